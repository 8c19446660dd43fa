/* Deterministic synthetic PCM for benchmarks and parity tests:
 * per stream a log sine sweep 20 Hz -> 0.45 fs over 10 s (right channel 1 % sharp),
 * counter-based uniform noise, and a 300-sample burst every half second so that
 * the psychoacoustic model switches to short blocks.  Host-only, plain C; the
 * same bytes feed the GPU path, the CPU oracle and (as WAV) the reference.
 */
#include <math.h>
#include <stdint.h>
#include "mp3mi.h"

static uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

static uint32_t key32(uint32_t seed, uint32_t stream, uint32_t ch, uint32_t n)
{
    return mix32(mix32(mix32(seed ^ 0x9e3779b9u * (stream + 1)) + ch * 0x85ebca6bu) + n);
}

void mp3mi_synth_pcm(int16_t *out, long n_per_ch, int channels, int rate_hz,
                     uint32_t stream, uint32_t seed)
{
    const double two_pi = 6.283185307179586;
    const double T = 10.0, f0 = 20.0, f1 = 0.45 * rate_hz;
    const double lr = log(f1 / f0);
    const double amp = 32767.0 * (0.15 + 0.25 * (double) ((stream * 37u) % 16u) / 15.0);
    static const double noise_amp[4] = { 1386.0, 90.0, 350.0, 5200.0 };
    const double namp = noise_amp[(stream / 3u) % 4u];
    const long half = rate_hz / 2;
    const long boff = (long) ((stream * 977u) % (uint32_t) half);
    long n;
    int c;
    for (n = 0; n < n_per_ch; n++) {
        double t = (double) n / rate_hz;
        double ph = two_pi * f0 * T / lr * (exp(lr * t / T) - 1.0);
        long bpos = (n + boff) % half;
        for (c = 0; c < channels; c++) {
            uint32_t h = key32(seed, stream, (uint32_t) c, (uint32_t) n);
            double v = amp * sin((c ? 1.01 : 1.0) * ph + 0.3 * (double) stream);
            v += namp * ((double) (h >> 8) / 8388608.0 - 1.0);
            if (bpos < 300)
                v += (h & 1u) ? 12000.0 : -12000.0;
            v = floor(v + 0.5);
            if (v > 32767.0) v = 32767.0;
            if (v < -32768.0) v = -32768.0;
            out[n * channels + c] = (int16_t) v;
        }
    }
}
