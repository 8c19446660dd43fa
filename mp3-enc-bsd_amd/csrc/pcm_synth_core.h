/* Deterministic synthetic PCM for benchmarks and parity tests, ONE definition for host and
 * device: per stream a log sine sweep 20 Hz -> 0.45 fs over 10 s (right channel 1 % sharp), counter-based
 * uniform noise, and a 300-sample burst every half second so that the psychoacoustic model switches to short
 * blocks.  The transcendental functions are dmath.h's (IEEE + - * / fma only: the device and the host build
 * agree bit for bit, tests/test_gpu_dmath.py), so mp3mi_synth_pcm (host, pcm_synth_host.cpp) and
 * mp3mi_synth_pcm_device (k_synth.hip) produce the same bytes; tests/test_synth.py pins their md5.
 */
#ifndef MP3MI_PCM_SYNTH_CORE_H
#define MP3MI_PCM_SYNTH_CORE_H
#include <stdint.h>
#include "dmath.h"

typedef struct {
    double K;      /* 2 pi f0 T / ln(f1 / f0) */
    double lr;     /* ln(f1 / f0) */
    double T;
    double rate;
    int32_t half;  /* burst period in samples */
} synth_params;

DM_FN uint32_t synth_mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

DM_FN uint32_t synth_key32(uint32_t seed, uint32_t stream, uint32_t ch, uint32_t n)
{
    return synth_mix32(synth_mix32(synth_mix32(seed ^ 0x9e3779b9u * (stream + 1)) + ch * 0x85ebca6bu) + n);
}

/* sample n of every channel of `stream` -> out[0 .. channels) */
DM_FN void synth_sample(const synth_params *P, uint32_t seed, uint32_t stream, int channels, long n, int16_t *out)
{
    const double amp = 32767.0 * (0.15 + 0.25 * (double) ((stream * 37u) % 16u) / 15.0);
    const uint32_t nsel = (stream / 3u) % 4u;
    const double namp = nsel == 0 ? 1386.0 : (nsel == 1 ? 90.0 : (nsel == 2 ? 350.0 : 5200.0));
    const long boff = (long) ((stream * 977u) % (uint32_t) P->half);
    const double t = (double) n / P->rate;
    const double ph = P->K * (dm_exp(P->lr * t / P->T) - 1.0);
    const long bpos = (n + boff) % P->half;
    for (int c = 0; c < channels; c++) {
        const uint32_t h = synth_key32(seed, stream, (uint32_t) c, (uint32_t) n);
        double v = amp * dm_sin((c ? 1.01 : 1.0) * ph + 0.3 * (double) stream);
        v += namp * ((double) (h >> 8) / 8388608.0 - 1.0);
        if (bpos < 300) v += (h & 1u) ? 12000.0 : -12000.0;
        v = __builtin_floor(v + 0.5);
        if (v > 32767.0) v = 32767.0;
        if (v < -32768.0) v = -32768.0;
        out[c] = (int16_t) v;
    }
}

#endif
