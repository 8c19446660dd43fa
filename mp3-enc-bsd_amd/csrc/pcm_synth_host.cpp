/* Host side of the deterministic PCM generator (pcm_synth_core.h): mp3mi_synth_pcm for tests, goldens and the
 * CPU baseline sample, and the parameter block both sides start from.  Compiled by g++ (never hipcc: dmath.h
 * is host code only outside __HIPCC__) with -ffp-contract=off like everything else. */
#include "mp3mi.h"
#include "pcm_synth_core.h"

extern "C" void mp3mi_synth_params(int rate_hz, synth_params *P)
{
    const double two_pi = 6.283185307179586;
    const double T = 10.0, f0 = 20.0, f1 = 0.45 * rate_hz;
    P->lr = dm_log(f1 / f0);
    P->K = two_pi * f0 * T / P->lr;
    P->T = T;
    P->rate = (double) rate_hz;
    P->half = rate_hz / 2;
}

extern "C" void mp3mi_synth_pcm(int16_t *out, long n_per_ch, int channels, int rate_hz, uint32_t stream, uint32_t seed)
{
    synth_params P;
    mp3mi_synth_params(rate_hz, &P);
    for (long n = 0; n < n_per_ch; n++) synth_sample(&P, seed, stream, channels, n, out + n * channels);
}
