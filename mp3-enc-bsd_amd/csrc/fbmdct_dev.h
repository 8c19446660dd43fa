// Device-side building blocks shared by k_fbmdct.hip (batched) and k_dropin.hip (the
// reference's per-call surface): LDS layout, the matrixing step of filter_subband
// (src/encode.c:393-408), the 18-slot filterbank of one granule and the MDCT of one granule
// (src/mdct.c:57-91, 105-511).
//
// Bit-exactness: all arithmetic is f64 with the reference's association order -- y[i] sums
// its 8 taps left to right, each subband sample accumulates its 31 products in table order
// starting from y[16], the long-block MDCT follows the flattened term/operand order of
// src/mdct.c:205-508 (mdct_prog); nothing may be contracted to FMA (-ffp-contract=off).
#ifndef MP3MI_FBMDCT_DEV_H
#define MP3MI_FBMDCT_DEV_H
#include "mp3mi_host.h"

#define FBM_GPB 4

struct fbm_lds {
    double y[2][64];
    double sb[2][18][32];  // [0] previous granule, [1] current granule (sign-flipped like mdct_sub does)
    double xr[576];
    double cos_l[18][36];
    double win[4][36];
    double cos_s[6][12];
    uint16_t prog[18][36];
    int16_t pcm[1056 + 32];
};

// s[sub] of filter_subband from the 64 folded window sums y (src/encode.c:398-408)
MP3MI_DEVFN double fbm_matrix(const double *y, const double *frow)
{
    double si = y[16];
    for (int j = 0; j < 16; j++) si = si + frow[j] * (y[j] + y[32 - j]);
    for (int j = 0; j < 15; j++) si = si + frow[16 + j] * (y[33 + j] - y[63 - j]);
    return si;
}

MP3MI_DEVFN void fbm_load_tables(fbm_lds &L, const mp3mi_tables *T)
{
    const int lane = wave_lane();
    for (int i = lane; i < 18 * 36; i += 64) {
        L.cos_l[i / 36][i % 36] = T->cos_l[i / 36][i % 36];
        L.prog[i / 36][i % 36] = T->mdct_prog[i / 36][i % 36];
    }
    for (int i = lane; i < 4 * 36; i += 64) L.win[i / 36][i % 36] = T->mdct_win[i / 36][i % 36];
    for (int i = lane; i < 72; i += 64) L.cos_s[i / 12][i % 12] = T->cos_s[i / 12][i % 12];
}

// 18 slots of one granule -> sb[dst]; pcm in LDS holds samples [576*g - 480, 576*g + 576)
MP3MI_DEVFN void fbm_filter_granule(fbm_lds &L, int dst, const double *enw, const double *frow)
{
    const int lane = wave_lane(), half = lane >> 5, sub = lane & 31;
    for (int pair = 0; pair < 9; pair++) {
        // y[i] = sum_k z[i+64k], z[i] = pcm[32q+31-i]/32768 * enwindow[i]   (src/encode.c:306-312, 393-397)
        for (int h = 0; h < 2; h++) {
            int slot = pair * 2 + h;
            int base = 480 + 32 * slot + 31 - lane; // index into L.pcm of tap 0 for y[lane]
            double acc = ((double) L.pcm[base] * (1.0 / 32768.0)) * enw[0];
            for (int k = 1; k < 8; k++) acc = acc + ((double) L.pcm[base - 64 * k] * (1.0 / 32768.0)) * enw[k];
            L.y[h][lane] = acc;
        }
        __syncthreads();
        {
            double si = fbm_matrix(L.y[half], frow);
            int slot = pair * 2 + half;
            // mdct_sub negates odd slots of odd subbands before use (src/mdct.c:57-60)
            if ((sub & 1) && (slot & 1)) si = si * -1.0;
            L.sb[dst][slot][sub] = si;
        }
        __syncthreads();
    }
}

MP3MI_DEVFN void fbm_load_pcm(fbm_lds &L, const int16_t *pcm, long n_per_ch, int channels, int ch, long g)
{
    // samples [576 g - 480, 576 g + 576) of this channel; outside the stream -> 0
    for (int i = wave_lane(); i < 1056; i += 64) {
        long t = 576 * g - 480 + i;
        L.pcm[i] = (t >= 0 && t < n_per_ch) ? pcm[t * channels + ch] : (int16_t) 0;
    }
}

// MDCT + alias reduction of one granule: L.sb[0] (previous) and L.sb[1] (current), both already
// sign-compensated, -> L.xr[band*18 + m].  Ends with a barrier.
MP3MI_DEVFN void fbm_mdct_granule(fbm_lds &L, const mp3mi_tables *T, int bt)
{
    const int lane = wave_lane();
    for (int o = lane; o < 576; o += 64) {
        const int band = o / 18, m = o % 18;
        double sum;
        if (bt == 2) { // three short transforms, out[3*mm + l]   (src/mdct.c:173-185)
            const int mm = m / 3, l = m % 3;
            sum = 0.0;
            for (int k = 0; k < 12; k++) {
                int idx = k + 6 * l + 6;
                double in = (idx < 18) ? L.sb[0][idx][band] : L.sb[1][idx - 18][band];
                sum = sum + (L.win[2][k] * in) * L.cos_s[mm][k];
            }
        } else if (bt != 0) { // start / stop windows, plain 36-term sum (src/mdct.c:188-198)
            sum = 0.0;
            for (int k = 0; k < 36; k++) {
                double in = (k < 18) ? L.sb[0][k][band] : L.sb[1][k - 18][band];
                sum = sum + (L.win[bt][k] * in) * L.cos_l[m][k];
            }
        } else { // long window, reference's grouped expression trees (src/mdct.c:199-509)
            double acc = 0.0;
            sum = 0.0;
            for (int e = 0; e < 36; e++) {
                const unsigned pe = L.prog[m][e];
                const int idx = (int) (pe & 63u);
                double in = (idx < 18) ? L.sb[0][idx][band] : L.sb[1][idx - 18][band];
                double fin = L.win[0][idx] * in;
                if (pe & 0x80u) acc = (pe & 0x40u) ? -fin : fin;
                else acc = (pe & 0x40u) ? acc - fin : acc + fin;
                if (pe & 0x100u) {
                    double c = L.cos_l[m][(pe >> 9) & 31u];
                    if (pe & 0x4000u) c = -c;
                    double p = acc * c;
                    sum = (pe & 0x8000u) ? p : sum + p;
                }
            }
        }
        L.xr[o] = sum;
    }
    __syncthreads();
    if (bt != 2) { // alias reduction butterflies (src/mdct.c:83-91)
        for (int i = lane; i < 31 * 8; i += 64) {
            const int band = i >> 3, k = i & 7;
            double up = L.xr[band * 18 + 17 - k], dn = L.xr[(band + 1) * 18 + k];
            double bu = up * T->cs[k] + dn * T->ca[k];
            double bd = dn * T->cs[k] - up * T->ca[k];
            L.xr[band * 18 + 17 - k] = bu;
            L.xr[(band + 1) * 18 + k] = bd;
        }
        __syncthreads();
    }
}

#endif
