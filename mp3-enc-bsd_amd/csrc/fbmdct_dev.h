// Device-side building blocks shared by k_fbmdct.hip (batched) and k_dropin.hip (the
// reference's per-call surface): LDS layout, the matrixing step of filter_subband
// (src/encode.c:393-408), the 18-slot filterbank of one granule and the MDCT of one granule
// (src/mdct.c:57-91, 105-511).
//
// Bit-exactness: all arithmetic is f64 with the reference's association order -- y[i] sums
// its 8 taps left to right, each subband sample accumulates its 31 products in table order
// starting from y[16], the long-block MDCT evaluates the reference's bracketed operand groups
// once per band (they recur in several outputs, up to an exact negation) and then accumulates
// each output's terms in the reference's order; nothing may be contracted to FMA
// (-ffp-contract=off).
#ifndef MP3MI_FBMDCT_DEV_H
#define MP3MI_FBMDCT_DEV_H
#include "mp3mi_host.h"

#define FBM_GPB 4

struct fbm_lds {
    double sb[2][18][32];  // [0] previous granule, [1] current granule (sign-flipped like mdct_sub does)
    union { // the filterbank's staging buffers and the MDCT's are never live at the same time
        struct {
            double y[2][64];
            int16_t pcm[1056 + 32];
        };
        struct {
            double xr[576];
            double V[32][27]; // long-block operand groups per band (26 used; odd stride spreads the banks)
        };
    };
    double win[4][36];
    double cos_s[6][12];
    double vcoef[18][18];
    uint8_t vidx[18][18], nterm[18];
    uint8_t g_ops[6][6], h_ops[2][18];
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
    for (int i = lane; i < 4 * 36; i += 64) L.win[i / 36][i % 36] = T->mdct_win[i / 36][i % 36];
    for (int i = lane; i < 72; i += 64) L.cos_s[i / 12][i % 12] = T->cos_s[i / 12][i % 12];
    for (int i = lane; i < 18 * 18; i += 64) {
        L.vcoef[i / 18][i % 18] = T->mdct_vcoef[i / 18][i % 18];
        L.vidx[i / 18][i % 18] = T->mdct_vidx[i / 18][i % 18];
    }
    if (lane < 18) L.nterm[lane] = T->mdct_nterm[lane];
    if (lane < 36) L.g_ops[lane / 6][lane % 6] = T->mdct_g_ops[lane / 6][lane % 6];
    if (lane < 36) L.h_ops[lane / 18][lane % 18] = T->mdct_h_ops[lane / 18][lane % 18];
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

// windowed input k of a band: win[0][k] * in[k], in = 18 previous then 18 current slots
MP3MI_DEVFN double fbm_fin(const fbm_lds &L, int band, int k)
{
    const double in = (k < 18) ? L.sb[0][k][band] : L.sb[1][k - 18][band];
    return L.win[0][k] * in;
}

// ordered signed sum of windowed inputs: ops[i] = index | 0x80 (subtract / negate)
MP3MI_DEVFN double fbm_group(const fbm_lds &L, int band, const uint8_t *ops, int n)
{
    double acc = fbm_fin(L, band, ops[0] & 0x3f);
    if (ops[0] & 0x80) acc = -acc;
    for (int i = 1; i < n; i++) {
        const double f = fbm_fin(L, band, ops[i] & 0x3f);
        acc = (ops[i] & 0x80) ? acc - f : acc + f;
    }
    return acc;
}

// MDCT + alias reduction of one granule: L.sb[0] (previous) and L.sb[1] (current), both already
// sign-compensated, -> L.xr[band*18 + m].  Ends with a barrier.
MP3MI_DEVFN void fbm_mdct_granule(fbm_lds &L, const mp3mi_tables *T, int bt)
{
    const int lane = wave_lane();
    if (bt == 0) { // long window (src/mdct.c:199-509)
        { // phase A: the 26 operand groups of every band; lane = band*2 + h, h picks the half of the list
            const int band = lane >> 1, h = lane & 1;
            for (int j = 0; j < 9; j++) {
                const double a = fbm_fin(L, band, h ? 18 + j : j), b = fbm_fin(L, band, h ? 35 - j : 17 - j);
                L.V[band][9 * h + j] = h ? a + b : a - b;
            }
            for (int c = 0; c < 3; c++) L.V[band][18 + 3 * h + c] = fbm_group(L, band, L.g_ops[3 * h + c], 6);
            L.V[band][24 + h] = fbm_group(L, band, L.h_ops[h], 18);
        }
        __syncthreads();
        if (lane < 54) { // phase B: lane owns output row m for every third band
            const int m = lane % 18, grp = lane / 18, nt = L.nterm[m];
            double coef[18];
            int vi[18];
#pragma unroll
            for (int t = 0; t < 18; t++) { coef[t] = L.vcoef[m][t]; vi[t] = L.vidx[m][t]; }
            for (int band = grp; band < 32; band += 3) {
                double sum = L.V[band][vi[0]] * coef[0];
#pragma unroll
                for (int t = 1; t < 18; t++) {
                    const double p = L.V[band][vi[t]] * coef[t];
                    sum = (t < nt) ? sum + p : sum;
                }
                L.xr[band * 18 + m] = sum;
            }
        }
    } else {
        for (int o = lane; o < 576; o += 64) {
            const int band = o / 18, m = o % 18;
            double sum = 0.0;
            if (bt == 2) { // three short transforms, out[3*mm + l]   (src/mdct.c:173-185)
                const int mm = m / 3, l = m % 3;
                for (int k = 0; k < 12; k++) {
                    int idx = k + 6 * l + 6;
                    double in = (idx < 18) ? L.sb[0][idx][band] : L.sb[1][idx - 18][band];
                    sum = sum + (L.win[2][k] * in) * L.cos_s[mm][k];
                }
            } else { // start / stop windows, plain 36-term sum (src/mdct.c:188-198)
                for (int k = 0; k < 36; k++) {
                    double in = (k < 18) ? L.sb[0][k][band] : L.sb[1][k - 18][band];
                    sum = sum + (L.win[bt][k] * in) * T->cos_l[m][k];
                }
            }
            L.xr[o] = sum;
        }
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
