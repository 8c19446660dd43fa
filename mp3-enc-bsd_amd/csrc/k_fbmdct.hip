// Polyphase analysis filterbank + MDCT + alias reduction.
//
// Replaces, for a whole batch, window_subband/filter_subband (src/encode.c:287-409) and
// mdct_sub/mdct (src/mdct.c:25-511).  One wavefront handles FBM_GPB consecutive granules of
// one (stream, channel): it recomputes the 18 subband slots of the granule before its first
// one (the reference keeps them in l3_sb_sample[ch][0]; the filterbank is feed-forward, so
// they are a pure function of the PCM) and then walks forward, keeping the previous
// granule's slots in LDS.
//
// Bit-exactness notes: all arithmetic is f64 with the reference's association order --
// y[i] sums its 8 taps left to right, each subband sample accumulates its 31 products in
// table order starting from y[16], the long-block MDCT follows the flattened term/operand
// order of src/mdct.c:205-508 (mdct_prog), and nothing may be contracted to FMA
// (-ffp-contract=off).
//
// HBM traffic per granule-channel: 1152 B of PCM in (+ the 480-sample tail shared with the
// neighbour, L2-resident) and 4608 B of xr out; everything else lives in LDS/registers.
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
            const double *y = L.y[half];
            double si = y[16];
            for (int j = 0; j < 16; j++) si = si + frow[j] * (y[j] + y[32 - j]);
            for (int j = 0; j < 15; j++) si = si + frow[16 + j] * (y[33 + j] - y[63 - j]);
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

__global__ void __launch_bounds__(64) k_fbmdct(const mp3mi_tables *__restrict__ T, mp3mi_geom geo,
                                               const int16_t *__restrict__ pcm_all,
                                               const mp3mi_psy_out *__restrict__ psy,
                                               double *__restrict__ xr_out, double *__restrict__ sb_dbg)
{
    __shared__ fbm_lds L;
    const int lane = wave_lane();
    const int C = geo.channels, G = 2 * geo.nf;
    const int blocks_per_sc = (G + FBM_GPB - 1) / FBM_GPB;
    int bid = (int) blockIdx.x;
    const int gb = bid % blocks_per_sc; bid /= blocks_per_sc;
    const int ch = bid % C;
    const int s = bid / C;
    const long n_per_ch = (long) geo.n_frames * 1152;
    const int16_t *pcm = pcm_all + (size_t) s * (size_t) n_per_ch * (size_t) C;

    // per-lane constants: 8 window taps for y[lane], the 31 filter coefficients of subband lane&31
    double enw[8], frow[31];
    for (int k = 0; k < 8; k++) enw[k] = T->enwindow[lane + 64 * k];
    for (int j = 0; j < 31; j++) frow[j] = T->filt[lane & 31][j];
    for (int i = lane; i < 18 * 36; i += 64) {
        L.cos_l[i / 36][i % 36] = T->cos_l[i / 36][i % 36];
        L.prog[i / 36][i % 36] = T->mdct_prog[i / 36][i % 36];
    }
    for (int i = lane; i < 4 * 36; i += 64) L.win[i / 36][i % 36] = T->mdct_win[i / 36][i % 36];
    for (int i = lane; i < 72; i += 64) L.cos_s[i / 12][i % 12] = T->cos_s[i / 12][i % 12];

    const int gl0 = gb * FBM_GPB;
    const long gabs0 = 2L * geo.f0 + gl0;

    // previous granule's slots
    fbm_load_pcm(L, pcm, n_per_ch, C, ch, gabs0 - 1);
    __syncthreads();
    if (gabs0 == 0) {
        for (int i = lane; i < 576; i += 64) L.sb[0][i / 32][i % 32] = 0.0;
        __syncthreads();
    } else
        fbm_filter_granule(L, 0, enw, frow);

    for (int gi = 0; gi < FBM_GPB && gl0 + gi < G; gi++) {
        const int gl = gl0 + gi;
        const long gabs = gabs0 + gi;
        const size_t rec = ((size_t) s * G + gl) * C + ch;
        const int bt = psy[rec].block_type;
        fbm_load_pcm(L, pcm, n_per_ch, C, ch, gabs);
        __syncthreads();
        fbm_filter_granule(L, 1, enw, frow);
        if (sb_dbg) { // raw subband samples as filter_subband returns them (debug / per-call shim)
            for (int i = lane; i < 576; i += 64) {
                int slot = i / 32, sub = i % 32;
                double v = L.sb[1][slot][sub];
                if ((sub & 1) && (slot & 1)) v = v * -1.0;
                sb_dbg[rec * 576 + i] = v;
            }
        }
        // ---- MDCT: output o = band*18 + m ----
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
        for (int i = lane; i < 576; i += 64) xr_out[rec * 576 + i] = L.xr[i];
        // current becomes previous (src/mdct.c:99-102)
        for (int i = lane; i < 576; i += 64) L.sb[0][i / 32][i % 32] = L.sb[1][i / 32][i % 32];
        __syncthreads();
    }
}

void mp3mi_launch_fbmdct(const mp3mi_tables *T, const mp3mi_geom &g, const int16_t *pcm,
                         const mp3mi_psy_out *psy, double *xr, double *sb_dbg, hipStream_t st)
{
    const int G = 2 * g.nf;
    const int blocks_per_sc = (G + FBM_GPB - 1) / FBM_GPB;
    const unsigned grid = (unsigned) (g.n_streams * g.channels * blocks_per_sc);
    hipLaunchKernelGGL(k_fbmdct, dim3(grid), dim3(64), 0, st, T, g, pcm, psy, xr, sb_dbg);
}
