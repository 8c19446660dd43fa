// Polyphase analysis filterbank + MDCT + alias reduction.
//
// Replaces, for a whole batch, window_subband/filter_subband (src/encode.c:287-409) and
// mdct_sub/mdct (src/mdct.c:25-511).  One wavefront handles FBM_GPB consecutive granules of
// one (stream, channel): it recomputes the 18 subband slots of the granule before its first
// one (the reference keeps them in l3_sb_sample[ch][0]; the filterbank is feed-forward, so
// they are a pure function of the PCM) and then walks forward, keeping the previous
// granule's slots in LDS.  Arithmetic and its ordering: fbmdct_dev.h.
//
// HBM traffic per granule-channel: 1152 B of PCM in (+ the 480-sample tail shared with the
// neighbour, L2-resident) and 4608 B of xr out; everything else lives in LDS/registers.
#include "fbmdct_dev.h"

__global__ void __launch_bounds__(64) k_fbmdct(const mp3mi_tables *__restrict__ T, mp3mi_geom geo,
                                               const int16_t *__restrict__ pcm_all,
                                               const mp3mi_psy_out *__restrict__ psy,
                                               double *__restrict__ xr_out, double *__restrict__ sb_dbg)
{
    __shared__ fbm_lds L;
    const int lane = wave_lane();
    const int C = geo.channels, G = geo.n_gran;
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
    fbm_load_tables(L, T);

    const int gl0 = gb * FBM_GPB;
    const long gabs0 = (long) geo.g0 + gl0;

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
        if (sb_dbg) { // raw subband samples as filter_subband returns them (parity tests)
            for (int i = lane; i < 576; i += 64) {
                int slot = i / 32, sub = i % 32;
                double v = L.sb[1][slot][sub];
                if ((sub & 1) && (slot & 1)) v = v * -1.0;
                sb_dbg[rec * 576 + i] = v;
            }
        }
        fbm_mdct_granule(L, T, bt);
        for (int i = lane; i < 576; i += 64) xr_out[rec * 576 + i] = L.xr[i];
        // current becomes previous (src/mdct.c:99-102)
        for (int i = lane; i < 576; i += 64) L.sb[0][i / 32][i % 32] = L.sb[1][i / 32][i % 32];
        __syncthreads();
    }
}

void mp3mi_launch_fbmdct(const mp3mi_tables *T, const mp3mi_geom &g, const int16_t *pcm,
                         const mp3mi_psy_out *psy, double *xr, double *sb_dbg, hipStream_t st)
{
    const int blocks_per_sc = (g.n_gran + FBM_GPB - 1) / FBM_GPB;
    const unsigned grid = (unsigned) (g.n_streams * g.channels * blocks_per_sc);
    hipLaunchKernelGGL(k_fbmdct, dim3(grid), dim3(64), 0, st, T, g, pcm, psy, xr, sb_dbg);
}
