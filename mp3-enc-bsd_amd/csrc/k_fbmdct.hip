// Polyphase analysis filterbank, then MDCT + alias reduction.
//
// Replaces, for a whole batch, window_subband/filter_subband (src/encode.c:287-409) and
// mdct_sub/mdct (src/mdct.c:25-511).  Two kernels with the subband samples of the chunk in HBM
// between them (4.6 KB per granule and channel, written once and read twice):
//
//   k_filter  one wavefront per (stream, granule, channel): the 18 slots of 32 subband samples.
//             The filterbank is feed-forward -- a pure function of 1056 PCM samples -- so every
//             granule is independent, including the one BEFORE the chunk, which the reference
//             would still hold in l3_sb_sample[ch][0] and which is recomputed here (granule slot 0).
//   k_mdct    one wavefront per (stream, channel, run of 22 granules): 36 inputs per band from two granules.
//
// Arithmetic and its ordering: fbmdct_dev.h.
#include "fbmdct_dev.h"

struct filter_lds {
    int16_t pcm[1056 + 32];
    double ud[2][2][32]; // [parity of the slot pair][slot of the pair]: y[16], y[j]+y[32-j] (j<16), y[33+j]-y[63-j] (j<15)
};

__global__ void __launch_bounds__(64, 3) k_filter(const mp3mi_tables *__restrict__ T, mp3mi_geom geo,
                                                  const int16_t *__restrict__ pcm_all, double *__restrict__ sbs,
                                                  double *__restrict__ sb_dbg)
{
    __shared__ filter_lds L;
    const int lane = wave_lane(), half = lane >> 5, sub = lane & 31;
    const int C = geo.channels, G1 = geo.n_gran + 1;
    int bid = (int) blockIdx.x;
    const int ch = bid % C; bid /= C;
    const int gi = bid % G1;   // granule slot: 0 is the granule before the chunk
    const int s = bid / G1;
    const long gabs = (long) geo.g0 - 1 + gi; // relative to the call's first granule
    double *out = sbs + (((size_t) s * G1 + gi) * C + ch) * 576;
    if (2 * geo.fabs0 + gabs < 0) { // before the stream: the reference's zero-initialised l3_sb_sample
        for (int i = lane; i < 576; i += 64) out[i] = 0.0;
        return;
    }
    const long n_pitch = (long) geo.n_frames * 1152; // row pitch of the PCM buffer
    const long n_per_ch = geo.n_samples ? (long) geo.n_samples[s] : n_pitch; // valid samples: the rest reads as zero (src/encode.c:162-166)
    const int16_t *pcm = pcm_all + (size_t) s * (size_t) n_pitch * (size_t) C;
    const int16_t *hist = geo.hist ? geo.hist + (size_t) s * MP3MI_PCM_HIST * (size_t) C : NULL;
    // samples [576 g - 480, 576 g + 576) of this channel; before the call's first sample the stream's history
    // (zeros at the start of a stream), beyond the stream's last sample 0
    const long t0 = 576 * gabs - 480;
    if (t0 >= 0 && t0 + 1056 <= n_per_ch) {
        // all 1056 samples lie inside the call's PCM (every granule but the first and the last few of a stream): one
        // scalar base, 32-bit lane offsets, no per-sample range tests (they cost ~30 instructions per sample)
        const int16_t *p0 = pcm + t0 * C + ch;
        const unsigned lo = (unsigned) lane * (unsigned) C, step = 64u * (unsigned) C;
        int16_t v[17];
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = p0[lo + step * (unsigned) k];
        v[16] = lane < 32 ? p0[lo + step * 16u] : (int16_t) 0;
#pragma unroll
        for (int k = 0; k < 16; k++) L.pcm[lane + 64 * k] = v[k];
        if (lane < 32) L.pcm[lane + 1024] = v[16];
    } else {
        int16_t v[17];
#pragma unroll
        for (int k = 0; k < 17; k++) {
            const int i = lane + 64 * k;
            const long t = t0 + i;
            const bool past = hist && t < 0 && t >= -MP3MI_PCM_HIST;
            v[k] = (i < 1056 && t >= 0 && t < n_per_ch) ? pcm[t * C + ch] : ((i < 1056 && past) ? hist[(t + MP3MI_PCM_HIST) * C + ch] : (int16_t) 0);
        }
#pragma unroll
        for (int k = 0; k < 17; k++)
            if (lane + 64 * k < 1056) L.pcm[lane + 64 * k] = v[k];
    }
    // per-lane constants: 8 window taps for y[lane] -- with the 1/32768 of src/encode.c:306-312 folded in: scaling
    // by a power of two commutes with the rounding of the product, (x / 32768) * w == x * (w / 32768) bit for
    // bit (no product comes near the subnormals) -- and the 31 filter coefficients of subband lane & 31
    double enw[8], frow[31];
#pragma unroll
    for (int k = 0; k < 8; k++) enw[k] = T->enwindow[lane + 64 * k] * (1.0 / 32768.0);
#pragma unroll
    for (int j = 0; j < 31; j++) frow[j] = T->filt[sub][j];
    // where this lane's y goes in the matrixing step: lane j < 16 forms y[j] + y[32-j], lane 16 passes
    // y[16] on, lane 33+j (j < 15) forms y[33+j] - y[63-j]; the other lanes only supply operands
    const int partner = (lane < 16) ? 32 - lane : ((lane >= 33 && lane < 48) ? 96 - lane : lane);
    const int udi = (lane < 16) ? 1 + lane : (lane == 16 ? 0 : ((lane >= 33 && lane < 48) ? lane - 16 : -1));
    __syncthreads();

    for (int pair = 0; pair < 9; pair++) {
        // y[i] = sum_k z[i+64k], z[i] = pcm[32q+31-i]/32768 * enwindow[i]   (src/encode.c:306-312, 393-397)
        double y[2];
        {
            int16_t tp[2][8];
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int k = 0; k < 8; k++) tp[h][k] = L.pcm[480 + 32 * (pair * 2 + h) + 31 - lane - 64 * k];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                double acc = (double) tp[h][0] * enw[0];
#pragma unroll
                for (int k = 1; k < 8; k++) acc = acc + (double) tp[h][k] * enw[k];
                y[h] = acc;
            }
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const double yp = __shfl(y[h], partner);
            const double v = (lane < 16) ? y[h] + yp : (lane == 16 ? y[h] : y[h] - yp);
            if (udi >= 0) L.ud[pair & 1][h][udi] = v;
        }
        __syncthreads();
        // s[sub] = y[16] + sum_j filt[j] (y[j] + y[32-j]) + sum_j filt[16+j] (y[33+j] - y[63-j])   (src/encode.c:398-408)
        {
            double u[32];
#pragma unroll
            for (int j = 0; j < 32; j++) u[j] = L.ud[pair & 1][half][j];
            double si = u[0];
#pragma unroll
            for (int j = 0; j < 31; j++) si = si + frow[j] * u[1 + j];
            const int slot = pair * 2 + half;
            if (sb_dbg && gi > 0) // raw subband samples as filter_subband returns them (parity tests)
                sb_dbg[(((size_t) s * geo.n_gran + gi - 1) * C + ch) * 576 + slot * 32 + sub] = si;
            // mdct_sub negates odd slots of odd subbands before use (src/mdct.c:57-60)
            if ((sub & 1) && (slot & 1)) si = si * -1.0;
            out[slot * 32 + sub] = si;
        }
        // the next pair writes the other ud buffer; the barrier above orders its reuse two pairs on
    }
}

// One wavefront transforms a run of MDCT_RUN consecutive granules of one (stream, channel): the tables are
// set up once, every granule's subband samples are read once (the current granule is the next one's
// "previous"), and the next granule's samples are requested before the current one is transformed -- into the
// registers the previous granule's just left.  168 VGPRs (no spills) and 15 KB of LDS: the kernel fits beside k_loop's
// resident wavefronts (batch.cpp).
#define MDCT_RUN 22
__global__ void __launch_bounds__(64, 3) k_mdct(const mp3mi_tables *__restrict__ T, mp3mi_geom geo,
                                                const double *__restrict__ sbs, const mp3mi_psy_out *__restrict__ psy,
                                                double *__restrict__ xr_out)
{
    __shared__ mdct_lds L;
    const int lane = wave_lane();
    const int C = geo.channels, G = geo.n_gran, NR = (G + MDCT_RUN - 1) / MDCT_RUN;
    int bid = (int) blockIdx.x;
    const int ch = bid % C; bid /= C;
    const int r = bid % NR;
    const int s = bid / NR;
    const int g_lo = r * MDCT_RUN, n = G - g_lo < MDCT_RUN ? G - g_lo : MDCT_RUN;
    const size_t rec0 = ((size_t) s * G + g_lo) * C + ch;
    const int btv = lane < n ? psy[rec0 + (size_t) lane * C].block_type : 0;
    const size_t pitch = (size_t) C * 576;
    const double *blk = sbs + (((size_t) s * (G + 1) + g_lo) * C + ch) * 576; // granule slot g_lo: the one before granule g_lo
    double vp[9], vc[9];
#pragma unroll
    for (int j = 0; j < 9; j++) { vp[j] = blk[lane + 64 * j]; vc[j] = blk[pitch + lane + 64 * j]; }
    mdct_regs R;
    mdct_load_tables(L, R, T);
    __syncthreads();
    for (int k = 0; k < n; k++) {
        const int bt = wave_readlane_i32(btv, k);
        mdct_store_inputs(L, vp, vc, bt);
#pragma unroll
        for (int j = 0; j < 9; j++) vp[j] = vc[j]; // the current granule is the next one's previous,
        if (k + 1 < n) {                            // and the next one's samples are on their way during the transform
#pragma unroll
            for (int j = 0; j < 9; j++) vc[j] = blk[(size_t) (k + 2) * pitch + lane + 64 * j];
        }
        mdct_granule(L, R, T, bt);
        double *out = xr_out + (rec0 + (size_t) k * C) * 576;
#pragma unroll
        for (int j = 0; j < 9; j++) out[lane + 64 * j] = L.xr[lane + 64 * j];
        __syncthreads(); // the next granule's inputs take the place of this result
    }
}

size_t mp3mi_sbs_bytes(const mp3mi_geom &g) { return (size_t) g.n_streams * (size_t) (g.n_gran + 1) * (size_t) g.channels * 576 * sizeof(double); }

void mp3mi_launch_fbmdct(const mp3mi_tables *T, const mp3mi_geom &g, const int16_t *pcm,
                         const mp3mi_psy_out *psy, double *sbs, double *xr, double *sb_dbg, hipStream_t st)
{
    hipLaunchKernelGGL(k_filter, dim3((unsigned) (g.n_streams * (g.n_gran + 1) * g.channels)), dim3(64), 0, st, T, g, pcm, sbs, sb_dbg);
    hipLaunchKernelGGL(k_mdct, dim3((unsigned) (g.n_streams * g.channels * ((g.n_gran + MDCT_RUN - 1) / MDCT_RUN))), dim3(64), 0, st, T, g, sbs, psy, xr);
}
