// Small kernels behind the reference's per-call symbols (include/mp3mi_dropin.h) that have no
// batched counterpart of the same granularity: window_subband, filter_subband, mdct_sub.  They
// reuse the device functions of the batched path (fbmdct_dev.h), so the arithmetic is the same
// code.  L3psycho_anal, iteration_loop and III_format_bitstream reuse k_fft/k_psy, k_loop and
// k_format directly with n_streams = 1 (dropin.cpp).
#include "fbmdct_dev.h"

// src/encode.c:361-409
__global__ void __launch_bounds__(64) k_filter_subband(const mp3mi_tables *__restrict__ T, const double *__restrict__ z,
                                                       double *__restrict__ s)
{
    __shared__ double y[64];
    const int lane = wave_lane();
    double acc = z[lane];
    for (int k = 1; k < 8; k++) acc = acc + z[lane + 64 * k];
    y[lane] = acc;
    __syncthreads();
    if (lane < 32) {
        double frow[31];
        for (int j = 0; j < 31; j++) frow[j] = T->filt[lane][j];
        s[lane] = fbm_matrix(y, frow);
    }
}

// window_subband (src/encode.c:287-316: ring[ch] is the 512-entry history x[k][], off the running offset) and the
// filter_subband that the reference's frame loops call right after it (src/musicin.c:640-642, 676-678, 722-724) in ONE launch: the 32 new samples arrive as a kernel argument, z[512] and s[32] go to zs -- a
// host-mapped buffer -- so that the call is one launch and one wait instead of three copies, two launches and two waits.
// filter_subband's arithmetic is that of the kernel above, operation by operation.
__global__ void __launch_bounds__(64) k_window_filter(const mp3mi_tables *__restrict__ T, double *__restrict__ ring, int off,
                                                      mp3mi_dropin_samples in, double *__restrict__ zs)
{
    __shared__ double zl[512];
    __shared__ double y[64];
    const int lane = wave_lane();
    if (lane < 32) ring[(31 - lane + off) & 511] = (double) in.v[lane] * (1.0 / 32768.0);
    __syncthreads();
    for (int i = lane; i < 512; i += 64) {
        const double v = ring[(i + off) & 511] * T->enwindow[i];
        zl[i] = v;
        zs[i] = v;
    }
    __syncthreads();
    double acc = zl[lane];
    for (int k = 1; k < 8; k++) acc = acc + zl[lane + 64 * k];
    y[lane] = acc;
    __syncthreads();
    if (lane < 32) {
        double frow[31];
        for (int j = 0; j < 31; j++) frow[j] = T->filt[lane][j];
        zs[512 + lane] = fbm_matrix(y, frow);
    }
}

// A whole FRAME of window_subband + filter_subband calls of up to two channels in one launch (dropin.cpp: look-ahead).
// Workgroup (ch, i) computes slot i of channel ch: its 512-sample window holds the frame's samples 32 (i - m) + j at
// z index 32 m + 31 - j (m < 16, j < 32) -- what the reference's ring holds after it has taken in slots 0..i
// (src/encode.c:287-316) --; samples before the frame are read out of the ring as it stood at the frame's start (ring,
// off0: neither is written here), where sample 32 q + j (q < 0) sits at (31 - j + off0 - 32 q) & 511.  z[512] and s[32]
// go to zs[ch][i][544].  The arithmetic per slot is k_window_filter's, operation by operation.
// sb_out (may be NULL): the caller's L3SBS as the reference's loop will have filled it by the time it calls mdct_sub --
// s of slot i goes to [ch][1 + i / 18][i % 18][32] as well -- for the mdct_sub that dropin.cpp launches right behind this kernel.
__global__ void __launch_bounds__(64) k_window_filter_frame(const mp3mi_tables *__restrict__ T, const double *__restrict__ ring_all,
                                                            int off0_a, int off0_b, const int16_t *__restrict__ samples, int n_slots,
                                                            double *__restrict__ zs, double *__restrict__ sb_out)
{
    __shared__ double zl[512];
    __shared__ double y[64];
    const int lane = wave_lane(), ch = (int) blockIdx.x / n_slots, i = (int) blockIdx.x % n_slots;
    const double *ring = ring_all + 512 * ch;
    const int16_t *smp = samples + (size_t) ch * 32 * n_slots;
    const int off0 = ch ? off0_b : off0_a;
    double *out = zs + ((size_t) ch * n_slots + i) * 544;
    for (int t = lane; t < 512; t += 64) {
        const int m = t >> 5, j = 31 - (t & 31), q = i - m;
        const double x = q >= 0 ? (double) smp[32 * q + j] * (1.0 / 32768.0) : ring[(31 - j + off0 - 32 * q) & 511];
        const double v = x * T->enwindow[t];
        zl[t] = v;
        out[t] = v;
    }
    __syncthreads();
    double acc = zl[lane];
    for (int k = 1; k < 8; k++) acc = acc + zl[lane + 64 * k];
    y[lane] = acc;
    __syncthreads();
    if (lane < 32) {
        double frow[31];
        for (int j = 0; j < 31; j++) frow[j] = T->filt[lane][j];
        const double sv = fbm_matrix(y, frow);
        out[512 + lane] = sv;
        if (sb_out) sb_out[(size_t) ch * 3 * 576 + (size_t) (1 + i / 18) * 576 + (size_t) (i % 18) * 32 + lane] = sv;
    }
}

// The end of a call's launches, as the host sees it: the last "kernel" of the sequence stores the call's number in
// host-mapped memory and the host spins on it -- a few microseconds after the store instead of the ~25 us a
// hipStreamSynchronize takes to come back (dropin.cpp, dropin_wait: four of them per frame).
__global__ void k_dropin_done(volatile unsigned *flag, unsigned seq)
{
    __threadfence_system(); // (the stream is in order: every kernel of the call has finished; their stores to host-mapped memory first)
    *flag = seq;
}

void mp3mi_launch_dropin_done(unsigned *flag, unsigned seq, hipStream_t st)
{
    hipLaunchKernelGGL(k_dropin_done, dim3(1), dim3(1), 0, st, (volatile unsigned *) flag, seq);
}

void mp3mi_launch_window_filter_frame(const mp3mi_tables *T, const double *ring, int off0_a, int off0_b, const int16_t *samples, int n_ch, int n_slots,
                                      double *zs, double *sb_out, hipStream_t st)
{
    hipLaunchKernelGGL(k_window_filter_frame, dim3((unsigned) (n_ch * n_slots)), dim3(64), 0, st, T, ring, off0_a, off0_b, samples, n_slots, zs, sb_out);
}

// src/mdct.c:25-103: sb_in is the caller's L3SBS [2][3][18][32] as the call finds it, sb_out the same as the call leaves it;
// bt[gr][2]; xr [gr][ch][576] with the call's number of channels (the layout k_loop reads).  One workgroup per (channel, granule): the reference transforms granule 0 from blocks 0 and 1,
// granule 1 from blocks 1 and 2 -- after negating the odd slots of the odd subbands of blocks 1 and 2 in place (src/mdct.c:57-60)
// -- and ends by copying block 2 over block 0 (src/mdct.c:98-103).  With the result in a buffer of its own the two granules of
// a channel do not wait for each other: each reads its two blocks from sb_in (negating 1 and 2 as it reads) and writes
// what the reference leaves of them.  flag != NULL: the last workgroup to finish tells the spinning host (dropin.cpp) --
// count: zero before the launch, zero again after it.
__global__ void __launch_bounds__(64) k_mdct_sub(const mp3mi_tables *__restrict__ T, const double *__restrict__ sb_in, double *__restrict__ sb_out,
                                                 const int32_t *__restrict__ bt, double *__restrict__ xr, double *__restrict__ xr_dev, int mode_gr,
                                                 unsigned *zero_me, volatile unsigned *flag, unsigned seq, unsigned *count)
{
    __shared__ mdct_lds L;
    const int lane = wave_lane(), ch = (int) blockIdx.x / mode_gr, gr = (int) blockIdx.x % mode_gr;
    if (zero_me && blockIdx.x == 0 && lane == 0) *zero_me = 0u; // (the list of the kernels behind this one: a memset less on the stream)
    const double *in = sb_in + (size_t) ch * 3 * 576;
    double *outc = sb_out + (size_t) ch * 3 * 576;
    mdct_regs R;
    mdct_load_tables(L, R, T);
    __syncthreads();
    double vp[9], vc[9];
#pragma unroll
    for (int j = 0; j < 9; j++) {
        const int i = lane + 64 * j, slot = i / 32, sub = i % 32;
        const bool neg = (sub & 1) && (slot & 1);
        const double p = in[gr * 576 + i], c = in[(gr + 1) * 576 + i];
        vp[j] = (neg && gr > 0) ? p * -1.0 : p; // (block 0 is the frame before's block 2: negated then)
        vc[j] = neg ? c * -1.0 : c;
        outc[(gr + 1) * 576 + i] = vc[j];
        if (gr == mode_gr - 1) outc[i] = vc[j];
    }
    const int b = bt[gr * 2 + ch];
    mdct_store_inputs(L, vp, vc, b);
    mdct_granule(L, R, T, b);
    const size_t xo = ((size_t) gr * (gridDim.x / (unsigned) mode_gr) + ch) * 576;
    for (int i = lane; i < 576; i += 64) xr[xo + i] = L.xr[i];
    if (xr_dev) // (a copy in device memory for the kernels launched behind this one: xr is the host's)
        for (int i = lane; i < 576; i += 64) xr_dev[xo + i] = L.xr[i];
    if (flag) {
        __threadfence_system(); // (every lane: its stores to host-mapped memory first)
        if (lane == 0) {
            const unsigned n = gridDim.x;
            if (n == 1 || atomicAdd(count, 1u) == n - 1u) {
                if (n > 1) *count = 0u;
                __threadfence_system();
                *flag = seq;
            }
        }
    }
}

void mp3mi_launch_window_filter(const mp3mi_tables *T, double *ring, int off, const mp3mi_dropin_samples &in, double *zs, hipStream_t st)
{
    hipLaunchKernelGGL(k_window_filter, dim3(1), dim3(64), 0, st, T, ring, off, in, zs);
}

void mp3mi_launch_filter_subband(const mp3mi_tables *T, const double *z, double *s, hipStream_t st)
{
    hipLaunchKernelGGL(k_filter_subband, dim3(1), dim3(64), 0, st, T, z, s);
}

void mp3mi_launch_mdct_sub(const mp3mi_tables *T, const double *sb_in, double *sb_out, const int32_t *bt, double *xr, double *xr_dev, int stereo, int mode_gr,
                           unsigned *zero_me, unsigned *flag, unsigned seq, unsigned *count, hipStream_t st)
{
    hipLaunchKernelGGL(k_mdct_sub, dim3((unsigned) (stereo * mode_gr)), dim3(64), 0, st, T, sb_in, sb_out, bt, xr, xr_dev, mode_gr, zero_me,
                       (volatile unsigned *) flag, seq, count);
}
