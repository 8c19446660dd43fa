// Polyphase analysis filterbank, then MDCT + alias reduction.
//
// Replaces, for a whole batch, window_subband/filter_subband (src/encode.c:287-409) and
// mdct_sub/mdct (src/mdct.c:25-511).  Two kernels with the subband samples of the chunk in HBM
// between them (4.6 KB per granule and channel, written once and read twice):
//
//   k_filter  one wavefront per (stream, channel, 64 slots); a lane computes the 32 subband samples of one slot.
//             The filterbank is feed-forward -- a slot is a pure function of 512 PCM samples -- so every
//             slot is independent, including those of the granule BEFORE the chunk, which the reference
//             would still hold in l3_sb_sample[ch][0] and which is recomputed here (granule slot 0).
//   k_mdct    one wavefront per (stream, channel, run of 22 granules): 36 inputs per band from two granules.
//
// Arithmetic and its ordering: fbmdct_dev.h.
#include "fbmdct_dev.h"

// k_filter: one LANE per slot.  A slot's 32 subband samples are a function of 512 PCM samples, and every operand
// of that function but the PCM itself is the same for all slots: the 512 window taps and the 31 x 32 matrixing
// coefficients.  With a lane per slot they are wavefront-uniform -- scalar loads, SGPR operands of the f64
// multiplies -- and the lane's own operands (its 32 folded window sums u) stay in registers: the inner loops read no
// LDS at all.  (A lane per SUBBAND, as before, broadcasts the 32 u's of a slot to all lanes through LDS: one 8-byte
// LDS operand per multiply-add pair, and the LDS pipe, shared by the CU's four SIMDs, was what bounded the kernel.)
// A wavefront takes 64 consecutive slots of one (stream, channel) -- 3.6 granules; the slots of the chunk, the
// granule before it included, are numbered through: slot sigma = 18 gi + q.
//
// The 1/32768 of src/encode.c:306-312 is applied to the finished sample instead of to every PCM value: scaling by a
// power of two commutes with every rounding on the way (products, sums; nothing comes near the subnormals), so
// (sum of (x/32768) w ...) == (sum of x w ...) / 32768 bit for bit; the sign flip of mdct_sub's odd slots of odd
// subbands (src/mdct.c:57-60) rides on the same multiply.
#define FILT_SLOTS 64
#define FILT_WIN (32 * (FILT_SLOTS - 1) + 512)            // PCM samples under a wavefront's slots
#define FILT_PAD(t) ((t) + 2 * ((t) >> 5))                 // a dword of padding per 32 samples: lane stride 17 dwords
struct filter_lds {
    union {
        int16_t pcm[FILT_PAD(FILT_WIN) + 2];
        double tr[FILT_SLOTS][17]; // 16 subbands of every slot on their way to coalesced stores (the PCM is dead by then)
    };
};

// y[i] of filter_subband for this lane's slot, times 32768: sum over k of pcm * enwindow[i + 64 k], in k order
// (src/encode.c:306-312, 393-397); P: this lane's window in the padded LDS copy
template <int I> MP3MI_DEVFN double filt_y(const int16_t *P, const mp3mi_tables *T)
{
    double acc = (double) P[FILT_PAD(511 - I)] * T->enwindow[I];
#pragma unroll
    for (int k = 1; k < 8; k++) acc = acc + (double) P[FILT_PAD(511 - I - 64 * k)] * T->enwindow[I + 64 * k];
    return acc;
}

// u[0] = y[16], u[1+j] = y[j] + y[32-j] (j < 16), u[17+j] = y[33+j] - y[63-j] (j < 15)   (src/encode.c:398-408);
// taken in ascending i so that the window taps are read in address order; a + b == b + a bit for bit
template <int I> MP3MI_DEVFN void filt_fold(double (&u)[32], const int16_t *P, const mp3mi_tables *T)
{
    if constexpr (I < 64) {
        if constexpr (I != 48) { // (y[48] has no partner: its coefficient is cos(pi/2))
            const double y = filt_y<I>(P, T);
            if constexpr (I < 16) u[1 + I] = y;
            else if constexpr (I == 16) u[0] = y;
            else if constexpr (I <= 32) u[1 + 32 - I] = u[1 + 32 - I] + y;
            else if constexpr (I < 48) u[17 + I - 33] = y;
            else u[17 + 63 - I] = u[17 + 63 - I] - y;
        }
        // (keeps the compiler from fetching all 504 taps ahead and spilling them: four slots' worth are in flight)
        if constexpr ((I & 3) == 3) __asm__ volatile("" ::: "memory");
        filt_fold<I + 1>(u, P, T);
    }
}

__global__ void __launch_bounds__(64, 3) k_filter(const mp3mi_tables *__restrict__ T, mp3mi_geom geo,
                                                  const int16_t *__restrict__ pcm_all, double *__restrict__ sbs,
                                                  double *__restrict__ sb_dbg)
{
    __shared__ filter_lds L;
    const int lane = wave_lane();
    const int C = geo.channels, G1 = geo.n_gran + 1, NS = G1 * 18, NB = (NS + FILT_SLOTS - 1) / FILT_SLOTS;
    int bid = (int) blockIdx.x;
    const int ch = bid % C; bid /= C;
    const int blk = bid % NB;
    const int s = bid / NB;
    const long n_pitch = (long) geo.n_frames * 1152; // row pitch of the PCM buffer
    const long n_per_ch = geo.n_samples ? (long) geo.n_samples[s] : n_pitch; // valid samples: the rest reads as zero (src/encode.c:162-166)
    const int16_t *pcm = pcm_all + (size_t) s * (size_t) n_pitch * (size_t) C;
    const int16_t *hist = geo.hist ? geo.hist + (size_t) s * MP3MI_PCM_HIST * (size_t) C : NULL;
    // slot sigma needs samples [32 sigma - 480, 32 sigma + 32) counted from the first sample of granule slot 0; before
    // the call's first sample the stream's history (zeros at the start of a stream), beyond the stream's last sample 0
    const long t0 = 576 * ((long) geo.g0 - 1) + 32L * FILT_SLOTS * blk - 480;
    if (t0 >= 0 && t0 + FILT_WIN <= n_per_ch) { // all inside the call's PCM: one scalar base, 32-bit lane offsets, no range tests
        const int16_t *p0 = pcm + t0 * C + ch;
        const unsigned lo = (unsigned) lane * (unsigned) C, step = 64u * (unsigned) C;
#pragma unroll
        for (int k0 = 0; k0 < FILT_WIN / 64; k0 += 13) {
            int16_t v[13];
#pragma unroll
            for (int k = 0; k < 13; k++) v[k] = p0[lo + step * (unsigned) (k0 + k)];
#pragma unroll
            for (int k = 0; k < 13; k++) L.pcm[FILT_PAD(lane + 64 * (k0 + k))] = v[k];
        }
        if (lane < FILT_WIN % 64) L.pcm[FILT_PAD(lane + 64 * (FILT_WIN / 64))] = p0[lo + step * (unsigned) (FILT_WIN / 64)];
    } else {
        for (int i = lane; i < FILT_WIN; i += 64) {
            const long t = t0 + i;
            const bool past = hist && t < 0 && t >= -MP3MI_PCM_HIST;
            L.pcm[FILT_PAD(i)] = (t >= 0 && t < n_per_ch) ? pcm[t * C + ch] : (past ? hist[(t + MP3MI_PCM_HIST) * C + ch] : (int16_t) 0);
        }
    }
    __syncthreads();
    double u[32];
    filt_fold<0>(u, &L.pcm[34 * lane], T);
    __syncthreads(); // every lane has read its PCM: the results may take its place

    // where the slots of this wavefront go: lane >> 4 picks the slot of a group of four, the walk below adds four
    const int sg = FILT_SLOTS * blk + (lane >> 4);
    const int gi = sg / 18, q = sg - 18 * gi;
#pragma unroll 1
    for (int half = 0; half < 2; half++) {
        // s[sub] = y[16] + sum_j filt[j] (y[j] + y[32-j]) + sum_j filt[16+j] (y[33+j] - y[63-j])   (src/encode.c:398-408)
#pragma unroll 1
        for (int sb = 0; sb < 16; sb++) {
            const double *f = T->filt[16 * half + sb];
            double si = u[0];
#pragma unroll
            for (int j = 0; j < 31; j++) si = si + f[j] * u[1 + j];
            L.tr[lane][sb] = si;
            __asm__ volatile("" ::: "memory"); // (likewise: one row of coefficients at a time)
        }
        __syncthreads();
        const int sub = 16 * half + (lane & 15);
        int sgm = sg, gim = gi, qm = q;
#pragma unroll 4
        for (int m = 0; m < 16; m++) { // element lane + 64 m of [slot][16]: slot 4 m + (lane >> 4)
            const double raw = L.tr[4 * m + (lane >> 4)][lane & 15] * 0x1p-15;
            if (sgm < NS) {
                // before the stream: the reference's zero-initialised l3_sb_sample
                const bool before = 2 * geo.fabs0 + (long) geo.g0 - 1 + gim < 0;
                if (sb_dbg && gim > 0) // raw subband samples as filter_subband returns them (parity tests)
                    sb_dbg[(((size_t) s * geo.n_gran + gim - 1) * C + ch) * 576 + qm * 32 + sub] = raw;
                // mdct_sub negates odd slots of odd subbands before use (src/mdct.c:57-60)
                sbs[(((size_t) s * G1 + gim) * C + ch) * 576 + qm * 32 + sub] = before ? 0.0 : (((sub & 1) && (qm & 1)) ? raw * -1.0 : raw);
            }
            sgm += 4; qm += 4;
            if (qm >= 18) { qm -= 18; gim++; }
        }
        __syncthreads(); // the other half's results take the same place
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
    hipLaunchKernelGGL(k_filter, dim3((unsigned) (g.n_streams * g.channels * (((g.n_gran + 1) * 18 + FILT_SLOTS - 1) / FILT_SLOTS))), dim3(64), 0, st, T, g, pcm, sbs, sb_dbg);
    hipLaunchKernelGGL(k_mdct, dim3((unsigned) (g.n_streams * g.channels * ((g.n_gran + MDCT_RUN - 1) / MDCT_RUN))), dim3(64), 0, st, T, g, sbs, psy, xr);
}
