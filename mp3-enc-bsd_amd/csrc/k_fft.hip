// Psychoacoustic model, feed-forward part: windowing, the 1024-point and three 256-point
// real split-radix FFTs, energies, phases and the unpredictability of lines 6..205.
//
// Replaces fft()/rsfft()/enphinew() (src/subs.c:38-123, 412-534) and src/l3psy.c:477-549 for
// every (stream, granule, channel) of a chunk at once; one wavefront per (granule, channel).
// The FFT arithmetic is single precision with the reference's exact butterfly DAG: the
// recursion is flattened on the host (tables_host.cpp) into barrier-separated segments of
// independent butterflies which the 64 lanes execute from LDS.
//
// Output per (granule, channel), consumed by k_psy:
//   energy_l[513] f32, energy_s[3][129] f32, cw_mid[50] f64 (cw of lines 6+4n..9+4n),
//   hist6[12] f32 = r[0..5], phi[0..5] of the long FFT (history for the next two granules).
// Algorithmic HBM bytes: 2304 B PCM in (1344-sample window, 58 % shared with neighbours),
// 4048 B out.
#include "mp3mi_host.h"
#include "dmath.h"

#define R_SQHALF 0.707106781186547524401 /* src/subs.c:27 */

// Diagnostic build only (-DMP3MI_FFT_PROFILE): cycles per phase, summed over all waves.
#if defined(MP3MI_FFT_PROFILE) && !defined(MP3MI_EMU)
__device__ unsigned long long g_fft_prof[8];
#define PROF_DECL unsigned long long prof_t = __builtin_amdgcn_s_memtime(), prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PROF(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); prof_acc[i] += n_ - prof_t; prof_t = n_; } while (0)
#define PROF_END do { if (lane == 0) for (int i_ = 0; i_ < 8; i_++) atomicAdd(&g_fft_prof[i_], prof_acc[i_]); } while (0)
#else
#define PROF_DECL
#define PROF(i)
#define PROF_END
#endif

// One wavefront transforms all C channels of a granule: every butterfly record is fetched once
// and applied to the C channels of the long (then of the three short) transforms, which also gives
// each lane C independent dependency chains.
// A workgroup is W such wavefronts (W granules) sharing ONE copy of the butterfly program in LDS:
// a record costs an LDS read instead of an L2 round trip per round, which is what the transform
// was waiting on when the program lived in global memory.
// Element e of the long transform is the C floats at x[e * C] (the channels of one position side by
// side), element e of short window sb those at x[(sb * 256 + e) * C]: a butterfly fetches and stores
// both channels of an operand with one 8-byte LDS access instead of two 4-byte ones.  The short windows
// reuse the space once the long spectrum is consumed.
template <int C> struct fft_wave_lds {
    float x[C * 1024];
};
template <int C> struct fft_vec;
template <> struct fft_vec<1> { float c[1]; };
template <> struct __attribute__((aligned(8))) fft_vec<2> { float c[2]; };
template <int C, int W> struct fft_lds {
    uint32_t prog_g[64 * (MP3MI_FFT_GROUNDS_L + 2)]; // + two rounds that are read ahead but never used
    uint4 prog_r[64 * (MP3MI_FFT_RROUNDS_L + 1)];
    fft_wave_lds<C> w[W];
};

// A butterfly is applied in two steps -- fetch the operands of all arrays, then compute and store --
// so that the rounds of a segment (which are independent of each other) can be taken two at a time:
// both rounds' LDS reads are in flight together and only one LDS latency is exposed per pair.
struct fft_idx { int a, b, c, d; bool on; };

template <int TYPE>
MP3MI_DEVFN fft_idx fft_decode(uint32_t w0, uint32_t w1)
{
    // operand indices are LDS positions (MP3MI_FFT_SWZ applied on the host); an idle lane (bit 31)
    // decodes to position 0, reads it and stores nothing
    fft_idx i;
    i.on = !(w0 >> 31);
    i.c = 0; i.d = 0;
    if (TYPE == FOP_ROT) {
        i.a = (int) (w0 & 0x3ffu); i.b = (int) ((w0 >> 16) & 0x3ffu);
    } else {
        i.a = (int) (w0 & 1023u); i.b = (int) ((w0 >> 10) & 1023u);
        if (TYPE == FOP_CROSS) { i.c = (int) (w1 & 1023u); i.d = (int) ((w1 >> 10) & 1023u); }
    }
    return i;
}

// NWIN transforms of N points with C channels each (layout above)
template <int TYPE, int C, int NWIN, int N>
MP3MI_DEVFN void fft_fetch(const float *x, const fft_idx &i, fft_vec<C> (&v)[NWIN][4])
{
#pragma unroll
    for (int w = 0; w < NWIN; w++) {
        const float *p = x + w * N * C;
        v[w][0] = *(const fft_vec<C> *) (p + i.a * C);
        if (TYPE != FOP_NEG) v[w][1] = *(const fft_vec<C> *) (p + i.b * C);
        if (TYPE == FOP_CROSS) { v[w][2] = *(const fft_vec<C> *) (p + i.c * C); v[w][3] = *(const fft_vec<C> *) (p + i.d * C); }
    }
}

template <int TYPE, int C, int NWIN, int N>
MP3MI_DEVFN void fft_finish(float *x, const fft_idx &i, const fft_vec<C> (&v)[NWIN][4], uint32_t w1, uint32_t w2, uint32_t w3)
{
    if (!i.on) return;
#pragma unroll
    for (int w = 0; w < NWIN; w++) {
        float *p = x + w * N * C;
        fft_vec<C> oa, ob, oc, od;
#pragma unroll
        for (int c = 0; c < C; c++) {
            const float va = v[w][0].c[c], vb = v[w][1].c[c];
            if (TYPE == FOP_ADDSUB) {
                ob.c[c] = va - vb;
                oa.c[c] = va + vb;
            } else if (TYPE == FOP_NEG) {
                oa.c[c] = -va;
            } else if (TYPE == FOP_CROSS) {
                const float r1 = va, r2 = vb, i1 = v[w][2].c[c], i2 = v[w][3].c[c];
                oc.c[c] = i1 - r2;
                ob.c[c] = r1 - i2;
                oa.c[c] = r1 + i2;
                od.c[c] = i1 + r2;
            } else if (TYPE == FOP_ROT) {
                const float cn = __builtin_bit_cast(float, w1), spc = __builtin_bit_cast(float, w2), smc = __builtin_bit_cast(float, w3);
                const float r1 = va, i1 = vb;
                const float t2 = cn * (r1 + i1);
                const float t1 = spc * r1 + t2;
                oa.c[c] = smc * i1 + t2;
                ob.c[c] = t1;
            } else if (TYPE == FOP_SQ1) {
                oa.c[c] = (float) (R_SQHALF * (double) (va + vb));
                ob.c[c] = (float) (R_SQHALF * (double) (vb - va));
            } else if (TYPE == FOP_SQ2) {
                oa.c[c] = (float) (R_SQHALF * (double) (vb - va));
                ob.c[c] = (float) (-R_SQHALF * (double) (va + vb));
            } else if (TYPE == FOP_SWAPNN) {
                oa.c[c] = -vb;
                ob.c[c] = -va;
            } else if (TYPE == FOP_SWAPN) {
                oa.c[c] = -vb;
                ob.c[c] = va;
            } else { // FOP_SWAP
                oa.c[c] = vb;
                ob.c[c] = va;
            }
        }
        *(fft_vec<C> *) (p + i.a * C) = oa;
        if (TYPE != FOP_NEG) *(fft_vec<C> *) (p + i.b * C) = ob;
        if (TYPE == FOP_CROSS) { *(fft_vec<C> *) (p + i.c * C) = oc; *(fft_vec<C> *) (p + i.d * C) = od; }
    }
}

// The butterfly records are two streams (one-word records; 16-byte rotation records) laid out in
// rounds of 64, one record per lane, idle lanes marked by bit 31 (tables_host.cpp).  pg/pr point at
// this lane's record of the segment's first round and are left at the next segment's.  The records
// of the following pair of rounds are read while the current pair's operands are in flight.
template <int TYPE, int C, int NWIN, int N>
MP3MI_DEVFN void fft_segment(float *x, const uint32_t *&pg, const uint4 *&pr, int rounds)
{
    const int gstep = (TYPE == FOP_CROSS) ? 128 : 64; // FOP_CROSS: the round's second words follow its first words
    uint4 c0 = {0x80000000u, 0, 0, 0}, c1 = c0;
    if (TYPE == FOP_ROT) { c0 = pr[0]; if (rounds > 1) c1 = pr[64]; }
    else {
        c0.x = pg[0];
        if (TYPE == FOP_CROSS) c0.y = pg[64];
        if (rounds > 1) { c1.x = pg[gstep]; if (TYPE == FOP_CROSS) c1.y = pg[gstep + 64]; }
    }
    int t = 0;
    for (; t + 1 < rounds; t += 2) {
        const fft_idx i0 = fft_decode<TYPE>(c0.x, c0.y), i1 = fft_decode<TYPE>(c1.x, c1.y);
        fft_vec<C> v0[NWIN][4], v1[NWIN][4];
        fft_fetch<TYPE, C, NWIN, N>(x, i0, v0);
        fft_fetch<TYPE, C, NWIN, N>(x, i1, v1);
        uint4 n0 = {0x80000000u, 0, 0, 0}, n1 = n0;
        if (TYPE == FOP_ROT) {
            pr += 128;
            if (t + 2 < rounds) { n0 = pr[0]; n1 = pr[64]; } // the second may belong to what follows: never used then
        } else {
            pg += 2 * gstep;
            if (t + 2 < rounds) {
                n0.x = pg[0]; n1.x = pg[gstep];
                if (TYPE == FOP_CROSS) { n0.y = pg[64]; n1.y = pg[gstep + 64]; }
            }
        }
        fft_finish<TYPE, C, NWIN, N>(x, i0, v0, c0.y, c0.z, c0.w);
        fft_finish<TYPE, C, NWIN, N>(x, i1, v1, c1.y, c1.z, c1.w);
        c0 = n0;
        c1 = n1;
    }
    if (t < rounds) { // odd round count: the last round alone
        const fft_idx i0 = fft_decode<TYPE>(c0.x, c0.y);
        fft_vec<C> v0[NWIN][4];
        fft_fetch<TYPE, C, NWIN, N>(x, i0, v0);
        fft_finish<TYPE, C, NWIN, N>(x, i0, v0, c0.y, c0.z, c0.w);
        if (TYPE == FOP_ROT) pr += 64; else pg += gstep;
    }
}

template <int C, int NWIN, int N>
MP3MI_DEVFN void fft_run(float *x, const int32_t *segs, int nseg, const uint32_t *pg, const uint4 *pr)
{
    int nxt = segs[0];
    for (int sidx = 0; sidx < nseg; sidx++) {
        const int sw = nxt;
        nxt = segs[sidx + 1 < nseg ? sidx + 1 : sidx];
        const int type = sw & 0xff, rounds = (sw >> 8) & 0xff;
        switch (type) {
        case FOP_ADDSUB: fft_segment<FOP_ADDSUB, C, NWIN, N>(x, pg, pr, rounds); break;
        case FOP_NEG: fft_segment<FOP_NEG, C, NWIN, N>(x, pg, pr, rounds); break;
        case FOP_CROSS: fft_segment<FOP_CROSS, C, NWIN, N>(x, pg, pr, rounds); break;
        case FOP_ROT: fft_segment<FOP_ROT, C, NWIN, N>(x, pg, pr, rounds); break;
        case FOP_SQ1: fft_segment<FOP_SQ1, C, NWIN, N>(x, pg, pr, rounds); break;
        case FOP_SQ2: fft_segment<FOP_SQ2, C, NWIN, N>(x, pg, pr, rounds); break;
        case FOP_SWAPNN: fft_segment<FOP_SWAPNN, C, NWIN, N>(x, pg, pr, rounds); break;
        case FOP_SWAPN: fft_segment<FOP_SWAPN, C, NWIN, N>(x, pg, pr, rounds); break;
        default: fft_segment<FOP_SWAP, C, NWIN, N>(x, pg, pr, rounds); break;
        }
        if (sw >> 16) wave_sync();
    }
}

// energy of bin i of an N-point transform for all C channels (src/subs.c:53-123); x points at element 0.
// Branch-free: bins 0 and N/2 are real (their "imaginary" operand is read from a valid place and not
// used).  The reference's floor test `(double) e < 0.0005` is the same as the float test against
// (float) 0.0005 = 0x3a03126f, the float next ABOVE 0.0005: below it both say yes, above it both say no,
// and at it the replacement value is e itself.
template <int C>
MP3MI_DEVFN fft_vec<C> fft_energy(const float *x, int N, int i)
{
    const bool real = (i == 0) || (i == N / 2);
    const fft_vec<C> re = *(const fft_vec<C> *) (x + MP3MI_FFT_SWZ(i) * C);
    const fft_vec<C> im = *(const fft_vec<C> *) (x + MP3MI_FFT_SWZ(real ? i : N - i) * C);
    fft_vec<C> e;
#pragma unroll
    for (int c = 0; c < C; c++) {
        const float rr = re.c[c] * re.c[c];
        const float ee = rr + im.c[c] * im.c[c];
        const float fl = ee < (float) 0.0005 ? (float) 0.0005 : ee;
        e.c[c] = real ? rr : fl;
    }
    return e;
}

template <int C, int W>
__global__ void __launch_bounds__(64 * W) k_fft(const mp3mi_tables *__restrict__ T, mp3mi_geom geo,
                                                const int16_t *__restrict__ pcm_all, float *__restrict__ energy_l,
                                                float *__restrict__ energy_s, float *__restrict__ bins)
{
    __shared__ fft_lds<C, W> LL;
    const int lane = wave_lane(), tid = (int) threadIdx.x;
    fft_wave_lds<C> &L = LL.w[tid >> 6];
    const int G = geo.n_gran, n_task = geo.n_streams * G;
    int task = (int) blockIdx.x * W + (tid >> 6);
    const bool valid = task < n_task; // the last workgroup may have idle wavefronts: they compute, but do not store
    task = valid ? task : n_task - 1;
    const int gl = task % G, s = task / G;
    const size_t rec0 = ((size_t) s * G + gl) * C;
    const long gabs = (long) geo.g0 + gl;
    const long n_pitch = (long) geo.n_frames * 1152; // row pitch of the PCM buffer
    const long n_per_ch = geo.n_samples ? (long) geo.n_samples[s] : n_pitch; // valid samples: the rest reads as zero (src/encode.c:162-166)
    const int16_t *pcm = pcm_all + (size_t) s * (size_t) n_pitch * (size_t) C;
    const long t0 = 576 * gabs - 768; // time of savebuf[0]  (src/l3psy.c:477-481)
    PROF_DECL;

    for (int i = tid; i < 64 * MP3MI_FFT_GROUNDS_L; i += 64 * W) LL.prog_g[i] = T->gops_l[i];
    for (int i = tid; i < 64 * MP3MI_FFT_RROUNDS_L; i += 64 * W) LL.prog_r[i] = ((const uint4 *) T->rops_l)[i];

    // the 1024-sample window of all channels: every load is issued before the first use
    {
        float wl[16];
        uint32_t smp[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const long t = t0 + lane + 64 * k;
            const bool in = t >= 0 && t < n_per_ch;
            if (C == 2) smp[k] = in ? ((const uint32_t *) pcm)[t] : 0u;
            else smp[k] = in ? (uint32_t) (uint16_t) pcm[t] : 0u;
            wl[k] = T->window[lane + 64 * k];
        }
#pragma unroll
        for (int k = 0; k < 16; k++)
#pragma unroll
            for (int c = 0; c < C; c++) {
                const float v = (float) (int) (int16_t) (c == 0 ? (smp[k] & 0xffffu) : (smp[k] >> 16));
                L.x[MP3MI_FFT_SWZ(lane + 64 * k) * C + c] = wl[k] * v; // src/l3psy.c:485
            }
    }
    __syncthreads();
    PROF(0);

    fft_run<C, 1, 1024>(&L.x[0], T->seg_l, T->n_seg_l, LL.prog_g + lane, LL.prog_r + lane);
    PROF(1);

    // the short windows' samples are requested now and land while the long spectrum is consumed
    uint32_t smp[12];
    float wsv[4];
#pragma unroll
    for (int k = 0; k < 12; k++) {
        const long t = t0 + 256 + lane + 64 * k;
        const bool in = t >= 0 && t < n_per_ch;
        if (C == 2) smp[k] = in ? ((const uint32_t *) pcm)[t] : 0u;
        else smp[k] = in ? (uint32_t) (uint16_t) pcm[t] : 0u;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) wsv[k] = T->window_s[lane + 64 * k];

    {
        float *el0 = energy_l + rec0 * MP3MI_HBLK_P;
#pragma unroll 3
        for (int i = lane; i < MP3MI_HBLK; i += 64) {
            const fft_vec<C> e = fft_energy<C>(L.x, 1024, i);
            if (valid) {
#pragma unroll
                for (int c = 0; c < C; c++) el0[c * MP3MI_HBLK_P + i] = e.c[c];
            }
        }
        // raw bins 0..5 for k_cw: re, im (bin 0 is real: im = -0 makes atan2(-im, re) the reference's atan2(0.0, x[0]))
        if (lane < 6 && valid) {
#pragma unroll
            for (int c = 0; c < C; c++) {
                bins[(rec0 + c) * MP3MI_FFT_BINS + 300 + lane] = L.x[MP3MI_FFT_SWZ(lane) * C + c];
                bins[(rec0 + c) * MP3MI_FFT_BINS + 306 + lane] = lane ? L.x[MP3MI_FFT_SWZ(1024 - lane) * C + c] : -0.0f;
            }
        }
    }
    wave_sync(); // the long spectrum is dead from here on
    // short windows: samples 256 + 128 sb + jj, sb < 3 (src/l3psy.c:520-523); the second half of every
    // window is also the first half of the next.  Sample 256 + lane + 64 k: sb = k >> 1, jj = lane + 64 (k & 1).
#pragma unroll
    for (int k = 0; k < 12; k++)
#pragma unroll
        for (int c = 0; c < C; c++) {
            const float v = (float) (int) (int16_t) (c == 0 ? (smp[k] & 0xffffu) : (smp[k] >> 16));
            const int sb = k >> 1, jj = lane + 64 * (k & 1);
            if (sb < 3) L.x[(sb * 256 + MP3MI_FFT_SWZ(jj)) * C + c] = wsv[k & 1] * v;
            if (sb >= 1 && sb < 4) L.x[((sb - 1) * 256 + MP3MI_FFT_SWZ(128 + jj)) * C + c] = wsv[2 + (k & 1)] * v;
        }
    __syncthreads(); // every wavefront is done with the long program
    for (int i = tid; i < 64 * MP3MI_FFT_GROUNDS_S; i += 64 * W) LL.prog_g[i] = T->gops_s[i];
    for (int i = tid; i < 64 * MP3MI_FFT_RROUNDS_S; i += 64 * W) LL.prog_r[i] = ((const uint4 *) T->rops_s)[i];
    __syncthreads();
    PROF(2);

    fft_run<C, 3, 256>(&L.x[0], T->seg_s, T->n_seg_s, LL.prog_g + lane, LL.prog_r + lane);
    PROF(3);

    // energies of the three short spectra (bin k of window sb, both channels per LDS read) and the raw
    // short lines 2..51 for k_cw (src/l3psy.c:531-549 reads these only); plain nested loops, no div/mod
#pragma unroll
    for (int sb = 0; sb < 3; sb++) {
        float *es0 = energy_s + rec0 * (3 * MP3MI_HBLK_S) + sb * MP3MI_HBLK_S;
        const float *xw = L.x + sb * 256 * C;
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const int k = lane + 64 * t;
            if (k < MP3MI_HBLK_S) {
                const fft_vec<C> e = fft_energy<C>(xw, 256, k);
                if (valid) {
#pragma unroll
                    for (int c = 0; c < C; c++) es0[c * (3 * MP3MI_HBLK_S) + k] = e.c[c];
                }
            }
        }
        if (lane < 50 && valid) {
            const fft_vec<C> re = *(const fft_vec<C> *) (xw + MP3MI_FFT_SWZ(2 + lane) * C);
            const fft_vec<C> im = *(const fft_vec<C> *) (xw + MP3MI_FFT_SWZ(254 - lane) * C);
#pragma unroll
            for (int c = 0; c < C; c++) {
                float *o = bins + (rec0 + c) * MP3MI_FFT_BINS + (sb * 50 + lane) * 2;
                o[0] = re.c[c];
                o[1] = im.c[c];
            }
        }
    }
    PROF(4);
    PROF_END;
}

// energy and phase of a raw bin (src/subs.c:53-123): phi = (float) atan2(-im, re).  Bins below the energy
// floor have phase 0 and never reach atan2; `exact` marks the real-valued bin 0, which has no floor.
// Two tiers: the phase is only needed as a FLOAT, so a plain-double atan2 (error < 2^-50) decides it unless
// the value lies within 2^-46 of the midpoint of two floats (dm_float_rounding_safe); then *unsafe is
// set and the caller repeats the wavefront with the correctly rounded dm_atan2 (EXACT).
template <bool EXACT>
MP3MI_DEVFN void cw_bin(float re, float im, bool exact, float *energy, float *phi, bool *unsafe)
{
    const float e = re * re + im * im;
    const bool low = !exact && e < (float) 0.0005; // == ((double) e < 0.0005), see fft_energy
    *energy = low ? (float) 0.0005 : e;
    float ph = 0.0f;
    if (!low) {
        const double y = -(double) im, x = (double) re;
        if (EXACT || y == 0.0 || x == 0.0) ph = (float) dm_atan2(y, x);
        else {
            const double v = dm_atan2_fast(y, x);
            if (!dm_float_rounding_safe(v)) *unsafe = true;
            ph = (float) v;
        }
    }
    *phi = ph;
}

// Phases and the unpredictability measure from the raw FFT bins, one wavefront per (granule, channel):
// lanes 0..49 the unpredictability of lines 6+4n..9+4n from the three short FFTs (src/l3psy.c:531-549),
// lanes 50..55 magnitude and phase of long lines 0..5 (src/l3psy.c:497-503).  Kept out of k_fft so that
// this double-precision chain runs at full occupancy instead of next to 150 KB of LDS.
__global__ void __launch_bounds__(64) k_cw(const float *__restrict__ bins, double *__restrict__ cw_mid, float *__restrict__ hist6,
                                          int force_exact)
{
    const int lane = wave_lane();
    const size_t rec = blockIdx.x;
    const float *b = bins + rec * MP3MI_FFT_BINS;
    float re[3] = {1.0f, 1.0f, 1.0f}, im[3] = {0.0f, 0.0f, 0.0f};
    if (lane < 50) {
#pragma unroll
        for (int sb = 0; sb < 3; sb++) { re[sb] = b[(sb * 50 + lane) * 2]; im[sb] = b[(sb * 50 + lane) * 2 + 1]; }
    } else if (lane < 56) {
        re[0] = b[300 + lane - 50];
        im[0] = b[306 + lane - 50];
    }
    float e[3], ph[3];
    bool unsafe = force_exact != 0;
    if (!unsafe) {
#pragma unroll
        for (int sb = 0; sb < 3; sb++) cw_bin<false>(re[sb], im[sb], lane == 50 && sb == 0, &e[sb], &ph[sb], &unsafe);
    }
    if (wave_any(unsafe)) { // rare: some phase too close to a float midpoint for the first tier
#pragma unroll
        for (int sb = 0; sb < 3; sb++) cw_bin<true>(re[sb], im[sb], lane == 50 && sb == 0, &e[sb], &ph[sb], &unsafe);
    }
    if (lane < 50) {
        const double r_prime = 2.0 * __builtin_sqrt((double) e[0]) - __builtin_sqrt((double) e[2]);
        const double phi_prime = 2.0 * (double) ph[0] - (double) ph[2];
        const double r2 = __builtin_sqrt((double) e[1]);
        const double phi2 = (double) ph[1];
        double s2, c2, sp, cp;
        dm_sincos(phi2, &s2, &c2);
        dm_sincos(phi_prime, &sp, &cp);
        const double t1 = r2 * c2 - r_prime * cp;
        const double t2 = r2 * s2 - r_prime * sp;
        const double t3 = r2 + __builtin_fabs(r_prime);
        double cw = 0.0;
        if (t3 != 0.0) cw = __builtin_sqrt(t1 * t1 + t2 * t2) / t3;
        cw_mid[rec * 50 + lane] = cw;
    } else if (lane < 56) {
        hist6[rec * 12 + lane - 50] = (float) __builtin_sqrt((double) e[0]); // r, src/l3psy.c:500
        hist6[rec * 12 + 6 + lane - 50] = ph[0];
    }
}

#if defined(MP3MI_FFT_PROFILE) && !defined(MP3MI_EMU)
extern "C" void mp3mi_debug_fft_profile(unsigned long long *out)
{
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fft_prof), sizeof(z));
    hipMemcpyToSymbol(HIP_SYMBOL(g_fft_prof), z, sizeof(z));
}
#endif

void mp3mi_launch_fft(const mp3mi_tables *T, const mp3mi_geom &g, const int16_t *pcm, float *energy_l,
                      float *energy_s, float *bins, double *cw_mid, float *hist6, hipStream_t st)
{
    // W: as many wavefronts as fit the 160 KB of LDS next to the shared program
    const int n_task = g.n_streams * g.n_gran;
    if (g.channels == 2) {
        const int W = 13;
        hipLaunchKernelGGL((k_fft<2, W>), dim3((unsigned) ((n_task + W - 1) / W)), dim3(64 * W), 0, st, T, g, pcm, energy_l, energy_s, bins);
    } else {
        const int W = 16;
        hipLaunchKernelGGL((k_fft<1, W>), dim3((unsigned) ((n_task + W - 1) / W)), dim3(64 * W), 0, st, T, g, pcm, energy_l, energy_s, bins);
    }
    hipLaunchKernelGGL(k_cw, dim3((unsigned) (n_task * g.channels)), dim3(64), 0, st, bins, cw_mid, hist6, (g.test_flags >> 1) & 1);
}
