// Psychoacoustic model, feed-forward part: windowing, the 1024-point and three 256-point
// real split-radix FFTs, energies, phases and the unpredictability of lines 6..205.
//
// Replaces fft()/rsfft()/enphinew() (src/subs.c:38-123, 412-534) and src/l3psy.c:477-549 for
// every (stream, granule, channel) of a chunk at once: k_fft<.., true> (the long transform) and
// k_fft<.., false> (the three short ones), one wavefront per (stream, granule) task and all channels,
// then k_cw.  The FFT arithmetic is single precision with the reference's exact butterfly DAG, as FUSED
// butterflies -- steps 1-4 of one recursion level for one index, 4 or 8 operands.  Those of the blocks of
// 256 points and more run in the lanes' registers, where the windowed samples arrive (fft_reg_long,
// fft_reg4); the rest of the recursion is flattened on the host (tables_host.cpp) into rounds of 64
// independent butterflies which the lanes execute from LDS; the data movement the reference ends with
// (step 5, bit reversal) is folded into the read-out.
//
// Output per (granule, channel), consumed by k_psy:
//   energy_l[513] f32, energy_s[3][129] f32, cw_mid[50] f64 (cw of lines 6+4n..9+4n),
//   hist6[12] f32 = r[0..5], phi[0..5] of the long FFT (history for the next two granules).
// Algorithmic HBM bytes: 2304 B PCM in (1344-sample window, 58 % shared with neighbours),
// 4048 B out.
#include "mp3mi_host.h"
#include "l12_dev.h"
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
// and applied to the C channels, which also gives each lane C independent dependency chains.
// A workgroup is W such wavefronts (W granules) sharing ONE copy of the butterfly program in LDS.
// Element e of an array is the C floats at x[e * C] (the channels of one position side by side): a
// butterfly fetches and stores both channels of an operand with one 8-byte LDS access.  The three short
// transforms are one program over elements [256 sb, 256 sb + 256) and reuse the space once the long
// spectrum is consumed; elements 1024 + lane are what idle lanes work on.
template <int C, bool LONG> struct fft_wave_lds {
    float x[C * ((LONG ? MP3MI_FFT_DUMMY : MP3MI_FFT_DUMMY_S) + 64)];
};
// fft_vec<C>: the C channels of one element.  For two channels a native 2-vector, so that the channel pair
// of an operand is one 8-byte LDS access and one packed instruction (v_pk_add_f32 / v_pk_mul_f32: the same
// IEEE results per component as the scalar ones) without any register shuffling.
template <int C> struct fft_pair;  // the same C floats as a plain struct (windowing, read-out)
template <> struct fft_pair<1> { float c[1]; };
template <> struct __attribute__((aligned(8))) fft_pair<2> { float c[2]; };
typedef float fft_v2f __attribute__((vector_size(8)));
#if defined(MP3MI_EMU)
#define FFT_MEMFN static inline
#define MP3MI_DEVFN_M inline
#else
#define FFT_MEMFN static __device__ __forceinline__
#define MP3MI_DEVFN_M __device__ __forceinline__
#endif
typedef uint32_t fft_v2u __attribute__((vector_size(8)));
template <int C> struct fft_vec;
template <> struct fft_vec<1> {
    typedef float V;
    FFT_MEMFN V splat(float f) { return f; }
    FFT_MEMFN V flip(V v, uint32_t signbit) { return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, v) ^ signbit); }
    FFT_MEMFN float get(V v, int) { return v; }
    FFT_MEMFN void set(V &v, int, float f) { v = f; }
};
template <> struct fft_vec<2> {
    typedef fft_v2f V;
    FFT_MEMFN V splat(float f) { return (V){f, f}; }
    FFT_MEMFN V flip(V v, uint32_t signbit) { return (V) ((fft_v2u) v ^ (fft_v2u){signbit, signbit}); }
    FFT_MEMFN float get(V v, int c) { return v[c]; }
    FFT_MEMFN void set(V &v, int c, float f) { v[c] = f; }
};
// a workgroup's LDS: the program, the rotation rows of the register rounds and the read-out table -- everything a task
// looks up per lane, one LDS round trip away instead of one to the L2 -- and the W wavefronts' arrays
template <int C, int W, bool LONG> struct fft_lds {
    static constexpr int PW = LONG ? MP3MI_FFT_PROG_WORDS : MP3MI_FFT_PROG_WORDS_S;  // capacity of the program in words
    static constexpr int ROWS = LONG ? MP3MI_FFT_REG_ROWS_L : 1, NRD = LONG ? MP3MI_HBLK : MP3MI_HBLK_S;
    uint32_t prog[PW] __attribute__((aligned(16)));
    uint4 regtw[ROWS * 64];
    uint4 leaf[64];
    uint32_t rd[(NRD + 63) / 64 * 64];
    float win[LONG ? 1024 : 256];
    fft_wave_lds<C, LONG> w[W];
};
// the workgroup's copy of the tables (all its wavefronts; a barrier follows)
template <int C, int W, bool LONG> MP3MI_DEVFN void fft_lds_fill(fft_lds<C, W, LONG> &LL, const mp3mi_tables *__restrict__ T, int tid)
{
    const int nw4 = (LONG ? T->fft_nword_l : T->fft_nword_s) / 4;
    const uint4 *src = (const uint4 *) (LONG ? T->fft_prog_l : T->fft_prog_s);
    for (int i = tid; i < nw4; i += 64 * W) ((uint4 *) LL.prog)[i] = src[i];
    const uint4 *rsrc = (const uint4 *) (LONG ? T->fft_regtw_l : T->fft_regtw_s);
    for (int i = tid; i < fft_lds<C, W, LONG>::ROWS * 64; i += 64 * W) LL.regtw[i] = rsrc[i];
    if (tid < 64) LL.leaf[tid] = ((const uint4 *) (LONG ? T->fft_leaf_l : T->fft_leaf_s))[tid];
    const uint32_t *dsrc = LONG ? T->fft_rd_l : T->fft_rd_s;
    for (int i = tid; i < fft_lds<C, W, LONG>::NRD; i += 64 * W) LL.rd[i] = dsrc[i];
    const float *wsrc = LONG ? T->window : T->window_s;
    for (int i = tid; i < (LONG ? 1024 : 256); i += 64 * W) LL.win[i] = wsrc[i];
}

// Operand of a record word: LDS element position in its low (hi = 0) or high half.  xw = the workgroup's
// arrays as bytes, woff = byte offset of this wavefront's array: position * element size + woff is ONE
// instruction (v_mad_u32_u16 reads either half of a register).
template <int C, int HI> MP3MI_DEVFN typename fft_vec<C>::V *fft_at(char *xw, uint32_t woff, uint32_t w)
{
    uint32_t off;
#if defined(MP3MI_EMU)
    off = (HI ? (w >> 16) : (w & 0xffffu)) * (4u * C) + woff;
#else
    if (C == 2) {
        if (HI) asm("v_mad_u32_u16 %0, %1, 8, %2 op_sel:[1,0,0,0]" : "=v"(off) : "v"(w), "v"(woff));
        else asm("v_mad_u32_u16 %0, %1, 8, %2" : "=v"(off) : "v"(w), "v"(woff));
    } else {
        if (HI) asm("v_mad_u32_u16 %0, %1, 4, %2 op_sel:[1,0,0,0]" : "=v"(off) : "v"(w), "v"(woff));
        else asm("v_mad_u32_u16 %0, %1, 4, %2" : "=v"(off) : "v"(w), "v"(woff));
    }
#endif
    return (typename fft_vec<C>::V *) (xw + off);
}

// twiddle stage of a fused butterfly on (r, i) (src/subs.c:330-339, 487-495): flags bit 0 rotates by
// (cn, spcn, smcn), bit 1 by SQHALF -- `second` is the form the reference uses for the pair (xr2, xi2)
template <int C, bool ROT, bool SQ>
MP3MI_DEVFN void fft_twiddle(typename fft_vec<C>::V &r, typename fft_vec<C>::V &i, uint32_t flags, bool second,
                             uint32_t cn, uint32_t spc, uint32_t smc)
{
    typedef fft_vec<C> F;
    typedef typename F::V V;
    const V r1 = r, i1 = i;
    if (ROT) {
        const V t2 = F::splat(__builtin_bit_cast(float, cn)) * (r1 + i1);
        const V t1 = F::splat(__builtin_bit_cast(float, spc)) * r1 + t2;
        const V ra = F::splat(__builtin_bit_cast(float, smc)) * i1 + t2;
        r = (flags & 1u) ? ra : r1;
        i = (flags & 1u) ? t1 : i1;
    }
    if (SQ) {
        const V sum = r1 + i1, dif = i1 - r1;
        V qa, qb;
#pragma unroll
        for (int c = 0; c < C; c++) {
            const float sm = F::get(sum, c), df = F::get(dif, c);
            F::set(qa, c, second ? (float) (R_SQHALF * (double) df) : (float) (R_SQHALF * (double) sm));
            F::set(qb, c, second ? (float) (-R_SQHALF * (double) sm) : (float) (R_SQHALF * (double) df));
        }
        r = (flags & 2u) ? qa : r;
        i = (flags & 2u) ? qb : i;
    }
}

// A round in two halves -- fetch (records, operand addresses, operands) and finish (arithmetic, stores) -- so
// that two rounds of the same rank, which touch different elements, can have their LDS reads in flight
// together: the kernel runs 3 wavefronts per SIMD and every round is two dependent LDS round trips.
// H = the round's header bits (bit 0: eight operands, bit 1: rotations, bit 2: SQHALF rotations).
template <int C, int H> struct fft_round {
    typedef fft_vec<C> F;
    typedef typename F::V V;
    static constexpr int N = (H & 1) ? 8 : 4;
    static constexpr bool ROT = (H & 2) != 0, SQ = (H & 4) != 0;
    V *p[N];
    V v[N];
    uint4 tw1, tw3;

    MP3MI_DEVFN_M void fetch(char *xw, uint32_t woff, const uint32_t *blk, int lane)
    {
        tw1 = tw3 = uint4{0, 0, 0, 0};
        if (N == 4) { // tables_host.cpp, FusedOp cls 0
            const uint2 ad = *(const uint2 *) (blk + 2 * lane);
            if (ROT) tw1 = *(const uint4 *) (blk + 128 + 4 * lane);
            else tw1.w = blk[128 + lane];
            p[0] = fft_at<C, 0>(xw, woff, ad.x); p[1] = fft_at<C, 1>(xw, woff, ad.x);
            p[2] = fft_at<C, 0>(xw, woff, ad.y); p[3] = fft_at<C, 1>(xw, woff, ad.y);
        } else { // FusedOp cls 1
            const uint4 ad = *(const uint4 *) (blk + 4 * lane);
            if (ROT) { tw1 = *(const uint4 *) (blk + 256 + 4 * lane); tw3 = *(const uint4 *) (blk + 512 + 4 * lane); }
            else tw1.w = blk[256 + lane];
            p[0] = fft_at<C, 0>(xw, woff, ad.x); p[1] = fft_at<C, 1>(xw, woff, ad.x);
            p[2] = fft_at<C, 0>(xw, woff, ad.y); p[3] = fft_at<C, 1>(xw, woff, ad.y);
            p[4 % N] = fft_at<C, 0>(xw, woff, ad.z); p[5 % N] = fft_at<C, 1>(xw, woff, ad.z);
            p[6 % N] = fft_at<C, 0>(xw, woff, ad.w); p[7 % N] = fft_at<C, 1>(xw, woff, ad.w);
        }
#pragma unroll
        for (int k = 0; k < N; k++) v[k] = *p[k];
    }

    // the arithmetic: the operands in v[] become the results that take their places
    MP3MI_DEVFN_M void compute()
    {
        const uint32_t flags = tw1.w;
        if (N == 4) { // steps 1-4 of rsrec for one n (src/subs.c:465-498), or two length-2 butterflies
            const V a = v[0], b = v[1], c = v[2], d = v[3];
            const V oa = a + b, oc = c + d;
            V u1 = a - b;
            V u2 = F::flip(c - d, flags & 0x80000000u); // src/subs.c:475-479
            fft_twiddle<C, ROT, SQ>(u1, u2, flags, false, tw1.x, tw1.y, tw1.z);
            v[0] = oa; v[1] = u1; v[2] = oc; v[3] = u2;
        } else { // steps 1-4 of srrec for one n
            const V ar0 = v[0], ar1 = v[1], br0 = v[2], br1 = v[3], ai0 = v[4 % N], ai1 = v[5 % N], bi0 = v[6 % N], bi1 = v[7 % N];
            // step 1 (src/subs.c:288-298)
            const V o0 = ar0 + ar1, o2 = br0 + br1, o4 = ai0 + ai1, o6 = bi0 + bi1;
            const V xr1 = ar0 - ar1, xr2 = br0 - br1, xi1 = ai0 - ai1, xi2 = bi0 - bi1;
            // step 2 (src/subs.c:301-312)
            V r1 = xr1 + xi2, i2 = xi1 + xr2, i1 = xi1 - xr2, r2 = xr1 - xi2;
            // steps 3 and 4 (src/subs.c:327-342)
            fft_twiddle<C, ROT, SQ>(r1, i1, flags, false, tw1.x, tw1.y, tw1.z);
            fft_twiddle<C, ROT, SQ>(r2, i2, flags, true, tw3.x, tw3.y, tw3.z);
            v[0] = o0; v[1] = r1; v[2] = o2; v[3] = r2; v[4 % N] = o4; v[5 % N] = i1; v[6 % N] = o6; v[7 % N] = i2;
        }
    }

    MP3MI_DEVFN_M void finish()
    {
        compute();
#pragma unroll
        for (int k = 0; k < N; k++) *p[k] = v[k];
    }
};

// The blocks of 256 points and more IN REGISTERS.  With element e of a transform in register e / 64 of lane e % 64 (which
// is how the windowed samples arrive), the four or eight operands of a butterfly of a block of m >= 256 points -- m / 2
// and m / 4 apart -- are registers of one lane: R(1024), R(512), C(256) and R(256) of the long transform, R(256) of each
// short one run here, before the data ever reaches LDS -- a third of all operand traffic, 8 of the long program's 24
// rounds and 3 of the short one's 16, with their address words, and two dependent LDS round trips each.  The same
// arithmetic as a round of the program (fft_round::compute); what a lane needs beside its registers is its row of rotations
// (tables_host.cpp, FftGen::reg_row), 16 bytes a butterfly from T->fft_regtw_*.
template <int C, int H> MP3MI_DEVFN void fft_reg4(typename fft_vec<C>::V &a, typename fft_vec<C>::V &b, typename fft_vec<C>::V &c,
                                                  typename fft_vec<C>::V &d, const uint4 tw)
{
    fft_round<C, H & 6> r; // (a, b, c, d) = x[n], x[n + m/2], x[n + m/4], x[n + 3m/4]
    r.v[0] = a; r.v[1] = b; r.v[2] = c; r.v[3] = d;
    r.tw1 = tw;
    r.compute();
    a = r.v[0]; b = r.v[1]; c = r.v[2]; d = r.v[3];
}
// The windowed samples of the long transform, x[k] = element 64 k + lane, and the register rounds on them; rt = the
// workgroup's copy of T->fft_regtw_l, at this lane.
template <int C> MP3MI_DEVFN void fft_reg_long(typename fft_vec<C>::V (&x)[16], const uint32_t (&smp)[16], const float *window,
                                               const uint4 *rt, int lane)
{
    typedef fft_vec<C> F;
    uint4 tw[MP3MI_FFT_REG_ROWS_L];
    {
        float wl[16];
#pragma unroll
        for (int k = 0; k < 16; k++) wl[k] = window[lane + 64 * k];
#pragma unroll
        for (int k = 0; k < 16; k++) {
#pragma unroll
            for (int c = 0; c < C; c++)
                F::set(x[k], c, wl[k] * (float) (int) (int16_t) (c == 0 ? (smp[k] & 0xffffu) : (smp[k] >> 16))); // src/l3psy.c:485, src/psy.c:264
        }
    }
#pragma unroll
    for (int r = 0; r < MP3MI_FFT_REG_ROWS_L; r++) tw[r] = rt[64 * r];
    // R(1024): n = lane + 64 j; lane 0 of j = 2 is n = m / 8, the SQHALF rotation
    fft_reg4<C, 2>(x[0], x[8], x[4], x[12], tw[0]);
    fft_reg4<C, 2>(x[1], x[9], x[5], x[13], tw[1]);
    fft_reg4<C, 6>(x[2], x[10], x[6], x[14], tw[2]);
    fft_reg4<C, 2>(x[3], x[11], x[7], x[15], tw[3]);
    // R(512) on elements 0..511 (n = m / 8 = 64: lane 0 of j = 1)
    fft_reg4<C, 2>(x[0], x[4], x[2], x[6], tw[4]);
    fft_reg4<C, 6>(x[1], x[5], x[3], x[7], tw[5]);
    // C(256): xr = elements 512..767, xi = 768..1023, n = lane (n = m / 8 = 32)
    {
        fft_round<C, 7> r;
        r.v[0] = x[8]; r.v[1] = x[10]; r.v[2] = x[9]; r.v[3] = x[11];
        r.v[4] = x[12]; r.v[5] = x[14]; r.v[6] = x[13]; r.v[7] = x[15];
        r.tw1 = tw[6]; r.tw3 = tw[7];
        r.compute();
        x[8] = r.v[0]; x[10] = r.v[1]; x[9] = r.v[2]; x[11] = r.v[3];
        x[12] = r.v[4]; x[14] = r.v[5]; x[13] = r.v[6]; x[15] = r.v[7];
    }
    // R(256) on elements 0..255
    fft_reg4<C, 6>(x[0], x[2], x[1], x[3], tw[8]);
}

// The blocks of 8 points and fewer IN REGISTERS, behind the program (tables_host.cpp, FftGen::in_leaf): a lane takes two runs of 8
// consecutive elements, A and B -- four 16-byte LDS reads each, a pair of elements (all channels) per read -- and runs what is left
// of the recursion on them: C(8) with all it spawns (kind 0), the two C(4) of a block C(16) (kind 1), or R(8) and the C(4) beside
// it (kind 2: one lane a transform).  The same arithmetic as a round of the program (fft_round::compute), with the rotation flags
// known at compile time; seven rounds of the long program and five of the short one -- two dependent LDS round trips each, on
// operands that no placement spreads over the banks (profiles/r06_experiments.txt, F10) -- become one.
template <int C> struct __attribute__((aligned(8 * C))) fft_elem_pair { typename fft_vec<C>::V lo, hi; };
template <int C, int H, uint32_t FLAGS>
MP3MI_DEVFN void fft_leaf4(typename fft_vec<C>::V &a, typename fft_vec<C>::V &b, typename fft_vec<C>::V &c, typename fft_vec<C>::V &d)
{
    uint4 tw = {0, 0, 0, FLAGS};
    fft_reg4<C, H>(a, b, c, d, tw);
}
template <int C, int H, uint32_t FLAGS>
MP3MI_DEVFN void fft_leaf8(typename fft_vec<C>::V &r0, typename fft_vec<C>::V &r1, typename fft_vec<C>::V &r2, typename fft_vec<C>::V &r3,
                           typename fft_vec<C>::V &i0, typename fft_vec<C>::V &i1, typename fft_vec<C>::V &i2, typename fft_vec<C>::V &i3)
{
    fft_round<C, H | 1> r;
    r.v[0] = r0; r.v[1] = r1; r.v[2] = r2; r.v[3] = r3; r.v[4] = i0; r.v[5] = i1; r.v[6] = i2; r.v[7] = i3;
    r.tw1 = uint4{0, 0, 0, FLAGS};
    r.tw3 = uint4{0, 0, 0, 0};
    r.compute();
    r0 = r.v[0]; r1 = r.v[1]; r2 = r.v[2]; r3 = r.v[3]; i0 = r.v[4]; i1 = r.v[5]; i2 = r.v[6]; i3 = r.v[7];
}
// (a pair of elements is read as ONE aligned access: the wavefronts' arrays must start and repeat on multiples of its size)
static_assert(sizeof(fft_wave_lds<2, true>) % 16 == 0 && sizeof(fft_wave_lds<2, false>) % 16 == 0 && sizeof(fft_wave_lds<1, true>) % 8 == 0 &&
              sizeof(fft_wave_lds<1, false>) % 8 == 0, "fft_leaves: a wavefront's array is not a whole number of element pairs");
typedef fft_lds<2, 12, true> fft_lds_2l;
typedef fft_lds<2, 16, false> fft_lds_2s;
typedef fft_lds<1, 16, true> fft_lds_1l;
typedef fft_lds<1, 16, false> fft_lds_1s;
static_assert(__builtin_offsetof(fft_lds_2l, w) % 16 == 0 && __builtin_offsetof(fft_lds_2s, w) % 16 == 0 && __builtin_offsetof(fft_lds_1l, w) % 8 == 0 &&
              __builtin_offsetof(fft_lds_1s, w) % 8 == 0, "fft_leaves: the wavefronts' arrays do not start on a pair boundary");
template <int C>
MP3MI_DEVFN void fft_leaves(char *xw, uint32_t woff, const uint4 *leaf_tab, int lane)
{
    typedef typename fft_vec<C>::V V;
    typedef fft_elem_pair<C> P;
    const uint4 ad = leaf_tab[lane];
    const uint32_t kind = (ad.x >> 14) & 3u;
    if (kind == 3u) return; // an idle lane (the three short transforms fill 48)
    const uint32_t ax = ad.x & 0x3fff3fffu;
    P *p[8];
    p[0] = (P *) fft_at<C, 0>(xw, woff, ax);   p[1] = (P *) fft_at<C, 1>(xw, woff, ax);
    p[2] = (P *) fft_at<C, 0>(xw, woff, ad.y); p[3] = (P *) fft_at<C, 1>(xw, woff, ad.y);
    p[4] = (P *) fft_at<C, 0>(xw, woff, ad.z); p[5] = (P *) fft_at<C, 1>(xw, woff, ad.z);
    p[6] = (P *) fft_at<C, 0>(xw, woff, ad.w); p[7] = (P *) fft_at<C, 1>(xw, woff, ad.w);
    V a[8], b[8];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const P pa = *p[j], pb = *p[4 + j];
        a[2 * j] = pa.lo; a[2 * j + 1] = pa.hi;
        b[2 * j] = pb.lo; b[2 * j + 1] = pb.hi;
    }
    // (the order of FftGen::leaf_ops: what a butterfly reads, the ones before it have written)
    if (kind == 0u) { // C(8), src/subs.c:288-342 for n = 0 and n = m / 8 = 1
        fft_leaf8<C, 0, 0u>(a[0], a[4], a[2], a[6], b[0], b[4], b[2], b[6]);
        fft_leaf8<C, 4, 2u>(a[1], a[5], a[3], a[7], b[1], b[5], b[3], b[7]);
    } else if (kind == 1u) { // the second C(4)
        fft_leaf8<C, 0, 0u>(a[4], a[6], a[5], a[7], b[4], b[6], b[5], b[7]);
    }
    if (kind <= 1u) {
        fft_leaf8<C, 0, 0u>(a[0], a[2], a[1], a[3], b[0], b[2], b[1], b[3]); // C(4)
        fft_leaf4<C, 0, 0u>(a[0], a[1], b[0], b[1]);                         // C(2), src/subs.c:243-250
        fft_leaf4<C, 0, 0u>(a[4], a[5], b[4], b[5]);
        if (kind == 0u) fft_leaf4<C, 0, 0u>(a[6], a[7], b[6], b[7]);
    } else { // R(8) in A (src/subs.c:465-498 for n = 0, 1; then R(4), R(2) and the C(2) of its upper half), C(4) in B
        fft_leaf4<C, 0, 0x80000000u>(a[0], a[4], a[2], a[6]);
        fft_leaf4<C, 4, 0x80000002u>(a[1], a[5], a[3], a[7]);
        fft_leaf4<C, 0, 0x80000000u>(a[0], a[2], a[1], a[3]);
        { V zc = fft_vec<C>::splat(0.0f), zd = zc; fft_leaf4<C, 0, 0u>(a[0], a[1], zc, zd); } // R(2): its other half is the dummy element
        fft_leaf4<C, 0, 0u>(a[4], a[5], a[6], a[7]);
        fft_leaf8<C, 0, 0u>(b[0], b[2], b[1], b[3], b[4], b[6], b[5], b[7]);
        fft_leaf4<C, 0, 0u>(b[0], b[1], b[4], b[5]);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        P pa, pb;
        pa.lo = a[2 * j]; pa.hi = a[2 * j + 1];
        pb.lo = b[2 * j]; pb.hi = b[2 * j + 1];
        *p[j] = pa;
        *p[4 + j] = pb;
    }
}

// The sequence of rounds is a compile-time constant (MP3MI_FFT_HDRS_*, checked against the generator at
// table build): the program runs as straight-line code, every block a constant offset from the lane's
// record address, with no per-round dispatch.
static constexpr uint8_t fft_hdrs_l[] = {MP3MI_FFT_HDRS_L}, fft_hdrs_s[] = {MP3MI_FFT_HDRS_S};
constexpr int fft_round_words(int h) { return ((h & 1) ? 256 : 128) + ((h & 2) ? ((h & 1) ? 512 : 256) : 64); }

template <int C, bool LONG, int R, int OFF>
MP3MI_DEVFN void fft_run(char *xw, uint32_t woff, const uint32_t *prog, int lane)
{
    constexpr int NL = (int) sizeof(fft_hdrs_l), NS = (int) sizeof(fft_hdrs_s), NR = LONG ? NL : NS;
    if constexpr (R < NR) {
        constexpr int h = LONG ? fft_hdrs_l[R < NL ? R : 0] : fft_hdrs_s[R < NS ? R : 0];
        if constexpr ((h & 8) == 0 && R + 1 < NR) { // the next round belongs to the same rank: both in flight together
            constexpr int h2 = LONG ? fft_hdrs_l[R + 1 < NL ? R + 1 : 0] : fft_hdrs_s[R + 1 < NS ? R + 1 : 0];
            fft_round<C, h & 7> a;
            fft_round<C, h2 & 7> b;
            a.fetch(xw, woff, prog + OFF, lane);
            b.fetch(xw, woff, prog + OFF + fft_round_words(h), lane);
            a.finish();
            b.finish();
            if constexpr ((h2 & 8) != 0) wave_sync(); // the next rank reads what this one wrote
            fft_run<C, LONG, R + 2, OFF + fft_round_words(h) + fft_round_words(h2)>(xw, woff, prog, lane);
        } else {
            fft_round<C, h & 7> a;
            a.fetch(xw, woff, prog + OFF, lane);
            a.finish();
            if constexpr ((h & 8) != 0) wave_sync();
            fft_run<C, LONG, R + 1, OFF + fft_round_words(h)>(xw, woff, prog, lane);
        }
    }
}

// energy of a bin for all C channels (src/subs.c:53-123); x points at element 0 of the transform and rd is
// the bin's read-out word (fft_rd_*: where its real and imaginary part ended up).  Branch-free: bins 0 and
// N/2 are real (`real`; their "imaginary" operand is the real one again and not used).  The reference's
// floor test `(double) e < 0.0005` is the same as the float test against (float) 0.0005 = 0x3a03126f, the
// float next ABOVE 0.0005: below it both say yes, above it both say no, and at it the replacement value is
// e itself.  Signs do not matter here.
template <int C>
MP3MI_DEVFN fft_pair<C> fft_energy(const float *x, uint32_t rd, bool real)
{
    const fft_pair<C> re = *(const fft_pair<C> *) (x + (rd & 0x7fffu) * C);
    const fft_pair<C> im = *(const fft_pair<C> *) (x + ((rd >> 16) & 0x7fffu) * C);
    fft_pair<C> e;
#pragma unroll
    for (int c = 0; c < C; c++) {
        const float rr = re.c[c] * re.c[c];
        const float ee = rr + im.c[c] * im.c[c];
        const float fl = ee < (float) 0.0005 ? (float) 0.0005 : ee;
        e.c[c] = real ? rr : fl;
    }
    return e;
}

// raw value of a bin with the sign the reference's step 5 leaves it with (src/subs.c:506-523)
template <int C>
MP3MI_DEVFN void fft_bin(const float *x, uint32_t rd, fft_pair<C> *re, fft_pair<C> *im)
{
    *re = *(const fft_pair<C> *) (x + (rd & 0x7fffu) * C);
    *im = *(const fft_pair<C> *) (x + ((rd >> 16) & 0x7fffu) * C);
#pragma unroll
    for (int c = 0; c < C; c++) {
        re->c[c] = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, re->c[c]) ^ ((rd & 0x8000u) << 16));
        im->c[c] = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, im->c[c]) ^ (rd & 0x80000000u));
    }
}

// N x 64 consecutive samples (all channels of a sample in one word) from time t_first on, sample lane + 64 k
// in smp[k]; t is relative to the call's first sample.  Before it (t < 0, at most MP3MI_PCM_HIST back) the stream's
// history buffer is read -- zeros at the start of a stream, the previous call's last samples after that -- and from
// n_per_ch on the stream reads as zero.  t_first is wave-uniform: when the whole span lies inside the call's
// samples -- all but its first and last granules -- the loads need no per-sample tests.
template <int C, int N>
MP3MI_DEVFN void fft_load_pcm(const int16_t *pcm, const int16_t *hist, long t_first, long n_per_ch, int lane, uint32_t (&smp)[N],
                              const int hlen = MP3MI_PCM_HIST)
{
    if (t_first >= 0 && t_first + 64 * N <= n_per_ch) {
        const int16_t *p = pcm + t_first * C;
#pragma unroll
        for (int k = 0; k < N; k++) {
            if (C == 2) smp[k] = ((const uint32_t *) p)[lane + 64 * k];
            else smp[k] = (uint32_t) (uint16_t) p[lane + 64 * k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < N; k++) {
            const long t = t_first + lane + 64 * k;
            const bool in = t >= 0 && t < n_per_ch, past = hist && t < 0 && t >= -hlen;
            const long th = past ? t + hlen : 0;
            if (C == 2) smp[k] = in ? ((const uint32_t *) pcm)[t] : (past ? ((const uint32_t *) hist)[th] : 0u);
            else smp[k] = in ? (uint32_t) (uint16_t) pcm[t] : (past ? (uint32_t) (uint16_t) hist[th] : 0u);
        }
    }
}

// Two kernels, one for the 1024-point transform and one for the three 256-point ones (LONG), each with ITS
// program resident in LDS and W wavefronts that work through the (stream, granule) tasks on their own: a
// workgroup takes every gridDim.x-th batch of W tasks, the program is loaded once per workgroup, and after
// that no wavefront ever waits for another (one workgroup fills a CU's LDS, so a workgroup that started and
// ended together would leave the CU idle while the next one loads its program and samples).
template <int C, int W, bool LONG>
__global__ void __launch_bounds__(64 * W) k_fft(const mp3mi_tables *__restrict__ T, mp3mi_geom geo,
                                                const int16_t *__restrict__ pcm_all, float *__restrict__ energy_l,
                                                float *__restrict__ energy_s, float *__restrict__ bins)
{
    typedef fft_vec<C> F;
    __shared__ fft_lds<C, W, LONG> LL;
    const int lane = wave_lane(), tid = (int) threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6); // wave-uniform, and known to be
    fft_wave_lds<C, LONG> &L = LL.w[wv];
    const int G = geo.n_gran, n_task = geo.n_streams * G;
    const long n_pitch = (long) geo.n_frames * 1152; // row pitch of the PCM buffer
    const int lane_swz = MP3MI_FFT_SWZ(lane);
    constexpr int NS = LONG ? 16 : 8; // words of samples a lane holds per task
    PROF_DECL;
    fft_lds_fill<C, W, LONG>(LL, T, tid);

    // A task's samples: the 1024-sample window from savebuf[0] on (src/l3psy.c:477-485), or -- the short windows are samples
    // 256 + 128 sb + jj, sb < 3 (src/l3psy.c:520-523) -- the 512 samples from savebuf[256] on.  They are asked for a task
    // AHEAD, in front of the read-out of the task before: what a wavefront waited for at the top of every task was this trip
    // to memory (a seventh of its time), and behind the program the registers to hold sixteen words are there.
    auto task_of = [&](int batch) { const int t = batch * W + wv; return t < n_task ? t : n_task - 1; };
    auto load_task = [&](int task, uint32_t (&smp)[NS]) {
        const int gl = task % G, s = task / G;
        const long n_per_ch = geo.n_samples ? (long) geo.n_samples[s] : n_pitch; // valid samples: the rest reads as zero (src/encode.c:162-166)
        const int16_t *pcm = pcm_all + (size_t) s * (size_t) n_pitch * (size_t) C;
        const int16_t *hist = geo.hist ? geo.hist + (size_t) s * MP3MI_PCM_HIST * (size_t) C : NULL;
        const long t0 = 576 * ((long) geo.g0 + gl) - 768; // time of savebuf[0]  (src/l3psy.c:477-481)
        fft_load_pcm<C, NS>(pcm, hist, LONG ? t0 : t0 + 256, n_per_ch, lane, smp);
    };
    uint32_t smp[NS];
    if ((int) blockIdx.x * W < n_task) load_task(task_of((int) blockIdx.x), smp);
    __syncthreads();
    PROF(LONG ? 0 : 4);

    for (int batch = (int) blockIdx.x; batch * W < n_task; batch += (int) gridDim.x) {
        const bool valid = batch * W + wv < n_task; // the last batch may have idle wavefronts: they compute, but do not store
        const int task = task_of(batch);
        const int gl = task % G, s = task / G;
        const size_t rec0 = ((size_t) s * G + gl) * C;
        const int next = batch + (int) gridDim.x;
        const bool more = next * W < n_task; // (workgroup-uniform)

        if constexpr (LONG) {
            {
                typename F::V x[16];
                fft_reg_long<C>(x, smp, LL.win, LL.regtw + lane, lane);
#if !defined(MP3MI_FFT_EXP_NO_STORE) // (diagnostic builds, tools/gpu_fft_conflicts.sh: one phase's LDS traffic taken out, results wrong, counters telling)
#pragma unroll
                for (int k = 0; k < 16; k++)
                    *(typename F::V *) (L.x + (lane_swz ^ MP3MI_FFT_SWZ(64 * k)) * C) = x[k]; // == SWZ(lane + 64 k): the map is linear
#else
                if (lane == 99) L.x[0] = F::get(x[0], 0) + F::get(x[15], 0);
#endif
            }
            wave_sync();
            PROF(1);
#if !defined(MP3MI_FFT_EXP_NO_PROG)
            fft_run<C, true, 0, 0>((char *) &LL.w[0], (uint32_t) (wv * (int) sizeof(fft_wave_lds<C, true>)), LL.prog, lane);
            fft_leaves<C>((char *) &LL.w[0], (uint32_t) (wv * (int) sizeof(fft_wave_lds<C, true>)), LL.leaf, lane);
            wave_sync();
#endif
            PROF(2);
            if (more) load_task(task_of(next), smp);
            // energies of the 513 lines: the read-out words of all nine steps first, then the spectrum, then the stores
            float *el0 = energy_l + rec0 * MP3MI_HBLK_P;
            uint32_t rdw[9];
#pragma unroll
            for (int k = 0; k < 9; k++) rdw[k] = LL.rd[lane + 64 * k < MP3MI_HBLK ? lane + 64 * k : 0];
            fft_pair<C> e[9];
#if !defined(MP3MI_FFT_EXP_NO_READOUT)
#pragma unroll
            for (int k = 0; k < 9; k++) e[k] = fft_energy<C>(L.x, rdw[k], lane + 64 * k == 0 || lane + 64 * k == 512);
#else
#pragma unroll
            for (int k = 0; k < 9; k++) for (int c = 0; c < C; c++) e[k].c[c] = (float) rdw[k];
#endif
            if (valid) {
#pragma unroll
                for (int k = 0; k < 9; k++) {
                    if (k < 8 || lane == 0) {
#pragma unroll
                        for (int c = 0; c < C; c++) el0[c * MP3MI_HBLK_P + lane + 64 * k] = e[k].c[c];
                    }
                }
            }
            // raw bins 0..5 for k_cw: re, im (bin 0 is real: im = -0 makes atan2(-im, re) the reference's atan2(0.0, x[0]))
            if (lane < 6 && valid) {
                fft_pair<C> re, im;
                fft_bin<C>(L.x, rdw[0], &re, &im);
#pragma unroll
                for (int c = 0; c < C; c++) {
                    bins[(rec0 + c) * MP3MI_FFT_BINS + 300 + lane] = re.c[c];
                    bins[(rec0 + c) * MP3MI_FFT_BINS + 306 + lane] = lane ? im.c[c] : -0.0f;
                }
            }
            wave_sync(); // the spectrum is dead: the next task's samples take its place
            PROF(3);
        } else {
            // Element 64 q + lane of window sb is sample 64 (2 sb + q) + lane (the second half of every window is also the
            // first half of the next): four registers a window, and R(256) of each window -- one rotation row for the
            // three -- runs on them before they go to LDS (fft_reg4).
            {
                float wsv[4];
#pragma unroll
                for (int k = 0; k < 4; k++) wsv[k] = LL.win[lane + 64 * k];
                const uint4 rtw = LL.regtw[lane];
#pragma unroll
                for (int sb = 0; sb < 3; sb++) {
                    typename F::V x[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const uint32_t w = smp[2 * sb + q];
#pragma unroll
                        for (int c = 0; c < C; c++) F::set(x[q], c, wsv[q] * (float) (int) (int16_t) (c == 0 ? (w & 0xffffu) : (w >> 16)));
                    }
                    fft_reg4<C, 6>(x[0], x[2], x[1], x[3], rtw);
                    // (element sb * 256 + 64 q + lane of the 768-element array of the three windows; the map is linear)
#pragma unroll
                    for (int q = 0; q < 4; q++) *(typename F::V *) (L.x + (lane_swz ^ MP3MI_FFT_SWZ(sb * 256 + 64 * q)) * C) = x[q];
                }
            }
            wave_sync();
            PROF(5);
            fft_run<C, false, 0, 0>((char *) &LL.w[0], (uint32_t) (wv * (int) sizeof(fft_wave_lds<C, false>)), LL.prog, lane);
            fft_leaves<C>((char *) &LL.w[0], (uint32_t) (wv * (int) sizeof(fft_wave_lds<C, false>)), LL.leaf, lane);
            wave_sync();
            PROF(6);
            if (more) load_task(task_of(next), smp);
            // energies of the three short spectra (bin k of window sb, both channels per LDS read) and the raw
            // short lines 2..51 for k_cw (src/l3psy.c:531-549 reads these only); plain nested loops, no div/mod
            uint32_t rds[3];
#pragma unroll
            for (int t = 0; t < 3; t++) rds[t] = LL.rd[lane + 64 * t < MP3MI_HBLK_S ? lane + 64 * t : 0];
            const uint32_t rdb = LL.rd[lane < 50 ? 2 + lane : 0];
#pragma unroll
            for (int sb = 0; sb < 3; sb++) {
                float *es0 = energy_s + rec0 * (3 * MP3MI_HBLK_S) + sb * MP3MI_HBLK_S;
                const float *xw = L.x + sb * 256 * C;
                // the read-out table holds window 0's positions; window sb's differ by the swizzle of its offset
                const uint32_t wsw = (uint32_t) (MP3MI_FFT_SWZ(sb * 256) ^ (sb * 256)) * 0x10001u;
#pragma unroll
                for (int t = 0; t < 3; t++) {
                    const int k = lane + 64 * t;
                    if (k < MP3MI_HBLK_S) {
                        const fft_pair<C> e = fft_energy<C>(xw, rds[t] ^ wsw, k == 0 || k == 128);
                        if (valid) {
#pragma unroll
                            for (int c = 0; c < C; c++) es0[c * (3 * MP3MI_HBLK_S) + k] = e.c[c];
                        }
                    }
                }
                if (lane < 50 && valid) {
                    fft_pair<C> re, im;
                    fft_bin<C>(xw, rdb ^ wsw, &re, &im);
#pragma unroll
                    for (int c = 0; c < C; c++) {
                        float *o = bins + (rec0 + c) * MP3MI_FFT_BINS + (sb * 50 + lane) * 2;
                        o[0] = re.c[c];
                        o[1] = im.c[c];
                    }
                }
            }
            wave_sync(); // the spectra are dead: the next task's samples take their place
            PROF(7);
        }
    }
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
#if defined(MP3MI_ULP_CENSUS) && !defined(MP3MI_EMU)
        if (!EXACT && y != 0.0 && x != 0.0) { // site 1: phi = (float) atan2(-im, re); a libm within one ulp of the exact value
            const long long dd = dm_float_midpoint_distance_ulps(dm_atan2(y, x));
            ULP_CENSUS(UC_PHASE, dd <= 1, dd <= (1LL << 20));
        }
#endif
        if (EXACT || y == 0.0 || x == 0.0) ph = (float) dm_atan2(y, x);
        else {
            const double v = dm_atan2_fast(y, x);
            if (!dm_float_rounding_safe(v)) *unsafe = true;
            ph = (float) v;
        }
    }
    *phi = ph;
}

// Phases and the unpredictability measure from the raw FFT bins of one (granule, channel) record by one wavefront:
// lanes 0..49 the unpredictability of lines 6+4n..9+4n from the three short FFTs (src/l3psy.c:531-549),
// lanes 50..55 magnitude and phase of long lines 0..5 (src/l3psy.c:497-503).  Kept out of k_fft so that
// this double-precision chain runs at full occupancy instead of next to 150 KB of LDS.
//
// The sines and cosines: c_w only ever reaches a bit through cb[b] += c_w * energy, a sum that is rounded to FLOAT at
// every step (src/l3psy.c:576, k_part).  FASTSC = true takes them from dm_sincos_fast (plain double, |error| < 2^-51);
// the c_w that comes out is within 6e-15 of the reference's (error chain: k_part, part_cw_safe), and k_part checks at
// every step that the float it rounds to cannot depend on that; the blocks of records where it could are repeated
// with FASTSC = false, the correctly rounded dm_sincos (k_cw_fix, then k_part again).
template <bool FASTSC>
MP3MI_DEVFN void cw_record(const float *__restrict__ bins, double *__restrict__ cw_mid, float *__restrict__ hist6, size_t rec, int force_exact)
{
    const int lane = wave_lane();
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
        if (FASTSC) {
            dm_sincos_fast(phi2, &s2, &c2);
            dm_sincos_fast(phi_prime, &sp, &cp);
        } else {
            dm_sincos(phi2, &s2, &c2);
            dm_sincos(phi_prime, &sp, &cp);
        }
        const double t1 = r2 * c2 - r_prime * cp;
        const double t2 = r2 * s2 - r_prime * sp;
        const double t3 = r2 + __builtin_fabs(r_prime);
        double cw = 0.0;
        if (t3 != 0.0) cw = __builtin_sqrt(t1 * t1 + t2 * t2) / t3;
        // Predicted and actual line bit for bit alike (digital silence: every energy at its floor, every phase 0; a
        // stationary bin): the reference subtracts a product from itself, whatever its libm returns for the sine --
        // its c_w is +0 exactly, as ours is.  -0.0 tells k_part that this zero is not a first-tier estimate (it adds
        // like +0.0 there): without it every record of a silent stream would go through the second tier.
        if (r2 == r_prime && phi2 == phi_prime) cw = -0.0;
        cw_mid[rec * 50 + lane] = cw;
    } else if (lane < 56 && FASTSC) { // (the same in both tiers: written once)
        hist6[rec * 12 + lane - 50] = (float) __builtin_sqrt((double) e[0]); // r, src/l3psy.c:500
        hist6[rec * 12 + 6 + lane - 50] = ph[0];
    }
}

// first tier: one wavefront per record.  exact_sc (MP3MI_CW_EXACT=1, tests): the second tier for every record.
__global__ void __launch_bounds__(64) k_cw(const float *__restrict__ bins, double *__restrict__ cw_mid, float *__restrict__ hist6,
                                          int force_exact, int exact_sc)
{
#if !defined(MP3MI_EMU)
    // beside k_loop (batch.cpp) this kernel gets what that one leaves: with the highest wave priority it is through in its
    // stand-alone time instead of three times that, and the chain k_cw -> k_part -> k_psy -> k_filter -> k_mdct ends before
    // the k_loop launch it runs beside does (profiles/r04_experiments.txt, prioA)
    __builtin_amdgcn_s_setprio(3);
#endif

    cw_record<true>(bins, cw_mid, hist6, blockIdx.x, force_exact);
    if (exact_sc) cw_record<false>(bins, cw_mid, hist6, blockIdx.x, force_exact);
}

// second tier: the records k_part listed, a wavefront each (a fixed grid walks the list)
__global__ void __launch_bounds__(64) k_cw_fix(const float *__restrict__ bins, double *__restrict__ cw_mid, float *__restrict__ hist6,
                                              int force_exact, const mp3mi_cw_fixlist *__restrict__ fix)
{
    const unsigned n_fix = fix->count < fix->cap ? fix->count : fix->cap;
    for (unsigned i = blockIdx.x; i < n_fix; i += gridDim.x) cw_record<false>(bins, cw_mid, hist6, fix->list[i], force_exact);
}

__global__ void __launch_bounds__(64) k_cw_fix_reset(mp3mi_cw_fixlist *fix, unsigned cap)
{
    if (threadIdx.x == 0) { fix->count = 0; fix->cap = cap; }
}

// Layers I and II (l12_dev.h): the 1024-point transform of every PASS of the psychoacoustic model for Layers I / II
// (src/psy.c:258-270: the same Hann window, the same fft() as Layer III's long transform), one wavefront per
// (stream, pass) task and all channels, with the same resident butterfly program.  Out: energy, magnitude and phase of all
// 513 lines per (pass, channel) record -- these layers take them of EVERY line (src/psy.c:282-292), not of 156.
template <int C, int W>
__global__ void __launch_bounds__(64 * W) k_fft12(const mp3mi_tables *__restrict__ T, l12_geom geo,
                                                  const int16_t *__restrict__ pcm_all, float *__restrict__ erp)
{
    __shared__ fft_lds<C, W, true> LL;
    const int lane = wave_lane(), tid = (int) threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    fft_wave_lds<C, true> &L = LL.w[wv];
    const int NP = geo.np, n_task = geo.n_streams * NP;
    const long n_pitch = (long) geo.n_frames * geo.spf;
    const int lane_swz = MP3MI_FFT_SWZ(lane);
    const bool force_exact = (geo.test_flags >> 1) & 1;
    fft_lds_fill<C, W, true>(LL, T, tid);
    __syncthreads();
    for (int batch = (int) blockIdx.x; batch * W < n_task; batch += (int) gridDim.x) {
        int task = batch * W + wv;
        const bool valid = task < n_task;
        task = valid ? task : n_task - 1;
        const int qi = task % NP, s = task / NP;
        const size_t rec0 = ((size_t) s * NP + qi) * C;
        const long qr = (long) geo.f0 * geo.layer - geo.lb + qi; // the pass counted from the call's first, and from the stream's
        const long q = geo.fabs0 * geo.layer + qr;
        const long n_per_ch = geo.n_samples ? (long) geo.n_samples[s] : n_pitch;
        const int16_t *pcm = pcm_all + (size_t) s * (size_t) n_pitch * (size_t) C;
        const long t0 = (long) geo.spp * (qr + 1) - geo.span; // time of savebuf[0], from the call's first sample  (src/psy.c:258-262)
        const int16_t *hist = geo.hist ? geo.hist + (size_t) s * L12_PCM_HIST * (size_t) C : NULL;
        {
            typedef fft_vec<C> F;
            uint32_t smp[16];
            typename F::V x[16];
            fft_load_pcm<C, 16>(pcm, hist, t0, n_per_ch, lane, smp, L12_PCM_HIST);
            fft_reg_long<C>(x, smp, LL.win, LL.regtw + lane, lane); // windowing (src/psy.c:264) + the blocks of 256 points and more
#pragma unroll
            for (int k = 0; k < 16; k++) *(typename F::V *) (L.x + (lane_swz ^ MP3MI_FFT_SWZ(64 * k)) * C) = x[k];
        }
        wave_sync();
        fft_run<C, true, 0, 0>((char *) &LL.w[0], (uint32_t) (wv * (int) sizeof(fft_wave_lds<C, true>)), LL.prog, lane);
        fft_leaves<C>((char *) &LL.w[0], (uint32_t) (wv * (int) sizeof(fft_wave_lds<C, true>)), LL.leaf, lane);
        wave_sync();
        // energy, magnitude and phase of every line straight from the spectrum in LDS (src/subs.c:53-123, src/psy.c:285-286):
        // erp[rec] = {energy, r = (float) sqrt((double) energy), phi}, rows of L12_ROW floats.  The transform is bound by
        // the LDS pipe and leaves the vector pipe idle more than half of the time: the phases' double-precision chain runs
        // in that shadow (as a kernel of its own it took 1.6 times the transform).  Phases in two tiers, as k_cw's (cw_bin).
        // Bins 0 and 512 are real: im = -0 makes atan2(-im, re) the reference's atan2(0.0, x) (src/subs.c:58, 93).
#pragma unroll 1
        for (int k = 0; k < 9; k++) {
            const int i = lane + 64 * k;
            const bool on = i < L12_HBLK; // (k = 8: line 512 only)
            fft_pair<C> re, im;
            fft_bin<C>(L.x, LL.rd[on ? i : 0], &re, &im);
#pragma unroll
            for (int c = 0; c < C; c++) {
                const bool real = i == 0 || i == 512;
                const float rr = re.c[c], ii = real ? -0.0f : im.c[c];
                float e, ph;
                bool unsafe = force_exact;
                if (!unsafe) cw_bin<false>(rr, ii, real, &e, &ph, &unsafe);
                if (wave_any(unsafe && on)) cw_bin<true>(rr, ii, real, &e, &ph, &unsafe);
                if (valid && on && q >= 0) {
                    float *o = erp + (rec0 + c) * (3 * L12_ROW);
                    o[i] = e;
                    o[L12_ROW + i] = __builtin_sqrtf(e); // == (float) sqrt((double) e): 53 >= 2 * 24 + 2 bits, the double rounding is innocuous
                    o[2 * L12_ROW + i] = ph;
                }
            }
        }
        if (valid && q < 0) { // a pass before the stream's first sample: the reference's initial r = phi = 0 (src/psy.c:158-162)
            for (int i = lane; i < L12_HBLK; i += 64)
                for (int c = 0; c < C; c++) {
                    float *o = erp + (rec0 + c) * (3 * L12_ROW);
                    o[i] = 0.0f; o[L12_ROW + i] = 0.0f; o[2 * L12_ROW + i] = 0.0f;
                }
        }
        wave_sync(); // the spectrum is dead: the next task's samples take its place
    }
}

void mp3mi_launch_fft12(const mp3mi_tables *T, const l12_geom &g, const int16_t *pcm, float *erp, hipStream_t st)
{
    const int n_task = g.n_streams * g.np;
    int n_cu = 256, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount;
    if (g.channels == 2) {
        const int W = 12, nb = (n_task + W - 1) / W;
        hipLaunchKernelGGL((k_fft12<2, W>), dim3((unsigned) (nb < n_cu ? nb : n_cu)), dim3(64 * W), 0, st, T, g, pcm, erp);
    } else {
        const int W = 16, nb = (n_task + W - 1) / W;
        hipLaunchKernelGGL((k_fft12<1, W>), dim3((unsigned) (nb < n_cu ? nb : n_cu)), dim3(64 * W), 0, st, T, g, pcm, erp);
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
                      float *energy_s, float *bins, double *cw_mid, float *hist6, hipStream_t st, int which)
{
    // which: bit 0 the two transforms, bit 1 k_cw (batch.cpp launches them apart: k_cw runs beside k_loop)
    // W: as many wavefronts as fit the 160 KB of LDS next to the shared program; one workgroup per CU (that
    // is all the LDS allows), each working through its share of the batches
    const int n_task = g.n_streams * g.n_gran;
    const int n_cu = mp3mi_current_cu_count();
    if (!(which & 1)) {
    } else if (g.channels == 2) {
        const int W = 12, WS = 16, nb = (n_task + W - 1) / W, nbs = (n_task + WS - 1) / WS;
        hipLaunchKernelGGL((k_fft<2, W, true>), dim3((unsigned) (nb < n_cu ? nb : n_cu)), dim3(64 * W), 0, st, T, g, pcm, energy_l, energy_s, bins);
        hipLaunchKernelGGL((k_fft<2, WS, false>), dim3((unsigned) (nbs < n_cu ? nbs : n_cu)), dim3(64 * WS), 0, st, T, g, pcm, energy_l, energy_s, bins);
    } else {
        const int W = 16, nb = (n_task + W - 1) / W, grid = nb < n_cu ? nb : n_cu;
        hipLaunchKernelGGL((k_fft<1, W, true>), dim3((unsigned) grid), dim3(64 * W), 0, st, T, g, pcm, energy_l, energy_s, bins);
        hipLaunchKernelGGL((k_fft<1, W, false>), dim3((unsigned) grid), dim3(64 * W), 0, st, T, g, pcm, energy_l, energy_s, bins);
    }
    if (which & 2) hipLaunchKernelGGL(k_cw, dim3((unsigned) (n_task * g.channels)), dim3(64), 0, st, bins, cw_mid, hist6, (g.test_flags >> 1) & 1, (g.test_flags >> 4) & 1);
}

// the second tier of the unpredictability for the records k_part listed (mp3mi_launch_psy)
void mp3mi_launch_cw_fix_reset(mp3mi_cw_fixlist *fix, unsigned cap, hipStream_t st)
{
    hipLaunchKernelGGL(k_cw_fix_reset, dim3(1), dim3(64), 0, st, fix, cap);
}
void mp3mi_launch_cw_fix(const mp3mi_geom &g, const float *bins, double *cw_mid, float *hist6, const mp3mi_cw_fixlist *fix, hipStream_t st)
{
    hipLaunchKernelGGL(k_cw_fix, dim3(2048), dim3(64), 0, st, bins, cw_mid, hist6, (g.test_flags >> 1) & 1, fix);
}

ULP_CENSUS_ACCESSOR(mp3mi_debug_ulp_census_fft)
