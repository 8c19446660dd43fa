// Iteration loop: allowed distortion, scfsi, bit reservoir, quantiser step search, Huffman
// table selection and bit counting, noise calculation and scalefactor amplification.
//
// Replaces iteration_loop (src/loop.c:232-362) with everything below it (src/loop.c:369-2140,
// src/pow_nint.h, src/reservoir.c) for a whole batch.  The search is serial per stream (the
// reservoir size threads through every granule), so ONE WAVEFRONT OWNS ONE STREAM and walks its
// frames in order; the 64 lanes share each granule's 288 PAIRS of lines (pair p = lines 2p, 2p + 1 -- what the
// Huffman tables code together -- belongs to lane p%64, slot p/64; slot 4 is half empty), all loop
// decisions are wave-uniform, bit counts are wave-reduced (DPP, several reductions in lock-step).
// The spectrum and the quantised values live in LDS; what stays in registers between passes is
// |xr|^(3/4) of the lane's ten lines and the per-band state.
//
// The stateless head of the loop (calc_xmin, quantanf_init, the values calc_scfsi stores) comes
// from k_mdct's tail (k_fbmdct.hip; k_prep.hip).  Band noise -- only ever compared with the allowed distortion -- is summed in
// ~10-line parts by all lanes, with the reference's sequential order as the fallback when a band
// lands within 1e-12 of its threshold.  Integer work is order-free.
//
// The kernel is instruction-issue-bound: what matters is the total instruction count of
// a pass and keeping memory operations out of its dependent chain (no scratch access inside the
// pass loops: lane-derived addresses are recomputed where used, see wave_lane_here).
//
// Small tables the search consults with wave-uniform indices (scalefactor band edges, Huffman
// table geometry, the region subdivision table) live one entry per lane in registers and are
// read with v_readlane; per-line Huffman code lengths sit in LDS; the quantiser boundary table
// and i^(4/3) are read through L1/L2.
//
// HBM per (granule, channel): 4608 B xr in, 472 B psy record and 472 B prep record in, 1152 B ix out,
// ~54 words of side information out.
#include "mp3mi_host.h"
#include "dmath.h"
#include <stdlib.h>

typedef struct {
    int32_t ResvSize;
    int32_t sc_en_tot[2][2], sc_en[2][2][21], sc_xm[2][2][21], sc_xrmax[2][2];
    int32_t addr[2][2][3];
    int32_t ref_abort; // last word (MP3MI_LOOP_STATE_ABORT_WORD): 0, or MP3MI_DEV_ABORT_* | frame << 8 -- an input the reference dies on
} mp3mi_loop_state;

struct loop_gr { // wave-uniform working copy of gr_info (src/l3side.h:60-87)
    int part2_3_length, big_values, count1, scalefac_compress;
    int wsf, block_type, table_select[3], region0_count, region1_count;
    int preflag, count1table_select, part2_length;
    int sfb_lmax, sfb_smax, address1, address2, address3, q;
};

// 7.6 KB per wavefront plus one 1.9 KB copy of the code-length tables per workgroup of LOOP_W = 4 wavefronts: the 16
// wavefronts per CU that 4096 streams on 256 CUs need take 129 KB of its 160 KB of LDS, and at 80 VGPRs 320 of a SIMD's
// 512 registers -- the feed-forward kernels of the next chunk find room beside them (batch.cpp).
struct loop_lds {
    double xr[576 + 2] __attribute__((aligned(16))); // the granule's spectrum (amplified in place; [576] = [577] = 0: the pair a finished noise job
                      // reads on, ix[576] = ix[577] = 0 too): kept here, not in registers, because the
                      // quantise/count passes only need |xr|^(3/4) (registers) and the 18 VGPRs decide
                      // between 4 wavefronts per SIMD with and without scratch spills
    int16_t ix[576 + 128] __attribute__((aligned(4))); // padded: the region walks read whole 64-pair steps and mask what lies past the end; pair p as one word
    int sf_gr0[2][21];
    // per-band state of the distortion loop, lane b = band b (long) / sfb * 3 + window (short): it is touched once per
    // iteration, between two runs of quantise+count passes, and lives here -- not in registers -- across them
    double band_xmin[36]; // allowed distortion (calc_xmin, doubled / pre-emphasised along the way)
    int band_sf[36];      // scalefactors of the iteration in progress
    int band_sfsave[36];  // and of the last iteration whose result stands (src/loop.c:505-519)
    mp3mi_loop_state st;
    // of the frame's side information only what a later granule or the end of the frame reads back; the records
    // themselves go straight to memory
    int p23[2][2];      // part2_3_length (ResvFrameEnd adds the stuffing bits, src/reservoir.c:190-224)
    int preflag0[2];    // granule 0's preflag (src/loop.c:1172-1176)
};

// The reference DIES on some inputs (an assert fails: tests/golden/coverage_notes.json, "reference_aborts").  A batch cannot
// die for one stream: the first such event is recorded in the stream's state (code | frame index << 8; wave-uniform, so
// it lives in a scalar register until the state goes back to memory), the search goes on with something harmless, and
// k_format / k_stream_tail void the stream's output (mp3mi.h, mp3mi_batch_stream_status).
#define LOOP_REF_ABORT(code, frame) do { if (ref_abort == 0) ref_abort = MP3MI_DEV_STATUS(code, frame); } while (0)

// Wavefronts (streams) per workgroup.  They share nothing but the code-length tables in LDS (1.9 KB that every stream
// would otherwise hold a copy of) and synchronise only per wavefront.
#define LOOP_W 4

// one entry per lane, read with wave_readlane_i32(reg, uniform index)
struct loop_regs {
    int desc_a;  // lane < 27: Huffman group descriptor of region maximum class `lane` (see loop_desc_index)
    int desc_b;  // lane < 27: offset of that group's cells in glut
};

__device__ static const int LOOP_PRETAB[21] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 3, 2};
// slen1 / slen2 of scalefac_compress k (src/loop.c:746-747: {0,0,0,0,3,1,1,1,2,2,2,3,3,3,4,4} and
// {0,1,2,3,0,1,2,3,1,2,3,1,2,3,2,3}) as nibble k of an immediate: a table in memory would be a dependent scalar
// load on the path between two passes
MP3MI_DEVFN int loop_slen1(int k) { return (int) ((0x4433322211130000ull >> (4 * k)) & 15ull); }
MP3MI_DEVFN int loop_slen2(int k) { return (int) ((0x3232132132103210ull >> (4 * k)) & 15ull); }

// Diagnostic build only (-DMP3MI_LOOP_PROFILE): cycles per phase, summed over all waves.
#if defined(MP3MI_LOOP_PROFILE) && !defined(MP3MI_EMU)
__device__ unsigned long long g_loop_prof[8];
__device__ unsigned long long g_loop_wave[2 * 65536];
__device__ unsigned long long g_loop_start[65536];
__device__ unsigned long long g_loop_work[65536]; // per stream: cycles from first to last instruction, HW_ID
__device__ unsigned long long g_cb_prof[8]; // phases inside loop_count_bits
#define CBPROF_ARG , unsigned long long *cbp
#define CBPROF_PASS , cb_acc
#define CBPROF_START unsigned long long cb_t = __builtin_amdgcn_s_memtime()
#define CBPROF(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); cbp[i] += n_ - cb_t; cb_t = n_; } while (0)
#define PROF_DECL unsigned long long cb_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long prof_t = __builtin_amdgcn_s_memtime(), prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; if (wave_lane() == 0 && block < 65536) g_loop_start[block] = __builtin_amdgcn_s_memrealtime()
#define PROF(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); prof_acc[i] += n_ - prof_t; prof_t = n_; } while (0)
#define PROF_END do { if (lane == 0) { for (int i_ = 0; i_ < 8; i_++) { atomicAdd(&g_loop_prof[i_], prof_acc[i_]); atomicAdd(&g_cb_prof[i_], cb_acc[i_]); } \
    if (s < 65536) { unsigned long long tot_ = 0; for (int i_ = 0; i_ < 8; i_++) tot_ += prof_acc[i_]; g_loop_wave[2 * s] = tot_; \
    g_loop_wave[2 * s + 1] = (unsigned long long) __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | ((unsigned long long) __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 32); } } } while (0)
#else
#define PROF_DECL
#define PROF(i)
#define PROF_END
#define CBPROF_ARG
#define CBPROF_PASS
#define CBPROF_START
#define CBPROF(i)
#endif

MP3MI_DEVFN int loop_nint(double in) { return (in < 0) ? (int) (in - 0.5) : (int) (in + 0.5); } // src/loop.c:2020

// ---- quantiser: ix = max{p in [0,2047] : tab[p] <= x}  (src/pow_nint.h:15-49, src/loop.c:1360-1428) ----
// A float estimate of x^(3/4)+0.4054 settles every line that is not within 2^-9 of a table
// boundary; the others are settled against the exact table.  The result never depends on the
// quality of the estimate.  Leaves the values in p[] and in L.ix (followed by a barrier).
// y34[j] = |xr[j]|^(3/4) in float (loop_power34), so that x^(3/4) = y34 * 2^(-3q/16) costs one multiply per pass.
// Only an estimate (the quantiser settles borderline lines exactly), so the raw 1-ulp hardware
// square root is enough; the correctly rounded expansion costs ~20 instructions per root.
// (LOOP_FAST_SQRTF / LOOP_FAST_EXP2F: mp3mi_dev.h; their error on the device is measured by k_debug.hip,
// mp3mi_debug_fastmath_bounds, and asserted by tests/test_gpu_tiers.py)
// Returns the largest y34 of the granule (wave-uniform): a step size that quantises it to zero
// quantises everything to zero.
// ---- which lines a lane owns ----
// Lane l holds pairs l, l + 64, .. l + 256 (LOOP_SLOTS slots; the pairs from 288 on -- slot 4 of lanes 32..63 -- do not exist:
// their y34 is 0, they quantise to 0, and what is stored for them lands in L.ix's padding).  Value j of a lane is line
// 2 * (l + 64 * (j / 2)) + j % 2.
#define LOOP_SLOTS 5
#define LOOP_NV (2 * LOOP_SLOTS)
MP3MI_DEVFN int loop_pair_of(int lane, int k) { return lane + 64 * k; }

MP3MI_DEVFN float loop_power34(const double xr[LOOP_NV], float y34[LOOP_NV])
{
    float m = 0.0f;
#pragma unroll
    for (int j = 0; j < LOOP_NV; j++) {
        const float a = (float) __builtin_fabs(xr[j]);
        y34[j] = LOOP_FAST_SQRTF(a * LOOP_FAST_SQRTF(a));
        m = y34[j] > m ? y34[j] : m;
    }
    return __builtin_bit_cast(float, wave_max_i32(__builtin_bit_cast(int, m))); // non-negative floats order like their bits
}

// |xr * sqrt(2)^n|^(3/4) = y34 * 2^(3n/8): when a band is amplified (n = 1) or pre-emphasised (n = pretab)
// the cached power is rescaled instead of taking two roots again.  One float rounding of the
// constant and one of the product per call: < 1.2e-7 relative (the quantiser's guard band budgets it).
MP3MI_DEVFN float loop_rescale34(float y34, int n)
{
    const float c = n == 1 ? 1.2968395546510096f : (n == 2 ? 1.681792830507429f : (n == 3 ? 2.1810154653305154f : 1.0f));
    return y34 * c;
}
// upper bound of the largest y34 after such a rescaling (the all-zero shortcut of the bisection reads it, which runs
// before the first amplification, and the quantiser, to know that no line reaches the table's end: loop_quantize)
#define LOOP_Y34MAX_GROW 2.1810157f
#define LOOP_Y34MAX_AMP 1.29684f /* one amplification (n = 1) */

MP3MI_DEVFN float loop_estimate(float y34, float cq) { return __builtin_fmaf(y34, cq, 0.4054f); } // of x^(3/4) + 0.4054

// true: every line quantises to 0 at this step (wave-uniform).  loop_estimate is monotone in y34 and lies above the upper
// estimate the quantiser rounds (see loop_quantize: (1 + 3.5e-6) y34 cq + 0.4054 + 2e-6 < 0.999 + 3.5e-6 + 2e-6 < 1 <=> below 0.5 after
// the quantiser's - 0.5), so the quantiser would leave every line at 0 unflagged.
MP3MI_DEVFN bool loop_all_zero(float y34max, int q)
{
    return loop_estimate(y34max, LOOP_FAST_EXP2F(-0.1875f * (float) q)) < 0.999f;
}

// What a pass needs to know about the quantised values besides the values themselves (which go to L.ix): the run
// lengths of calc_runlen for long blocks, the two region maxima of short blocks.  Taken while every value is still
// in a register, so that the values of a lane are never alive across the counting.
struct loop_qinfo {
    int n_nz, n_big; // lines up to the last non-zero PAIR / the last pair with a value above 1 (2 * pairs: even; 0 = none)
    int m1, m2;      // short blocks: this lane's maximum over lines [0, 36) / [36, 576)
};

// The quantiser's first tier, two lines -- a pair -- at a time.  With t = x^(3/4) + 0.4054 the answer is floor(t) (clamped to the
// table's end).  u = y34 * cq + 0.4054 - 0.5 estimates t - 0.5 with an error below 2.8e-6 y34 cq relative (the two 1-ulp roots,
// exp2, the roundings, at most 17 rescalings of y34 by loop_rescale34: 16 amplifications, one pre-emphasis) plus the roundings of
// what is computed here (the constants' products and the multiply-add: 2e-7 relative).  An UPPER and a LOWER estimate
//     u_hi = y34 * cq (1 + 3.5e-6) + (0.4054 - 0.5 + 2e-6),   u_lo = y34 * cq (1 - 3.5e-6) + (0.4054 - 0.5 - 2e-6)
// bracket t - 0.5, and v_cvt_pknorm_u16_f32 -- on a / 65535 it returns a rounded to the NEAREST integer, computed from the exact
// product and clamped to [0, 65535] (every float of the range against double arithmetic: mp3mi_debug_pknorm_bound, k_debug.hip;
// tests/test_gpu_tiers.py) -- rounds both, a pair per instruction, straight into the halves of the pair word x | y << 16:
//     n_hi >= u_hi - 0.5 > t - 1  =>  n_hi >= floor(t);     n_lo <= u_lo + 0.5 < t  =>  n_lo <= floor(t)
// (n_lo < t gives n_lo <= floor(t) also for an integer t).  So a line whose two roundings agree is settled, whatever the quality
// of the estimate; the others -- those within the band of a table boundary: small values, the common case, almost never --
// are settled against the exact table.  Five instructions per pair: two packed multiply-adds, two conversions, one compare-accumulate.
// scale = 1 / 65535: the conversion's; the constants are rounded ONCE, here, by the compiler.
#define LOOP_Q_HI ((1.0 + 3.5e-6) / 65535.0)
#define LOOP_Q_LO ((1.0 - 3.5e-6) / 65535.0)
#define LOOP_Q_CHI ((0.4054 - 0.5 + 2e-6) / 65535.0)
#define LOOP_Q_CLO ((0.4054 - 0.5 - 2e-6) / 65535.0)
struct loop_qscale { float a_hi, a_lo; };
MP3MI_DEVFN loop_qscale loop_quant_scale(int q)
{
    const float cq = LOOP_FAST_EXP2F(-0.1875f * (float) q);
    loop_qscale s = {cq * (float) LOOP_Q_HI, cq * (float) LOOP_Q_LO};
    return s;
}
MP3MI_DEVFN unsigned loop_quant_pair(float y0, float y1, float a, float c)
{
#if defined(MP3MI_EMU)
    return LOOP_PKNORM_U16(__builtin_fmaf(y0, a, c), __builtin_fmaf(y1, a, c));
#else
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 y = {y0, y1}, av = {a, a}, cv = {c, c};
    const f32x2 u = __builtin_elementwise_fma(y, av, cv); // v_pk_fma_f32: both lines in one instruction
    return LOOP_PKNORM_U16(u.x, u.y);
#endif
}

// ix = max{p in [0,2047] : tab[p] <= x}  (src/pow_nint.h:15-49, src/loop.c:1360-1428) for the lane's ten lines: the pair
// words go to L.ix (followed by a barrier).  y34[j] = |xr[j]|^(3/4) in float (loop_power34), so that x^(3/4) = y34 * 2^(-3q/16)
// costs one multiply per pass.  Only an estimate (the quantiser settles borderline lines exactly), so the raw 1-ulp hardware
// square root is enough; the correctly rounded expansion costs ~20 instructions per root.
// (LOOP_FAST_SQRTF / LOOP_FAST_EXP2F: mp3mi_dev.h; their error on the device is measured by k_debug.hip,
// mp3mi_debug_fastmath_bounds, and asserted by tests/test_gpu_tiers.py)
// force_exact (MP3MI_QUANT_EXACT=1, tests): every line is settled against the exact table, whatever the estimate says.
MP3MI_DEVFN loop_qinfo loop_quantize(const mp3mi_tables *T, loop_lds &L, const float y34[LOOP_NV], float y34max, int q, bool all_zero, bool force_exact, bool shortb)
{
    const int lane = wave_lane_here();
    unsigned *ixw = (unsigned *) L.ix; // pair p as one word: x | y << 16
    loop_qinfo qi = {0, 0, 0, 0};
    if (all_zero && !force_exact) {
#pragma unroll
        for (int k = 0; k < LOOP_SLOTS; k++) ixw[loop_pair_of(lane, k)] = 0u;
        wave_sync();
        return qi;
    }
    unsigned pw[LOOP_SLOTS];
    const loop_qscale qs = loop_quant_scale(q);
    unsigned differ = 0u; // some line's upper and lower estimate round to different integers
#pragma unroll
    for (int k = 0; k < LOOP_SLOTS; k++) {
        pw[k] = loop_quant_pair(y34[2 * k], y34[2 * k + 1], qs.a_hi, (float) LOOP_Q_CHI);
        const unsigned lo = loop_quant_pair(y34[2 * k], y34[2 * k + 1], qs.a_lo, (float) LOOP_Q_CLO);
        differ += pw[k] ^ lo; // (five terms below 2^28: no overflow; v_xad_u32)
    }
    // The rare tier.  From 2047.5 on the answer is the table's last entry; whether any line gets there is known from the granule's
    // largest y34 (an upper bound, wave-uniform): almost never -- so the clamps are not in the pass but in here, where the
    // lines whose two estimates differ are settled against the exact table.
    const bool over = !(loop_estimate(y34max, qs.a_hi * 65535.0f) < 2047.0f);
    if (force_exact || over || wave_any(differ != 0u)) {
        const double ostep = 1.0 / T->step[q - MP3MI_STEP_MIN];
#pragma unroll
        for (int j = 0; j < LOOP_NV; j++) {
            const int k = j >> 1, sh = 16 * (j & 1);
            // (the lower estimate again, here: nothing of it lives through the common path)
            unsigned lo = (loop_quant_pair(y34[2 * k], y34[2 * k + 1], qs.a_lo, (float) LOOP_Q_CLO) >> sh) & 0xffffu;
            int pp = (int) ((pw[k] >> sh) & 0xffffu);
            lo = lo < 2047u ? lo : 2047u;
            pp = pp < 2047 ? pp : 2047;
            const int line = 2 * loop_pair_of(lane, k) + (j & 1);
            if ((force_exact || lo != (unsigned) pp) && line < 576) {
                const double x = __builtin_fabs(L.xr[line]) * ostep;
                while (pp > 0 && x < T->pow_nint_tab[pp]) pp--;
                while (pp < 2047 && x >= T->pow_nint_tab[pp + 1]) pp++;
            }
            pw[k] = (pw[k] & ~(0xffffu << sh)) | ((unsigned) pp << sh);
        }
    }
#pragma unroll
    for (int k = 0; k < LOOP_SLOTS; k++) ixw[loop_pair_of(lane, k)] = pw[k];
    // Run lengths (calc_runlen, src/loop.c:1488-1520), by pairs: big_values and count1 only ask for the last PAIR that holds a
    // non-zero value / a value above 1 (loop_count_bits).  And the short-block maxima.
    if (shortb) { // (a short block's big_values is 288 whatever the values: no run lengths)
#pragma unroll
        for (int k = 0; k < LOOP_SLOTS; k++) {
            const int x = (int) (pw[k] & 0xffffu), y = (int) (pw[k] >> 16);
            const int v = x > y ? x : y;
            const bool low = loop_pair_of(lane, k) < 18; // lines [0, 36)
            qi.m1 = (low && v > qi.m1) ? v : qi.m1;
            qi.m2 = (!low && v > qi.m2) ? v : qi.m2;
        }
    } else {
        // min(x, 2) | min(y, 2) of the lane's five pairs as 2-bit fields of one word, slot k in bits 2k, 2k + 1: its highest set bit
        // names the lane's last non-zero pair, the highest set ODD bit its last pair with a value above 1.
        // (both halves of a pair word clamped to 2 by ONE packed instruction, the five words shifted together while x and y are
        // still apart -- the x codes in bits 0..9, the y codes in bits 16..25 --, then one OR of the halves: a field reads 0, 1, 2 or 3,
        // 3 = one value 1 beside one above 1, which the two questions below answer as they answer 2)
        const mp3mi_u16x2 two2 = {2, 2};
        unsigned wp = __builtin_bit_cast(unsigned, LOOP_PK_MIN_U16(__builtin_bit_cast(mp3mi_u16x2, pw[0]), two2));
#pragma unroll
        for (int k = 1; k < LOOP_SLOTS; k++)
            wp |= __builtin_bit_cast(unsigned, LOOP_PK_MIN_U16(__builtin_bit_cast(mp3mi_u16x2, pw[k]), two2)) << (2 * k);
        const unsigned w = (wp & 0xffffu) | (wp >> 16);
        // ONE reduction -- the OR of the words names the last slot that holds a non-zero pair / a pair with a value above 1 -- and a
        // lane mask per run length for the last lane of that slot.
        const unsigned wo = wave_or_u32(w);
        const int t_nz = 31 - __clz((int) wo), t_big = 31 - __clz((int) (wo & 0x2AAu)); // -1 for an empty word
        const int k_nz = t_nz >> 1, k_big = t_big >> 1;
        const unsigned long long m_nz = __ballot((w & (3u << ((2 * k_nz) & 31))) != 0u), m_big = __ballot((w & (2u << ((2 * k_big) & 31))) != 0u);
        qi.n_nz = t_nz < 0 ? 0 : 2 * (64 * k_nz + 64 - __clzll((long long) m_nz));
        qi.n_big = t_big < 0 ? 0 : 2 * (64 * k_big + 64 - __clzll((long long) m_big));
    }
    wave_sync();
    return qi;
}

// ---- Huffman table choice and code-length look-up, grouped ----
// new_choose_table (src/loop.c:1793-1897) only ever compares tables of one "group": {1},
// {2,3}, {5,6}, {7,8,9}, {10,11,12}, {13,15}, {15,24+}, {16+,24+}; within a group all tables have
// the same geometry.  The group is a function of the region maximum alone:
//   max < 16        -> entry max        of the descriptor table
//   max >= 16       -> entry 15 + bit length of (max - 15)   (the ht[].linmax thresholds of
//                      src/loop.c:1872-1888 are all of the form 2^k - 1)
// R.desc_a / R.desc_b hold that table one entry per lane (built in tables_host.cpp):
//   desc_a = c0 | c1 << 5 | c2 << 10 | ylen << 15 | linbits(c0) << 20 | linbits(c1) << 24
//   desc_b = offset of the group's cells in T->glut
// T->glut holds, per group and (x,y) cell, the code lengths of all its tables packed 5 bits
// each, so ONE LDS read prices a pair for every candidate.
#define GL_C1 897 /* count1 tables A | B << 5 */

MP3MI_DEVFN int loop_desc_index(int max)
{
    return max < 16 ? max : 15 + (32 - __clz(max - 15));
}

// descriptor of class `cls` (run once per wave, lane = cls): the candidate searches over ht[].xlen
// and ht[].linmax of src/loop.c:1812-1817, 1872-1888, 1919-1939 written as comparisons
MP3MI_DEVFN void loop_desc_init(const mp3mi_tables *T, int cls, int *desc_a, int *desc_b)
{
    *desc_a = 0;
    *desc_b = 0;
    if (cls == 0 || cls > 26) return;
    const int max = cls < 16 ? cls : 15 + (1 << (cls - 16));
    int c0, c1 = 0, c2 = 0, off, ylen;
    if (max < 15) {
        if (max <= 1) { c0 = 1; off = 0; ylen = 2; }
        else if (max == 2) { c0 = 2; c1 = 3; off = 4; ylen = 3; }
        else if (max == 3) { c0 = 5; c1 = 6; off = 13; ylen = 4; }
        else if (max <= 5) { c0 = 7; c1 = 8; c2 = 9; off = 29; ylen = 6; }
        else if (max <= 7) { c0 = 10; c1 = 11; c2 = 12; off = 65; ylen = 8; }
        else { c0 = 13; c1 = 15; off = 129; ylen = 16; }
    } else {
        const int m = max - 15;
        c0 = m <= 0 ? 15 : (m <= 1 ? 16 : (m <= 3 ? 17 : (m <= 7 ? 18 : (m <= 15 ? 19 : (m <= 63 ? 20 : (m <= 255 ? 21 : (m <= 1023 ? 22 : 23)))))));
        c1 = m <= 15 ? 24 : (m <= 31 ? 25 : (m <= 63 ? 26 : (m <= 127 ? 27 : (m <= 255 ? 28 : (m <= 511 ? 29 : (m <= 2047 ? 30 : 31))))));
        off = c0 == 15 ? 385 : 641;
        ylen = 16;
    }
    *desc_a = c0 | (c1 << 5) | (c2 << 10) | (ylen << 15) | ((int) T->ht_linbits[c0] << 20) | ((int) T->ht_linbits[c1] << 24);
    *desc_b = off;
}

// code lengths of pair (x, y), sign bits included (glut) and the linbits of src/loop.c:172-225 added,
// for the (up to) three tables of a group, spread to 10-bit fields.
// da/db: the group's descriptor words (may differ per lane).
MP3MI_DEVFN int loop_pair_cost3(const uint16_t *GL, int da, int db, int x, int y)
{
    const int xc = x > 15 ? 15 : x, yc = y > 15 ? 15 : y;
    const int nesc = (x > 14) + (y > 14);
    const int ylen = (da >> 15) & 31, lb = ((da >> 20) & 15) | (((da >> 24) & 15) << 10);
    const int e = GL[db + xc * ylen + yc];
    const int spread = (e & 31) | (((e >> 5) & 31) << 10) | (((e >> 10) & 31) << 20);
    return spread + nesc * lb;
}

// new_choose_table's decision from the candidates' bit sums (src/loop.c:1819-1897): '<=' moves to the later table among
// tables without linbits, '<' among the linbits pair -- taken on values that live in ONE LANE of vector registers (the sums as the reduction leaves them): the
// descriptor is taken into a vector register too, so that every instruction of the decision is a vector one.
// s01 = candidate 0's sum | candidate 1's << 16.  NC3: some region of the pass has a third candidate.
template <bool NC3>
MP3MI_DEVFN int loop_pick_v(int da, int s01, int s2, int *sum)
{
    int dav = da;
#if !defined(MP3MI_EMU)
    asm volatile("" : "+v"(dav));
#endif
    const int c0 = dav & 31, c1 = (dav >> 5) & 31;
    const int s0 = s01 & 0xffff, s1 = (int) ((unsigned) s01 >> 16);
    // the linbits pair compares with '<' (s1 + 1 <= s0), the others with '<='; no second candidate: never
    const int t1 = c1 ? s1 + (c0 >= 15 ? 1 : 0) : 0x7fffffff;
    const bool take1 = t1 <= s0;
    int best = take1 ? s1 : s0, choice = take1 ? c1 : c0;
    if (NC3) {
        const int c2 = (dav >> 10) & 31;
        const int t2 = c2 ? s2 : 0x7fffffff;
        const bool take2 = t2 <= best;
        best = take2 ? s2 : best;
        choice = take2 ? c2 : choice;
    }
    *sum = best;
    return choice;
}

// Cost of the pairs of lines [lo, hi) under the candidate tables of one group (descriptor dA/dB), per lane:
// *a01 = candidate 0 | candidate 1 << 16, *a2 = third candidate.  Walks 64 pairs per step straight out of
// L.ix.  ESC: the group's tables have linbits (x or y > 14 then costs them); NC3: it has a third candidate.
// (k15: the constant 15 in a scalar register -- loop_walk_consts --: compiled without the machine-level hoisting of loop
// invariants, a step would otherwise set it up again, in a vector register)
struct loop_walk_k { int k15; unsigned m1f; };
MP3MI_DEVFN loop_walk_k loop_walk_consts(void)
{
    loop_walk_k k = {15, 0x1f001fu};
#if !defined(MP3MI_EMU)
    asm volatile("" : "+s"(k.k15), "+s"(k.m1f));
#endif
    return k;
}
// byte address of the cell of pair word xy = x | y << 16 in a group whose values stay below 16: x * ylen2 + 2 y + dB2, as two
// multiply-adds that read the halves of the word where they lie (no unpacking, no clamp)
MP3MI_DEVFN unsigned loop_cell_of_word(unsigned xy, int ylen2, int dB2)
{
#if defined(MP3MI_EMU)
    return (xy & 0xffffu) * (unsigned) ylen2 + 2u * (xy >> 16) + (unsigned) dB2;
#else
    unsigned idx; // (one statement: between two the compiler puts a wait state it cannot know to be unnecessary)
    asm("v_mad_u32_u16 %0, %1, 2, %2 op_sel:[1,0,0,0]\n\tv_mad_u32_u16 %0, %1, %3, %0" : "=&v"(idx) : "v"(xy), "s"(dB2), "s"(ylen2));
    return idx;
#endif
}
// One step of a walk: the 64 pairs [w, w + 64) of a region, a pair word xy per lane.  LAST: the region's last step, which can reach past
// its end -- `inside` says whether this lane's pair still belongs to it; the others are priced as some cell all the same and masked out
// of the sums (a conditional load costs three scalar instructions and two branches per step).
template <bool ESC, bool NC3, bool LAST>
MP3MI_DEVFN void loop_walk_step(const uint16_t *GL, unsigned xy, bool inside, int ylen2, int dB2, int lb01, const loop_walk_k &K, int &s01, int &s2)
{
    unsigned at;
    if (ESC) { // values above 15 occur: clamped (which also keeps a pair past the end inside the table, whatever the padding holds)
        const int x = (int) (xy & 0xffffu), y = (int) (xy >> 16);
        const int xc = x > K.k15 ? K.k15 : x, yc = y > K.k15 ? K.k15 : y;
        unsigned cell = 2u * (unsigned) yc + (unsigned) dB2; // (kept apart: re-associated, the sum takes three instructions instead of two)
#if !defined(MP3MI_EMU)
        asm volatile("" : "+v"(cell));
        at = (unsigned) __umul24((unsigned) xc, (unsigned) ylen2) + cell; // (xc <= 15: a 24-bit multiply-add)
#else
        at = (unsigned) (xc * ylen2) + cell;
#endif
    } else { // the region's maximum is below 16: so is every value in it -- only a pair PAST its end can be larger, and counts as (0, 0)
        if (LAST) xy = inside ? xy : 0u;
        at = loop_cell_of_word(xy, ylen2, dB2);
    }
    unsigned e = *(const uint16_t *) ((const char *) GL + at);
    if (LAST) e = inside ? e : 0u;
    int c = (int) (((e << 11) | e) & K.m1f); // candidate 0's length | candidate 1's << 16: a shift-or and a mask
    if (ESC) c += (int) ((e >> 10) & 3u) * lb01; // (how many of x, y are escapes: in the cell, tables_host.cpp)
    s01 += c;
    if (NC3) s2 += (int) (e >> 10); // (a cell's bit 15 is clear: the third length needs no mask)
}

template <bool ESC, bool NC3>
MP3MI_DEVFN void loop_region_walk(const uint16_t *GL, const unsigned *ixw, int lane, int lo, int hi, int dA, int dB, const loop_walk_k &K, int *a01, int *a2)
{
    const int ylen2 = 2 * ((dA >> 15) & 31), dB2 = 2 * dB, lb01 = ((dA >> 20) & 15) | (((dA >> 24) & 15) << 16);
    int s01 = 0, s2 = 0;
    // Up to four whole steps of 64 pairs (a region holds at most 288) and a last, partial one: written out, every step at a constant
    // offset from the lane's first pair -- as a loop each step paid a pointer increment and two scalar additions of bookkeeping.
    const unsigned *p = ixw + (lo >> 1) + lane; // this lane's pair of the first step
    const int span = hi > lo ? hi - lo : 0, nfull = span >> 7, rest = (span & 127) >> 1; // (lo, hi even; a walk only runs on a region that holds a value: lo < hi)
    if (nfull > 0) {
        loop_walk_step<ESC, NC3, false>(GL, p[0], true, ylen2, dB2, lb01, K, s01, s2);
        if (nfull > 1) {
            loop_walk_step<ESC, NC3, false>(GL, p[64], true, ylen2, dB2, lb01, K, s01, s2);
            if (nfull > 2) {
                loop_walk_step<ESC, NC3, false>(GL, p[128], true, ylen2, dB2, lb01, K, s01, s2);
                if (nfull > 3) loop_walk_step<ESC, NC3, false>(GL, p[192], true, ylen2, dB2, lb01, K, s01, s2);
            }
        }
    }
    // pairs past the end of the region are read all the same (L.ix is padded) and masked out of the sums
    if (rest) loop_walk_step<ESC, NC3, true>(GL, p[64 * nfull], lane < rest, ylen2, dB2, lb01, K, s01, s2);
    *a01 = s01;
    *a2 = s2;
}

MP3MI_DEVFN void loop_region_cost(const uint16_t *GL, const unsigned *ixw, int lane, int lo, int hi, int m, int dA, int dB, const loop_walk_k &K, int *a01, int *a2)
{
    *a01 = 0;
    *a2 = 0;
    if (m == 0) return; // no table, no bits (src/loop.c:1771-1777)
    const bool esc = (dA >> 20) != 0, nc3 = ((dA >> 10) & 31) != 0; // tables with linbits never come in threes
    if (esc) loop_region_walk<true, false>(GL, ixw, lane, lo, hi, dA, dB, K, a01, a2);
    else if (nc3) loop_region_walk<false, true>(GL, ixw, lane, lo, hi, dA, dB, K, a01, a2);
    else loop_region_walk<false, false>(GL, ixw, lane, lo, hi, dA, dB, K, a01, a2);
}

// calc_runlen + count1_bitcount + subdivide + bigv_tab_select + bigv_bitcount
// (src/loop.c:1488-2014) on the freshly quantised values (pair words in registers, L.ix in LDS).
// Returns the Huffman bit count and fills g.  Written branch-free over the lanes: region
// membership is a predicate, never a divergent branch.
MP3MI_DEVFN int loop_count_bits(const mp3mi_tables *T, const loop_regs &R, loop_lds &L, const uint16_t *GL, loop_gr &g, const loop_qinfo &qi, bool all_zero CBPROF_ARG)
{
    CBPROF_START;
    const int lane = wave_lane_here();
#if defined(LOOP_EXP_SALU) && !defined(MP3MI_EMU) // experiment (profiles/r06_experiments.txt, F6): what does a pass pay for N more scalar / vector instructions?
    { int t_ = 0;
#pragma unroll
      for (int i_ = 0; i_ < LOOP_EXP_SALU; i_++) asm volatile("s_add_u32 %0, %0, 1" : "+s"(t_) : : "scc"); }
#endif
#if defined(LOOP_EXP_VALU) && !defined(MP3MI_EMU)
    { int t_ = lane;
#pragma unroll
      for (int i_ = 0; i_ < LOOP_EXP_VALU; i_++) asm volatile("v_add_u32_e32 %0, 1, %0" : "+v"(t_)); }
#endif
#if defined(LOOP_EXP_VALU3) && !defined(MP3MI_EMU)
    { int t_ = lane;
#pragma unroll
      for (int i_ = 0; i_ < LOOP_EXP_VALU3; i_++) asm volatile("v_add3_u32 %0, %0, %0, 1" : "+v"(t_)); }
#endif
    const bool shortb = g.wsf && g.block_type == 2;
    const unsigned *ixw = (const unsigned *) L.ix; // (x, y) of pair pr as one word: x | y << 16
    int bits = 0, nslot = 9; // nslot: slots (of 64 lines) that can hold a non-zero value
    int c1part = 0;          // this lane's part of the count1 region's bits: table A | table B << 16
    if (shortb) {
        g.count1 = 0;
        g.big_values = 288;
        g.count1table_select = 1; // count1_bitcount with no quadruples: sum0 == sum1 -> table B
    } else if (all_zero) {
        g.count1 = 0;
        g.big_values = 0;
        g.count1table_select = 1;
        nslot = 0;
    } else {
        const int hh[2] = {qi.n_nz, qi.n_big}; // from the quantiser, taken while the values were in registers
        // hh[0] = n: lines up to the last non-zero one, hh[1] = b <= n: lines up to the last one above 1.  The
        // reference's i = 2 * (top / 2 + 1) is n rounded up to even (0 for n = 0); everything is non-negative,
        // so the divisions are shifts (src/loop.c:1488-1520)
        const unsigned n_nz = (unsigned) hh[0], n_big = (unsigned) hh[1];
        nslot = (int) ((n_nz + 63u) >> 6);
        const unsigned i0 = (n_nz + 1u) & ~1u;
        const unsigned c1 = (i0 - n_big) >> 2;
        g.count1 = (int) c1;
        g.big_values = (int) ((i0 - 4u * c1) >> 1);
        // count1 region: table A vs table B (values are 0/1, so v+2w+4x+8y comes from two words)
        int s01 = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) { // at most 144 quadruples
            if (64 * k >= g.count1) break;
            const int qd = lane + 64 * k;
            const bool in = qd < g.count1;
            // (quadruples past the end are read all the same -- still inside this wavefront's LDS -- and masked out of the sum)
            const int w0 = g.big_values + 2 * qd;
            const unsigned a = ixw[w0], b = ixw[w0 + 1];
            // values 0 / 1 in the four halves of two words: v | w << 16 and x | y << 16 -> v + 2 w + 4 x + 8 y in three instructions
            const unsigned t = a | (b << 2);
            const int pp = (int) ((t | (t >> 15)) & 15u);
            const int e = GL[GL_C1 + pp]; // code length + sign bits: table A | table B << 5
            const int c = (e & 31) | ((e >> 5) << 16);
            s01 += in ? c : 0;
        }
        c1part = s01; // reduced further down, together with the region maxima
    }
    CBPROF(0); // run lengths + count1 region
    // subdivide (src/loop.c:1638-1706); address1..3 keep their old values when big_values == 0.
    // (Results go through plain locals and are assigned once: stores to the fields from several branches
    // made the compiler keep them in scratch memory.)
    {
        int r0c = 0, r1c = 0, ad1 = g.address1, ad2 = g.address2, ad3 = g.address3;
        if (g.big_values != 0) {
            const int bvr = 2 * g.big_values;
            if (g.wsf == 0) {
                // scfb_anz = number of band edges below bvr picks the counts from subdv_table, both lowered until
                // the regions end at or below bvr: a function of big_values alone, tabulated at table build
                const unsigned sd = T->subdiv_lut[g.big_values];
                r0c = (int) (sd & 15u);
                r1c = (int) ((sd >> 4) & 15u);
                ad1 = (int) ((sd >> 8) & 1023u);
                ad2 = (int) (sd >> 18);
                ad3 = bvr;
            } else {
                const bool sb = g.block_type == 2;
                r0c = sb ? 8 : 7;
                r1c = sb ? 36 : 13;
                ad1 = sb ? 36 : T->sfb_l[8]; // (36 at every MPEG-1 rate; start / stop blocks are rare)
                ad2 = bvr;
                ad3 = 0;
            }
        }
        g.region0_count = r0c; g.region1_count = r1c;
        g.address1 = ad1; g.address2 = ad2; g.address3 = ad3;
    }
    g.table_select[0] = g.table_select[1] = g.table_select[2] = 0;
    CBPROF(1); // subdivide
    if (nslot == 0) { // nothing but zeros: every region maximum is 0, no table, no bits
        g.count1table_select = 1; // count1_bitcount without quadruples: sum0 == sum1 -> table B
        return bits;
    }
    if (shortb) {
        // region maxima over lines [0,36) and [36,576); pair (6m+w, 6m+3+w), m<96, w<3
        int m1 = qi.m1, m2 = qi.m2; // this lane's part, from the quantiser
        {
            int mv[2] = {m1, m2};
            wave_reduce_i32<0, 2>(mv);
            m1 = mv[0]; m2 = mv[1];
        }
        // choose_table (src/loop.c:1908-1947): the first table that can hold the maximum
        const int da0 = wave_readlane_i32(R.desc_a, loop_desc_index(m1)), db0 = wave_readlane_i32(R.desc_b, loop_desc_index(m1));
        const int da1 = wave_readlane_i32(R.desc_a, loop_desc_index(m2)), db1 = wave_readlane_i32(R.desc_b, loop_desc_index(m2));
        const int t0 = m1 ? (da0 & 31) : 0, t1 = m2 ? (da1 & 31) : 0;
        g.table_select[0] = t0;
        g.table_select[1] = t1;
        int sum = 0;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int pr = lane + 64 * k;
            const bool in = pr < 288;
            const int m = in ? pr / 3 : 0, w = in ? pr - 3 * m : 0;
            const int x = L.ix[6 * m + w], y = L.ix[6 * m + 3 + w];
            const bool first = m < 6;
            const int c = loop_pair_cost3(GL, first ? da0 : da1, first ? db0 : db1, x, y) & 0x3ff;
            sum += (in && (first ? t0 : t1)) ? c : 0;
        }
        return wave_sum_i32(sum);
    }
    // long / start / stop blocks: regions [0,a1), [a1,a2), [a2,e2) (src/loop.c:1771-1777).  The
    // reference's enable tests (a1 > 0, a2 > a1, e2 > a2) are exactly "the range is not empty",
    // and region 2 is non-empty only right after subdivide set a1 <= a2 <= e2, so the three
    // ranges never overlap.  Lines from slot nslot on are zero, so the maxima may skip them; but a
    // region can reach past big_values (stale or clamped addresses) and a pair of zeros still has a
    // code length, so the pricing loop runs to the end of the last region.
    // (the addresses may come out of LDS: make their uniformity explicit, they bound the loops below)
    const int a1 = __builtin_amdgcn_readfirstlane(g.address1), a2 = __builtin_amdgcn_readfirstlane(g.address2);
    const int e2 = __builtin_amdgcn_readfirstlane(2 * g.big_values);
    // Region 0 = lines [0, a1), 1 = [a1, a2), 2 = [a2, e2).  Each region is walked on its own, 64
    // consecutive lines (32 pairs: one word each) per step straight out of L.ix, so a line is visited once,
    // by the region it belongs to, with that region's wave-uniform descriptor; lines from slot nslot on are
    // zero.  (Written out per region: arrays indexed by the region would live in scratch memory.)
    const int nzend = 64 * nslot;
    auto region_max = [&](int lo, int hi) {        hi = hi < nzend ? hi : nzend; // both even
        mp3mi_u16x2 m = {0, 0}; // the maxima of the x and of the y of this lane's pairs: one packed instruction a step
        const unsigned *p = ixw + (lo >> 1) + lane;
        const int span = hi > lo ? hi - lo : 0, nfull = span >> 7, rest = (span & 127) >> 1; // whole steps of 64 pairs (nothing to mask), and a last one
        if (nfull > 0) {
            m = LOOP_PK_MAX_U16(m, __builtin_bit_cast(mp3mi_u16x2, p[0]));
            if (nfull > 1) {
                m = LOOP_PK_MAX_U16(m, __builtin_bit_cast(mp3mi_u16x2, p[64]));
                if (nfull > 2) {
                    m = LOOP_PK_MAX_U16(m, __builtin_bit_cast(mp3mi_u16x2, p[128]));
                    if (nfull > 3) m = LOOP_PK_MAX_U16(m, __builtin_bit_cast(mp3mi_u16x2, p[192]));
                }
            }
        }
        if (rest) {
            // read unconditionally (L.ix is padded: pairs past the end exist) and mask: a conditional load costs
            // three scalar instructions and two branches per step
            const unsigned xy = lane < rest ? p[64 * nfull] : 0u;
            m = LOOP_PK_MAX_U16(m, __builtin_bit_cast(mp3mi_u16x2, xy));
        }
        return (int) (m.x > m.y ? m.x : m.y); // this lane's part
    };
    // the three region maxima and the count1 region's two bit sums: four reductions in lock-step
    int red[4] = {c1part, region_max(0, a1), region_max(a1, a2), region_max(a2, e2)};
    wave_reduce_i32<1, 3>(red);
    CBPROF(2); // region maxima + reduction
    {
        const int sum0 = red[0] & 0xffff, sum1 = (red[0] >> 16) & 0xffff; // table A vs table B (src/loop.c:1531-1580)
        if (sum0 < sum1) { g.count1table_select = 0; bits = sum0; }
        else { g.count1table_select = 1; bits = sum1; }
    }
    const int m0 = red[1], m1 = red[2], m2 = red[3];
    const int mx[3] = {m0, m1, m2};
    int da[3], db[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const int idx = loop_desc_index(mx[r]);
        da[r] = wave_readlane_i32(R.desc_a, idx);
        db[r] = wave_readlane_i32(R.desc_b, idx);
    }
    // cost of every candidate over its region: per lane three 10-bit partial sums per region.  A region can
    // reach past big_values (stale or clamped addresses) and a pair of zeros still has a code length, so the
    // pricing walks to the end of the region, not only over the non-zero slots.
    // Per region two partial sums per lane: candidates 0 and 1 in the halves of one word (the layout the
    // reduction wants), the third candidate -- only the groups {7,8,9} and {10,11,12} have one -- in another.
    int s01p[3], s2p[3];
    const loop_walk_k K = loop_walk_consts(); // (once a pass, not once a region)
    loop_region_cost(GL, ixw, lane, 0, a1, m0, da[0], db[0], K, &s01p[0], &s2p[0]);
    loop_region_cost(GL, ixw, lane, a1, a2, m1, da[1], db[1], K, &s01p[1], &s2p[1]);
    loop_region_cost(GL, ixw, lane, a2, e2, m2, da[2], db[2], K, &s01p[2], &s2p[2]);
    CBPROF(3); // descriptors + region walks
    const bool third = (((da[0] | da[1] | da[2]) >> 10) & 31) != 0; // (descriptors of empty regions are zero)
    // The sums stay where the reduction leaves them -- lane 63 of a vector register -- and new_choose_table's decision
    // is taken THERE, by vector instructions on values no other lane holds; four lane reads bring back the bit count and
    // the three tables.  As scalar code (until round 4) the decision cost ~20 scalar instructions per
    // region, and a scalar instruction costs this kernel more than a vector one.  An empty
    // region's descriptor and sums are zero: table 0, no bits, as src/loop.c:1771-1777 leaves it.
    {
        // ... and for the three regions AT ONCE: the sums of regions 0, 1, 2 are moved to lanes 61, 62, 63 of one register (the
        // reductions leave each in lane 63 of its own), their descriptors to the same lanes of another, and one run of the
        // decision serves all three; a sum over the three lanes is the bit count, three lane reads are the tables.
        int dav = wave_put_lane<63>(wave_put_lane<62>(wave_put_lane<61>(0, da[0]), da[1]), da[2]);
        int best, choice;
        if (third) { // five reductions in lock-step
            int fs[5] = {s01p[0], s01p[1], s01p[2], s2p[0] | (s2p[1] << 16), s2p[2]};
            wave_reduce_keep_i32<5, 0>(fs);
            const int x01 = wave_tail_place(wave_tail_place(fs[2], fs[1], 1), fs[0], 2);
            const int x2 = wave_tail_place(wave_tail_place(fs[4], (int) ((unsigned) fs[3] >> 16), 1), fs[3] & 0xffff, 2);
            choice = loop_pick_v<true>(dav, x01, x2, &best);
        } else {
            wave_reduce_keep_i32<3, 0>(s01p);
            const int x01 = wave_tail_place(wave_tail_place(s01p[2], s01p[1], 1), s01p[0], 2);
            choice = loop_pick_v<false>(dav, x01, 0, &best);
        }
        bits += wave_tail_sum3(best);
#pragma unroll
        for (int r = 0; r < 3; r++) g.table_select[r] = wave_readlane_i32(choice, 61 + r);
    }
    CBPROF(4); // cost reductions + picks
    return bits;
}

// scfsi_m: bit b = scfsi[ch][b] of this frame when gr == 1, 0 for gr == 0 (whose scalefactors are always sent)
MP3MI_DEVFN int loop_part2_length(const loop_gr &g, int scfsi_m)
{ // src/loop.c:731-780
    const int slen1 = loop_slen1(g.scalefac_compress), slen2 = loop_slen2(g.scalefac_compress);
    if (g.wsf == 1 && g.block_type == 2) return 18 * slen1 + 18 * slen2;
    int bits = 0;
    if ((scfsi_m & 1) == 0) bits += 6 * slen1;
    if ((scfsi_m & 2) == 0) bits += 5 * slen1;
    if ((scfsi_m & 4) == 0) bits += 5 * slen2;
    if ((scfsi_m & 8) == 0) bits += 5 * slen2;
    return bits;
}

// Sequential sum of the noise terms (|xr| - ix^(4/3) step)^2 of lines first + k*stride, k < count, in index
// order (src/loop.c:1030-1060); xr and ix come from LDS, i^(4/3) from the table.  Every lane runs the
// same loop with its own range (a partial-sum job, or a whole band for the exact tier).
MP3MI_DEVFN double loop_noise_sum(const mp3mi_tables *T, const loop_lds &L, double step, int first, int count, int stride)
{
    // (addresses as in loop_noise_jobs: 32-bit byte offsets, all shifts of the one line index)
    const char *lds = (const char *) &L.xr[0];
    const unsigned ix_off = (unsigned) ((const char *) &L.ix[0] - (const char *) &L.xr[0]);
    const char *tab = (const char *) &T->pow43[0];
    double sum = 0.0;
    unsigned line = (unsigned) first;
    int k = 0;
    for (; k + 2 <= count; k += 2) { // two terms in flight, added in order (the rare tier: it must not be what sets the kernel's register count)
        double t[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const unsigned ln = line + (unsigned) (u * stride);
            const unsigned q = *(const uint16_t *) (lds + ix_off + 2u * ln);
            t[u] = __builtin_fabs(*(const double *) (lds + 8u * ln)) - *(const double *) (tab + (size_t) (8u * q)) * step;
        }
        line += (unsigned) (2 * stride);
#pragma unroll
        for (int u = 0; u < 2; u++) sum = sum + t[u] * t[u];
    }
    for (; k < count; k++) {
        const unsigned q = *(const uint16_t *) (lds + ix_off + 2u * line);
        const double t = __builtin_fabs(*(const double *) (lds + 8u * line)) - *(const double *) (tab + (size_t) (8u * q)) * step;
        line += (unsigned) stride;
        sum = sum + t * t;
    }
    return sum;
}

// The partial-sum jobs of the first tier: all lanes take the same number of steps (kmax, the longest job,
// a multiple of 4) and a job that is through reads pair 288 -- xr = 0, ix = 0, terms of exactly +0 -- instead
// of dropping out: no divergent loop, no remainder loop.
// LONG blocks: a job is a run of whole PAIRS of lines (tables_host.cpp: first and count are even), two pairs a step: per pair ONE
// 16-byte read of the spectrum, ONE word of quantised values, two table reads -- four terms in flight on half the address
// arithmetic and half the LDS reads of a line-by-line walk.
typedef double loop_f64x2 __attribute__((ext_vector_type(2)));
MP3MI_DEVFN double loop_noise_jobs_long(const mp3mi_tables *T, const loop_lds &L, double step, int first, int count, int kmax)
{
    // Addresses as 32-bit byte offsets from three bases, all shifts of the one pair index: as `L.ix[line]` and
    // `T->pow43[L.ix[line]]` the compiler derived the second LDS address from the first by a 64-bit multiply-add (a quarter-rate
    // instruction) and sign-extended the table index to 64 bits (L.ix holds magnitudes).
    const char *lds = (const char *) &L.xr[0];
    const unsigned ix_off = (unsigned) ((const char *) &L.ix[0] - (const char *) &L.xr[0]);
    const char *tab = (const char *) &T->pow43[0];
    double sum = 0.0;
    unsigned pr = (unsigned) first >> 1;
    const unsigned npairs = (unsigned) count >> 1;
#if !defined(LOOP_NOISE_PAIRS)
#define LOOP_NOISE_PAIRS 4 /* pairs in flight per step: the jobs of all three rates are at most four pairs long -- one step, no loop */
#endif
    for (int k = 0; 2 * k < kmax; k += LOOP_NOISE_PAIRS) {
        // the quantised values first, then the table entries they select -- the long latency --, the spectrum two pairs at a time behind
        // them: what is in flight at once decides the kernel's register count
        unsigned p[LOOP_NOISE_PAIRS], w[LOOP_NOISE_PAIRS];
        double q43[2 * LOOP_NOISE_PAIRS];
#pragma unroll
        for (int u = 0; u < LOOP_NOISE_PAIRS; u++) {
            p[u] = (unsigned) (k + u) < npairs ? pr + (unsigned) u : 288u;
            w[u] = *(const unsigned *) (lds + ix_off + 4u * p[u]);
        }
#pragma unroll
        for (int u = 0; u < LOOP_NOISE_PAIRS; u++) {
            q43[2 * u] = *(const double *) (tab + (size_t) ((w[u] << 3) & 0x7fff8u));
            q43[2 * u + 1] = *(const double *) (tab + (size_t) ((w[u] >> 13) & 0x7fff8u));
        }
        pr += (unsigned) LOOP_NOISE_PAIRS;
#pragma unroll
        for (int u = 0; u < LOOP_NOISE_PAIRS; u++) {
#if !defined(MP3MI_EMU)
            if ((u & 1) == 0) asm volatile("" ::: "memory"); // (the spectrum's reads stay here, two pairs at a time, behind the table reads)
#endif
#if defined(MP3MI_EMU)
            const double x0 = *(const double *) (lds + 16u * p[u]), x1 = *(const double *) (lds + 16u * p[u] + 8u);
#else
            const loop_f64x2 xx = *(const loop_f64x2 *) (lds + 16u * p[u]);
            const double x0 = xx.x, x1 = xx.y;
#endif
            const double t0 = __builtin_fabs(x0) - q43[2 * u] * step, t1 = __builtin_fabs(x1) - q43[2 * u + 1] * step;
            // (this is the FIRST tier: its sum only has to lie within 1e-12 of the reference's -- loop_noise_close -- and a fused
            // square-and-add differs from the reference's two roundings by an ulp of the running sum per term, < 2.2e-14 over the
            // 192 terms of the widest band; the difference itself stays two roundings: fused, a value of 8000 could move a term by 1.6e-12)
            sum = __builtin_fma(t0, t0, sum);
            sum = __builtin_fma(t1, t1, sum);
        }
    }
    return sum;
}
// SHORT blocks (one granule in twenty): a job's lines are three apart; line by line, two terms a step (what is in flight here, not the
// kernel's common path, would otherwise set its register count).
MP3MI_DEVFN double loop_noise_jobs_short(const mp3mi_tables *T, const loop_lds &L, double step, int first, int count, int kmax)
{
    const char *lds = (const char *) &L.xr[0];
    const unsigned ix_off = (unsigned) ((const char *) &L.ix[0] - (const char *) &L.xr[0]);
    const char *tab = (const char *) &T->pow43[0];
    double sum = 0.0;
    unsigned line = (unsigned) first;
    for (int k = 0; k < kmax; k += 2) {
        double t[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const unsigned ln = (unsigned) (k + u) < (unsigned) count ? line + (unsigned) (3 * u) : 576u;
            const double x = *(const double *) (lds + 8u * ln);
            const unsigned q = *(const uint16_t *) (lds + ix_off + 2u * ln);
            t[u] = __builtin_fabs(x) - *(const double *) (tab + (size_t) (8u * q)) * step;
        }
        line += 6u;
#pragma unroll
        for (int u = 0; u < 2; u++) sum = __builtin_fma(t[u], t[u], sum); // (first tier: see loop_noise_jobs_long)
    }
    return sum;
}

// the reference's sequential band sums (src/loop.c:1030-1060)
MP3MI_DEVFN double loop_noise_exact(const mp3mi_tables *T, const loop_lds &L, double step, bool bandlane, int sfirst, int scount, int sstride)
{
    const double sum = loop_noise_sum(T, L, step, sfirst, bandlane ? scount : 0, sstride);
    return bandlane ? sum / (double) scount : 0.0;
}

// is the partial-sum noise of this band lane too close to its threshold to decide `noise > xmin`?
MP3MI_DEVFN bool loop_noise_close(bool bandlane, double xfsf, double xmin)
{
    return bandlane && xmin > 0.0 && __builtin_fabs(xfsf - xmin) <= 1e-12 * xmin;
}

// The kernel's arguments as ONE record.  The search needs a dozen pointers and launch constants a few times per granule,
// per frame or per stream -- and every scalar register matters in between: held in registers across the whole kernel they were
// what the compiler spilled (131 scalar registers into the lanes of three vector registers, written and read back by vector
// instructions in the granule's set-up and in every iteration of the distortion loop).  Now the record stays where the launch
// put it -- the kernel argument segment -- and a value is fetched by a scalar load WHERE IT IS USED (LOOP_ARG: the pointer
// behind an optimisation barrier, so that the loads are not hoisted to the kernel's top again).
struct loop_kargs { // (the table block is a kernel argument of its own: a read-only, no-alias pointer, whose loads are scalar loads)
    mp3mi_geom geo;
    const double *xr_all;
    const mp3mi_psy_out *psy;
    const mp3mi_loop_prep *prep;
    const int32_t *bits_per_frame;
    mp3mi_loop_state *state;
    int16_t *ix_out;
    mp3mi_frame_side *side_out;
    unsigned *gate_count;
    mp3mi_loop_place place;
};
#if defined(MP3MI_EMU)
typedef const loop_kargs *loop_kargs_p;
#define LOOP_ARG(field) (ka->field)
#define loop_kargs_ptr(ka) (ka)
#else
typedef const __attribute__((address_space(4))) loop_kargs *loop_kargs_p;
MP3MI_DEVFN loop_kargs_p loop_kargs_here(loop_kargs_p p)
{
    int z = 0;
    asm volatile("" : "+s"(z));
    return (loop_kargs_p) ((const __attribute__((address_space(4))) char *) p + z);
}
#define LOOP_ARG(field) (loop_kargs_here(ka)->field)
#define loop_kargs_ptr(ka) loop_kargs_here(ka)
#endif

// frames of the chunk [f0, f0 + nf) that a stream with n valid samples per channel still has
MP3MI_DEVFN int loop_frames_here(int f0, int nf, int n)
{
    const int total = (n + 1151) / 1152, left = total - f0;
    return left < 0 ? 0 : (left < nf ? left : nf);
}

// ---- placement: which stream does this wavefront take? ----
// The kernel ends when its most loaded SIMD ends.  The hardware decides where a workgroup runs,
// so instead of stream = blockIdx a wavefront looks at where it landed (HW_ID / XCC_ID) and takes a
// stream from the list sorted by the previous chunk's cost such that every SIMD gets a heavy, a
// light and two middle streams: the r-th wavefront to arrive on the SIMD with arrival ticket i
// takes sorted position r*NS + i (r even) or (r+1)*NS - 1 - i (r odd), NS = place.n_simd.
// Any wavefront that finds its position missing or taken (more streams than slots, an uneven
// spread) takes the next free one from a scan counter, so every stream is taken exactly once.
MP3MI_DEVFN int loop_place_stream(loop_kargs_p ka, int n_streams, int block)
{
#if defined(MP3MI_EMU)
    return ka->place.order ? ka->place.order[block] : block;
#else
    mp3mi_loop_place pl; // (field by field: the record lives in the kernel argument segment)
    pl.order = LOOP_ARG(place.order);
    if (!pl.order) return block;
    pl.taken = LOOP_ARG(place.taken); pl.simd_slots = LOOP_ARG(place.simd_slots); pl.simd_idx = LOOP_ARG(place.simd_idx);
    pl.ticket = LOOP_ARG(place.ticket); pl.scan = LOOP_ARG(place.scan); pl.n_simd = LOOP_ARG(place.n_simd);
    int pos = -1;
    if (wave_lane() == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));  // XCC_ID
        // simd_id[5:4] cu_id[11:8] sh_id[12] se_id[15:13]
        const unsigned key = ((xcc & 7u) << 10) | (((hw >> 13) & 7u) << 7) | (((hw >> 12) & 1u) << 6) | (((hw >> 8) & 15u) << 2) | ((hw >> 4) & 3u);
        const unsigned r = atomicAdd(&pl.simd_slots[key], 1u);
        unsigned idx = 0xffffffffu;
        if (r == 0) {
            idx = atomicAdd(pl.ticket, 1u);
            __hip_atomic_store(&pl.simd_idx[key], idx + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else if (r < 4) {
            for (int spin = 0; spin < 20000; spin++) { // the first arrival on this SIMD publishes within microseconds
                const unsigned v = __hip_atomic_load(&pl.simd_idx[key], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                if (v) { idx = v - 1u; break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        if (r < 4 && idx < (unsigned) pl.n_simd) {
            const int p = (r & 1u) ? (int) ((r + 1u) * pl.n_simd - 1u - idx) : (int) (r * pl.n_simd + idx);
            if (p < n_streams && atomicExch(&pl.taken[p], 1u) == 0u) pos = p;
        }
        while (pos < 0) { // n_streams wavefronts claim n_streams positions: this always finds one
            const unsigned p = atomicAdd(pl.scan, 1u);
            if (p >= (unsigned) n_streams) { pos = block; break; } // cannot happen; never spin forever
            if (atomicExch(&pl.taken[p], 1u) == 0u) pos = (int) p;
        }
        pos = pl.order[pos];
    }
    return __builtin_amdgcn_readfirstlane(pos);
#endif
}

// order[rank] = stream, streams ranked by cost (descending, ties by index): one thread per stream.
__global__ void __launch_bounds__(256) k_rank(const int *__restrict__ cost, int *__restrict__ order, int n)
{
    const int i = (int) (blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const int ci = cost[i];
    int rank = 0;
    for (int j = 0; j < n; j++) {
        const int cj = cost[j];
        rank += (cj > ci || (cj == ci && j < i)) ? 1 : 0;
    }
    order[rank] = i;
}

void mp3mi_launch_rank(const int *cost, int *order, int n, hipStream_t st)
{
    hipLaunchKernelGGL(k_rank, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, cost, order, n);
}

// 80 VGPRs: four resident wavefronts per SIMD leave 192 of its 512 registers to the kernels of the next chunk.
// One stream, start to end, by one wavefront (k_loop below).
MP3MI_DEVFN void loop_stream(const mp3mi_tables *__restrict__ T, loop_kargs_p ka, loop_lds &L, const uint16_t *GL, int block)
{
    const int lane = wave_lane();
    const int C = LOOP_ARG(geo.channels), G = 2 * LOOP_ARG(geo.nf);
    const int s = loop_place_stream(ka, LOOP_ARG(geo.n_streams), block);
    int work = 0; // cost of this stream in this launch: 4 per quantise+count pass, 5 per distortion-loop iteration
    const int bitsPerFrame = LOOP_ARG(bits_per_frame)[s];
    const int mean_bits = (bitsPerFrame - (32 + (C == 1 ? 136 : 256) + (LOOP_ARG(geo.crc) ? 16 : 0))) / 2; // src/musicin.c:729-746
    PROF_DECL;
    // Residency census (batch.cpp, k_gate): every wavefront counts itself in when it starts.  The
    // counter only ever grows; nothing in this kernel waits on it.
    {
        unsigned *gate_count = LOOP_ARG(gate_count);
        if (gate_count && lane == 0) atomicAdd(gate_count, 1u);
    }
#if !defined(MP3MI_EMU)
    // this wavefront is on the critical path of the whole batch: let it issue ahead of the
    // feed-forward kernels of the next chunk that run beside it (batch.cpp); adjusted
    // per frame by the pacing below
    __builtin_amdgcn_s_setprio(2);
#endif

    loop_regs R;
    loop_desc_init(T, lane, &R.desc_a, &R.desc_b);

    for (int i = lane; i < (int) (sizeof(mp3mi_loop_state) / 4); i += 64) ((int *) &L.st)[i] = ((const int *) &LOOP_ARG(state)[s])[i];
    L.ix[576 + lane] = 0; L.ix[640 + lane] = 0; // (the padding: pairs that do not exist)
    if (lane < 2) L.xr[576 + lane] = 0.0;
    wave_sync();
    int ref_abort = __builtin_amdgcn_readfirstlane(L.st.ref_abort); // (sticky: the first event of the stream stands)

    // ragged batch: frames of this stream beyond its last (zero-filled) one are not encoded
    const int nf_s = LOOP_ARG(geo.n_samples) ? loop_frames_here(LOOP_ARG(geo.f0), LOOP_ARG(geo.nf), LOOP_ARG(geo.n_samples)[s]) : LOOP_ARG(geo.nf);
    for (int fl = 0; fl < nf_s; fl++) {
        // ResvFrameBegin (src/reservoir.c:45-93); main_data_begin*8 == ResvSize by construction
        int ResvSize = L.st.ResvSize;
        int ResvMax = (bitsPerFrame > 7680) ? 0 : 7680 - bitsPerFrame;
        if (ResvMax > 4088) ResvMax = 4088;
        const int main_data_begin = ResvSize / 8;
        int resvDrain = 0;
        wave_sync();

        for (int gr = 0; gr < 2; gr++)
            for (int ch = 0; ch < C; ch++) {
                const int lane = wave_lane_here(); // nothing derived from the lane index outlives this granule
                const int gl = 2 * fl + gr;
                const size_t rec = ((size_t) s * G + gl) * C + ch;
                // (the granule's pointers in one go: one laundered base, the scalar loads behind it issue together)
                loop_kargs_p kg = loop_kargs_ptr(ka);
                const mp3mi_psy_out *po = &kg->psy[rec];
                loop_gr g;
                g.block_type = po->block_type;
                g.wsf = g.block_type != 0;
                const bool shortb = g.wsf && g.block_type == 2;
                g.sfb_lmax = shortb ? 0 : 21; // gr_deco, src/loop.c:2063
                g.sfb_smax = shortb ? 0 : 12;
                g.address1 = L.st.addr[gr][ch][0];
                g.address2 = L.st.addr[gr][ch][1];
                g.address3 = L.st.addr[gr][ch][2];
                const int nband = shortb ? 36 : 21;   // band lanes
                float y34[LOOP_NV], y34max;
                {
                    double xr[LOOP_NV];
                    const double *xr_all = kg->xr_all;
#pragma unroll
                    for (int k = 0; k < LOOP_SLOTS; k++) { // a pair = 16 bytes per lane
                        const int pr = loop_pair_of(lane, k);
                        const bool there = k < 4 || pr < 288;
                        xr[2 * k] = there ? xr_all[rec * 576 + 2 * pr] : 0.0;
                        xr[2 * k + 1] = there ? xr_all[rec * 576 + 2 * pr + 1] : 0.0;
                    }
                    y34max = loop_power34(xr, y34);
#pragma unroll
                    for (int k = 0; k < LOOP_SLOTS; k++) {
                        const int pr = loop_pair_of(lane, k);
                        if (k < 4 || pr < 288) { L.xr[2 * pr] = xr[2 * k]; L.xr[2 * pr + 1] = xr[2 * k + 1]; }
                    }
                }

                // ---- calc_xmin (src/loop.c:1085-1118) and the values calc_scfsi stores (src/loop.c:631-667)
                //      were computed by k_mdct's tail (k_prep); only the stateful decision of calc_scfsi happens here ----
                PROF(0);
                const mp3mi_loop_prep *pp = &kg->prep[rec];
                // per-band state (allowed distortion, scalefactors): L.band_*
                if (lane < 36) {
                    L.band_xmin[lane] = lane < nband ? pp->xmin[lane] : 0.0;
                    L.band_sf[lane] = 0;
                    L.band_sfsave[lane] = 0;
                }
                if (lane == 0) {
                    L.st.sc_xrmax[gr][ch] = pp->sc_xrmax;
                    L.st.sc_en_tot[gr][ch] = pp->sc_en_tot;
                }
                if (!shortb && lane < 21) {
                    L.st.sc_en[gr][ch][lane] = pp->sc_en[lane];
                    L.st.sc_xm[gr][ch][lane] = pp->sc_xm[lane];
                }
                const int nonzero = pp->nonzero;
                int scfsi_m = 0; // this granule's scfsi bits (wave-uniform): what the search asks for between passes
                wave_sync();
                if (gr == 1) {
                    int condition = 0;
                    for (int gr2 = 0; gr2 < 2; gr2++) {
                        if (L.st.sc_xrmax[ch][gr2] != 0) condition++; // [ch][gr2], sic
                        if (!shortb) condition++;
                    }
                    condition++; // abs(en_tot[0]-en_tot[1]) is a pointer difference in the reference
                    int d = 0, dx = 0;
                    if (lane < 21) {
                        d = abs(L.st.sc_en[ch][0][lane] - L.st.sc_en[ch][1][lane]);
                        dx = abs(L.st.sc_xm[ch][0][lane] - L.st.sc_xm[ch][1][lane]);
                    }
                    if (wave_sum_i32(d) < 100) condition++;
                    if (condition == 6) {
                        for (int band = 0; band < 4; band++) {
                            const int lo = (band == 0) ? 0 : (band == 1 ? 6 : (band == 2 ? 11 : 16));
                            const int hi = (band == 0) ? 6 : (band == 1 ? 11 : (band == 2 ? 16 : 21));
                            const bool in = lane >= lo && lane < hi;
                            const int s0 = wave_sum_i32(in ? d : 0), s1 = wave_sum_i32(in ? dx : 0);
                            const int v = (s0 < 10 && s1 < 10) ? 1 : 0;
                            scfsi_m |= v << band;
                        }
                    }
                    if (lane < 4) LOOP_ARG(side_out)[(size_t) s * LOOP_ARG(geo.nf) + fl].scfsi[ch][lane] = (scfsi_m >> lane) & 1; // (always decided in granule 1)
                }
                wave_sync();

                // ---- ResvMaxBits (src/reservoir.c:101-134) ----
                int max_bits;
                {
                    const int mb = mean_bits / C;
                    max_bits = mb > 4095 ? 4095 : mb;
                    if (ResvMax != 0) {
                        const int more_bits = (int) (po->pe * 3.1 - (double) mb);
#if defined(MP3MI_ULP_CENSUS) && !defined(MP3MI_EMU)
                        if (lane == 0) { // site 3b: (int)(pe * 3.1 - mean_bits); pe within 1.8e-12 of the reference's (k_psy.hip)
                            const double t = po->pe * 3.1 - (double) mb, fr = __builtin_fabs(t - __builtin_rint(t));
                            ULP_CENSUS(UC_PE_RESV, fr <= 6e-12, fr <= 6e-12 * 1048576.0);
                        }
#endif
                        int add_bits = 0;
                        if (more_bits > 100) {
                            const int frac = (ResvSize * 6) / 10;
                            add_bits = frac < more_bits ? frac : more_bits;
                        }
                        const int over_bits = ResvSize - ((ResvMax * 8) / 10) - add_bits;
                        if (over_bits > 0) add_bits += over_bits;
                        max_bits += add_bits;
                        if (max_bits > 4095) max_bits = 4095;
                    }
                }

                // ---- reset of iteration variables (src/loop.c:318-344) ----
                g.part2_3_length = 0; g.big_values = 0; g.count1 = 0; g.scalefac_compress = 0;
                g.table_select[0] = g.table_select[1] = g.table_select[2] = 0;
                g.region0_count = 0; g.region1_count = 0; g.part2_length = 0; g.preflag = 0;
                g.count1table_select = 0; g.q = 0;
#pragma unroll
                for (int k = 0; k < LOOP_SLOTS; k++) ((unsigned *) L.ix)[loop_pair_of(lane, k)] = 0u;
                wave_sync();

                if (nonzero) {
                    g.q = pp->q0; // quantanf_init (src/loop.c:369-402), from k_prep
                    const int test_flags = LOOP_ARG(geo.test_flags); // (tests: an exact tier only)

                    // ---- outer_loop (src/loop.c:415-558) ----
                    int iteration = 0, bits = 0, over, status, save_preflag, save_compress;
                    do {
                        const int lane = wave_lane_here(); // ... nor an iteration of the distortion loop
                        iteration++;
                        work += 5;
                        g.part2_length = loop_part2_length(g, scfsi_m);
                        int huff_bits = max_bits - g.part2_length;
                        if (huff_bits < 0) {
                            // assert( max_bits >= 0 ) of inner_loop (src/loop.c:579): the scalefactors alone exceed the
                            // granule's budget and the reference dies.  Without the assert its loop -- and the one below --
                            // would raise the step for ever: no count is <= a negative budget.  The stream is void from
                            // here on; any budget lets the search run out.
                            LOOP_REF_ABORT(MP3MI_DEV_ABORT_HUFF_BITS, LOOP_ARG(geo.fabs0) + LOOP_ARG(geo.f0) + fl);
                            huff_bits = max_bits;
                        }
                        // bin_search_StepSize (src/loop.c:2119-2140, first iteration only) and inner_loop (src/loop.c:569-606)
                        // as ONE loop around ONE copy of the quantise+count pass (the pass is ~3 k instructions; a second
                        // inlined copy is instruction-cache pressure for nothing).  The bisection probes (top + bot) / 2
                        // until the count equals max_bits or the probes are one step apart; inner_loop then raises the step
                        // until the bits fit -- and its first pass repeats the bisection's last probe (same step, same xr),
                        // which is taken over instead of recomputed.
                        {
                            bool bisect = iteration == 1;
                            int top = g.q, bot = 200, next = g.q, last = g.q;
                            for (;;) {
                                if (bisect) {
                                    last = next;
                                    next = (top + bot) / 2;
                                    g.q = next;
                                }
                                PROF(1);
                                const bool quant_exact = (test_flags & 8) != 0; // MP3MI_QUANT_EXACT=1: the quantiser's exact tier only (tests)
                                const bool az = !quant_exact && loop_all_zero(y34max, g.q);
                                work += 4;
                                const loop_qinfo qi = loop_quantize(T, L, y34, y34max, g.q, az, quant_exact, shortb);
                                PROF(2);
                                bits = loop_count_bits(T, R, L, GL, g, qi, az CBPROF_PASS);
                                PROF(3);
                                wave_sync();
                                if (bisect) {
                                    if (bits > max_bits) top = next; else bot = next;
                                    if (bits != max_bits && abs(last - next) > 1) continue;
                                    bisect = false; // this probe is inner_loop's first pass
                                }
                                if (!(bits > huff_bits)) break;
                                g.q += 1;
                            }
                        }

                        PROF(1);
                        // ---- the distortion side of the iteration.  What it knows about "its" bands and noise jobs a
                        //      lane fetches HERE (two 8-byte loads of packed constants, mp3mi_tables::lane_bands / lane_jobs),
                        //      and the per-band state comes from / goes back to L.band_*: nothing of it is alive during
                        //      the quantise+count passes above ----
                        const bool bandlane = lane < nband;
                        // (registers: the partial sums below are where the kernel's register count is set -- ten y34, four terms in
                        // flight --, so nothing that is only needed behind them is fetched or unpacked in front of them)
                        unsigned long long jobs = T->lane_jobs[shortb][lane];
                        const int jfirst = (int) (jobs & 1023ull), jcount = (int) ((jobs >> 10) & 255ull);
                        const int sstride = shortb ? 3 : 1;
                        const int jmax4 = (T->nj_max[shortb] + 3) & ~3; // the longest job, in steps of four terms
                        double xfsf_r = 0.0;
                        // calc_noise (src/loop.c:1007-1067).  The noise of a band is only ever COMPARED with the
                        // allowed distortion, and it is a sum of non-negative terms, so any summation order
                        // agrees with the reference's sequential one to within 2(n-1) ulp (n <= 102 lines:
                        // < 2.3e-14 relative; the fused square-and-add of the partial sums adds an ulp per term, the
                        // multiplication by 1 / n in place of the division 1.5).  First tier: every band is cut into parts of ~10 lines summed by
                        // different lanes.  Only if a band lands within 1e-12 of its threshold is the
                        // reference's order used (loop_noise_exact) -- at both places that compare.
                        bool xfsf_exact = (test_flags & 1) != 0;
                        const double noise_step = T->step[g.q - MP3MI_STEP_MIN];
                        double vjob = 0.0;
                        if (!xfsf_exact) vjob = shortb ? loop_noise_jobs_short(T, L, noise_step, jfirst, jcount, jmax4) : loop_noise_jobs_long(T, L, noise_step, jfirst, jcount, jmax4);
                        // band of each of this lane's five PAIRS (a band's edges are even: both lines of a pair are of one band), 6 bits
                        // each; short blocks: the lines of a pair belong to two windows, i.e. two band lanes: 10 fields of a 64-bit word
                        // (asked for here: its latency passes under the band sums' reduction)
                        const unsigned long long bandpack = T->lane_bands[shortb][wave_lane_here()];
                        const double inv_lines = T->lane_inv_lines[shortb][wave_lane_here()]; // 1 / (lines of this band lane's band)
                        // the band lane's state (fetched behind the partial sums, see above)
                        const int bl = wave_lane_here();
                        double xmin_r = bandlane ? L.band_xmin[bl] : 0.0;
                        int sf_r = bandlane ? L.band_sf[bl] : 0;
                        {
                            if (!xfsf_exact) {
                                // The jobs of a band sit in consecutive lanes: a segmented sum by doubling (lane l takes
                                // over the sum of lane l + d where that is a job of the same band: bit of jseg) leaves
                                // the band's sum in its first job's lane, where the band lane fetches it.  Any order
                                // of these non-negative terms is as good as another here (see above).
                                double v = vjob;
                                jobs = wave_opaque_u64(jobs);
                                const int jseg = (int) ((jobs >> 18) & 31ull), pj0 = (int) ((jobs >> 23) & 63ull), scount = (int) ((jobs >> 35) & 255ull);
                                // (a band has at most 16 jobs -- tables_host.cpp refuses a table with more -- so four doublings do)
                                const int lane4 = 4 * lane;
                                { const double o = wave_down_f64<1>(v, lane4); v = v + ((jseg & 1) ? o : 0.0); }
                                { const double o = wave_down_f64<2>(v, lane4); v = v + ((jseg & 2) ? o : 0.0); }
                                { const double o = wave_down_f64<4>(v, lane4); v = v + ((jseg & 4) ? o : 0.0); }
                                { const double o = wave_down_f64<8>(v, lane4); v = v + ((jseg & 8) ? o : 0.0); }
                                const double sum = __shfl(v, pj0);
                                (void) scount;
                                xfsf_r = bandlane ? sum * inv_lines : 0.0; // (first tier: within 1.5 ulp of the reference's quotient; the exact tier divides)
                                if (wave_any(loop_noise_close(bandlane, xfsf_r, xmin_r))) xfsf_exact = true;
                            }
                            if (xfsf_exact) xfsf_r = loop_noise_exact(T, L, noise_step, bandlane, (int) ((jobs >> 43) & 1023ull), (int) ((jobs >> 35) & 255ull), sstride);
                        }
                        if (bandlane) L.band_sfsave[lane] = sf_r; // the result of this iteration stands (src/loop.c:505-519)
                        save_preflag = g.preflag;
                        save_compress = g.scalefac_compress;
                        // bands whose noise exceeds the allowed distortion (bit b = band lane b)
                        const unsigned long long viol = __ballot(bandlane && xfsf_r > xmin_r);

                        PROF(4);
                        // preemphasis (src/loop.c:1161-1214)
                        {
                            bool skip = false;
                            if (scfsi_m) {
                                g.preflag = L.preflag0[ch];
                                skip = true;
                            }
                            if (!skip && g.block_type != 2 && g.preflag == 0) {
                                if ((viol & 0x1E0000ull) == 0x1E0000ull) { // sfb 17..20 all violate
                                    g.preflag = 1;
                                    if (lane < g.sfb_lmax) xmin_r = xmin_r * T->pretab_xmin[LOOP_PRETAB[lane]];
                                    // the thresholds moved: amp_scalefac_bands compares against the new ones
                                    if (!xfsf_exact && wave_any(loop_noise_close(bandlane, xfsf_r, xmin_r))) {
                                        xfsf_exact = true;
                                        xfsf_r = loop_noise_exact(T, L, noise_step, bandlane, (int) ((jobs >> 43) & 1023ull), (int) ((jobs >> 35) & 255ull), sstride);
                                    }
#pragma unroll
                                    for (int k = 0; k < LOOP_SLOTS; k++) { // (long blocks only: a pair is of one band)
                                        const int b = (int) (((unsigned) bandpack >> (6 * k)) & 63u); // (63: a pair that does not exist)
                                        const int line = 2 * loop_pair_of(lane, k);
                                        if (b < g.sfb_lmax) {
                                            const double f = T->pretab_xr[LOOP_PRETAB[b]];
                                            L.xr[line] = L.xr[line] * f;
                                            L.xr[line + 1] = L.xr[line + 1] * f;
                                            y34[2 * k] = loop_rescale34(y34[2 * k], LOOP_PRETAB[b]);
                                            y34[2 * k + 1] = loop_rescale34(y34[2 * k + 1], LOOP_PRETAB[b]);
                                        }
                                    }
                                    y34max = y34max * LOOP_Y34MAX_GROW;
                                }
                            }
                        }
                        wave_sync();

                        // amp_scalefac_bands (src/loop.c:1225-1350)
                        {
                            int copySF = 0, preventSF = 0;
                            if (scfsi_m) {
                                if (iteration == 1) copySF = 1; else preventSF = 1;
                            }
                            const double ifqstep = T->sqrt2, ifqstep2 = ifqstep * ifqstep;
                            bool amp = false;
                            if (bandlane) {
                                bool skipband = false;
                                if (!shortb && (copySF || preventSF)) {
                                    const int sb4 = (lane < 6) ? 0 : (lane < 11 ? 1 : (lane < 16 ? 2 : 3));
                                    if ((scfsi_m >> sb4) & 1) {
                                        if (copySF) sf_r = L.sf_gr0[ch][lane];
                                        skipband = true;
                                    }
                                }
                                if (!skipband && xfsf_r > xmin_r) {
                                    amp = true;
                                    xmin_r = xmin_r * ifqstep2;
                                    sf_r = sf_r + 1;
                                }
                                L.band_xmin[lane] = xmin_r; // (pre-emphasised and / or doubled, or as it was)
                                L.band_sf[lane] = sf_r;
                            }
                            const unsigned long long ampmask = __ballot(amp); // bit b = band lane b amplified
                            over = __popcll(ampmask);
                            if (over) {
                                // The amplified bands are a few runs of consecutive lines, so most of the five slots of 64
                                // pairs hold none of them: a slot's lines are skipped as a whole,
                                // and inside a slot only the amplified lines touch the LDS.  ampmask only has bits of
                                // band lanes, so lines above the last band (b >= nband) find a zero bit.
                                if (!shortb) { // a pair is of one band; 21 band lanes: the mask is one word
                                    const unsigned amp32 = (unsigned) ampmask;
#pragma unroll
                                    for (int k = 0; k < LOOP_SLOTS; k++) {
                                        const unsigned b = ((unsigned) bandpack >> (6 * k)) & 63u;
                                        const int line = 2 * loop_pair_of(lane, k);
                                        if (b < 32u && ((amp32 >> b) & 1u)) { // (a slot none of whose pairs is amplified: the branch over an empty execution mask)
                                            L.xr[line] = L.xr[line] * ifqstep;
                                            L.xr[line + 1] = L.xr[line + 1] * ifqstep;
                                            y34[2 * k] = y34[2 * k] * 1.2968395546510096f; // loop_rescale34(y34, 1)
                                            y34[2 * k + 1] = y34[2 * k + 1] * 1.2968395546510096f;
                                        }
                                    }
                                } else {
#pragma unroll
                                    for (int j = 0; j < LOOP_NV; j++) {
                                        const unsigned b = (unsigned) ((bandpack >> (6 * j)) & 63ull);
                                        const bool f = ((ampmask >> b) & 1ull) != 0;
                                        const int line = 2 * loop_pair_of(lane, j >> 1) + (j & 1);
                                        if (f) {
                                            L.xr[line] = L.xr[line] * ifqstep;
                                            y34[j] = y34[j] * 1.2968395546510096f;
                                        }
                                    }
                                }
                                y34max = y34max * LOOP_Y34MAX_AMP;
                            }
                        }
                        wave_sync(); // amplified lines in L.xr are read by other lanes' noise sums

                        PROF(5);
                        // loop_break (src/loop.c:1131-1152) then scale_bitcount (src/loop.c:792-860)
                        {
                            status = !wave_any(bandlane && sf_r == 0);
                            if (status == 0) {
                                int m1, m2;
                                if (shortb) {
                                    m1 = (lane < 18) ? sf_r : 0;
                                    m2 = (lane >= 18 && lane < 36) ? sf_r : 0;
                                } else {
                                    m1 = (lane < 11) ? sf_r : 0;
                                    m2 = (lane >= 11 && lane < 21) ? sf_r : 0;
                                }
                                int mmv[2] = {m1, m2};
                                wave_reduce_i32<0, 2>(mmv); // in lock-step: each fills the other's DPP wait states
                                const int mm1 = mmv[0], mm2 = mmv[1];
                                // the first k with mm1 < 2^slen1[k] and mm2 < 2^slen2[k] only depends on the bit lengths
                                // of the two maxima: tabulated, a nibble per (length of mm1 <= 4, length of mm2 <= 3)
                                const int bl1 = 32 - __clz(mm1), bl2 = 32 - __clz(mm2);
                                int ep = 2;
                                if (bl1 <= 4 && bl2 <= 3) {
                                    const unsigned long long w = bl1 < 4 ? (0xdcb4a98476543210ull >> (16 * bl1)) : 0xfeeeull;
                                    g.scalefac_compress = (int) ((w >> (4 * bl2)) & 15ull);
                                    ep = 0;
                                }
                                status = ep;
                            }
                        }
                    } while (status == 0 && over > 0);
                    g.preflag = save_preflag;
                    g.scalefac_compress = save_compress;
                    g.part2_length = loop_part2_length(g, scfsi_m);
                    g.part2_3_length = g.part2_length + bits;
                }

                // ResvAdjust (src/reservoir.c:141-145), global_gain (src/loop.c:357)
                ResvSize += (mean_bits / C) - g.part2_3_length;
                const int global_gain = loop_nint((double) g.q + 210.0);
                if (global_gain >= 256) LOOP_REF_ABORT(MP3MI_DEV_ABORT_GLOBAL_GAIN, LOOP_ARG(geo.fabs0) + LOOP_ARG(geo.f0) + fl); // assert, src/loop.c:358

                PROF(6);
                // ---- hand the granule over: signed ix (src/l3bitstream.c:115-125) and side info ----
                loop_kargs_p ko = loop_kargs_ptr(ka);
                unsigned *ix_out = (unsigned *) ko->ix_out;
#pragma unroll
                for (int k = 0; k < LOOP_SLOTS; k++) {
                    const int pr = loop_pair_of(lane, k);
                    if (k < 4 || pr < 288) {
                        const unsigned w = ((const unsigned *) L.ix)[pr]; // what the last pass (or the reset) left
                        int x = (int) (w & 0xffffu), y = (int) (w >> 16);
                        if (L.xr[2 * pr] < 0 && x > 0) x = -x;
                        if (L.xr[2 * pr + 1] < 0 && y > 0) y = -y;
                        ix_out[rec * 288 + pr] = ((unsigned) x & 0xffffu) | ((unsigned) y << 16);
                    }
                }
                {
                    mp3mi_gr_side *o = &ko->side_out[(size_t) s * ko->geo.nf + fl].gr[gr][ch]; // straight to memory
                    if (lane == 0) {
                        o->part2_3_length = g.part2_3_length; o->big_values = g.big_values; o->count1 = g.count1;
                        o->global_gain = global_gain; o->scalefac_compress = g.scalefac_compress;
                        o->window_switching_flag = g.wsf; o->block_type = g.block_type;
                        o->table_select[0] = g.table_select[0]; o->table_select[1] = g.table_select[1];
                        o->table_select[2] = g.table_select[2];
                        o->region0_count = g.region0_count; o->region1_count = g.region1_count;
                        o->preflag = g.preflag; o->count1table_select = g.count1table_select;
                        o->part2_length = g.part2_length;
                        L.p23[gr][ch] = g.part2_3_length;
                        if (gr == 0) L.preflag0[ch] = g.preflag;
                        L.st.addr[gr][ch][0] = g.address1; L.st.addr[gr][ch][1] = g.address2; L.st.addr[gr][ch][2] = g.address3;
                    }
                    if (lane < 39) { // the scalefactors of the last iteration whose result stands (all 0 without a search)
                        const int sfv = lane < nband ? L.band_sfsave[lane] : 0;
                        o->scalefac[lane] = sfv;
                        if (gr == 0 && lane < 21) L.sf_gr0[ch][lane] = shortb ? 0 : sfv;
                    }
                }
                wave_sync();
            }

        // ---- ResvFrameEnd (src/reservoir.c:155-226) ----
        if (C == 2 && (mean_bits & 1)) ResvSize += 1;
        {
            int over_bits = ResvSize - ResvMax;
            if (over_bits < 0) over_bits = 0;
            ResvSize -= over_bits;
            int stuffingBits = over_bits;
            if ((over_bits = ResvSize % 8)) { stuffingBits += over_bits; ResvSize -= over_bits; }
            if (lane == 0) {
                bool moved = false; // some part2_3_length took stuffing bits: the records in memory follow
                if (stuffingBits) {
                    moved = true;
                    if (L.p23[0][0] + stuffingBits < 4095)
                        L.p23[0][0] += stuffingBits;
                    else {
                        for (int gr = 0; gr < 2; gr++)
                            for (int ch = 0; ch < C; ch++) {
                                if (stuffingBits == 0) break;
                                const int extra = 4095 - L.p23[gr][ch];
                                const int now = extra < stuffingBits ? extra : stuffingBits;
                                L.p23[gr][ch] += now;
                                stuffingBits -= now;
                            }
                        resvDrain = stuffingBits;
                    }
                }
                mp3mi_frame_side *o = &LOOP_ARG(side_out)[(size_t) s * LOOP_ARG(geo.nf) + fl];
                if (moved)
                    for (int gr = 0; gr < 2; gr++)
                        for (int ch = 0; ch < C; ch++) o->gr[gr][ch].part2_3_length = L.p23[gr][ch];
                o->main_data_begin = main_data_begin;
                o->resvDrain = resvDrain;
                L.st.ResvSize = ResvSize;
            }
        }
        wave_sync();
#if !defined(MP3MI_EMU)
        // Pacing: the kernel ends when its slowest stream ends, and streams differ by up to 1.5x in
        // work.  gate_count[1] counts the frames finished by all streams of this launch; a stream
        // behind the average raises its wave priority (it then issues ahead of the three other
        // wavefronts of its SIMD), a stream ahead of it lowers it.  Purely a scheduling hint.
        unsigned *gate_count = LOOP_ARG(gate_count);
        if (gate_count) {
            const unsigned done_all = __builtin_amdgcn_readfirstlane((int) (lane == 0 ? atomicAdd(gate_count + 1, 1u) + 1u : 0u));
            // (frames per wavefront so far against the average over the wavefronts of the launch: a wavefront on its
            // second stream has that stream's frames on top of the first one's)
            const int n_act = LOOP_ARG(geo.n_streams); // (fetched and converted here, once per frame)
            const float lead = (float) (fl + 1) - (float) done_all / (float) n_act;
            if (lead < -1.0f) __builtin_amdgcn_s_setprio(3);
            else if (lead < 0.0f) __builtin_amdgcn_s_setprio(2);
            else if (lead < 1.0f) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
#endif
    }
    {
        const int ln = wave_lane_here(); // (not the address the state was loaded through, kept alive across the whole kernel)
        if (ln == 0) L.st.ref_abort = ref_abort;
        wave_sync();
        for (int i = ln; i < (int) (sizeof(mp3mi_loop_state) / 4); i += 64) ((int *) &LOOP_ARG(state)[s])[i] = ((const int *) &L.st)[i];
    }
    {
        int *cost = LOOP_ARG(place.cost);
        if (cost && lane == 0) cost[s] = work;
    }
#if defined(MP3MI_LOOP_PROFILE) && !defined(MP3MI_EMU)
    if (lane == 0 && s < 65536) g_loop_work[s] = (unsigned long long) work | ((unsigned long long) __builtin_amdgcn_s_memrealtime() << 20);
#endif
    PROF(7);
    PROF_END;
}

// what every wavefront of a workgroup does first: its share of the code-length tables (the workgroup's only barrier;
// from there on every wavefront is on its own)
#define LOOP_KERNEL_PROLOGUE                                                                                         \
    __shared__ loop_lds LL[LOOP_W];                                                                                  \
    __shared__ uint16_t GL[928]; /* code lengths grouped as new_choose_table compares them (mp3mi_tables::glut) */    \
    const int wv = __builtin_amdgcn_readfirstlane((int) threadIdx.x >> 6);                                           \
    loop_lds &L = LL[wv];                                                                                            \
    for (int i = (int) threadIdx.x; i < 928; i += 64 * LOOP_W) GL[i] = T->glut[i];                                   \
    __syncthreads()

// As many wavefronts as streams, each takes one.
__global__ void __attribute__((amdgpu_num_vgpr(80))) __launch_bounds__(64 * LOOP_W) k_loop(const mp3mi_tables *__restrict__ T, loop_kargs A)
{
    LOOP_KERNEL_PROLOGUE;
    const int block = (int) blockIdx.x * LOOP_W + wv;
#if defined(MP3MI_EMU)
    loop_kargs_p ka = &A;
#else
    // (the record follows the table pointer in the kernel argument segment)
    loop_kargs_p ka = (loop_kargs_p) ((const __attribute__((address_space(4))) char *) __builtin_amdgcn_kernarg_segment_ptr() + sizeof(void *));
    static_assert(alignof(loop_kargs) == 8, "loop_kargs follows an 8-byte argument");
#endif
    if (block < A.geo.n_streams) loop_stream(T, ka, L, GL, block);
}

#if defined(MP3MI_LOOP_PROFILE) && !defined(MP3MI_EMU)
extern "C" void mp3mi_debug_loop_profile(unsigned long long *out)
{
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_loop_prof), sizeof(z));
    hipMemcpyToSymbol(HIP_SYMBOL(g_loop_prof), z, sizeof(z));
}
extern "C" void mp3mi_debug_cb_profile(unsigned long long *out)
{
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cb_prof), sizeof(z));
    hipMemcpyToSymbol(HIP_SYMBOL(g_cb_prof), z, sizeof(z));
}
extern "C" void mp3mi_debug_loop_work(unsigned long long *out, int n_streams)
{
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_loop_work), sizeof(unsigned long long) * (size_t) n_streams);
}
extern "C" void mp3mi_debug_loop_starts(unsigned long long *out, int n_streams)
{
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_loop_start), sizeof(unsigned long long) * (size_t) n_streams);
}
extern "C" void mp3mi_debug_loop_waves(unsigned long long *out, int n_streams)
{
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_loop_wave), sizeof(unsigned long long) * 2 * (size_t) n_streams);
}
#endif

size_t mp3mi_loop_state_size(void) { return sizeof(mp3mi_loop_state); }

// workgroups that are resident together: four of LOOP_W = 4 wavefronts per CU (four wavefronts per SIMD: what stays
// resident beside the feed-forward kernels, batch.cpp)
static int loop_wg_cap(void) { return mp3mi_current_cu_count() * 4; }

// streams one launch of k_loop holds resident, a wavefront each
int mp3mi_loop_resident(void) { return loop_wg_cap() * LOOP_W; }

// A wavefront per stream.  batch.cpp cuts a batch into parts of at most mp3mi_loop_resident() streams, so that a launch is
// resident at once; a larger launch (options.loop_part_streams, tests) is valid all the same: the hardware starts the
// workgroups beyond the resident ones as others end.
void mp3mi_launch_loop(const mp3mi_tables *T, const mp3mi_geom &g, const double *xr, const mp3mi_psy_out *psy,
                       const mp3mi_loop_prep *prep, const int32_t *bits_per_frame, void *loop_state, int16_t *ix,
                       mp3mi_frame_side *side, unsigned *gate_count, mp3mi_loop_place place, hipStream_t st)
{
    const int want = (g.n_streams + LOOP_W - 1) / LOOP_W;
    loop_kargs A;
    A.geo = g; A.xr_all = xr; A.psy = psy; A.prep = prep; A.bits_per_frame = bits_per_frame;
    A.state = (mp3mi_loop_state *) loop_state; A.ix_out = ix; A.side_out = side; A.gate_count = gate_count; A.place = place;
    hipLaunchKernelGGL(k_loop, dim3((unsigned) want), dim3(64 * LOOP_W), 0, st, T, A);
}

// Holds the front stream back until the k_loop launch whose census target is `target` has (all but
// a few of) its wavefronts running, so that the feed-forward kernels queued behind this one fill the
// chip BEHIND k_loop instead of taking its wave slots.  One wavefront, a BOUNDED wait (max_ticks of
// the 100 MHz real-time counter): k_loop fills every register file, so the slot this wavefront
// occupies keeps one k_loop wavefront out until it leaves -- hence "all but a few", and hence short.
__global__ void __launch_bounds__(64) k_gate(const unsigned *__restrict__ count, unsigned target, unsigned max_ticks)
{
#if !defined(MP3MI_EMU)
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((int) (__hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long) max_ticks) break;
        __builtin_amdgcn_s_sleep(32);
    }
#endif
}

void mp3mi_launch_gate(const unsigned *count, unsigned target, unsigned max_ticks, hipStream_t st)
{
    hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, st, count, target, max_ticks);
}

// Holds the loop stream back in front of the LAST k_loop of a call until the call AFTER it has had its first transforms
// (k_fft takes whole CUs' LDS: run behind that k_loop's start they would wait for its end, and everything of the next
// call's first chunk with them -- 20 ms of an otherwise idle chip per call), or until the host lets go (a call that waits
// for results: batch.cpp, hold_release), or until max_ticks of the 100 MHz counter have passed: the hold only ever
// changes WHEN the kernel behind it starts, never what it computes, so running out is harmless.  flag[] is host memory
// mapped into the device's address space; tickets only ever grow.  flag[0]: the highest ticket a NEXT CALL has let go
// (k_hold_release, on the device, behind that call's first transforms); flag[1 + ticket % 8]: the ticket the HOST lets go -- that
// one and no other: the host runs far ahead of the device, and when it waits for the last call it must not let go the holds of
// the calls before it, which the device has not reached yet and whose successors are already queued.  (A ring of eight words:
// a release the device has not looked at yet survives the next seven.)
__global__ void __launch_bounds__(64) k_hold(const unsigned *flag, unsigned ticket, unsigned max_ticks)
{
#if !defined(MP3MI_EMU)
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((int) (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - ticket) < 0 &&
           __hip_atomic_load(flag + 1 + (ticket & 7u), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != ticket) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long) max_ticks) break;
        __builtin_amdgcn_s_sleep(64);
    }
#endif
}
__global__ void __launch_bounds__(64) k_hold_release(unsigned *flag, unsigned ticket)
{
#if defined(MP3MI_EMU)
    if (threadIdx.x == 0 && (int) (*flag - ticket) < 0) *flag = ticket;
#else
    if (threadIdx.x == 0) __hip_atomic_fetch_max(flag, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
}
void mp3mi_launch_hold(const unsigned *flag, unsigned ticket, unsigned max_ticks, hipStream_t st)
{
    hipLaunchKernelGGL(k_hold, dim3(1), dim3(64), 0, st, flag, ticket, max_ticks);
}
void mp3mi_launch_hold_release(unsigned *flag, unsigned ticket, hipStream_t st)
{
    hipLaunchKernelGGL(k_hold_release, dim3(1), dim3(64), 0, st, flag, ticket);
}

ULP_CENSUS_ACCESSOR(mp3mi_debug_ulp_census_loop)
