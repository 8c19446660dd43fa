/* Shared declarations of the device side: table block, per-granule records, wave helpers.
 *
 * Layout rule: every inter-kernel buffer is stream-major; within a stream the chunk's
 * granules are consecutive.  One wavefront (64 lanes) is the unit of work everywhere.
 */
#ifndef MP3MI_DEV_H
#define MP3MI_DEV_H

#include <stdint.h>

#if defined(MP3MI_EMU)
#include "hipemu.h"
#define MP3MI_DEVFN static inline
#else
#include <hip/hip_runtime.h>
#define MP3MI_DEVFN static __device__ __forceinline__
#endif

#define MP3MI_CBANDS 63
#define MP3MI_CBANDS_S 42
#define MP3MI_HBLK 513
#define MP3MI_HBLK_S 129
#define MP3MI_HBLK_P 544      /* row pitch of energy_l in floats: rows start on a 128-byte line (k_part reads whole lines) */
#define MP3MI_PART_P 64       /* row pitch of the partition energies eb / cb handed from k_part to k_psy */
#define MP3MI_FFT_BINS 312   /* raw bins handed from k_fft to k_cw per (granule, channel): short lines 2..51 of the three
                                windows as (re, im), then re[6], im[6] of long lines 0..5 */
#if defined(MP3MI_FFT_SWZ_RUNTIME) /* tools/exp/fft_swz_search.cpp: the swizzle as data */
extern unsigned mp3mi_fft_swz_col[10];
static inline int mp3mi_fft_swz_rt(int p)
{
    int r = p;
    for (int b = 4; b < 10; b++)
        if (p & (1 << b)) r ^= (int) mp3mi_fft_swz_col[b];
    return r;
}
#define MP3MI_FFT_SWZ(p) mp3mi_fft_swz_rt(p)
#else
/* LDS index of element p of an FFT array: index bits 4..9 are XORed into the low five bits, bit b with the constant
 * MP3MI_FFT_SWZ_COLS names for it (bit 4's constant stays below 16: the map is a bijection; all of them even: elements 2 j and
 * 2 j + 1 stay neighbours, which fft_leaves' 16-byte accesses rely on -- table build checks it).  GF(2)-linear, so
 * SWZ(a ^ b) == SWZ(a) ^ SWZ(b): the kernels split an index into its lane part and a compile-time part.  The
 * constants -- and with them the placement of the butterflies, fft_placement.h -- come out of a search for the
 * fewest LDS bank conflicts of the butterfly programs (tools/exp/fft_swz_search.cpp); the three short transforms
 * are swizzled as ONE 768-element array, so that the same butterfly of two windows does not fall on the same banks. */
#define MP3MI_FFT_SWZ_COLS 8, 24, 2, 10, 8, 20
#define MP3MI_FFT_SWZ_PICK(c4, c5, c6, c7, c8, c9, p) \
    ((p) ^ ((((p) >> 4) & 1) * (c4)) ^ ((((p) >> 5) & 1) * (c5)) ^ ((((p) >> 6) & 1) * (c6)) ^ ((((p) >> 7) & 1) * (c7)) ^ \
     ((((p) >> 8) & 1) * (c8)) ^ ((((p) >> 9) & 1) * (c9)))
#define MP3MI_FFT_SWZ_APPLY(cols, p) MP3MI_FFT_SWZ_PICK(cols, p)
#define MP3MI_FFT_SWZ(p) MP3MI_FFT_SWZ_APPLY(MP3MI_FFT_SWZ_COLS, p)
#endif
/* FFT butterfly programs (tables_host.cpp): rounds of 64 fused butterflies, one per lane */
#define MP3MI_FFT_DUMMY 1024      /* elements 1024 + lane: what the idle lanes of a round work on */
#define MP3MI_FFT_DUMMY_S 768     /* the same behind the three 256-point transforms */
#define MP3MI_FFT_MAX_ROUNDS 48
#define MP3MI_FFT_REG_ROWS_L 9    /* rows of fft_regtw_l */
#define MP3MI_FFT_PROG_WORDS 5376 /* capacity of the long program in 32-bit words; checked at table build */
#define MP3MI_FFT_PROG_WORDS_S 4224 /* capacity of the short program */
/* the header words of the rounds of the long and of the short program (fft_hdr_* below): k_fft is compiled
   for exactly these sequences -- straight-line code, no per-round dispatch -- and table build checks that the
   generator still produces them (MP3MI_FFT_INFO=1 prints the lists) */
#define MP3MI_FFT_HDRS_L 15, 6, 15, 6, 15, 6, 15, 6, 15
#define MP3MI_FFT_HDRS_S 2, 14, 6, 15, 6, 15, 6, 15
#define MP3MI_PCM_HIST 1056   /* samples per channel a call needs from before its first sample: the filterbank of the granule
                                 before the call (k_filter recomputes it: 576 + 480 taps); the FFT window reaches back 768 */
#define MP3MI_POW43_N 8208
#define MP3MI_STEP_MIN (-400)
#define MP3MI_STEP_N 801

/* Inputs the reference dies on (an assertion fails; tests/golden/coverage_notes.json): status codes, the public names
 * are MP3MI_STREAM_* in include/mp3mi.h (batch.cpp checks that they agree).  A stream's status word is the LAST word of
 * its mp3mi_loop_state: code | index of the frame it happened in << 8 (the number of frames: in the final flush). */
#define MP3MI_DEV_ABORT_GLOBAL_GAIN 1 /* assert( cod_info->global_gain < 256 ), src/loop.c:358 */
#define MP3MI_DEV_ABORT_HUFF_BITS 2   /* assert( max_bits >= 0 ), src/loop.c:579 */
#define MP3MI_DEV_ABORT_FLUSH_SLOT 3  /* assert( l ), src/formatBitstream.c:390, from BF_FlushBitstream's remainder call */
#define MP3MI_DEV_ABORT_REPORTED 0x40000000 /* set in the status word once a streaming call has counted the stream in `voided` (k_format.hip) */
/* the status word: the frame field is 22 bits wide and SATURATES (a stream fed call by call passes frame 2^22 after ~30 h of
 * audio), so that it never reaches the flag above or the sign */
#define MP3MI_DEV_STATUS_FRAME_MAX 0x3fffff
#define MP3MI_DEV_STATUS(code, frame) ((int32_t) (code) | (int32_t) (((frame) < MP3MI_DEV_STATUS_FRAME_MAX ? (frame) : MP3MI_DEV_STATUS_FRAME_MAX) << 8))

/* Data movement the reference's FFT ends with; folded into the read-out tables fft_rd_* (tables_host.cpp) */
enum {
    FOP_NEG = 1,    /* x[a]=-x[a]                                        (src/subs.c:523) */
    FOP_SWAPNN = 6, /* t=x[a]; x[a]=-x[b]; x[b]=-t                       (src/subs.c:509-513) */
    FOP_SWAPN = 7   /* t=x[a]; x[a]=-x[b]; x[b]=t                        (src/subs.c:516-522) */
};

/* All read-only tables, one block in device memory.  Values are produced on the host with
 * the host's libm exactly as the reference's init code does (tables_host.cpp). */
typedef struct {
    int32_t rate_idx;
    int32_t sfb_l[23], sfb_s[14];
    uint8_t sfb_of_line_l[576];      /* long sfb index of each line (21 = above sfb 20) */
    uint8_t sfb_of_line_s[576];      /* short: sfb*3+window of line l*3+w              */
    /* calc_noise as partial sums (k_loop): job j < 64 adds nj_count elements of the noise terms from
       element nj_first on (stride 1 for long blocks [0], 3 for short blocks [1]); the parts of band b
       are the jobs [nj_job0[b], nj_job0[b] + nj_njobs[b]) */
    int16_t nj_first[2][64];
    uint8_t nj_count[2][64], nj_job0[2][36], nj_njobs[2][36];
    uint8_t nj_max[2];               /* the longest job */
    uint8_t nj_seg[2][64];           /* bit d (1, 2, 4, 8, 16): job j + d belongs to the same band as job j (bands have < 32 jobs) */
    /* what lane l of k_loop needs to know about "its" lines and noise jobs, packed so that one 8-byte load each brings it in
       where it is used (the distortion loop) instead of a dozen registers living through the whole search; [0] long, [1] short:
       lane_bands: [0] band (= band lane) of the lane's PAIR k < 5 -- lines 2 (l + 64 k), + 1: one band, k_loop.hip -- in bits 6k .. 6k+5;
                   [1] of its value j < 10 -- line 2 (l + 64 (j / 2)) + j % 2 -- in bits 6j .. 6j+5 (63: the line does not exist);
       lane_jobs: nj_first (10 bits) | nj_count << 10 (8) | nj_seg << 18 (5) | nj_job0 << 23 (6) | nj_njobs << 29 (6) |
                  lines of band l << 35 (8) | first line of band l << 43 (10; short: l = sfb * 3 + window -> first * 3 + window) */
    uint64_t lane_bands[2][64], lane_jobs[2][64];
    /* 1 / (lines of band l): the first tier of calc_noise (k_loop.hip) takes a band's mean by a multiplication -- its value only has to
       lie within 1e-12 of the reference's quotient --, the exact tier divides */
    double lane_inv_lines[2][64];
    /* subdivide (src/loop.c:1596-1706) for blocks without window switching, by big_values:
       region0_count | region1_count << 4 | address1 << 8 | address2 << 18 (tables_host.cpp) */
    uint32_t subdiv_lut[289];
    /* psy */
    float window[1024], window_s[256];
    int32_t numlines_pe[MP3MI_CBANDS];       /* numlines[] as left by L3para_read (quirk) */
    int32_t part_l_start[MP3MI_CBANDS + 1];  /* lines of long partition b: [start, end)   */
    int32_t part_s_start[MP3MI_CBANDS_S + 1];
    int32_t part_l_covered, part_s_covered;  /* lines >= covered belong to partition 0 (quirk) */
    double minval[MP3MI_CBANDS], qthr_l[MP3MI_CBANDS], norm_l[MP3MI_CBANDS];
    double qthr_s[MP3MI_CBANDS_S], exp_snr_s[MP3MI_CBANDS_S];
    double s3_l[MP3MI_CBANDS][MP3MI_CBANDS];
    double s3_lt[MP3MI_CBANDS][MP3MI_PART_P]; /* transposed, [k][b]: what k_psy's lanes b read together at the rates with dense rows */
    int32_t s3_lo[MP3MI_CBANDS], s3_hi[MP3MI_CBANDS];
    int32_t bu_l[21], bo_l[21], bu_s[12], bo_s[12];
    double w1_l[21], w2_l[21], w1_s[12], w2_s[12];
    /* FFT programs: per round a header word (bit 0: eight-operand butterflies, bit 1: some lane rotates by a
       twiddle, bit 2: some lane rotates by SQHALF, bit 3: the next round reads what this rank wrote) and its
       blocks in fft_prog_*; fft_rd_*: LDS position of bin i's real part | sign << 15 | imaginary part's << 16 |
       sign << 31 once the butterflies are done (step 5 and the bit reversal of the reference folded in) */
    int32_t fft_nround_l, fft_nround_s, fft_nword_l, fft_nword_s;
    uint32_t fft_hdr_l[MP3MI_FFT_MAX_ROUNDS], fft_hdr_s[MP3MI_FFT_MAX_ROUNDS];
    uint32_t fft_prog_l[MP3MI_FFT_PROG_WORDS] __attribute__((aligned(16))), fft_prog_s[MP3MI_FFT_PROG_WORDS_S] __attribute__((aligned(16)));
    uint32_t fft_rd_l[MP3MI_HBLK], fft_rd_s[MP3MI_HBLK_S];
    /* the rotations of the butterflies k_fft runs in registers, before the program: [row][lane] {cn, spcn, smcn, flags}
       (tables_host.cpp, FftGen::build) */
    uint32_t fft_regtw_l[MP3MI_FFT_REG_ROWS_L * 256] __attribute__((aligned(16))), fft_regtw_s[256] __attribute__((aligned(16)));
    /* the blocks of 8 points and fewer, which k_fft finishes in registers behind the program: [lane] the LDS positions of the lane's
       eight pairs of elements (four of run A, four of run B, 16 bits each), the kind of leaf in bits 14..15 of word 0 (3: an idle
       lane) (tables_host.cpp, FftGen::in_leaf) */
    uint32_t fft_leaf_l[256] __attribute__((aligned(16))), fft_leaf_s[256] __attribute__((aligned(16)));
    /* filterbank + MDCT */
    double enwindow[512];
    double filt[32][32];             /* the 31 used columns per subband: 0..15, 33..47  */
    double mdct_win[4][36], cos_s[6][12], cos_l[18][36], ca[8], cs[8];
    /* long-block (type 0) transform, src/mdct.c:199-509, in shared-subexpression form: per band 26
       values V (0-8: fin[j]-fin[17-j]; 9-17: fin[18+j]+fin[35-j]; 18-23: the six 6-operand groups;
       24-25: the two 18-operand groups, operand lists below, bit 7 = subtract/negate); output m is
       the ordered sum over t < mdct_nterm[m] of V[mdct_vidx[m][t]] * mdct_vcoef[m][t] */
    uint8_t mdct_vidx[18][18], mdct_nterm[18];
    uint8_t mdct_full_row[12], mdct_small_row[6]; /* rows whose terms are V[0..17] in order / rows with at most 6 terms */
    uint8_t mdct_g_ops[6][18], mdct_h_ops[2][18];
    double mdct_vcoef[18][18];
    /* quantiser */
    double pow_nint_tab[2049];       /* (i-0.4054)^(4/3); [0]=0, [2048]=+inf sentinel   */
    double pow43[MP3MI_POW43_N];     /* i^(4/3) */
    double step[MP3MI_STEP_N];       /* 2^(q/4), q = MP3MI_STEP_MIN + i */
    double pretab_xr[4], pretab_xmin[4]; /* sqrt(2)^n, sqrt(2)^(2n), n = 0..3 */
    double sqrt2, log2;
    /* Huffman */
    uint16_t ht_off[34];
    uint8_t ht_xlen[34], ht_ylen[34], ht_linbits[34];
    uint16_t ht_linmax[34];
    uint8_t ht_len[1440];
    uint32_t ht_code[1440];
    /* code lengths grouped the way new_choose_table compares tables: per group and (x,y) cell
       the lengths of its tables INCLUDING the cell's sign bits, 5 bits each.  Groups at offsets 0 {1}, 4 {2,3}, 13 {5,6},
       29 {7,8,9}, 65 {10,11,12}, 129 {13,15}, 385 {15,24}, 641 {16,24}, 897 {32,33}; the cells of the two groups with linbits
       carry in bits 10..11 how many of x, y are escapes */
    uint16_t glut[928];
} mp3mi_tables;

/* Output of the psychoacoustic stage for one (granule, channel); mirrors what
 * L3psycho_anal hands back to its caller (src/l3psy.h:33). */
typedef struct {
    double pe;
    double ratio_l[21];
    double ratio_s[12][3];
    int32_t block_type;
    int32_t pad;
} mp3mi_psy_out;

/* Stateless per-(granule, channel) inputs of the iteration loop, computed for all granules in
 * parallel (k_mdct's tail; k_prep for the records it lists) so that the serial kernel starts from them: allowed
 * distortion (calc_xmin, src/loop.c:1085), the integer log-energies calc_scfsi stores (src/loop.c:631-667)
 * and the start value of the quantiser step (quantanf_init, src/loop.c:369).  One 472-byte record per (granule,
 * channel), record-major: k_mdct's lanes write a record's fields side by side and k_loop's wavefront reads them so. */
typedef struct {
    double xmin[36];                 /* long: [sfb], sfb < 21; short: [sfb*3 + window], sfb < 12 */
    int32_t sc_en[21], sc_xm[21];    /* written for non-short granules only */
    int32_t sc_en_tot, sc_xrmax, q0, nonzero;
} mp3mi_loop_prep;

/* Side information of one (granule, channel) as the iteration loop leaves it
 * (subset of gr_info, src/l3side.h:60-87, that the formatter needs). */
typedef struct {
    int32_t part2_3_length, big_values, count1, global_gain, scalefac_compress;
    int32_t window_switching_flag, block_type, table_select[3];
    int32_t region0_count, region1_count, preflag, count1table_select, part2_length;
    int32_t scalefac[39];            /* long: [0..20]; short: [sfb*3+window], sfb<12 (slot 36.. unused) */
} mp3mi_gr_side;

typedef struct {
    int32_t main_data_begin, resvDrain, scfsi[2][4];
    mp3mi_gr_side gr[2][2];
} mp3mi_frame_side;

/* The quantiser's first tier (k_loop.hip, loop_quantize) estimates x^(3/4) with the raw hardware square root and
 * exp2 (v_sqrt_f32, v_exp_f32: 1 ulp each); its guard band budgets 7e-7 relative for the two roots, the exp2
 * and the roundings between them.  mp3mi_debug_fastmath_bounds (k_debug.hip) measures these very expressions
 * on the device.  The CPU test build substitutes libm. */
#if defined(MP3MI_EMU)
#define LOOP_FAST_SQRTF(x) __builtin_sqrtf(x)
#define LOOP_FAST_EXP2F(x) __builtin_exp2f(x)
/* v_cvt_pknorm_u16_f32 as the device executes it (mp3mi_debug_pknorm_bound, k_debug.hip): each operand clamped to [0, 1],
   times 65535 -- exact in double -- rounded to the nearest integer, ties to even; a | b << 16 */
static inline unsigned mp3mi_emu_pknorm_u16(float a, float b)
{
    const double ca = a > 0.0f ? (a < 1.0f ? (double) a : 1.0) : 0.0, cb = b > 0.0f ? (b < 1.0f ? (double) b : 1.0) : 0.0;
    return (unsigned) __builtin_rint(ca * 65535.0) | ((unsigned) __builtin_rint(cb * 65535.0) << 16);
}
#define LOOP_PKNORM_U16(a, b) mp3mi_emu_pknorm_u16((a), (b))
/* v_pk_max_u16: the larger of each half (k_loop's region maxima run on pair words x | y << 16) */
struct mp3mi_u16x2 { unsigned short x, y; };
static inline mp3mi_u16x2 mp3mi_emu_pk_max_u16(mp3mi_u16x2 a, mp3mi_u16x2 b) { mp3mi_u16x2 r = {a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y}; return r; }
#define LOOP_PK_MAX_U16(a, b) mp3mi_emu_pk_max_u16((a), (b))
static inline mp3mi_u16x2 mp3mi_emu_pk_min_u16(mp3mi_u16x2 a, mp3mi_u16x2 b) { mp3mi_u16x2 r = {a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y}; return r; }
#define LOOP_PK_MIN_U16(a, b) mp3mi_emu_pk_min_u16((a), (b))
#else
#define LOOP_FAST_SQRTF(x) __builtin_amdgcn_sqrtf(x)
#define LOOP_FAST_EXP2F(x) __builtin_amdgcn_exp2f(x) /* |x| < 80 here: no denormal range to care for */
typedef unsigned short mp3mi_u16x2 __attribute__((ext_vector_type(2)));
#define LOOP_PKNORM_U16(a, b) __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pknorm_u16((a), (b)))
#define LOOP_PK_MAX_U16(a, b) __builtin_elementwise_max((a), (b))
#define LOOP_PK_MIN_U16(a, b) __builtin_elementwise_min((a), (b))
#endif

/* Diagnostic build only (-DMP3MI_ULP_CENSUS, tools/gpu_ulp_census.sh; never the product build): how often does a
 * value that came out of a transcendental sit so close to the rounding or comparison it feeds that a libm which is off by
 * one ulp -- as glibc's may be -- could decide it differently?  Per site three counters: calls, calls within the band a
 * one-ulp error of every libm result involved can move the value by ("near"), calls within a band 2^20 times wider
 * ("wide": the statistics behind an estimate where "near" is too rare to be seen).  DESIGN.md section 2. */
#if defined(MP3MI_ULP_CENSUS) && !defined(MP3MI_EMU)
enum { UC_PHASE, UC_NB, UC_PE_ATTACK, UC_PE_RESV, UC_QUANTANF, UC_SCFSI_LOG, UC_CW_STEP, UC_CW_RANGE,
       UC_L12_C, UC_L12_BC, UC_L12_EXP, UC_L12_SNR, /* Layers I / II (k_l12.hip) */
       UC_CW_REACH, /* k_part: partitions with a near step of UC_CW_STEP (calls), those whose sum changes when the steps' floats move by one ulp (near) */
       UC_CW_NB,    /* k_psy: thresholds nb whose spread unpredictability such a changed sum reaches (calls), those that come out as another float (near) */
       UC_N };
static __device__ unsigned long long g_ulp_census[UC_N][3]; /* (one copy per translation unit: no relocatable device code here) */
/* adds this translation unit's counters to out[UC_N][3] and clears them */
#define ULP_CENSUS_ACCESSOR(name)                                                                       \
    extern "C" void name(unsigned long long *out)                                                       \
    {                                                                                                   \
        unsigned long long h[UC_N][3], z[UC_N][3] = {};                                                 \
        hipDeviceSynchronize();                                                                         \
        hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ulp_census), sizeof(h));                                    \
        hipMemcpyToSymbol(HIP_SYMBOL(g_ulp_census), z, sizeof(z));                                      \
        for (int i = 0; i < UC_N; i++) for (int j = 0; j < 3; j++) out[3 * i + j] += h[i][j];          \
    }
#define ULP_CENSUS(site, near_, wide_) do { atomicAdd(&g_ulp_census[site][0], 1ull); if (wide_) atomicAdd(&g_ulp_census[site][2], 1ull); \
                                            if (near_) atomicAdd(&g_ulp_census[site][1], 1ull); } while (0)
#else
#define ULP_CENSUS(site, near_, wide_) do { } while (0)
#define ULP_CENSUS_ACCESSOR(name)
#endif

/* ---- wave helpers (64 lanes) ---- */
MP3MI_DEVFN int wave_lane(void) { return (int) (threadIdx.x & 63); }
/* The lane index behind an optimisation barrier: everything derived from it (addresses, masks) is
 * recomputed where it is used instead of being hoisted to the top of a long kernel and kept alive
 * (in registers, then in scratch) across all of it. */
MP3MI_DEVFN int wave_lane_here(void)
{
    int l = (int) (threadIdx.x & 63);
#if !defined(MP3MI_EMU)
    asm volatile("" : "+v"(l));
#endif
    return l;
}

/* A value behind an optimisation barrier: what is computed from the result cannot be hoisted above this point.
 * Used at the top of RARELY executed blocks inside k_loop's distortion loop, whose per-line address and constant
 * arithmetic the compiler would otherwise hoist to the top of the granule and keep alive -- in scratch memory --
 * across the whole search. */
MP3MI_DEVFN unsigned long long wave_opaque_u64(unsigned long long v)
{
#if !defined(MP3MI_EMU)
    unsigned lo = (unsigned) v, hi = (unsigned) (v >> 32);
    asm volatile("" : "+v"(lo), "+v"(hi));
    v = ((unsigned long long) hi << 32) | lo;
#endif
    return v;
}

#if defined(MP3MI_EMU)
MP3MI_DEVFN int wave_sum_i32(int v)
{
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
MP3MI_DEVFN int wave_max_i32(int v)
{
    for (int m = 32; m >= 1; m >>= 1) { int o = __shfl_xor(v, m); v = o > v ? o : v; }
    return v;
}
/* value of v in lane `lane`; lane must be wave-uniform */
MP3MI_DEVFN int wave_readlane_i32(int v, int lane) { return __shfl(v, lane); }
#else
/* DPP reductions (gfx9 family): xor-1 and xor-2 inside each quad, then half-row and row mirrors
 * give every lane of a 16-lane row the row total; row_bcast15 / row_bcast31 carry the totals
 * across rows into lane 63, which is read back as a wave-uniform scalar.  The broadcasts run with all rows
 * enabled (rows that have no source take 0, the identity of a sum and of a maximum of non-negative values;
 * row 2 also picks up row 1's total in the first broadcast, which never reaches lane 63's chain): a step
 * with a partial row mask has to keep the other rows' old values and costs three instructions, not one. */
#define MP3MI_DPP(v, ctrl, rmask) __builtin_amdgcn_update_dpp(ident, (v), (ctrl), (rmask), 0xf, false)
MP3MI_DEVFN int wave_sum_i32(int v)
{
    const int ident = 0;
    v += MP3MI_DPP(v, 0xB1, 0xf);  /* quad_perm [1,0,3,2] */
    v += MP3MI_DPP(v, 0x4E, 0xf);  /* quad_perm [2,3,0,1] */
    v += MP3MI_DPP(v, 0x141, 0xf); /* row_half_mirror */
    v += MP3MI_DPP(v, 0x140, 0xf); /* row_mirror */
    v += MP3MI_DPP(v, 0x142, 0xf); /* row_bcast15: row 3 takes row 2's total, row 1 row 0's */
    v += MP3MI_DPP(v, 0x143, 0xf); /* row_bcast31: row 3 takes lane 31's = rows 0 + 1 */
    return __builtin_amdgcn_readlane(v, 63);
}
MP3MI_DEVFN int wave_max_i32(int v) /* values >= 0 */
{
    const int ident = 0;
    int o;
    o = MP3MI_DPP(v, 0xB1, 0xf); v = o > v ? o : v;
    o = MP3MI_DPP(v, 0x4E, 0xf); v = o > v ? o : v;
    o = MP3MI_DPP(v, 0x141, 0xf); v = o > v ? o : v;
    o = MP3MI_DPP(v, 0x140, 0xf); v = o > v ? o : v;
    o = MP3MI_DPP(v, 0x142, 0xf); v = o > v ? o : v;
    o = MP3MI_DPP(v, 0x143, 0xf); v = o > v ? o : v;
    return __builtin_amdgcn_readlane(v, 63);
}
MP3MI_DEVFN int wave_readlane_i32(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
#endif

/* A wave-uniform pointer displaced by a zero the compiler cannot see through: addresses derived from the result are
 * formed where they are used (scalar loads with an immediate offset) instead of being hoisted out of a loop by the
 * dozen -- or the loads themselves -- and kept in SGPRs and, when those run out, in VGPR lanes.  (The pointer itself
 * is not laundered: it would lose what makes its loads scalar, that it is a read-only kernel argument.) */
template <typename P> MP3MI_DEVFN const P *wave_uniform_here(const P *p)
{
#if !defined(MP3MI_EMU)
    int z = 0;
    asm volatile("" : "+s"(z));
    p = (const P *) ((const char *) p + z);
#endif
    return p;
}

/* Several reductions at once, step by step in lock-step.  A DPP instruction that reads the result of
 * the previous VALU instruction needs two wait states (an s_nop each time in a lone reduction: 6 DPP +
 * 6 s_nop); interleaved, the other chains' steps fill those slots.  NSUM sums first, then NMAX maxima
 * (values >= 0); results are wave-uniform. */
template <int NSUM, int NMAX>
MP3MI_DEVFN void wave_reduce_i32(int (&v)[NSUM + NMAX])
{
#if defined(MP3MI_EMU)
    for (int k = 0; k < NSUM; k++) v[k] = wave_sum_i32(v[k]);
    for (int k = NSUM; k < NSUM + NMAX; k++) v[k] = wave_max_i32(v[k]);
#else
    const int ident = 0;
#define MP3MI_RSTEP(ctrl, rmask)                                                                   \
    _Pragma("unroll") for (int k = 0; k < NSUM + NMAX; k++) {                                      \
        const int o = MP3MI_DPP(v[k], ctrl, rmask);                                                \
        v[k] = (k < NSUM) ? v[k] + o : (o > v[k] ? o : v[k]);                                      \
    }
    MP3MI_RSTEP(0xB1, 0xf)
    MP3MI_RSTEP(0x4E, 0xf)
    MP3MI_RSTEP(0x141, 0xf)
    MP3MI_RSTEP(0x140, 0xf)
    MP3MI_RSTEP(0x142, 0xf)
    MP3MI_RSTEP(0x143, 0xf)
#undef MP3MI_RSTEP
#pragma unroll
    for (int k = 0; k < NSUM + NMAX; k++) v[k] = __builtin_amdgcn_readlane(v[k], 63);
#endif
}
/* The same, but the results stay in LANE 63 of the vector registers (what the other lanes hold is unspecified): for a
 * caller that goes on computing with them there and reads back only its final values (k_loop's table choice). */
template <int NSUM, int NMAX>
MP3MI_DEVFN void wave_reduce_keep_i32(int (&v)[NSUM + NMAX])
{
#if defined(MP3MI_EMU)
    wave_reduce_i32<NSUM, NMAX>(v);
#else
    const int ident = 0;
#define MP3MI_RSTEP(ctrl, rmask)                                                                   \
    _Pragma("unroll") for (int k = 0; k < NSUM + NMAX; k++) {                                      \
        const int o = MP3MI_DPP(v[k], ctrl, rmask);                                                \
        v[k] = (k < NSUM) ? v[k] + o : (o > v[k] ? o : v[k]);                                      \
    }
    MP3MI_RSTEP(0xB1, 0xf)
    MP3MI_RSTEP(0x4E, 0xf)
    MP3MI_RSTEP(0x141, 0xf)
    MP3MI_RSTEP(0x140, 0xf)
    MP3MI_RSTEP(0x142, 0xf)
    MP3MI_RSTEP(0x143, 0xf)
#undef MP3MI_RSTEP
#endif
}
/* The tails of three reductions side by side: what lane 63 of `src` holds goes to lane 63 - n of `old` (n = 1, 2), the lanes
 * from 63 - n + 1 up keep `old`, what the lanes below 63 - n hold afterwards is unspecified.  wave_put_lane: one lane takes a
 * wave-uniform value.  wave_tail_sum3: lane 61's + lane 62's + lane 63's value, wave-uniform. */
MP3MI_DEVFN int wave_tail_place(int old, int src, int n)
{
#if defined(MP3MI_EMU)
    const int v = __shfl(src, 63);
    return wave_lane() == 63 - n ? v : old;
#else
    return n == 1 ? __builtin_amdgcn_update_dpp(old, src, 0x101, 0x8, 0x8, false)  /* row_shl:1, row 3, lanes 60..63 */
                  : __builtin_amdgcn_update_dpp(old, src, 0x102, 0x8, 0x8, false); /* row_shl:2 */
#endif
}
template <int LANE> MP3MI_DEVFN int wave_put_lane(int old, int uniform_value)
{
#if defined(MP3MI_EMU)
    return wave_lane() == LANE ? uniform_value : old;
#else
    int r = old;
    asm("v_writelane_b32 %0, %1, %2" : "+v"(r) : "s"(uniform_value), "i"(LANE));
    return r;
#endif
}
MP3MI_DEVFN int wave_tail_sum3(int v)
{
#if defined(MP3MI_EMU)
    return __shfl(v, 61) + __shfl(v, 62) + __shfl(v, 63);
#else
    int t = v + __builtin_amdgcn_update_dpp(0, v, 0x111, 0x8, 0x8, true); /* row_shr:1: lane 63 takes lane 62's */
    t += __builtin_amdgcn_update_dpp(0, v, 0x112, 0x8, 0x8, true);        /* row_shr:2: ... and lane 61's */
    return __builtin_amdgcn_readlane(t, 63);
#endif
}
/* the double that lane (this lane + D) mod 64 holds: two ds_bpermute_b32 whose lane offset is the instruction's immediate
 * (lane4 = 4 * this lane) -- __shfl_down computes the partner's address and clamps it at the wavefront's end, three vector
 * instructions a call that a caller who discards what comes from beyond its segment does not need */
template <int D> MP3MI_DEVFN double wave_down_f64(double v, int lane4)
{
#if defined(MP3MI_EMU)
    return __shfl(v, (wave_lane() + D) & 63);
#else
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const int lo = __builtin_amdgcn_ds_bpermute(lane4 + 4 * D, (int) (unsigned) b);
    const int hi = __builtin_amdgcn_ds_bpermute(lane4 + 4 * D, (int) (unsigned) (b >> 32));
    return __builtin_bit_cast(double, ((unsigned long long) (unsigned) hi << 32) | (unsigned long long) (unsigned) lo);
#endif
}
/* OR of a word over the wavefront (wave-uniform result) */
MP3MI_DEVFN unsigned wave_or_u32(unsigned v)
{
#if defined(MP3MI_EMU)
    int x = (int) v;
    for (int m = 32; m >= 1; m >>= 1) x |= __shfl_xor(x, m);
    return (unsigned) x;
#else
    const int ident = 0;
    int x = (int) v;
    x |= MP3MI_DPP(x, 0xB1, 0xf);
    x |= MP3MI_DPP(x, 0x4E, 0xf);
    x |= MP3MI_DPP(x, 0x141, 0xf);
    x |= MP3MI_DPP(x, 0x140, 0xf);
    x |= MP3MI_DPP(x, 0x142, 0xf);
    x |= MP3MI_DPP(x, 0x143, 0xf);
    return (unsigned) __builtin_amdgcn_readlane(x, 63);
#endif
}
/* wave maximum of unsigned values: DPP steps with the maximum folded into the move (a lane without a source reads 0) */
MP3MI_DEVFN unsigned wave_max_u32(unsigned v)
{
#if defined(MP3MI_EMU)
    for (int m = 32; m >= 1; m >>= 1) { const unsigned o = __shfl_xor(v, m); v = o > v ? o : v; }
    return v;
#else
    const int ident = 0;
#define MP3MI_UMAX_STEP(ctrl) { const unsigned o = (unsigned) MP3MI_DPP((int) v, ctrl, 0xf); v = o > v ? o : v; }
    MP3MI_UMAX_STEP(0xB1) MP3MI_UMAX_STEP(0x4E) MP3MI_UMAX_STEP(0x141) MP3MI_UMAX_STEP(0x140) MP3MI_UMAX_STEP(0x142) MP3MI_UMAX_STEP(0x143)
#undef MP3MI_UMAX_STEP
    return (unsigned) __builtin_amdgcn_readlane((int) v, 63);
#endif
}
MP3MI_DEVFN int wave_min_i32(int v)
{
    for (int m = 32; m >= 1; m >>= 1) { int o = __shfl_xor(v, m); v = o < v ? o : v; }
    return v;
}
MP3MI_DEVFN double wave_max_f64(double v)
{
    for (int m = 32; m >= 1; m >>= 1) { double o = __shfl_xor(v, m); v = o > v ? o : v; }
    return v;
}
MP3MI_DEVFN int wave_bcast_i32(int v, int lane) { return __shfl(v, lane); }
MP3MI_DEVFN double wave_bcast_f64(double v, int lane) { return __shfl(v, lane); }
MP3MI_DEVFN int wave_any(int p) { return __ballot(p) != 0ull; }

/* Orders the LDS traffic of ONE wavefront: what any lane wrote before is visible to every lane
 * after.  The hardware executes a wave's LDS instructions in order, so on the device this only has
 * to stop the compiler from moving accesses across; the emulator needs a real rendezvous. */
#if defined(MP3MI_EMU)
MP3MI_DEVFN void wave_sync(void) { __syncthreads(); }
#else
MP3MI_DEVFN void wave_sync(void)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
#endif

#endif
