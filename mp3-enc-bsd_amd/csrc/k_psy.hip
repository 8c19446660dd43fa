// Psychoacoustic model, threshold part: unpredictability of lines 0..5, partition energies,
// spreading, tonality, SNR, thresholds with pre-echo control, perceptual entropy, block-type
// decision and the signal-to-mask ratios per scalefactor band.
//
// Replaces src/l3psy.c:452-456, 496-512, 555-739 (and sprdngf1/2, :1062-1084).  The part of
// L3psycho_anal that carries state from call to call lives here: one wavefront owns one
// (stream, channel) and walks the chunk's granules in order, keeping in registers/LDS what
// the reference keeps in function statics (r/phi history of the last two granules, nb_1,
// nb_2, blocktype_old, the held ratio[] / ratio_s[]).  Lane b handles threshold partition b.
//
// Precision map: cb/ecb/nb accumulate in f32 with f64 products rounded
// at every step, everything else is f64; log/exp/sin/cos come from dmath.h.
#include "mp3mi_host.h"
#include "dmath.h"

#define R_LN_TO_LOG10 0.2302585093 /* src/common.h:204 */

typedef struct {
    float r1[6], p1[6], r2[6], p2[6]; /* r and phi of the previous / second previous granule */
    double nb_1[MP3MI_CBANDS], nb_2[MP3MI_CBANDS];
    double ratio_l[21], ratio_s[36];
    int32_t blocktype_old, pad;
} mp3mi_psy_state;

struct psy_lds {
    double eb[MP3MI_CBANDS], thr[MP3MI_CBANDS], pev[MP3MI_CBANDS];
    float cb[MP3MI_CBANDS];
#if defined(MP3MI_ULP_CENSUS) && !defined(MP3MI_EMU)
    float cb_lo[MP3MI_CBANDS], cb_hi[MP3MI_CBANDS]; // k_part's shadow sums (census site UC_CW_NB)
#endif
    double ebs[MP3MI_CBANDS_S], thrs[MP3MI_CBANDS_S];
    double held_l[21], held_s[36];
    double pe;
};

// ---- k_part: partition energy and weighted unpredictability (src/l3psy.c:496-512, 565-578) ----
// eb[b] and cb[b] are ORDER-SENSITIVE sums over the lines of partition b (cb even rounds to float
// at every step), i.e. serial chains; they depend on nothing but this granule's spectrum and the
// r/phi of the two granules before it.  So, as in k_prep, the lanes of a wavefront are 64
// different (granule, channel) records, each walking its own 513 lines in index order, closing
// partitions as it passes their last line; lines beyond the table's coverage continue partition
// 0's chain.  A lane fetches one whole 128-byte line (32 energies) of its row at a time and uses it
// up before touching the next: with 64 lanes x 20 wavefronts x 32 CUs walking different rows, lines
// fetched 16 bytes at a time were evicted from the 4 MB L2 between uses (8x over-fetch measured).
// k_psy then starts every granule from eb/cb.
struct __attribute__((aligned(16))) psy_f4 { float x, y, z, w; };

struct part_walk {
    double eb, eb0;
    float cb, cb0;
    int b, pend; // open partition and the line after its last (wave-uniform)
#if defined(MP3MI_ULP_CENSUS) && !defined(MP3MI_EMU)
    // census site UC_CW_REACH: the open partition's sum as it would stand had every float of a "near" step come out one
    // float ulp lower / higher -- does the difference survive to the value that leaves the kernel?
    float cb_lo, cb_hi;
    int near_steps;
#endif
};
#if defined(MP3MI_ULP_CENSUS) && !defined(MP3MI_EMU)
#define PART_CENSUS_ARG , size_t cstride
#define PART_CENSUS_PASS , geo.census_cb_stride
#else
#define PART_CENSUS_ARG
#define PART_CENSUS_PASS
#endif

// A closed partition's 64 values (one per lane = record) wait in an LDS tile; every eight partitions the tile
// goes out transposed, 64 contiguous bytes of eb (32 of cb) per record, instead of one 8-byte store per lane
// and partition into 64 different lines.
struct part_tile {
    double eb[8][65];
    float cb[8][65];
    unsigned rec[64]; // the lanes' records (PART_NO_REC: none)
};
#define PART_NO_REC 0xffffffffu

// partitions [b_last & ~7, b_last] of the wavefront's records Lt.rec[0 .. 63] (partition 0 is stored at the end)
MP3MI_DEVFN void part_flush(part_tile &Lt, int b_last, double *__restrict__ eb_all, float *__restrict__ cb_all)
{
    const int lane = wave_lane_here();
    wave_sync();
    const int base = b_last & ~7, np = (b_last & 7) + 1, p = lane & 7;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int r = 8 * i + (lane >> 3);
        const unsigned rr = Lt.rec[r];
        if (p < np && base + p >= 1 && rr != PART_NO_REC) {
            eb_all[(size_t) rr * MP3MI_PART_P + base + p] = Lt.eb[p][r];
            cb_all[(size_t) rr * MP3MI_PART_P + base + p] = Lt.cb[p][r];
        }
    }
    wave_sync(); // the tile is written again
}

// Can the float that d = cb + cw * e rounds to depend on cw's last bits?  cw is k_cw's first tier:
//   * ours against exact arithmetic: the sines and cosines within 2^-51 = 4.4e-16 (dm_sincos_fast), so
//     t1 = r2 c2 - r' cp and t2 within 6.6e-16 t3 (t3 = r2 + |r'| bounds every operand; the products and the
//     difference round too), sqrt(t1^2 + t2^2) within 1.4e-15 t3, the quotient by t3 within 1.6e-15;
//   * the reference's against exact arithmetic, with a libm that is good to one ulp: 1.3e-15 by the same chain;
// so the two differ by less than 2.9e-15; 6e-15 is assumed.  d itself is formed in double twice (product, sum):
// |d - d_ref| < 6e-15 e + 2^-52 cw e + 2^-52 d <= (54 e / d + 4) ulps of d, an ulp of d being >= 2^-53 d.  (float) d
// is safe when d's distance from the nearest midpoint of two floats -- the low 29 bits of its mantissa against
// 0x10000000 -- is larger than that (dm_float_rounding_safe_ulps, with the division multiplied out).
MP3MI_DEVFN bool part_cw_safe(double d, double e)
{
    const long long b = dm_bits(d);
    long long dist = (b & 0x1fffffffLL) - 0x10000000LL;
    dist = dist < 0 ? -dist : dist;
    const bool range = b >= 0x3810000000000000LL && b < 0x47f0000000000000LL; // a normal float (and positive, not nan)
    return range && (double) dist * d > 54.0 * e + 4.0 * d;
}

// CHECK: cw is a first-tier value; *amb is set when a float rounding could depend on its last bits.
template <bool CHECK>
MP3MI_DEVFN void part_line(const mp3mi_tables *T, part_walk &W, part_tile &Lt, int j, float ef, double cw,
                           double *__restrict__ eb_all, float *__restrict__ cb_all, bool *amb PART_CENSUS_ARG)
{
    const double e = (double) ef;
    if (j < T->part_l_covered) {
        W.eb = W.eb + e;
        const double d = (double) W.cb + cw * e;
        // (cw == -0.0: k_cw's mark for a c_w that is exactly zero in the reference as well -- nothing to check)
        if (CHECK && dm_bits(cw) != (long long) 0x8000000000000000ull && !part_cw_safe(d, e)) *amb = true;
#if defined(MP3MI_ULP_CENSUS) && !defined(MP3MI_EMU)
        // site 6: the float this step rounds to, with c_w from correctly rounded sines (lines 0..5, and the records that
        // went through the second tier): could the reference's own sines, one ulp off, round it the other way?
        if (dm_bits(cw) != (long long) 0x8000000000000000ull && cw != 0.4 && e != 0.0 && Lt.rec[wave_lane_here()] != PART_NO_REC) { // (0.4: the constant of lines 206.., src/l3psy.c:555-556 -- no libm in it; e == 0: d == cb whatever c_w is)
            const long long bb = dm_bits(d);
            long long dist = (bb & 0x1fffffffLL) - 0x10000000LL;
            dist = dist < 0 ? -dist : dist;
            const bool range = bb >= 0x3810000000000000LL && bb < 0x47f0000000000000LL;
            // (the band: the reference performs the same operations in the same order, so only its four libm results differ,
            // by at most one ulp each: c_w moves by < 5e-16 -- 1.6e-16 through t1, t2, the root and the quotient, plus a
            // few flipped roundings on the way --, d by < 5e-16 e + an ulp of d = (4.5 e / d + 1) ulps; 4.5 e + 2 d is counted)
            ULP_CENSUS(UC_CW_STEP, !CHECK && range && !((double) dist * d > 4.5 * e + 2.0 * d), !CHECK && range && !((double) dist * d > (4.5 * e + 2.0 * d) * 1048576.0));
            // ... and if it did: the shadow sums take this step's float one ulp down / up (every near step of the partition,
            // both directions at once: the worst case), and the partition's close compares what comes of it
            if (!CHECK) {
                float lo = (float) ((double) W.cb_lo + cw * e), hi = (float) ((double) W.cb_hi + cw * e);
                if (range && !((double) dist * d > 4.5 * e + 2.0 * d)) {
                    lo = __builtin_bit_cast(float, __builtin_bit_cast(int, lo) - (lo > 0.0f ? 1 : 0));
                    hi = __builtin_bit_cast(float, __builtin_bit_cast(int, hi) + 1);
                    W.near_steps++;
                }
                W.cb_lo = lo; W.cb_hi = hi;
            }
            // (sums below the floats' normal range -- a line-0 energy of next to nothing -- are counted apart: the grid there
            // is absolute, 2^-149, and a relative 1e-16 does not reach a midpoint; k_part sends them to the second tier all the same)
            if (!CHECK && !range) ULP_CENSUS(UC_CW_RANGE, 0, 0);
        }
#endif
        W.cb = (float) d;
#if defined(MP3MI_ULP_CENSUS) && !defined(MP3MI_EMU)
        if (!(dm_bits(cw) != (long long) 0x8000000000000000ull && cw != 0.4 && e != 0.0 && Lt.rec[wave_lane_here()] != PART_NO_REC) || CHECK) { // steps outside the census: the shadows follow the sum
            W.cb_lo = (float) ((double) W.cb_lo + cw * e);
            W.cb_hi = (float) ((double) W.cb_hi + cw * e);
        }
        if (j + 1 == W.pend) { // the partition closes: did a one-ulp difference at its near steps reach the sum that goes on?
            if (!CHECK && W.near_steps > 0 && Lt.rec[wave_lane_here()] != PART_NO_REC) ULP_CENSUS(UC_CW_REACH, W.cb_lo != W.cb || W.cb_hi != W.cb, 0);
            // ... and the shadow sums go on to k_psy (census site UC_CW_NB: do they reach a threshold?).  Partition 0 also takes
            // the lines beyond the table's coverage (a constant c_w: the same addend for all three): its shadows keep the
            // DIFFERENCE to the sum as it stands here, k_psy adds it to the final value
            const unsigned rr = Lt.rec[wave_lane_here()];
            if (cstride && rr != PART_NO_REC && W.b < MP3MI_CBANDS) {
                cb_all[cstride + (size_t) rr * MP3MI_PART_P + W.b] = W.cb_lo; // (a first-tier step moves the shadows like the sum)
                cb_all[2 * cstride + (size_t) rr * MP3MI_PART_P + W.b] = W.cb_hi;
                if (W.b == 0) cb_all[(size_t) rr * MP3MI_PART_P + MP3MI_CBANDS] = W.cb; // (the row's one free slot: partition 0's sum at its close)
            }
            W.cb_lo = W.cb_hi = 0.0f;
            W.near_steps = 0;
        }
#endif
        while (W.b < MP3MI_CBANDS && j + 1 == W.pend) { // closes this partition and any empty ones after it
            if (W.b == 0) { W.eb0 = W.eb; W.cb0 = W.cb; }
            else {
                const int lane = wave_lane_here();
                Lt.eb[W.b & 7][lane] = W.eb;
                Lt.cb[W.b & 7][lane] = W.cb;
            }
            if ((W.b & 7) == 7) part_flush(Lt, W.b, eb_all, cb_all);
            W.eb = 0.0; W.cb = 0.0f;
            W.b++;
            W.pend = W.b < MP3MI_CBANDS ? T->part_l_start[W.b + 1] : -1;
        }
    } else { // lines beyond the table's coverage fall into partition 0 (src/l3psy.c:572-577 with partition_l == 0)
        W.eb0 = W.eb0 + e;
        W.cb0 = (float) ((double) W.cb0 + cw * e);
    }
}

__global__ void __launch_bounds__(64) k_part(const mp3mi_tables *__restrict__ T, mp3mi_geom geo,
                                             const float *__restrict__ energy_l, const double *__restrict__ cw_mid,
                                             const float *__restrict__ hist6, const mp3mi_psy_state *__restrict__ state,
                                             double *__restrict__ eb_all, float *__restrict__ cb_all,
                                             mp3mi_cw_fixlist *__restrict__ fix, int second)
{
#if !defined(MP3MI_EMU)
    // beside k_loop (batch.cpp) this kernel gets what that one leaves: with the highest wave priority it is through in its
    // stand-alone time instead of three times that, and the chain k_cw -> k_part -> k_psy -> k_filter -> k_mdct ends before
    // the k_loop launch it runs beside does (profiles/r04_experiments.txt, prioA)
    __builtin_amdgcn_s_setprio(3);
#endif

    // second == 0: the lanes are the 64 consecutive records of block blockIdx.x.  With fix given, cw_mid holds k_cw's
    // FIRST tier: every rounding it feeds is checked (part_cw_safe) and a record with one that could go either way is
    // put on the list.  second == 1: the lanes are the records of the list, on the second-tier values k_cw_fix has
    // written for them meanwhile.  fix == NULL (MP3MI_CW_EXACT=1): second-tier values throughout, nothing to check.
    const bool check = !second && fix != NULL;
    bool amb = false;
    const int lane = wave_lane();
    const int C = geo.channels, G = geo.n_gran;
    const size_t n_rec = (size_t) geo.n_streams * (size_t) G * (size_t) C;
    bool live;
    size_t rec;
    if (second) {
        const unsigned n_fix = fix->count < fix->cap ? fix->count : fix->cap;
        if ((size_t) blockIdx.x * 64 >= n_fix) return;
        live = (size_t) blockIdx.x * 64 + lane < n_fix;
        rec = live ? fix->list[(size_t) blockIdx.x * 64 + lane] : n_rec - 1;
    } else {
        live = (size_t) blockIdx.x * 64 + lane < n_rec;
        rec = live ? (size_t) blockIdx.x * 64 + lane : n_rec - 1;
    }
    __shared__ part_tile Lt;
    Lt.rec[lane] = live ? (unsigned) rec : PART_NO_REC;
    const int ch = (int) (rec % C), gl = (int) ((rec / C) % G);
    const size_t s = rec / ((size_t) C * G);
    const mp3mi_psy_state *st = &state[s * C + ch];
    const float *er = energy_l + rec * MP3MI_HBLK_P;
    const double *cwr = cw_mid + rec * 50;

    part_walk W;
    W.eb = W.eb0 = 0.0; W.cb = W.cb0 = 0.0f;
#if defined(MP3MI_ULP_CENSUS) && !defined(MP3MI_EMU)
    W.cb_lo = W.cb_hi = 0.0f;
    W.near_steps = 0;
#endif
    W.b = 0; W.pend = T->part_l_start[1];
    while (W.b < MP3MI_CBANDS && W.pend == 0) { W.b++; W.pend = W.b < MP3MI_CBANDS ? T->part_l_start[W.b + 1] : -1; } // (never: partition 0 has lines)

    // lines 0..5: unpredictability from this granule's r/phi and the two granules before it (src/l3psy.c:496-512)
#pragma unroll 1
    for (int j = 0; j < 6; j++) {
        const float rn = hist6[rec * 12 + j], pn = hist6[rec * 12 + 6 + j];
        const float r1 = gl >= 1 ? hist6[(rec - C) * 12 + j] : st->r1[j], p1 = gl >= 1 ? hist6[(rec - C) * 12 + 6 + j] : st->p1[j];
        const float r2 = gl >= 2 ? hist6[(rec - 2 * C) * 12 + j] : (gl == 1 ? st->r1[j] : st->r2[j]);
        const float p2 = gl >= 2 ? hist6[(rec - 2 * C) * 12 + 6 + j] : (gl == 1 ? st->p1[j] : st->p2[j]);
        const double r_prime = 2.0 * (double) r1 - (double) r2;
        const double phi_prime = 2.0 * (double) p1 - (double) p2;
        double sn, cn, sp, cp;
        dm_sincos((double) pn, &sn, &cn);
        dm_sincos(phi_prime, &sp, &cp);
        const double t1 = (double) rn * cn - r_prime * cp;
        const double t2 = (double) rn * sn - r_prime * sp;
        const double t3 = (double) rn + __builtin_fabs(r_prime);
        double cw = (t3 != 0.0) ? __builtin_sqrt(t1 * t1 + t2 * t2) / t3 : 0.0;
        if ((double) rn == r_prime && (double) pn == phi_prime) cw = -0.0; // an exact zero in the reference too (k_fft.hip, cw_record)
        part_line<false>(T, W, Lt, j, er[j], cw, eb_all, cb_all, &amb PART_CENSUS_PASS); // (correctly rounded sines: nothing to check)
    }
    // lines 6..511 in blocks of 32 = one 128-byte line of the energy row; the unpredictability of lines
    // 6+4n..9+4n is cw_mid[n] (src/l3psy.c:531-549), from line 206 on the constant 0.4 (src/l3psy.c:555-556).
    // Line l of block k uses cw_mid[8k - 2 + ((l - 6) >> 2) + 2]: nine values per block.
#pragma unroll 1
    for (int k = 0; k < 16; k++) {
        psy_f4 blk[8];
#pragma unroll
        for (int q = 0; q < 8; q++) blk[q] = *(const psy_f4 *) (er + 32 * k + 4 * q);
        double cwv[9];
        if (32 * k < 206) {
#pragma unroll
            for (int q = 0; q < 9; q++) {
                const int m = 8 * k - 2 + q;
                cwv[q] = (m >= 0 && m < 50) ? cwr[m] : 0.4;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 9; q++) cwv[q] = 0.4;
        }
        const float ev[32] = {blk[0].x, blk[0].y, blk[0].z, blk[0].w, blk[1].x, blk[1].y, blk[1].z, blk[1].w,
                              blk[2].x, blk[2].y, blk[2].z, blk[2].w, blk[3].x, blk[3].y, blk[3].z, blk[3].w,
                              blk[4].x, blk[4].y, blk[4].z, blk[4].w, blk[5].x, blk[5].y, blk[5].z, blk[5].w,
                              blk[6].x, blk[6].y, blk[6].z, blk[6].w, blk[7].x, blk[7].y, blk[7].z, blk[7].w};
#pragma unroll
        for (int l = 0; l < 32; l++) {
            const int jj = 32 * k + l;
            if (jj < 6) continue; // done above (only in block 0; wave-uniform)
            const int q = ((l - 6) >> 2) + 2; // arithmetic shift: l < 6 never reaches here with k == 0
            if (jj < 206 && check) part_line<true>(T, W, Lt, jj, ev[l], cwv[q], eb_all, cb_all, &amb PART_CENSUS_PASS);
            else part_line<false>(T, W, Lt, jj, ev[l], jj < 206 ? cwv[q] : 0.4, eb_all, cb_all, &amb PART_CENSUS_PASS);
        }
    }
    part_line<false>(T, W, Lt, 512, er[512], 0.4, eb_all, cb_all, &amb PART_CENSUS_PASS);
    for (int b = W.b < 1 ? 1 : W.b; b < MP3MI_CBANDS; b++) { // partitions without lines
        Lt.eb[b & 7][lane] = 0.0;
        Lt.cb[b & 7][lane] = 0.0f;
        if ((b & 7) == 7) part_flush(Lt, b, eb_all, cb_all);
    }
    if (((MP3MI_CBANDS - 1) & 7) != 7) part_flush(Lt, MP3MI_CBANDS - 1, eb_all, cb_all); // the last, partial group
    if (live) {
        eb_all[rec * MP3MI_PART_P] = W.eb0;
        cb_all[rec * MP3MI_PART_P] = W.cb0;
    }
    if (check && amb && live) { // (rare: a list entry per record, in no particular order)
        const unsigned i = atomicAdd(&fix->count, 1u);
        if (i < fix->cap) fix->list[i] = (unsigned) rec;
    }
}

// k_part for a HANDFUL of records (the drop-in L3psycho_anal: one or two per launch): a wavefront per record, a LANE PER
// PARTITION -- the partitions' sums are independent chains of at most a few dozen lines, where k_part's lane walks all
// 513 lines of its record (64 us per launch when the wavefront has two records to show for it).  The same sums in the
// same order: eb in double, cb rounded to float at every step (src/l3psy.c:565-578); lines beyond the table's coverage
// go on top of partition 0's (lane 0 walks on).  cw_mid must hold SECOND-tier values (k_cw with MP3MI_TEST_CW_EXACT):
// nothing is checked here, nothing is listed.
__global__ void __launch_bounds__(64) k_part_wave(const mp3mi_tables *__restrict__ T, mp3mi_geom geo, const float *__restrict__ energy_l,
                                                  const double *__restrict__ cw_mid, const float *__restrict__ hist6,
                                                  const mp3mi_psy_state *__restrict__ state, double *__restrict__ eb_all, float *__restrict__ cb_all)
{
    const int lane = wave_lane();
    const int C = geo.channels, G = geo.n_gran;
    const size_t rec = blockIdx.x;
    const int ch = (int) (rec % C), gl = (int) ((rec / C) % G);
    const size_t s = rec / ((size_t) C * G);
    const mp3mi_psy_state *st = &state[s * C + ch];
    const float *er = energy_l + rec * MP3MI_HBLK_P;
    const double *cwr = cw_mid + rec * 50;
    if (lane >= MP3MI_CBANDS) return;
    int j0 = T->part_l_start[lane], j1 = T->part_l_start[lane + 1];
    if (j1 > T->part_l_covered) j1 = T->part_l_covered;
    double eb = 0.0;
    float cb = 0.0f;
    for (int pass = 0; pass < 2; pass++) { // (lane 0: its own lines, then the uncovered ones)
        for (int j = j0; j < j1; j++) {
            double cw;
            if (j < 6) { // unpredictability from this granule's r / phi and the two granules before it (src/l3psy.c:496-512): as k_part
                const float rn = hist6[rec * 12 + j], pn = hist6[rec * 12 + 6 + j];
                const float r1 = gl >= 1 ? hist6[(rec - C) * 12 + j] : st->r1[j], p1 = gl >= 1 ? hist6[(rec - C) * 12 + 6 + j] : st->p1[j];
                const float r2 = gl >= 2 ? hist6[(rec - 2 * C) * 12 + j] : (gl == 1 ? st->r1[j] : st->r2[j]);
                const float p2 = gl >= 2 ? hist6[(rec - 2 * C) * 12 + 6 + j] : (gl == 1 ? st->p1[j] : st->p2[j]);
                const double r_prime = 2.0 * (double) r1 - (double) r2;
                const double phi_prime = 2.0 * (double) p1 - (double) p2;
                double sn, cn, sp, cp;
                dm_sincos((double) pn, &sn, &cn);
                dm_sincos(phi_prime, &sp, &cp);
                const double t1 = (double) rn * cn - r_prime * cp;
                const double t2 = (double) rn * sn - r_prime * sp;
                const double t3 = (double) rn + __builtin_fabs(r_prime);
                cw = (t3 != 0.0) ? __builtin_sqrt(t1 * t1 + t2 * t2) / t3 : 0.0;
                if ((double) rn == r_prime && (double) pn == phi_prime) cw = -0.0;
            } else
                cw = j < 206 ? cwr[(j - 6) >> 2] : 0.4;
            const double e = (double) er[j];
            eb = eb + e;
            cb = (float) ((double) cb + cw * e);
        }
        if (lane != 0) break;
        j0 = T->part_l_covered; j1 = MP3MI_HBLK;
    }
    eb_all[rec * MP3MI_PART_P + lane] = eb;
    cb_all[rec * MP3MI_PART_P + lane] = cb;
}

void mp3mi_launch_part_wave(const mp3mi_tables *T, const mp3mi_geom &g, const float *energy_l, const double *cw_mid, const float *hist6,
                            const void *psy_state, double *eb_all, float *cb_all, hipStream_t st)
{
    const size_t n_rec = (size_t) g.n_streams * (size_t) g.n_gran * (size_t) g.channels;
    hipLaunchKernelGGL(k_part_wave, dim3((unsigned) n_rec), dim3(64), 0, st, T, g, energy_l, cw_mid, hist6, (const mp3mi_psy_state *) psy_state, eb_all, cb_all);
}

// The kernel is a chain of dependent steps per granule (k_part's sums in, spreading, threshold, entropy, ratios out),
// so what it costs is latency: the next granule's partition sums are requested before the current one is worked
// on, and the spreading function -- a row of s3_l per lane -- never comes from memory inside the loop: at 44.1 kHz
// (SPARSE) a row has at most PSY_S3_W non-zero entries, and the PSY_W wavefronts of a workgroup share one copy of
// these runs in LDS ([entry][lane]: conflict-free); at the other rates the rows are dense and are read column by column
// from the transposed table (one coalesced line per step).  Every wavefront works on its own (stream, channel) with
// its own psy_lds: apart from the table load, synchronisation is per wavefront (wave_sync).
#define PSY_S3_W 17
#define PSY_W 4
template <bool SPARSE>
__global__ void __launch_bounds__(64 * PSY_W, 3) k_psy(const mp3mi_tables *__restrict__ T, mp3mi_geom geo,
                                            const double *__restrict__ eb_all, const float *__restrict__ cb_all,
                                            const float *__restrict__ energy_s, const float *__restrict__ hist6,
                                            mp3mi_psy_state *__restrict__ state, mp3mi_psy_out *__restrict__ out)
{
#if !defined(MP3MI_EMU)
    // beside k_loop (batch.cpp) this kernel gets what that one leaves: with the highest wave priority it is through in its
    // stand-alone time instead of three times that, and the chain k_cw -> k_part -> k_psy -> k_filter -> k_mdct ends before
    // the k_loop launch it runs beside does (profiles/r04_experiments.txt, prioA)
    __builtin_amdgcn_s_setprio(3);
#endif

    __shared__ psy_lds LL[PSY_W];
    __shared__ double s3rows[SPARSE ? PSY_S3_W : 1][64];
    const int wv = __builtin_amdgcn_readfirstlane((int) threadIdx.x >> 6);
    psy_lds &L = LL[wv];
    const bool psy_exact = (geo.test_flags & 4) != 0; // MP3MI_PSY_EXACT=1: thresholds from dm_log / dm_exp only (tests)
    const int lane = wave_lane();
    const int C = geo.channels, G = geo.n_gran;
    // (the last workgroup may hold wavefronts without a task: they leave after the workgroup's one barrier)
    const int n_task = geo.n_streams * C;
    int task = (int) blockIdx.x * PSY_W + wv;
    const bool live = task < n_task;
    task = live ? task : n_task - 1;
    const int ch = task % C, s = task / C;
    mp3mi_psy_state *st = &state[(size_t) s * C + ch];
    const int b = lane; // partition owned by this lane (lane 63 idles in partition loops)

    // restore carried state
    float r1 = 0, p1 = 0, r2 = 0, p2 = 0;
    double nb_1 = 0, nb_2 = 0;
    if (lane < 6) { r1 = st->r1[lane]; p1 = st->p1[lane]; r2 = st->r2[lane]; p2 = st->p2[lane]; }
    if (b < MP3MI_CBANDS) { nb_1 = st->nb_1[b]; nb_2 = st->nb_2[b]; }
    if (lane < 21) L.held_l[lane] = st->ratio_l[lane];
    if (lane < 36) L.held_s[lane] = st->ratio_s[lane];
    int bt_old = st->blocktype_old;
    // per-lane constants
    double minval = 0, qthr_l = 0, norm_l = 0, numl = 0;
    int pl0 = 0, pl1 = 0, s3lo = 0, s3hi = -1;
    if (b < MP3MI_CBANDS) {
        minval = T->minval[b]; qthr_l = T->qthr_l[b]; norm_l = T->norm_l[b];
        numl = (double) T->numlines_pe[b];
        pl0 = T->part_l_start[b]; pl1 = T->part_l_start[b + 1];
        if (SPARSE) { s3lo = T->s3_lo[b]; s3hi = T->s3_hi[b]; } else { s3lo = 0; s3hi = MP3MI_CBANDS - 1; }
    }
    if (SPARSE && wv == 0) { // lane b's run of the spreading function, entries s3lo .. s3lo + PSY_S3_W - 1 (0 past its end)
#pragma unroll
        for (int i = 0; i < PSY_S3_W; i++) s3rows[i][lane] = (b < MP3MI_CBANDS && s3lo + i <= s3hi) ? T->s3_l[b][s3lo + i] : 0.0;
    }
    // partition sums of the first granule (every later one is requested one granule ahead)
    double eb_next = 0.0;
    float cb_next = 0.0f;
    if (b < MP3MI_CBANDS && G > 0) {
        const size_t rec0 = ((size_t) s * G) * C + ch;
        eb_next = eb_all[rec0 * MP3MI_PART_P + b];
        cb_next = cb_all[rec0 * MP3MI_PART_P + b];
    }
    __syncthreads();
    if (!live) return;

    for (int gl = 0; gl < G; gl++) {
        const size_t rec = ((size_t) s * G + gl) * C + ch;
        // partition energy and weighted unpredictability come from k_part; r/phi of lines 0..5 are only
        // carried along so that the state handed to the next chunk holds the last two granules'
        if (lane < 6) {
            const float rn = hist6[rec * 12 + lane], pn = hist6[rec * 12 + 6 + lane];
            r2 = r1; p2 = p1; r1 = rn; p1 = pn;
        }
#if defined(MP3MI_ULP_CENSUS) && !defined(MP3MI_EMU)
        if (b < MP3MI_CBANDS) { // the shadows of this granule's sums (k_part wrote them on the second-tier run only for listed records: the others carry the sum itself)
            const size_t cs = geo.census_cb_stride;
            float lo = cb_all[cs + rec * MP3MI_PART_P + b], hi = cb_all[2 * cs + rec * MP3MI_PART_P + b];
            if (b == 0) { // partition 0: the shadows stopped at its close, the uncovered lines came on top (see k_part)
                const float at_close = cb_all[rec * MP3MI_PART_P + MP3MI_CBANDS], fin = cb_all[rec * MP3MI_PART_P];
                lo = fin + (lo - at_close); hi = fin + (hi - at_close);
            }
            L.cb_lo[b] = cs ? lo : cb_all[rec * MP3MI_PART_P + b];
            L.cb_hi[b] = cs ? hi : cb_all[rec * MP3MI_PART_P + b];
        }
#endif
        if (b < MP3MI_CBANDS) {
            L.eb[b] = eb_next;
            L.cb[b] = cb_next;
            if (gl + 1 < G) { // in flight while this granule is worked on
                eb_next = eb_all[(rec + C) * MP3MI_PART_P + b];
                cb_next = cb_all[(rec + C) * MP3MI_PART_P + b];
            }
        }
        wave_sync();

        double thr = 0.0, ebv = 0.0;
        // spreading (src/l3psy.c:586-605, 1062-1084)
        float ecb = 0.0f;
        double ctb = 0.0;
#if defined(MP3MI_ULP_CENSUS) && !defined(MP3MI_EMU)
        double ctb_lo = 0.0, ctb_hi = 0.0;
#define PSY_CENSUS_SPREAD(sv, k) { ctb_lo = ctb_lo + (sv) * (double) L.cb_lo[k]; ctb_hi = ctb_hi + (sv) * (double) L.cb_hi[k]; }
#else
#define PSY_CENSUS_SPREAD(sv, k)
#endif
        if (SPARSE) { // every lane takes PSY_S3_W steps; a step past the row's end adds nothing (table build checks the width)
#pragma unroll
            for (int i = 0; i < PSY_S3_W; i++) {
                const int k = s3lo + i;
                if (b < MP3MI_CBANDS && k <= s3hi) {
                    const double sv = s3rows[i][lane];
                    ecb = (float) ((double) ecb + sv * L.eb[k]);
                    ctb = ctb + sv * (double) L.cb[k];
                    PSY_CENSUS_SPREAD(sv, k)
                }
            }
        } else if (b < MP3MI_CBANDS) {
#pragma unroll 1
            for (int k0 = 0; k0 < MP3MI_CBANDS; k0 += 9) { // 63 = 7 x 9: nine loads in flight, then the ordered sums
                double svv[9];
#pragma unroll
                for (int u = 0; u < 9; u++) svv[u] = T->s3_lt[k0 + u][b];
#pragma unroll
                for (int u = 0; u < 9; u++) {
                    if (svv[u] != 1.0) { // src/l3psy.c:596-603: entries that are exactly 1 are skipped at these rates
                        ecb = (float) ((double) ecb + svv[u] * L.eb[k0 + u]);
                        ctb = ctb + svv[u] * (double) L.cb[k0 + u];
                        PSY_CENSUS_SPREAD(svv[u], k0 + u)
                    }
                }
            }
        }
        // tonality, SNR, threshold (src/l3psy.c:610-636).  Only nb -- a FLOAT -- leaves this block, so the
        // log and the exp are first taken in plain double: log off by < 2^-50 max(1,|log|) (|log| < 4.7)
        // moves tbb by < 2^-48.9, snr by < 2^-44.3, the exponent by < 2^-46.4, and the product with
        // exp (itself < 2^-50) by < 2^-45.9 relative; unless it lies within 2^-44 (256 ulps) of the midpoint
        // of two floats, nb is decided.  Otherwise the wavefront repeats the block with dm_log / dm_exp.
        float nb = 0.0f;
        for (int tier = psy_exact ? 1 : 0; tier < 2; tier++) {
            double pr = 0.0;
            if (b < MP3MI_CBANDS) {
                double cbb, tbb, snr;
                if ((double) ecb != 0.0) {
                    cbb = ctb / (double) ecb;
                    if (cbb < 0.01) cbb = 0.01;
                    cbb = tier ? dm_log(cbb) : dm_log_fast(cbb);
                } else
                    cbb = 0.0;
                tbb = -0.299 - 0.43 * cbb;
                tbb = (0.0 > tbb) ? 0.0 : tbb;
                tbb = (1.0 < tbb) ? 1.0 : tbb;
                snr = 29.0 * tbb + 6.0 * (1.0 - tbb);
                snr = (minval > snr) ? minval : snr;
                const double arg = -snr * R_LN_TO_LOG10;
                pr = ((double) ecb * norm_l) * (tier ? dm_exp(arg) : dm_exp_fast(arg));
                nb = (float) pr;
            }
            if (tier == 0 && !wave_any(pr != 0.0 && !dm_float_rounding_safe_ulps(pr, 256))) break;
        }
#if defined(MP3MI_ULP_CENSUS) && !defined(MP3MI_EMU)
        if (b < MP3MI_CBANDS && (double) ecb != 0.0) { // site 2: nb = (float)(ecb norm exp(-snr ln10/10)), log and exp one ulp off: 20 ulps of the product
            double cbb = ctb / (double) ecb;
            if (cbb < 0.01) cbb = 0.01;
            cbb = dm_log(cbb);
            double tbb = -0.299 - 0.43 * cbb;
            tbb = (0.0 > tbb) ? 0.0 : tbb;
            tbb = (1.0 < tbb) ? 1.0 : tbb;
            double snr = 29.0 * tbb + 6.0 * (1.0 - tbb);
            snr = (minval > snr) ? minval : snr;
            const double prx = ((double) ecb * norm_l) * dm_exp(-snr * R_LN_TO_LOG10);
            const long long dd = dm_float_midpoint_distance_ulps(prx);
            ULP_CENSUS(UC_NB, dd <= 20, dd <= (20LL << 20));
            // site UC_CW_NB: this partition's threshold with k_part's shadow sums in the spreading -- every float of a near
            // step of the c_w site one ulp lower, and one ulp higher -- : calls = thresholds a shadow reaches at all (the
            // spread unpredictability differs), near = thresholds that come out as another float
            if (ctb_lo != ctb || ctb_hi != ctb) {
                bool differs = false;
                for (int w = 0; w < 2; w++) {
                    double c2 = (w ? ctb_hi : ctb_lo) / (double) ecb;
                    if (c2 < 0.01) c2 = 0.01;
                    c2 = dm_log(c2);
                    double t2 = -0.299 - 0.43 * c2;
                    t2 = (0.0 > t2) ? 0.0 : t2;
                    t2 = (1.0 < t2) ? 1.0 : t2;
                    double s2 = 29.0 * t2 + 6.0 * (1.0 - t2);
                    s2 = (minval > s2) ? minval : s2;
                    differs |= (float) (((double) ecb * norm_l) * dm_exp(-s2 * R_LN_TO_LOG10)) != (float) prx;
                }
                ULP_CENSUS(UC_CW_NB, differs, 0);
            }
        }
#endif
        if (b < MP3MI_CBANDS) {
            const double a2 = 2.0 * nb_1, a16 = 16.0 * nb_2;
            const double inner = (a2 < a16) ? a2 : a16;
            const double t1 = ((double) nb < inner) ? (double) nb : inner;
            thr = (qthr_l > t1) ? qthr_l : t1;
            nb_2 = nb_1;
            nb_1 = (double) nb;
            ebv = L.eb[b];
            L.thr[b] = thr;
            // perceptual entropy term (src/l3psy.c:639-645)
            const double lg = dm_log((thr + 1.0) / (ebv + 1.0));
            L.pev[b] = numl * ((0.0 < lg) ? 0.0 : lg);
        }
        wave_sync();
        if (lane == 0) {
            double pe = 0.0;
            for (int k = 0; k < MP3MI_CBANDS; k++) pe = pe - L.pev[k];
            L.pe = pe;
        }
        wave_sync();
        const double pe = L.pe;
        const bool attack = !(pe < 1800.0);
#if defined(MP3MI_ULP_CENSUS) && !defined(MP3MI_EMU)
        // site 3a: pe is a sum of 63 terms numlines * log(..); every log one ulp off moves it by < 513 * 2^-48 = 1.8e-12
        if (lane == 0) ULP_CENSUS(UC_PE_ATTACK, __builtin_fabs(pe - 1800.0) <= 2e-12, __builtin_fabs(pe - 1800.0) <= 2e-12 * 1048576.0);
#endif

        // block type state machine (src/l3psy.c:651-668, 689-694, 732-739)
        int blocktype, bt_out;
        if (!attack) {
            blocktype = (bt_old == 2) ? 3 : 0;
        } else {
            blocktype = 2;
            if (bt_old == 0) bt_old = 1;
            if (bt_old == 3) bt_old = 2;
        }
        bt_out = bt_old;
        bt_old = blocktype;

        // outputs: ratios are the values held BEFORE this granule updates them (src/l3psy.c:452-456)
        if (lane < 21) out[rec].ratio_l[lane] = L.held_l[lane];
        if (lane < 36) out[rec].ratio_s[lane / 3][lane % 3] = L.held_s[lane];
        if (lane == 0) { out[rec].pe = pe; out[rec].block_type = bt_out; out[rec].pad = 0; }
        wave_sync();

        if (!attack) { // long-block ratios (src/l3psy.c:671-684)
            if (lane < 21) {
                const int bu = T->bu_l[lane], bo = T->bo_l[lane];
                const double w1 = T->w1_l[lane], w2 = T->w2_l[lane];
                double en = w1 * L.eb[bu] + w2 * L.eb[bo];
                double thm = w1 * L.thr[bu] + w2 * L.thr[bo];
                for (int k = bu + 1; k < bo; k++) { en = en + L.eb[k]; thm = thm + L.thr[k]; }
                L.held_l[lane] = (en != 0.0) ? thm / en : 0.0;
            }
        } else { // short-block ratios for the three windows (src/l3psy.c:696-729)
            for (int sblock = 0; sblock < 3; sblock++) {
                if (b < MP3MI_CBANDS_S) {
                    const float *es = energy_s + rec * (3 * MP3MI_HBLK_S) + (size_t) sblock * MP3MI_HBLK_S;
                    double e = 0.0;
                    for (int j = T->part_s_start[b]; j < T->part_s_start[b + 1]; j++) e = e + (double) es[j];
                    if (b == 0)
                        for (int j = T->part_s_covered; j < MP3MI_HBLK_S; j++) e = e + (double) es[j];
                    L.ebs[b] = e;
                }
                wave_sync();
                if (b < MP3MI_CBANDS_S) {
                    float ecb = 0.0f;
                    for (int k = 0; k < MP3MI_CBANDS_S; k++) ecb = (float) ((double) ecb + T->s3_l[b][k] * L.ebs[k]);
                    const float nb = (float) (((double) ecb * T->norm_l[b]) * T->exp_snr_s[b]);
                    const double q = T->qthr_s[b];
                    L.thrs[b] = (q > (double) nb) ? q : (double) nb;
                }
                wave_sync();
                if (lane < 12) {
                    const int bu = T->bu_s[lane], bo = T->bo_s[lane];
                    const double w1 = T->w1_s[lane], w2 = T->w2_s[lane];
                    double en = w1 * L.ebs[bu] + w2 * L.ebs[bo];
                    double thm = w1 * L.thrs[bu] + w2 * L.thrs[bo];
                    for (int k = bu + 1; k < bo; k++) { en = en + L.ebs[k]; thm = thm + L.thrs[k]; }
                    L.held_s[lane * 3 + sblock] = (en != 0.0) ? thm / en : 0.0;
                }
                wave_sync();
            }
        }
        wave_sync();
    }

    if (lane < 6) { st->r1[lane] = r1; st->p1[lane] = p1; st->r2[lane] = r2; st->p2[lane] = p2; }
    if (b < MP3MI_CBANDS) { st->nb_1[b] = nb_1; st->nb_2[b] = nb_2; }
    if (lane < 21) st->ratio_l[lane] = L.held_l[lane];
    if (lane < 36) st->ratio_s[lane] = L.held_s[lane];
    if (lane == 0) st->blocktype_old = bt_old;
}

size_t mp3mi_psy_state_size(void) { return sizeof(mp3mi_psy_state); }

// bins, fix: for the second tier of the unpredictability -- k_cw_fix and k_part again on the records the first k_part
// lists (fix holds at least as many entries as there are records); with MP3MI_CW_EXACT=1 (test_flags bit 4) k_cw has
// already written second-tier values everywhere and nothing is checked.
void mp3mi_launch_psy(const mp3mi_tables *T, const mp3mi_geom &g, const float *energy_l, const float *energy_s,
                      double *cw_mid, float *hist6, const float *bins, mp3mi_cw_fixlist *fix, void *psy_state, double *eb_all, float *cb_all,
                      mp3mi_psy_out *out, hipStream_t st, int which)
{
    // which: bit 0 the partition sums (k_part, with the second tier of the unpredictability), bit 1 k_psy
    const size_t n_rec = (size_t) g.n_streams * (size_t) g.n_gran * (size_t) g.channels;
    const unsigned nblk = (unsigned) ((n_rec + 63) / 64);
    if (!(which & 1)) {
    } else if ((g.test_flags >> 4) & 1)
        hipLaunchKernelGGL(k_part, dim3(nblk), dim3(64), 0, st, T, g, energy_l, cw_mid, hist6, (const mp3mi_psy_state *) psy_state, eb_all, cb_all,
                           (mp3mi_cw_fixlist *) NULL, 0);
    else {
        mp3mi_launch_cw_fix_reset(fix, (unsigned) n_rec, st);
        hipLaunchKernelGGL(k_part, dim3(nblk), dim3(64), 0, st, T, g, energy_l, cw_mid, hist6, (const mp3mi_psy_state *) psy_state, eb_all, cb_all, fix, 0);
#if defined(MP3MI_EXP_PART_TWICE) // experiment (profiles/r06_experiments.txt, F5): what does the step pay for k_part's 28 GB of reads?  The same launch once more.
        mp3mi_launch_cw_fix_reset(fix, (unsigned) n_rec, st);
        hipLaunchKernelGGL(k_part, dim3(nblk), dim3(64), 0, st, T, g, energy_l, cw_mid, hist6, (const mp3mi_psy_state *) psy_state, eb_all, cb_all, fix, 0);
#endif
        mp3mi_launch_cw_fix(g, bins, cw_mid, hist6, fix, st);
        // (as many blocks as the list could need: those beyond its end leave at once)
        hipLaunchKernelGGL(k_part, dim3(nblk), dim3(64), 0, st, T, g, energy_l, cw_mid, hist6, (const mp3mi_psy_state *) psy_state, eb_all, cb_all, fix, 1);
    }
    const unsigned grid = (unsigned) ((g.n_streams * g.channels + PSY_W - 1) / PSY_W);
    if (!(which & 2)) return;
    if (g.rate_idx == 0)
        hipLaunchKernelGGL(k_psy<true>, dim3(grid), dim3(64 * PSY_W), 0, st, T, g, eb_all, cb_all, energy_s, hist6, (mp3mi_psy_state *) psy_state, out);
    else
        hipLaunchKernelGGL(k_psy<false>, dim3(grid), dim3(64 * PSY_W), 0, st, T, g, eb_all, cb_all, energy_s, hist6, (mp3mi_psy_state *) psy_state, out);
}

ULP_CENSUS_ACCESSOR(mp3mi_debug_ulp_census_psy)
