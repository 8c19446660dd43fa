// Stateless head of the iteration loop for a (stream, granule, channel) record: calc_xmin (src/loop.c:1085-1118),
// the integer log-energies that calc_scfsi stores (src/loop.c:631-667) and quantanf_init (src/loop.c:369-402).
// None of this depends on the bit reservoir, so it is lifted out of the serial kernel (k_loop), which starts each
// granule from the mp3mi_loop_prep record.
//
// Until round 3 this kernel computed every record, reading the whole spectrum a second time (34 GB per 4096 x 383
// step).  Now k_mdct's tail computes the records while the granule's spectrum is still in its LDS (k_fbmdct.hip:
// the band energies in the reference's order, everything that only feeds an integer rounding in any order, with a
// margin) and LISTS the records it could not decide; this kernel works through the list the reference's way -- and
// through every record when there is no list: the drop-in iteration_loop (dropin.cpp), whose spectrum comes from the
// caller, and MP3MI_TEST_PREP_EXACT (tests compare the two paths).
//
// Everything here is an ORDER-SENSITIVE f64 sum over the 576 lines of a granule (the total
// energy, the band energies, the sum of logs), i.e. a serial chain per granule.  So the lanes
// of a wavefront are 64 different granules, each walking its own 576 lines in index order,
// one whole 128-byte line (16 values) at a time.
//
// quantanf_init needs sum(log(xr^2)) only to round v = 8*ln(sfm) to an integer.  The first tier takes no logarithm
// per line at all: sum log(t_i) = ln2 * sum e_i + log(prod m_i) with t_i = m_i 2^e_i, m_i in [1, 2) -- an integer
// add and one multiplication per line (576 mantissas cannot overflow: the product stays below 2^576), one logarithm
// per granule.  How far that is from the reference's value S_ref -- the sequential f64 sum of 576 rounded logs:
//   * S_ref against exact arithmetic: each log within 1 ulp (|log t| < 1420: 2.3e-13), each of the 575 additions
//     within half an ulp of a partial sum below 576 * 1420 < 2^20 (1.2e-10): < 6.8e-8 in all (5e-10 on audio,
//     whose logs stay within +-25);
//   * ours against exact arithmetic: 575 multiplications (6.4e-14 relative on the product = absolute on its
//     log; in any order), dm_log_fast (2^-50), the product with ln2 (< 2^20 * 2^-53) and one addition: < 3e-10;
// so S moves by < 6.9e-8 and v = 8 (S / 576 - ln(tot / 576)) by < 9.6e-10: unless v lies within 2e-9 max(1, |v|)
// of a rounding boundary of nint() the integer is decided.  Otherwise (probability ~1e-7 per granule) the
// wavefront repeats the walk the reference's way, with the correctly rounded dm_log per line.
// MP3MI_PREP_EXACT=1 forces the second tier (tests run both).
#include "mp3mi_host.h"
#include "dmath.h"

#if defined(MP3MI_EMU)
#define PREP_SCHED_FENCE()
#else
#define PREP_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
#define PREP_BLOCK 16 /* lines per step: one 128-byte line of the spectrum */

struct __attribute__((aligned(16))) prep_d2 { double x, y; };

MP3MI_DEVFN int prep_ilog2(const mp3mi_tables *T, double v) // (int)(log(v)/log(2)), src/loop.c:633-667
{
#if defined(MP3MI_ULP_CENSUS) && !defined(MP3MI_EMU)
    if (v != 0.0) { // site 5: (int)(log(v) / log(2)): a log one ulp off moves the quotient by two of its ulps
        const double q = dm_log(v) / T->log2, fr = __builtin_fabs(q - __builtin_rint(q)), u = __builtin_fabs(q) * 0x1p-52;
        ULP_CENSUS(UC_SCFSI_LOG, fr <= 2.0 * u, fr <= 2.0 * u * 1048576.0);
    }
#endif
    return (v == 0.0) ? 0 : (int) (dm_log(v) / T->log2);
}

struct prep_walk_state {
    double tot, slog, accL, a0, a1, a2, amax;
    bool amb;
    int bandL, bandS, edgeL, edgeS; // next band to close and where it ends (wave-uniform)
};

// One spectral line.  w = line % 3 (the short-block window), line1 = line + 1.  Band energies are
// parked RAW in out->xmin[] when a band closes; prep_finish turns them into xmin and the log-energies.
template <bool EXACT>
MP3MI_DEVFN void prep_line(const mp3mi_tables *T, prep_walk_state &S, double &prod, int &esum, double x, int w, int line1, bool shortb, bool live,
                           mp3mi_loop_prep *out)
{
    const double sq = x * x, ax = __builtin_fabs(x);
    S.tot = S.tot + sq;
    S.accL = S.accL + sq;
    if (w == 0) S.a0 = S.a0 + sq; else if (w == 1) S.a1 = S.a1 + sq; else S.a2 = S.a2 + sq;
    S.amax = ax > S.amax ? ax : S.amax;
    if (EXACT) {
        double lg = 0.0;
        if (x != 0.0) lg = dm_log(sq); // (also below the normal range, and log(0) = -inf when xr^2 underflows)
        S.slog = S.slog + lg;
    } else {
        // sq = m 2^e: the product takes m, the exponent sum e; a zero line takes neither (src/loop.c:380-385), and
        // xr != 0 with xr^2 below the normal range is left to the second tier
        const long long sb = dm_bits(sq);
        const int ef = (int) (sb >> 52); // (sq >= 0: no sign bit)
        if (x != 0.0 && ef == 0) S.amb = true;
        const double m = dm_from_bits((sb & 0x000fffffffffffffLL) | 0x3ff0000000000000LL);
        prod = prod * (ef != 0 ? m : 1.0);
        esum += ef != 0 ? ef - 1023 : 0;
    }
    if (line1 == S.edgeL) { // a long scalefactor band ends here
        if (S.bandL < 21 && !shortb && live) out->xmin[S.bandL] = S.accL;
        S.accL = 0.0;
        S.bandL++;
        S.edgeL = S.bandL < 22 ? T->sfb_l[S.bandL + 1] : 577;
    }
    if (w == 2 && line1 == S.edgeS) { // a short scalefactor band ends here for all three windows
        if (S.bandS < 12 && shortb && live) {
            out->xmin[S.bandS * 3 + 0] = S.a0;
            out->xmin[S.bandS * 3 + 1] = S.a1;
            out->xmin[S.bandS * 3 + 2] = S.a2;
        }
        S.a0 = S.a1 = S.a2 = 0.0;
        S.bandS++;
        S.edgeS = S.bandS < 13 ? 3 * T->sfb_s[S.bandS + 1] : 577;
    }
}

template <int PH>
MP3MI_DEVFN void prep_block(const mp3mi_tables *T, prep_walk_state &S, double &prod, int &esum, const prep_d2 (&v)[PREP_BLOCK / 2], int k, bool shortb,
                            bool live, mp3mi_loop_prep *out)
{
#pragma unroll
    for (int j = 0; j < PREP_BLOCK; j++) {
        const double x = (j & 1) ? v[j >> 1].y : v[j >> 1].x;
        prep_line<false>(T, S, prod, esum, x, (PH + j) % 3, k + j + 1, shortb, live, out);
        PREP_SCHED_FENCE(); // one line at a time: the wavefronts of the SIMD hide the latency, not ILP across lines
    }
}

__global__ void __launch_bounds__(64, 3) k_prep(const mp3mi_tables *__restrict__ T, mp3mi_geom geo,
                                                const double *__restrict__ xr_all, const mp3mi_psy_out *__restrict__ psy,
                                                mp3mi_loop_prep *__restrict__ prep, const mp3mi_prep_fixlist *__restrict__ fix, int force_exact)
{
    const int lane = wave_lane();
    const size_t n_rec = (size_t) geo.n_streams * (size_t) geo.n_gran * (size_t) geo.channels;
    const size_t n_todo = fix ? (size_t) fix->count : n_rec; // (wave-uniform: the list was closed by the kernel before)
    for (size_t base = (size_t) blockIdx.x * 64; base < n_todo; base += (size_t) gridDim.x * 64) {
        const bool live = base + lane < n_todo;
        const size_t item = live ? base + lane : n_todo - 1;
        const size_t rec = fix ? (size_t) fix->list[item] : item;
        const mp3mi_psy_out *po = &psy[rec];
        mp3mi_loop_prep *out = &prep[rec];
        const double *row = xr_all + rec * 576;
        const bool shortb = po->block_type == 2;

        prep_walk_state S;
        int tp = 0;
        for (int pass = 0; pass < 2; pass++) {
            const bool exact = pass == 1 || force_exact != 0;
            S.tot = S.slog = S.accL = S.a0 = S.a1 = S.a2 = S.amax = 0.0;
            S.amb = false;
            S.bandL = S.bandS = 0;
            S.edgeL = T->sfb_l[1];
            S.edgeS = 3 * T->sfb_s[1];
            if (!exact) {
                double prod = 1.0; // first tier: product of the mantissas of the non-zero xr^2,
                int esum = 0;      // sum of their exponents
#pragma unroll 1
                for (int k = 0; k < 576; k += PREP_BLOCK) {
                    prep_d2 v[PREP_BLOCK / 2];
#pragma unroll
                    for (int q = 0; q < PREP_BLOCK / 2; q++) v[q] = *(const prep_d2 *) (row + k + 2 * q);
                    // 16 = 1 mod 3: the short-block window of the block's first line cycles 0, 1, 2 (wave-uniform)
                    const int ph = k % 3;
                    if (ph == 0) prep_block<0>(T, S, prod, esum, v, k, shortb, live, out);
                    else if (ph == 1) prep_block<1>(T, S, prod, esum, v, k, shortb, live, out);
                    else prep_block<2>(T, S, prod, esum, v, k, shortb, live, out);
                }
                S.slog = (double) esum * 0x1.62e42fefa39efp-1 + dm_log_fast(prod);
            } else { // second tier, rare: plain line-by-line walk
                double unused_p = 1.0;
                int unused_e = 0;
#pragma unroll 1
                for (int k = 0; k < 576; k++) prep_line<true>(T, S, unused_p, unused_e, row[k], k % 3, k + 1, shortb, live, out);
            }
            // quantanf_init (src/loop.c:369-402)
            tp = 0;
            if (S.tot != 0.0) {
                const double sfm = dm_exp(S.slog / 576.0) / (S.tot / 576.0);
                const double v = 8.0 * dm_log(sfm);
                tp = (v < 0) ? (int) (v - 0.5) : (int) (v + 0.5); // nint, src/loop.c:2020
#if defined(MP3MI_ULP_CENSUS) && !defined(MP3MI_EMU)
                if (exact && live) { // site 4: nint(8 ln sfm) from 576 logs, an exp and a log, each one ulp off: < 1e-12 absolute (header)
                    const double av = __builtin_fabs(v), fr = __builtin_fabs((av - __builtin_floor(av)) - 0.5);
                    ULP_CENSUS(UC_QUANTANF, fr <= 1e-12, fr <= 1e-12 * 1048576.0);
                }
#endif
                if (tp < -100) tp = -100;
                if (!exact) { // is nint(v) independent of the last bits of the logs?
                    const double av = __builtin_fabs(v), fr = av - __builtin_floor(av);
                    if (!(__builtin_fabs(fr - 0.5) > 2e-9 * (av > 1.0 ? av : 1.0))) S.amb = true; // also catches NaN
                }
            } else
                S.amb = false;
            if (exact || !wave_any(S.amb && live)) break;
        }
        if (!live) continue; // (no collective below)
        out->q0 = tp - 70;
        out->sc_en_tot = prep_ilog2(T, S.tot);
        out->sc_xrmax = (int) S.amax;
        out->nonzero = (S.amax != 0.0) ? 1 : 0;
        // calc_xmin (src/loop.c:1085-1118) and calc_scfsi's stored values (src/loop.c:642-667) from the parked energies
        if (shortb) {
            for (int b = 0; b < 12; b++) {
                const double cnt = (double) (T->sfb_s[b + 1] - T->sfb_s[b]);
                for (int w = 0; w < 3; w++) out->xmin[b * 3 + w] = po->ratio_s[b][w] * out->xmin[b * 3 + w] / cnt;
            }
        } else {
#pragma unroll 1
            for (int b = 0; b < 21; b++) {
                const double en = out->xmin[b];
                const double xmin = po->ratio_l[b] * en / (double) (T->sfb_l[b + 1] - T->sfb_l[b]);
                out->xmin[b] = xmin;
                out->sc_en[b] = prep_ilog2(T, en);   // truncation to int as the reference's statics do
                out->sc_xm[b] = prep_ilog2(T, xmin);
            }
        }
    }
}

void mp3mi_launch_prep(const mp3mi_tables *T, const mp3mi_geom &g, const double *xr, const mp3mi_psy_out *psy,
                       mp3mi_loop_prep *prep, const mp3mi_prep_fixlist *fix, int force_exact, hipStream_t st)
{
    const size_t n_rec = (size_t) g.n_streams * (size_t) g.n_gran * (size_t) g.channels;
    // a list is short (its length is not known here): a few wavefronts walk it
    const unsigned grid = fix ? 32u : (unsigned) ((n_rec + 63) / 64);
    hipLaunchKernelGGL(k_prep, dim3(grid), dim3(64), 0, st, T, g, xr, psy, prep, fix, force_exact);
}

ULP_CENSUS_ACCESSOR(mp3mi_debug_ulp_census_prep)
