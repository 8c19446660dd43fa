// Stateless head of the iteration loop, for every (stream, granule, channel) of a chunk in
// parallel: calc_xmin (src/loop.c:1085-1118), the integer log-energies that calc_scfsi
// stores (src/loop.c:631-667) and quantanf_init (src/loop.c:369-402).  None of this depends
// on the bit reservoir, so it is lifted out of the serial kernel (k_loop) which starts each
// granule from the mp3mi_loop_prep record written here.
//
// Everything here is an ORDER-SENSITIVE f64 sum over the 576 lines of a granule (the total
// energy, the band energies, the sum of logs), i.e. a serial chain per granule.  So the lanes
// of a wavefront are 64 different granules, each walking its own 576 lines in index order;
// the spectra are read from HBM coalesced, 24 lines of 64 granules at a time, and transposed
// through LDS (row stride 25 doubles: conflict-free column walk).
//
// quantanf_init needs sum(log(xr^2)) only to round 8*ln(sfm) to an integer.  The first tier
// uses dm_log_fast (plain double, |error| < 2^-50 max(1,|log|)), which moves 8*ln(sfm) by less
// than 1e-12; unless the value lies within 1e-9 of a rounding boundary of nint() the integer
// is already decided.  Otherwise (probability ~1e-9 per granule) the wavefront repeats the walk
// with the correctly rounded dm_log, which is what the reference's libm call amounts to.
// MP3MI_PREP_EXACT=1 forces the second tier (tests run both).
#include "mp3mi_host.h"
#include "dmath.h"

#define PREP_TILE 24 /* lines per LDS tile; a multiple of 3 so that short-block windows stay aligned */
#define PREP_ROW (PREP_TILE + 1)

struct prep_lds {
    double x[64][PREP_ROW];
};

MP3MI_DEVFN int prep_ilog2(const mp3mi_tables *T, double v) // (int)(log(v)/log(2)), src/loop.c:633-667
{
    return (v == 0.0) ? 0 : (int) (dm_log(v) / T->log2);
}

__global__ void __launch_bounds__(64) k_prep(const mp3mi_tables *__restrict__ T, mp3mi_geom geo,
                                             const double *__restrict__ xr_all, const mp3mi_psy_out *__restrict__ psy,
                                             mp3mi_loop_prep *__restrict__ prep, int force_exact)
{
    __shared__ prep_lds L;
    const int lane = wave_lane();
    const size_t n_rec = (size_t) geo.n_streams * (size_t) geo.n_gran * (size_t) geo.channels;
    const size_t rec0 = (size_t) blockIdx.x * 64;
    const bool live = rec0 + lane < n_rec;
    const size_t rec = live ? rec0 + lane : n_rec - 1;
    const mp3mi_psy_out *po = &psy[rec];
    mp3mi_loop_prep *out = &prep[rec];
    const bool shortb = po->block_type == 2;
    // transposing copy: lanes 0..23 carry granule 2*it, lanes 24..47 granule 2*it + 1
    const int cp_g = lane >= PREP_TILE ? 1 : 0, cp_line = lane - PREP_TILE * cp_g;
    const bool cp_on = lane < 2 * PREP_TILE;

    for (int pass = 0; pass < 2; pass++) {
        const bool exact = pass == 1 || force_exact != 0;
        double tot = 0.0, slog = 0.0, accL = 0.0, a0 = 0.0, a1 = 0.0, a2 = 0.0, amax = 0.0;
        bool amb = false;
        int bandL = 0, bandS = 0;               // next band to close (wave-uniform)
        int edgeL = T->sfb_l[1], edgeS = 3 * T->sfb_s[1];
        for (int t = 0; t < 576 / PREP_TILE; t++) {
            wave_sync();
#pragma unroll 8
            for (int it = 0; it < 32; it++) {
                const size_t r = rec0 + 2 * it + cp_g;
                if (cp_on) L.x[2 * it + cp_g][cp_line] = (r < n_rec) ? xr_all[r * 576 + t * PREP_TILE + cp_line] : 0.0;
            }
            wave_sync();
            for (int st = 0; st < PREP_TILE / 3; st++) {
                const int k = t * PREP_TILE + 3 * st;
                const double xs[3] = {L.x[lane][3 * st], L.x[lane][3 * st + 1], L.x[lane][3 * st + 2]};
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const double x = xs[j], sq = x * x, ax = __builtin_fabs(x);
                    tot = tot + sq;
                    accL = accL + sq;
                    if (j == 0) a0 = a0 + sq; else if (j == 1) a1 = a1 + sq; else a2 = a2 + sq;
                    amax = ax > amax ? ax : amax;
                    double lg = 0.0;
                    if (x != 0.0) {
                        if (sq < 0x1p-1022) amb = true; // below the normal range: only dm_log handles it
                        else lg = exact ? dm_log(sq) : dm_log_fast(sq);
                    }
                    slog = slog + lg;
                    if (k + j + 1 == edgeL) { // a long scalefactor band ends here (wave-uniform)
                        if (bandL < 21 && !shortb && live) {
                            const double en = accL;
                            const double xmin = po->ratio_l[bandL] * en / (double) (edgeL - T->sfb_l[bandL]);
                            out->xmin[bandL] = xmin;
                            out->sc_en[bandL] = prep_ilog2(T, en);   // truncation to int as the reference's statics do
                            out->sc_xm[bandL] = prep_ilog2(T, xmin);
                        }
                        accL = 0.0;
                        bandL++;
                        edgeL = bandL < 22 ? T->sfb_l[bandL + 1] : 577;
                    }
                }
                if (k + 3 == edgeS) { // a short scalefactor band ends here for all three windows
                    if (bandS < 12 && shortb && live) {
                        const double cnt = (double) (T->sfb_s[bandS + 1] - T->sfb_s[bandS]);
                        out->xmin[bandS * 3 + 0] = po->ratio_s[bandS][0] * a0 / cnt;
                        out->xmin[bandS * 3 + 1] = po->ratio_s[bandS][1] * a1 / cnt;
                        out->xmin[bandS * 3 + 2] = po->ratio_s[bandS][2] * a2 / cnt;
                    }
                    a0 = a1 = a2 = 0.0;
                    bandS++;
                    edgeS = bandS < 13 ? 3 * T->sfb_s[bandS + 1] : 577;
                }
            }
        }
        // quantanf_init (src/loop.c:369-402)
        int tp = 0;
        if (tot != 0.0) {
            const double sfm = dm_exp(slog / 576.0) / (tot / 576.0);
            const double v = 8.0 * dm_log(sfm);
            tp = (v < 0) ? (int) (v - 0.5) : (int) (v + 0.5); // nint, src/loop.c:2020
            if (tp < -100) tp = -100;
            if (!exact) { // is nint(v) independent of the last bits of the logs?
                const double av = __builtin_fabs(v), fr = av - __builtin_floor(av);
                if (!(__builtin_fabs(fr - 0.5) > 1e-9 * (av > 1.0 ? av : 1.0))) amb = true; // also catches NaN
            }
        } else
            amb = false;
        if (live) {
            out->q0 = tp - 70;
            out->sc_en_tot = prep_ilog2(T, tot);
            out->sc_xrmax = (int) amax;
            out->nonzero = (amax != 0.0) ? 1 : 0;
        }
        if (exact || !wave_any(amb && live)) break;
    }
}

void mp3mi_launch_prep(const mp3mi_tables *T, const mp3mi_geom &g, const double *xr, const mp3mi_psy_out *psy,
                       mp3mi_loop_prep *prep, int force_exact, hipStream_t st)
{
    const size_t n_rec = (size_t) g.n_streams * (size_t) g.n_gran * (size_t) g.channels;
    hipLaunchKernelGGL(k_prep, dim3((unsigned) ((n_rec + 63) / 64)), dim3(64), 0, st, T, g, xr, psy, prep, force_exact);
}
