// Stateless head of the iteration loop, for every (stream, granule, channel) of a chunk in
// parallel: calc_xmin (src/loop.c:1085-1118), the integer log-energies that calc_scfsi
// stores (src/loop.c:631-667) and quantanf_init (src/loop.c:369-402).  None of this depends
// on the bit reservoir, so it is lifted out of the serial kernel (k_loop) which starts each
// granule from the mp3mi_loop_prep record written here.
//
// One wavefront per (granule, channel).  Order-sensitive f64 sums (band energies, the
// 576-term total, the 576-term sum of logs) are formed by one lane per sum in index order;
// the terms are computed in parallel.  log/exp come from dmath.h.
#include "mp3mi_host.h"
#include "dmath.h"

struct prep_lds {
    double tmp[576];
    double total;
};

MP3MI_DEVFN double prep_seq_sum(const double *tmp, int first, int count, int stride)
{
    double sum = 0.0;
    int k = 0;
    for (; k + 8 <= count; k += 8) {
        const double *q = &tmp[first + k * stride];
        const double t0 = q[0], t1 = q[stride], t2 = q[2 * stride], t3 = q[3 * stride];
        const double t4 = q[4 * stride], t5 = q[5 * stride], t6 = q[6 * stride], t7 = q[7 * stride];
        sum = sum + t0; sum = sum + t1; sum = sum + t2; sum = sum + t3;
        sum = sum + t4; sum = sum + t5; sum = sum + t6; sum = sum + t7;
    }
    for (; k < count; k++) sum = sum + tmp[first + k * stride];
    return sum;
}

__global__ void __launch_bounds__(64) k_prep(const mp3mi_tables *__restrict__ T, mp3mi_geom geo,
                                             const double *__restrict__ xr_all, const mp3mi_psy_out *__restrict__ psy,
                                             mp3mi_loop_prep *__restrict__ prep)
{
    __shared__ prep_lds L;
    const int lane = wave_lane();
    const size_t rec = blockIdx.x;
    const mp3mi_psy_out *po = &psy[rec];
    mp3mi_loop_prep *out = &prep[rec];
    const bool shortb = po->block_type == 2;
    const int nband = shortb ? 36 : 21;

    double xr[9], amax = 0.0;
#pragma unroll
    for (int j = 0; j < 9; j++) {
        xr[j] = xr_all[rec * 576 + lane + 64 * j];
        L.tmp[lane + 64 * j] = xr[j] * xr[j];
        const double a = __builtin_fabs(xr[j]);
        amax = a > amax ? a : amax;
    }
    amax = wave_max_f64(amax);
    __syncthreads();

    // band energies on the band lanes, the total (src/loop.c:636-637, 378-386) on lane 63
    int first = 0, count = 0, stride = 1;
    if (lane < nband) {
        if (shortb) {
            const int sfb = lane / 3, w = lane - 3 * sfb;
            first = T->sfb_s[sfb] * 3 + w;
            count = T->sfb_s[sfb + 1] - T->sfb_s[sfb];
            stride = 3;
        } else {
            first = T->sfb_l[lane];
            count = T->sfb_l[lane + 1] - first;
        }
    } else if (lane == 63)
        count = 576;
    const double en = prep_seq_sum(L.tmp, first, count, stride);
    if (lane == 63) L.total = en;
    double xmin = 0.0;
    if (lane < nband) {
        const double ratio = shortb ? po->ratio_s[lane / 3][lane % 3] : po->ratio_l[lane];
        xmin = ratio * en / (double) count;
        out->xmin[lane] = xmin;
    }
    if (!shortb && lane < 21) { // src/loop.c:642-667, truncation to int as the reference's statics do
        out->sc_en[lane] = (en == 0.0) ? 0 : (int) (dm_log(en) / T->log2);
        out->sc_xm[lane] = (xmin == 0.0) ? 0 : (int) (dm_log(xmin) / T->log2);
    }
    __syncthreads();
    const double en_total = L.total;
    __syncthreads();

    // quantanf_init: sum of log(xr^2) over the non-zero lines, in index order
#pragma unroll
    for (int j = 0; j < 9; j++) L.tmp[lane + 64 * j] = (xr[j] != 0) ? dm_log(xr[j] * xr[j]) : 0.0;
    __syncthreads();
    if (lane == 63) {
        const double s1 = prep_seq_sum(L.tmp, 0, 576, 1);
        int tp = 0;
        if (en_total != 0.0) {
            const double sfm = dm_exp(s1 / 576.0) / (en_total / 576.0);
            const double v = 8.0 * dm_log(sfm);
            tp = (v < 0) ? (int) (v - 0.5) : (int) (v + 0.5); // nint, src/loop.c:2020
            if (tp < -100) tp = -100;
        }
        out->q0 = tp - 70;
        out->sc_en_tot = (en_total == 0.0) ? 0 : (int) (dm_log(en_total) / T->log2);
        out->sc_xrmax = (int) amax;
        out->nonzero = (amax != 0.0) ? 1 : 0;
    }
}

void mp3mi_launch_prep(const mp3mi_tables *T, const mp3mi_geom &g, const double *xr, const mp3mi_psy_out *psy,
                       mp3mi_loop_prep *prep, hipStream_t st)
{
    const unsigned grid = (unsigned) (g.n_streams * g.n_gran * g.channels);
    hipLaunchKernelGGL(k_prep, dim3(grid), dim3(64), 0, st, T, g, xr, psy, prep);
}
