// Layers I and II (l12_dev.h): psychoacoustic model 2 behind the FFT, and the frame encoder.
//
//   k12_psy    one wavefront per (stream, pass, channel) record: the unpredictability of all 513 lines, the partition
//              sums, spreading, tonality, masking threshold per line and -- Layer II -- the signal-to-mask ratio of
//              the 32 subbands (src/psy.c:282-386), by runs of consecutive passes of one (stream, channel).
//   k12_alloc  one wavefront per (stream, frame), a lane per (subband, channel): scale factors, transmission pattern,
//              joint-stereo bound, the greedy bit allocation, CRC, quantisation, and the frame's bits assembled in LDS
//              (src/encode.c:512-1416, src/common.c:1251-1327).
//
// Floating point: the reference's types expression by expression (FLOAT = float, FLT_EVAL_METHOD 0, no contraction).
// Run-time transcendentals (sin cos log exp on data) come from dmath.h in two tiers, as on the Layer III path: a plain
// double value with a proven error bound decides every FLOAT rounding it can, and the rest is repeated with the
// correctly rounded function.
#include "mp3mi_host.h"
#include "l12_dev.h"
#include "dmath.h"

#define R_LN_TO_LOG10 0.2302585093

// does every value within err of v round to the same float?  (rounding is monotone: the two ends decide)
MP3MI_DEVFN bool l12_float_decided(double v, double err) { return (float) (v - err) == (float) (v + err); }

struct psy12_lds {
    float e[L12_ROW], c[L12_ROW], thr[L12_ROW];
    float thr_prev[L12_ROW]; // Layer I: the thresholds of the pass before, before pre-echo control ("lthr" / 32, src/psy.c:355-361)
    float2 g2[64]; // grouped_e, grouped_c of partition k
    float nb[64];
};

// c[j] of src/psy.c:283-292 for one line, in two tiers.
//
// First tier (psy12_c0): with a = r, b = r' (floats: a - b, and 4 a b, are exact doubles) and D = phi - phi',
//     |a e^(i phi) - b e^(i phi')|^2  =  a^2 + b^2 - 2 a b cos D  =  (a - b)^2 + 4 a b sin^2(D / 2)
// -- ONE sine instead of the reference's two sines and two cosines, and for a b >= 0 a sum of two non-negative terms:
// no cancellation, however well the line is predicted.  dm_sin_fast_rel is within 2^-49 RELATIVE of the sine (D / 2 is
// exact: the subtraction is checked with its TwoSum error term, the halving is a power of two), so cw = sqrt(x) / (a + |b|)
// is within 2^-48 RELATIVE of the mathematical quotient (for a b < 0, where the second term is subtracted from
// (a + |b|)^2, within 2^-51 / cw absolute).  The REFERENCE's double arithmetic is not that close to it: each of
// t1 = r cos phi - r' cos phi', t2 = ... carries (r + |r'|) 2^-52 of rounding (two products, a libm that may be off by
// an ulp, the difference), their root sqrt 2 times that, and the divisor is r + |r'|: its quotient lies within 2^-51.5 +
// 3 2^-53 cw of the mathematical one.  So the float is decided unless it lies within 2^-50 + 2^-47 cw (+ 2^-51 / cw) of a
// rounding boundary: 2^-15 of the lines at cw = 2^-10 -- and every line predicted better than 2^-25, whose float
// the reference's own rounding noise decides.  *undecided says whether the float could differ.
// Second tier (psy12_c_exact): the reference's formula operation by operation with the correctly rounded dm_sincos,
// behind the loop.
MP3MI_DEVFN float psy12_c0(float r_new, float phi_new, float r_old, float r_oldest, float phi_old, float phi_oldest, bool *undecided)
{
    const float r_prime = (float) (2.0 * (double) r_old - (double) r_oldest);
    const float phi_prime = (float) (2.0 * (double) phi_old - (double) phi_oldest);
    const double a = (double) r_new, b = (double) r_prime, pn = (double) phi_new, pp = (double) phi_prime;
    const double dl = pn - pp, bb = dl - pn, dl_err = (pn - (dl - bb)) + (-pp - bb); // TwoSum: dl + dl_err == pn - pp exactly
    int near_multiple;
    const double sn = dm_sin_fast_rel(0.5 * dl, &near_multiple);
    const double d = a - b, m = (a + a) * (b + b);
    const double x = d * d + m * (sn * sn);
    const double t3 = a + __builtin_fabs(b);
    double cw = 0.0;
    if (t3 != 0.0) cw = __builtin_sqrt(x > 0.0 ? x : 0.0) / t3;
    const double err = 0x1p-50 + cw * 0x1p-47 + (m >= 0.0 ? 0.0 : 0x1p-51 / cw);
    if (t3 != 0.0 && !(dl_err == 0.0 && !near_multiple && (m >= 0.0 || cw > 0x1p-20) && l12_float_decided(cw, err))) *undecided = true;
    return (float) cw;
}

MP3MI_DEVFN float psy12_c_exact(float r_new, float phi_new, float r_old, float r_oldest, float phi_old, float phi_oldest)
{
    const float r_prime = (float) (2.0 * (double) r_old - (double) r_oldest);
    const float phi_prime = (float) (2.0 * (double) phi_old - (double) phi_oldest);
    const double rn = (double) r_new, rp = (double) r_prime;
    double s2, c2, sp, cp;
    dm_sincos((double) phi_new, &s2, &c2);
    dm_sincos((double) phi_prime, &sp, &cp);
    const double t1 = rn * c2 - rp * cp;
    const double t2 = rn * s2 - rp * sp;
    const double t3 = rn + __builtin_fabs(rp);
    double cw = 0.0;
    if (t3 != 0.0) cw = __builtin_sqrt(t1 * t1 + t2 * t2) / t3;
    return (float) cw;
}

// the signal-to-mask ratio of subband sb from the energies and thresholds of its lines, src/psy.c:367-386.
// The logarithm in two tiers: dm_log_fast is within 2^-50 max(1, |log x|) of the logarithm.
MP3MI_DEVFN float psy12_snr_band(const float *e, const float *fthr, int sb, bool exact)
{
    const int j = 16 * sb;
    // one walk for both forms -- the 13 lower subbands take the smallest threshold of their 17 lines, the upper ones the
    // sum (src/psy.c:367-386) --: a wavefront that branched would walk twice
    float minthres = 60802371420160.0f, sumthres = 0.0f, sum_energy = 0.0f;
#pragma unroll
    for (int k = 0; k < 17; k++) {
        const float f = fthr[j + k];
        if (minthres > f) minthres = f;
        sumthres = sumthres + f;
        sum_energy = sum_energy + e[j + k];
    }
    const float x = sb < 13 ? (float) ((double) sum_energy / ((double) minthres * 17.0)) : sum_energy / sumthres;
    const bool normal = x >= 0x1p-126f && x < __builtin_inff();
#if defined(MP3MI_ULP_CENSUS) && !defined(MP3MI_EMU)
    if (normal) { // site: (float)(4.342944819 log x): one ulp of the logarithm
        const double lv = dm_log((double) x), v = 4.342944819 * lv;
        const double ulp = dm_from_bits((dm_bits(lv) & 0x7ff0000000000000LL)) * 0x1p-52;
        ULP_CENSUS(UC_L12_SNR, !l12_float_decided(v, 4.342944819 * ulp), !l12_float_decided(v, 4.342944819 * ulp * 1048576.0));
    }
#endif
    if (!exact && normal) {
        const double lv = dm_log_fast((double) x), v = 4.342944819 * lv;
        const double al = __builtin_fabs(lv);
        if (l12_float_decided(v, 0x1p-47 * (al > 1.0 ? al : 1.0))) return (float) v;
    }
    return (float) (4.342944819 * dm_log((double) x));
}

// diagnostic build: the site of c[j] (see ULP_CENSUS, mp3mi_dev.h)
#if defined(MP3MI_ULP_CENSUS) && !defined(MP3MI_EMU)
#define L12_CENSUS_C(rn_, pn_, ro_, roo_, po_, poo_)                                                                         \
    if (on) { /* c[j] = (float)(sqrt(t1^2 + t2^2) / t3); one ulp of each sine and cosine moves the quotient by < 2^-52 */   \
        const float r_prime = (float) (2.0 * (double) (ro_) - (double) (roo_)), phi_prime = (float) (2.0 * (double) (po_) - (double) (poo_)); \
        if (!((double) (rn_) + __builtin_fabs((double) r_prime) == 0.0) && !((rn_) == r_prime && (pn_) == phi_prime)) {      \
            double s2, c2, sp, cp;                                                                                          \
            dm_sincos((double) (pn_), &s2, &c2);                                                                            \
            dm_sincos((double) phi_prime, &sp, &cp);                                                                        \
            const double t1 = (double) (rn_) * c2 - (double) r_prime * cp, t2 = (double) (rn_) * s2 - (double) r_prime * sp;  \
            const double cwx = __builtin_sqrt(t1 * t1 + t2 * t2) / ((double) (rn_) + __builtin_fabs((double) r_prime));      \
            ULP_CENSUS(UC_L12_C, !l12_float_decided(cwx, 0x1p-52), !l12_float_decided(cwx, 0x1p-32));                       \
        }                                                                                                                   \
    }
#else
#define L12_CENSUS_C(rn_, pn_, ro_, roo_, po_, poo_)
#endif
#define L12_PSY_RUN 32 /* passes of one (stream, channel) a wavefront of k12_psy takes in a row */

__global__ void __launch_bounds__(64, 3) k12_psy(const mp3mi_tables_l12 *__restrict__ T, l12_geom geo,
                                              const float *__restrict__ erp, float *__restrict__ snr)
{
    __shared__ psy12_lds L;
    const int lane = wave_lane();
    const int C = geo.channels, NP = geo.np;
    // Records of this launch: the chunk's own passes.  A persistent wavefront takes RUNS of consecutive passes of one (stream, channel): the magnitudes
    // and phases of the two passes before the current one -- what the prediction of a line is made of -- stay in
    // registers from pass to pass (lane l: lines l + 64 k), so a record costs three rows of memory reads, not seven
    // (plus four per run to start it).  Layer I's pre-echo control looks at the thresholds of the pass before: a run starts
    // one pass early there (its thresholds stay in LDS for the next pass; nothing of it is written).
    const int qi0 = geo.lb, nq = NP - qi0, warm = geo.layer == 1 ? 1 : 0;
    const int n_run = (nq + L12_PSY_RUN - 1) / L12_PSY_RUN;
    const unsigned n_item = (unsigned) geo.n_streams * (unsigned) n_run * (unsigned) C;
    const bool cw_exact = (geo.test_flags >> 5) & 1, psy_exact = (geo.test_flags >> 2) & 1;
    const double tmn = T->tmn[lane < L12_CB ? lane : 0];
    const float bm = T->bmaxv[lane < L12_CB ? lane : 0], rn_nl = T->rn_nl[lane < L12_CB ? lane : 0];
    const int pj0 = lane < T->npart ? T->part_first[lane] : 0, pj1 = lane < T->npart ? T->part_first[lane + 1] : 0;
#pragma unroll 1
    for (unsigned bid = blockIdx.x; bid < n_item; bid += gridDim.x) {
    const int ch = (int) (bid % (unsigned) C);
    const int run = (int) ((bid / (unsigned) C) % (unsigned) n_run);
    const int s = (int) (bid / (unsigned) (C * n_run));
    const int qa = qi0 + run * L12_PSY_RUN - warm, qb = qa + warm + L12_PSY_RUN < NP ? qa + warm + L12_PSY_RUN : NP;
    float h_ro[9], h_po[9], h_roo[9], h_poo[9]; // r, phi of the pass before ("old") and of the one before that ("oldest")
    {
        const float *r_o = erp + (((size_t) s * NP + qa - 1) * C + ch) * (3 * L12_ROW);
        const float *r_oo = erp + (((size_t) s * NP + qa - 2) * C + ch) * (3 * L12_ROW);
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const int ii = lane + 64 * k < L12_HBLK ? lane + 64 * k : 0;
            h_ro[k] = r_o[L12_ROW + ii]; h_po[k] = r_o[2 * L12_ROW + ii];
            h_roo[k] = r_oo[L12_ROW + ii]; h_poo[k] = r_oo[2 * L12_ROW + ii];
        }
    }
#pragma unroll 1
    for (int qi = qa; qi < qb; qi++) {
    const long q = (geo.fabs0 + geo.f0) * geo.layer - geo.lb + qi; // the pass counted from the stream's first
    const size_t rec = ((size_t) s * NP + qi) * C + ch;
    const float *r_n = erp + rec * (3 * L12_ROW);               // this pass: energy, r, phi
    const float *r_o = erp + (rec - (size_t) C) * (3 * L12_ROW);     // (the second tier reads the passes before from memory)
    const float *r_oo = erp + (rec - 2 * (size_t) C) * (3 * L12_ROW);
    wave_sync(); // the record before is done with the LDS

    // ---- unpredictability of every line, src/psy.c:282-292
    unsigned redo = 0; // bit k: line lane + 64 k needs the second tier
    {
        float en[9], rn[9], pn[9];
#pragma unroll
        for (int k = 0; k < 9; k++) { // (all 27 loads in flight together)
            const int ii = lane + 64 * k < L12_HBLK ? lane + 64 * k : 0;
            en[k] = r_n[ii]; rn[k] = r_n[L12_ROW + ii]; pn[k] = r_n[2 * L12_ROW + ii];
        }
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const int i = lane + 64 * k;
            const bool on = i < L12_HBLK;
            bool undecided = cw_exact;
            float c = 0.0f;
            if (!undecided) c = psy12_c0(rn[k], pn[k], h_ro[k], h_roo[k], h_po[k], h_poo[k], &undecided);
            if (undecided && on) redo |= 1u << k;
            L12_CENSUS_C(rn[k], pn[k], h_ro[k], h_roo[k], h_po[k], h_poo[k]);
            if (on) { L.e[i] = en[k]; L.c[i] = c; }
            h_roo[k] = h_ro[k]; h_poo[k] = h_po[k]; // the next pass's "oldest" and "old"
            h_ro[k] = rn[k]; h_po[k] = pn[k];
        }
    }
    if (q < 0) continue; // (Layer I's warm-up pass of a stream's first run lies before the stream: the initial lthr stands in for it, below; its r = phi = 0 are history all the same)
    // the second tier -- the reference's formula with correctly rounded sines and cosines -- for the lines the first could not
    // decide, behind the loop: it is rare, and inlined into the loop its double-double arithmetic would set the loop's
    // register budget
    if (wave_any(redo != 0)) {
#pragma unroll 1
        for (int k = 0; k < 9; k++) {
            const bool mine = (redo >> k) & 1u;
            if (!wave_any(mine)) continue;
            const int i = lane + 64 * k, ii = mine ? i : 0;
            const float cx = psy12_c_exact(r_n[L12_ROW + ii], r_n[2 * L12_ROW + ii], r_o[L12_ROW + ii], r_oo[L12_ROW + ii],
                                           r_o[2 * L12_ROW + ii], r_oo[2 * L12_ROW + ii]);
            if (mine) L.c[i] = cx;
        }
    }
    wave_sync();

    // ---- partition sums in line order, src/psy.c:297-306: a lane per partition
    {
        float ge = 0.0f, gc = 0.0f;
        {
#pragma unroll 4
            for (int j = pj0; j < pj1; j++) {
                const float ev = L.e[j];
                ge = ge + ev;
                gc = gc + ev * L.c[j];
            }
        }
        L.g2[lane] = make_float2(ge, gc);
    }
    wave_sync();

    // ---- spreading, tonality, masking level of partition j = lane, src/psy.c:312-348
    {
        float nbv = 0.0f;
        const int j = lane < L12_CB ? lane : 0;
        float ecb = 0.0f, cb = 0.0f;
        // row j of the spreading function in two halves of 32 (8 loads of 16 bytes each, served by the CU's vector
        // cache: the table is 16 KB), fetched here and not kept: 64 registers for the whole kernel otherwise
#pragma unroll
        for (int h = 0; h < 2; h++) {
            float srow[32];
            const float4 *rp = (const float4 *) (T->spread_r[j] + 32 * h);
#if !defined(MP3MI_EMU)
            __asm__ volatile("" : "+v"(rp));
#endif
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const float4 v = rp[k];
                srow[4 * k] = v.x; srow[4 * k + 1] = v.y; srow[4 * k + 2] = v.z; srow[4 * k + 3] = v.w;
            }
#pragma unroll
            for (int k = 0; k < 32; k++) { // (the reference skips s == 0: adding 0 * x changes nothing; k = 63 is padding: 0 * 0)
                const float2 g = L.g2[32 * h + k];
                ecb = ecb + srow[k] * g.x;
                cb = cb + srow[k] * g.y;
#if !defined(MP3MI_EMU)
                if ((k & 7) == 7) __asm__ volatile("" ::: "memory"); // (eight broadcast reads in flight, not all: registers)
#endif
            }
        }
        if (ecb != 0.0f) cb = cb / ecb;
        else cb = 0.0f;
        if ((double) cb < .05) cb = (float) 0.05;
        else if ((double) cb > .5) cb = (float) 0.5;
        const double nmt = 5.5;
        float bc;
        {
            bool done = false;
            bc = 0.0f;
            if (!psy_exact) { // log within 3 2^-50 on [0.05, 0.5]: v within 2^-44
                const double tb = -0.434294482 * dm_log_fast((double) cb) - 0.301029996;
                const double v = tmn * tb + nmt * (1.0 - tb);
                if (l12_float_decided(v, 0x1p-44)) { bc = (float) v; done = true; }
            }
            if (wave_any(!done)) {
                const double tb = -0.434294482 * dm_log((double) cb) - 0.301029996;
                const float bx = (float) (tmn * tb + nmt * (1.0 - tb));
                bc = done ? bc : bx;
            }
        }
#if defined(MP3MI_ULP_CENSUS) && !defined(MP3MI_EMU)
        if (lane < L12_CB) { // site: bc = (float)(tmn tb + nmt (1 - tb)); one ulp of the logarithm (2^-51 on [0.69, 3]) moves it by < (tmn - 5.5) 0.434 2^-51
            const double tb = -0.434294482 * dm_log((double) cb) - 0.301029996, v = tmn * tb + nmt * (1.0 - tb);
            const double band = __builtin_fabs(tmn - nmt) * 0.434294482 * 0x1p-51;
            ULP_CENSUS(UC_L12_BC, !l12_float_decided(v, band), !l12_float_decided(v, band * 1048576.0));
        }
#endif
        bc = (bc > bm) ? bc : bm;
#if defined(MP3MI_ULP_CENSUS) && !defined(MP3MI_EMU)
        if (lane < L12_CB) { // site: (float) exp(-bc ln10/10): one ulp of the exponential
            const double v = dm_exp((double) -bc * R_LN_TO_LOG10);
            ULP_CENSUS(UC_L12_EXP, !l12_float_decided(v, v * 0x1p-52), !l12_float_decided(v, v * 0x1p-32));
        }
#endif
        {
            const double arg = (double) -bc * R_LN_TO_LOG10;
            bool done = false;
            float bx = 0.0f;
            if (!psy_exact) { // exp within 2^-50 relative
                const double v = dm_exp_fast(arg);
                if (l12_float_decided(v, v * 0x1p-48)) { bx = (float) v; done = true; }
            }
            if (wave_any(!done)) {
                const float by = (float) dm_exp(arg);
                bx = done ? bx : by;
            }
            bc = bx;
        }
        if (rn_nl != 0.0f) nbv = ecb * bc / rn_nl;
        L.nb[lane] = lane < L12_CB ? nbv : 0.0f;
    }
    wave_sync();

    // ---- threshold of every line before pre-echo control, src/psy.c:349-353 ("temp1": the larger of two floats)
    {
        const uint8_t *pt = T->partition;
        const float *at = T->absthr;
#if !defined(MP3MI_EMU)
        __asm__ volatile("" : "+v"(pt), "+v"(at)); // (the tables do not depend on the record: not hoisted out of the record loop, 18 registers)
#endif
        for (int i = lane; i < L12_HBLK; i += 64) {
            const float t = L.nb[pt[i]], a = at[i];
            const float v = (t > a) ? t : a;
            if (geo.layer == 1) {
                // pre-echo control, src/psy.c:355-361: limited by 32 x the threshold of the pass before (lthr starts at
                // 60802371420160.0, :161-162), floored at 0.00316 x its own value
                const double temp1 = (double) v;
                const float lthr = q >= 1 ? (float) (32.0 * (double) L.thr_prev[i]) : 60802371420160.0f;
                float f = (temp1 < (double) lthr) ? v : lthr;
                const double temp2 = temp1 * 0.00316;
                f = (temp2 > (double) f) ? (float) temp2 : f;
                L.thr[i] = f;
                L.thr_prev[i] = v;
            } else L.thr[i] = v;
        }
    }
    if (qi < qa + warm) continue; // (Layer I's warm-up pass: its thresholds were all that was wanted)
    wave_sync();
    if (lane < 32) snr[rec * 32 + lane] = psy12_snr_band(L.e, L.thr, lane, psy_exact);
    }
    }
}

void mp3mi_launch_l12_psy(const mp3mi_tables_l12 *T, const l12_geom &g, const float *erp, float *snr, hipStream_t st)
{
    const int nq = g.np - g.lb;
    const size_t n_item = (size_t) g.n_streams * ((nq + L12_PSY_RUN - 1) / L12_PSY_RUN) * g.channels;
    const int n_wave = mp3mi_current_cu_count() * 12; // resident wavefronts: 12 per CU at the kernel's 168 registers
    hipLaunchKernelGGL(k12_psy, dim3((unsigned) (n_item < (size_t) n_wave ? n_item : (size_t) n_wave)), dim3(64), 0, st, T, g, erp, snr);
}

// ------------------------------------------------------------------------------------------------------------------
// the frame encoder
// ------------------------------------------------------------------------------------------------------------------
// pattern[class0][class1] of II_transmission_pattern (src/encode.c:647-651) as a code:
// 0 = 0x123, 1 = 0x122, 2 = 0x133, 3 = 0x113, 4 = 0x111, 5 = 0x222, 6 = 0x333, 7 = 0x444
__device__ static const uint8_t L12_PATTERN[5][5] = {{0, 1, 1, 2, 0}, {3, 4, 4, 7, 3}, {4, 4, 4, 6, 3}, {5, 5, 5, 6, 0}, {0, 1, 1, 2, 0}};
#define L12_IMG_WORDS 448 /* 1792 bytes: the largest frame is 1728 (Layer II, 384 kbps at 32 kHz) */
struct alloc12_lds {
    unsigned img[L12_IMG_WORDS];
    double multiple[64];
    double snr[18];  // the signal-to-noise ratio of a quantiser: read once per granted step, in the loop's dependent chain
    uint16_t al[32][16][4];
    int ba[2][32], sf[2][32];
    unsigned crc;
};

MP3MI_DEVFN void l12_or(unsigned *w, unsigned v)
{
#if defined(MP3MI_EMU)
    *w |= v;
#else
    atomicOr(w, v);
#endif
}
// the low n bits of val at bit position pos (MSB first) of a word image, src/common.c:1134-1161
MP3MI_DEVFN void l12_put(unsigned *img, int pos, unsigned val, int n)
{
    if (n <= 0) return;
    if (n < 32) val &= (1u << n) - 1u;
    const int wi = pos >> 5, off = pos & 31;
    if (off + n <= 32) l12_or(&img[wi], val << (32 - off - n));
    else {
        const int n2 = off + n - 32;
        l12_or(&img[wi], val >> n2);
        l12_or(&img[wi + 1], val << (32 - n2));
    }
}
// exclusive prefix sum over the wave in lane order; *total = the wave's sum
MP3MI_DEVFN int l12_scan(int v, int *total)
{
    const int lane = wave_lane();
    int incl = v;
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, (unsigned) d);
        if (lane >= d) incl += o;
    }
    *total = __shfl(incl, 63);
    return incl - v;
}
// index of the scalefactor just above the peak m: the largest j <= 62 with m <= multiple[j], 0 if there is none
// (src/encode.c:529-534, 551-556; multiple[] falls with j)
MP3MI_DEVFN int l12_scale_index(const double *mult, double m)
{
    int lo = 0, hi = 63; // count of j in [0, 63) with multiple[j] >= m lies in [lo, hi]
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1; // is multiple[mid - 1] >= m ?
        if (mult[mid - 1] >= m) lo = mid;
        else hi = mid - 1;
    }
    return lo > 0 ? lo - 1 : 0;
}
// a double as a 64-bit key whose unsigned order is the REVERSE of the doubles' order (no NaN; -0 never occurs here):
// the smallest ratio has the largest key, and 0 -- below every key of a number -- marks a band that is not a candidate
MP3MI_DEVFN unsigned long long l12_key(double d)
{
    const long long b = dm_bits(d);
    return ~(unsigned long long) (b ^ ((b >> 63) | (long long) 0x8000000000000000ull));
}
// wave maximum of 64-bit keys: DPP steps inside the rows, row broadcasts across them; a lane without a source reads 0,
// the identity (bound_ctrl), so a step is two DPP moves, one 64-bit compare and two selects
MP3MI_DEVFN unsigned long long l12_wave_max_u64(unsigned long long v)
{
#if defined(MP3MI_EMU)
    for (int m = 32; m >= 1; m >>= 1) { const unsigned long long o = __shfl_xor(v, m); v = o > v ? o : v; }
    return v;
#else
#define L12_MAX_STEP(ctrl)                                                                                  \
    {                                                                                                       \
        const unsigned olo = (unsigned) __builtin_amdgcn_update_dpp(0, (int) (unsigned) v, ctrl, 0xf, 0xf, true); \
        const unsigned ohi = (unsigned) __builtin_amdgcn_update_dpp(0, (int) (unsigned) (v >> 32), ctrl, 0xf, 0xf, true); \
        const unsigned long long o = ((unsigned long long) ohi << 32) | olo;                                \
        v = o > v ? o : v;                                                                                  \
    }
    L12_MAX_STEP(0xB1)  /* quad_perm [1,0,3,2] */
    L12_MAX_STEP(0x4E)  /* quad_perm [2,3,0,1] */
    L12_MAX_STEP(0x141) /* row_half_mirror */
    L12_MAX_STEP(0x140) /* row_mirror */
    L12_MAX_STEP(0x142) /* row_bcast15 */
    L12_MAX_STEP(0x143) /* row_bcast31 */
#undef L12_MAX_STEP
    return ((unsigned long long) (unsigned) __builtin_amdgcn_readlane((int) (unsigned) (v >> 32), 63) << 32) |
           (unsigned) __builtin_amdgcn_readlane((int) (unsigned) v, 63);
#endif
}
MP3MI_DEVFN void l12_update_crc(unsigned data, unsigned length, unsigned *crc)
{ // src/common.c:1309-1323
    unsigned masking = 1u << length;
    while ((masking >>= 1)) {
        const unsigned carry = *crc & 0x8000u;
        *crc <<= 1;
        if (!carry ^ !(data & masking)) *crc ^= 0x8005u;
    }
    *crc &= 0xffffu;
}

// LAYER 1 or 2; NPART = parts of 12 samples per subband and frame (1 or 3)
template <int LAYER>
__global__ void __launch_bounds__(64) k12_alloc(const mp3mi_tables_l12 *__restrict__ T, l12_geom geo,
                                                const l12_stream_cfg *__restrict__ cfg, const double *__restrict__ sbs,
                                                const float *__restrict__ snr, uint8_t *__restrict__ out, size_t out_stride,
                                                uint32_t *__restrict__ out_len, l12_frame_dbg *__restrict__ dbg)
{
    constexpr int NPART = LAYER == 1 ? 1 : 3, NSLOT = 12 * NPART;
    __shared__ alloc12_lds L;
    const int lane = wave_lane();
    const int C = geo.channels, stereo = C;
    const int fl = (int) (blockIdx.x % (unsigned) geo.nf), s = (int) (blockIdx.x / (unsigned) geo.nf);
    const long n = (long) geo.f0 + fl; // frame index in the stream
    const long n_frames_s = geo.n_samples ? ((long) geo.n_samples[s] + geo.spf - 1) / geo.spf : (long) geo.n_frames;
    if (n >= n_frames_s) {
        if (n_frames_s == 0 && n == 0 && lane == 0 && geo.whole_file) { out[(size_t) s * out_stride] = 0; out_len[s] = 1; }
        return;
    }
    const l12_stream_cfg cf = cfg[s];
    const int sblimit = cf.sblimit;
    // Two-channel Layer I at 32 kbps and 44.1 / 48 kHz: header and allocation fields alone (32 + 256 bits, + 16 with -e) are
    // more than the frame's 256 bits.  The reference's budget goes negative, nothing is allocated, no padding is written --
    // and the frame it has written is LONGER than its slot; the next one follows it directly (src/encode.c:997-998,
    // src/musicin.c:657).  Reproduced: such a stream's frames are 36 (38) bytes apart.  (Joint stereo lowers its bound
    // until the fields fit, src/encode.c:907-918; Layer II's smallest frame holds its fields at every rate.)
    const int fields = LAYER == 1 && geo.actual_mode != 1 ? 32 + (geo.crc ? 16 : 0) + 128 * stereo : 0;
    const int frame_bytes = (cf.frame_bits > fields ? cf.frame_bits : fields) / 8;
    const int sb = C == 2 ? lane >> 1 : lane, ch = C == 2 ? lane & 1 : 0;
    const bool act = sb < sblimit && lane < 32 * C; // a (subband, channel) the layer codes
    const int sbc = sb < 32 ? sb : 31;

    for (int i = lane; i < L12_IMG_WORDS; i += 64) L.img[i] = 0;
    L.multiple[lane] = T->multiple[lane];
    if (lane < 18) L.snr[lane] = T->snr[lane];
    if (LAYER == 2)
        for (int i = lane; i < 32 * 16 * 4 / 2; i += 64) ((uint32_t *) L.al)[i] = ((const uint32_t *) T->alloc[cf.table])[i];
    wave_sync();

    // ---- the frame's subband samples of this lane's (subband, channel), src/musicin.c:622-626, 662-666.  k_filter
    // stored them for Layer III's MDCT: odd slots of odd subbands negated (src/mdct.c:57-60) -- undone here, exactly.
    double x[NSLOT];
    {
        const int G1 = geo.n_gran + 1;
        const double *base = sbs + ((size_t) s * G1 * C + (lane < 32 * C ? ch : 0)) * 576 + sbc;
        int sg = geo.slot0 + fl * NSLOT, gi = sg / 18, qq = sg - 18 * gi;
#pragma unroll
        for (int u = 0; u < NSLOT; u++) {
            const double v = base[(size_t) gi * C * 576 + qq * 32];
            x[u] = ((sbc & 1) && (qq & 1)) ? -v : v;
            qq++;
            if (qq == 18) { qq = 0; gi++; }
        }
    }

    // ---- scale factors, src/encode.c:512-561; joint stereo: of the channels' mean as well (:469-494, musicin.c:629-632)
    unsigned scalar[3] = {0, 0, 0}, j_scale[3] = {0, 0, 0};
#pragma unroll
    for (int t = 0; t < NPART; t++) {
        double m = __builtin_fabs(x[12 * t]);
#pragma unroll
        for (int j = 1; j < 12; j++) { const double a = __builtin_fabs(x[12 * t + j]); m = a > m ? a : m; }
        scalar[t] = sb < sblimit ? (unsigned) l12_scale_index(L.multiple, m) : 63u;
    }
    if (geo.actual_mode == 1) {
#pragma unroll
        for (int t = 0; t < NPART; t++) {
            double m = 0.0;
#pragma unroll
            for (int j = 0; j < 12; j++) {
                const double o = __shfl_xor(x[12 * t + j], 1);
                const double a = __builtin_fabs(.5 * ((ch ? o : x[12 * t + j]) + (ch ? x[12 * t + j] : o)));
                m = (j == 0 || a > m) ? a : m;
            }
            j_scale[t] = sb < sblimit ? (unsigned) l12_scale_index(L.multiple, m) : 63u;
        }
    }

    // ---- signal-to-mask ratios of this frame, src/musicin.c:639-643, 681-686 and src/psy.c:391-395
    double smr = 0.0;
    if (lane < 32 * C) {
        const size_t rec = ((size_t) s * geo.np + geo.lb + (size_t) fl * LAYER) * C + ch;
        float v = snr[rec * 32 + sbc];
        if (LAYER == 2) { const float w = snr[(rec + (size_t) C) * 32 + sbc]; v = (v > w) ? v : w; }
        smr = (double) v;
    }

    // ---- Layer II: transmission pattern, src/encode.c:638-691
    unsigned scfsi = 0;
    if (LAYER == 2 && act) {
        const int d0 = (int) (scalar[0] - scalar[1]), d1 = (int) (scalar[1] - scalar[2]);
        const int c0 = d0 <= -3 ? 0 : d0 < 0 ? 1 : d0 == 0 ? 2 : d0 < 3 ? 3 : 4;
        const int c1 = d1 <= -3 ? 0 : d1 < 0 ? 1 : d1 == 0 ? 2 : d1 < 3 ? 3 : 4;
        switch (L12_PATTERN[c0][c1]) {
        case 0: scfsi = 0; break;
        case 1: scfsi = 3; scalar[2] = scalar[1]; break;
        case 2: scfsi = 3; scalar[1] = scalar[2]; break;
        case 3: scfsi = 1; scalar[1] = scalar[0]; break;
        case 4: scfsi = 2; scalar[1] = scalar[2] = scalar[0]; break;
        case 5: scfsi = 2; scalar[0] = scalar[2] = scalar[1]; break;
        case 6: scfsi = 2; scalar[0] = scalar[1] = scalar[2]; break;
        default:
            scfsi = 2;
            if (scalar[0] > scalar[2]) scalar[0] = scalar[2];
            scalar[1] = scalar[2] = scalar[0];
        }
    }
    const int sfs = scfsi == 0 ? 3 : scfsi == 2 ? 1 : 2;                 // sfsPerScfsi, src/encode.c:825
    const int sfs_o = __shfl_xor(sfs, 1);                                // the other channel's (C == 2)
    const double smr_o = __shfl_xor(smr, 1);
    const double *snrt = L.snr;
    const int maxAlloc = LAYER == 2 ? (1 << L.al[sbc][0][1]) - 1 : 15;   // src/encode.c:838

    // ---- joint stereo: how many subbands stay stereo, src/encode.c:882-948
    int mode = geo.actual_mode, mode_ext = 0, jsbound = sblimit;
    if (geo.actual_mode == 1) {
        mode = 0;
        for (int me = 4;;) { // me == 4: plain stereo; then mode_ext 3, 2, 1, 0
            const int jsb = me == 4 ? sblimit : (me == 3 ? 16 : me == 2 ? 12 : me == 1 ? 8 : 4); // js_bound, src/common.c:320-331 (Layers I and II alike)
            int bits = 0;
            if (act && (sb < jsb || ch == 0)) { // *_bits_for_nonoise, src/encode.c:782-860
                if (LAYER == 1) {
                    int k = 0;
                    for (; k < 14; ++k)
                        if ((-smr + snrt[k]) >= 0.0) break;
                    if (stereo == 2 && sb >= jsb)
                        for (; k < 14; ++k)
                            if ((-smr_o + snrt[k]) >= 0.0) break;
                    if (k > 0) bits = (k + 1) * 12 + 6 * ((sb >= jsb) ? stereo : 1);
                } else {
                    int ba = 0;
                    for (; ba < maxAlloc - 1; ++ba)
                        if ((-smr + snrt[L.al[sbc][ba][3] + ((ba > 0) ? 1 : 0)]) >= 0.0) break;
                    if (stereo == 2 && sb >= jsb)
                        for (; ba < maxAlloc - 1; ++ba)
                            if ((-smr_o + snrt[L.al[sbc][ba][3] + ((ba > 0) ? 1 : 0)]) >= 0.0) break;
                    if (ba > 0) {
                        bits = 12 * (L.al[sbc][ba][2] * L.al[sbc][ba][1]) + 2 + 6 * sfs;
                        if (stereo == 2 && sb >= jsb) bits += 2 + 6 * sfs_o;
                    }
                }
            }
            int hdr; // bits of the header and of the allocation fields
            if (LAYER == 1) hdr = 32 + 4 * ((jsb * stereo) + (32 - jsb));
            else {
                const int bb = (act && (sb < jsb || ch == 0)) ? L.al[sbc][0][1] : 0;
                hdr = 32 + (geo.crc ? 16 : 0) + wave_sum_i32(bb);
            }
            const int rq = hdr + wave_sum_i32(bits);
            if (me == 4) {
                if (!(rq > cf.frame_bits)) break;
                mode = 1;
                me = 3;
                continue;
            }
            jsbound = jsb;
            mode_ext = me;
            if (!((rq > cf.frame_bits) && (me > 0))) break;
            --me;
        }
    }

    // ---- *_a_bit_allocation, src/encode.c:974-1172: the band with the smallest mask-to-noise ratio gets the next step,
    // the first in (subband, channel) order among equals -- lane order -- until a step does not fit; then that band is
    // closed and the others go on.  Every lane keeps its ratio as a 64-bit key whose unsigned order is the reverse of the
    // doubles' order (0: not a candidate), what its next step would cost, and the key it would have after that step.
    //
    // Steps are granted in ROUNDS where they can be: a band's ratio only rises with its steps, so every current key above
    // T = the largest key any band would have AFTER its next step belongs to a step that the reference takes before all
    // others (every other step -- a band's later one, or a current one at or below T -- comes after them); if the steps
    // of that set fit together, each of them fits when its turn comes, and they are granted at once.  (T is taken from
    // the keys' upper halves, rounded up: a smaller set, never a wrong one.)  Where the set does not fit -- the end
    // game -- or the bands are coupled (above the joint-stereo bound the allocation applies to both channels), the
    // reference's one-band-at-a-time order is followed literally: one wave maximum per step.
    int ba = 0, used = 0, adb = cf.frame_bits;
    double mnr = snrt[0] - smr;
    {
        const bool code = act && (sb < jsbound || ch == 0); // has an allocation field of its own
        int bbal;
        if (LAYER == 1) bbal = 4 * ((jsbound * stereo) + (32 - jsbound));
        else bbal = wave_sum_i32(code ? (int) L.al[sbc][0][1] : 0);
        adb -= bbal + (geo.crc ? 16 : 0) + 32;
        const int ad = adb;
        const bool coupled = C == 2 && jsbound < (LAYER == 1 ? 32 : sblimit);
        int spent = 0; // bspl + bscf + bsel
        auto entry = [&](int idx) -> uint2 { return *(const uint2 *) L.al[sbc][idx < 16 ? idx : 15]; }; // {steps | bits << 16, group | quant << 16}
        auto bits12 = [](uint2 e) -> int { return 12 * (int) ((e.y & 0xffffu) * (e.x >> 16)); };
        auto keyof = [&](double m, bool open) -> unsigned long long { return (open && (LAYER == 1 || 999999.0 > m)) ? l12_key(m) : 0ull; };
        // the band's NEXT step: its cost (sample bits, and with the first step scale factor and scfsi bits), the ratio
        // after it, and whether it is the band's last (then the band is closed: src/encode.c:1041, 1147)
        int cur12 = 0, need = 0;
        double mnr_nxt = 0.0;
        bool last_nxt = false;
        auto look_ahead = [&]() {
            if (LAYER == 1) {
                int scale = used ? 0 : 6;
                if (sb >= jsbound) scale *= stereo;
                need = (used ? 12 : 24) + scale;
                mnr_nxt = -smr + snrt[ba + 1 < 17 ? ba + 1 : 17];
                last_nxt = ba + 1 == 14;
            } else {
                const uint2 e = entry(ba + 1);
                need = bits12(e);
                if (used) need -= cur12;
                else {
                    need += 2 + 6 * sfs;
                    if (stereo == 2 && sb >= jsbound) need += 2 + 6 * sfs_o;
                }
                mnr_nxt = -smr + snrt[(e.y >> 16) + 1];
                last_nxt = ba + 1 >= maxAlloc;
            }
        };
        auto take_step = [&]() { // the band gets its next step
            ba++;
            used = last_nxt ? 2 : 1;
            mnr = mnr_nxt;
            if (LAYER == 2) cur12 = bits12(entry(ba));
        };
        look_ahead();
        unsigned long long key = keyof(mnr, act), nkey = keyof(mnr_nxt, act && !last_nxt);
        for (;;) {
            unsigned long long k = key;
            if (LAYER == 1) { // src/encode.c:1012: small starts at mnr[0][0] + 1, whatever state that band is in
                const double lim = wave_bcast_f64(mnr, 0) + 1;
                if (!(lim > mnr)) k = 0ull;
            }
            if (!coupled) { // a round
                if (!__ballot(k != 0ull)) break;
                const unsigned hi = wave_max_u32(k != 0ull ? (unsigned) (nkey >> 32) : 0u); // (keys of numbers are far below all ones)
                const bool in = k > (((unsigned long long) hi << 32) | 0xffffffffull);
                if (__ballot(in)) {
                    const int cost = wave_sum_i32(in ? need : 0);
                    if (ad >= spent + cost) {
                        spent += cost;
                        if (in) {
                            take_step();
                            if (used != 2) look_ahead();
                            key = keyof(mnr, used != 2);
                            nkey = keyof(mnr_nxt, used != 2 && !last_nxt);
                        }
                        continue;
                    }
                }
            }
            const unsigned long long small = l12_wave_max_u64(k); // (the key of the smallest ratio)
            if (small == 0ull) break;
            const unsigned long long tie = __ballot(k == small);
            const int win = __ffsll((long long) tie) - 1;
            const int wsb = C == 2 ? win >> 1 : win;
            const bool me_win = lane == win, me_oth = C == 2 && lane == (win ^ 1) && wsb >= jsbound;
            const int wneed = wave_readlane_i32(need, win);
            const bool fits = ad >= spent + wneed;
            if (fits) spent += wneed;
            else {
                // The band does not get its step and is closed.  When what is left is less than ANY open band's next
                // step, every remaining round of the reference's loop does the same -- picks a band, finds no room,
                // closes it -- and changes nothing that is looked at afterwards (the allocation, the bits left): stop.
                const int mn = wave_min_i32((key != 0ull && !me_win) ? need : 0x7fffffff);
                if (ad - spent < mn) break;
            }
            if (me_win) {
                if (fits) take_step();
                else used = 2;
                if (used != 2) look_ahead();
                key = keyof(mnr, used != 2);
                nkey = keyof(mnr_nxt, used != 2 && !last_nxt);
            }
            if (C == 2 && wsb >= jsbound) { // above the joint-stereo bound the allocation applies to both channels
                const int wba = wave_readlane_i32(ba, win), wused = wave_readlane_i32(used, win);
                if (me_oth) {
                    ba = wba;
                    used = wused;
                    if (LAYER == 1) mnr = -smr + snrt[ba];
                    else {
                        const uint2 e_cur = entry(ba);
                        mnr = -smr + snrt[(e_cur.y >> 16) + 1];
                        cur12 = bits12(e_cur);
                    }
                    if (used != 2) look_ahead();
                    key = keyof(mnr, used != 2);
                    nkey = keyof(mnr_nxt, used != 2 && !last_nxt);
                }
            }
        }
        adb = ad - spent;
    }
    if (!act) ba = 0;

    // ---- CRC over header and allocation (Layer II: and scfsi), src/common.c:1251-1307
    unsigned crc = 0;
    if (geo.crc || dbg) {
        if (lane < 32 * C) { L.ba[ch][sbc] = ba; L.sf[ch][sbc] = (int) scfsi; }
        wave_sync();
    }
    if (geo.crc) {
        if (lane == 0) {
            unsigned c = 0xffff;
            l12_update_crc((unsigned) cf.bitrate_index, 4, &c);
            l12_update_crc((unsigned) geo.rate_idx, 2, &c);
            l12_update_crc(0, 1, &c);
            l12_update_crc(0, 1, &c);
            l12_update_crc((unsigned) mode, 2, &c);
            l12_update_crc((unsigned) mode_ext, 2, &c);
            l12_update_crc((unsigned) (geo.hdr_flags >> 3) & 1u, 1, &c);
            l12_update_crc((unsigned) (geo.hdr_flags >> 2) & 1u, 1, &c);
            l12_update_crc((unsigned) geo.hdr_flags & 3u, 2, &c);
            const int nsb = LAYER == 1 ? 32 : sblimit;
            for (int i = 0; i < nsb; i++)
                for (int k = 0; k < ((i < jsbound) ? stereo : 1); k++)
                    l12_update_crc((unsigned) L.ba[k][i], LAYER == 1 ? 4u : (unsigned) L.al[i][0][1], &c);
            if (LAYER == 2)
                for (int i = 0; i < sblimit; i++)
                    for (int k = 0; k < stereo; k++)
                        if (L.ba[k][i]) l12_update_crc((unsigned) L.sf[k][i], 2, &c);
            L.crc = c;
        }
        wave_sync();
        crc = L.crc;
    }

    if (dbg) { // the seams of oracle/stage_dump_l12.h
        l12_frame_dbg *d = dbg + (size_t) s * geo.nf + fl;
        if (lane < 32 * C) {
            d->ltmin[ch][sbc] = smr;
            for (int t = 0; t < 3; t++) d->scalar[ch][t][sbc] = t < NPART ? (int) scalar[t] : 0;
            d->scfsi[ch][sbc] = (int) scfsi;
            d->bit_alloc[ch][sbc] = ba;
            if (ch == 0)
                for (int t = 0; t < 3; t++) d->j_scale[t][sbc] = (t < NPART && geo.actual_mode == 1) ? (int) j_scale[t] : 0;
        }
        if (C == 1 && lane >= 32) {
            const int b2 = lane - 32;
            d->ltmin[1][b2] = 0.0;
            for (int t = 0; t < 3; t++) d->scalar[1][t][b2] = 0;
            d->scfsi[1][b2] = 0;
            d->bit_alloc[1][b2] = 0;
        }
        if (lane == 0) {
            d->mode = mode; d->mode_ext = mode_ext; d->jsbound = jsbound; d->sblimit = sblimit;
            d->adb_left = adb; d->crc = (int) crc;
        }
    }

    // ---- the frame's bits, src/encode.c:418-437, 708-748, 1328-1416
    const bool own = act && (sb < jsbound || ch == 0); // writes allocation field and samples (above the bound: channel 0's lane)
    int pos = 0, tot;
    if (lane == 0) {
        unsigned h = 0xfffu << 20;           // syncword
        h |= 1u << 19;                       // ID: MPEG-1
        h |= (unsigned) (4 - LAYER) << 17;
        h |= (geo.crc ? 0u : 1u) << 16;      // protection bit: set = no CRC
        h |= (unsigned) cf.bitrate_index << 12;
        h |= (unsigned) geo.rate_idx << 10;  // (padding 0, private 0: src/musicin.c:566-581)
        h |= (unsigned) mode << 6;
        h |= (unsigned) mode_ext << 4;
        h |= (unsigned) geo.hdr_flags & 15u;
        L.img[0] = h;
        if (geo.crc) L.img[1] = crc << 16;
    }
    wave_sync(); // (plain stores before the atomic ORs)
    pos = 32 + (geo.crc ? 16 : 0);
    {   // allocation fields
        const int nb = LAYER == 1 ? ((lane < 32 * C && (sb < jsbound || ch == 0)) ? 4 : 0) : (own ? (int) L.al[sbc][0][1] : 0);
        const int ofs = l12_scan(nb, &tot);
        if (nb) l12_put(L.img, pos + ofs, (unsigned) ba, nb);
        pos += tot;
    }
    if (LAYER == 2) { // scfsi, then the scale factors it selects
        const int nb = (act && ba) ? 2 : 0;
        int ofs = l12_scan(nb, &tot);
        if (nb) l12_put(L.img, pos + ofs, scfsi, 2);
        pos += tot;
        const int ns = (act && ba) ? 6 * sfs : 0;
        ofs = l12_scan(ns, &tot);
        if (ns) {
            int p = pos + ofs;
            l12_put(L.img, p, scalar[0], 6); p += 6;
            if (scfsi == 0) { l12_put(L.img, p, scalar[1], 6); p += 6; }
            if (scfsi != 2) l12_put(L.img, p, scalar[2], 6);
        }
        pos += tot;
    } else {
        const int ns = (lane < 32 * C && ba) ? 6 : 0;
        const int ofs = l12_scan(ns, &tot);
        if (ns) l12_put(L.img, pos + ofs, scalar[0], 6);
        pos += tot;
    }
    {   // samples: quantised (src/encode.c:1207-1325) and written group by group
        const bool joint = stereo == 2 && sb >= jsbound;
        const bool wr = own && ba > 0;
        int nbits, grp = 3, steps = 0, qn = 0, qnt;
        if (LAYER == 1) { nbits = ba + 1; qn = ba; qnt = ba > 0 ? ba - 1 : 0; }
        else {
            const uint16_t *e = L.al[sbc][ba];
            steps = e[0]; nbits = e[1]; grp = e[2]; qnt = e[3];
            while ((1L << qn) < (long) steps) qn++;
            qn--;
            if (qn < 0) qn = 0; // (a lane without allocation: nothing of it is written)
        }
        const double qa = T->qa[qnt], qb = T->qb[qnt];
        const int per = LAYER == 1 ? nbits : (grp == 3 ? 3 * nbits : nbits); // bits of this lane per row of the sample section
        int rowbits;
        const int ofs = l12_scan(wr ? per : 0, &rowbits);
#pragma unroll
        for (int t = 0; t < NPART; t++) {
            const double div = L.multiple[joint ? j_scale[t] : scalar[t]];
#pragma unroll
            for (int j0 = 0; j0 < 12; j0 += (LAYER == 1 ? 1 : 3)) {
                unsigned q3[3];
#pragma unroll
                for (int jj = 0; jj < (LAYER == 1 ? 1 : 3); jj++) {
                    double v = x[12 * t + j0 + jj];
                    if (geo.actual_mode == 1) { // (the shuffle runs on every lane, the value is used above the bound)
                        const double o = __shfl_xor(v, 1);
                        if (joint) v = .5 * ((ch ? o : v) + (ch ? v : o));
                    }
                    double d = v / div;
                    d = d * qa + qb;
                    unsigned sig = 1;
                    if (!(d >= 0)) { sig = 0; d += 1.0; }
                    unsigned qv = (unsigned) (d * (double) (1L << qn));
                    if (sig) qv |= 1u << qn;
                    q3[jj] = qv;
                }
                if (wr) {
                    const int row = LAYER == 1 ? j0 : (t * 4 + j0 / 3);
                    const int p = pos + row * rowbits + ofs;
                    if (LAYER == 1) l12_put(L.img, p, q3[0], nbits);
                    else if (grp == 3) {
                        l12_put(L.img, p, q3[0], nbits);
                        l12_put(L.img, p + nbits, q3[1], nbits);
                        l12_put(L.img, p + 2 * nbits, q3[2], nbits);
                    } else l12_put(L.img, p, q3[0] + q3[1] * (unsigned) steps + q3[2] * (unsigned) steps * (unsigned) steps, nbits);
                }
            }
        }
    }
    wave_sync();
    // ---- the frame goes to byte n * frame_bytes of the stream's row; what is left of it is zero (src/musicin.c:657, 703).
    // The file carries one byte beyond its last frame (close_bit_stream_w, src/common.c:843-868): the last frame's wavefront adds it.
    {
        uint8_t *o = out + (size_t) s * out_stride + (size_t) n * (size_t) frame_bytes;
        for (int i = lane; i < frame_bytes; i += 64) o[i] = (uint8_t) (L.img[i >> 2] >> (24 - 8 * (i & 3)));
        if (n == n_frames_s - 1 && lane == 0) {
            if (geo.whole_file) o[frame_bytes] = 0;
            out_len[s] = (uint32_t) (n_frames_s * frame_bytes + (geo.whole_file ? 1 : 0));
        }
    }
}

void mp3mi_launch_l12_alloc(const mp3mi_tables_l12 *T, const l12_geom &g, const l12_stream_cfg *cfg, const double *sbs,
                            const float *snr, uint8_t *out, size_t out_stride, uint32_t *out_len, l12_frame_dbg *dbg, hipStream_t st)
{
    const unsigned grid = (unsigned) ((size_t) g.n_streams * g.nf);
    if (g.layer == 1) hipLaunchKernelGGL(k12_alloc<1>, dim3(grid), dim3(64), 0, st, T, g, cfg, sbs, snr, out, out_stride, out_len, dbg);
    else hipLaunchKernelGGL(k12_alloc<2>, dim3(grid), dim3(64), 0, st, T, g, cfg, sbs, snr, out, out_stride, out_len, dbg);
}

// ---- streaming: what a stream carries from call to call is PCM history only (l12_dev.h).  The last L12_PCM_HIST samples
// before the next call's first one -- this call's tail, preceded by the old history's when the call was shorter --
// for the FFT windows, and the last MP3MI_PCM_HIST of them in k_filter's layout.  Out of place: in and out are two buffers.
__global__ void __launch_bounds__(256) k12_hist_save(l12_geom geo, const int16_t *__restrict__ pcm, const int16_t *__restrict__ hist_in,
                                                     int16_t *__restrict__ hist_out, const int16_t *__restrict__ fb_in, int16_t *__restrict__ fb_out)
{
    const int s = (int) blockIdx.x, C = geo.channels;
    const long n_call = (long) geo.n_frames * geo.spf;
    const int16_t *src = pcm + (size_t) s * (size_t) n_call * C;
    for (int pass = 0; pass < 2; pass++) {
        const int H = pass ? MP3MI_PCM_HIST : L12_PCM_HIST;
        const int16_t *in = (pass ? fb_in : hist_in) + (size_t) s * H * C;
        int16_t *outp = (pass ? fb_out : hist_out) + (size_t) s * H * C;
        for (int i = (int) threadIdx.x; i < H * C; i += 256) {
            const long t = (long) (i / C) + n_call - H; // time of the sample, from the call's first
            outp[i] = t >= 0 ? src[t * C + i % C] : in[(t + H) * C + i % C];
        }
    }
}
void mp3mi_launch_l12_hist_save(const l12_geom &g, const int16_t *pcm, const int16_t *hist_in, int16_t *hist_out, const int16_t *fb_in,
                                int16_t *fb_out, hipStream_t st)
{
    hipLaunchKernelGGL(k12_hist_save, dim3((unsigned) g.n_streams), dim3(256), 0, st, g, pcm, hist_in, hist_out, fb_in, fb_out);
}
// close_bit_stream_w: the byte under construction is written too (src/common.c:843-868) -- frames end on byte boundaries, so it is 0
__global__ void __launch_bounds__(64) k12_flush(int n_streams, uint8_t *__restrict__ out, size_t out_stride, uint32_t *__restrict__ out_len)
{
    const int s = (int) (blockIdx.x * 64 + threadIdx.x);
    if (s < n_streams) { out[(size_t) s * out_stride] = 0; out_len[s] = 1; }
}
void mp3mi_launch_l12_flush(int n_streams, uint8_t *out, size_t out_stride, uint32_t *out_len, hipStream_t st)
{
    hipLaunchKernelGGL(k12_flush, dim3((unsigned) ((n_streams + 63) / 64)), dim3(64), 0, st, n_streams, out, out_stride, out_len);
}

ULP_CENSUS_ACCESSOR(mp3mi_debug_ulp_census_l12)
