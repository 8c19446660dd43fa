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

// One wavefront transforms all C channels of a granule: every butterfly record is fetched once
// and applied to the C long (then 3C short) arrays, which also gives each lane C independent
// dependency chains.  es/ps reuse the long arrays' space once the long spectrum is consumed.
template <int C> struct fft_lds {
    union {
        float xl[C][1024];
        struct { float es[C][3][129]; float ps[C][3][52]; } sp;
    };
    float xs[C][3][256];
};

template <int TYPE, int NARR, int STRIDE>
MP3MI_DEVFN void fft_apply(float *x, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3)
{
    int a, b, c = 0, d = 0;
    if (TYPE == FOP_ROT) {
        a = (int) (w0 & 0xffffu); b = (int) (w0 >> 16);
    } else {
        a = (int) (w0 & 1023u); b = (int) ((w0 >> 10) & 1023u); c = (int) (w0 >> 20); d = c + b - a;
    }
#pragma unroll
    for (int arr = 0; arr < NARR; arr++) {
        float *v = x + arr * STRIDE;
        if (TYPE == FOP_ADDSUB) {
            const float t = v[a] + v[b];
            v[b] = v[a] - v[b];
            v[a] = t;
        } else if (TYPE == FOP_NEG) {
            v[a] = -v[a];
        } else if (TYPE == FOP_CROSS) {
            const float r1 = v[a], r2 = v[b], i1 = v[c], i2 = v[d];
            v[c] = i1 - r2;
            v[b] = r1 - i2;
            v[a] = r1 + i2;
            v[d] = i1 + r2;
        } else if (TYPE == FOP_ROT) {
            const float cn = __builtin_bit_cast(float, w1), spc = __builtin_bit_cast(float, w2), smc = __builtin_bit_cast(float, w3);
            const float r1 = v[a], i1 = v[b];
            const float t2 = cn * (r1 + i1);
            const float t1 = spc * r1 + t2;
            v[a] = smc * i1 + t2;
            v[b] = t1;
        } else if (TYPE == FOP_SQ1) {
            const float r1 = v[a], i1 = v[b];
            v[a] = (float) (R_SQHALF * (double) (r1 + i1));
            v[b] = (float) (R_SQHALF * (double) (i1 - r1));
        } else if (TYPE == FOP_SQ2) {
            const float r2 = v[a], i2 = v[b];
            v[a] = (float) (R_SQHALF * (double) (i2 - r2));
            v[b] = (float) (-R_SQHALF * (double) (r2 + i2));
        } else if (TYPE == FOP_SWAPNN) {
            const float t = v[a];
            v[a] = -v[b];
            v[b] = -t;
        } else if (TYPE == FOP_SWAPN) {
            const float t = v[a];
            v[a] = -v[b];
            v[b] = t;
        } else { // FOP_SWAP
            const float t = v[a];
            v[a] = v[b];
            v[b] = t;
        }
    }
}

// one segment of independent butterflies; the record of the next round is requested before the
// current one is applied so that its latency hides behind the LDS work
template <int TYPE, int NARR, int STRIDE>
MP3MI_DEVFN void fft_segment(float *x, const uint32_t *gops, const mp3mi_fftop *rops, int start, int count)
{
    const int lane = wave_lane();
    if (TYPE == FOP_ROT) {
        const uint4 *rp = (const uint4 *) (rops + start);
        uint4 cur = rp[lane < count ? lane : 0];
        for (int e = lane; e < count; e += 64) {
            const uint4 nxt = rp[e + 64 < count ? e + 64 : 0];
            fft_apply<TYPE, NARR, STRIDE>(x, cur.x, cur.y, cur.z, cur.w);
            cur = nxt;
        }
    } else {
        const uint32_t *gp = gops + start;
        uint32_t cur = gp[lane < count ? lane : 0];
        for (int e = lane; e < count; e += 64) {
            const uint32_t nxt = gp[e + 64 < count ? e + 64 : 0];
            fft_apply<TYPE, NARR, STRIDE>(x, cur, 0, 0, 0);
            cur = nxt;
        }
    }
}

template <int NARR, int STRIDE>
MP3MI_DEVFN void fft_run(float *x, const mp3mi_fftseg *segs, int nseg, const uint32_t *gops, const mp3mi_fftop *rops)
{
    for (int sidx = 0; sidx < nseg; sidx++) {
        const int type = segs[sidx].type, start = segs[sidx].start, count = segs[sidx].count;
        switch (type) {
        case FOP_ADDSUB: fft_segment<FOP_ADDSUB, NARR, STRIDE>(x, gops, rops, start, count); break;
        case FOP_NEG: fft_segment<FOP_NEG, NARR, STRIDE>(x, gops, rops, start, count); break;
        case FOP_CROSS: fft_segment<FOP_CROSS, NARR, STRIDE>(x, gops, rops, start, count); break;
        case FOP_ROT: fft_segment<FOP_ROT, NARR, STRIDE>(x, gops, rops, start, count); break;
        case FOP_SQ1: fft_segment<FOP_SQ1, NARR, STRIDE>(x, gops, rops, start, count); break;
        case FOP_SQ2: fft_segment<FOP_SQ2, NARR, STRIDE>(x, gops, rops, start, count); break;
        case FOP_SWAPNN: fft_segment<FOP_SWAPNN, NARR, STRIDE>(x, gops, rops, start, count); break;
        case FOP_SWAPN: fft_segment<FOP_SWAPN, NARR, STRIDE>(x, gops, rops, start, count); break;
        default: fft_segment<FOP_SWAP, NARR, STRIDE>(x, gops, rops, start, count); break;
        }
        if (segs[sidx].barrier) __syncthreads();
    }
}

// energy and phase of bin i of an N-point transform held as x (src/subs.c:53-123)
MP3MI_DEVFN void fft_bin(const float *x, int N, int i, bool want_phi, float *energy, float *phi)
{
    if (i == 0 || i == N / 2) {
        *energy = x[i] * x[i];
        *phi = want_phi ? (float) dm_atan2(0.0, (double) x[i]) : 0.0f;
        return;
    }
    const float re = x[i], im = x[N - i];
    float e = re * re + im * im;
    if ((double) e < 0.0005) {
        *energy = (float) 0.0005;
        *phi = 0.0f;
    } else {
        *energy = e;
        *phi = want_phi ? (float) dm_atan2(-(double) im, (double) re) : 0.0f;
    }
}

template <int C>
__global__ void __launch_bounds__(64) k_fft(const mp3mi_tables *__restrict__ T, mp3mi_geom geo,
                                            const int16_t *__restrict__ pcm_all, float *__restrict__ energy_l,
                                            float *__restrict__ energy_s, double *__restrict__ cw_mid,
                                            float *__restrict__ hist6)
{
    __shared__ fft_lds<C> L;
    const int lane = wave_lane();
    const int G = geo.n_gran;
    const int gl = (int) blockIdx.x % G, s = (int) blockIdx.x / G;
    const size_t rec0 = ((size_t) s * G + gl) * C;
    const long gabs = (long) geo.g0 + gl;
    const long n_per_ch = (long) geo.n_frames * 1152;
    const int16_t *pcm = pcm_all + (size_t) s * (size_t) n_per_ch * (size_t) C;
    const long t0 = 576 * gabs - 768; // time of savebuf[0]  (src/l3psy.c:477-481)

    for (int j = lane; j < 1024; j += 64) {
        const long t = t0 + j;
        const bool in = t >= 0 && t < n_per_ch;
        const float w = T->window[j];
        int v[C];
        if (C == 2) {
            const uint32_t both = in ? ((const uint32_t *) pcm)[t] : 0u;
            v[0] = (int) (int16_t) (both & 0xffffu);
            v[C - 1] = (int) (int16_t) (both >> 16);
        } else {
            v[0] = in ? (int) pcm[t] : 0;
        }
#pragma unroll
        for (int c = 0; c < C; c++) {
            L.xl[c][j] = w * (float) v[c];                        // src/l3psy.c:485
            // short windows: samples 256 + 128 sb + jj, sb < 3 (src/l3psy.c:520-523); the middle 128
            // samples of every window are also the first 128 of the next
            if (j >= 256) {
                const int q = j - 256, sb = q >> 7, jj = q & 127;
                if (sb < 3) L.xs[c][sb][jj] = T->window_s[jj] * (float) v[c];
                if (sb >= 1 && sb < 4) L.xs[c][sb - 1][128 + jj] = T->window_s[128 + jj] * (float) v[c];
            }
        }
    }
    __syncthreads();

    fft_run<C, 1024>(&L.xl[0][0], T->seg_l, T->n_seg_l, T->gops_l, T->rops_l);

    for (int c = 0; c < C; c++)
        for (int i = lane; i < MP3MI_HBLK; i += 64) {
            float e, p;
            fft_bin(L.xl[c], 1024, i, i < 6, &e, &p);
            energy_l[(rec0 + c) * MP3MI_HBLK + i] = e;
            if (i < 6) {
                hist6[(rec0 + c) * 12 + i] = (float) __builtin_sqrt((double) e); // r, src/l3psy.c:500
                hist6[(rec0 + c) * 12 + 6 + i] = p;
            }
        }
    __syncthreads(); // xl is dead from here on: es/ps take its place

    fft_run<3 * C, 256>(&L.xs[0][0][0], T->seg_s, T->n_seg_s, T->gops_s, T->rops_s);

    for (int i = lane; i < C * 3 * MP3MI_HBLK_S; i += 64) {
        const int csb = i / MP3MI_HBLK_S, k = i % MP3MI_HBLK_S, c = csb / 3, sb = csb % 3;
        float e, p;
        fft_bin(L.xs[c][sb], 256, k, false, &e, &p);
        L.sp.es[c][sb][k] = e;
        energy_s[(rec0 + c) * (3 * MP3MI_HBLK_S) + sb * MP3MI_HBLK_S + k] = e;
    }
    for (int i = lane; i < C * 150; i += 64) { // phases of short lines 2..51 (src/l3psy.c:531-549 reads these only)
        const int csb = i / 50, k = 2 + i % 50, c = csb / 3, sb = csb % 3;
        float e, p;
        fft_bin(L.xs[c][sb], 256, k, true, &e, &p);
        L.sp.ps[c][sb][k] = p;
    }
    __syncthreads();

    for (int i = lane; i < C * 50; i += 64) { // unpredictability of lines 6+4n..9+4n from the three short FFTs (src/l3psy.c:531-549)
        const int c = i / 50, n = i % 50, k = n + 2;
        const double r_prime = 2.0 * __builtin_sqrt((double) L.sp.es[c][0][k]) - __builtin_sqrt((double) L.sp.es[c][2][k]);
        const double phi_prime = 2.0 * (double) L.sp.ps[c][0][k] - (double) L.sp.ps[c][2][k];
        const double r2 = __builtin_sqrt((double) L.sp.es[c][1][k]);
        const double phi2 = (double) L.sp.ps[c][1][k];
        double s2, c2, sp, cp;
        dm_sincos(phi2, &s2, &c2);
        dm_sincos(phi_prime, &sp, &cp);
        const double t1 = r2 * c2 - r_prime * cp;
        const double t2 = r2 * s2 - r_prime * sp;
        const double t3 = r2 + __builtin_fabs(r_prime);
        double cw = 0.0;
        if (t3 != 0.0) cw = __builtin_sqrt(t1 * t1 + t2 * t2) / t3;
        cw_mid[(rec0 + c) * 50 + n] = cw;
    }
}

void mp3mi_launch_fft(const mp3mi_tables *T, const mp3mi_geom &g, const int16_t *pcm, float *energy_l,
                      float *energy_s, double *cw_mid, float *hist6, hipStream_t st)
{
    const unsigned grid = (unsigned) (g.n_streams * g.n_gran);
    if (g.channels == 2)
        hipLaunchKernelGGL(k_fft<2>, dim3(grid), dim3(64), 0, st, T, g, pcm, energy_l, energy_s, cw_mid, hist6);
    else
        hipLaunchKernelGGL(k_fft<1>, dim3(grid), dim3(64), 0, st, T, g, pcm, energy_l, energy_s, cw_mid, hist6);
}
