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

struct fft_lds {
    float xl[1024];
    float xs[3][256];
    float es[3][129];
    float ps[3][52];
};

template <int NARR, int STRIDE>
MP3MI_DEVFN void fft_run(float *x, const mp3mi_fftseg *segs, int nseg, const mp3mi_fftop *ops)
{
    const int lane = wave_lane();
    for (int sidx = 0; sidx < nseg; sidx++) {
        const int type = segs[sidx].type, start = segs[sidx].start, count = segs[sidx].count;
        for (int e = lane; e < count; e += 64) {
            const mp3mi_fftop op = ops[start + e];
            const int a = (int) (op.w[0] & 0xffffu), b = (int) (op.w[0] >> 16);
            for (int arr = 0; arr < NARR; arr++) {
                float *v = x + arr * STRIDE;
                switch (type) {
                case FOP_ADDSUB: {
                    float t = v[a] + v[b];
                    v[b] = v[a] - v[b];
                    v[a] = t;
                } break;
                case FOP_NEG: v[a] = -v[a]; break;
                case FOP_CROSS: {
                    const int c = (int) (op.w[1] & 0xffffu), d = (int) (op.w[1] >> 16);
                    float r1 = v[a], r2 = v[b], i1 = v[c], i2 = v[d];
                    v[c] = i1 - r2;
                    v[b] = r1 - i2;
                    v[a] = r1 + i2;
                    v[d] = i1 + r2;
                } break;
                case FOP_ROT: {
                    const float cn = __builtin_bit_cast(float, op.w[1]), spc = __builtin_bit_cast(float, op.w[2]),
                                smc = __builtin_bit_cast(float, op.w[3]);
                    float r1 = v[a], i1 = v[b];
                    float t2 = cn * (r1 + i1);
                    float t1 = spc * r1 + t2;
                    v[a] = smc * i1 + t2;
                    v[b] = t1;
                } break;
                case FOP_SQ1: {
                    float r1 = v[a], i1 = v[b];
                    v[a] = (float) (R_SQHALF * (double) (r1 + i1));
                    v[b] = (float) (R_SQHALF * (double) (i1 - r1));
                } break;
                case FOP_SQ2: {
                    float r2 = v[a], i2 = v[b];
                    v[a] = (float) (R_SQHALF * (double) (i2 - r2));
                    v[b] = (float) (-R_SQHALF * (double) (r2 + i2));
                } break;
                case FOP_SWAPNN: {
                    float t = v[a];
                    v[a] = -v[b];
                    v[b] = -t;
                } break;
                case FOP_SWAPN: {
                    float t = v[a];
                    v[a] = -v[b];
                    v[b] = t;
                } break;
                default: { // FOP_SWAP
                    float t = v[a];
                    v[a] = v[b];
                    v[b] = t;
                } break;
                }
            }
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

__global__ void __launch_bounds__(64) k_fft(const mp3mi_tables *__restrict__ T, mp3mi_geom geo,
                                            const int16_t *__restrict__ pcm_all, float *__restrict__ energy_l,
                                            float *__restrict__ energy_s, double *__restrict__ cw_mid,
                                            float *__restrict__ hist6)
{
    __shared__ fft_lds L;
    const int lane = wave_lane();
    const int C = geo.channels, G = geo.n_gran;
    int bid = (int) blockIdx.x;
    const int ch = bid % C; bid /= C;
    const int gl = bid % G;
    const int s = bid / G;
    const size_t rec = ((size_t) s * G + gl) * C + ch;
    const long gabs = (long) geo.g0 + gl;
    const long n_per_ch = (long) geo.n_frames * 1152;
    const int16_t *pcm = pcm_all + (size_t) s * (size_t) n_per_ch * (size_t) C;
    const long t0 = 576 * gabs - 768; // time of savebuf[0]  (src/l3psy.c:477-481)

    for (int j = lane; j < 1024; j += 64) {
        long t = t0 + j;
        int16_t v = (t >= 0 && t < n_per_ch) ? pcm[t * C + ch] : (int16_t) 0;
        L.xl[j] = T->window[j] * (float) v;           // src/l3psy.c:485
    }
    for (int j = lane; j < 768; j += 64) {
        const int sb = j >> 8, jj = j & 255;
        long t = t0 + 128 * (2 + sb) + jj;
        int16_t v = (t >= 0 && t < n_per_ch) ? pcm[t * C + ch] : (int16_t) 0;
        L.xs[sb][jj] = T->window_s[jj] * (float) v;   // src/l3psy.c:520-523
    }
    __syncthreads();

    fft_run<1, 0>(L.xl, T->seg_l, T->n_seg_l, T->ops_l);
    fft_run<3, 256>(&L.xs[0][0], T->seg_s, T->n_seg_s, T->ops_s);

    for (int i = lane; i < MP3MI_HBLK; i += 64) {
        float e, p;
        fft_bin(L.xl, 1024, i, i < 6, &e, &p);
        energy_l[rec * MP3MI_HBLK + i] = e;
        if (i < 6) {
            hist6[rec * 12 + i] = (float) __builtin_sqrt((double) e); // r, src/l3psy.c:500
            hist6[rec * 12 + 6 + i] = p;
        }
    }
    for (int i = lane; i < 3 * MP3MI_HBLK_S; i += 64) {
        const int sb = i / MP3MI_HBLK_S, k = i % MP3MI_HBLK_S;
        float e, p;
        fft_bin(L.xs[sb], 256, k, k >= 2 && k < 52, &e, &p);
        L.es[sb][k] = e;
        if (k < 52) L.ps[sb][k] = p;
        energy_s[rec * (3 * MP3MI_HBLK_S) + i] = e;
    }
    __syncthreads();

    if (lane < 50) { // unpredictability of lines 6+4n..9+4n from the three short FFTs (src/l3psy.c:531-549)
        const int k = lane + 2;
        const double r_prime = 2.0 * __builtin_sqrt((double) L.es[0][k]) - __builtin_sqrt((double) L.es[2][k]);
        const double phi_prime = 2.0 * (double) L.ps[0][k] - (double) L.ps[2][k];
        const double r2 = __builtin_sqrt((double) L.es[1][k]);
        const double phi2 = (double) L.ps[1][k];
        double s2, c2, sp, cp;
        dm_sincos(phi2, &s2, &c2);
        dm_sincos(phi_prime, &sp, &cp);
        const double t1 = r2 * c2 - r_prime * cp;
        const double t2 = r2 * s2 - r_prime * sp;
        const double t3 = r2 + __builtin_fabs(r_prime);
        double cw = 0.0;
        if (t3 != 0.0) cw = __builtin_sqrt(t1 * t1 + t2 * t2) / t3;
        cw_mid[rec * 50 + lane] = cw;
    }
}

void mp3mi_launch_fft(const mp3mi_tables *T, const mp3mi_geom &g, const int16_t *pcm, float *energy_l,
                      float *energy_s, double *cw_mid, float *hist6, hipStream_t st)
{
    const unsigned grid = (unsigned) (g.n_streams * g.n_gran * g.channels);
    hipLaunchKernelGGL(k_fft, dim3(grid), dim3(64), 0, st, T, g, pcm, energy_l, energy_s, cw_mid, hist6);
}
