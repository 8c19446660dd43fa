// Host-side construction of the read-only table block (mp3mi_tables).
//
// Every value that the reference computes once at start-up with libm is computed here the
// same way with the host's libm (identical glibc on the GPU box), so the kernels never
// evaluate an init-time transcendental themselves:
//   Hann windows            src/l3psy.c:194-195      spreading matrix   src/l3psy.c:818-848
//   FFT twiddles            src/subs.c:278-286,452-457
//   analysis filter matrix  src/encode.c:331-345     MDCT windows/cos   src/mdct.c:129-171
//   alias butterflies       src/mdct.c:37-45         quantiser tables   src/pow_nint.c:13-20,
//                                                                       src/loop.c:1017-1021
// plus the FFT butterfly program: the reference's recursive split-radix real FFT
// (src/subs.c:185-362, 412-523) flattened into barrier-separated segments of independent
// butterflies so that 64 lanes can execute it with the identical arithmetic DAG.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
#include "mp3mi_host.h"
#include "mp3mi_tables_gen.h"

#define R_PI 3.14159265358979
#define R_LN_TO_LOG10 0.2302585093
#define R_TWOPI 6.28318530717958647692

static const int SFB_L[3][23] = {
    {0,4,8,12,16,20,24,30,36,44,52,62,74,90,110,134,162,196,238,288,342,418,576},
    {0,4,8,12,16,20,24,30,36,42,50,60,72,88,106,128,156,190,230,276,330,384,576},
    {0,4,8,12,16,20,24,30,36,44,54,66,82,102,126,156,194,240,296,364,448,550,576}};
static const int SFB_S[3][14] = {
    {0,4,8,12,16,22,30,40,52,66,84,106,136,192},
    {0,4,8,12,16,22,28,38,50,64,80,100,126,192},
    {0,4,8,12,16,22,30,42,58,78,104,138,180,192}};
static const double ALIAS_C[8] = {-0.6,-0.535,-0.33,-0.185,-0.095,-0.041,-0.0142,-0.0037};

namespace {

struct RawOp {
    int phase, type;
    unsigned a, b, c, d;
    float f0, f1, f2;
};

struct Twiddle { std::vector<float> t; int nel; };

Twiddle make_twiddle(int logm, bool three)
{
    int m = 1 << logm, m4 = m / 4, m8 = m / 8, nel = m4 - 2, e = 0;
    Twiddle tw;
    tw.nel = nel;
    tw.t.assign((size_t) (three ? 6 : 3) * (nel > 0 ? nel : 1), 0.0f);
    for (int n = 1; n < m4; n++) {
        if (n == m8) continue;
        float ang = (float) (n * R_TWOPI / m);
        float c = (float) cos((double) ang), s = (float) sin((double) ang); /* C semantics: double libm on the float angle */
        tw.t[e] = c;
        tw.t[nel + e] = -(s + c);
        tw.t[2 * nel + e] = s - c;
        if (three) {
            ang = (float) (3 * n * R_TWOPI / m);
            c = (float) cos((double) ang);
            s = (float) sin((double) ang);
            tw.t[3 * nel + e] = c;
            tw.t[4 * nel + e] = -(s + c);
            tw.t[5 * nel + e] = s - c;
        }
        e++;
    }
    return tw;
}

struct FftGen {
    std::vector<RawOp> ops;
    Twiddle tw_rs[11], tw_sr[11];
    int post1, post2, brphase;

    void op(int phase, int type, unsigned a, unsigned b = 0, unsigned c = 0, unsigned d = 0,
            float f0 = 0, float f1 = 0, float f2 = 0)
    {
        RawOp o = {phase, type, a, b, c, d, f0, f1, f2};
        ops.push_back(o);
    }

    void cplx(int xr, int xi, int logm, int rank)
    {
        if (logm <= 0) return;
        if (logm == 1) {
            op(3 * rank, FOP_ADDSUB, xr, xr + 1);
            op(3 * rank, FOP_ADDSUB, xi, xi + 1);
            return;
        }
        int m = 1 << logm, m2 = m / 2, m4 = m2 / 2, m8 = m4 / 2;
        for (int n = 0; n < m2; n++) {
            op(3 * rank, FOP_ADDSUB, xr + n, xr + n + m2);
            op(3 * rank, FOP_ADDSUB, xi + n, xi + n + m2);
        }
        for (int n = 0; n < m4; n++)
            op(3 * rank + 1, FOP_CROSS, xr + m2 + n, xr + m2 + n + m4, xi + m2 + n, xi + m2 + n + m4);
        const Twiddle &tw = tw_sr[logm];
        for (int n = 1, e = 0; n < m4; n++) {
            unsigned r1 = xr + m2 + n, r2 = r1 + m4, i1 = xi + m2 + n, i2 = i1 + m4;
            if (n == m8) {
                op(3 * rank + 2, FOP_SQ1, r1, i1);
                op(3 * rank + 2, FOP_SQ2, r2, i2);
            } else {
                op(3 * rank + 2, FOP_ROT, r1, i1, 0, 0, tw.t[e], tw.t[tw.nel + e], tw.t[2 * tw.nel + e]);
                op(3 * rank + 2, FOP_ROT, r2, i2, 0, 0, tw.t[3 * tw.nel + e], tw.t[4 * tw.nel + e], tw.t[5 * tw.nel + e]);
                e++;
            }
        }
        cplx(xr, xi, logm - 1, rank + 1);
        cplx(xr + m2, xi + m2, logm - 2, rank + 1);
        cplx(xr + 3 * (m / 4), xi + 3 * (m / 4), logm - 2, rank + 1);
    }

    void real(int o, int logm, int rank)
    {
        if (logm <= 0) return;
        if (logm == 1) {
            op(3 * rank, FOP_ADDSUB, o, o + 1);
            return;
        }
        int m = 1 << logm, m2 = m / 2, m4 = m2 / 2, m8 = m4 / 2;
        for (int n = 0; n < m2; n++) op(3 * rank, FOP_ADDSUB, o + n, o + n + m2);
        for (int n = 0; n < m4; n++) op(3 * rank + 1, FOP_NEG, o + m2 + m4 + n);
        const Twiddle &tw = tw_rs[logm];
        for (int n = 1, e = 0; n < m4; n++) {
            unsigned r1 = o + m2 + n, i1 = r1 + m4;
            if (n == m8) op(3 * rank + 2, FOP_SQ1, r1, i1);
            else {
                op(3 * rank + 2, FOP_ROT, r1, i1, 0, 0, tw.t[e], tw.t[tw.nel + e], tw.t[2 * tw.nel + e]);
                e++;
            }
        }
        real(o, logm - 1, rank + 1);
        cplx(o + m2, o + 3 * (m / 4), logm - 2, rank + 1);
        for (int n = 0; n < m8; n++) op(post1, FOP_SWAPNN, o + m2 + m4 + n, o + m - 1 - n);
        for (int n = 0; n < m8; n++) op(post2, FOP_SWAPN, o + m2 + 1 + 2 * n, o + m - 2 - 2 * n);
        if (logm == 2) op(post1, FOP_NEG, o + 3);
    }

    void build(int logN, uint32_t *gops, int max_gops, mp3mi_fftop *rops, int max_rops, int32_t *segw, int32_t *n_seg)
    {
        ops.clear();
        post1 = 3 * logN + 1;
        post2 = post1 + 1;
        brphase = post2 + 1;
        real(0, logN, 0);
        const int N = 1 << logN;
        for (int i = 0; i < N; i++) { /* bit reversal */
            int j = 0;
            for (int b = 0; b < logN; b++)
                if (i & (1 << b)) j |= 1 << (logN - 1 - b);
            if (j > i) op(brphase, FOP_SWAP, i, j);
        }
        std::stable_sort(ops.begin(), ops.end(), [](const RawOp &x, const RawOp &y) {
            return x.phase != y.phase ? x.phase < y.phase : x.type < y.type;
        });
        /* the arrays live in LDS with the low five address bits XORed with the next five (MP3MI_FFT_SWZ):
           the small transforms' operands, which in natural order share a few banks, then spread over all */
        for (size_t i = 0; i < ops.size(); i++) {
            RawOp &o = ops[i];
            o.a = MP3MI_FFT_SWZ(o.a);
            if (o.type != FOP_NEG) o.b = MP3MI_FFT_SWZ(o.b);
            if (o.type == FOP_CROSS) { o.c = MP3MI_FFT_SWZ(o.c); o.d = MP3MI_FFT_SWZ(o.d); }
        }
        /* The butterflies of a segment are independent, so their order is free: place them so that the
           32 lanes of a half wave (the unit the LDS serves a 4-byte access in) address 32 different
           banks with every operand.  Greedy: first the butterflies that fit without any collision, then
           -- a half wave left partly idle would cost a whole round, a collision costs a cycle -- the
           ones that collide least. */
        for (size_t s0 = 0; s0 < ops.size();) {
            size_t s1 = s0;
            while (s1 < ops.size() && ops[s1].phase == ops[s0].phase && ops[s1].type == ops[s0].type) s1++;
            std::vector<RawOp> rest(ops.begin() + s0, ops.begin() + s1), placed;
            const int type = ops[s0].type;
            const int nopnd = (type == FOP_NEG) ? 1 : (type == FOP_CROSS ? 4 : 2);
            while (!rest.empty()) {
                int used[4][32];
                memset(used, 0, sizeof(used));
                std::vector<RawOp> group;
                auto cost = [&](const RawOp &o) {
                    const unsigned ad[4] = {o.a, o.b, o.c, o.d};
                    int c = 0;
                    for (int k = 0; k < nopnd; k++) c += used[k][ad[k] & 31];
                    return c;
                };
                auto take = [&](size_t idx, bool flip) {
                    RawOp o = rest[idx];
                    if (flip) std::swap(o.a, o.b);
                    const unsigned ad[4] = {o.a, o.b, o.c, o.d};
                    for (int k = 0; k < nopnd; k++) used[k][ad[k] & 31]++;
                    group.push_back(o);
                    rest.erase(rest.begin() + (long) idx);
                };
                for (size_t i = 0; i < rest.size() && group.size() < 32;) { /* collision-free candidates, in order */
                    RawOp f = rest[i];
                    std::swap(f.a, f.b);
                    if (cost(rest[i]) == 0) take(i, false);
                    else if (type == FOP_SWAP && cost(f) == 0) take(i, true); /* an exchange is symmetric */
                    else i++;
                }
                while (group.size() < 32 && !rest.empty()) { /* fill up with the least harmful */
                    size_t best = 0;
                    int bc = 1 << 30;
                    for (size_t i = 0; i < rest.size(); i++) {
                        const int c = cost(rest[i]);
                        if (c < bc) { bc = c; best = i; }
                    }
                    take(best, false);
                }
                placed.insert(placed.end(), group.begin(), group.end());
            }
            std::copy(placed.begin(), placed.end(), ops.begin() + (long) s0);
            s0 = s1;
        }
        /* Every segment is padded to whole rounds of 64 records (bit 31 = idle lane): round t of the
           g (r) stream is records [64 t, 64 t + 64), one per lane.  The kernel keeps both streams in
           LDS and walks them in order, so a segment is just (type, rounds, barrier). */
        struct Seg { int type, count, barrier; };
        Seg segs[MP3MI_MAX_FFT_SEGS];
        int ns = 0, ng = 0, nr = 0;
        auto pad = [&](int type) {
            if (type == FOP_ROT) {
                while (nr % 64) { mp3mi_fftop w = {{0x80000000u, 0, 0, 0}}; if (nr >= max_rops) abort(); rops[nr++] = w; }
            } else {
                while (ng % 64) { if (ng >= max_gops) abort(); gops[ng++] = 0x80000000u; }
            }
        };
        std::vector<uint32_t> second; /* FOP_CROSS: the round's second words (c | d << 10) follow its first words */
        auto flush_cross = [&]() {
            if (second.empty()) return;
            pad(FOP_CROSS);
            for (size_t k = 0; k < second.size(); k++) { if (ng >= max_gops) abort(); gops[ng++] = second[k]; }
            pad(FOP_CROSS);
            second.clear();
        };
        for (size_t i = 0; i < ops.size(); i++) {
            const RawOp &o = ops[i];
            const bool rot = o.type == FOP_ROT;
            if (i == 0 || o.phase != ops[i - 1].phase || o.type != ops[i - 1].type) {
                if (ns >= MP3MI_MAX_FFT_SEGS) { fprintf(stderr, "mp3mi: too many fft segments\n"); abort(); }
                if (ns > 0) {
                    segs[ns - 1].barrier = (o.phase != ops[i - 1].phase);
                    flush_cross();
                    pad(segs[ns - 1].type);
                }
                segs[ns].type = o.type;
                segs[ns].count = 0;
                segs[ns].barrier = 1;
                ns++;
            }
            segs[ns - 1].count++;
            if (rot) {
                if (nr >= max_rops) { fprintf(stderr, "mp3mi: fft program too large\n"); abort(); }
                mp3mi_fftop w;
                w.w[0] = o.a | (o.b << 16);
                memcpy(&w.w[1], &o.f0, 4);
                memcpy(&w.w[2], &o.f1, 4);
                memcpy(&w.w[3], &o.f2, 4);
                rops[nr++] = w;
            } else {
                if (ng >= max_gops) { fprintf(stderr, "mp3mi: fft program too large\n"); abort(); }
                gops[ng++] = o.a | (o.b << 10);
                if (o.type == FOP_CROSS) {
                    second.push_back(o.c | (o.d << 10));
                    if (second.size() == 64) flush_cross();
                }
            }
        }
        flush_cross();
        pad(segs[ns - 1].type);
        if (ng != max_gops || nr != max_rops) { fprintf(stderr, "mp3mi: fft program size %d/%d, expected %d/%d\n", ng, nr, max_gops, max_rops); abort(); }
        for (int k = 0; k < ns; k++) segw[k] = segs[k].type | (((segs[k].count + 63) / 64) << 8) | (segs[k].barrier << 16);
        *n_seg = ns;
    }
};

} // namespace

extern "C" int mp3mi_build_tables(mp3mi_tables *T, int rate_idx)
{
    const int ri = rate_idx;
    if (ri < 0 || ri > 2) return -1;
    memset(T, 0, sizeof(*T));
    T->rate_idx = ri;
    for (int i = 0; i < 23; i++) T->sfb_l[i] = SFB_L[ri][i];
    for (int i = 0; i < 14; i++) T->sfb_s[i] = SFB_S[ri][i];
    for (int sfb = 0; sfb < 22; sfb++)
        for (int l = SFB_L[ri][sfb]; l < SFB_L[ri][sfb + 1]; l++) T->sfb_of_line_l[l] = (uint8_t) sfb;
    for (int sfb = 0; sfb < 13; sfb++)
        for (int l = SFB_S[ri][sfb]; l < SFB_S[ri][sfb + 1]; l++)
            for (int w = 0; w < 3; w++) T->sfb_of_line_s[l * 3 + w] = (uint8_t) (sfb * 3 + w);

    // noise-sum jobs: split every band into parts of at most `target` elements, smallest target that fits 64 jobs
    for (int t = 0; t < 2; t++) {
        const int nb = t ? 36 : 21;
        for (int target = 4; target <= 192; target++) {
            int jobs = 0;
            for (int b = 0; b < nb; b++) {
                const int w = t ? SFB_S[ri][b / 3 + 1] - SFB_S[ri][b / 3] : SFB_L[ri][b + 1] - SFB_L[ri][b];
                jobs += (w + target - 1) / target;
            }
            if (jobs > 64) continue;
            int j = 0;
            for (int b = 0; b < nb; b++) {
                const int e0 = t ? SFB_S[ri][b / 3] : SFB_L[ri][b];
                const int w = (t ? SFB_S[ri][b / 3 + 1] : SFB_L[ri][b + 1]) - e0;
                const int parts = (w + target - 1) / target;
                T->nj_job0[t][b] = (uint8_t) j;
                T->nj_njobs[t][b] = (uint8_t) parts;
                for (int q = 0, off = 0; q < parts; q++) {
                    const int cnt = (w - off + (parts - q) - 1) / (parts - q); // even split of what is left
                    T->nj_first[t][j] = (int16_t) (t ? (e0 + off) * 3 + b % 3 : e0 + off);
                    T->nj_count[t][j] = (uint8_t) cnt;
                    off += cnt;
                    j++;
                }
            }
            break;
        }
    }

    for (unsigned i = 0; i < 1024; i++) T->window[i] = (float) (0.5 * (1 - cos(2.0 * R_PI * (i - 0.5) / 1024)));
    for (unsigned i = 0; i < 256; i++) T->window_s[i] = (float) (0.5 * (1 - cos(2.0 * R_PI * (i - 0.5) / 256)));

    const int cb_l = T_PL_COUNT[ri], cb_s = T_PS_COUNT[ri];
    double bval_l[MP3MI_CBANDS];
    int k2 = 0;
    for (int i = 0; i < cb_l; i++) {
        T->numlines_pe[i] = T_PL_NUMLINES[ri][i];
        T->part_l_start[i] = k2;
        k2 += T_PL_NUMLINES[ri][i];
        T->minval[i] = T_PL_MINVAL[ri][i];
        T->qthr_l[i] = T_PL_QTHR[ri][i];
        T->norm_l[i] = T_PL_NORM[ri][i];
        bval_l[i] = T_PL_BVAL[ri][i];
    }
    for (int i = cb_l; i <= MP3MI_CBANDS; i++) T->part_l_start[i] = k2;
    /* lines the table does not cover keep the reference's zero-initialised partition index,
       i.e. they are summed into partition 0 after its own lines (src/l3psy.c:131, 808-809) */
    T->part_l_covered = k2;
    if (k2 > MP3MI_HBLK) return -2;
    for (int i = 0; i < cb_l; i++)
        for (int j = 0; j < cb_l; j++) {
            double tempx, x, tempy, temp;
            if (j >= i) tempx = (bval_l[i] - bval_l[j]) * 3.0;
            else tempx = (bval_l[i] - bval_l[j]) * 1.5;
            if (tempx >= 0.5 && tempx <= 2.5) { temp = tempx - 0.5; x = 8.0 * (temp * temp - 2.0 * temp); }
            else x = 0.0;
            tempx += 0.474;
            tempy = 15.811389 + 7.5 * tempx - 17.5 * sqrt(1.0 + tempx * tempx);
            T->s3_l[i][j] = (tempy <= -60.0) ? 0.0 : exp((x + tempy) * R_LN_TO_LOG10);
        }
    k2 = 0;
    for (int i = 0; i < cb_s; i++) {
        T->numlines_pe[i] = T_PS_NUMLINES[ri][i]; /* the short table overwrites the long one: src/l3psy.c:868 */
        T->part_s_start[i] = k2;
        k2 += T_PS_NUMLINES[ri][i];
        T->qthr_s[i] = T_PS_QTHR[ri][i];
        T->exp_snr_s[i] = exp((double) T_PS_SNR[ri][i] * R_LN_TO_LOG10);
    }
    for (int i = cb_s; i <= MP3MI_CBANDS_S; i++) T->part_s_start[i] = k2;
    T->part_s_covered = k2;
    if (k2 > MP3MI_HBLK_S) return -2;
    /* entries of qthr_s/exp_snr_s beyond cb_s: the reference loops b < 42 over statics that
       were never written there: qthr_s = 0, SNR_s = 0 -> exp(0) = 1 */
    for (int i = cb_s; i < MP3MI_CBANDS_S; i++) T->exp_snr_s[i] = exp(0.0 * R_LN_TO_LOG10);
    for (int i = 0; i < MP3MI_CBANDS; i++) { T->s3_lo[i] = T_S3_LO[i]; T->s3_hi[i] = T_S3_HI[i]; }
    for (int i = 0; i < 21; i++) {
        T->bu_l[i] = T_SL_BU[ri][i]; T->bo_l[i] = T_SL_BO[ri][i];
        T->w1_l[i] = T_SL_W1[ri][i]; T->w2_l[i] = T_SL_W2[ri][i];
    }
    for (int i = 0; i < 12; i++) {
        T->bu_s[i] = T_SS_BU[ri][i]; T->bo_s[i] = T_SS_BO[ri][i];
        T->w1_s[i] = T_SS_W1[ri][i]; T->w2_s[i] = T_SS_W2[ri][i];
    }

    {
        FftGen *g = new FftGen();
        for (int i = 4; i <= 10; i++) { g->tw_rs[i] = make_twiddle(i, false); g->tw_sr[i] = make_twiddle(i, true); }
        g->build(10, T->gops_l, 64 * MP3MI_FFT_GROUNDS_L, T->rops_l, 64 * MP3MI_FFT_RROUNDS_L, T->seg_l, &T->n_seg_l);
        g->build(8, T->gops_s, 64 * MP3MI_FFT_GROUNDS_S, T->rops_s, 64 * MP3MI_FFT_RROUNDS_S, T->seg_s, &T->n_seg_s);
        delete g;
    }

    for (int i = 0; i < 512; i++) T->enwindow[i] = T_ENWINDOW[i];
    for (int i = 0; i < 32; i++) {
        double row[64];
        for (int k = 0; k < 64; k++) {
            double f = 1e9 * cos((double) ((2 * i + 1) * (16 - k) * R_PI / 64));
            if (f >= 0) modf(f + 0.5, &f);
            else modf(f - 0.5, &f);
            row[k] = f * 1e-9;
        }
        for (int j = 0; j < 16; j++) T->filt[i][j] = row[j];
        for (int j = 0; j < 15; j++) T->filt[i][16 + j] = row[33 + j];
        T->filt[i][31] = 0.0;
    }

    double (*win)[36] = T->mdct_win;
    for (int i = 0; i < 36; i++) win[0][i] = sin(R_PI / 36 * (i + 0.5));
    for (int i = 0; i < 18; i++) win[1][i] = sin(R_PI / 36 * (i + 0.5));
    for (int i = 18; i < 24; i++) win[1][i] = 1.0;
    for (int i = 24; i < 30; i++) win[1][i] = sin(R_PI / 12 * (i + 0.5 - 18));
    for (int i = 30; i < 36; i++) win[1][i] = 0.0;
    for (int i = 0; i < 6; i++) win[3][i] = 0.0;
    for (int i = 6; i < 12; i++) win[3][i] = sin(R_PI / 12 * (i + 0.5 - 6));
    for (int i = 12; i < 18; i++) win[3][i] = 1.0;
    for (int i = 18; i < 36; i++) win[3][i] = sin(R_PI / 36 * (i + 0.5));
    for (int i = 0; i < 12; i++) win[2][i] = sin(R_PI / 12 * (i + 0.5));
    for (int i = 12; i < 36; i++) win[2][i] = 0.0;
    {
        int N = 12;
        for (int m = 0; m < N / 2; m++)
            for (int k = 0; k < N; k++)
                T->cos_s[m][k] = cos((R_PI / (2 * N)) * (2 * k + 1 + N / 2) * (2 * m + 1)) / (N / 4);
        N = 36;
        for (int m = 0; m < N / 2; m++)
            for (int k = 0; k < N; k++)
                T->cos_l[m][k] = cos((R_PI / (2 * N)) * (2 * k + 1 + N / 2) * (2 * m + 1)) / (N / 4);
    }
    for (int k = 0; k < 8; k++) {
        double sq = sqrt(1.0 + ALIAS_C[k] * ALIAS_C[k]);
        T->ca[k] = ALIAS_C[k] / sq;
        T->cs[k] = 1.0 / sq;
    }
    /* Long-block MDCT in shared-subexpression form.  Every bracketed operand group of
       src/mdct.c:205-508 is one of 26 per-band values V: d1[j] = fin[j]-fin[17-j], s2[j] =
       fin[18+j]+fin[35-j] (j<9), six 6-operand groups and two 18-operand groups -- or the exact
       negation of one (rounding is symmetric, so negating every operand negates the sum bit for
       bit).  A term is then V[idx] * (+-cos_l[m][k]); rows keep the reference's term order. */
    {
        int n_g = 0, n_h = 0;
        for (int m = 0; m < 18; m++) {
            int nt = 0;
            for (int t = T_MDCTL_ROW[m]; t < T_MDCTL_ROW[m + 1]; t++, nt++) {
                const int o0 = T_MDCTL_TERM_OP[t], o1 = T_MDCTL_TERM_OP[t + 1], nops = o1 - o0;
                const unsigned char *ops = &T_MDCTL_OPS[o0];
                double coef = T->cos_l[m][T_MDCTL_TERM_K[t] & 0x7f];
                if (T_MDCTL_TERM_K[t] & 0x80) coef = -coef;
                int vidx = -1, sgn = 0; /* term value = sgn * V[vidx] */
                if (nops == 2) {
                    const int a = ops[0] & 0x3f, b = ops[1] & 0x3f, na = ops[0] >> 7, nb = ops[1] >> 7;
                    if (a < 9 && b == 17 - a && na != nb) { vidx = a; sgn = na ? -1 : 1; }                 /* +-(fin[a]-fin[17-a]) */
                    else if (a >= 18 && a < 27 && b == 53 - a && na == nb) { vidx = 9 + (a - 18); sgn = na ? -1 : 1; } /* +-(fin[a]+fin[35-j]) */
                } else {
                    uint8_t (*canon)[18] = (nops == 6) ? T->mdct_g_ops : T->mdct_h_ops;
                    int &ncanon = (nops == 6) ? n_g : n_h;
                    const int maxc = (nops == 6) ? 6 : 2, base = (nops == 6) ? 18 : 24;
                    if (nops != 6 && nops != 18) return -4;
                    for (int c = 0; c < ncanon && vidx < 0; c++) {
                        bool same = true, neg = true;
                        for (int o = 0; o < nops; o++) {
                            if ((canon[c][o] & 0x3f) != (ops[o] & 0x3f)) { same = neg = false; break; }
                            if ((canon[c][o] >> 7) != (ops[o] >> 7)) same = false; else neg = false;
                        }
                        if (same) { vidx = base + c; sgn = 1; }
                        else if (neg) { vidx = base + c; sgn = -1; }
                    }
                    if (vidx < 0) {
                        if (ncanon >= maxc) return -4;
                        for (int o = 0; o < nops; o++) canon[ncanon][o] = ops[o];
                        vidx = base + ncanon++;
                        sgn = 1;
                    }
                }
                if (vidx < 0 || nt >= 18) return -4;
                T->mdct_vidx[m][nt] = (uint8_t) vidx;
                T->mdct_vcoef[m][nt] = (sgn < 0) ? -coef : coef;
            }
            T->mdct_nterm[m] = (uint8_t) nt;
            for (; nt < 18; nt++) { T->mdct_vidx[m][nt] = 0; T->mdct_vcoef[m][nt] = 0.0; }
        }
        if (n_g != 6 || n_h != 2) return -4;
    }

    T->pow_nint_tab[0] = 0.0;
    for (int i = 1; i < 2048; i++) T->pow_nint_tab[i] = pow((double) i - 0.4054, 4.0 / 3.0);
    T->pow_nint_tab[2048] = HUGE_VAL;
    for (int i = 0; i < MP3MI_POW43_N; i++) T->pow43[i] = pow((double) i, 4.0 / 3.0);
    for (int i = 0; i < MP3MI_STEP_N; i++) T->step[i] = pow(2.0, (double) (MP3MI_STEP_MIN + i) * 0.25);
    for (int n = 0; n < 4; n++) {
        T->pretab_xr[n] = pow(sqrt(2.), (double) n);
        T->pretab_xmin[n] = pow(sqrt(2.), 2.0 * (double) n);
    }
    T->sqrt2 = sqrt(2.0);
    T->log2 = log(2.0);

    size_t n_ht = sizeof(T_HT_PACKED) / sizeof(T_HT_PACKED[0]);
    if (n_ht > 1440) return -3;
    for (size_t i = 0; i < n_ht; i++) {
        T->ht_len[i] = (uint8_t) (T_HT_PACKED[i] & 0xff);
        T->ht_code[i] = T_HT_PACKED[i] >> 8;
    }
    {
        static const int groups[9][5] = { /* offset, cells, table a, table b, table c (0 = none) */
            {0, 4, 1, 0, 0}, {4, 9, 2, 3, 0}, {13, 16, 5, 6, 0}, {29, 36, 7, 8, 9}, {65, 64, 10, 11, 12},
            {129, 256, 13, 15, 0}, {385, 256, 15, 24, 0}, {641, 256, 16, 24, 0}, {897, 16, 32, 33, 0}};
        memset(T->glut, 0, sizeof(T->glut));
        for (int gi = 0; gi < 9; gi++)
            for (int c = 0; c < groups[gi][1]; c++) {
                /* sign bits of the cell (src/loop.c:172-225, 1560-1584): one per non-zero value */
                const int ylen = T_HT_YLEN[groups[gi][2]];
                const int sg = (gi == 8) ? __builtin_popcount((unsigned) c) : (c / ylen != 0) + (c % ylen != 0);
                unsigned e = 0;
                for (int f = 0; f < 3; f++) {
                    const int t = groups[gi][2 + f];
                    if (!t) continue;
                    const unsigned len = (T_HT_PACKED[T_HT_OFF[t] + c] & 0xff) + (unsigned) sg;
                    if (len > 31) return -5;
                    e |= len << (5 * f);
                }
                T->glut[groups[gi][0] + c] = (uint16_t) e;
            }
    }
    for (int i = 0; i < 34; i++) {
        T->ht_off[i] = T_HT_OFF[i];
        T->ht_xlen[i] = T_HT_XLEN[i];
        T->ht_ylen[i] = T_HT_YLEN[i];
        T->ht_linbits[i] = T_HT_LINBITS[i];
        T->ht_linmax[i] = T_HT_LINMAX[i];
    }
    return 0;
}
