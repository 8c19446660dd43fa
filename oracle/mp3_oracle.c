/* TEST INFRASTRUCTURE -- CPU restatement of the reference's Layer III hot path.
 *
 * Plain scalar C with glibc libm, explicit per-stream state (the reference keeps
 * its state in function statics), one function per reference function; every
 * function cites the reference file:line it follows.  See mp3_oracle.h for who
 * may use this and for the parity status (PINNED against oracle/_ref).
 *
 * Build: gcc -O2 -ffp-contract=off (contraction changes the output bits).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <assert.h>
#include "mp3_oracle.h"
#include "oracle_tables_gen.h"

/* ---- constants restated verbatim ---- */
#define R_PI 3.14159265358979          /* src/common.h:200, not M_PI */
#define R_LN_TO_LOG10 0.2302585093     /* src/common.h:204 */
#define R_TWOPI 6.28318530717958647692 /* src/subs.c:26 */
#define R_SQHALF 0.707106781186547524401 /* src/subs.c:27 */
#define CBANDS 63
#define CBANDS_S 42
#define HBLK 513
#define HBLK_S 129

enum { BT_NORM = 0, BT_START = 1, BT_SHORT = 2, BT_STOP = 3 };

/* Table B.8 scalefactor bands, rows = MPEG-1 sampling_frequency code (src/loop.c:80-91) */
static const int SFB_L[3][23] = {
    {0,4,8,12,16,20,24,30,36,44,52,62,74,90,110,134,162,196,238,288,342,418,576},
    {0,4,8,12,16,20,24,30,36,42,50,60,72,88,106,128,156,190,230,276,330,384,576},
    {0,4,8,12,16,20,24,30,36,44,54,66,82,102,126,156,194,240,296,364,448,550,576}};
static const int SFB_S[3][14] = {
    {0,4,8,12,16,22,30,40,52,66,84,106,136,192},
    {0,4,8,12,16,22,28,38,50,64,80,100,126,192},
    {0,4,8,12,16,22,30,42,58,78,104,138,180,192}};
static const int PRETAB[21] = {0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,3,2}; /* src/loop.c:150 */
static const int SCFSI_BAND_L[5] = {0, 6, 11, 16, 21};                      /* src/loop.c:157 */
static const int SLEN1[16] = {0,0,0,0,3,1,1,1,2,2,2,3,3,3,4,4};             /* src/loop.c:740 */
static const int SLEN2[16] = {0,1,2,3,0,1,2,3,1,2,3,1,2,3,2,3};
static const int SUBDV[23][2] = {{0,0},{0,0},{0,0},{0,0},{0,0},{0,1},{1,1},{1,1},{1,2},{2,2},{2,3},
    {2,3},{3,4},{3,4},{3,4},{4,5},{4,5},{4,6},{5,6},{5,6},{5,7},{6,7},{6,7}}; /* src/loop.c:1596 */
static const int BITRATES[15] = {0,32,40,48,56,64,80,96,112,128,160,192,224,256,320}; /* src/common.c:124 */
static const double ALIAS_C[8] = {-0.6,-0.535,-0.33,-0.185,-0.095,-0.041,-0.0142,-0.0037}; /* src/mdct.c:23 */

typedef struct {
    unsigned part2_3_length, big_values, count1, global_gain, scalefac_compress;
    unsigned window_switching_flag, block_type, mixed_block_flag;
    unsigned table_select[3];
    int subblock_gain[3];
    unsigned region0_count, region1_count, preflag, scalefac_scale, count1table_select;
    unsigned part2_length, sfb_lmax, sfb_smax, address1, address2, address3;
    double quantizerStepSize;
} gr_info_t;

typedef struct {
    int main_data_begin;
    unsigned private_bits;
    int resvDrain;
    unsigned scfsi[2][4];
    gr_info_t gr[2][2];
} side_info_t;

typedef struct {
    /* init-time tables, each computed the way the reference computes it */
    float window[1024], window_s[256];                      /* src/l3psy.c:194-195 */
    int numlines[CBANDS], partition_l[HBLK], partition_s[HBLK_S];
    double minval[CBANDS], qthr_l[CBANDS], norm_l[CBANDS];
    double qthr_s[CBANDS_S], SNR_s[CBANDS_S];
    double s3_l[CBANDS][CBANDS];
    int bu_l[21], bo_l[21], bu_s[12], bo_s[12];
    double w1_l[21], w2_l[21], w1_s[12], w2_s[12];
    float *tw_rs[11], *tw_sr[11];                           /* FFT twiddles, index = logm */
    double filt[32][64];                                    /* src/encode.c:331-345 */
    double win[4][36], cos_s[6][12], cos_l[18][36];         /* src/mdct.c:129-171 */
    double ca[8], cs[8];                                    /* src/mdct.c:37-45 */
    double pow_nint_tab[4096];                              /* src/pow_nint.c:13-20 */
    double pow43[8300];                                     /* src/loop.c:1017-1021 and direct pow() */
} tables_t;

#define SI_MAX_BYTES 40
typedef struct {
    int frameLength, SILength;
    uint8_t bytes[SI_MAX_BYTES];
} si_entry_t;

struct mp3o_stream {
    int rate_idx, rate_hz, kbps, bitrate_index, channels, mode;
    int crc, copyright, original, emphasis; /* the driver's -e -c -o -d (src/musicin.c:263-275) */
    int ref_abort;                          /* MP3O_ABORT_*: the reference would have died here (see mp3_oracle.h) */
    int bitsPerFrame, mean_bits;
    tables_t *T;
    /* psy model state (function statics of src/l3psy.c:57-131 and the caller's sam[]) */
    short savebuf[2][1344];
    float r[2][2][HBLK], phi_sav[2][2][HBLK];
    int age_new, age_old, age_oldest;
    double nb_1[2][CBANDS], nb_2[2][CBANDS];
    double ratio[2][21], ratio_s[2][12][3];
    int blocktype_old[2];
    /* filterbank ring (src/encode.c:292-296) */
    double fb_x[2][512];
    int fb_off[2];
    /* previous granule's subband samples = l3_sb_sample[ch][0] (src/mdct.c:99-102) */
    double sb[2][3][18][32];
    /* calc_scfsi statics (src/loop.c:618-621) */
    int sc_en_tot[2][2], sc_en[2][2][21], sc_xm[2][2][21], sc_xrmax[2][2];
    /* reservoir (src/reservoir.c:36-37) */
    int ResvSize, ResvMax;
    /* caller-held frame data that persists across frames (statics in main) */
    side_info_t side;
    int l3_enc[2][2][576];
    int scalefac_l[2][2][22], scalefac_s[2][2][13][3];
    /* formatter (src/formatBitstream.c:25-28, 278-279) */
    int BitCount, ThisFrameSize, BitsRemaining;
    si_entry_t *queue;
    int q_head, q_len, q_cap;
    /* bit writer (src/common.c:1134-1161): MSB first, linear */
    uint8_t *out;
    size_t out_bits, out_cap;
    int closed;
    size_t out_len;
    fft_seam_t *fft_seam; /* where psy_granule leaves the transforms' outputs of its next call (NULL: nowhere) */
    long fft_seam_left;   /* records the buffer still takes */
};

/* ------------------------------------------------------------------------- */
/* bit writer                                                                */
/* ------------------------------------------------------------------------- */
static void put_bits(mp3o_stream *s, unsigned val, int n)
{ /* src/common.c:1134-1161: the low n bits of val, most significant first */
    int j;
    if (s->out_bits + 64 > s->out_cap * 8) {
        size_t ncap = s->out_cap ? s->out_cap * 2 : 65536;
        s->out = (uint8_t *) realloc(s->out, ncap);
        memset(s->out + s->out_cap, 0, ncap - s->out_cap);
        s->out_cap = ncap;
    }
    for (j = n - 1; j >= 0; j--) {
        if ((val >> j) & 1u)
            s->out[s->out_bits >> 3] |= (uint8_t) (0x80u >> (s->out_bits & 7));
        s->out_bits++;
    }
}

/* ------------------------------------------------------------------------- */
/* table construction                                                        */
/* ------------------------------------------------------------------------- */
static float *make_twiddles(int logm, int complex3)
{ /* src/subs.c:452-457 (rsrec) and src/subs.c:278-286 (srrec) */
    int m = 1 << logm, m4 = m / 4, m8 = m / 8, nel = m4 - 2, n, e = 0;
    float *t = (float *) calloc((size_t) (complex3 ? 6 : 3) * (nel > 0 ? nel : 1), sizeof(float));
    for (n = 1; n < m4; n++) {
        float ang, c, sn;
        if (n == m8) continue;
        ang = (float) (n * R_TWOPI / m);
        c = (float) cos(ang);
        sn = (float) sin(ang);
        t[e] = c;
        t[nel + e] = -(sn + c);
        t[2 * nel + e] = sn - c;
        if (complex3) {
            ang = (float) (3 * n * R_TWOPI / m);
            c = (float) cos(ang);
            sn = (float) sin(ang);
            t[3 * nel + e] = c;
            t[4 * nel + e] = -(sn + c);
            t[5 * nel + e] = sn - c;
        }
        e++;
    }
    return t;
}

static tables_t *make_tables(int ri)
{
    tables_t *T = (tables_t *) calloc(1, sizeof(tables_t));
    int i, j, k, k2, m, N, cb_l = T_PL_COUNT[ri], cb_s = T_PS_COUNT[ri];
    double bval_l[CBANDS], bval_s[CBANDS];

    /* Hann windows, src/l3psy.c:194-195 (i is unsigned there; i-0.5 is evaluated in double) */
    for (i = 0; i < 1024; i++) T->window[i] = (float) (0.5 * (1 - cos(2.0 * R_PI * (i - 0.5) / 1024)));
    for (i = 0; i < 256; i++) T->window_s[i] = (float) (0.5 * (1 - cos(2.0 * R_PI * (i - 0.5) / 256)));

    /* L3para_read, src/l3psy.c:770-994 */
    for (i = 0, k2 = 0; i < cb_l; i++) {
        T->numlines[i] = T_PL_NUMLINES[ri][i];
        T->minval[i] = T_PL_MINVAL[ri][i];
        T->qthr_l[i] = T_PL_QTHR[ri][i];
        T->norm_l[i] = T_PL_NORM[ri][i];
        bval_l[i] = T_PL_BVAL[ri][i];
        for (k = 0; k < T->numlines[i]; k++) T->partition_l[k2++] = i;
    }
    assert(k2 <= HBLK);
    for (i = 0; i < cb_l; i++)
        for (j = 0; j < cb_l; j++) {
            double tempx, x, tempy, temp;
            if (j >= i) tempx = (bval_l[i] - bval_l[j]) * 3.0;
            else tempx = (bval_l[i] - bval_l[j]) * 1.5;
            if (tempx >= 0.5 && tempx <= 2.5) {
                temp = tempx - 0.5;
                x = 8.0 * (temp * temp - 2.0 * temp);
            } else x = 0.0;
            tempx += 0.474;
            tempy = 15.811389 + 7.5 * tempx - 17.5 * sqrt(1.0 + tempx * tempx);
            if (tempy <= -60.0) T->s3_l[i][j] = 0.0;
            else T->s3_l[i][j] = exp((x + tempy) * R_LN_TO_LOG10);
        }
    /* short-block read overwrites numlines[0..cb_s) -- src/l3psy.c:868 (quirk kept) */
    for (i = 0, k2 = 0; i < cb_s; i++) {
        T->numlines[i] = T_PS_NUMLINES[ri][i];
        T->qthr_s[i] = T_PS_QTHR[ri][i];
        T->SNR_s[i] = T_PS_SNR[ri][i];
        bval_s[i] = T_PS_BVAL[ri][i];
        for (k = 0; k < T->numlines[i]; k++) T->partition_s[k2++] = i;
    }
    (void) bval_s; /* s3_s / norm_s are computed by the reference but never used (src/l3psy.c:893-920) */
    assert(k2 <= HBLK_S);
    for (i = 0; i < 21; i++) {
        T->bu_l[i] = T_SL_BU[ri][i]; T->bo_l[i] = T_SL_BO[ri][i];
        T->w1_l[i] = T_SL_W1[ri][i]; T->w2_l[i] = T_SL_W2[ri][i];
    }
    for (i = 0; i < 12; i++) {
        T->bu_s[i] = T_SS_BU[ri][i]; T->bo_s[i] = T_SS_BO[ri][i];
        T->w1_s[i] = T_SS_W1[ri][i]; T->w2_s[i] = T_SS_W2[ri][i];
    }

    for (i = 4; i <= 10; i++) {
        T->tw_rs[i] = make_twiddles(i, 0);
        T->tw_sr[i] = make_twiddles(i, 1);
    }

    /* create_ana_filter, src/encode.c:331-345; PI64 expands textually to PI/64 */
    for (i = 0; i < 32; i++)
        for (k = 0; k < 64; k++) {
            double f = 1e9 * cos((double) ((2 * i + 1) * (16 - k) * R_PI / 64));
            if (f >= 0) modf(f + 0.5, &f);
            else modf(f - 0.5, &f);
            T->filt[i][k] = f * 1e-9;
        }

    /* mdct windows and cosine tables, src/mdct.c:129-171 */
    for (i = 0; i < 36; i++) T->win[0][i] = sin(R_PI / 36 * (i + 0.5));
    for (i = 0; i < 18; i++) T->win[1][i] = sin(R_PI / 36 * (i + 0.5));
    for (i = 18; i < 24; i++) T->win[1][i] = 1.0;
    for (i = 24; i < 30; i++) T->win[1][i] = sin(R_PI / 12 * (i + 0.5 - 18));
    for (i = 30; i < 36; i++) T->win[1][i] = 0.0;
    for (i = 0; i < 6; i++) T->win[3][i] = 0.0;
    for (i = 6; i < 12; i++) T->win[3][i] = sin(R_PI / 12 * (i + 0.5 - 6));
    for (i = 12; i < 18; i++) T->win[3][i] = 1.0;
    for (i = 18; i < 36; i++) T->win[3][i] = sin(R_PI / 36 * (i + 0.5));
    for (i = 0; i < 12; i++) T->win[2][i] = sin(R_PI / 12 * (i + 0.5));
    for (i = 12; i < 36; i++) T->win[2][i] = 0.0;
    N = 12;
    for (m = 0; m < N / 2; m++)
        for (k = 0; k < N; k++)
            T->cos_s[m][k] = cos((R_PI / (2 * N)) * (2 * k + 1 + N / 2) * (2 * m + 1)) / (N / 4);
    N = 36;
    for (m = 0; m < N / 2; m++)
        for (k = 0; k < N; k++)
            T->cos_l[m][k] = cos((R_PI / (2 * N)) * (2 * k + 1 + N / 2) * (2 * m + 1)) / (N / 4);
    for (k = 0; k < 8; k++) {
        double sq = sqrt(1.0 + ALIAS_C[k] * ALIAS_C[k]);
        T->ca[k] = ALIAS_C[k] / sq;
        T->cs[k] = 1.0 / sq;
    }

    for (i = 1; i < 4096; i++) T->pow_nint_tab[i] = pow((double) i - 0.4054, 4.0 / 3.0);
    for (i = 0; i < 8300; i++) T->pow43[i] = pow((double) i, 4.0 / 3.0);
    return T;
}

static void free_tables(tables_t *T)
{
    int i;
    for (i = 4; i <= 10; i++) { free(T->tw_rs[i]); free(T->tw_sr[i]); }
    free(T);
}

/* ------------------------------------------------------------------------- */
/* real split-radix FFT, single precision (src/subs.c)                       */
/* ------------------------------------------------------------------------- */
static void fft_complex_rec(const tables_t *T, float *xr, float *xi, int logm)
{ /* srrec, src/subs.c:185-362 */
    int m, m2, m4, m8, nel, n, e;
    float tmp1, tmp2;
    const float *tw;
    if (logm == 0) return;
    if (logm == 1) {
        tmp1 = xr[0] + xr[1]; xr[1] = xr[0] - xr[1]; xr[0] = tmp1;
        tmp1 = xi[0] + xi[1]; xi[1] = xi[0] - xi[1]; xi[0] = tmp1;
        return;
    }
    if (logm == 2) {
        tmp1 = xr[0] + xr[2]; xr[2] = xr[0] - xr[2]; xr[0] = tmp1;
        tmp1 = xi[0] + xi[2]; xi[2] = xi[0] - xi[2]; xi[0] = tmp1;
        tmp1 = xr[1] + xr[3]; xr[3] = xr[1] - xr[3]; xr[1] = tmp1;
        tmp1 = xi[1] + xi[3]; xi[3] = xi[1] - xi[3]; xi[1] = tmp1;
        tmp1 = xr[0] + xr[1]; xr[1] = xr[0] - xr[1]; xr[0] = tmp1;
        tmp1 = xi[0] + xi[1]; xi[1] = xi[0] - xi[1]; xi[0] = tmp1;
        tmp1 = xr[2] + xi[3];
        tmp2 = xi[2] + xr[3];
        xi[2] = xi[2] - xr[3];
        xr[3] = xr[2] - xi[3];
        xr[2] = tmp1;
        xi[3] = tmp2;
        return;
    }
    m = 1 << logm; m2 = m / 2; m4 = m2 / 2; m8 = m4 / 2;
    for (n = 0; n < m2; n++) { /* step 1 */
        tmp1 = xr[n] + xr[n + m2]; xr[n + m2] = xr[n] - xr[n + m2]; xr[n] = tmp1;
        tmp2 = xi[n] + xi[n + m2]; xi[n + m2] = xi[n] - xi[n + m2]; xi[n] = tmp2;
    }
    for (n = 0; n < m4; n++) { /* step 2 */
        float *r1 = xr + m2 + n, *r2 = r1 + m4, *i1 = xi + m2 + n, *i2 = i1 + m4;
        tmp1 = *r1 + *i2;
        tmp2 = *i1 + *r2;
        *i1 = *i1 - *r2;
        *r2 = *r1 - *i2;
        *r1 = tmp1;
        *i2 = tmp2;
    }
    nel = m4 - 2;
    tw = (logm >= 4) ? T->tw_sr[logm] : NULL;
    for (n = 1, e = 0; n < m4; n++) { /* steps 3 & 4 */
        float *r1 = xr + m2 + n, *r2 = r1 + m4, *i1 = xi + m2 + n, *i2 = i1 + m4;
        if (n == m8) {
            tmp1 = (float) (R_SQHALF * (*r1 + *i1));
            *i1 = (float) (R_SQHALF * (*i1 - *r1));
            *r1 = tmp1;
            tmp2 = (float) (R_SQHALF * (*i2 - *r2));
            *i2 = (float) (-R_SQHALF * (*r2 + *i2));
            *r2 = tmp2;
        } else {
            tmp2 = tw[e] * (*r1 + *i1);
            tmp1 = tw[nel + e] * *r1 + tmp2;
            *r1 = tw[2 * nel + e] * *i1 + tmp2;
            *i1 = tmp1;
            tmp2 = tw[3 * nel + e] * (*r2 + *i2);
            tmp1 = tw[4 * nel + e] * *r2 + tmp2;
            *r2 = tw[5 * nel + e] * *i2 + tmp2;
            *i2 = tmp1;
            e++;
        }
    }
    fft_complex_rec(T, xr, xi, logm - 1);
    fft_complex_rec(T, xr + m2, xi + m2, logm - 2);
    fft_complex_rec(T, xr + 3 * (m / 4), xi + 3 * (m / 4), logm - 2);
}

static void fft_real_rec(const tables_t *T, float *x, int logm)
{ /* rsrec, src/subs.c:412-523 */
    int m, m2, m4, m8, nel, n, e;
    float tmp1, tmp2;
    const float *tw;
    if (logm == 0) return;
    if (logm == 1) {
        tmp1 = x[0] + x[1]; x[1] = x[0] - x[1]; x[0] = tmp1;
        return;
    }
    m = 1 << logm; m2 = m / 2; m4 = m2 / 2; m8 = m4 / 2;
    for (n = 0; n < m2; n++) {
        tmp1 = x[n] + x[n + m2]; x[n + m2] = x[n] - x[n + m2]; x[n] = tmp1;
    }
    for (n = 0; n < m4; n++) x[m2 + m4 + n] = -x[m2 + m4 + n];
    nel = m4 - 2;
    tw = (logm >= 4) ? T->tw_rs[logm] : NULL;
    for (n = 1, e = 0; n < m4; n++) {
        float *r1 = x + m2 + n, *i1 = r1 + m4;
        if (n == m8) {
            tmp1 = (float) (R_SQHALF * (*r1 + *i1));
            *i1 = (float) (R_SQHALF * (*i1 - *r1));
            *r1 = tmp1;
        } else {
            tmp2 = tw[e] * (*r1 + *i1);
            tmp1 = tw[nel + e] * *r1 + tmp2;
            *r1 = tw[2 * nel + e] * *i1 + tmp2;
            *i1 = tmp1;
            e++;
        }
    }
    fft_real_rec(T, x, logm - 1);
    fft_complex_rec(T, x + m2, x + 3 * (m / 4), logm - 2);
    { /* step 5: sign change and reorder */
        float *r1 = x + m2 + m4, *r2 = x + m - 1;
        for (n = 0; n < m8; n++) {
            tmp1 = *r1;
            *r1++ = -*r2;
            *r2-- = -tmp1;
        }
        r1 = x + m2 + 1;
        r2 = x + m - 2;
        for (n = 0; n < m8; n++) {
            tmp1 = *r1;
            *r1++ = -*r2;
            *r2-- = tmp1;
            r1++;
            r2--;
        }
        if (logm == 2) x[3] = -x[3];
    }
}

static void bit_reverse(float *x, int logm)
{ /* BR_permute, src/subs.c:136-177 (Evans' algorithm = plain bit-reversal permutation) */
    int n = 1 << logm, i, j, b;
    for (i = 0; i < n; i++) {
        for (j = 0, b = 0; b < logm; b++)
            if (i & (1 << b)) j |= 1 << (logm - 1 - b);
        if (j > i) { float t = x[i]; x[i] = x[j]; x[j] = t; }
    }
}

static void fft_energy_phase(const tables_t *T, float *x, float *energy, float *phi, int N)
{ /* fft + enphinew, src/subs.c:38-123 */
    int logm = (N == 1024) ? 10 : 8, i, h = N / 2;
    fft_real_rec(T, x, logm);
    bit_reverse(x, logm);
    energy[0] = x[0] * x[0];
    phi[0] = (float) atan2(0.0, (double) x[0]);
    for (i = 1; i < h; i++) {
        float re = x[i], im = x[N - i];
        energy[i] = re * re + im * im;
        if (energy[i] < 0.0005) {
            energy[i] = (float) 0.0005;
            phi[i] = 0.0f;
        } else
            phi[i] = (float) atan2(-(double) im, (double) re);
    }
    for (i = 1; i < h; i++) {
        energy[h + i] = energy[h - i];
        phi[h + i] = -phi[h - i];
    }
    energy[h] = x[h] * x[h];
    phi[h] = (float) atan2(0.0, (double) x[h]);
}

/* ------------------------------------------------------------------------- */
/* psychoacoustic model 2, Layer III branch (src/l3psy.c:443-740)            */
/* ------------------------------------------------------------------------- */
#define MAXD(a, b) (((a) > (b)) ? (a) : (b))
#define MIND(a, b) (((a) < (b)) ? (a) : (b))

static void psy_granule(mp3o_stream *s, const short *buffer, int chn, double ratio_d[21],
                        double ratio_ds[12][3], double *pe, gr_info_t *cod_info)
{
    const tables_t *T = s->T;
    float wsamp[1024], energy[1024], phi[1024], energy_s[3][256], phi_s[3][256];
    float cb[CBANDS], ecb[CBANDS], nb[CBANDS];
    double cw[HBLK], eb[CBANDS], ctb[CBANDS], thr[CBANDS], SNR_l[CBANDS], en[21], thm[21];
    short *savebuf = s->savebuf[chn];
    int b, j, k, sb, sblock, blocktype, nw, old, oldest;

    /* outputs are the PREVIOUS call's ratios for this channel, :452-456 */
    for (j = 0; j < 21; j++) ratio_d[j] = s->ratio[chn][j];
    for (j = 0; j < 12; j++)
        for (k = 0; k < 3; k++) ratio_ds[j][k] = s->ratio_s[chn][j][k];

    if (chn == 0) { /* :458-470 */
        if (s->age_new == 0) { s->age_new = 1; s->age_old = 0; s->age_oldest = 1; }
        else { s->age_new = 0; s->age_old = 1; s->age_oldest = 0; }
    }
    nw = s->age_new; old = s->age_old; oldest = s->age_oldest;

    for (j = 0; j < 768; j++) savebuf[j] = savebuf[j + 576]; /* :477-481 */
    for (j = 768; j < 1344; j++) savebuf[j] = *buffer++;

    for (j = 0; j < 1024; j++) wsamp[j] = T->window[j] * savebuf[j];
    fft_energy_phase(T, wsamp, energy, phi, 1024);
    if (s->fft_seam) { /* (fft_seam.h; wsamp holds the spectrum: re at [i], im at [N - i]) */
        memcpy(s->fft_seam->energy_l, energy, sizeof(s->fft_seam->energy_l));
        for (j = 0; j < 6; j++) {
            s->fft_seam->phi_l[j] = phi[j];
            s->fft_seam->re_l[j] = wsamp[j];
            s->fft_seam->im_l[j] = j ? wsamp[1024 - j] : 0.0f;
        }
    }

    for (j = 0; j < 6; j++) { /* :496-512 */
        double r_prime = 2.0 * s->r[chn][old][j] - s->r[chn][oldest][j];
        double phi_prime = 2.0 * s->phi_sav[chn][old][j] - s->phi_sav[chn][oldest][j];
        double t1, t2, t3;
        s->r[chn][nw][j] = (float) sqrt((double) energy[j]);
        s->phi_sav[chn][nw][j] = phi[j];
        t1 = s->r[chn][nw][j] * cos((double) phi[j]) - r_prime * cos(phi_prime);
        t2 = s->r[chn][nw][j] * sin((double) phi[j]) - r_prime * sin(phi_prime);
        t3 = s->r[chn][nw][j] + fabs(r_prime);
        if (t3 != 0.0) cw[j] = sqrt(t1 * t1 + t2 * t2) / t3;
        else cw[j] = 0;
    }

    for (sblock = 0; sblock < 3; sblock++) { /* :518-527 */
        for (j = 0, k = 128 * (2 + sblock); j < 256; j++, k++) wsamp[j] = T->window_s[j] * savebuf[k];
        fft_energy_phase(T, wsamp, energy_s[sblock], phi_s[sblock], 256);
        if (s->fft_seam) {
            memcpy(s->fft_seam->energy_s[sblock], energy_s[sblock], sizeof(s->fft_seam->energy_s[sblock]));
            for (j = 0; j < 50; j++) {
                s->fft_seam->phi_s[sblock][j] = phi_s[sblock][2 + j];
                s->fft_seam->re_s[sblock][j] = wsamp[2 + j];
                s->fft_seam->im_s[sblock][j] = wsamp[256 - 2 - j];
            }
        }
    }
    if (s->fft_seam) { /* the next call's record follows this one: call order = [frame][gr][ch] */
        if (--s->fft_seam_left > 0) s->fft_seam++;
        else s->fft_seam = NULL;
    }
    for (j = 6; j < 206; j += 4) { /* :531-549 */
        double r_prime, phi_prime, r2, phi2, t1, t2, t3;
        k = (j + 2) >> 2;
        r_prime = 2.0 * sqrt((double) energy_s[0][k]) - sqrt((double) energy_s[2][k]);
        phi_prime = 2.0 * phi_s[0][k] - phi_s[2][k];
        r2 = sqrt((double) energy_s[1][k]);
        phi2 = phi_s[1][k];
        t1 = r2 * cos(phi2) - r_prime * cos(phi_prime);
        t2 = r2 * sin(phi2) - r_prime * sin(phi_prime);
        t3 = r2 + fabs(r_prime);
        if (t3 != 0.0) cw[j] = sqrt(t1 * t1 + t2 * t2) / t3;
        else cw[j] = 0.0;
        cw[j + 1] = cw[j + 2] = cw[j + 3] = cw[j];
    }
    for (j = 206; j < HBLK; j++) cw[j] = 0.4;

    for (b = 0; b < CBANDS; b++) { eb[b] = 0.0; cb[b] = 0.0f; } /* :565-578 */
    for (j = 0; j < HBLK; j++) {
        int tp = T->partition_l[j];
        eb[tp] += energy[j];
        cb[tp] = (float) (cb[tp] + cw[j] * energy[j]);
    }

    for (b = 0; b < CBANDS; b++) { ecb[b] = 0.0f; ctb[b] = 0.0; } /* :586-605 */
    if (s->rate_idx == 0) { /* 44.1 kHz sparse form, sprdngf1/2 :1062-1084 */
        for (b = 0; b < CBANDS; b++)
            for (k = T_S3_LO[b]; k <= T_S3_HI[b]; k++) ecb[b] = (float) (ecb[b] + T->s3_l[b][k] * eb[k]);
        for (b = 0; b < CBANDS; b++)
            for (k = T_S3_LO[b]; k <= T_S3_HI[b]; k++) ctb[b] += T->s3_l[b][k] * cb[k];
    } else {
        for (b = 0; b < CBANDS; b++)
            for (k = 0; k < CBANDS; k++)
                if (T->s3_l[b][k] != 1.0) {
                    ecb[b] = (float) (ecb[b] + T->s3_l[b][k] * eb[k]);
                    ctb[b] += T->s3_l[b][k] * cb[k];
                }
    }

    for (b = 0; b < CBANDS; b++) { /* :610-623 */
        double cbb, tbb;
        if (ecb[b] != 0.0) {
            cbb = ctb[b] / ecb[b];
            if (cbb < 0.01) cbb = 0.01;
            cbb = log(cbb);
        } else cbb = 0.0;
        tbb = -0.299 - 0.43 * cbb;
        tbb = MIND(1.0, MAXD(0.0, tbb));
        SNR_l[b] = MAXD(T->minval[b], 29.0 * tbb + 6.0 * (1.0 - tbb));
    }
    for (b = 0; b < CBANDS; b++) /* :626-627 */
        nb[b] = (float) (ecb[b] * T->norm_l[b] * exp(-SNR_l[b] * R_LN_TO_LOG10));
    for (b = 0; b < CBANDS; b++) { /* pre-echo control :629-636 */
        double t = MIND((double) nb[b], MIND(2.0 * s->nb_1[chn][b], 16.0 * s->nb_2[chn][b]));
        thr[b] = MAXD(T->qthr_l[b], t);
        s->nb_2[chn][b] = s->nb_1[chn][b];
        s->nb_1[chn][b] = nb[b];
    }
    *pe = 0.0; /* :639-645 */
    for (b = 0; b < CBANDS; b++) {
        double tp = MIND(0.0, log((thr[b] + 1.0) / (eb[b] + 1.0)));
        *pe -= T->numlines[b] * tp;
    }

    blocktype = BT_NORM;
    if (*pe < 1800) { /* :651-685 */
        if (s->blocktype_old[chn] == BT_SHORT) blocktype = BT_STOP;
        else blocktype = BT_NORM;
        for (sb = 0; sb < 21; sb++) {
            en[sb] = T->w1_l[sb] * eb[T->bu_l[sb]] + T->w2_l[sb] * eb[T->bo_l[sb]];
            thm[sb] = T->w1_l[sb] * thr[T->bu_l[sb]] + T->w2_l[sb] * thr[T->bo_l[sb]];
            for (b = T->bu_l[sb] + 1; b < T->bo_l[sb]; b++) { en[sb] += eb[b]; thm[sb] += thr[b]; }
            if (en[sb] != 0.0) s->ratio[chn][sb] = thm[sb] / en[sb];
            else s->ratio[chn][sb] = 0.0;
        }
    } else { /* attack :687-730 */
        blocktype = BT_SHORT;
        if (s->blocktype_old[chn] == BT_NORM) s->blocktype_old[chn] = BT_START;
        if (s->blocktype_old[chn] == BT_STOP) s->blocktype_old[chn] = BT_SHORT;
        for (sblock = 0; sblock < 3; sblock++) {
            for (b = 0; b < CBANDS_S; b++) { eb[b] = 0.0; ecb[b] = 0.0f; }
            for (j = 0; j < HBLK_S; j++) eb[T->partition_s[j]] += energy_s[sblock][j];
            for (b = 0; b < CBANDS_S; b++)
                for (k = 0; k < CBANDS_S; k++) ecb[b] = (float) (ecb[b] + T->s3_l[b][k] * eb[k]);
            for (b = 0; b < CBANDS_S; b++) {
                nb[b] = (float) (ecb[b] * T->norm_l[b] * exp((double) T->SNR_s[b] * R_LN_TO_LOG10));
                thr[b] = MAXD(T->qthr_s[b], (double) nb[b]);
            }
            for (sb = 0; sb < 12; sb++) {
                en[sb] = T->w1_s[sb] * eb[T->bu_s[sb]] + T->w2_s[sb] * eb[T->bo_s[sb]];
                thm[sb] = T->w1_s[sb] * thr[T->bu_s[sb]] + T->w2_s[sb] * thr[T->bo_s[sb]];
                for (b = T->bu_s[sb] + 1; b < T->bo_s[sb]; b++) { en[sb] += eb[b]; thm[sb] += thr[b]; }
                if (en[sb] != 0.0) s->ratio_s[chn][sb][sblock] = thm[sb] / en[sb];
                else s->ratio_s[chn][sb][sblock] = 0.0;
            }
        }
    }
    cod_info->block_type = (unsigned) s->blocktype_old[chn]; /* :732-739 */
    s->blocktype_old[chn] = blocktype;
    cod_info->window_switching_flag = (cod_info->block_type == BT_NORM) ? 0 : 1;
    cod_info->mixed_block_flag = 0;
}

/* ------------------------------------------------------------------------- */
/* polyphase analysis filterbank (src/encode.c:287-409)                      */
/* ------------------------------------------------------------------------- */
static void filterbank_slot(mp3o_stream *s, const short *in32, int ch, double out[32])
{
    const tables_t *T = s->T;
    double z[512], y[64], ysum[16], ysub[16];
    int i, j, off = s->fb_off[ch];
    for (i = 0; i < 32; i++) s->fb_x[ch][31 - i + off] = (double) in32[i] / 32768; /* :306-307 */
    for (i = 0; i < 512; i++) z[i] = s->fb_x[ch][(i + off) & 511] * T_ENWINDOW[i];
    s->fb_off[ch] = (off + 480) & 511;
    for (i = 0; i < 64; i++) /* :393-397 */
        y[i] = z[i] + z[i + 64] + z[i + 128] + z[i + 192] + z[i + 256] + z[i + 320] + z[i + 384] + z[i + 448];
    for (i = 0; i < 16; i++) ysum[i] = y[i] + y[32 - i];
    for (i = 0; i < 15; i++) ysub[i] = y[33 + i] - y[63 - i];
    for (i = 0; i < 32; i++) {
        double si = y[16];
        for (j = 0; j < 16; j++) si += T->filt[i][j] * ysum[j];
        for (j = 0; j < 15; j++) si += T->filt[i][33 + j] * ysub[j];
        out[i] = si;
    }
}

/* ------------------------------------------------------------------------- */
/* MDCT (src/mdct.c)                                                         */
/* ------------------------------------------------------------------------- */
static void mdct_one(const tables_t *T, const double in[36], double *out, int block_type)
{ /* mdct, src/mdct.c:105-511 */
    double fin[36], sum;
    int k, l, m;
    if (block_type == 2) {
        for (l = 0; l < 3; l++)
            for (m = 0; m < 6; m++) {
                for (sum = 0.0, k = 0; k < 12; k++) sum += T->win[2][k] * in[k + 6 * l + 6] * T->cos_s[m][k];
                out[3 * m + l] = sum;
            }
    } else if (block_type != 0) {
        for (k = 0; k < 36; k++) fin[k] = T->win[block_type][k] * in[k];
        for (m = 0; m < 18; m++) {
            for (sum = 0.0, k = 0; k < 36; k++) sum += fin[k] * T->cos_l[m][k];
            out[m] = sum;
        }
    } else { /* hand-unrolled long transform :199-509, term/operand order from T_MDCTL_* */
        for (k = 0; k < 36; k++) fin[k] = T->win[0][k] * in[k];
        for (m = 0; m < 18; m++) {
            int t;
            sum = 0.0;
            for (t = T_MDCTL_ROW[m]; t < T_MDCTL_ROW[m + 1]; t++) {
                int o, o0 = T_MDCTL_TERM_OP[t], o1 = T_MDCTL_TERM_OP[t + 1];
                double acc, c;
                unsigned char op = T_MDCTL_OPS[o0];
                acc = (op & 0x80) ? -fin[op & 0x7f] : fin[op & 0x7f];
                for (o = o0 + 1; o < o1; o++) {
                    op = T_MDCTL_OPS[o];
                    if (op & 0x80) acc = acc - fin[op & 0x7f];
                    else acc = acc + fin[op & 0x7f];
                }
                c = T->cos_l[m][T_MDCTL_TERM_K[t] & 0x7f];
                if (T_MDCTL_TERM_K[t] & 0x80) c = -c;
                if (t == T_MDCTL_ROW[m]) sum = acc * c;
                else sum += acc * c;
            }
            out[m] = sum;
        }
    }
}

static void mdct_frame(mp3o_stream *s, double xr[2][2][576])
{ /* mdct_sub, src/mdct.c:25-103 */
    const tables_t *T = s->T;
    double in[36];
    int gr, ch, band, k;
    for (gr = 0; gr < 2; gr++)
        for (ch = 0; ch < s->channels; ch++) {
            int block_type = (int) s->side.gr[gr][ch].block_type;
            double (*enc)[18] = (double (*)[18]) xr[gr][ch];
            for (band = 1; band < 32; band += 2)
                for (k = 1; k < 18; k += 2) s->sb[ch][gr + 1][k][band] *= -1.0;
            for (band = 0; band < 32; band++) {
                for (k = 0; k < 18; k++) {
                    in[k] = s->sb[ch][gr][k][band];
                    in[k + 18] = s->sb[ch][gr + 1][k][band];
                }
                mdct_one(T, in, enc[band], block_type);
            }
            if (block_type != 2)
                for (band = 0; band < 31; band++)
                    for (k = 0; k < 8; k++) {
                        double bu = enc[band][17 - k] * T->cs[k] + enc[band + 1][k] * T->ca[k];
                        double bd = enc[band + 1][k] * T->cs[k] - enc[band][17 - k] * T->ca[k];
                        enc[band][17 - k] = bu;
                        enc[band + 1][k] = bd;
                    }
        }
    for (ch = 0; ch < s->channels; ch++) memcpy(s->sb[ch][0], s->sb[ch][2], sizeof(s->sb[0][0]));
}

/* ------------------------------------------------------------------------- */
/* iteration loop (src/loop.c) and bit reservoir (src/reservoir.c)           */
/* ------------------------------------------------------------------------- */
typedef struct { double l[21]; double s[12][3]; } xmin_t;

static int r_nint(double in)
{ /* src/loop.c:2020-2029 (HAVE_NINT undefined) */
    return (in < 0) ? (int) (in - 0.5) : (int) (in + 0.5);
}

static int pow_nint(const tables_t *T, double x)
{ /* src/pow_nint.h:15-49: probe by doubling, then bisect; saturates at 2047 */
    const double *tab = T->pow_nint_tab;
    int step = 1, pos = 1, p = 0;
    while (pos < 2048) {
        if (x < tab[pos]) break;
        p = pos;
        pos += step;
        step <<= 1;
    }
    step >>= 1;
    pos -= step;
    step >>= 1;
    if (step) {
        while (step) {
            if (x < tab[pos]) pos -= step;
            else { p = pos; pos += step; }
            step >>= 1;
        }
        if (x >= tab[pos]) p = pos;
    }
    return p;
}

static int count_bit(const int *ix, unsigned start, unsigned end, unsigned table)
{ /* src/loop.c:172-225 */
    unsigned linbits, ylen, off;
    int i, sum = 0;
    if (table == 0) return 0;
    ylen = T_HT_YLEN[table];
    linbits = T_HT_LINBITS[table];
    off = T_HT_OFF[table];
    for (i = (int) start; i < (int) end; i += 2) {
        int x = ix[i], y = ix[i + 1];
        if (table > 15) {
            if (x > 14) { x = 15; sum += linbits; }
            if (y > 14) { y = 15; sum += linbits; }
        }
        sum += T_HT_PACKED[off + x * ylen + y] & 0xff;
        if (x != 0) sum++;
        if (y != 0) sum++;
    }
    return sum;
}

static int pair_bits(unsigned table, int x, int y)
{ /* HuffmanCode with a NULL holder, src/huffcode.h:16-139: bits only */
    unsigned ylen, linbits;
    int bits = 0;
    if (table == 0) return 0;
    if (x < 0) x = -x;
    if (y < 0) y = -y;
    ylen = T_HT_YLEN[table];
    linbits = T_HT_LINBITS[table];
    if (table > 15) {
        if (x > 14) { x = 15; bits += linbits; }
        if (y > 14) { y = 15; bits += linbits; }
    }
    bits += T_HT_PACKED[T_HT_OFF[table] + x * ylen + y] & 0xff;
    if (x != 0) bits++;
    if (y != 0) bits++;
    return bits;
}

static int ix_max(const int *ix, unsigned begin, unsigned end)
{ /* src/loop.c:1441-1452 */
    int i, max = 0;
    for (i = (int) begin; i < (int) end; i++) {
        int x = abs(ix[i]);
        if (x > max) max = x;
    }
    return max;
}

static double xr_max(const double *xr, unsigned begin, unsigned end)
{ /* src/loop.c:1463-1472 */
    int i;
    double max = 0.0, temp;
    for (i = (int) begin; i < (int) end; i++)
        if ((temp = fabs(xr[i])) > max) max = temp;
    return max;
}

static void gr_deco(gr_info_t *g)
{ /* src/loop.c:2063-2081 (mixed_block_flag is always 0) */
    if (g->window_switching_flag != 0 && g->block_type == 2) {
        if (g->mixed_block_flag == 0) { g->sfb_lmax = 0; g->sfb_smax = 0; }
        else { g->sfb_lmax = 8; g->sfb_smax = 3; }
    } else { g->sfb_lmax = 21; g->sfb_smax = 12; }
}

static void calc_xmin(const mp3o_stream *s, const double *xr, const double ratio_l[21],
                      double ratio_s[12][3], const gr_info_t *g, xmin_t *xm)
{ /* src/loop.c:1085-1118 */
    const int *bl = SFB_L[s->rate_idx], *bs = SFB_S[s->rate_idx];
    unsigned sfb;
    int l, b;
    for (sfb = g->sfb_smax; sfb < 12; sfb++) {
        int start = bs[sfb], end = bs[sfb + 1];
        double bw = end - start, en;
        for (b = 0; b < 3; b++) {
            for (en = 0.0, l = start; l < end; l++) en += xr[l * 3 + b] * xr[l * 3 + b];
            xm->s[sfb][b] = ratio_s[sfb][b] * en / bw;
        }
    }
    for (sfb = 0; sfb < g->sfb_lmax; sfb++) {
        int start = bl[sfb], end = bl[sfb + 1];
        double bw = end - start, en;
        for (en = 0.0, l = start; l < end; l++) en += xr[l] * xr[l];
        xm->l[sfb] = ratio_l[sfb] * en / bw;
    }
}

static void calc_scfsi(mp3o_stream *s, const double *xr, const xmin_t *xm, int ch, int gr)
{ /* src/loop.c:615-715, including its index transpositions and int truncations */
    const int *bl = SFB_L[s->rate_idx];
    const gr_info_t *g = &s->side.gr[gr][ch];
    int sfb, i, condition = 0, scfsi_band;
    double temp, log2 = log(2.0);
    s->sc_xrmax[gr][ch] = (int) xr_max(xr, 0, 576);
    for (temp = 0.0, i = 0; i < 576; i++) temp += xr[i] * xr[i];
    if (temp == 0.0) s->sc_en_tot[gr][ch] = 0;
    else s->sc_en_tot[gr][ch] = (int) (log(temp) / log2);
    if (g->window_switching_flag == 0 || g->block_type != 2)
        for (sfb = 0; sfb < 21; sfb++) {
            int start = bl[sfb], end = bl[sfb + 1];
            for (temp = 0.0, i = start; i < end; i++) temp += xr[i] * xr[i];
            if (temp == 0.0) s->sc_en[gr][ch][sfb] = 0;
            else s->sc_en[gr][ch][sfb] = (int) (log(temp) / log2);
            if (xm->l[sfb] == 0.0) s->sc_xm[gr][ch][sfb] = 0;
            else s->sc_xm[gr][ch][sfb] = (int) (log(xm->l[sfb]) / log2);
        }
    if (gr == 1) {
        int gr2, tp;
        for (gr2 = 0; gr2 < 2; gr2++) {
            if (s->sc_xrmax[ch][gr2] != 0.0) condition++; /* [ch][gr2], sic */
            if (g->window_switching_flag == 0 || g->block_type != 2) condition++;
        }
        /* abs(en_tot[0] - en_tot[1]) is a pointer difference (= 2) in the reference: always < 10 */
        condition++;
        for (tp = 0, sfb = 0; sfb < 21; sfb++) tp += abs(s->sc_en[ch][0][sfb] - s->sc_en[ch][1][sfb]);
        if (tp < 100) condition++;
        if (condition == 6) {
            for (scfsi_band = 0; scfsi_band < 4; scfsi_band++) {
                int sum0 = 0, sum1 = 0;
                for (sfb = SCFSI_BAND_L[scfsi_band]; sfb < SCFSI_BAND_L[scfsi_band + 1]; sfb++) {
                    sum0 += abs(s->sc_en[ch][0][sfb] - s->sc_en[ch][1][sfb]);
                    sum1 += abs(s->sc_xm[ch][0][sfb] - s->sc_xm[ch][1][sfb]);
                }
                s->side.scfsi[ch][scfsi_band] = (sum0 < 10 && sum1 < 10) ? 1 : 0;
            }
        } else
            for (scfsi_band = 0; scfsi_band < 4; scfsi_band++) s->side.scfsi[ch][scfsi_band] = 0;
    }
}

static int part2_length(const mp3o_stream *s, int gr, int ch)
{ /* src/loop.c:731-780, MPEG-1 branch */
    const gr_info_t *g = &s->side.gr[gr][ch];
    int slen1 = SLEN1[g->scalefac_compress], slen2 = SLEN2[g->scalefac_compress], bits = 0;
    if (g->window_switching_flag == 1 && g->block_type == 2) {
        bits += 18 * slen1 + 18 * slen2;
    } else {
        if (gr == 0 || s->side.scfsi[ch][0] == 0) bits += 6 * slen1;
        if (gr == 0 || s->side.scfsi[ch][1] == 0) bits += 5 * slen1;
        if (gr == 0 || s->side.scfsi[ch][2] == 0) bits += 5 * slen2;
        if (gr == 0 || s->side.scfsi[ch][3] == 0) bits += 5 * slen2;
    }
    return bits;
}

static int scale_bitcount(mp3o_stream *s, int gr, int ch)
{ /* src/loop.c:792-860 */
    static const int pow2[5] = {1, 2, 4, 8, 16};
    gr_info_t *g = &s->side.gr[gr][ch];
    int i, k, sfb, max1 = 0, max2 = 0, ep = 2;
    if (g->window_switching_flag != 0 && g->block_type == 2) {
        for (i = 0; i < 3; i++) {
            for (sfb = 0; sfb < 6; sfb++)
                if (s->scalefac_s[gr][ch][sfb][i] > max1) max1 = s->scalefac_s[gr][ch][sfb][i];
            for (sfb = 6; sfb < 12; sfb++)
                if (s->scalefac_s[gr][ch][sfb][i] > max2) max2 = s->scalefac_s[gr][ch][sfb][i];
        }
    } else {
        for (sfb = 0; sfb < 11; sfb++)
            if (s->scalefac_l[gr][ch][sfb] > max1) max1 = s->scalefac_l[gr][ch][sfb];
        for (sfb = 11; sfb < 21; sfb++)
            if (s->scalefac_l[gr][ch][sfb] > max2) max2 = s->scalefac_l[gr][ch][sfb];
    }
    for (k = 0; k < 16; k++)
        if (max1 < pow2[SLEN1[k]] && max2 < pow2[SLEN2[k]]) { ep = 0; break; }
    if (ep == 0) g->scalefac_compress = (unsigned) k;
    return ep;
}

static void quantize(const mp3o_stream *s, const double *xr, int *ix, const gr_info_t *g)
{ /* src/loop.c:1360-1428; subblock_gain is always 0 so the short-block view uses the same step */
    double step, ostep;
    int i;
    if (g->quantizerStepSize == 0.0) step = 1.0;
    else step = pow(2.0, g->quantizerStepSize * 0.25);
    ostep = 1.0 / step;
    for (i = 0; i < 576; i++) ix[i] = pow_nint(s->T, fabs(xr[i]) * ostep);
}

static void calc_runlen(const int *ix, gr_info_t *g)
{ /* src/loop.c:1488-1519 */
    int i, rzero = 0;
    if (g->window_switching_flag && g->block_type == 2) {
        g->count1 = 0;
        g->big_values = 288;
    } else {
        for (i = 576; i > 1; i -= 2)
            if (ix[i - 1] == 0 && ix[i - 2] == 0) rzero++;
            else break;
        g->count1 = 0;
        for (; i > 3; i -= 4)
            if (abs(ix[i - 1]) <= 1 && abs(ix[i - 2]) <= 1 && abs(ix[i - 3]) <= 1 && abs(ix[i - 4]) <= 1)
                g->count1++;
            else break;
        g->big_values = (unsigned) (i / 2);
    }
    assert(2 * rzero + 4 * g->count1 + 2 * g->big_values == 576 || (g->block_type == 2));
}

static int count1_bitcount(const int *ix, gr_info_t *g)
{ /* src/loop.c:1531-1591 */
    int i, k, sum0 = 0, sum1 = 0;
    for (i = (int) g->big_values * 2, k = 0; k < (int) g->count1; i += 4, k++) {
        int v = abs(ix[i]), w = abs(ix[i + 1]), x = abs(ix[i + 2]), y = abs(ix[i + 3]);
        int p = v + (w << 1) + (x << 2) + (y << 3);
        int signbits = (v != 0) + (w != 0) + (x != 0) + (y != 0);
        sum0 += signbits + (int) (T_HT_PACKED[T_HT_OFF[32] + p] & 0xff);
        sum1 += signbits + (int) (T_HT_PACKED[T_HT_OFF[33] + p] & 0xff);
    }
    if (sum0 < sum1) { g->count1table_select = 0; return sum0; }
    g->count1table_select = 1;
    return sum1;
}

static void subdivide(const mp3o_stream *s, gr_info_t *g)
{ /* src/loop.c:1638-1706 */
    const int *bl = SFB_L[s->rate_idx];
    if (g->big_values == 0) {
        g->region0_count = 0;
        g->region1_count = 0;
    } else {
        int bigvalues_region = 2 * (int) g->big_values;
        if (g->window_switching_flag == 0) {
            int scfb_anz = 0, thiscount, index;
            while (bl[scfb_anz] < bigvalues_region) scfb_anz++;
            assert(scfb_anz < 23);
            thiscount = SUBDV[scfb_anz][0];
            index = thiscount + 1;
            while (thiscount && bl[index] > bigvalues_region) { thiscount--; index--; }
            g->region0_count = (unsigned) thiscount;
            thiscount = SUBDV[scfb_anz][1];
            index = (int) g->region0_count + thiscount + 2;
            while (thiscount && bl[index] > bigvalues_region) { thiscount--; index--; }
            g->region1_count = (unsigned) thiscount;
            g->address1 = (unsigned) bl[g->region0_count + 1];
            g->address2 = (unsigned) bl[g->region0_count + g->region1_count + 2];
            g->address3 = (unsigned) bigvalues_region;
        } else if (g->block_type == 2 && g->mixed_block_flag == 0) {
            g->region0_count = 8;
            g->region1_count = 36;
            g->address1 = 36;
            g->address2 = (unsigned) bigvalues_region;
            g->address3 = 0;
        } else {
            g->region0_count = 7;
            g->region1_count = 13;
            g->address1 = (unsigned) bl[g->region0_count + 1];
            g->address2 = (unsigned) bigvalues_region;
            g->address3 = 0;
        }
    }
}

static int new_choose_table(const int *ix, unsigned begin, unsigned end)
{ /* src/loop.c:1793-1897 */
    int i, max, choice[2] = {0, 0}, sum[2];
    max = ix_max(ix, begin, end);
    if (max == 0) return 0;
    if (max < 15) {
        for (i = 0; i < 14; i++)
            if (T_HT_XLEN[i] > max) { choice[0] = i; break; }
        sum[0] = count_bit(ix, begin, end, (unsigned) choice[0]);
        switch (choice[0]) {
        case 2:
            sum[1] = count_bit(ix, begin, end, 3);
            if (sum[1] <= sum[0]) choice[0] = 3;
            break;
        case 5:
            sum[1] = count_bit(ix, begin, end, 6);
            if (sum[1] <= sum[0]) choice[0] = 6;
            break;
        case 7:
            sum[1] = count_bit(ix, begin, end, 8);
            if (sum[1] <= sum[0]) { choice[0] = 8; sum[0] = sum[1]; }
            sum[1] = count_bit(ix, begin, end, 9);
            if (sum[1] <= sum[0]) choice[0] = 9;
            break;
        case 10:
            sum[1] = count_bit(ix, begin, end, 11);
            if (sum[1] <= sum[0]) { choice[0] = 11; sum[0] = sum[1]; }
            sum[1] = count_bit(ix, begin, end, 12);
            if (sum[1] <= sum[0]) choice[0] = 12;
            break;
        case 13:
            sum[1] = count_bit(ix, begin, end, 15);
            if (sum[1] <= sum[0]) choice[0] = 15;
            break;
        default: break;
        }
    } else {
        max -= 15;
        for (i = 15; i < 24; i++)
            if ((int) T_HT_LINMAX[i] >= max) { choice[0] = i; break; }
        for (i = 24; i < 32; i++)
            if ((int) T_HT_LINMAX[i] >= max) { choice[1] = i; break; }
        sum[0] = count_bit(ix, begin, end, (unsigned) choice[0]);
        sum[1] = count_bit(ix, begin, end, (unsigned) choice[1]);
        if (sum[1] < sum[0]) choice[0] = choice[1];
    }
    return choice[0];
}

static int choose_table(int max)
{ /* src/loop.c:1908-1947 */
    int i, choice = 0;
    if (max == 0) return 0;
    if (max < 15) {
        for (i = 0; i < 15; i++)
            if (T_HT_XLEN[i] > max) { choice = i; break; }
    } else {
        max -= 15;
        for (i = 15; i < 32; i++)
            if ((int) T_HT_LINMAX[i] >= max) { choice = i; break; }
    }
    return choice;
}

static void bigv_tab_select(const mp3o_stream *s, const int *ix, gr_info_t *g)
{ /* src/loop.c:1717-1780 */
    g->table_select[0] = g->table_select[1] = g->table_select[2] = 0;
    if (g->window_switching_flag && g->block_type == 2) {
        const int *bs = SFB_S[s->rate_idx];
        int sfb, window, line, max1 = 0, max2 = 0;
        for (sfb = 0; sfb < 13; sfb++) {
            int start = bs[sfb], end = bs[sfb + 1];
            int *pmax = (start < 12) ? &max1 : &max2;
            for (window = 0; window < 3; window++)
                for (line = start; line < end; line += 2) {
                    int x = abs(ix[line * 3 + window]), y = abs(ix[(line + 1) * 3 + window]);
                    if (x > *pmax) *pmax = x;
                    if (y > *pmax) *pmax = y;
                }
        }
        g->table_select[0] = (unsigned) choose_table(max1);
        g->table_select[1] = (unsigned) choose_table(max2);
    } else {
        if (g->address1 > 0) g->table_select[0] = (unsigned) new_choose_table(ix, 0, g->address1);
        if (g->address2 > g->address1)
            g->table_select[1] = (unsigned) new_choose_table(ix, g->address1, g->address2);
        if (g->big_values * 2 > g->address2)
            g->table_select[2] = (unsigned) new_choose_table(ix, g->address2, g->big_values * 2);
    }
}

static int bigv_bitcount(const mp3o_stream *s, const int *ix, const gr_info_t *g)
{ /* src/loop.c:1954-2014 */
    int bits = 0;
    if (g->window_switching_flag && g->block_type == 2) {
        const int *bs = SFB_S[s->rate_idx];
        int sfb, window, line;
        for (sfb = 0; sfb < 13; sfb++) {
            int start = bs[sfb], end = bs[sfb + 1];
            unsigned t = (start < 12) ? g->table_select[0] : g->table_select[1];
            for (window = 0; window < 3; window++)
                for (line = start; line < end; line += 2)
                    bits += pair_bits(t, ix[line * 3 + window], ix[(line + 1) * 3 + window]);
        }
    } else {
        if (g->table_select[0]) bits += count_bit(ix, 0, g->address1, g->table_select[0]);
        if (g->table_select[1]) bits += count_bit(ix, g->address1, g->address2, g->table_select[1]);
        if (g->table_select[2]) bits += count_bit(ix, g->address2, g->address3, g->table_select[2]);
    }
    return bits;
}

#if defined(MP3O_CENSUS)
/* tools/pass_census.py: how the search spends its quantise + count passes (build with -DMP3O_CENSUS; single-threaded) */
long long mp3o_census[16];
enum { CEN_GC, CEN_BISECT, CEN_BISECT_OVER, CEN_BISECT_EQUAL, CEN_OUTER, CEN_INNER_FIRST, CEN_INNER_EXTRA, CEN_PASS_ALLZERO,
       CEN_OUTER_NO_AMP, CEN_BISECT_ALLZERO, CEN_SHORT_GC };
#define CENSUS(i, n) (mp3o_census[i] += (n))
#else
#define CENSUS(i, n) ((void) 0)
#endif

static int count_bits(const mp3o_stream *s, const int *ix, gr_info_t *g)
{ /* src/loop.c:2099-2113 */
    int bits;
    calc_runlen(ix, g);
    if (ix_max(ix, 0, 576) > 8192) return 100000;
    bits = count1_bitcount(ix, g);
    subdivide(s, g);
    bigv_tab_select(s, ix, g);
    bits += bigv_bitcount(s, ix, g);
    return bits;
}

static void bin_search_StepSize(const mp3o_stream *s, int desired_rate, double start, int *ix,
                                const double *xrs, gr_info_t *g)
{ /* src/loop.c:2119-2140; aint() truncates through long */
    double top = start, bot = 200, next = start, last;
    int bit;
    do {
        last = next;
        next = (double) (long) ((top + bot) / 2.0);
        g->quantizerStepSize = next;
        quantize(s, xrs, ix, g);
        bit = count_bits(s, ix, g);
        CENSUS(CEN_BISECT, 1); CENSUS(CEN_BISECT_OVER, bit > desired_rate); CENSUS(CEN_BISECT_EQUAL, bit == desired_rate);
        CENSUS(CEN_BISECT_ALLZERO, ix_max(ix, 0, 576) == 0);
        if (bit > desired_rate) top = next;
        else bot = next;
    } while (bit != desired_rate && fabs(last - next) > 1.0);
}

static int inner_loop(const mp3o_stream *s, const double *xrs, int *ix, int max_bits, gr_info_t *g)
{ /* src/loop.c:569-606 */
    int bits;
    if (max_bits < 0) { /* assert( max_bits >= 0 ), src/loop.c:579: the reference dies here (and the loop below would not end) */
        if (!((mp3o_stream *) s)->ref_abort) ((mp3o_stream *) s)->ref_abort = MP3O_ABORT_HUFF_BITS;
        return 0;
    }
    g->quantizerStepSize -= 1.0;
    CENSUS(CEN_INNER_FIRST, 1); CENSUS(CEN_INNER_EXTRA, -1);
    do {
        do {
            g->quantizerStepSize += 1.0;
            quantize(s, xrs, ix, g);
        } while (ix_max(ix, 0, 576) > 8191 + 14);
        CENSUS(CEN_INNER_EXTRA, 1); CENSUS(CEN_PASS_ALLZERO, ix_max(ix, 0, 576) == 0);
        calc_runlen(ix, g);
        bits = count1_bitcount(ix, g);
        subdivide(s, g);
        bigv_tab_select(s, ix, g);
        bits += bigv_bitcount(s, ix, g);
    } while (bits > max_bits);
    return bits;
}

static void calc_noise(const mp3o_stream *s, const double *xr, const int *ix, const gr_info_t *g,
                       double xfsf[4][21])
{ /* src/loop.c:1007-1067; pow43[] holds pow(i, 4/3) for every reachable i */
    const int *bl = SFB_L[s->rate_idx], *bs = SFB_S[s->rate_idx];
    const double *p43 = s->T->pow43;
    double step = pow(2.0, g->quantizerStepSize * 0.25), sum, bw, temp;
    unsigned sfb;
    int l, i;
    for (sfb = 0; sfb < g->sfb_lmax; sfb++) {
        int start = bl[sfb], end = bl[sfb + 1];
        bw = end - start;
        for (sum = 0.0, l = start; l < end; l++) {
            temp = fabs(xr[l]) - p43[ix[l]] * step;
            sum += temp * temp;
        }
        xfsf[0][sfb] = sum / bw;
    }
    for (i = 0; i < 3; i++)
        for (sfb = g->sfb_smax; sfb < 12; sfb++) {
            int start = bs[sfb], end = bs[sfb + 1];
            bw = end - start;
            for (sum = 0.0, l = start; l < end; l++) {
                temp = fabs(xr[l * 3 + i]) - p43[ix[l * 3 + i]] * step;
                sum += temp * temp;
            }
            xfsf[i + 1][sfb] = sum / bw;
        }
}

static void preemphasis(mp3o_stream *s, double *xr, double xfsf[4][21], xmin_t *xm, int gr, int ch)
{ /* src/loop.c:1161-1214 */
    const int *bl = SFB_L[s->rate_idx];
    gr_info_t *g = &s->side.gr[gr][ch];
    int i, sfb, scfsi_band, over;
    if (gr == 1)
        for (scfsi_band = 0; scfsi_band < 4; scfsi_band++)
            if (s->side.scfsi[ch][scfsi_band]) {
                g->preflag = s->side.gr[0][ch].preflag;
                return;
            }
    if (g->block_type != 2 && g->preflag == 0) {
        over = 0;
        for (sfb = 17; sfb < 21; sfb++)
            if (xfsf[0][sfb] > xm->l[sfb]) over++;
        if (over == 4) {
            double ifqstep = sqrt(2.);
            g->preflag = 1;
            for (sfb = 0; sfb < (int) g->sfb_lmax; sfb++) {
                xm->l[sfb] *= pow(ifqstep, 2.0 * (double) PRETAB[sfb]);
                for (i = bl[sfb]; i < bl[sfb + 1]; i++) xr[i] *= pow(ifqstep, (double) PRETAB[sfb]);
            }
        }
    }
}

static int amp_scalefac_bands(mp3o_stream *s, double *xr, double xfsf[4][21], xmin_t *xm, int gr,
                              int ch, int iteration)
{ /* src/loop.c:1225-1350; scalefac_scale is always 0 so ifqstep = sqrt(2) */
    const int *bl = SFB_L[s->rate_idx], *bs = SFB_S[s->rate_idx];
    gr_info_t *g = &s->side.gr[gr][ch];
    int start, end, l, sfb, i, scfsi_band, over = 0, copySF = 0, preventSF = 0;
    double ifqstep = sqrt(2.0), ifqstep2;
    if (gr == 1)
        for (scfsi_band = 0; scfsi_band < 4; scfsi_band++)
            if (s->side.scfsi[ch][scfsi_band]) {
                if (iteration == 1) copySF = 1;
                else preventSF = 1;
                break;
            }
    ifqstep2 = ifqstep * ifqstep;
    scfsi_band = 0;
    for (sfb = 0; sfb < (int) g->sfb_lmax; sfb++) {
        if (copySF || preventSF) {
            if (sfb == SCFSI_BAND_L[scfsi_band + 1]) scfsi_band += 1;
            if (s->side.scfsi[ch][scfsi_band]) {
                if (copySF) s->scalefac_l[gr][ch][sfb] = s->scalefac_l[0][ch][sfb];
                continue;
            }
        }
        if (xfsf[0][sfb] > xm->l[sfb]) {
            over++;
            xm->l[sfb] *= ifqstep2;
            s->scalefac_l[gr][ch][sfb]++;
            start = bl[sfb];
            end = bl[sfb + 1];
            for (l = start; l < end; l++) xr[l] *= ifqstep;
        }
    }
    for (i = 0; i < 3; i++)
        for (sfb = (int) g->sfb_smax; sfb < 12; sfb++)
            if (xfsf[i + 1][sfb] > xm->s[sfb][i]) {
                over++;
                xm->s[sfb][i] *= ifqstep2;
                s->scalefac_s[gr][ch][sfb][i]++;
                start = bs[sfb];
                end = bs[sfb + 1];
                for (l = start; l < end; l++) xr[l * 3 + i] *= ifqstep;
            }
    return over;
}

static int loop_break(const mp3o_stream *s, const gr_info_t *g, int gr, int ch)
{ /* src/loop.c:1131-1152 */
    int i, sfb, temp = 1;
    for (sfb = 0; sfb < (int) g->sfb_lmax; sfb++)
        if (s->scalefac_l[gr][ch][sfb] == 0) temp = 0;
    for (sfb = (int) g->sfb_smax; sfb < 12; sfb++)
        for (i = 0; i < 3; i++)
            if (s->scalefac_s[gr][ch][sfb][i] == 0) temp = 0;
    return temp;
}

static int quantanf_init(const double *xr)
{ /* src/loop.c:369-402 */
    int i, tp = 0;
    double sfm, sum1 = 0.0, sum2 = 0.0;
    for (i = 0; i < 576; i++)
        if (xr[i] != 0) {
            double tpd = xr[i] * xr[i];
            sum1 += log(tpd);
            sum2 += tpd;
        }
    if (sum2 != 0.0) {
        sfm = exp(sum1 / 576.0) / (sum2 / 576.0);
        tp = r_nint(8.0 * log(sfm));
        if (tp < -100.0) tp = (int) -100.0;
    }
    return (int) (tp - 70.0);
}

static int outer_loop(mp3o_stream *s, double *xr, int max_bits, xmin_t *xm, int gr, int ch)
{ /* src/loop.c:415-558 */
    gr_info_t *g = &s->side.gr[gr][ch];
    int *ix = s->l3_enc[gr][ch];
    int scalesave_l[21], scalesave_s[13][3];
    int sfb, i, bits, huff_bits, save_preflag, save_compress, over, status, iteration = 0;
    double xfsf[4][21];
    CENSUS(CEN_GC, 1); CENSUS(CEN_SHORT_GC, g->window_switching_flag && g->block_type == 2);
    do {
        iteration += 1;
        CENSUS(CEN_OUTER, 1);
        g->part2_length = (unsigned) part2_length(s, gr, ch);
        huff_bits = max_bits - (int) g->part2_length;
        if (iteration == 1) bin_search_StepSize(s, max_bits, g->quantizerStepSize, ix, xr, g);
        bits = inner_loop(s, xr, ix, huff_bits, g);
        calc_noise(s, xr, ix, g, xfsf);
        for (sfb = 0; sfb < 21; sfb++) scalesave_l[sfb] = s->scalefac_l[gr][ch][sfb];
        for (sfb = 0; sfb < 13; sfb++)
            for (i = 0; i < 3; i++) scalesave_s[sfb][i] = s->scalefac_s[gr][ch][sfb][i];
        save_preflag = (int) g->preflag;
        save_compress = (int) g->scalefac_compress;
        preemphasis(s, xr, xfsf, xm, gr, ch);
        over = amp_scalefac_bands(s, xr, xfsf, xm, gr, ch, iteration);
        CENSUS(CEN_OUTER_NO_AMP, over == 0);
        if ((status = loop_break(s, g, gr, ch)) == 0) status = scale_bitcount(s, gr, ch);
    } while (status == 0 && over > 0);
    g->preflag = (unsigned) save_preflag;
    g->scalefac_compress = (unsigned) save_compress;
    for (sfb = 0; sfb < 21; sfb++) s->scalefac_l[gr][ch][sfb] = scalesave_l[sfb];
    for (i = 0; i < 3; i++)
        for (sfb = 0; sfb < 12; sfb++) s->scalefac_s[gr][ch][sfb][i] = scalesave_s[sfb][i];
    g->part2_length = (unsigned) part2_length(s, gr, ch);
    g->part2_3_length = g->part2_length + (unsigned) bits;
    return (int) g->part2_3_length;
}

static int ResvMaxBits(const mp3o_stream *s, double pe, int mean_bits)
{ /* src/reservoir.c:101-134 */
    int more_bits, max_bits, add_bits, over_bits;
    mean_bits /= s->channels;
    max_bits = mean_bits;
    if (max_bits > 4095) max_bits = 4095;
    if (s->ResvMax == 0) return max_bits;
    more_bits = (int) (pe * 3.1 - mean_bits);
    add_bits = 0;
    if (more_bits > 100) {
        int frac = (s->ResvSize * 6) / 10;
        add_bits = (frac < more_bits) ? frac : more_bits;
    }
    over_bits = s->ResvSize - ((s->ResvMax * 8) / 10) - add_bits;
    if (over_bits > 0) add_bits += over_bits;
    max_bits += add_bits;
    if (max_bits > 4095) max_bits = 4095;
    return max_bits;
}

static void ResvFrameEnd(mp3o_stream *s, int mean_bits)
{ /* src/reservoir.c:155-226 */
    int stereo = s->channels, gr, ch, stuffingBits, over_bits;
    if (stereo == 2 && (mean_bits & 1)) s->ResvSize += 1;
    over_bits = s->ResvSize - s->ResvMax;
    if (over_bits < 0) over_bits = 0;
    s->ResvSize -= over_bits;
    stuffingBits = over_bits;
    if ((over_bits = s->ResvSize % 8)) {
        stuffingBits += over_bits;
        s->ResvSize -= over_bits;
    }
    if (stuffingBits) {
        gr_info_t *g = &s->side.gr[0][0];
        if (g->part2_3_length + stuffingBits < 4095) g->part2_3_length += (unsigned) stuffingBits;
        else {
            for (gr = 0; gr < 2; gr++)
                for (ch = 0; ch < stereo; ch++) {
                    int extraBits, bitsThisGr;
                    g = &s->side.gr[gr][ch];
                    if (stuffingBits == 0) break;
                    extraBits = 4095 - (int) g->part2_3_length;
                    bitsThisGr = extraBits < stuffingBits ? extraBits : stuffingBits;
                    g->part2_3_length += (unsigned) bitsThisGr;
                    stuffingBits -= bitsThisGr;
                }
            s->side.resvDrain = stuffingBits;
        }
    }
}

static void iteration_loop(mp3o_stream *s, double pe[2][2], double xr_org[2][2][576],
                           double ratio_l[2][2][21], double ratio_s[2][2][12][3])
{ /* src/loop.c:232-362 */
    double xr[2][2][576];
    xmin_t xm;
    int gr, ch, sfb, i, max_bits, mean_bits = s->mean_bits;
    s->side.resvDrain = 0;
    memcpy(xr, xr_org, sizeof(xr));
    /* ResvFrameBegin, src/reservoir.c:45-93 */
    assert(s->side.main_data_begin * 8 == s->ResvSize);
    s->ResvMax = (s->bitsPerFrame > 7680) ? 0 : 7680 - s->bitsPerFrame;
    if (s->ResvMax > 4088) s->ResvMax = 4088;
    for (gr = 0; gr < 2; gr++)
        for (ch = 0; ch < s->channels; ch++) {
            gr_info_t *g = &s->side.gr[gr][ch];
            gr_deco(g);
            calc_xmin(s, xr[gr][ch], ratio_l[gr][ch], ratio_s[gr][ch], g, &xm);
            calc_scfsi(s, xr[gr][ch], &xm, ch, gr);
            max_bits = ResvMaxBits(s, pe[gr][ch], mean_bits);
            for (sfb = 0; sfb < 21; sfb++) s->scalefac_l[gr][ch][sfb] = 0;
            for (sfb = 0; sfb < 13; sfb++)
                for (i = 0; i < 3; i++) s->scalefac_s[gr][ch][sfb][i] = 0;
            g->part2_3_length = 0; g->big_values = 0; g->count1 = 0; g->scalefac_compress = 0;
            g->table_select[0] = g->table_select[1] = g->table_select[2] = 0;
            g->subblock_gain[0] = g->subblock_gain[1] = g->subblock_gain[2] = 0;
            g->region0_count = 0; g->region1_count = 0; g->part2_length = 0; g->preflag = 0;
            g->scalefac_scale = 0; g->quantizerStepSize = 0.0; g->count1table_select = 0;
            if (fabs(xr_max(xr[gr][ch], 0, 576)) != 0.0) {
                g->quantizerStepSize = (double) quantanf_init(xr[gr][ch]);
                g->part2_3_length = (unsigned) outer_loop(s, xr[gr][ch], max_bits, &xm, gr, ch);
            }
            s->ResvSize += (mean_bits / s->channels) - (int) g->part2_3_length; /* ResvAdjust */
            g->global_gain = (unsigned) r_nint(g->quantizerStepSize + 210.0);
            if (g->global_gain >= 256 && !s->ref_abort) s->ref_abort = MP3O_ABORT_GLOBAL_GAIN; /* assert, src/loop.c:358 */
        }
    ResvFrameEnd(s, mean_bits);
}

/* ------------------------------------------------------------------------- */
/* bitstream formatting (src/l3bitstream.c, src/formatBitstream.c)           */
/* ------------------------------------------------------------------------- */
typedef struct { uint8_t *p; int nbits; } bytebits_t;

static void bb_put(bytebits_t *b, unsigned val, int n)
{
    int j;
    for (j = n - 1; j >= 0; j--) {
        if ((val >> j) & 1u) b->p[b->nbits >> 3] |= (uint8_t) (0x80u >> (b->nbits & 7));
        b->nbits++;
    }
}

static void queue_side_info(mp3o_stream *s)
{ /* encodeSideInfo, src/l3bitstream.c:314-458 + store_side_info, src/formatBitstream.c:301 */
    si_entry_t *e;
    bytebits_t b;
    int gr, ch, i;
    if (s->q_len == s->q_cap) {
        int ncap = s->q_cap ? s->q_cap * 2 : 8, k;
        si_entry_t *nq = (si_entry_t *) calloc((size_t) ncap, sizeof(si_entry_t));
        for (k = 0; k < s->q_len; k++) nq[k] = s->queue[(s->q_head + k) % (s->q_cap ? s->q_cap : 1)];
        free(s->queue);
        s->queue = nq;
        s->q_cap = ncap;
        s->q_head = 0;
    }
    e = &s->queue[(s->q_head + s->q_len) % s->q_cap];
    s->q_len++;
    memset(e, 0, sizeof(*e));
    b.p = e->bytes;
    b.nbits = 0;
    bb_put(&b, 0xfff, 12);
    bb_put(&b, 1, 1);                       /* version: MPEG-1 */
    bb_put(&b, 4 - 3, 2);                   /* layer III */
    bb_put(&b, s->crc ? 0u : 1u, 1);        /* !error_protection (src/l3bitstream.c:325) */
    bb_put(&b, (unsigned) s->bitrate_index, 4);
    bb_put(&b, (unsigned) s->rate_idx, 2);
    bb_put(&b, 0, 1);                       /* padding: never (src/musicin.c:566-581) */
    bb_put(&b, 0, 1);                       /* extension: uninitialised in the reference, observed 0 */
    bb_put(&b, (unsigned) s->mode, 2);
    bb_put(&b, 0, 2);                       /* mode_ext */
    bb_put(&b, (unsigned) s->copyright, 1);
    bb_put(&b, (unsigned) s->original, 1);
    bb_put(&b, (unsigned) s->emphasis, 2);
    if (s->crc) bb_put(&b, 0, 16);          /* the CRC word, never computed for Layer III (src/l3bitstream.c:312, 338-342) */
    bb_put(&b, (unsigned) s->side.main_data_begin, 9);
    bb_put(&b, s->side.private_bits, s->channels == 2 ? 3 : 5);
    for (ch = 0; ch < s->channels; ch++)
        for (i = 0; i < 4; i++) bb_put(&b, s->side.scfsi[ch][i], 1);
    for (gr = 0; gr < 2; gr++)
        for (ch = 0; ch < s->channels; ch++) {
            const gr_info_t *g = &s->side.gr[gr][ch];
            bb_put(&b, g->part2_3_length, 12);
            bb_put(&b, g->big_values, 9);
            bb_put(&b, g->global_gain, 8);
            bb_put(&b, g->scalefac_compress, 4);
            bb_put(&b, g->window_switching_flag, 1);
            if (g->window_switching_flag) {
                bb_put(&b, g->block_type, 2);
                bb_put(&b, g->mixed_block_flag, 1);
                for (i = 0; i < 2; i++) bb_put(&b, g->table_select[i], 5);
                for (i = 0; i < 3; i++) bb_put(&b, (unsigned) g->subblock_gain[i], 3);
            } else {
                assert(g->block_type == 0);
                for (i = 0; i < 3; i++) bb_put(&b, g->table_select[i], 5);
                bb_put(&b, g->region0_count, 4);
                bb_put(&b, g->region1_count, 3);
            }
            bb_put(&b, g->preflag, 1);
            bb_put(&b, g->scalefac_scale, 1);
            bb_put(&b, g->count1table_select, 1);
        }
    e->frameLength = s->bitsPerFrame;
    e->SILength = b.nbits;
    assert(b.nbits == 32 + (s->channels == 2 ? 256 : 136) + (s->crc ? 16 : 0));
}

static int write_side_info(mp3o_stream *s)
{ /* src/formatBitstream.c:250-270 + get_side_info :369 */
    si_entry_t *e;
    int i;
    if (s->q_len <= 0) { /* get_side_info's assert( l ), src/formatBitstream.c:390: the reference dies here */
        if (!s->ref_abort) s->ref_abort = MP3O_ABORT_FLUSH_SLOT;
        return 0;
    }
    e = &s->queue[s->q_head];
    s->q_head = (s->q_head + 1) % s->q_cap;
    s->q_len--;
    s->ThisFrameSize = e->frameLength;
    for (i = 0; i < e->SILength / 8; i++) put_bits(s, e->bytes[i], 8);
    return e->SILength;
}

static void main_bits(mp3o_stream *s, unsigned val, unsigned nbits)
{ /* WriteMainDataBits, src/formatBitstream.c:218-247 */
    assert(nbits <= 32);
    if (s->BitCount == s->ThisFrameSize) {
        s->BitCount = write_side_info(s);
        s->BitsRemaining = s->ThisFrameSize - s->BitCount;
    }
    if (nbits == 0) return;
    if ((int) nbits > s->BitsRemaining) {
        unsigned extra = val >> (nbits - (unsigned) s->BitsRemaining);
        nbits -= (unsigned) s->BitsRemaining;
        put_bits(s, extra, s->BitsRemaining);
        s->BitCount = write_side_info(s);
        s->BitsRemaining = s->ThisFrameSize - s->BitCount;
        put_bits(s, val, (int) nbits);
    } else
        put_bits(s, val, (int) nbits);
    s->BitCount += (int) nbits;
    s->BitsRemaining -= (int) nbits;
    assert(s->BitCount <= s->ThisFrameSize && s->BitsRemaining >= 0);
}

/* BF_addEntry drops zero-length elements (src/formatBitstream.c:536-548) */
#define MAIN_ENTRY(s, v, n) do { if ((n) != 0) main_bits((s), (unsigned) (v), (unsigned) (n)); } while (0)

static int emit_pair(mp3o_stream *s, unsigned table, int x, int y)
{ /* HuffmanCode, src/huffcode.h:16-139 */
    unsigned signx = 0, signy = 0, linbitsx = 0, linbitsy = 0, linbits, ylen, idx, code, ext = 0, e;
    int cbits = 0, xbits = 0;
    if (table == 0) return 0;
    if (x < 0) { x = -x; signx = 1; }
    if (y < 0) { y = -y; signy = 1; }
    ylen = T_HT_YLEN[table];
    linbits = T_HT_LINBITS[table];
    if (table > 15) {
        if (x > 14) { linbitsx = (unsigned) x - 15; x = 15; }
        if (y > 14) { linbitsy = (unsigned) y - 15; y = 15; }
        idx = (unsigned) x * ylen + (unsigned) y;
        e = T_HT_PACKED[T_HT_OFF[table] + idx];
        code = e >> 8;
        cbits = (int) (e & 0xff);
        if (x > 14) { ext |= linbitsx; xbits += (int) linbits; }
        if (x != 0) { ext <<= 1; ext |= signx; xbits += 1; }
        if (y > 14) { ext <<= linbits; ext |= linbitsy; xbits += (int) linbits; }
        if (y != 0) { ext <<= 1; ext |= signy; xbits += 1; }
    } else {
        idx = (unsigned) x * ylen + (unsigned) y;
        e = T_HT_PACKED[T_HT_OFF[table] + idx];
        code = e >> 8;
        cbits = (int) (e & 0xff);
        if (x != 0) { code <<= 1; code |= signx; cbits += 1; }
        if (y != 0) { code <<= 1; code |= signy; cbits += 1; }
    }
    if (cbits) main_bits(s, code, (unsigned) cbits);
    if (xbits) main_bits(s, ext, (unsigned) xbits);
    return cbits + xbits;
}

static void huffman_code_bits(mp3o_stream *s, const int *ix, const gr_info_t *g)
{ /* Huffmancodebits, src/l3bitstream.c:516-716 */
    const int *bl = SFB_L[s->rate_idx], *bs = SFB_S[s->rate_idx];
    int bigvalues = (int) g->big_values * 2, bitsWritten = 0, i, count1End, stuffingBits;
    if (bigvalues) {
        if (!g->mixed_block_flag && g->window_switching_flag && g->block_type == 2) {
            int sfb, window, line;
            for (sfb = 0; sfb < 13; sfb++) {
                int start = bs[sfb], end = bs[sfb + 1];
                unsigned t = (start < 12) ? g->table_select[0] : g->table_select[1];
                for (window = 0; window < 3; window++)
                    for (line = start; line < end; line += 2)
                        bitsWritten += emit_pair(s, t, ix[line * 3 + window], ix[(line + 1) * 3 + window]);
            }
        } else {
            int region1Start = bl[g->region0_count + 1];
            int region2Start = bl[g->region0_count + 1 + g->region1_count + 1];
            for (i = 0; i < bigvalues; i += 2) {
                unsigned t;
                if (i < region1Start) t = g->table_select[0];
                else if (i < region2Start) t = g->table_select[1];
                else t = g->table_select[2];
                if (t) bitsWritten += emit_pair(s, t, ix[i], ix[i + 1]);
            }
        }
    }
    count1End = bigvalues + (int) g->count1 * 4;
    for (i = bigvalues; i < count1End; i += 4) { /* L3_huffman_coder_count1 :727-767 */
        int q[4], k, p;
        unsigned sg[4], e;
        for (k = 0; k < 4; k++) {
            q[k] = ix[i + k];
            if (q[k] > 0) sg[k] = 0;
            else { q[k] = -q[k]; sg[k] = 1; }
        }
        p = q[0] + (q[1] << 1) + (q[2] << 2) + (q[3] << 3);
        e = T_HT_PACKED[T_HT_OFF[32 + g->count1table_select] + p];
        MAIN_ENTRY(s, e >> 8, e & 0xff);
        bitsWritten += (int) (e & 0xff);
        for (k = 0; k < 4; k++)
            if (q[k]) { main_bits(s, sg[k], 1); bitsWritten += 1; }
    }
    if ((stuffingBits = (int) g->part2_3_length - (int) g->part2_length - bitsWritten)) {
        int words = stuffingBits / 32, rem = stuffingBits % 32;
        assert(stuffingBits > 0);
        while (words--) main_bits(s, ~0u, 32);
        if (rem) main_bits(s, ~0u, (unsigned) rem);
    }
}

static void format_frame(mp3o_stream *s, double xr[2][2][576])
{ /* III_format_bitstream, src/l3bitstream.c:67-162 + BF_BitstreamFrame, src/formatBitstream.c:52-80 */
    int gr, ch, i, sfb, window, k, fwdFrame = 0, fwdSI = 0;
    for (gr = 0; gr < 2; gr++)
        for (ch = 0; ch < s->channels; ch++)
            for (i = 0; i < 576; i++)
                if (xr[gr][ch][i] < 0 && s->l3_enc[gr][ch][i] > 0) s->l3_enc[gr][ch][i] *= -1;
    queue_side_info(s);
    for (gr = 0; gr < 2; gr++) /* encodeMainData :174-310 via main_data :194 */
        for (ch = 0; ch < s->channels; ch++) {
            const gr_info_t *g = &s->side.gr[gr][ch];
            int slen1 = SLEN1[g->scalefac_compress], slen2 = SLEN2[g->scalefac_compress];
            if (g->window_switching_flag == 1 && g->block_type == 2) {
                for (sfb = 0; sfb < 6; sfb++)
                    for (window = 0; window < 3; window++) MAIN_ENTRY(s, s->scalefac_s[gr][ch][sfb][window], slen1);
                for (sfb = 6; sfb < 12; sfb++)
                    for (window = 0; window < 3; window++) MAIN_ENTRY(s, s->scalefac_s[gr][ch][sfb][window], slen2);
            } else {
                if (gr == 0 || s->side.scfsi[ch][0] == 0)
                    for (sfb = 0; sfb < 6; sfb++) MAIN_ENTRY(s, s->scalefac_l[gr][ch][sfb], slen1);
                if (gr == 0 || s->side.scfsi[ch][1] == 0)
                    for (sfb = 6; sfb < 11; sfb++) MAIN_ENTRY(s, s->scalefac_l[gr][ch][sfb], slen1);
                if (gr == 0 || s->side.scfsi[ch][2] == 0)
                    for (sfb = 11; sfb < 16; sfb++) MAIN_ENTRY(s, s->scalefac_l[gr][ch][sfb], slen2);
                if (gr == 0 || s->side.scfsi[ch][3] == 0)
                    for (sfb = 16; sfb < 21; sfb++) MAIN_ENTRY(s, s->scalefac_l[gr][ch][sfb], slen2);
            }
            huffman_code_bits(s, s->l3_enc[gr][ch], g);
        }
    if (s->side.resvDrain) { /* drain_into_ancillary_data :492-509 */
        int words = s->side.resvDrain / 32, rem = s->side.resvDrain % 32;
        for (i = 0; i < words; i++) main_bits(s, 0, 32);
        if (rem) main_bits(s, 0, (unsigned) rem);
    }
    assert(s->BitsRemaining % 8 == 0);
    for (k = 0; k < s->q_len; k++) {
        const si_entry_t *e = &s->queue[(s->q_head + k) % s->q_cap];
        fwdFrame += e->frameLength;
        fwdSI += e->SILength;
    }
    s->side.main_data_begin = s->BitsRemaining / 8 + fwdFrame / 8 - fwdSI / 8;
}

/* ------------------------------------------------------------------------- */
/* public interface                                                          */
/* ------------------------------------------------------------------------- */
mp3o_stream *mp3o_open(int rate_hz, int kbps, int channels)
{
    static const double s_freq[3] = {44.1, 48, 32}; /* src/common.c:113 */
    mp3o_stream *s;
    int ri, bi, whole_SpF;
    if (rate_hz == 44100) ri = 0;
    else if (rate_hz == 48000) ri = 1;
    else if (rate_hz == 32000) ri = 2;
    else return NULL;
    for (bi = 1; bi < 15; bi++)
        if (BITRATES[bi] == kbps) break;
    if (bi == 15 || (channels != 1 && channels != 2)) return NULL;
    s = (mp3o_stream *) calloc(1, sizeof(*s));
    s->rate_idx = ri; s->rate_hz = rate_hz; s->kbps = kbps; s->bitrate_index = bi;
    s->channels = channels;
    s->mode = (channels == 1) ? 3 : 0;
    /* src/musicin.c:562-566, 723-746 */
    whole_SpF = (int) (((double) 1152 / s_freq[ri]) * ((double) kbps / 8.0));
    s->bitsPerFrame = 8 * whole_SpF;
    s->mean_bits = (s->bitsPerFrame - (32 + (channels == 1 ? 136 : 256))) / 2;
    s->T = make_tables(ri);
    s->age_new = 0; s->age_old = 1; s->age_oldest = 0; /* src/l3psy.c:73 */
    return s;
}

int mp3o_set_options(mp3o_stream *s, int mode, int error_protection, int copyright, int original, int emphasis)
{ /* the driver's -m -e -c -o -d (src/musicin.c:226-275); before the first frame */
    if (mode == 1) return -1; /* joint stereo: refused for Layer III (src/musicin.c:548-552) */
    if ((s->channels == 1) != (mode == 3)) return -1;
    s->mode = mode;
    s->crc = error_protection != 0;
    s->copyright = copyright != 0;
    s->original = original != 0;
    s->emphasis = emphasis & 3;
    /* src/musicin.c:728-746 */
    s->mean_bits = (s->bitsPerFrame - (32 + (s->channels == 1 ? 136 : 256) + (s->crc ? 16 : 0))) / 2;
    return 0;
}

int mp3o_ref_abort(const mp3o_stream *s) { return s->ref_abort; }

void mp3o_close(mp3o_stream *s)
{
    if (!s) return;
    free_tables(s->T);
    free(s->queue);
    free(s->out);
    free(s);
}

void mp3o_encode_frame(mp3o_stream *s, const int16_t pcm[2][1152], stage_dump_t *d)
{ /* the Layer III case of the frame loop, src/musicin.c:708-788 */
    double pe[2][2], ratio_l[2][2][21], ratio_s[2][2][12][3], xr[2][2][576];
    int gr, ch, j, i, k;
    memset(pe, 0, sizeof(pe)); memset(ratio_l, 0, sizeof(ratio_l)); memset(ratio_s, 0, sizeof(ratio_s));
    memset(xr, 0, sizeof(xr));
    if (d) {
        memset(d, 0, sizeof(*d));
        d->magic = STAGE_DUMP_MAGIC;
    }
    for (gr = 0; gr < 2; gr++)
        for (ch = 0; ch < s->channels; ch++) {
            psy_granule(s, &pcm[ch][gr * 576], ch, ratio_l[gr][ch], ratio_s[gr][ch], &pe[gr][ch],
                        &s->side.gr[gr][ch]);
            if (d) d->psy_block_type[gr][ch] = (int32_t) s->side.gr[gr][ch].block_type;
        }
    for (gr = 0; gr < 2; gr++)
        for (ch = 0; ch < s->channels; ch++)
            for (j = 0; j < 18; j++)
                filterbank_slot(s, &pcm[ch][gr * 576 + j * 32], ch, s->sb[ch][gr + 1][j]);
    if (d) {
        memcpy(d->pe, pe, sizeof(pe));
        memcpy(d->ratio_l, ratio_l, sizeof(ratio_l));
        memcpy(d->ratio_s, ratio_s, sizeof(ratio_s));
        for (ch = 0; ch < s->channels; ch++)
            for (gr = 0; gr < 2; gr++) memcpy(d->sb_sample[ch][gr], s->sb[ch][gr + 1], sizeof(d->sb_sample[0][0]));
    }
    mdct_frame(s, xr);
    if (d) memcpy(d->xr, xr, sizeof(xr));
    iteration_loop(s, pe, xr, ratio_l, ratio_s);
    if (d) {
        for (gr = 0; gr < 2; gr++)
            for (ch = 0; ch < s->channels; ch++) {
                const gr_info_t *g = &s->side.gr[gr][ch];
                memcpy(d->l3_enc[gr][ch], s->l3_enc[gr][ch], sizeof(s->l3_enc[0][0]));
                d->gi[gr][ch].part2_3_length = (int32_t) g->part2_3_length;
                d->gi[gr][ch].big_values = (int32_t) g->big_values;
                d->gi[gr][ch].count1 = (int32_t) g->count1;
                d->gi[gr][ch].global_gain = (int32_t) g->global_gain;
                d->gi[gr][ch].scalefac_compress = (int32_t) g->scalefac_compress;
                d->gi[gr][ch].window_switching_flag = (int32_t) g->window_switching_flag;
                d->gi[gr][ch].block_type = (int32_t) g->block_type;
                d->gi[gr][ch].mixed_block_flag = (int32_t) g->mixed_block_flag;
                for (k = 0; k < 3; k++) {
                    d->gi[gr][ch].table_select[k] = (int32_t) g->table_select[k];
                    d->gi[gr][ch].subblock_gain[k] = g->subblock_gain[k];
                }
                d->gi[gr][ch].region0_count = (int32_t) g->region0_count;
                d->gi[gr][ch].region1_count = (int32_t) g->region1_count;
                d->gi[gr][ch].preflag = (int32_t) g->preflag;
                d->gi[gr][ch].scalefac_scale = (int32_t) g->scalefac_scale;
                d->gi[gr][ch].count1table_select = (int32_t) g->count1table_select;
                d->gi[gr][ch].part2_length = (int32_t) g->part2_length;
                for (i = 0; i < 22; i++) d->scalefac_l[gr][ch][i] = s->scalefac_l[gr][ch][i];
                for (i = 0; i < 13; i++)
                    for (k = 0; k < 3; k++) d->scalefac_s[gr][ch][i][k] = s->scalefac_s[gr][ch][i][k];
            }
        d->main_data_begin = s->side.main_data_begin;
        d->resvDrain = s->side.resvDrain;
        for (ch = 0; ch < s->channels; ch++)
            for (i = 0; i < 4; i++) d->scfsi[ch][i] = (int32_t) s->side.scfsi[ch][i];
    }
    format_frame(s, xr);
}

void mp3o_flush(mp3o_stream *s)
{ /* BF_FlushBitstream, src/formatBitstream.c:87-120, then close_bit_stream_w, src/common.c:968 */
    int k, fwdFrame = 0, fwdSI = 0;
    if (s->closed) return;
    for (k = 0; k < s->q_len; k++) {
        const si_entry_t *e = &s->queue[(s->q_head + k) % s->q_cap];
        fwdFrame += e->frameLength;
        fwdSI += e->SILength;
    }
    if (s->q_len) {
        int bitsRemaining = fwdFrame - fwdSI, words = bitsRemaining / 32;
        while (words--) main_bits(s, 0, 32);
        main_bits(s, 0, (unsigned) (bitsRemaining % 32));
    }
    /* empty_buffer(bs, buf_byte_idx) also writes the byte under construction (src/common.c:843-868),
       so the file carries one byte beyond the last complete one */
    put_bits(s, 0, 0);
    s->out_len = (s->out_bits >> 3) + 1;
    s->closed = 1;
}

const uint8_t *mp3o_output(const mp3o_stream *s, size_t *len)
{
    *len = s->closed ? s->out_len : (s->out_bits >> 3);
    return s->out;
}

size_t mp3o_encode_pcm(int rate_hz, int kbps, int channels, const int16_t *pcm, size_t n_total,
                       uint8_t **out, stage_dump_t *dumps, int max_dumps)
{
    int ab = 0;
    size_t n = mp3o_encode_pcm_ex(rate_hz, kbps, channels, NULL, pcm, n_total, out, dumps, max_dumps, &ab);
    if (ab) { /* callers of the plain entry do not expect inputs the reference dies on */
        fprintf(stderr, "mp3_oracle: the reference aborts on this input (MP3O_ABORT_* = %d)\n", ab);
        abort();
    }
    return n;
}

size_t mp3o_encode_pcm_ex(int rate_hz, int kbps, int channels, const char *mode, const int16_t *pcm, size_t n_total,
                          uint8_t **out, stage_dump_t *dumps, int max_dumps, int *ref_abort)
{
    mp3o_stream *s = mp3o_open(rate_hz, kbps, channels);
    size_t per_frame = (size_t) 1152 * (size_t) channels, pos = 0, len;
    int16_t buf[2][1152];
    int frame = 0, j;
    const uint8_t *p;
    *out = NULL;
    *ref_abort = 0;
    if (!s) return 0;
    if (mode && mode[0]) { /* as oracle/ref_harness.c: the -m letter, then e / c / o */
        const int m = mode[0] == 'm' ? 3 : mode[0] == 'd' ? 2 : mode[0] == 'j' ? 1 : 0;
        if (mp3o_set_options(s, m, strchr(mode + 1, 'e') != NULL, strchr(mode + 1, 'c') != NULL, strchr(mode + 1, 'o') != NULL, 0)) {
            mp3o_close(s);
            return 0;
        }
    }
    while (pos < n_total) { /* get_audio / read_samples, src/encode.c:123-256 */
        size_t n = n_total - pos < per_frame ? n_total - pos : per_frame;
        memset(buf, 0, sizeof(buf));
        for (j = 0; j < 1152; j++) {
            if (channels == 2) {
                buf[0][j] = ((size_t) (2 * j) < n) ? pcm[pos + 2 * j] : 0;
                buf[1][j] = ((size_t) (2 * j + 1) < n) ? pcm[pos + 2 * j + 1] : 0;
            } else
                buf[0][j] = ((size_t) j < n) ? pcm[pos + j] : 0;
        }
        pos += n;
        mp3o_encode_frame(s, (const int16_t (*)[1152]) buf, (dumps && frame < max_dumps) ? &dumps[frame] : NULL);
        if (dumps && frame < max_dumps) dumps[frame].frame_index = frame;
        if (s->ref_abort) break; /* the reference is dead: nothing after this frame exists */
        frame++;
    }
    if (!s->ref_abort) mp3o_flush(s);
    if (s->ref_abort) {
        *ref_abort = s->ref_abort | (frame << 8); /* the frame it died in (the number of frames: in the final flush) */
        *out = (uint8_t *) malloc(1);
        mp3o_close(s);
        return 0;
    }
    p = mp3o_output(s, &len);
    *out = (uint8_t *) malloc(len ? len : 1);
    memcpy(*out, p, len);
    mp3o_close(s);
    return len;
}

/* The whole stream again, for the FFT seam only (fft_seam.h): seams[] receives one record per L3psycho_anal call of the
 * first max_frames frames, in call order ([frame][gr][ch]).  Returns the number of records written. */
long mp3o_encode_pcm_fft_seam(int rate_hz, int kbps, int channels, const int16_t *pcm, size_t n_total, fft_seam_t *seams, int max_frames)
{
    mp3o_stream *s = mp3o_open(rate_hz, kbps, channels);
    size_t per_frame = (size_t) 1152 * (size_t) channels, pos = 0;
    int16_t buf[2][1152];
    int frame = 0, j;
    const long want = (long) max_frames * 2 * channels;
    if (!s || !seams || max_frames <= 0) { mp3o_close(s); return 0; }
    s->fft_seam = seams;
    s->fft_seam_left = want;
    while (pos < n_total && frame < max_frames && !s->ref_abort) {
        size_t n = n_total - pos < per_frame ? n_total - pos : per_frame;
        memset(buf, 0, sizeof(buf));
        for (j = 0; j < 1152; j++) {
            if (channels == 2) {
                buf[0][j] = ((size_t) (2 * j) < n) ? pcm[pos + 2 * j] : 0;
                buf[1][j] = ((size_t) (2 * j + 1) < n) ? pcm[pos + 2 * j + 1] : 0;
            } else
                buf[0][j] = ((size_t) j < n) ? pcm[pos + j] : 0;
        }
        pos += n;
        mp3o_encode_frame(s, (const int16_t (*)[1152]) buf, NULL);
        frame++;
    }
    {
        const long left = s->fft_seam ? s->fft_seam_left : 0;
        mp3o_close(s);
        return want - left;
    }
}

/* Layers I and II: the same translation unit, because they share the window, the FFT, the
 * filterbank and the bit writer above -- as the reference's layers do */
#include "mp12_oracle.inc"
