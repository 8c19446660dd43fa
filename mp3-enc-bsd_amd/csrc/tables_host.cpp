// Host-side construction of the read-only table block (mp3mi_tables).
//
// Every value that the reference computes once at start-up with libm is computed the same way with glibc's libm --
// but NOT on the machine that encodes: the expressions below marked LIBM_TABLE are compiled only into the generator
// (-DMP3MI_TABLE_GEN, `make blob`), which runs in the environment the golden vectors come from and writes their values,
// bit for bit, into tables_blob.bin; the library links that file (build/tables_blob.o) and copies the values out of it.
// The product therefore calls no transcendental of the host's libm at all, and a host with another libm emits the
// same stream (tests/test_table_pins.py::test_tables_do_not_depend_on_this_hosts_libm).  What comes out of libm:
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
#include <mutex>
#include <vector>
#include "mp3mi_host.h"
#include "mp3mi_tables_gen.h"
#include "mdct_shape.h"
#include "l12_dev.h"
#include "mp3mi_tables_l12_gen.h"

#if defined(MP3MI_FFT_INFO) /* -DMP3MI_FFT_INFO: print the round headers the generator produces (after changing it) */
#define MP3MI_FFT_INFO_ON 1
#else
#define MP3MI_FFT_INFO_ON 0
#endif
#define R_PI 3.14159265358979
#define R_LN_TO_LOG10 0.2302585093
#define R_TWOPI 6.28318530717958647692

/* ---- tables_blob.bin: the values that come out of libm, as the reference environment computes them ----
 * "MP3MITB1", n entries {id, rate index or -1 for all rates, offset, size}, the data, FNV-1a 64 of all before. */
enum {
    TB_WINDOW = 1, TB_WINDOW_S, TB_S3_L, TB_EXP_SNR_S, TB_FILT, TB_MDCT_WIN, TB_COS_S, TB_COS_L, TB_CA, TB_CS,
    TB_POW_NINT, TB_POW43, TB_STEP, TB_PRETAB_XR, TB_PRETAB_XMIN, TB_SQRT2, TB_LOG2,
    TB_L12_SPREAD, /* Layers I / II: the spreading function of src/psy.c:200-216 */
    TB_TWIDDLE = 64 /* + 2 * logm + three */
};
struct tb_entry { uint32_t id; int32_t rate; uint32_t offset, size; };
struct tb_header { char magic[8]; uint32_t n_entries, pad; };

static uint64_t tb_fnv(const void *p, size_t n)
{
    const unsigned char *b = (const unsigned char *) p;
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001b3ull; }
    return h;
}

#if defined(MP3MI_TABLE_GEN)
/* generator: compute with libm, record */
static std::vector<tb_entry> tb_entries;
static std::vector<unsigned char> tb_data;
static void tb_put(uint32_t id, int rate, const void *src, size_t bytes)
{
    for (const tb_entry &e : tb_entries)
        if (e.id == id && e.rate == rate) return; /* (the tables of a rate are built more than once) */
    tb_entry e = {id, rate, (uint32_t) tb_data.size(), (uint32_t) bytes};
    tb_entries.push_back(e);
    tb_data.insert(tb_data.end(), (const unsigned char *) src, (const unsigned char *) src + bytes);
    while (tb_data.size() % 8) tb_data.push_back(0);
}
#define LIBM_TABLE(id, rate, dst, bytes, ...) do { __VA_ARGS__; tb_put((id), (rate), (dst), (bytes)); } while (0)
#else
/* product: copy out of the linked blob */
extern "C" const unsigned char _binary_tables_blob_bin_start[], _binary_tables_blob_bin_end[];
/* Threads: table builds are serialised (g_tables_mutex, taken by the three entry points at the end of this file), so
 * the statics here and in the generator code below have one user at a time; the blob's checksum is verified once per
 * process (std::call_once). */
static int tb_failed = 0; /* some entry was missing or the blob is damaged: build_tables_unpinned reports it */
static int tb_get(uint32_t id, int rate, void *dst, size_t bytes)
{
    const unsigned char *b = _binary_tables_blob_bin_start;
    const size_t total = (size_t) (_binary_tables_blob_bin_end - _binary_tables_blob_bin_start);
    static std::once_flag checked_once;
    static int checked = -1; /* 1 good, -1 bad */
    std::call_once(checked_once, [&]() {
        uint64_t h = 0;
        if (total > sizeof(tb_header) + 8 && !memcmp(b, "MP3MITB1", 8)) {
            memcpy(&h, b + total - 8, 8);
            if (h == tb_fnv(b, total - 8)) checked = 1;
        }
        if (checked < 0) fprintf(stderr, "mp3mi: the linked table blob (csrc/tables_blob.bin) is damaged\n");
    });
    if (checked < 0) { tb_failed = 1; return -1; }
    tb_header hd;
    memcpy(&hd, b, sizeof(hd));
    for (uint32_t i = 0; i < hd.n_entries; i++) {
        tb_entry e;
        memcpy(&e, b + sizeof(hd) + (size_t) i * sizeof(e), sizeof(e));
        if (e.id != id || (e.rate != rate && e.rate != -1)) continue;
        if (e.size != bytes || (size_t) e.offset + e.size > total) break;
        memcpy(dst, b + sizeof(hd) + (size_t) hd.n_entries * sizeof(tb_entry) + e.offset, bytes);
        return 0;
    }
    fprintf(stderr, "mp3mi: table blob has no entry %u for rate index %d of %zu bytes -- regenerate it (make -C csrc blob)\n", id, rate, bytes);
    tb_failed = 1;
    return -1;
}
#define LIBM_TABLE(id, rate, dst, bytes, ...) do { if (tb_get((id), (rate), (dst), (bytes))) return -7; } while (0)
#endif

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

struct Twiddle { std::vector<float> t; int nel; };

Twiddle make_twiddle(int logm, bool three)
{
    int m = 1 << logm, m4 = m / 4, m8 = m / 8, nel = m4 - 2, e = 0;
    Twiddle tw;
    tw.nel = nel;
    tw.t.assign((size_t) (three ? 6 : 3) * (nel > 0 ? nel : 1), 0.0f);
#if !defined(MP3MI_TABLE_GEN)
    (void) tb_get((uint32_t) (TB_TWIDDLE + 2 * logm + (three ? 1 : 0)), -1, tw.t.data(), tw.t.size() * sizeof(float)); /* (a failure is remembered: tb_failed) */
    (void) m8; (void) e;
    return tw;
#else
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
    tb_put((uint32_t) (TB_TWIDDLE + 2 * logm + (three ? 1 : 0)), -1, tw.t.data(), tw.t.size() * sizeof(float));
    return tw;
#endif
}

/* One fused butterfly of the flattened FFT.  cls 0 ("R", four operands a b c d):
 *     s1 = a + b; s2 = c + d; u1 = a - b; u2 = c - d (negated if neg);  a <- s1; c <- s2; (b, d) <- twiddle(u1, u2)
 * which is steps 1-4 of rsrec for one n (src/subs.c:465-498: a = x[n], b = x[n+m/2], c = x[n+m/4], d = x[n+3m/4]),
 * and, without neg and twiddle, the two length-2 butterflies of a complex block of size 2 (src/subs.c:243-250).
 * cls 1 ("C", eight operands): steps 1-4 of srrec for one n (src/subs.c:288-342).
 * kind: 0 none, 1 twiddle rotation, 2 SQHALF rotation. */
struct FusedOp {
    int cls, kind, neg;
    int logm; /* the block it belongs to has 2^logm points (k_fft.hip runs the blocks of 256 points and more in registers) */
    int id; /* window * (butterflies of the rank) + index within the rank: names the butterfly in a stored placement */
    unsigned p[8];
    float tw[6];
};

/* a butterfly in the dependency DAG of a program (FftGen::build) */
struct FftNode { FusedOp o; int lp[8]; std::vector<int> succ, pred; int npred, height; bool done; };
/* a program's schedule: per step (a wave barrier after each) the butterflies of class 0 and of class 1 in lane order */
typedef std::vector<std::vector<int> > FftSchedule; /* [2 * step + class] -> node ids */

struct FftGen {
    Twiddle tw_rs[11], tw_sr[11];
    std::vector<std::vector<FusedOp> > rank_ops; /* [rank] */
    struct Post { int type; unsigned a, b; };
    std::vector<Post> post1, post2;
    const std::vector<uint16_t> *(*stored_order)(int logN, int rank, int cls) = NULL;
    void (*schedule_hook)(int logN, const std::vector<FftNode> &nodes, FftSchedule &sched) = NULL; /* the offline search */
    int leaf_cycles = 0; /* of the last build: LDS cycles of a 16-byte access to every lane's run A plus one to its run B (fft_leaves) */

    void add(int rank, const FusedOp &o)
    {
        if ((int) rank_ops.size() <= rank) rank_ops.resize((size_t) rank + 1);
        rank_ops[(size_t) rank].push_back(o);
    }

    void cplx(unsigned xr, unsigned xi, int logm, int rank)
    {
        if (logm <= 0) return;
        FusedOp o;
        memset(&o, 0, sizeof(o));
        if (logm == 1) {
            o.cls = 0;
            o.logm = 1;
            o.p[0] = xr; o.p[1] = xr + 1; o.p[2] = xi; o.p[3] = xi + 1;
            add(rank, o);
            return;
        }
        const int m = 1 << logm, m2 = m / 2, m4 = m2 / 2, m8 = m4 / 2;
        const Twiddle &tw = tw_sr[logm];
        for (int n = 0, e = 0; n < m4; n++) {
            memset(&o, 0, sizeof(o));
            o.cls = 1;
            o.logm = logm;
            o.p[0] = xr + n; o.p[1] = xr + n + m2; o.p[2] = xr + n + m4; o.p[3] = xr + n + m4 + m2;
            o.p[4] = xi + n; o.p[5] = xi + n + m2; o.p[6] = xi + n + m4; o.p[7] = xi + n + m4 + m2;
            if (n == 0) o.kind = 0;
            else if (n == m8) o.kind = 2;
            else {
                o.kind = 1;
                for (int k = 0; k < 6; k++) o.tw[k] = tw.t[(size_t) k * tw.nel + e];
                e++;
            }
            add(rank, o);
        }
        cplx(xr, xi, logm - 1, rank + 1);
        cplx(xr + m2, xi + m2, logm - 2, rank + 1);
        cplx(xr + 3 * (m / 4), xi + 3 * (m / 4), logm - 2, rank + 1);
    }

    void real(unsigned o0, int logm, int rank, unsigned dummy)
    {
        if (logm <= 0) return;
        FusedOp o;
        memset(&o, 0, sizeof(o));
        if (logm == 1) {
            o.cls = 0;
            o.logm = 1;
            o.p[0] = o0; o.p[1] = o0 + 1; o.p[2] = dummy; o.p[3] = dummy;
            add(rank, o);
            return;
        }
        const int m = 1 << logm, m2 = m / 2, m4 = m2 / 2, m8 = m4 / 2;
        const Twiddle &tw = tw_rs[logm];
        for (int n = 0, e = 0; n < m4; n++) {
            memset(&o, 0, sizeof(o));
            o.cls = 0;
            o.logm = logm;
            o.neg = 1; /* step 2, src/subs.c:475-479 */
            o.p[0] = o0 + n; o.p[1] = o0 + n + m2; o.p[2] = o0 + n + m4; o.p[3] = o0 + n + m4 + m2;
            if (n == 0) o.kind = 0;
            else if (n == m8) o.kind = 2;
            else {
                o.kind = 1;
                for (int k = 0; k < 3; k++) o.tw[k] = tw.t[(size_t) k * tw.nel + e];
                e++;
            }
            add(rank, o);
        }
        real(o0, logm - 1, rank + 1, dummy);
        cplx(o0 + m2, o0 + 3 * (m / 4), logm - 2, rank + 1);
        /* step 5 (src/subs.c:506-523) only relabels and negates finished values: folded into the read-out table */
        for (int n = 0; n < m8; n++) { Post p = {FOP_SWAPNN, o0 + m2 + m4 + n, o0 + m - 1 - n}; post1.push_back(p); }
        for (int n = 0; n < m8; n++) { Post p = {FOP_SWAPN, o0 + m2 + 1 + 2 * n, o0 + m - 2 - 2 * n}; post2.push_back(p); }
        if (logm == 2) { Post p = {FOP_NEG, o0 + 3, 0}; post1.push_back(p); }
    }

    /* LDS cycles of one round's operand k under the model the placement works to: an 8-byte store is served 16 lanes a
       cycle (32 banks), an 8-byte load 32 lanes a cycle (64 banks); lanes of a group that fall on the same bank pair
       take a cycle each.  pos[l]: the operand's element position in lane l. */
    static int round_cycles(const unsigned *pos)
    {
        int total = 0;
        for (int g = 0; g < 64; g += 16) {
            int cnt[16] = {0}, mx = 0;
            for (int l = g; l < g + 16; l++) { const int c = ++cnt[pos[l] & 15]; mx = c > mx ? c : mx; }
            total += mx;
        }
        for (int g = 0; g < 64; g += 32) {
            int cnt[32] = {0}, mx = 0;
            for (int l = g; l < g + 32; l++) { const int c = ++cnt[pos[l] & 31]; mx = c > mx ? c : mx; }
            total += mx;
        }
        return total;
    }

    /* nwin transforms of 2^logN points at element offsets w << logN; returns the number of program words */
    /* The butterflies of the blocks of 256 points and more -- R(1024), R(512), C(256), R(256) of the long transform, R(256) of
       each short one -- are not part of the program: with element e of a transform in register e / 64 of lane e % 64 all their
       operands are registers of ONE lane (m / 4 >= 64), and k_fft.hip runs them there before the data ever reaches LDS.
       What a lane needs for them is regtw[r][lane] = {cn, spcn, smcn, flags} (a second row {c3n, spc3n, smc3n, 0} for C(256)):
       long: r = 0..3 R(1024) n = lane + 64 r; 4, 5 R(512) n = lane + 64 (r - 4); 6, 7 C(256) n = lane; 8 R(256) n = lane;
       short: r = 0 R(256) n = lane, the same for the three windows. */
    static bool in_registers(const FusedOp &o) { return o.logm >= 8; }
    /* The other end of the recursion: the blocks of 8 points and fewer are not part of the program either.  Every one of them lies
       within one or two runs of 8 consecutive elements, 8-aligned: a lane of k_fft.hip takes two such runs A and B (16 elements, four
       16-byte LDS reads each) after the last round and takes them through the rest of the recursion in its registers (fft_leaves):
         kind 0   C(8) with its real parts in A and its imaginary parts in B, down to the length-2 butterflies;
         kind 1   the two C(4) a block C(16) spawns: real parts A[0..3] / A[4..7], imaginary parts B[0..3] / B[4..7];
         kind 2   the two children of a block R(16): R(8) in A, and in B = A + 8 the C(4) whose imaginary parts follow its real ones.
       1024 points: 42 + 21 + 1 lanes; the three transforms of 256 points: 3 x (10 + 5 + 1).  leaf_ops lists the butterflies of a kind in
       the order fft_leaves runs them -- build() checks that they are exactly the recursion's. */
    static bool in_leaf(const FusedOp &o) { return o.logm <= 3; }
    struct LeafOp { int cls, logm, kind, neg; int e[8]; /* operand: run (0 = A, 1 = B) * 8 + index; -1 = the dummy element */ };
    static const std::vector<LeafOp> &leaf_ops(int kind)
    {
        enum { A0, A1, A2, A3, A4, A5, A6, A7, B0, B1, B2, B3, B4, B5, B6, B7 };
        static const std::vector<LeafOp> ops[3] = {
            {   {1, 3, 0, 0, {A0, A4, A2, A6, B0, B4, B2, B6}}, {1, 3, 2, 0, {A1, A5, A3, A7, B1, B5, B3, B7}},
                {1, 2, 0, 0, {A0, A2, A1, A3, B0, B2, B1, B3}}, {0, 1, 0, 0, {A0, A1, B0, B1}}, {0, 1, 0, 0, {A4, A5, B4, B5}},
                {0, 1, 0, 0, {A6, A7, B6, B7}} },
            {   {1, 2, 0, 0, {A4, A6, A5, A7, B4, B6, B5, B7}},
                {1, 2, 0, 0, {A0, A2, A1, A3, B0, B2, B1, B3}}, {0, 1, 0, 0, {A0, A1, B0, B1}}, {0, 1, 0, 0, {A4, A5, B4, B5}} },
            {   {0, 3, 0, 1, {A0, A4, A2, A6}}, {0, 3, 2, 1, {A1, A5, A3, A7}}, {0, 2, 0, 1, {A0, A2, A1, A3}}, {0, 1, 0, 0, {A0, A1, -1, -1}},
                {0, 1, 0, 0, {A4, A5, A6, A7}}, {1, 2, 0, 0, {B0, B2, B1, B3, B4, B6, B5, B7}}, {0, 1, 0, 0, {B0, B1, B4, B5}} },
        };
        return ops[kind];
    }
    void reg_row(uint32_t *dst, bool cplx_blk, int logm, int n0, int second) const
    {
        const int m = 1 << logm, m8 = m / 8;
        const Twiddle &tw = cplx_blk ? tw_sr[logm] : tw_rs[logm];
        for (int l = 0; l < 64; l++) {
            const int n = n0 + l, kind = n == 0 ? 0 : (n == m8 ? 2 : 1);
            uint32_t w[4] = {0, 0, 0, 0};
            if (kind == 1) {
                const int e = n - 1 - (n > m8 ? 1 : 0);
                for (int k = 0; k < 3; k++) memcpy(&w[k], &tw.t[(size_t) (3 * second + k) * tw.nel + e], 4);
            }
            if (!second) w[3] = (kind == 1 ? 1u : 0u) | (kind == 2 ? 2u : 0u) | (cplx_blk ? 0u : 0x80000000u);
            memcpy(dst + 4 * l, w, 16);
        }
    }
    /* the row and lane of regtw that serve butterfly o of a transform whose window starts at element w0 */
    static bool reg_slot(int logN, const FusedOp &o, int first_elem, int *row, int *lane)
    {
        const int n_abs = first_elem; /* p[0] before the swizzle, relative to the window */
        if (logN == 10) {
            if (o.cls == 0 && o.logm == 10) { *row = n_abs / 64; *lane = n_abs % 64; return n_abs < 256; }
            if (o.cls == 0 && o.logm == 9) { *row = 4 + n_abs / 64; *lane = n_abs % 64; return n_abs < 128; }
            if (o.cls == 1 && o.logm == 8) { *row = 6; *lane = n_abs - 512; return n_abs >= 512 && n_abs < 576; }
            if (o.cls == 0 && o.logm == 8) { *row = 8; *lane = n_abs; return n_abs < 64; }
            return false;
        }
        if (o.cls == 0 && o.logm == 8) { *row = 0; *lane = n_abs; return n_abs < 64; }
        return false;
    }

    int build(int logN, int nwin, uint32_t *hdr, int max_rounds, int32_t *n_rounds, uint32_t *prog, int max_words, uint32_t *rd, uint32_t *regtw, uint32_t *leaf)
    {
        const int N = 1 << logN;
        const unsigned dummy = logN == 10 ? MP3MI_FFT_DUMMY : MP3MI_FFT_DUMMY_S; /* where the idle lanes of a round work: behind the transforms */
        for (int e = 0; e < nwin * N; e++)
            if (MP3MI_FFT_SWZ(e) < 0 || MP3MI_FFT_SWZ(e) >= nwin * N) { fprintf(stderr, "mp3mi: the fft swizzle leaves the array\n"); abort(); }
        rank_ops.clear();
        post1.clear();
        post2.clear();
        real(0, logN, 0, dummy);
        /* read-out: where bin i's real and imaginary part are once the butterflies are done, and with which sign */
        {
            std::vector<int> src((size_t) N), sg((size_t) N, 0);
            for (int i = 0; i < N; i++) src[(size_t) i] = i;
            auto apply = [&](const Post &p) {
                if (p.type == FOP_NEG) sg[p.a] ^= 1;
                else {
                    const int ta = src[p.a], sa = sg[p.a];
                    src[p.a] = src[p.b]; sg[p.a] = sg[p.b] ^ 1;
                    src[p.b] = ta; sg[p.b] = sa ^ (p.type == FOP_SWAPNN ? 1 : 0);
                }
            };
            for (size_t i = 0; i < post1.size(); i++) apply(post1[i]);
            for (size_t i = 0; i < post2.size(); i++) apply(post2[i]);
            for (int i = 0; i < N; i++) { /* bit reversal, src/subs.c:136-177 */
                int j = 0;
                for (int b = 0; b < logN; b++)
                    if (i & (1 << b)) j |= 1 << (logN - 1 - b);
                if (j > i) { std::swap(src[(size_t) i], src[(size_t) j]); std::swap(sg[(size_t) i], sg[(size_t) j]); }
            }
            for (int i = 0; i <= N / 2; i++) {
                const int k = (i == 0 || i == N / 2) ? i : N - i;
                rd[i] = (uint32_t) MP3MI_FFT_SWZ(src[(size_t) i]) | ((uint32_t) sg[(size_t) i] << 15) |
                        ((uint32_t) MP3MI_FFT_SWZ(src[(size_t) k]) << 16) | ((uint32_t) sg[(size_t) k] << 31);
            }
        }
        int nr = 0, nw = 0;
        /* The butterflies form a DAG: one depends on those that last wrote its operands -- the level above IN ITS OWN
           sub-transform.  The recursion's ranks are one valid order, but a wasteful one: the sub-transforms of
           different sizes that a split-radix level spawns run out at different depths, so the deep ranks hold a few
           butterflies each, and every (rank, class) pays whole rounds of 64 lanes (1222 butterflies in 26 rounds for
           1024 points, 789 in 19 for the three 256-point transforms).  List scheduling instead: steps separated by a
           wave barrier; a step takes, per class, the ready butterflies in whole rounds, the longest remaining chains
           first, and a partial round only when it holds a butterfly of the longest chain (everything else can wait
           for a later step and fill its round). */
        typedef FftNode Node;
        std::vector<Node> nodes;
        for (size_t rank = 0; rank < rank_ops.size(); rank++)
            for (int w = 0; w < nwin; w++)
                for (size_t i = 0; i < rank_ops[rank].size(); i++) {
                    Node nd;
                    nd.o = rank_ops[rank][i];
                    nd.o.id = (int) nodes.size();
                    const int nop = nd.o.cls ? 8 : 4;
                    for (int k = 0; k < 8; k++) nd.lp[k] = (k < nop && nd.o.p[k] != dummy) ? w * N + (int) nd.o.p[k] : -1;
                    for (int k = 0; k < nop; k++)
                        nd.o.p[k] = (nd.o.p[k] == dummy) ? nd.o.p[k] : (unsigned) MP3MI_FFT_SWZ(w * N + (int) nd.o.p[k]);
                    nd.npred = 0; nd.height = 1; nd.done = false;
                    nodes.push_back(nd);
                }
        if (logN == 10) {
            for (int r = 0; r < 4; r++) reg_row(regtw + 256 * r, false, 10, 64 * r, 0);
            for (int r = 0; r < 2; r++) reg_row(regtw + 256 * (4 + r), false, 9, 64 * r, 0);
            reg_row(regtw + 256 * 6, true, 8, 0, 0);
            reg_row(regtw + 256 * 7, true, 8, 0, 1);
            reg_row(regtw + 256 * 8, false, 8, 0, 0);
        } else reg_row(regtw, false, 8, 0, 0);
        {   /* every butterfly that is left to the registers has its row there, with exactly its rotation */
            size_t n_reg = 0;
            for (size_t i = 0; i < nodes.size(); i++) {
                const FusedOp &o = nodes[i].o;
                if (!in_registers(o)) continue;
                int row = 0, lane = 0;
                bool ok = reg_slot(logN, o, nodes[i].lp[0] % N, &row, &lane);
                uint32_t want[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                if (o.kind == 1) memcpy(want, o.tw, 12);
                want[3] = (o.kind == 1 ? 1u : 0u) | (o.kind == 2 ? 2u : 0u) | (o.neg ? 0x80000000u : 0u);
                if (o.cls && o.kind == 1) memcpy(want + 4, o.tw + 3, 12);
                ok = ok && memcmp(regtw + 256 * row + 4 * lane, want, 16) == 0;
                if (ok && o.cls) ok = memcmp(regtw + 256 * (row + 1) + 4 * lane, want + 4, 16) == 0;
                if (!ok) { fprintf(stderr, "mp3mi: fft register rounds do not match the butterflies of the recursion\n"); abort(); }
                n_reg++;
            }
            if (n_reg != (size_t) nwin * (logN == 10 ? 256 + 128 + 64 + 64 : 64)) { fprintf(stderr, "mp3mi: fft register rounds: %zu butterflies\n", n_reg); abort(); }
        }
        {   /* the lanes of fft_leaves: which two runs of 8 elements each takes, and as what (in_leaf above) */
            struct Leaf { int kind, a, b; };
            std::vector<Leaf> leaves;
            std::vector<char> c8_at((size_t) nwin * (size_t) N, 0), used(nodes.size(), 0);
            for (size_t i = 0; i < nodes.size(); i++)
                if (nodes[i].o.cls == 1 && nodes[i].o.logm == 3 && nodes[i].o.kind == 0) c8_at[(size_t) nodes[i].lp[0]] = 1;
            for (size_t i = 0; i < nodes.size(); i++) {
                const FusedOp &o = nodes[i].o;
                const int *lp = nodes[i].lp;
                if (o.cls == 1 && o.logm == 3 && o.kind == 0) { Leaf l = {0, lp[0], lp[4]}; leaves.push_back(l); }
                else if (o.cls == 0 && o.logm == 3 && o.kind == 0) { Leaf l = {2, lp[0], lp[0] + 8}; leaves.push_back(l); }
                else if (o.cls == 1 && o.logm == 2 && lp[0] % 8 == 0 && lp[4] != lp[0] + 4 && !c8_at[(size_t) lp[0]]) { Leaf l = {1, lp[0], lp[4]}; leaves.push_back(l); }
            }
            bool ok = leaves.size() <= 64;
            for (size_t li = 0; ok && li < leaves.size(); li++) {
                const Leaf &lf = leaves[li];
                ok = lf.a % 8 == 0 && lf.b % 8 == 0 && lf.a != lf.b;
                const std::vector<LeafOp> &ops = leaf_ops(lf.kind);
                for (size_t q = 0; ok && q < ops.size(); q++) {
                    const LeafOp &lo = ops[q];
                    int want[8];
                    for (int k = 0; k < 8; k++) want[k] = (k < (lo.cls ? 8 : 4) && lo.e[k] >= 0) ? (lo.e[k] < 8 ? lf.a + lo.e[k] : lf.b + lo.e[k] - 8) : -1;
                    bool found = false;
                    for (size_t i = 0; !found && i < nodes.size(); i++) {
                        const FusedOp &o = nodes[i].o;
                        if (used[i] || o.cls != lo.cls || o.logm != lo.logm || o.kind != lo.kind || o.neg != lo.neg) continue;
                        if (memcmp(nodes[i].lp, want, sizeof(want)) == 0) { used[i] = 1; found = true; }
                    }
                    ok = found;
                }
            }
            size_t n_leaf = 0, n_used = 0;
            for (size_t i = 0; i < nodes.size(); i++) { n_leaf += in_leaf(nodes[i].o) ? 1 : 0; n_used += used[i] ? 1 : 0; ok = ok && (used[i] != 0) == in_leaf(nodes[i].o); }
            if (!ok) { fprintf(stderr, "mp3mi: fft leaves do not match the butterflies of the recursion (%zu leaves, %zu of %zu butterflies)\n", leaves.size(), n_used, n_leaf); abort(); }
            /* the lanes: a 16-byte LDS access is served 16 lanes a cycle over the 64 banks -- the runs of the 16 lanes of a group should
               start in 16 different groups of four banks (position / 2 mod 16; the same for the four accesses of a run) */
            auto quad = [&](int e) { return (MP3MI_FFT_SWZ(e) >> 1) & 15; };
            auto cycles = [&](const std::vector<Leaf> &ln) { /* of one access to run A plus one to run B, all lanes */
                int cyc = 0;
                for (size_t g = 0; g < ln.size(); g += 16) {
                    int ca[16] = {0}, cb[16] = {0}, ma = 0, mb = 0;
                    for (size_t l = g; l < g + 16 && l < ln.size(); l++) { ma = std::max(ma, ++ca[quad(ln[l].a)]); mb = std::max(mb, ++cb[quad(ln[l].b)]); }
                    cyc += ma + mb;
                }
                return cyc;
            };
            auto crowding = [&](const std::vector<Leaf> &ln) { /* what the search goes down on: the squares of the lanes per group of banks */
                int sq = 0;
                for (size_t g = 0; g < ln.size(); g += 16) {
                    int ca[16] = {0}, cb[16] = {0};
                    for (size_t l = g; l < g + 16 && l < ln.size(); l++) { ca[quad(ln[l].a)]++; cb[quad(ln[l].b)]++; }
                    for (int q = 0; q < 16; q++) sq += ca[q] * ca[q] + cb[q] * cb[q];
                }
                return sq;
            };
            std::vector<Leaf> lanes; /* greedy by groups of 16 lanes, then swaps of two lanes while they help */
            std::vector<char> taken(leaves.size(), 0);
            while (lanes.size() < leaves.size()) {
                int ua[16] = {0}, ub[16] = {0};
                for (int g = 0; g < 16 && lanes.size() < leaves.size(); g++) {
                    size_t best = 0;
                    int bc = 1 << 30;
                    for (size_t i = 0; i < leaves.size(); i++) {
                        if (taken[i]) continue;
                        const int c = ua[quad(leaves[i].a)] + ub[quad(leaves[i].b)];
                        if (c < bc) { bc = c; best = i; }
                    }
                    taken[best] = 1;
                    ua[quad(leaves[best].a)]++;
                    ub[quad(leaves[best].b)]++;
                    lanes.push_back(leaves[best]);
                }
            }
            for (bool better = true; better;) {
                better = false;
                for (size_t i = 0; i < lanes.size(); i++)
                    for (size_t j = i + 1; j < lanes.size(); j++) {
                        if (i / 16 == j / 16) continue;
                        const int before = crowding(lanes);
                        std::swap(lanes[i], lanes[j]);
                        if (crowding(lanes) < before) better = true;
                        else std::swap(lanes[i], lanes[j]);
                    }
            }
            leaf_cycles = cycles(lanes);
            for (int l = 0; l < 64; l++) {
                uint32_t w[4] = {3u << 14, 0, 0, 0}; /* kind 3: an idle lane */
                if ((size_t) l < lanes.size()) {
                    unsigned pp[8];
                    for (int j = 0; j < 8; j++) {
                        const int e = (j < 4 ? lanes[(size_t) l].a : lanes[(size_t) l].b) + 2 * (j & 3);
                        pp[j] = (unsigned) MP3MI_FFT_SWZ(e);
                        if ((pp[j] & 1u) || (unsigned) MP3MI_FFT_SWZ(e + 1) != pp[j] + 1u || pp[j] >= (1u << 14)) { fprintf(stderr, "mp3mi: the fft swizzle splits a pair of elements\n"); abort(); }
                    }
                    for (int k = 0; k < 4; k++) w[k] = pp[2 * k] | (pp[2 * k + 1] << 16);
                    w[0] |= (uint32_t) lanes[(size_t) l].kind << 14;
                }
                memcpy(leaf + 4 * l, w, 16);
            }
            if (MP3MI_FFT_INFO_ON)
                fprintf(stderr, "mp3mi: fft 2^%d: %zu leaves; a 16-byte access to run A and one to run B of the 64 lanes in %d cycles (conflict-free %zu)\n", logN, lanes.size(), leaf_cycles, 2 * ((lanes.size() + 15) / 16));
        }
        {
            std::vector<int> last_writer((size_t) nwin * (size_t) N, -1);
            for (size_t i = 0; i < nodes.size(); i++) { /* (rank order: a butterfly's producers come before it) */
                for (int k = 0; k < 8; k++) {
                    const int e = nodes[i].lp[k];
                    if (e < 0) continue;
                    const int pr = last_writer[(size_t) e];
                    if (pr >= 0 && pr != (int) i && std::find(nodes[(size_t) pr].succ.begin(), nodes[(size_t) pr].succ.end(), (int) i) == nodes[(size_t) pr].succ.end()) {
                        nodes[(size_t) pr].succ.push_back((int) i);
                        nodes[i].pred.push_back(pr);
                        nodes[i].npred++;
                    }
                }
                for (int k = 0; k < 8; k++) if (nodes[i].lp[k] >= 0) last_writer[(size_t) nodes[i].lp[k]] = (int) i;
            }
            for (size_t i = nodes.size(); i-- > 0;) /* (chains of the program: the leaves come behind all of it) */
                for (size_t j = 0; j < nodes[i].succ.size(); j++)
                    if (!in_leaf(nodes[(size_t) nodes[i].succ[j]].o)) nodes[i].height = std::max(nodes[i].height, 1 + nodes[(size_t) nodes[i].succ[j]].height);
        }
        FftSchedule sched;
        size_t n_left = 0;
        for (size_t i = 0; i < nodes.size(); i++) {
            if (in_leaf(nodes[i].o)) {
                nodes[i].done = true; /* no part of the schedule; nothing of the program waits for a leaf */
                for (size_t j = 0; j < nodes[i].succ.size(); j++)
                    if (!in_leaf(nodes[(size_t) nodes[i].succ[j]].o)) { fprintf(stderr, "mp3mi: an fft leaf feeds the program\n"); abort(); }
            } else n_left++;
        }
        for (int step = 0; n_left > 0; step++) {
            std::vector<int> ready[2];
            int maxh = 0;
            for (size_t i = 0; i < nodes.size(); i++)
                if (!nodes[i].done && nodes[i].npred == 0) { ready[nodes[i].o.cls].push_back((int) i); maxh = std::max(maxh, nodes[i].height); }
            std::vector<int> chosen[2];
            for (int cls = 0; cls < 2; cls++) {
                std::stable_sort(ready[cls].begin(), ready[cls].end(), [&](int a, int b) { return nodes[(size_t) a].height > nodes[(size_t) b].height; });
                size_t take = ready[cls].size() / 64 * 64;
                bool critical = false;
                for (size_t i = take; i < ready[cls].size(); i++) critical |= nodes[(size_t) ready[cls][i]].height == maxh;
                if (critical) take = ready[cls].size();
                chosen[cls].assign(ready[cls].begin(), ready[cls].begin() + (long) take);
            }
            if (chosen[0].empty() && chosen[1].empty()) { /* (cannot happen: the longest chain's head is ready) */
                const int c = ready[1].size() > ready[0].size() ? 1 : 0;
                chosen[c] = ready[c];
            }
            for (int cls = 0; cls < 2; cls++) {
                const int nopnd = cls ? 8 : 4;
                std::vector<FusedOp> rest, sq;
                /* rotations first; the few SQHALF rotations go last so that only the last round of the class
                   pays for that arithmetic */
                static const int kind_order[3] = {1, 0, 2};
                for (int ko = 0; ko < 3; ko++)
                    for (size_t i = 0; i < chosen[cls].size(); i++) {
                        const FusedOp &o = nodes[(size_t) chosen[cls][i]].o;
                        if (o.kind != kind_order[ko]) continue;
                        (o.kind == 2 ? sq : rest).push_back(o);
                    }
                /* The butterflies of a rank are independent, so their order is free: place them so that the 32
                   lanes an 8-byte LDS read is served in address 32 different bank pairs with every operand, and
                   the 16 lanes a store is served in 16 different ones.  Greedy: first what fits without any
                   collision, then -- an idle lane costs as much as a busy one, a collision one cycle -- what
                   collides least. */
                std::vector<FusedOp> placed;
                while (!rest.empty()) {
                    int usedr[8][32], usedw[8][2][16];
                    memset(usedr, 0, sizeof(usedr));
                    memset(usedw, 0, sizeof(usedw));
                    std::vector<FusedOp> group;
                    auto cost = [&](const FusedOp &o) {
                        const size_t q = group.size() / 16;
                        int c = 0;
                        for (int k = 0; k < nopnd; k++) c += usedr[k][o.p[k] & 31] + usedw[k][q][o.p[k] & 15];
                        return c;
                    };
                    auto take = [&](size_t idx) {
                        const FusedOp o = rest[idx];
                        const size_t q = group.size() / 16;
                        for (int k = 0; k < nopnd; k++) { usedr[k][o.p[k] & 31]++; usedw[k][q][o.p[k] & 15]++; }
                        group.push_back(o);
                        rest.erase(rest.begin() + (long) idx);
                    };
                    for (size_t i = 0; i < rest.size() && group.size() < 32;) {
                        if (cost(rest[i]) == 0) take(i);
                        else i++;
                    }
                    while (group.size() < 32 && !rest.empty()) {
                        size_t best = 0;
                        int bc = 1 << 30;
                        for (size_t i = 0; i < rest.size(); i++) {
                            const int c = cost(rest[i]);
                            if (c < bc) { bc = c; best = i; }
                        }
                        take(best);
                    }
                    placed.insert(placed.end(), group.begin(), group.end());
                }
                placed.insert(placed.end(), sq.begin(), sq.end());
                std::vector<int> ids;
                for (size_t i = 0; i < placed.size(); i++) ids.push_back(placed[i].id);
                sched.push_back(ids);
            }
            for (int cls = 0; cls < 2; cls++)
                for (size_t i = 0; i < chosen[cls].size(); i++) {
                    Node &nd = nodes[(size_t) chosen[cls][i]];
                    nd.done = true;
                    n_left--;
                    for (size_t j = 0; j < nd.succ.size(); j++) nodes[(size_t) nd.succ[j]].npred--;
                }
        }
        /* That schedule -- list scheduling, greedy lanes -- leaves up to twice the conflict-free cycle count in the
           deep steps (many short transforms, eight operands that must all spread at once).  A long offline search
           (simulated annealing over swaps of butterflies of the same kind within a list and, where the dependencies
           allow, between steps: tools/exp/fft_swz_search.cpp) does better; its result is stored as lists of butterfly
           ids (fft_placement.h) and replayed here if it is a schedule of exactly these butterflies that respects
           every dependency; otherwise the one above stands. */
        if (stored_order) {
            FftSchedule st;
            for (size_t li = 0;; li++) {
                const std::vector<uint16_t> *ord = stored_order(logN, (int) (li / 2), (int) (li & 1));
                if (!ord) { if (li & 1) { st.push_back(std::vector<int>()); continue; } break; }
                st.push_back(std::vector<int>(ord->begin(), ord->end()));
            }
            std::vector<int> step_of(nodes.size(), -1);
            bool ok = !st.empty();
            size_t count = 0;
            for (size_t li = 0; ok && li < st.size(); li++)
                for (size_t i = 0; ok && i < st[li].size(); i++) {
                    const int id = st[li][i];
                    ok = id >= 0 && (size_t) id < nodes.size() && step_of[(size_t) id] < 0 && nodes[(size_t) id].o.cls == (int) (li & 1);
                    if (ok) { step_of[(size_t) id] = (int) (li / 2); count++; }
                }
            size_t n_prog = 0;
            for (size_t i = 0; i < nodes.size(); i++) n_prog += in_leaf(nodes[i].o) ? 0 : 1;
            for (size_t i = 0; ok && i < nodes.size(); i++) ok = (step_of[i] >= 0) != in_leaf(nodes[i].o);
            ok = ok && count == n_prog;
            for (size_t i = 0; ok && i < nodes.size(); i++)
                for (size_t j = 0; ok && !in_leaf(nodes[i].o) && j < nodes[i].pred.size(); j++) ok = step_of[(size_t) nodes[i].pred[j]] < step_of[i];
            if (ok) sched = st;
            else if (MP3MI_FFT_INFO_ON) fprintf(stderr, "mp3mi: the stored fft schedule for 2^%d points does not fit, list schedule kept\n", logN);
        }
        if (schedule_hook) schedule_hook(logN, nodes, sched);
        for (size_t li = 0; li < sched.size(); li += 2) {
            int last_round_of_rank = -1;
            for (int cls = 0; cls < 2 && li + (size_t) cls < sched.size(); cls++) {
                const int nopnd = cls ? 8 : 4;
                std::vector<FusedOp> placed;
                for (size_t i = 0; i < sched[li + (size_t) cls].size(); i++)
                    if (!in_registers(nodes[(size_t) sched[li + (size_t) cls][i]].o)) placed.push_back(nodes[(size_t) sched[li + (size_t) cls][i]].o);
                if (MP3MI_FFT_INFO_ON && placed.size() != sched[li + (size_t) cls].size())
                    fprintf(stderr, "mp3mi: fft 2^%d step %zu class %d: %zu of %zu butterflies left to the program\n", logN, li / 2, cls, placed.size(), sched[li + (size_t) cls].size());
                /* rounds of 64: block 0 = operand positions (R: 2 words per lane, C: 4), then either the
                   twiddle block(s) {cn, spc, smc, flags} (C: a second one {c3n, spc3n, smc3n, 0}) or, in a round
                   without rotations, one word of flags per lane.  flags: bit 0 rotation, bit 1 SQHALF rotation,
                   bit 31 negate u2. */
                for (size_t r0 = 0; r0 < placed.size(); r0 += 64) {
                    bool has_rot = false, has_sq = false;
                    for (size_t l = r0; l < r0 + 64 && l < placed.size(); l++)
                        if (placed[l].cls >= 0) { has_rot |= placed[l].kind == 1; has_sq |= placed[l].kind == 2; }
                    const int words = (cls ? 256 : 128) + (has_rot ? (cls ? 512 : 256) : 64);
                    if (nr >= max_rounds || nw + words > max_words) { fprintf(stderr, "mp3mi: fft program too large\n"); abort(); }
                    uint32_t *blk = prog + nw;
                    for (int l = 0; l < 64; l++) {
                        FusedOp o;
                        memset(&o, 0, sizeof(o));
                        o.cls = -1;
                        if (r0 + (size_t) l < placed.size()) o = placed[r0 + (size_t) l];
                        if (o.cls < 0) { /* idle lane: works on its own dummy element */
                            memset(&o, 0, sizeof(o));
                            for (int k = 0; k < 8; k++) o.p[k] = (unsigned) (dummy + l);
                        }
                        for (int k = 0; k < nopnd; k++) if (o.p[k] == dummy) o.p[k] = (unsigned) (dummy + l);
                        const uint32_t flags = (o.kind == 1 ? 1u : 0u) | (o.kind == 2 ? 2u : 0u) | (o.neg ? 0x80000000u : 0u);
                        const int aw = cls ? 4 : 2;
                        for (int k = 0; k < aw; k++) blk[l * aw + k] = o.p[2 * k] | (o.p[2 * k + 1] << 16);
                        uint32_t *t = blk + 64 * aw;
                        if (has_rot) {
                            memcpy(&t[l * 4], o.tw, 12);
                            t[l * 4 + 3] = flags;
                            if (cls) { memcpy(&t[256 + l * 4], o.tw + 3, 12); t[256 + l * 4 + 3] = 0; }
                        } else t[l] = flags;
                    }
                    hdr[nr] = (uint32_t) cls | (has_rot ? 2u : 0u) | (has_sq ? 4u : 0u);
                    last_round_of_rank = nr;
                    nr++;
                    nw += words;
                }
            }
            if (last_round_of_rank >= 0) hdr[last_round_of_rank] |= 8u; /* the next step reads what this one wrote */
        }
        *n_rounds = nr;
        return nw;
    }
};

#if !defined(MP3MI_FFT_SWZ_RUNTIME)
#include "fft_placement.h"
/* the stored placement of one list (step, class) of the transform of 2^logN points; only for the swizzle it was made for */
static const std::vector<uint16_t> *fft_stored_order(int logN, int step, int cls)
{
    static const int now[6] = {MP3MI_FFT_SWZ_COLS}, then[6] = {MP3MI_FFT_PLACEMENT_COLS};
    for (int i = 0; i < 6; i++) if (now[i] != then[i]) return NULL;
    static std::vector<uint16_t> v;
    for (size_t i = 0; i < sizeof(FFT_PLACEMENT) / sizeof(FFT_PLACEMENT[0]); i++)
        if (FFT_PLACEMENT[i].logN == logN && FFT_PLACEMENT[i].rank == step && FFT_PLACEMENT[i].cls == cls) {
            v.assign(FFT_PLACEMENT[i].ids, FFT_PLACEMENT[i].ids + FFT_PLACEMENT[i].n);
            return &v;
        }
    return NULL;
}
#else
static const std::vector<uint16_t> *fft_stored_order(int, int, int) { return NULL; }
#endif

} // namespace

static int build_tables_unpinned(mp3mi_tables *T, int rate_idx)
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
            /* long blocks [0]: k_loop sums a job by PAIRS of lines (one 16-byte and one 4-byte LDS read per two terms), so
               bands -- whose edges are even at every rate, checked here -- are cut into parts of whole pairs: unit = 2 */
            const int unit = t ? 1 : 2;
            if (target % unit) continue;
            int jobs = 0;
            for (int b = 0; b < nb; b++) {
                const int w = t ? SFB_S[ri][b / 3 + 1] - SFB_S[ri][b / 3] : SFB_L[ri][b + 1] - SFB_L[ri][b];
                if (!t && ((w & 1) || (SFB_L[ri][b] & 1))) return -11;
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
                    const int cnt = unit * (((w - off) / unit + (parts - q) - 1) / (parts - q)); // even split of what is left, in units
                    T->nj_first[t][j] = (int16_t) (t ? (e0 + off) * 3 + b % 3 : e0 + off);
                    T->nj_count[t][j] = (uint8_t) cnt;
                    if (cnt > T->nj_max[t]) T->nj_max[t] = (uint8_t) cnt;
                    off += cnt;
                    j++;
                }
            }
            /* segment flags for the band sums by doubling (k_loop): job j of band b, bit d: job j + d is of band b too */
            for (int b = 0; b < nb; b++) {
                if (T->nj_njobs[t][b] > 16) return -7; /* (k_loop sums a band's jobs in four doublings) */
                for (int q = 0; q < T->nj_njobs[t][b]; q++)
                    for (int d = 1; d <= 16; d <<= 1)
                        if (q + d < T->nj_njobs[t][b]) T->nj_seg[t][T->nj_job0[t][b] + q] |= (uint8_t) d;
            }
            break;
        }
    }

    for (int t = 0; t < 2; t++) { /* k_loop's per-lane constants (mp3mi_dev.h) */
        const int nb = t ? 36 : 21;
        for (int l = 0; l < 64; l++) {
            uint64_t bp = 0;
            if (t) { /* short blocks: value j of lane l is line 2 (l + 64 (j / 2)) + j % 2 (k_loop.hip, loop_pair_of) */
                for (int j = 0; j < 10; j++) {
                    const int line = 2 * (l + 64 * (j / 2)) + j % 2;
                    bp |= (uint64_t) (line < 576 ? T->sfb_of_line_s[line] : 63) << (6 * j);
                }
            } else { /* long blocks: a band's edges are even (checked above), both lines of PAIR l + 64 k are of one band */
                for (int k = 0; k < 5; k++) {
                    const int line = 2 * (l + 64 * k);
                    if (line < 576 && T->sfb_of_line_l[line] != T->sfb_of_line_l[line + 1]) return -12;
                    bp |= (uint64_t) (line < 576 ? T->sfb_of_line_l[line] : 63) << (6 * k);
                }
            }
            T->lane_bands[t][l] = bp;
            uint64_t first = 0, count = 0;
            if (l < nb) {
                if (t) { const int sfb = l / 3, w = l % 3; first = (uint64_t) (SFB_S[ri][sfb] * 3 + w); count = (uint64_t) (SFB_S[ri][sfb + 1] - SFB_S[ri][sfb]); }
                else { first = (uint64_t) SFB_L[ri][l]; count = (uint64_t) (SFB_L[ri][l + 1] - SFB_L[ri][l]); }
            }
            if (count > 255 || first > 1023 || T->nj_first[t][l] > 1023 || T->nj_first[t][l] < 0) return -10;
            T->lane_jobs[t][l] = (uint64_t) T->nj_first[t][l] | ((uint64_t) T->nj_count[t][l] << 10) | ((uint64_t) (T->nj_seg[t][l] & 31) << 18) |
                                 ((uint64_t) (l < nb ? T->nj_job0[t][l] : 0) << 23) | ((uint64_t) (l < nb ? T->nj_njobs[t][l] : 0) << 29) |
                                 (count << 35) | (first << 43);
            T->lane_inv_lines[t][l] = count ? 1.0 / (double) count : 0.0; /* (an IEEE division: the same on every host) */
        }
    }

    LIBM_TABLE(TB_WINDOW, -1, T->window, sizeof(T->window),
               for (unsigned i = 0; i < 1024; i++) T->window[i] = (float) (0.5 * (1 - cos(2.0 * R_PI * (i - 0.5) / 1024))));
    LIBM_TABLE(TB_WINDOW_S, -1, T->window_s, sizeof(T->window_s),
               for (unsigned i = 0; i < 256; i++) T->window_s[i] = (float) (0.5 * (1 - cos(2.0 * R_PI * (i - 0.5) / 256))));

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
    (void) bval_l;
    LIBM_TABLE(TB_S3_L, ri, T->s3_l, sizeof(T->s3_l),
               for (int i = 0; i < cb_l; i++)
                   for (int j = 0; j < cb_l; j++) {
                       double tempx; double x; double tempy; double temp;
                       if (j >= i) tempx = (bval_l[i] - bval_l[j]) * 3.0;
                       else tempx = (bval_l[i] - bval_l[j]) * 1.5;
                       if (tempx >= 0.5 && tempx <= 2.5) { temp = tempx - 0.5; x = 8.0 * (temp * temp - 2.0 * temp); }
                       else x = 0.0;
                       tempx += 0.474;
                       tempy = 15.811389 + 7.5 * tempx - 17.5 * sqrt(1.0 + tempx * tempx);
                       T->s3_l[i][j] = (tempy <= -60.0) ? 0.0 : exp((x + tempy) * R_LN_TO_LOG10);
                   });
    memset(T->s3_lt, 0, sizeof(T->s3_lt));
    for (int i = 0; i < MP3MI_CBANDS; i++)
        for (int j = 0; j < MP3MI_CBANDS; j++) T->s3_lt[j][i] = T->s3_l[i][j];
    if (ri == 0) /* k_psy<SPARSE> keeps a row's non-zero run in PSY_S3_W = 17 LDS entries */
        for (int i = 0; i < MP3MI_CBANDS; i++)
            if (T_S3_HI[i] - T_S3_LO[i] + 1 > 17) return -9;
    k2 = 0;
    for (int i = 0; i < cb_s; i++) {
        T->numlines_pe[i] = T_PS_NUMLINES[ri][i]; /* the short table overwrites the long one: src/l3psy.c:868 */
        T->part_s_start[i] = k2;
        k2 += T_PS_NUMLINES[ri][i];
        T->qthr_s[i] = T_PS_QTHR[ri][i];
    }
    for (int i = cb_s; i <= MP3MI_CBANDS_S; i++) T->part_s_start[i] = k2;
    T->part_s_covered = k2;
    if (k2 > MP3MI_HBLK_S) return -2;
    /* entries of qthr_s/exp_snr_s beyond cb_s: the reference loops b < 42 over statics that
       were never written there: qthr_s = 0, SNR_s = 0 -> exp(0) = 1 */
    LIBM_TABLE(TB_EXP_SNR_S, ri, T->exp_snr_s, sizeof(T->exp_snr_s),
               for (int i = 0; i < cb_s; i++) T->exp_snr_s[i] = exp((double) T_PS_SNR[ri][i] * R_LN_TO_LOG10);
               for (int i = cb_s; i < MP3MI_CBANDS_S; i++) T->exp_snr_s[i] = exp(0.0 * R_LN_TO_LOG10));
    for (int i = 0; i < MP3MI_CBANDS; i++) { T->s3_lo[i] = T_S3_LO[i]; T->s3_hi[i] = T_S3_HI[i]; }
    for (int i = 0; i < 21; i++) {
        T->bu_l[i] = T_SL_BU[ri][i]; T->bo_l[i] = T_SL_BO[ri][i];
        T->w1_l[i] = T_SL_W1[ri][i]; T->w2_l[i] = T_SL_W2[ri][i];
    }
    for (int i = 0; i < 12; i++) {
        T->bu_s[i] = T_SS_BU[ri][i]; T->bo_s[i] = T_SS_BO[ri][i];
        T->w1_s[i] = T_SS_W1[ri][i]; T->w2_s[i] = T_SS_W2[ri][i];
    }

    {   /* subdivide of a granule without window switching, for every big_values (src/loop.c:1596-1625, 1638-1679):
           scfb_anz = band edges below 2*big_values pick the counts from subdv_table; both are then lowered
           until the regions end at or below 2*big_values */
        static const int SUBDV0[23] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 5, 5, 6, 6};
        static const int SUBDV1[23] = {0, 0, 0, 0, 0, 1, 1, 1, 2, 2, 3, 3, 4, 4, 4, 5, 5, 6, 6, 6, 7, 7, 7};
        for (int bv = 0; bv <= 288; bv++) {
            const int bvr = 2 * bv;
            int anz = 0, K = -1;
            for (int i = 0; i < 23; i++) { if (T->sfb_l[i] < bvr) anz++; if (T->sfb_l[i] <= bvr) K++; }
            int c0 = SUBDV0[anz], c1 = SUBDV1[anz];
            const int lim0 = K - 1 > 0 ? K - 1 : 0;
            c0 = c0 < lim0 ? c0 : lim0;          /* while (cnt && edge[cnt+1] > bvr) cnt-- */
            const int lim1 = K - c0 - 2 > 0 ? K - c0 - 2 : 0;
            c1 = c1 < lim1 ? c1 : lim1;          /* while (cnt && edge[r0+cnt+2] > bvr) cnt-- */
            T->subdiv_lut[bv] = (uint32_t) c0 | ((uint32_t) c1 << 4) | ((uint32_t) T->sfb_l[c0 + 1] << 8) | ((uint32_t) T->sfb_l[c0 + c1 + 2] << 18);
        }
    }

    {
        FftGen *g = new FftGen();
        g->stored_order = fft_stored_order;
        for (int i = 4; i <= 10; i++) { g->tw_rs[i] = make_twiddle(i, false); g->tw_sr[i] = make_twiddle(i, true); }
        T->fft_nword_l = g->build(10, 1, T->fft_hdr_l, MP3MI_FFT_MAX_ROUNDS, &T->fft_nround_l, T->fft_prog_l, MP3MI_FFT_PROG_WORDS, T->fft_rd_l, T->fft_regtw_l, T->fft_leaf_l);
        T->fft_nword_s = g->build(8, 3, T->fft_hdr_s, MP3MI_FFT_MAX_ROUNDS, &T->fft_nround_s, T->fft_prog_s, MP3MI_FFT_PROG_WORDS_S, T->fft_rd_s, T->fft_regtw_s, T->fft_leaf_s);
        {   /* the kernel is compiled for exactly this sequence of rounds (k_fft.hip) */
            static const uint8_t hl[] = {MP3MI_FFT_HDRS_L}, hs[] = {MP3MI_FFT_HDRS_S};
            bool same = T->fft_nround_l == (int) sizeof(hl) && T->fft_nround_s == (int) sizeof(hs);
            for (int i = 0; same && i < T->fft_nround_l; i++) same = T->fft_hdr_l[i] == hl[i];
            for (int i = 0; same && i < T->fft_nround_s; i++) same = T->fft_hdr_s[i] == hs[i];
            if (MP3MI_FFT_INFO_ON) {
                fprintf(stderr, "#define MP3MI_FFT_HDRS_L");
                for (int i = 0; i < T->fft_nround_l; i++) fprintf(stderr, "%s %u", i ? "," : "", T->fft_hdr_l[i]);
                fprintf(stderr, "\n#define MP3MI_FFT_HDRS_S");
                for (int i = 0; i < T->fft_nround_s; i++) fprintf(stderr, "%s %u", i ? "," : "", T->fft_hdr_s[i]);
                fprintf(stderr, "\n");
            }
            if (!same) { fprintf(stderr, "mp3mi: fft program does not match MP3MI_FFT_HDRS_* (mp3mi_dev.h)\n"); delete g; return -6; }
        }
        if (MP3MI_FFT_INFO_ON) fprintf(stderr, "mp3mi: fft program long %d rounds %d words, short %d rounds %d words\n", T->fft_nround_l, T->fft_nword_l, T->fft_nround_s, T->fft_nword_s);
        delete g;
    }

    for (int i = 0; i < 512; i++) T->enwindow[i] = T_ENWINDOW[i];
    LIBM_TABLE(TB_FILT, -1, T->filt, sizeof(T->filt),
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
               });

    double (*win)[36] = T->mdct_win;
    LIBM_TABLE(TB_MDCT_WIN, -1, T->mdct_win, sizeof(T->mdct_win),
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
               for (int i = 12; i < 36; i++) win[2][i] = 0.0);
    (void) win;
    LIBM_TABLE(TB_COS_S, -1, T->cos_s, sizeof(T->cos_s),
               const int N = 12;
               for (int m = 0; m < N / 2; m++)
                   for (int k = 0; k < N; k++)
                       T->cos_s[m][k] = cos((R_PI / (2 * N)) * (2 * k + 1 + N / 2) * (2 * m + 1)) / (N / 4));
    LIBM_TABLE(TB_COS_L, -1, T->cos_l, sizeof(T->cos_l),
               const int N = 36;
               for (int m = 0; m < N / 2; m++)
                   for (int k = 0; k < N; k++)
                       T->cos_l[m][k] = cos((R_PI / (2 * N)) * (2 * k + 1 + N / 2) * (2 * m + 1)) / (N / 4));
    LIBM_TABLE(TB_CA, -1, T->ca, sizeof(T->ca),
               for (int k = 0; k < 8; k++) T->ca[k] = ALIAS_C[k] / sqrt(1.0 + ALIAS_C[k] * ALIAS_C[k]));
    LIBM_TABLE(TB_CS, -1, T->cs, sizeof(T->cs),
               for (int k = 0; k < 8; k++) T->cs[k] = 1.0 / sqrt(1.0 + ALIAS_C[k] * ALIAS_C[k]));
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
        {   /* the shape k_mdct's long-block code relies on: twelve rows over V[0..17] in order, six of <= 6 terms */
            int nf = 0, nsm = 0;
            for (int m = 0; m < 18; m++) {
                bool full = T->mdct_nterm[m] == 18;
                for (int t = 0; full && t < 18; t++) full = T->mdct_vidx[m][t] == t;
                if (full) { if (nf >= 12) return -4; T->mdct_full_row[nf++] = (uint8_t) m; }
                else if (T->mdct_nterm[m] >= 1 && T->mdct_nterm[m] <= 6) { if (nsm >= 6) return -4; T->mdct_small_row[nsm++] = (uint8_t) m; }
                else return -4;
            }
            if (nf != 12 || nsm != 6) return -4;
        }
        {   /* ... and the compile-time copy of that shape k_mdct is built on (fbmdct_dev.h) */
            bool ok = true;
            for (int c = 0; c < 6; c++) for (int i = 0; i < 6; i++) ok = ok && T->mdct_g_ops[c][i] == MDCT_G_OPS[c][i];
            for (int c = 0; c < 2; c++) for (int i = 0; i < 18; i++) ok = ok && T->mdct_h_ops[c][i] == MDCT_H_OPS[c][i];
            for (int r = 0; r < 12; r++) ok = ok && T->mdct_full_row[r] == MDCT_FULL_ROW[r];
            for (int r = 0; r < 6; r++) {
                const int m = T->mdct_small_row[r];
                ok = ok && m == MDCT_SMALL_ROW[r] && T->mdct_nterm[m] == MDCT_SMALL_NT[r];
                for (int t = 0; ok && t < MDCT_SMALL_NT[r]; t++) ok = ok && T->mdct_vidx[m][t] == MDCT_SMALL_IDX[r][t];
            }
            if (!ok) { fprintf(stderr, "mp3mi: the long-block MDCT's shape does not match MDCT_* (fbmdct_dev.h)\n"); return -4; }
        }
    }

    LIBM_TABLE(TB_POW_NINT, -1, T->pow_nint_tab, sizeof(T->pow_nint_tab),
               T->pow_nint_tab[0] = 0.0;
               for (int i = 1; i < 2048; i++) T->pow_nint_tab[i] = pow((double) i - 0.4054, 4.0 / 3.0);
               T->pow_nint_tab[2048] = HUGE_VAL);
    LIBM_TABLE(TB_POW43, -1, T->pow43, sizeof(T->pow43),
               for (int i = 0; i < MP3MI_POW43_N; i++) T->pow43[i] = pow((double) i, 4.0 / 3.0));
    LIBM_TABLE(TB_STEP, -1, T->step, sizeof(T->step),
               for (int i = 0; i < MP3MI_STEP_N; i++) T->step[i] = pow(2.0, (double) (MP3MI_STEP_MIN + i) * 0.25));
    LIBM_TABLE(TB_PRETAB_XR, -1, T->pretab_xr, sizeof(T->pretab_xr),
               for (int n = 0; n < 4; n++) T->pretab_xr[n] = pow(sqrt(2.), (double) n));
    LIBM_TABLE(TB_PRETAB_XMIN, -1, T->pretab_xmin, sizeof(T->pretab_xmin),
               for (int n = 0; n < 4; n++) T->pretab_xmin[n] = pow(sqrt(2.), 2.0 * (double) n));
    LIBM_TABLE(TB_SQRT2, -1, &T->sqrt2, sizeof(T->sqrt2), T->sqrt2 = sqrt(2.0));
    LIBM_TABLE(TB_LOG2, -1, &T->log2, sizeof(T->log2), T->log2 = log(2.0));

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
                /* the two groups with linbits have no third table: bits 10..11 of their cells hold how many of x, y are escapes
                   (== 15), so that k_loop's walk prices the linbits without testing the values (k_loop.hip, loop_walk_step) */
                if (gi == 6 || gi == 7) e |= (unsigned) ((c / 16 == 15) + (c % 16 == 15)) << 10;
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
#if !defined(MP3MI_TABLE_GEN)
    if (tb_failed) return -7;
#endif
    return 0;
}

/* ---- pins: the tables as the reference-equivalent environment builds them ----
 * One changed bit in a member that came out of libm (windows, twiddles, spreading function, MDCT cosines, the power
 * tables) changes the bitstream.  tables_pins.h holds an FNV-1a hash of every member for each sampling rate, generated
 * where the golden vectors were generated (tools/gen_table_pins.py: glibc 2.35, the build the reference's goldens come
 * from).  The generator refuses to write a blob whose tables do not hash to the pins (another libm); the library
 * checks the tables it assembled from the blob against them on every build -- a damaged or mismatched blob fails
 * loudly (MP3MI_ERR_TABLES) instead of emitting a different stream.  There is no switch to turn the check off. */
#define MP3MI_TABLE_MEMBERS(X) \
    X(rate_idx) X(sfb_l) X(sfb_s) X(sfb_of_line_l) X(sfb_of_line_s) X(nj_first) X(nj_count) X(nj_job0) X(nj_njobs) X(nj_max) \
    X(nj_seg) X(lane_bands) X(lane_jobs) X(lane_inv_lines) X(subdiv_lut) X(window) X(window_s) X(numlines_pe) X(part_l_start) X(part_s_start) X(part_l_covered) \
    X(part_s_covered) X(minval) X(qthr_l) X(norm_l) X(qthr_s) X(exp_snr_s) X(s3_l) X(s3_lt) X(s3_lo) X(s3_hi) X(bu_l) X(bo_l) X(bu_s) \
    X(bo_s) X(w1_l) X(w2_l) X(w1_s) X(w2_s) X(fft_nround_l) X(fft_nround_s) X(fft_nword_l) X(fft_nword_s) X(fft_hdr_l) \
    X(fft_hdr_s) X(fft_prog_l) X(fft_prog_s) X(fft_rd_l) X(fft_rd_s) X(fft_regtw_l) X(fft_regtw_s) X(fft_leaf_l) X(fft_leaf_s) X(enwindow) X(filt) X(mdct_win) X(cos_s) X(cos_l) X(ca) \
    X(cs) X(mdct_vidx) X(mdct_nterm) X(mdct_full_row) X(mdct_small_row) X(mdct_g_ops) X(mdct_h_ops) X(mdct_vcoef) \
    X(pow_nint_tab) X(pow43) X(step) X(pretab_xr) X(pretab_xmin) X(sqrt2) X(log2) X(ht_off) X(ht_xlen) X(ht_ylen) \
    X(ht_linbits) X(ht_linmax) X(ht_len) X(ht_code) X(glut)

#define X(m) +1
enum { MP3MI_N_TABLE_MEMBERS = 0 MP3MI_TABLE_MEMBERS(X) };
#undef X
#define X(m) #m,
static const char *const TABLE_MEMBER_NAMES[MP3MI_N_TABLE_MEMBERS] = {MP3MI_TABLE_MEMBERS(X)};
#undef X
#include "tables_l12_pins.h"
#include "tables_pins.h" /* MP3MI_TABLE_PINS_N, MP3MI_TABLE_PINS[3][MP3MI_TABLE_PINS_N] */

static uint64_t fnv1a64(const void *p, size_t n)
{
    const unsigned char *b = (const unsigned char *) p;
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001b3ull; }
    return h;
}

static void table_hashes(const mp3mi_tables *T, uint64_t *out)
{
    int i = 0;
#define X(m) out[i++] = fnv1a64(&T->m, sizeof(T->m));
    MP3MI_TABLE_MEMBERS(X)
#undef X
}

/* hashes of the table members for rate_idx as THIS host builds them (no comparison with the pins); names[i]
 * receives the member names.  Returns the number of members, or a negative error. */
static std::mutex g_tables_mutex; /* one table build at a time (batches may be created from several threads: mp3mi.h) */

extern "C" int mp3mi_tables_digest(int rate_idx, uint64_t *hashes, const char **names, int cap)
{
    if (rate_idx < 0 || rate_idx > 2 || !hashes || cap < MP3MI_N_TABLE_MEMBERS) return -1;
    std::lock_guard<std::mutex> lock(g_tables_mutex);
    mp3mi_tables *T = (mp3mi_tables *) calloc(1, sizeof(mp3mi_tables));
    if (!T) return -1;
    const int rc = build_tables_unpinned(T, rate_idx);
    if (rc == 0) {
        table_hashes(T, hashes);
        if (names) for (int i = 0; i < MP3MI_N_TABLE_MEMBERS; i++) names[i] = TABLE_MEMBER_NAMES[i];
    }
    free(T);
    return rc == 0 ? MP3MI_N_TABLE_MEMBERS : rc;
}

extern "C" int mp3mi_build_tables(mp3mi_tables *T, int rate_idx)
{
    std::lock_guard<std::mutex> lock(g_tables_mutex);
    const int rc = build_tables_unpinned(T, rate_idx);
    if (rc == -7) return -8; /* the blob lacks an entry or is damaged: reported as MP3MI_ERR_TABLES */
    if (rc != 0) return rc;
    if (MP3MI_TABLE_PINS_N != MP3MI_N_TABLE_MEMBERS) {
        fprintf(stderr, "mp3mi: tables_pins.h lists %d members, mp3mi_tables has %d -- regenerate it (tools/gen_table_pins.py)\n",
                (int) MP3MI_TABLE_PINS_N, (int) MP3MI_N_TABLE_MEMBERS);
        return -8;
    }
    uint64_t h[MP3MI_N_TABLE_MEMBERS];
    table_hashes(T, h);
    int bad = 0;
    for (int i = 0; i < MP3MI_N_TABLE_MEMBERS; i++)
        if (h[i] != MP3MI_TABLE_PINS[rate_idx][i]) {
            fprintf(stderr, "mp3mi: table member '%s' (rate index %d) differs from its pinned value (csrc/tables_pins.h): the table blob and "
                            "the table code do not belong together; the bitstream would not be bit-exact\n",
                    TABLE_MEMBER_NAMES[i], rate_idx);
            bad++;
        }
    return bad ? -8 : 0;
}


/* ---- Layers I and II (l12_dev.h): psychoacoustic model 2 of src/psy.c and the coding tables ----
 * Everything but the spreading function is plain float / double arithmetic on constants (the FLOAT intermediates of
 * src/psy.c:151-198 kept as the reference declares them); the spreading function takes a square root and an
 * exponential and comes out of the blob like the Layer III tables.  Pinned by its hash per sampling rate. */
static const uint64_t L12_SPREAD_PINS[3] = {L12_SPREAD_PIN_0, L12_SPREAD_PIN_1, L12_SPREAD_PIN_2};

static int build_tables_l12_unpinned(mp3mi_tables_l12 *T, int ri, int layer, float (*s)[L12_CB])
{
    static const float crit_band[27] = {0, 100, 200, 300, 400, 510, 630, 770, 920, 1080, 1270, 1480, 1720, 2000, 2320, 2700,
                                        3150, 3700, 4400, 5300, 6400, 7700, 9500, 12000, 15500, 25000, 30000};
    static const float bmax[27] = {20.0, 20.0, 20.0, 20.0, 20.0, 17.0, 15.0, 10.0, 7.0, 4.4, 4.5, 4.5, 4.5, 4.5,
                                   4.5, 4.5, 4.5, 4.5, 4.5, 4.5, 4.5, 4.5, 4.5, 4.5, 3.5, 3.5, 3.5};
    static const double rates[3] = {44100.0, 48000.0, 32000.0};
    static const double SNR[18] = {0.00, 7.00, 11.00, 16.00, 20.84, 25.28, 31.59, 37.75, 43.84, /* src/encode.c:777-780 */
                                   49.89, 55.93, 61.96, 67.98, 74.01, 80.03, 86.05, 92.01, 98.01};
    static const double QA[17] = {0.750000000, 0.625000000, 0.875000000, 0.562500000, 0.937500000, /* src/encode.c:1195-1199 */
                                  0.968750000, 0.984375000, 0.992187500, 0.996093750, 0.998046875,
                                  0.999023438, 0.999511719, 0.999755859, 0.999877930, 0.999938965,
                                  0.999969482, 0.999984741};
    static const double QB[17] = {-0.250000000, -0.375000000, -0.125000000, -0.437500000, -0.062500000, /* :1201-1205 */
                                  -0.031250000, -0.015625000, -0.007812500, -0.003906250, -0.001953125,
                                  -0.000976563, -0.000488281, -0.000244141, -0.000122070, -0.000061035,
                                  -0.000030518, -0.000015259};
    if (ri < 0 || ri > 2 || (layer != 1 && layer != 2)) return -1;
    memset(T, 0, sizeof(*T));
    T->rate_idx = ri;
    T->layer = layer;
    float fthr[L12_HBLK], cbval[L12_CB], rnorm[L12_CB], freq_mult, bval_lo;
    int numlines[L12_CB], partition[L12_HBLK];
    double temp1, temp2;
    unsigned i, j;
    memset(cbval, 0, sizeof(cbval)); memset(numlines, 0, sizeof(numlines)); /* mem_alloc zero-fills, src/common.c:541 */
    /* src/psy.c:165-198 */
    freq_mult = rates[ri] / 1024;
    for (i = 0; i < L12_HBLK; i++) {
        temp1 = i * freq_mult;
        j = 1;
        while (temp1 > crit_band[j]) j++;
        fthr[i] = j - 1 + (temp1 - crit_band[j - 1]) / (crit_band[j] - crit_band[j - 1]);
    }
    partition[0] = 0;
    temp2 = 1;
    cbval[0] = fthr[0];
    bval_lo = fthr[0];
    for (i = 1; i < L12_HBLK; i++) {
        if ((fthr[i] - bval_lo) > 0.33) {
            partition[i] = partition[i - 1] + 1;
            if (partition[i] >= L12_CB) return -2;
            cbval[partition[i - 1]] = cbval[partition[i - 1]] / temp2;
            cbval[partition[i]] = fthr[i];
            bval_lo = fthr[i];
            numlines[partition[i - 1]] = temp2;
            temp2 = 1;
        } else {
            partition[i] = partition[i - 1];
            cbval[partition[i]] += fthr[i];
            temp2++;
        }
    }
    numlines[partition[i - 1]] = temp2;
    cbval[partition[i - 1]] = cbval[partition[i - 1]] / temp2;
    T->npart = partition[L12_HBLK - 1] + 1;
    /* the spreading function, src/psy.c:200-216 */
    LIBM_TABLE(TB_L12_SPREAD, ri, s, sizeof(float) * L12_CB * L12_CB,
        for (int jj = 0; jj < L12_CB; jj++)
            for (int ii = 0; ii < L12_CB; ii++) {
                double t1 = (cbval[ii] - cbval[jj]) * 1.05, t2, t3;
                if (t1 >= 0.5 && t1 <= 2.5) {
                    t2 = t1 - 0.5;
                    t2 = 8.0 * (t2 * t2 - 2.0 * t2);
                } else t2 = 0;
                t1 += 0.474;
                t3 = 15.811389 + 7.5 * t1 - 17.5 * sqrt((double) (1.0 + t1 * t1));
                if (t3 <= -100) s[ii][jj] = 0;
                else {
                    t3 = (t2 + t3) * R_LN_TO_LOG10;
                    s[ii][jj] = exp(t3);
                }
            });
    for (j = 0; j < L12_CB; j++) { /* src/psy.c:219-228 */
        temp1 = 15.5 + cbval[j];
        T->tmn[j] = (temp1 > 24.5) ? temp1 : 24.5;
        rnorm[j] = 0;
        for (i = 0; i < L12_CB; i++) rnorm[j] += s[j][i];
    }
    for (j = 0; j < L12_CB; j++) {
        for (i = 0; i < L12_CB; i++) T->spread_r[j][i] = s[j][i];
        T->cbval[j] = cbval[j];
        T->rnorm[j] = rnorm[j];
        unsigned k = cbval[j] + 0.5; /* src/psy.c:336 */
        if (k >= 27) return -2;
        T->bmaxv[j] = bmax[k];
        T->rn_nl[j] = (rnorm[j] && numlines[j]) ? rnorm[j] * numlines[j] : 0.0f; /* src/psy.c:346-347 */
    }
    for (i = 0; i < L12_HBLK; i++) {
        T->absthr[i] = T12_ABSTHR[ri][i];
        T->partition[i] = (uint8_t) partition[i];
        if (i == 0 || partition[i] != partition[i - 1]) T->part_first[partition[i]] = (int16_t) i;
    }
    for (int b = T->npart; b <= 64; b++) T->part_first[b] = L12_HBLK;
    memcpy(T->multiple, T12_MULTIPLE, sizeof(T->multiple));
    memcpy(T->snr, SNR, sizeof(SNR)); memcpy(T->qa, QA, sizeof(QA)); memcpy(T->qb, QB, sizeof(QB));
    if (layer == 1) { /* what the layer's first frame does to the shared statics, src/encode.c:900-905, 1226-1231 */
        T->snr[2] = T->snr[3];
        for (i = 3; i < 16; i++) T->snr[i] = T->snr[i + 2];
        T->qa[1] = T->qa[2]; T->qb[1] = T->qb[2];
        for (i = 2; i < 15; i++) { T->qa[i] = T->qa[i + 2]; T->qb[i] = T->qb[i + 2]; }
    }
    memcpy(T->alloc, T12_ALLOC, sizeof(T->alloc));
    for (i = 0; i < 4; i++) T->sblimit[i] = T12_ALLOC_SBLIMIT[i];
    return 0;
}

extern "C" int mp3mi_build_tables_l12(mp3mi_tables_l12 *T, int rate_idx, int layer)
{
    std::lock_guard<std::mutex> lock(g_tables_mutex);
    static float s[L12_CB][L12_CB];
    const int rc = build_tables_l12_unpinned(T, rate_idx, layer, s);
    if (rc == -7) return -8;
    if (rc != 0) return rc;
    if (fnv1a64(s, sizeof(s)) != L12_SPREAD_PINS[rate_idx]) {
        fprintf(stderr, "mp3mi: the Layer I/II spreading function (rate index %d) differs from its pinned value: the table blob and the "
                        "table code do not belong together\n", rate_idx);
        return -8;
    }
    return 0;
}

#if defined(MP3MI_TABLE_GEN)
/* The generator (make -C csrc blob): builds the tables of the three rates with THIS host's libm, checks them against the
 * pins -- this host must be the environment the goldens come from -- and writes tables_blob.bin.
 *   gen_table_blob out.bin [--no-pin-check]     (the latter only to bootstrap new pins after a layout change) */
int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s out.bin [--no-pin-check]\n", argv[0]); return 2; }
    const bool check = !(argc > 2 && !strcmp(argv[2], "--no-pin-check"));
    mp3mi_tables *T = (mp3mi_tables *) calloc(1, sizeof(mp3mi_tables));
    for (int ri = 0; ri < 3; ri++) {
        const int rc = build_tables_unpinned(T, ri);
        if (rc != 0) { fprintf(stderr, "table build failed for rate index %d: %d\n", ri, rc); return 1; }
        if (check) {
            uint64_t h[MP3MI_N_TABLE_MEMBERS];
            table_hashes(T, h);
            for (int i = 0; i < MP3MI_N_TABLE_MEMBERS; i++)
                if (MP3MI_TABLE_PINS_N != MP3MI_N_TABLE_MEMBERS || h[i] != MP3MI_TABLE_PINS[ri][i]) {
                    fprintf(stderr, "member '%s' (rate index %d) differs from its pin: this host's libm is not the reference environment's; "
                                    "no blob written\n", TABLE_MEMBER_NAMES[i], ri);
                    return 1;
                }
        }
    }
    free(T);
    {   /* Layers I / II: the spreading function of the three rates */
        static mp3mi_tables_l12 T12;
        static float sp[L12_CB][L12_CB];
        for (int ri = 0; ri < 3; ri++) {
            const int rc = build_tables_l12_unpinned(&T12, ri, 2, sp);
            if (rc != 0) { fprintf(stderr, "Layer I/II table build failed for rate index %d: %d\n", ri, rc); return 1; }
            const uint64_t h = fnv1a64(sp, sizeof(sp));
            printf("L12_SPREAD_PIN_%d 0x%016llxull\n", ri, (unsigned long long) h);
            if (check && h != L12_SPREAD_PINS[ri]) { fprintf(stderr, "Layer I/II spreading function (rate index %d) differs from its pin; no blob written\n", ri); return 1; }
        }
    }
    tb_header hd;
    memset(&hd, 0, sizeof(hd));
    memcpy(hd.magic, "MP3MITB1", 8);
    hd.n_entries = (uint32_t) tb_entries.size();
    std::vector<unsigned char> out((const unsigned char *) &hd, (const unsigned char *) &hd + sizeof(hd));
    out.insert(out.end(), (const unsigned char *) tb_entries.data(), (const unsigned char *) (tb_entries.data() + tb_entries.size()));
    out.insert(out.end(), tb_data.begin(), tb_data.end());
    const uint64_t h = tb_fnv(out.data(), out.size());
    out.insert(out.end(), (const unsigned char *) &h, (const unsigned char *) &h + 8);
    FILE *f = fopen(argv[1], "wb");
    if (!f || fwrite(out.data(), 1, out.size(), f) != out.size()) { perror(argv[1]); return 1; }
    fclose(f);
    printf("%s: %zu entries, %zu bytes\n", argv[1], tb_entries.size(), out.size());
    return 0;
}
#endif
