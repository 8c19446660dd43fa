// Experiment (round 6, F10): does a WIDER family of swizzles than the six XOR columns of MP3MI_FFT_SWZ_COLS spread the operands of the
// shipped rounds better?  pos = e ^ Fa[(e >> 4) & 3] ^ Fb[e >> 6] keeps what the kernels rely on (SWZ(lane + 64 k) == SWZ(lane) ^ SWZ(64 k))
// with Fa[3] and all of Fb[1..15] free (80 bits instead of 29).  Annealing over the map's bits for balanced bank histograms, then
// over the lanes of every round: no better than the shipped columns (long: 890 - 899 against 850 for the shipped map under the
// same lane annealing).   usage: fft_swz_family <1 long | 0 short> <0 columns only | 1 wide family>
//   cd mp3-enc-bsd_amd/csrc && g++ -O2 -ffp-contract=off -std=c++17 -DMP3MI_EMU -I. -I../../include -I../../tests/hipemu \
//       ../../tools/exp/fft_swz_family.cpp tables_host.cpp build/tables_blob.o -o /tmp/fft_swz_family -lm
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <random>
#include <math.h>
#include <algorithm>
#include <vector>
#include "mp3mi_host.h"
static int fft_round_words(int h) { return ((h & 1) ? 256 : 128) + ((h & 2) ? ((h & 1) ? 512 : 256) : 64); }
static int colcost(const unsigned *pos)
{
    int total = 0;
    for (int g = 0; g < 64; g += 16) { int cnt[16] = {0}, mx = 0; for (int l = g; l < g + 16; l++) { int c = ++cnt[pos[l] & 15]; mx = c > mx ? c : mx; } total += mx; }
    for (int g = 0; g < 64; g += 32) { int cnt[32] = {0}, mx = 0; for (int l = g; l < g + 32; l++) { int c = ++cnt[pos[l] & 31]; mx = c > mx ? c : mx; } total += mx; }
    return total;
}
struct Round { int N; unsigned e[8][64]; };
static std::vector<Round> rounds;
static unsigned Fa[4], Fb[16];
static int NB; // entries of Fb in use
static inline unsigned S(unsigned e) { return e >= 1024u ? e : e ^ Fa[(e >> 4) & 3] ^ Fb[e >> 6]; }
static std::mt19937_64 rng(7);
static int proxy()
{
    int tot = 0;
    for (const Round &R : rounds)
        for (int k = 0; k < R.N; k++) {
            int c32[32] = {0}, c16[16] = {0};
            for (int l = 0; l < 64; l++) { const unsigned p = S(R.e[k][l]); c32[p & 31]++; c16[p & 15]++; }
            for (int i = 0; i < 32; i++) tot += c32[i] > 2 ? (c32[i] - 2) * (c32[i] - 2) : 0;
            for (int i = 0; i < 16; i++) tot += c16[i] > 4 ? (c16[i] - 4) : 0;
        }
    return tot;
}
static int annealed(long tries, bool verbose)
{
    std::uniform_real_distribution<double> U(0, 1);
    int total = 0, ideal = 0, r = 0;
    for (const Round &R : rounds) {
        unsigned p[8][64]; int perm[64];
        for (int l = 0; l < 64; l++) { perm[l] = l; for (int k = 0; k < R.N; k++) p[k][l] = S(R.e[k][l]); }
        auto cost = [&]() { int c = 0; unsigned q[64]; for (int k = 0; k < R.N; k++) { for (int l = 0; l < 64; l++) q[l] = p[k][perm[l]]; c += colcost(q); } return c; };
        int cur = cost(), best = cur;
        for (long t = 0; t < tries && best > 6 * R.N; t++) {
            const double temp = 1.0 * pow(0.05, (double) t / tries);
            const int a = rng() % 64, b = rng() % 64;
            if ((a >> 4) == (b >> 4)) continue;
            std::swap(perm[a], perm[b]);
            const int c = cost();
            if (c <= cur || U(rng) < exp(-(double) (c - cur) / temp)) { cur = c; if (c < best) best = c; }
            else std::swap(perm[a], perm[b]);
        }
        if (verbose) printf("  round %2d: %3d (conflict-free %d)\n", r, best, 6 * R.N);
        total += best; ideal += 6 * R.N; r++;
    }
    if (verbose) printf("  total %d, conflict-free %d\n", total, ideal);
    return total;
}
int main(int argc, char **argv)
{
    static mp3mi_tables T;
    mp3mi_build_tables(&T, 0);
    const int L = argc > 1 ? atoi(argv[1]) : 1;
    const int mode = argc > 2 ? atoi(argv[2]) : 0; // 0: linear columns only (the shipped family), 1: free Fa[3] and Fb[]
    static int inv[2048];
    for (int q = 0; q < 2048; q++) inv[q] = q;
    for (int q = 0; q < 1024; q++) inv[MP3MI_FFT_SWZ(q)] = q;
    const int nr = L ? T.fft_nround_l : T.fft_nround_s;
    const uint32_t *hdr = L ? T.fft_hdr_l : T.fft_hdr_s, *prog = L ? T.fft_prog_l : T.fft_prog_s;
    int off = 0;
    for (int r = 0; r < nr; r++) {
        const int h = (int) hdr[r], N = (h & 1) ? 8 : 4, aw = N / 2;
        Round R; R.N = N;
        for (int l = 0; l < 64; l++) for (int k = 0; k < N; k++) { const uint32_t w = prog[off + l * aw + k / 2]; const unsigned pos = (k & 1) ? (w >> 16) : (w & 0xffffu); R.e[k][l] = (L ? pos >= 1024u : pos >= 768u) ? 1024u + (pos & 63) : (unsigned) inv[pos]; }
        rounds.push_back(R);
        off += fft_round_words(h);
    }
    NB = L ? 16 : 12;
    // the shipped map in this parametrisation
    static const unsigned cols[6] = {MP3MI_FFT_SWZ_COLS};
    auto from_cols = [&](const unsigned *c) { Fa[0] = 0; Fa[1] = c[0]; Fa[2] = c[1]; Fa[3] = c[0] ^ c[1]; for (int h = 0; h < 16; h++) { Fb[h] = 0; for (int b = 0; b < 4; b++) if (h & (1 << b)) Fb[h] ^= c[2 + b]; } };
    from_cols(cols);
    printf("shipped map: proxy %d, annealed lanes %d\n", proxy(), annealed(100000, false));
    std::uniform_real_distribution<double> U(0, 1);
    int bestp = 1 << 30; unsigned bFa[4], bFb[16];
    for (int restart = 0; restart < 6; restart++) {
        unsigned c[6];
        for (int i = 0; i < 6; i++) c[i] = rng() & (i == 0 ? 15 : 31);
        from_cols(c);
        if (mode == 1 && !L) { /* short: windows stay linear: Fb[4 w + j] = Fb[4 w] ^ Fb[j] */ }
        int cur = proxy();
        const long tries = 200000;
        for (long t = 0; t < tries; t++) {
            const double temp = 3.0 * pow(0.1 / 3.0, (double) t / tries);
            unsigned sFa[4], sFb[16]; memcpy(sFa, Fa, sizeof(Fa)); memcpy(sFb, Fb, sizeof(Fb));
            if (mode == 0) { c[rng() % 6] ^= 1u << (rng() % 5); c[0] &= 15; from_cols(c); }
            else {
                const int which = rng() % (3 + (L ? 15 : 5));
                const unsigned bit = 1u << (rng() % 5);
                if (which < 3) { Fa[1 + which] ^= bit; Fa[1] &= 15; if ((Fa[2] ^ Fa[3]) & 16) Fa[3] ^= 16; }
                else if (L) Fb[1 + (which - 3)] ^= bit;
                else { static const int fr[5] = {1, 2, 3, 4, 8}; Fb[fr[which - 3]] ^= bit; for (int w = 1; w < 3; w++) for (int j = 1; j < 4; j++) Fb[4 * w + j] = Fb[4 * w] ^ Fb[j]; }
            }
            const int p = proxy();
            if (p <= cur || U(rng) < exp(-(double) (p - cur) / temp)) { cur = p; if (p < bestp) { bestp = p; memcpy(bFa, Fa, sizeof(Fa)); memcpy(bFb, Fb, sizeof(Fb)); } }
            else { memcpy(Fa, sFa, sizeof(Fa)); memcpy(Fb, sFb, sizeof(Fb)); if (mode == 0) { c[0] = Fa[1]; c[1] = Fa[2]; c[2] = Fb[1]; c[3] = Fb[2]; c[4] = Fb[4]; c[5] = Fb[8]; } }
        }
        printf("restart %d: proxy %d (best so far %d)\n", restart, cur, bestp);
        fflush(stdout);
    }
    memcpy(Fa, bFa, sizeof(Fa)); memcpy(Fb, bFb, sizeof(Fb));
    printf("best: proxy %d; Fa = %u %u %u %u; Fb =", bestp, Fa[0], Fa[1], Fa[2], Fa[3]);
    for (int h = 0; h < NB; h++) printf(" %u", Fb[h]);
    printf("\n");
    annealed(400000, true);
    return 0;
}
