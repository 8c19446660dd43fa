// Experiment (round 6, F10): how close to conflict-free can the 64 butterflies of each round of the shipped programs be arranged on
// the lanes when NOTHING moves between rounds -- annealing over lane swaps, one round at a time.  (Long 885 -> 847 of 576, short
// 708 -> 679 of 432: what is left is the SET of butterflies a step holds, not their order.)
//   cd mp3-enc-bsd_amd/csrc && g++ -O2 -ffp-contract=off -std=c++17 -DMP3MI_EMU -I. -I../../include -I../../tests/hipemu \
//       ../../tools/exp/fft_round_floor.cpp tables_host.cpp build/tables_blob.o -o /tmp/fft_round_floor -lm
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <random>
#include <math.h>
#include <algorithm>
#include "mp3mi_host.h"
static int fft_round_words(int h) { return ((h & 1) ? 256 : 128) + ((h & 2) ? ((h & 1) ? 512 : 256) : 64); }
static int colcost(const unsigned *pos)
{
    int total = 0;
    for (int g = 0; g < 64; g += 16) { int cnt[16] = {0}, mx = 0; for (int l = g; l < g + 16; l++) { int c = ++cnt[pos[l] & 15]; mx = c > mx ? c : mx; } total += mx; }
    for (int g = 0; g < 64; g += 32) { int cnt[32] = {0}, mx = 0; for (int l = g; l < g + 32; l++) { int c = ++cnt[pos[l] & 31]; mx = c > mx ? c : mx; } total += mx; }
    return total;
}
int main()
{
    static mp3mi_tables T;
    mp3mi_build_tables(&T, 0);
    std::mt19937_64 rng(1);
    std::uniform_real_distribution<double> U(0, 1);
    for (int L = 1; L >= 0; L--) {
        const int nr = L ? T.fft_nround_l : T.fft_nround_s;
        const uint32_t *hdr = L ? T.fft_hdr_l : T.fft_hdr_s, *prog = L ? T.fft_prog_l : T.fft_prog_s;
        int off = 0, tot0 = 0, tot1 = 0, ideal = 0;
        for (int r = 0; r < nr; r++) {
            const int h = (int) hdr[r], N = (h & 1) ? 8 : 4, aw = N / 2;
            const uint32_t *blk = prog + off;
            unsigned p[8][64]; int perm[64];
            for (int l = 0; l < 64; l++) { perm[l] = l; for (int k = 0; k < N; k++) { const uint32_t w = blk[l * aw + k / 2]; p[k][l] = (k & 1) ? (w >> 16) : (w & 0xffffu); } }
            auto cost = [&]() { int c = 0; unsigned q[64]; for (int k = 0; k < N; k++) { for (int l = 0; l < 64; l++) q[l] = p[k][perm[l]]; c += colcost(q); } return c; };
            int cur = cost(); const int start = cur; int best = cur;
            const long tries = 400000;
            for (long t = 0; t < tries && best > 6 * N; t++) {
                const double temp = 1.0 * pow(0.05 / 1.0, (double) t / tries);
                const int a = rng() % 64, b = rng() % 64;
                if ((a >> 4) == (b >> 4)) continue;
                std::swap(perm[a], perm[b]);
                const int c = cost();
                if (c <= cur || U(rng) < exp(-(double) (c - cur) / temp)) { cur = c; if (c < best) best = c; }
                else std::swap(perm[a], perm[b]);
            }
            printf("%s round %2d (%d operands): as shipped %3d, lanes of this round rearranged %3d, conflict-free %3d\n", L ? "long " : "short", r, N, start, best, 6 * N);
            tot0 += start; tot1 += best; ideal += 6 * N;
            off += fft_round_words(h);
        }
        printf("%s total: shipped %d, per-round rearranged %d, ideal %d\n", L ? "long" : "short", tot0, tot1, ideal);
    }
    return 0;
}
