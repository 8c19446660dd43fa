// Test helper (tests/test_fft_tables.py): properties of the FFT tables mp3mi_build_tables produces, checked on the host.
//   - the leaves (mp3mi_tables::fft_leaf_*, k_fft.hip fft_leaves): every lane's eight pairs are whole pairs (even positions), the
//     runs of all lanes together cover every element of the transform(s) exactly once, and the kinds come in the numbers the
//     recursion has (1024 points: 42 x C(8), 21 x two C(4), 1 x R(8) + C(4); three times 256: 30, 15, 3);
//   - the program: every round's operands lie inside the arrays (or on the idle lanes' dummy elements), no element is an
//     operand of two butterflies of one round, and the LDS cycles of all rounds under the placement's model
//     (an 8-byte load collides on position mod 32 within 32 lanes, a store on position mod 16 within 16) are printed.
// Prints one line per transform: "<long|short> rounds R words W leaves L kinds a b c idle d cycles C conflict_free F".
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <vector>
#include "mp3mi_host.h"

static int round_words(int h) { return ((h & 1) ? 256 : 128) + ((h & 2) ? ((h & 1) ? 512 : 256) : 64); }
static int fail(const char *what, int a, int b) { printf("FAIL %s (%d, %d)\n", what, a, b); return 1; }

int main(int argc, char **argv)
{
    static mp3mi_tables T;
    for (int ri = 0; ri < 3; ri++) {
        if (mp3mi_build_tables(&T, ri) != 0) return fail("table build", ri, 0);
        for (int L = 1; L >= 0; L--) {
            const int n_elem = L ? 1024 : 768, dummy = L ? MP3MI_FFT_DUMMY : MP3MI_FFT_DUMMY_S;
            const uint32_t *leaf = L ? T.fft_leaf_l : T.fft_leaf_s;
            std::vector<int> seen((size_t) n_elem, 0);
            int kinds[4] = {0, 0, 0, 0};
            for (int l = 0; l < 64; l++) {
                const uint32_t *w = leaf + 4 * l;
                const int kind = (int) ((w[0] >> 14) & 3u);
                kinds[kind]++;
                if (kind == 3) continue;
                for (int j = 0; j < 8; j++) {
                    uint32_t word = w[j / 2];
                    if (j / 2 == 0) word &= 0x3fff3fffu;
                    const int pos = (int) ((j & 1) ? (word >> 16) : (word & 0xffffu));
                    if ((pos & 1) || pos + 1 >= n_elem) return fail("leaf pair position", l, pos);
                    seen[(size_t) pos]++;
                    seen[(size_t) pos + 1]++;
                }
                for (int run = 0; run < 2; run++) { // the four pairs of a run lie in ONE aligned run of 8 positions
                    int base = -1;
                    for (int j = 4 * run; j < 4 * run + 4; j++) {
                        uint32_t word = w[j / 2];
                        if (j / 2 == 0) word &= 0x3fff3fffu;
                        const int pos = (int) ((j & 1) ? (word >> 16) : (word & 0xffffu));
                        if (base < 0) base = pos & ~7;
                        if ((pos & ~7) != base) return fail("leaf run", l, pos);
                    }
                }
            }
            for (int e = 0; e < n_elem; e++) if (seen[(size_t) e] != 1) return fail("leaf coverage", e, seen[(size_t) e]);
            const int want[2][4] = {{30, 15, 3, 16}, {42, 21, 1, 0}};
            for (int k = 0; k < 4; k++) if (kinds[k] != want[L][k]) return fail("leaf kinds", k, kinds[k]);
            const int nr = L ? T.fft_nround_l : T.fft_nround_s, nw = L ? T.fft_nword_l : T.fft_nword_s;
            const uint32_t *hdr = L ? T.fft_hdr_l : T.fft_hdr_s, *prog = L ? T.fft_prog_l : T.fft_prog_s;
            int off = 0, cycles = 0, ideal = 0;
            for (int r = 0; r < nr; r++) {
                const int h = (int) hdr[r], N = (h & 1) ? 8 : 4, aw = N / 2;
                std::vector<int> used((size_t) dummy + 64, 0);
                for (int k = 0; k < N; k++) {
                    unsigned pos[64];
                    for (int l = 0; l < 64; l++) {
                        const uint32_t w = prog[off + l * aw + k / 2];
                        pos[l] = (k & 1) ? (w >> 16) : (w & 0xffffu);
                        if (pos[l] >= (unsigned) dummy + 64u || (pos[l] >= (unsigned) n_elem && pos[l] < (unsigned) dummy)) return fail("operand position", r, (int) pos[l]);
                        if (pos[l] < (unsigned) n_elem && used[pos[l]]++) return fail("operand of two butterflies of a round", r, (int) pos[l]);
                    }
                    for (int g = 0; g < 64; g += 32) { int cnt[32] = {0}, mx = 0; for (int l = g; l < g + 32; l++) { const int c = ++cnt[pos[l] & 31]; if (c > mx) mx = c; } cycles += mx; }
                    for (int g = 0; g < 64; g += 16) { int cnt[16] = {0}, mx = 0; for (int l = g; l < g + 16; l++) { const int c = ++cnt[pos[l] & 15]; if (c > mx) mx = c; } cycles += mx; }
                    ideal += 6;
                }
                off += round_words(h);
            }
            if (off != nw) return fail("program words", off, nw);
            if (ri == 0)
                printf("%s rounds %d words %d leaves %d kinds %d %d %d idle %d cycles %d conflict_free %d\n", L ? "long" : "short", nr, nw,
                       kinds[0] + kinds[1] + kinds[2], kinds[0], kinds[1], kinds[2], kinds[3], cycles, ideal);
        }
    }
    (void) argc; (void) argv;
    return 0;
}
