// Experiment (round 6, F10): the LDS cycles of the SHIPPED butterfly programs under the model of FftGen::round_cycles (an 8-byte load is
// served 32 lanes a cycle and collides on pos & 31, an 8-byte store 16 lanes a cycle and collides on pos & 15), next to the
// conflict-free count (6 cycles per operand column): the excess is what SQ_LDS_BANK_CONFLICT counts per task on the device
// (long: 885 - 576 = 309 modelled, 309 counted -- profiles/r06_experiments.txt F8 / F10).
//   cd mp3-enc-bsd_amd/csrc && g++ -O2 -ffp-contract=off -std=c++17 -DMP3MI_EMU -I. -I../../include -I../../tests/hipemu \
//       ../../tools/exp/fft_model_cycles.cpp tables_host.cpp build/tables_blob.o -o /tmp/fft_model_cycles -lm
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include "mp3mi_host.h"
static int fft_round_words(int h) { return ((h & 1) ? 256 : 128) + ((h & 2) ? ((h & 1) ? 512 : 256) : 64); }
int main()
{
    static mp3mi_tables T;
    if (mp3mi_build_tables(&T, 0) != 0) { printf("build failed\n"); return 1; }
    for (int L = 1; L >= 0; L--) {
        const int nr = L ? T.fft_nround_l : T.fft_nround_s;
        const uint32_t *hdr = L ? T.fft_hdr_l : T.fft_hdr_s, *prog = L ? T.fft_prog_l : T.fft_prog_s;
        int off = 0, tot_ld = 0, tot_st = 0, ideal_ld = 0, ideal_st = 0;
        for (int r = 0; r < nr; r++) {
            const int h = (int) hdr[r], N = (h & 1) ? 8 : 4, aw = N / 2;
            const uint32_t *blk = prog + off;
            int rl = 0, rs = 0;
            for (int k = 0; k < N; k++) {
                unsigned pos[64];
                for (int l = 0; l < 64; l++) { const uint32_t w = blk[l * aw + k / 2]; pos[l] = (k & 1) ? (w >> 16) : (w & 0xffffu); }
                for (int g = 0; g < 64; g += 32) { int cnt[32] = {0}, mx = 0; for (int l = g; l < g + 32; l++) { int c = ++cnt[pos[l] & 31]; if (c > mx) mx = c; } rl += mx; }
                for (int g = 0; g < 64; g += 16) { int cnt[16] = {0}, mx = 0; for (int l = g; l < g + 16; l++) { int c = ++cnt[pos[l] & 15]; if (c > mx) mx = c; } rs += mx; }
            }
            printf("%s round %2d hdr %2d: operands %d  load cycles %3d (ideal %2d)  store cycles %3d (ideal %2d)\n", L ? "long " : "short", r, h, N, rl, 2 * N, rs, 4 * N);
            tot_ld += rl; tot_st += rs; ideal_ld += 2 * N; ideal_st += 4 * N;
            off += fft_round_words(h);
        }
        printf("%s: load %d (ideal %d), store %d (ideal %d): modelled conflict cycles %d\n", L ? "long" : "short", tot_ld, ideal_ld, tot_st, ideal_st, tot_ld - ideal_ld + tot_st - ideal_st);
    }
    return 0;
}
