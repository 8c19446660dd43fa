// Generator aid (not part of the library): searches (a) the LDS swizzle of the FFT arrays -- which low address bits an
// element's index bits 4..9 are XORed into -- and (b) the placement of every rank's butterflies on rounds and lanes,
// for the fewest LDS cycles of the butterfly programs under the model of FftGen::round_cycles (an 8-byte store
// serves 16 lanes a cycle, an 8-byte load 32).  (b) is simulated annealing over swaps of butterflies of the same kind;
// its result is written as mp3-enc-bsd_amd/csrc/fft_placement.h, which tables_host.cpp replays.
//   g++ -O2 -std=c++17 -DMP3MI_EMU -DMP3MI_FFT_SWZ_RUNTIME -Imp3-enc-bsd_amd/csrc -Iinclude -Itests/hipemu \
//       tools/exp/fft_swz_search.cpp tests/hipemu/hipemu.cpp -o /tmp/fft_swz_search
//   /tmp/fft_swz_search search <seconds>                         candidates for the swizzle (greedy placement as the yardstick)
//   /tmp/fft_swz_search anneal <c4> .. <c9> <sweeps> [out.h]     anneal the placement under one swizzle, write the header
#include "../../mp3-enc-bsd_amd/csrc/tables_host.cpp"
#include <map>
#include <random>
#include <time.h>
#include <math.h>
unsigned mp3mi_fft_swz_col[10];

static FftGen *G;
static std::vector<uint32_t> prog_l(4 * MP3MI_FFT_PROG_WORDS), prog_s(4 * MP3MI_FFT_PROG_WORDS_S);
static uint32_t hdr[256], rdl[MP3MI_HBLK], rds_[MP3MI_HBLK_S];

static int cost_of(int logN, int nwin, uint32_t *prog, int maxw, uint32_t *rd, int *rounds, bool verbose)
{
    int32_t nr = 0;
    const int nw = G->build(logN, nwin, hdr, 256, &nr, prog, 4 * maxw, rd);
    int off = 0, tot = (nw > maxw || nr > MP3MI_FFT_MAX_ROUNDS) ? 1000000 : 0; // must fit the product's tables
    for (int r = 0; r < nr; r++) {
        const int h = (int) hdr[r], N = (h & 1) ? 8 : 4;
        for (int k = 0; k < N; k++) {
            unsigned pos[64];
            for (int l = 0; l < 64; l++) {
                const uint32_t w = prog[off + (N / 2) * l + k / 2];
                pos[l] = (k & 1) ? (w >> 16) : (w & 0xffff);
            }
            tot += FftGen::round_cycles(pos);
        }
        off += ((h & 1) ? 256 : 128) + ((h & 2) ? ((h & 1) ? 512 : 256) : 64);
    }
    if (rounds) *rounds = nr;
    if (verbose) {
        int ideal = 0;
        for (int r = 0; r < nr; r++) ideal += ((hdr[r] & 1) ? 8 : 4) * 6;
        printf("  logN %d: %d rounds, %d words, %d LDS cycles (conflict-free: %d)\n", logN, nr, nw, tot, ideal);
    }
    return tot;
}

static int total_cost(bool verbose = false)
{
    int rl, rs;
    const int a = cost_of(10, 1, prog_l.data(), MP3MI_FFT_PROG_WORDS, rdl, &rl, verbose);
    const int b = cost_of(8, 3, prog_s.data(), MP3MI_FFT_PROG_WORDS_S, rds_, &rs, verbose);
    return a + b;
}

// ---- annealing of one list ----
static long g_sweeps = 2000;
static std::map<std::tuple<int, int, int>, std::vector<uint16_t>> g_orders;
static std::mt19937_64 g_rng(20261003);

static void anneal_hook(int logN, int rank, int cls, std::vector<FusedOp> &placed, int nopnd)
{
    const size_t n = placed.size(), nround = (n + 63) / 64;
    if (n >= 2) {
        std::vector<unsigned> pos(nround * 8 * 64);
        std::vector<int> cyc(nround * 8);
        auto posof = [&](size_t i, int k) -> unsigned {
            if (i >= n) return (unsigned) (MP3MI_FFT_DUMMY + (i & 63));
            return placed[i].p[k] == MP3MI_FFT_DUMMY ? (unsigned) (MP3MI_FFT_DUMMY + (i & 63)) : placed[i].p[k];
        };
        long cur = 0;
        for (size_t r = 0; r < nround; r++)
            for (int k = 0; k < nopnd; k++) {
                for (int l = 0; l < 64; l++) pos[(r * 8 + k) * 64 + l] = posof(r * 64 + l, k);
                cur += cyc[r * 8 + k] = FftGen::round_cycles(&pos[(r * 8 + k) * 64]);
            }
        const long start = cur, ideal = (long) nround * nopnd * 6;
        std::vector<FusedOp> best = placed;
        long bestc = cur;
        const long tries = g_sweeps * (long) n;
        std::uniform_real_distribution<double> U(0.0, 1.0);
        for (long t = 0; t < tries && bestc > ideal; t++) {
            const double temp = 0.8 * pow(0.02 / 0.8, (double) t / (double) tries);
            const size_t i = g_rng() % n, j = g_rng() % n;
            if (i == j || placed[i].kind != placed[j].kind) continue;
            const size_t ri = i / 64, rj = j / 64;
            int before = 0, after = 0, ci[8], cj[8];
            for (int k = 0; k < nopnd; k++) { before += cyc[ri * 8 + k]; if (rj != ri) before += cyc[rj * 8 + k]; }
            std::swap(placed[i], placed[j]);
            for (int k = 0; k < nopnd; k++) { pos[(ri * 8 + k) * 64 + (i & 63)] = posof(i, k); pos[(rj * 8 + k) * 64 + (j & 63)] = posof(j, k); }
            for (int k = 0; k < nopnd; k++) {
                ci[k] = FftGen::round_cycles(&pos[(ri * 8 + k) * 64]);
                cj[k] = rj != ri ? FftGen::round_cycles(&pos[(rj * 8 + k) * 64]) : 0;
                after += ci[k] + cj[k];
            }
            const int d = after - before;
            if (d <= 0 || U(g_rng) < exp(-(double) d / temp)) {
                for (int k = 0; k < nopnd; k++) { cyc[ri * 8 + k] = ci[k]; if (rj != ri) cyc[rj * 8 + k] = cj[k]; }
                cur += d;
                if (cur < bestc) { bestc = cur; best = placed; }
            } else {
                std::swap(placed[i], placed[j]);
                for (int k = 0; k < nopnd; k++) { pos[(ri * 8 + k) * 64 + (i & 63)] = posof(i, k); pos[(rj * 8 + k) * 64 + (j & 63)] = posof(j, k); }
            }
        }
        placed = best;
        printf("    logN %2d rank %2d class %d: %4zu butterflies, %2zu rounds: %5ld -> %5ld cycles (conflict-free %ld)\n", logN, rank, cls, n, nround, start, bestc, ideal);
        fflush(stdout);
    }
    std::vector<uint16_t> ids;
    for (size_t i = 0; i < n; i++) ids.push_back((uint16_t) placed[i].id);
    g_orders[std::make_tuple(logN, rank, cls)] = ids;
}

int main(int argc, char **argv)
{
    G = new FftGen();
    for (int i = 4; i <= 10; i++) { G->tw_rs[i] = make_twiddle(i, false); G->tw_sr[i] = make_twiddle(i, true); }
    for (int b = 0; b < 10; b++) mp3mi_fft_swz_col[b] = b >= 5 ? 1u << (b - 5) : 0;
    if (argc >= 3 && !strcmp(argv[1], "search")) {
        const double budget = atof(argv[2]);
        printf("p ^ ((p >> 5) & 31): cost %d\n", total_cost(true));
        std::mt19937 rng(12345);
        int bestc = 1 << 30;
        const time_t t0 = time(NULL);
        while (difftime(time(NULL), t0) < budget) {
            for (int b = 4; b < 10; b++) mp3mi_fft_swz_col[b] = rng() & (b == 4 ? 15u : 31u);
            int c = total_cost();
            for (bool improved = true; improved;) {
                improved = false;
                for (int b = 4; b < 10; b++)
                    for (int bit = 0; bit < (b == 4 ? 4 : 5); bit++) {
                        mp3mi_fft_swz_col[b] ^= 1u << bit;
                        const int c2 = total_cost();
                        if (c2 < c) { c = c2; improved = true; }
                        else mp3mi_fft_swz_col[b] ^= 1u << bit;
                    }
            }
            if (c < bestc + 30) {
                if (c < bestc) bestc = c;
                printf("cost %d  cols[4..9] =", c);
                for (int b = 4; b < 10; b++) printf(" %u", mp3mi_fft_swz_col[b]);
                printf("\n");
                fflush(stdout);
            }
        }
        return 0;
    }
    if (argc >= 9 && !strcmp(argv[1], "anneal")) {
        for (int b = 4; b < 10; b++) mp3mi_fft_swz_col[b] = (unsigned) atoi(argv[2 + b - 4]);
        g_sweeps = atol(argv[8]);
        printf("greedy placement: cost %d\n", total_cost(true));
        G->placement_hook = anneal_hook;
        std::map<std::tuple<int, int, int>, std::vector<uint16_t>> all;
        const int c = total_cost(true);
        printf("annealed placement: cost %d\n", c);
        if (argc >= 10) {
            FILE *f = fopen(argv[9], "w");
            fprintf(f, "/* GENERATED by tools/exp/fft_swz_search.cpp (anneal %s %s %s %s %s %s, %ld sweeps) -- do not edit.\n", argv[2], argv[3], argv[4], argv[5], argv[6], argv[7], g_sweeps);
            fprintf(f, " * The order of every rank's butterflies (FusedOp::id) on the rounds and lanes of the FFT programs: %d LDS cycles\n * per long + three short transforms under the model of FftGen::round_cycles.  Valid for the swizzle named above\n * (MP3MI_FFT_SWZ_COLS, mp3mi_dev.h); tables_host.cpp falls back to its greedy order where a list does not fit. */\n", c);
            fprintf(f, "#define MP3MI_FFT_PLACEMENT_COLS %s, %s, %s, %s, %s, %s\n", argv[2], argv[3], argv[4], argv[5], argv[6], argv[7]);
            int idx = 0;
            for (auto &kv : g_orders) {
                fprintf(f, "static const uint16_t FFT_PLACE_%d[] = {", idx++);
                for (size_t i = 0; i < kv.second.size(); i++) fprintf(f, "%s%u", i ? "," : "", kv.second[i]);
                fprintf(f, "};\n");
            }
            fprintf(f, "static const struct { int logN, rank, cls, n; const uint16_t *ids; } FFT_PLACEMENT[] = {\n");
            idx = 0;
            for (auto &kv : g_orders) {
                fprintf(f, "    {%d, %d, %d, %zu, FFT_PLACE_%d},\n", std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first), kv.second.size(), idx);
                idx++;
            }
            fprintf(f, "};\n");
            fclose(f);
        }
        return 0;
    }
    fprintf(stderr, "usage: see the head of tools/exp/fft_swz_search.cpp\n");
    return 2;
}
