// Generator aid (not part of the library): searches (a) the LDS swizzle of the FFT arrays -- which low address bits an
// element's index bits 4..9 are XORed into -- and (b) the placement of every rank's butterflies on rounds and lanes,
// for the fewest LDS cycles of the butterfly programs under the model of FftGen::round_cycles (an 8-byte store
// serves 16 lanes a cycle, an 8-byte load 32).  (b) is simulated annealing over swaps of butterflies of the same kind;
// its result is written as mp3-enc-bsd_amd/csrc/fft_placement.h, which tables_host.cpp replays.
//   g++ -O2 -std=c++17 -DMP3MI_EMU -DMP3MI_FFT_SWZ_RUNTIME -Imp3-enc-bsd_amd/csrc -Iinclude -Itests/hipemu \
//       tools/exp/fft_swz_search.cpp tests/hipemu/hipemu.cpp tests/hipemu/_build/tables_blob.o -o /tmp/fft_swz_search
//   (the blob object comes out of make -C tests/hipemu; the butterflies of the blocks of 256 points and more run in
//   registers and are no part of the programs the search sees)
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
    static uint32_t regtw[MP3MI_FFT_REG_ROWS_L * 256];
    static uint32_t leaf[256];
    const int nw = G->build(logN, nwin, hdr, 256, &nr, prog, 4 * maxw, rd, regtw, leaf);
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
static bool g_quiet = false;
static std::map<std::tuple<int, int, int>, std::vector<uint16_t>> g_orders;
static std::mt19937_64 g_rng(20261003);

// Simulated annealing over a whole program's schedule: swap two butterflies of the same class and kind -- within a
// list (lanes / rounds) or between two steps when every dependency still points from an earlier step to a later one.
static void anneal_hook(int logN, const std::vector<FftNode> &nodes, FftSchedule &sched)
{
    const size_t NN = nodes.size(), NL = sched.size();
    std::vector<int> list_of(NN, -1), at(NN, -1); // (-1: no part of the schedule -- the blocks of 8 points and fewer, FftGen::in_leaf)
    std::vector<size_t> base(NL + 1, 0); // first round of each list
    for (size_t li = 0; li < NL; li++) {
        base[li + 1] = base[li] + (sched[li].size() + 63) / 64;
        for (size_t i = 0; i < sched[li].size(); i++) { list_of[(size_t) sched[li][i]] = (int) li; at[(size_t) sched[li][i]] = (int) i; }
    }
    const size_t NR = base[NL];
    std::vector<unsigned> pos(NR * 8 * 64);
    std::vector<int> cyc(NR * 8, 0);
    auto nop = [&](size_t li) { return (li & 1) ? 8 : 4; };
    auto posof = [&](size_t li, size_t i, int k) -> unsigned {
        if (i >= sched[li].size()) return (unsigned) (MP3MI_FFT_DUMMY + (i & 63));
        const FusedOp &o = nodes[(size_t) sched[li][i]].o;
        return o.p[k] == MP3MI_FFT_DUMMY ? (unsigned) (MP3MI_FFT_DUMMY + (i & 63)) : o.p[k];
    };
    long cur = 0, ideal = 0;
    for (size_t li = 0; li < NL; li++)
        for (size_t r = base[li]; r < base[li + 1]; r++)
            for (int k = 0; k < nop(li); k++) {
                for (int l = 0; l < 64; l++) pos[(r * 8 + k) * 64 + l] = posof(li, (r - base[li]) * 64 + l, k);
                cur += cyc[r * 8 + k] = FftGen::round_cycles(&pos[(r * 8 + k) * 64]);
                ideal += 6;
            }
    const long start = cur;
    FftSchedule best = sched;
    long bestc = cur;
    // pools of interchangeable butterflies
    std::vector<int> pool[2][3];
    for (size_t i = 0; i < NN; i++) if (list_of[i] >= 0) pool[nodes[i].o.cls][nodes[i].o.kind].push_back((int) i);
    const long tries = g_sweeps * (long) NN;
    std::uniform_real_distribution<double> U(0.0, 1.0);
    long cross = 0;
    for (long t = 0; t < tries && bestc > ideal; t++) {
        const double temp = 0.8 * pow(0.02 / 0.8, (double) t / (double) tries);
        const int a = (int) (g_rng() % NN);
        if (list_of[(size_t) a] < 0) continue;
        const std::vector<int> &pl = pool[nodes[(size_t) a].o.cls][nodes[(size_t) a].o.kind];
        const int b = pl[g_rng() % pl.size()];
        if (a == b) continue;
        size_t la = (size_t) list_of[(size_t) a], lb = (size_t) list_of[(size_t) b];
        if (la != lb) { // between steps: every dependency must still point forward
            const int x = la < lb ? a : b, y = la < lb ? b : a; // x moves to the later step, y to the earlier one
            const int s_early = (int) (std::min(la, lb) / 2), s_late = (int) (std::max(la, lb) / 2);
            bool ok = true;
            for (size_t j = 0; ok && j < nodes[(size_t) x].succ.size(); j++) ok = list_of[(size_t) nodes[(size_t) x].succ[j]] < 0 || list_of[(size_t) nodes[(size_t) x].succ[j]] / 2 > s_late;
            for (size_t j = 0; ok && j < nodes[(size_t) y].pred.size(); j++) ok = list_of[(size_t) nodes[(size_t) y].pred[j]] / 2 < s_early;
            if (!ok) continue;
        }
        const size_t ia = (size_t) at[(size_t) a], ib = (size_t) at[(size_t) b];
        const size_t ra = base[la] + ia / 64, rb = base[lb] + ib / 64;
        const int n8 = nop(la);
        int before = 0, after = 0, ca[8], cb[8];
        for (int k = 0; k < n8; k++) { before += cyc[ra * 8 + k]; if (rb != ra) before += cyc[rb * 8 + k]; }
        auto do_swap = [&]() {
            std::swap(sched[la][ia], sched[lb][ib]);
            for (int k = 0; k < n8; k++) { pos[(ra * 8 + k) * 64 + (ia & 63)] = posof(la, ia, k); pos[(rb * 8 + k) * 64 + (ib & 63)] = posof(lb, ib, k); }
        };
        do_swap();
        for (int k = 0; k < n8; k++) {
            ca[k] = FftGen::round_cycles(&pos[(ra * 8 + k) * 64]);
            cb[k] = rb != ra ? FftGen::round_cycles(&pos[(rb * 8 + k) * 64]) : 0;
            after += ca[k] + cb[k];
        }
        const int d = after - before;
        if (d <= 0 || U(g_rng) < exp(-(double) d / temp)) {
            for (int k = 0; k < n8; k++) { cyc[ra * 8 + k] = ca[k]; if (rb != ra) cyc[rb * 8 + k] = cb[k]; }
            std::swap(list_of[(size_t) a], list_of[(size_t) b]);
            std::swap(at[(size_t) a], at[(size_t) b]);
            cur += d;
            if (la != lb) cross++;
            if (cur < bestc) { bestc = cur; best = sched; }
        } else do_swap();
    }
    sched = best;
    if (!g_quiet) printf("    2^%d points: %zu butterflies in %zu lists, %zu rounds: %ld -> %ld cycles (conflict-free %ld); %ld accepted moves between steps\n",
                         logN, NN, NL, NR, start, bestc, ideal, cross);
    for (size_t li = 0; li < NL; li++) {
        std::vector<uint16_t> ids;
        for (size_t i = 0; i < sched[li].size(); i++) ids.push_back((uint16_t) sched[li][i]);
        g_orders[std::make_tuple(logN, (int) (li / 2), (int) (li & 1))] = ids;
    }
}

int main(int argc, char **argv)
{
    G = new FftGen();
    for (int i = 4; i <= 10; i++) { G->tw_rs[i] = make_twiddle(i, false); G->tw_sr[i] = make_twiddle(i, true); }
    for (int b = 0; b < 10; b++) mp3mi_fft_swz_col[b] = b >= 5 ? 2u << ((b - 5) & 3) : 0;
    if (argc >= 3 && !strcmp(argv[1], "search")) {
        const double budget = atof(argv[2]);
        printf("a first even map: cost %d\n", total_cost(true));
        std::mt19937 rng(12345);
        int bestc = 1 << 30;
        const time_t t0 = time(NULL);
        while (difftime(time(NULL), t0) < budget) {
            for (int b = 4; b < 10; b++) mp3mi_fft_swz_col[b] = rng() & (b == 4 ? 14u : 30u); // (even: a pair of elements stays a pair -- fft_leaves reads 16 bytes at a time)
            int c = total_cost();
            for (bool improved = true; improved;) {
                improved = false;
                for (int b = 4; b < 10; b++)
                    for (int bit = 1; bit < (b == 4 ? 4 : 5); bit++) {
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
    if (argc >= 5 && !strcmp(argv[1], "search2")) { // hill climb on the swizzle with the ANNEALED placement as the yardstick: search2 <seconds> <sweeps> <seed> [c4..c9 to start from]
        const double budget = atof(argv[2]);
        g_sweeps = atol(argv[3]);
        std::mt19937 rng((unsigned) atoi(argv[4]));
        G->schedule_hook = anneal_hook;
        g_quiet = true;
        const time_t t0 = time(NULL);
        bool first = argc >= 11;
        int bestc = 1 << 30;
        while (difftime(time(NULL), t0) < budget) {
            if (first) { for (int b = 4; b < 10; b++) mp3mi_fft_swz_col[b] = (unsigned) atoi(argv[5 + b - 4]); first = false; }
            else for (int b = 4; b < 10; b++) mp3mi_fft_swz_col[b] = rng() & (b == 4 ? 14u : 30u); // (even: a pair of elements stays a pair -- fft_leaves reads 16 bytes at a time)
            g_rng.seed(1); int c = total_cost();
            for (bool improved = true; improved && difftime(time(NULL), t0) < budget;) {
                improved = false;
                for (int b = 4; b < 10; b++)
                    for (int bit = 1; bit < (b == 4 ? 4 : 5); bit++) {
                        mp3mi_fft_swz_col[b] ^= 1u << bit;
                        g_rng.seed(1); const int c2 = total_cost();
                        if (c2 < c) { c = c2; improved = true; }
                        else mp3mi_fft_swz_col[b] ^= 1u << bit;
                    }
            }
            if (c < bestc) {
                bestc = c;
                printf("annealed cost %d  cols[4..9] =", c);
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
        G->schedule_hook = anneal_hook;
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
