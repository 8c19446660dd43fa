// Experiment: the cost model of a SIMD that holds FOUR wavefronts (k_loop's regime: 16 single-stream wavefronts per
// CU), for mixes of vector and scalar instructions, taken branches, v_readlane, VOPC into SGPRs, DPP and dependent
// LDS reads.  Every workgroup takes 10 KB of LDS so that exactly 16 are resident per CU.
// hipcc --offload-arch=gfx950 -O3 tools/exp/issue_model.hip -o /tmp/issue_model && /tmp/issue_model
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP16(x) REP8(x) REP8(x)
#define REP32(x) REP16(x) REP16(x)

// NV vector + NS scalar instructions per iteration, interleaved as evenly as the counts allow; DEP: one dependent chain
// per kind instead of 8 independent ones
template <int NV, int NS, bool DEP>
__global__ void __launch_bounds__(64) k_mix(float *out, int iters)
{
    __shared__ float pad[2560]; // 10 KB: 16 workgroups per CU
    float a[8];
    uint32_t s[8];
    for (int i = 0; i < 8; i++) { a[i] = (float) threadIdx.x + i; s[i] = (uint32_t) iters + i; }
    pad[threadIdx.x] = a[0];
    const float c = 1.0001f;
    for (int it = 0; it < iters; it++) {
        constexpr int N = NV > NS ? NV : NS;
#pragma unroll
        for (int i = 0; i < N; i++) {
            // spread the shorter kind evenly over the longer one
            const bool dov = NV >= NS ? true : ((i * NV) / N != ((i + 1) * NV) / N);
            const bool dos = NS >= NV ? true : ((i * NS) / N != ((i + 1) * NS) / N);
            if (dov) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[DEP ? 0 : (i & 7)]) : "v"(c));
            if (dos) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s[DEP ? 0 : (i & 7)]) : : "scc");
        }
    }
    float r = pad[(threadIdx.x + 1) & 63];
    for (int i = 0; i < 8; i++) r += a[i] + (float) s[i];
    out[blockIdx.x * 64 + threadIdx.x] = r;
}

// special instruction kinds, 32 per iteration
template <int MODE>
__global__ void __launch_bounds__(64) k_kind(float *out, int iters)
{
    __shared__ float pad[2560];
    float a[8];
    uint32_t s[8];
    int idx = threadIdx.x;
    for (int i = 0; i < 8; i++) { a[i] = (float) threadIdx.x + i; s[i] = (uint32_t) iters + i; }
    for (int i = threadIdx.x; i < 2560; i += 64) pad[i] = (float) ((i * 7) & 63);
    __syncthreads();
    const float c = 1.0001f;
    unsigned long long m = 0;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) { // 32 taken branches (each skips nothing)
            REP32(asm volatile("s_branch 0\n" ::: "memory");)
        }
        if (MODE == 1) { // 16 x (v_cmp into an SGPR pair, s_cmp on it, s_cselect)
#pragma unroll
            for (int i = 0; i < 16; i++) {
                unsigned long long mm;
                asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(mm) : "v"(a[i & 7]), "v"(c));
                asm volatile("s_cmp_lg_u64 %1, 0\n s_cselect_b32 %0, %0, 7" : "+s"(s[i & 7]) : "s"(mm) : "scc");
            }
        }
        if (MODE == 2) { // 16 x (v_readlane, dependent s_add)
#pragma unroll
            for (int i = 0; i < 16; i++) {
                uint32_t t;
                asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(t) : "v"(a[i & 7]));
                asm volatile("s_add_u32 %0, %0, %1" : "+s"(s[i & 7]) : "s"(t) : "scc");
            }
        }
        if (MODE == 3) { // 32 DPP adds, one dependent chain (a wave reduction's shape)
            REP32(asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[0]));)
        }
        if (MODE == 4) { // 32 DPP adds, four chains in lock-step
            REP8(asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                              : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));)
        }
        if (MODE == 5) { // 8 dependent LDS reads (index chain), nothing else
#pragma unroll
            for (int i = 0; i < 8; i++) idx = (int) pad[idx & 63] + (idx & 1);
        }
        if (MODE == 6) { // 8 dependent LDS reads + 24 independent vector adds
#pragma unroll
            for (int i = 0; i < 8; i++) {
                idx = (int) pad[idx & 63] + (idx & 1);
                asm volatile("v_add_f32 %0, %0, %3\n v_add_f32 %1, %1, %3\n v_add_f32 %2, %2, %3" : "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(c));
            }
        }
        if (MODE == 7) { // 16 x (s_cmp, not-taken s_cbranch)
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("s_cmp_eq_u32 %0, 12345\n s_cbranch_scc1 0" : : "s"(s[i & 7]) : "scc");
        }
        if (MODE == 8) { // 32 v_cndmask with an SGPR-pair mask
            asm volatile("s_mov_b64 %0, 0x5555" : "=s"(m));
            REP32(asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[1]) : "v"(c), "s"(m));)
        }
        if (MODE == 9) { // 32 s_waitcnt lgkmcnt(0) with nothing outstanding
            REP32(asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");)
        }
        if (MODE == 10) { // 32 s_nop 0
            REP32(asm volatile("s_nop 0");)
        }
    }
    float r = pad[(threadIdx.x + 1) & 63] + (float) idx + (float) (m & 1);
    for (int i = 0; i < 8; i++) r += a[i] + (float) s[i];
    out[blockIdx.x * 64 + threadIdx.x] = r;
}

template <typename F> static float timeit(F launch)
{
    hipEvent_t e0, e1;
    (void) hipEventCreate(&e0);
    (void) hipEventCreate(&e1);
    float ms = 0, best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        (void) hipEventRecord(e0);
        launch();
        (void) hipEventRecord(e1);
        (void) hipEventSynchronize(e1);
        (void) hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best;
}

#define MIX(NV, NS, DEP) do { \
    const float ms = timeit([&] { hipLaunchKernelGGL((k_mix<NV, NS, DEP>), dim3(blocks), dim3(64), 0, 0, out, iters); }); \
    printf("%2d VALU + %2d SALU %s  %8.3f ms  %7.1f cycles per iteration per SIMD  (%5.2f per instruction)\n", NV, NS, DEP ? "dependent  " : "independent", ms, \
           ms * 1e-3 * clk / iters, ms * 1e-3 * clk / iters / (NV + NS)); } while (0)
#define KIND(MODE, name, n) do { \
    const float ms = timeit([&] { hipLaunchKernelGGL((k_kind<MODE>), dim3(blocks), dim3(64), 0, 0, out, iters); }); \
    printf("%-58s %8.3f ms  %7.1f cycles per iteration per SIMD  (%5.2f per instruction of %d)\n", name, ms, ms * 1e-3 * clk / iters, ms * 1e-3 * clk / iters / n, n); } while (0)

int main()
{
    float *out;
    (void) hipMalloc(&out, 8192 * 64 * sizeof(float));
    const int iters = 20000, blocks = 4096; // 4 wavefronts per SIMD
    const double clk = 2.4e9;
    hipLaunchKernelGGL((k_mix<32, 0, false>), dim3(blocks), dim3(64), 0, 0, out, iters * 4); // warm the clocks
    (void) hipDeviceSynchronize();
    MIX(32, 0, false); MIX(0, 32, false); MIX(16, 16, false); MIX(24, 8, false); MIX(8, 24, false);
    MIX(32, 32, false); MIX(32, 16, false); MIX(16, 32, false); MIX(32, 8, false); MIX(32, 4, false);
    MIX(32, 0, true); MIX(0, 32, true); MIX(16, 16, true); MIX(32, 32, true); MIX(32, 16, true);
    KIND(0, "32 s_branch (taken)", 32);
    KIND(1, "16 x (v_cmp -> sgpr pair, s_cmp_lg_u64, s_cselect)", 48);
    KIND(2, "16 x (v_readlane, s_add)", 32);
    KIND(3, "32 x (s_nop 1, v_add_f32_dpp) one chain", 64);
    KIND(4, "32 v_add_f32_dpp, four chains in lock-step", 32);
    KIND(5, "8 dependent LDS reads (+ cvt, and, add each)", 8);
    KIND(6, "8 dependent LDS reads + 24 v_add_f32", 8);
    KIND(7, "16 x (s_cmp, s_cbranch not taken)", 32);
    KIND(8, "32 v_cndmask with SGPR mask", 32);
    KIND(9, "32 s_waitcnt lgkmcnt(0)", 32);
    KIND(10, "32 s_nop 0", 32);
    return 0;
}
