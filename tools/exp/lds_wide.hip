// Experiment (round 6, F10): what do wide LDS reads of a wavefront cost, and does SQ_LDS_BANK_CONFLICT count their passes?  k_fft's butterfly
// program reads its per-lane records as 16-byte words (lane-consecutive) and its operands as 8-byte elements at searched positions;
// the counter attributes 25 % conflict cycles to the program (profiles/r06_experiments.txt, F8).  One kernel per pattern, 4 waves per SIMD:
//   0 ds_read_b32 lane-consecutive   1 ds_read_b64 lane-consecutive   2 ds_read_b128 lane-consecutive   3 four ds_read_b32 (structure of arrays)
//   4 ds_read_b64 stride 16 B (every other element)   5 ds_read_b128 stride 32 B   6 ds_write_b64 consecutive   7 ds_read2_b64 (two consecutive elements)
//   16-byte accesses at per-lane addresses (fft_leaves): how many lanes does the LDS serve a cycle, over how many banks?
//   8 ds_write_b128 consecutive   9 ds_write_b128, the 16 lanes of a group on 16 different groups of four banks but lanes 2 i, 2 i + 1 eight groups apart
//   (collide if a write is served 8 lanes a cycle over 32 banks)   10 ds_read_b128, the same addresses
//   11 ds_write_b128, lanes i and i + 8 on the same group of four banks, another row (collide if 16 lanes a cycle)   12 ds_read_b128, the same addresses
// hipcc --offload-arch=gfx950 -O3 tools/exp/lds_wide.hip -o /tmp/lds_wide && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d /tmp/lw -- /tmp/lds_wide
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int MODE> __global__ void __launch_bounds__(256) k(float *out, int iters)
{
    __shared__ float x[4][4096];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = lane; i < 4096; i += 64) x[wv][i] = (float) i;
    __syncthreads();
    const unsigned base = (unsigned) (size_t) &x[wv][0];
    unsigned addr = base;
    if (MODE == 0 || MODE == 3) addr += 4u * lane;
    if (MODE == 1 || MODE == 6 || MODE == 7) addr += 8u * lane;
    if (MODE == 2) addr += 16u * lane;
    if (MODE == 4) addr += 16u * lane;
    if (MODE == 5) addr += 32u * lane;
    if (MODE == 8) addr += 16u * lane;
    if (MODE == 9 || MODE == 10) { const unsigned i = lane & 15u; addr += 16u * (16u * (lane >> 4) + ((i >> 1) | ((i & 1u) << 3))); }
    if (MODE == 11 || MODE == 12) { const unsigned i = lane & 15u; addr += 16u * (i & 7u) + 256u * (i >> 3) + 512u * (lane >> 4); }
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if (MODE == 0) { float v; asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory"); a0 += v; }
            if (MODE == 1 || MODE == 4) { double v; asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory"); a0 += (float) v; }
            if (MODE == 8 || MODE == 9 || MODE == 11) { typedef float f4_ __attribute__((ext_vector_type(4))); f4_ v = {a0, a0, a0, a0}; asm volatile("ds_write_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : : "v"(addr), "v"(v) : "memory"); a0 += 1.0f; }
            if (MODE == 2 || MODE == 5 || MODE == 10 || MODE == 12) { float4 v; asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory"); a0 += v.x + v.w; }
            if (MODE == 3) { float v0, v1, v2, v3; asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:256\n ds_read_b32 %2, %4 offset:512\n ds_read_b32 %3, %4 offset:768\n s_waitcnt lgkmcnt(0)" : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(addr) : "memory"); a0 += v0 + v1 + v2 + v3; }
            if (MODE == 6) { double v = a0; asm volatile("ds_write_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : : "v"(addr), "v"(v) : "memory"); a0 += 1.0f; }
            if (MODE == 7) { float4 v; asm volatile("ds_read2_b64 %0, %1 offset1:1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory"); a0 += v.x + v.w; }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;
}

template <int MODE> static void run(float *out, const char *name, int bytes_per_lane)
{
    const int iters = 20000, blocks = 1024;
    hipEvent_t e0, e1;
    (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    float ms = 0, best = 1e9f;
    for (int rep = 0; rep < 2; rep++) {
        (void) hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
        (void) hipEventRecord(e1); (void) hipEventSynchronize(e1); (void) hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    printf("mode %d %-34s %8.3f ms   %6.1f bytes per CU and ns\n", MODE, name, best, (double) blocks * 4 * 64 * bytes_per_lane * 8.0 * iters / (best * 1e6) / 256.0);
    fflush(stdout);
}

int main()
{
    float *out;
    (void) hipMalloc(&out, 1024 * 256 * sizeof(float));
    run<0>(out, "ds_read_b32 consecutive", 4);
    run<1>(out, "ds_read_b64 consecutive", 8);
    run<2>(out, "ds_read_b128 consecutive", 16);
    run<3>(out, "4 x ds_read_b32 (SoA)", 16);
    run<4>(out, "ds_read_b64 stride 16 B", 8);
    run<5>(out, "ds_read_b128 stride 32 B", 16);
    run<6>(out, "ds_write_b64 consecutive", 8);
    run<7>(out, "ds_read2_b64 consecutive pairs", 16);
    run<8>(out, "ds_write_b128 consecutive", 16);
    run<9>(out, "ds_write_b128 2i, 2i+1 8 quads apart", 16);
    run<10>(out, "ds_read_b128 2i, 2i+1 8 quads apart", 16);
    run<11>(out, "ds_write_b128 i, i+8 same quad", 16);
    run<12>(out, "ds_read_b128 i, i+8 same quad", 16);
    return 0;
}
