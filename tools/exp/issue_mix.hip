// Experiment: what bounds a SIMD with 4 resident wavefronts -- the vector pipe alone, or every issued
// instruction?  Loops of independent / dependent f32, f64 and scalar instructions.
// hipcc --offload-arch=gfx950 -O3 tools/exp/issue_mix.hip -o /tmp/issue_mix && /tmp/issue_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int MODE>
__global__ void __launch_bounds__(64) k(float *out, int iters)
{
    float a[16];
    double d[16];
    uint32_t s[8];
    for (int i = 0; i < 16; i++) { a[i] = (float) threadIdx.x + i; d[i] = (double) threadIdx.x + i; }
    for (int i = 0; i < 8; i++) s[i] = (uint32_t) iters + i;
    const float c = 1.0001f;
    const double cd = 1.0001;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (MODE == 0 || MODE == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (MODE == 2 || MODE == 3) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(cd));
            if (MODE == 4) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(cd));
            if (MODE == 5) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[0]) : "v"(c));  // one dependent chain
            if (MODE == 6) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[0]) : "v"(cd)); // one dependent chain
            if (MODE == 7 || MODE == 8) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (MODE == 9) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(cd));
            if (MODE == 1 || MODE == 3 || MODE == 8 || MODE == 10) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s[i & 7]) : : "scc");
            if (MODE == 11) { asm volatile("s_add_u32 %0, %0, 3" : "+s"(s[i & 7]) : : "scc"); asm volatile("s_add_u32 %0, %0, 5" : "+s"(s[(i + 4) & 7]) : : "scc"); }
        }
    }
    float r = 0;
    for (int i = 0; i < 16; i++) r += a[i] + (float) d[i];
    for (int i = 0; i < 8; i++) r += (float) s[i];
    out[blockIdx.x * 64 + threadIdx.x] = r;
}
template <int MODE> static float run(float *out, int blocks, int iters)
{
    hipEvent_t e0, e1;
    (void) hipEventCreate(&e0);
    (void) hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        (void) hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters);
        (void) hipEventRecord(e1);
        (void) hipEventSynchronize(e1);
        (void) hipEventElapsedTime(&ms, e0, e1);
    }
    return ms;
}
int main()
{
    float *out;
    (void) hipMalloc(&out, 8192 * 64 * sizeof(float));
    const int iters = 20000;
    const char *names[12] = {"16 v_add_f32", "16 v_add_f32 + 16 s_add", "16 v_add_f64", "16 v_add_f64 + 16 s_add", "16 v_fma_f64",
                             "16 dependent v_add_f32", "16 dependent v_add_f64", "16 v_add_u32", "16 v_add_u32 + 16 s_add", "16 v_mul_f64",
                             "16 s_add", "32 s_add"};
    run<0>(out, 4096, iters); // warm the clocks
    for (int wps = 8; wps <= 8; wps *= 2) { // full occupancy: the dispatcher packs smaller grids onto fewer CUs
        const int blocks = 1024 * wps;
        float ms[12] = {run<0>(out, blocks, iters), run<1>(out, blocks, iters), run<2>(out, blocks, iters), run<3>(out, blocks, iters),
                        run<4>(out, blocks, iters), run<5>(out, blocks, iters), run<6>(out, blocks, iters), run<7>(out, blocks, iters),
                        run<8>(out, blocks, iters), run<9>(out, blocks, iters), run<10>(out, blocks, iters), run<11>(out, blocks, iters)};
        for (int m = 0; m < 12; m++)
            printf("waves/SIMD %d  %-26s %8.3f ms  %6.1f cycles per iteration per SIMD (2.4 GHz)\n", wps, names[m], ms[m], ms[m] * 1e-3 * 2.4e9 / iters);
    }
    return 0;
}
