// Experiment: issue cost (cycles per instruction per SIMD) of the f64 conversions and of f64 multiplies with an SGPR
// operand, at 4 wavefronts per SIMD.  hipcc --offload-arch=gfx950 -O3 tools/exp/cvt_rate.hip -o /tmp/cvt_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP32(x) REP16(x) REP16(x)

template <int MODE> __global__ void __launch_bounds__(64) k_kind(double *out, int iters, double sc)
{
    __shared__ float pad[2560]; // 10 KB: 16 workgroups per CU
    double a[4];
    int x[4];
    float f[4];
    for (int i = 0; i < 4; i++) { a[i] = threadIdx.x + i; x[i] = threadIdx.x * 3 + i; f[i] = threadIdx.x + 0.5f * i; }
    pad[threadIdx.x] = f[0];
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) { REP32(asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a[0]) : "v"(x[0])); asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a[1]) : "v"(x[1]));) }
        if (MODE == 1) { REP32(asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[0]) : "v"(f[0])); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[1]) : "v"(f[1]));) }
        if (MODE == 2) { REP32(asm volatile("v_mul_f64 %0, %1, %2" : "=v"(a[0]) : "v"(a[2]), "v"(a[3])); asm volatile("v_mul_f64 %0, %1, %2" : "=v"(a[1]) : "v"(a[3]), "v"(a[2]));) }
        if (MODE == 3) { REP32(asm volatile("v_mul_f64 %0, %1, %2" : "=v"(a[0]) : "s"(sc), "v"(a[3])); asm volatile("v_mul_f64 %0, %1, %2" : "=v"(a[1]) : "s"(sc), "v"(a[2]));) }
        if (MODE == 4) { REP32(asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[0]) : "v"(a[3])); asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[1]) : "v"(a[2]));) } // two dependent chains
        if (MODE == 5) { REP32(asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[0]) : "v"(a[3])); asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[0]) : "v"(a[2]));) } // one dependent chain
        if (MODE == 6) { REP32(asm volatile("v_mul_f64 %0, %2, %3\n v_add_f64 %1, %1, %0" : "=&v"(a[0]), "+v"(a[1]) : "s"(sc), "v"(a[2]));asm volatile("v_mul_f64 %0, %2, %3\n v_add_f64 %1, %1, %0" : "=&v"(a[0]), "+v"(a[1]) : "s"(sc), "v"(a[3]));) } // the matrixing chain
        if (MODE == 7) { REP32(asm volatile("v_ashrrev_i32 %0, 16, %1" : "=v"(x[2]) : "v"(x[0])); asm volatile("v_bfe_i32 %0, %1, 0, 16" : "=v"(x[3]) : "v"(x[1]));) }
    }
    double r = pad[(threadIdx.x + 1) & 63];
    for (int i = 0; i < 4; i++) r += a[i] + x[i] + f[i];
    out[blockIdx.x * 64 + threadIdx.x] = r;
}

template <typename F> static float timeit(F launch)
{
    hipEvent_t e0, e1;
    (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    float ms = 0, best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        (void) hipEventRecord(e0); launch(); (void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
        (void) hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best;
}
#define KIND(MODE, name) do { \
    const float ms = timeit([&] { hipLaunchKernelGGL((k_kind<MODE>), dim3(blocks), dim3(64), 0, 0, out, iters, 1.000001); }); \
    printf("%-44s %8.3f ms  %6.2f cycles per instruction per SIMD\n", name, ms, ms * 1e-3 * clk / iters / 64); } while (0)
int main()
{
    double *out;
    (void) hipMalloc(&out, 8192 * 64 * sizeof(double));
    const int iters = 10000, blocks = 4096; // 4 wavefronts per SIMD
    const double clk = 2.4e9;
    hipLaunchKernelGGL((k_kind<2>), dim3(blocks), dim3(64), 0, 0, out, iters * 4, 1.0);
    (void) hipDeviceSynchronize();
    KIND(0, "v_cvt_f64_i32"); KIND(1, "v_cvt_f64_f32"); KIND(2, "v_mul_f64 v, v"); KIND(3, "v_mul_f64 s, v");
    KIND(4, "v_add_f64, two chains"); KIND(5, "v_add_f64, one chain"); KIND(6, "v_mul_f64 s,v + dependent v_add_f64 chain"); KIND(7, "v_ashrrev_i32 / v_bfe_i32");
    return 0;
}
