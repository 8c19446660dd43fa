// Experiment: issue cost of v_add_f32 vs v_pk_add_f32 / v_pk_mul_f32 with 1..4 wavefronts per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/exp/pk_rate.hip -o /tmp/pk_rate && /tmp/pk_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float v2f __attribute__((vector_size(8)));
template <int MODE>
__global__ void __launch_bounds__(64) k(float *out, int iters)
{
    float a[16];
    v2f p[8];
    for (int i = 0; i < 16; i++) a[i] = (float) threadIdx.x + i;
    for (int i = 0; i < 8; i++) p[i] = (v2f){a[2 * i], a[2 * i + 1]};
    const float c = 1.0001f;
    const v2f c2 = {c, c};
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        } else if (MODE == 4) { // 16 pk ops: same instruction count as mode 0
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
        }
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += a[i];
    for (int i = 0; i < 8; i++) s += p[i][0] + p[i][1];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
int main()
{
    float *out;
    (void) hipMalloc(&out, 4096 * 64 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    const char *names[5] = {"16 v_add_f32", "8 v_pk_add_f32", "8 v_pk_mul_f32", "16 v_mul_f32", "16 v_pk_add_f32"};
    for (int wps = 1; wps <= 4; wps *= 2)
        for (int mode = 0; mode < 5; mode++) {
            const int blocks = 1024 * wps; // 1024 SIMDs
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, out, iters); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, out, iters); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, out, iters); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64), 0, 0, out, iters); break;
                default: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(64), 0, 0, out, iters); break;
                }
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double cyc = ms * 1e-3 * 2.4e9 / iters; // cycles per loop iteration at 2.4 GHz (nominal)
            printf("waves/SIMD %d  %-16s %8.3f ms  %6.1f cycles per iteration per SIMD\n", wps, names[mode], ms, cyc);
        }
    return 0;
}
