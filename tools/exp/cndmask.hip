// Experiment: why did 32 x v_cndmask_b32_e32 (vcc) cost 22.8 cycles each in tools/exp/issue_kinds.hip?  Variants.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int MODE>
__global__ void __launch_bounds__(64) k(float *out, int iters)
{
    __shared__ float pad[2560];
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = (float) threadIdx.x + i;
    pad[threadIdx.x] = a[0];
    const float c = 1.0001f;
    unsigned long long m;
    asm volatile("s_mov_b64 %0, 0x5555" : "=s"(m));
    if (MODE == 1 || MODE == 5) asm volatile("s_mov_b64 vcc, 0x5555" ::: "vcc");
    for (int it = 0; it < iters; it++) {
        if (MODE == 2) asm volatile("s_mov_b64 vcc, 0x5555" ::: "vcc");
#pragma unroll
        for (int i = 0; i < 32; i++) {
            float &x = a[i & 7];
            if (MODE == 0 || MODE == 1 || MODE == 2) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x) : "v"(c));
            if (MODE == 3) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1\n v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x) : "v"(c) : "vcc");
            if (MODE == 4) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(x) : "v"(c));
            if (MODE == 5) asm volatile("v_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(x) : "v"(a[(i + 1) & 7]), "v"(c));
            if (MODE == 6) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1\n s_nop 0\n v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x) : "v"(c) : "vcc");
            if (MODE == 7) asm volatile("v_cmp_lt_f32_e32 vcc, %1, %2\n v_add_f32_e32 %0, %0, %2" : "+v"(x) : "v"(a[(i + 1) & 7]), "v"(c) : "vcc");
            if (MODE == 8) asm volatile("v_cmp_lt_f32_e64 %3, %0, %1\n v_cndmask_b32_e64 %0, %0, %1, %3" : "+v"(x) : "v"(c), "v"(c), "s"(m));
        }
    }
    float r = pad[(threadIdx.x + 1) & 63] + (float) (m & 1);
    for (int i = 0; i < 8; i++) r += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = r;
}
template <int MODE> static void run(float *out, const char *name, int per)
{
    const int iters = 20000, blocks = 4096;
    hipEvent_t e0, e1;
    (void) hipEventCreate(&e0);
    (void) hipEventCreate(&e1);
    float ms = 0, best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        (void) hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters);
        (void) hipEventRecord(e1);
        (void) hipEventSynchronize(e1);
        (void) hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    printf("%-60s %8.3f ms  %6.2f cycles per instruction and SIMD at 2.4 GHz\n", name, best, best * 1e-3 * 2.4e9 / iters / per / 4);
}
int main()
{
    float *out;
    (void) hipMalloc(&out, 8192 * 64 * sizeof(float));
    hipLaunchKernelGGL(k<7>, dim3(4096), dim3(64), 0, 0, out, 100000);
    (void) hipDeviceSynchronize();
    run<0>(out, "32 v_cndmask_e32 vcc (vcc never written)", 32);
    run<1>(out, "32 v_cndmask_e32 vcc (s_mov vcc before the loop)", 32);
    run<2>(out, "32 v_cndmask_e32 vcc (s_mov vcc every iteration)", 32);
    run<3>(out, "32 x (v_cmp_e32 vcc, v_cndmask_e32 vcc)", 64);
    run<4>(out, "32 v_cndmask_e64 with vcc as the SGPR pair", 32);
    run<5>(out, "32 v_cndmask_e32 vcc, no read-modify-write", 32);
    run<6>(out, "32 x (v_cmp_e32 vcc, s_nop 0, v_cndmask_e32 vcc)", 96);
    run<7>(out, "32 x (v_cmp_e32 vcc, v_add_f32)", 64);
    run<8>(out, "32 x (v_cmp_e64 sgpr, v_cndmask_e64 sgpr)", 64);
    return 0;
}
