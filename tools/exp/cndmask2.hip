// Experiment: v_cndmask_b32_e32 reading vcc is ~23 cycles per instruction when nothing wrote vcc just before it
// (tools/exp/cndmask.hip).  Which patterns around one v_cmp are fast?
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
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            float &x = a[i & 7], &y = a[(i + 1) & 7], &z = a[(i + 2) & 7], &w = a[(i + 3) & 7];
            if (MODE == 0) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %2\n v_cndmask_b32_e32 %0, %0, %2, vcc\n v_cndmask_b32_e32 %1, %1, %2, vcc" : "+v"(x), "+v"(y) : "v"(c) : "vcc");
            if (MODE == 1) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %4\n v_cndmask_b32_e32 %0, %0, %4, vcc\n v_cndmask_b32_e32 %1, %1, %4, vcc\n v_cndmask_b32_e32 %2, %2, %4, vcc\n v_cndmask_b32_e32 %3, %3, %4, vcc" : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(c) : "vcc");
            if (MODE == 2) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %2\n v_add_f32_e32 %1, %1, %2\n v_cndmask_b32_e32 %0, %0, %2, vcc" : "+v"(x), "+v"(y) : "v"(c) : "vcc");
            if (MODE == 3) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %2\n v_cndmask_b32_e32 %0, %0, %2, vcc\n v_add_f32_e32 %1, %1, %2\n v_cndmask_b32_e32 %1, %1, %2, vcc" : "+v"(x), "+v"(y) : "v"(c) : "vcc");
            if (MODE == 4) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %2\n v_add_f32_e32 %1, %1, %2\n v_add_f32_e32 %1, %1, %2\n v_add_f32_e32 %1, %1, %2\n v_add_f32_e32 %1, %1, %2\n v_cndmask_b32_e32 %0, %0, %2, vcc" : "+v"(x), "+v"(y) : "v"(c) : "vcc");
            if (MODE == 5) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %2\n s_nop 4\n v_cndmask_b32_e32 %0, %0, %2, vcc\n s_nop 4\n v_cndmask_b32_e32 %1, %1, %2, vcc" : "+v"(x), "+v"(y) : "v"(c) : "vcc");
            if (MODE == 6) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %2\n v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n v_cndmask_b32_e32 %0, %0, %2, vcc" : "+v"(x), "+v"(y) : "v"(c) : "vcc");
            if (MODE == 7) asm volatile("v_cmp_lt_f32_e64 %3, %0, %2\n v_cndmask_b32_e64 %0, %0, %2, %3\n v_cndmask_b32_e64 %1, %1, %2, %3" : "+v"(x), "+v"(y) : "v"(c), "s"(m));
            if (MODE == 8) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %2\n s_and_b64 vcc, vcc, exec\n v_cndmask_b32_e32 %0, %0, %2, vcc" : "+v"(x), "+v"(y) : "v"(c) : "vcc", "scc");
            if (MODE == 9) asm volatile("s_mov_b64 vcc, %3\n v_cndmask_b32_e32 %0, %0, %2, vcc" : "+v"(x), "+v"(y) : "v"(c), "s"(m) : "vcc");
            if (MODE == 10) asm volatile("s_mov_b64 vcc, %3\n s_nop 3\n v_cndmask_b32_e32 %0, %0, %2, vcc" : "+v"(x), "+v"(y) : "v"(c), "s"(m) : "vcc");
        }
    }
    float r = pad[(threadIdx.x + 1) & 63] + (float) (m & 1);
    for (int i = 0; i < 8; i++) r += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = r;
}
template <int MODE> static void run(float *out, const char *name)
{
    const int iters = 40000, blocks = 4096;
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
    printf("%-64s %8.3f ms  %6.2f cycles per GROUP and SIMD at 2.4 GHz\n", name, best, best * 1e-3 * 2.4e9 / iters / 8 / 4);
}
int main()
{
    float *out;
    (void) hipMalloc(&out, 8192 * 64 * sizeof(float));
    hipLaunchKernelGGL(k<2>, dim3(4096), dim3(64), 0, 0, out, 400000);
    (void) hipDeviceSynchronize();
    run<0>(out, "v_cmp vcc, cndmask, cndmask");
    run<1>(out, "v_cmp vcc, 4 x cndmask");
    run<2>(out, "v_cmp vcc, v_add, cndmask");
    run<3>(out, "v_cmp vcc, cndmask, v_add, cndmask");
    run<4>(out, "v_cmp vcc, 4 x v_add, cndmask");
    run<5>(out, "v_cmp vcc, s_nop 4, cndmask, s_nop 4, cndmask");
    run<6>(out, "v_cmp vcc, v_addc_co (reads+writes vcc), cndmask");
    run<7>(out, "v_cmp_e64 sgpr, 2 x cndmask_e64 sgpr");
    run<8>(out, "v_cmp vcc, s_and_b64 vcc vcc exec, cndmask");
    run<9>(out, "s_mov_b64 vcc, cndmask");
    run<10>(out, "s_mov_b64 vcc, s_nop 3, cndmask");
    return 0;
}
