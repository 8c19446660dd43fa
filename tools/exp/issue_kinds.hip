// Experiment: issue cost per instruction KIND on a SIMD that holds four wavefronts (k_loop's regime), 32 instructions of
// one kind per loop iteration, independent over 8 registers unless the name says otherwise.  Does a 64-bit encoding
// (VOP3, DPP, SDWA, literal) cost more than a 32-bit one?
// hipcc --offload-arch=gfx950 -O3 tools/exp/issue_kinds.hip -o /tmp/issue_kinds && /tmp/issue_kinds
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int MODE>
__global__ void __launch_bounds__(64) k_kind(float *out, int iters)
{
    __shared__ float pad[2560]; // 10 KB: 16 workgroups per CU
    float a[8];
    uint32_t s[8];
    for (int i = 0; i < 8; i++) { a[i] = (float) threadIdx.x + i; s[i] = (uint32_t) iters + i; }
    for (int i = threadIdx.x; i < 2560; i += 64) pad[i] = (float) ((i * 7) & 63);
    __syncthreads();
    const float c = 1.0001f;
    unsigned long long m = 0x5555;
    asm volatile("s_mov_b64 %0, 0x5555" : "=s"(m));
    const unsigned ldsaddr = threadIdx.x * 4;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 32; i++) {
            float &x = a[i & 7];
            uint32_t &sx = s[i & 7];
            if (MODE == 0) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(x) : "v"(c));
            if (MODE == 1) asm volatile("v_add_f32_e64 %0, %0, %1" : "+v"(x) : "v"(c));
            if (MODE == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(c));
            if (MODE == 3) asm volatile("v_add_f32_e32 %0, 0x3f800347, %0" : "+v"(x));                 // literal: 8 bytes
            if (MODE == 4) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x) : "v"(c));
            if (MODE == 5) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x) : "v"(c), "s"(m));
            if (MODE == 6) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x) : "v"(c));
            if (MODE == 7) asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(x) : "v"(c));
            if (MODE == 8) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(x), "v"(c) : "vcc");
            if (MODE == 9) { unsigned long long mm; asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(mm) : "v"(x), "v"(c)); }
            if (MODE == 10) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sx) : : "scc");
            if (MODE == 11) asm volatile("s_add_u32 %0, %0, 0x12345" : "+s"(sx) : : "scc");              // literal: 8 bytes
            if (MODE == 12) asm volatile("s_mov_b32 %0, 7" : "=s"(sx));
            if (MODE == 13) asm volatile("s_addk_i32 %0, 5" : "+s"(sx) : : "scc");
            if (MODE == 14) { float t; asm volatile("ds_read_b32 %0, %1" : "=v"(t) : "v"(ldsaddr) : "memory"); }    // never waited for inside the block
            if (MODE == 15) asm volatile("ds_write_b16 %0, %1" : : "v"(ldsaddr), "v"(x) : "memory");
            if (MODE == 16) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(x) : "v"(c));
            if (MODE == 17) asm volatile("v_max_u32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "+v"(x));
            if (MODE == 18) asm volatile("v_add_f64 %0, %0, %1" : "+v"(*(double *) &a[(i & 3) * 2]) : "v"(1.0001));
            if (MODE == 19) asm volatile("v_mul_u32_u24_e32 %0, %0, %1" : "+v"(x) : "v"(c));
            if (MODE == 20) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(c));
            if (MODE == 21) asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(sx) : "v"(x));
            if (MODE == 22) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x));
            if (MODE == 23) asm volatile("v_add_f32_e32 %0, %0, %1\n s_nop 0" : "+v"(x) : "v"(c));
            if (MODE == 24) asm volatile("s_and_b64 %0, %0, exec" : "+s"(m) : : "scc");
            if (MODE == 25) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(x));
        }
        if (MODE == 14) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float r = pad[(threadIdx.x + 1) & 63] + (float) (m & 1);
    for (int i = 0; i < 8; i++) r += a[i] + (float) s[i];
    out[blockIdx.x * 64 + threadIdx.x] = r;
}

static double g_clk = 2.4e9;
template <int MODE> static void run(float *out, const char *name)
{
    const int iters = 20000, blocks = 4096;
    hipEvent_t e0, e1;
    (void) hipEventCreate(&e0);
    (void) hipEventCreate(&e1);
    float ms = 0, best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        (void) hipEventRecord(e0);
        hipLaunchKernelGGL(k_kind<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters);
        (void) hipEventRecord(e1);
        (void) hipEventSynchronize(e1);
        (void) hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    printf("%-44s %8.3f ms  %6.2f cycles per instruction and SIMD at 2.4 GHz, %5.2f x v_add_f32_e32\n", name, best,
           best * 1e-3 * g_clk / iters / 32 / 4, best / (float) g_clk);
}

int main()
{
    float *out;
    (void) hipMalloc(&out, 8192 * 64 * sizeof(float));
    hipLaunchKernelGGL(k_kind<0>, dim3(4096), dim3(64), 0, 0, out, 100000); // warm the clocks
    (void) hipDeviceSynchronize();
    run<0>(out, "v_add_f32_e32 (VOP2, 4 bytes)");
    run<1>(out, "v_add_f32_e64 (VOP3, 8 bytes)");
    run<2>(out, "v_fma_f32 (VOP3)");
    run<3>(out, "v_add_f32_e32 + literal (8 bytes)");
    run<4>(out, "v_cndmask_b32_e32 (vcc)");
    run<5>(out, "v_cndmask_b32_e64 (SGPR pair)");
    run<6>(out, "v_lshl_add_u32 (VOP3)");
    run<7>(out, "v_add_u32_e32");
    run<8>(out, "v_cmp_lt_f32_e32 (vcc)");
    run<9>(out, "v_cmp_lt_f32_e64 (SGPR pair)");
    run<10>(out, "s_add_u32 inline constant");
    run<11>(out, "s_add_u32 literal (8 bytes)");
    run<12>(out, "s_mov_b32");
    run<13>(out, "s_addk_i32");
    run<14>(out, "ds_read_b32 (32 in flight, one wait)");
    run<15>(out, "ds_write_b16");
    run<16>(out, "v_mov_b32_e32");
    run<17>(out, "v_max_u32_sdwa");
    run<18>(out, "v_add_f64");
    run<19>(out, "v_mul_u32_u24_e32");
    run<20>(out, "v_mad_u32_u24 (VOP3)");
    run<21>(out, "v_readlane_b32");
    run<22>(out, "v_add_f32_dpp");
    run<23>(out, "v_add_f32_e32 + s_nop 0 (per pair)");
    run<24>(out, "s_and_b64 with exec");
    run<25>(out, "v_bfe_u32 (VOP3)");
    return 0;
}
