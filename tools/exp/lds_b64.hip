// Experiment: what an 8-byte LDS access of 64 lanes costs for different bank patterns -- the model behind the FFT
// placement (FftGen::round_cycles: loads 32 lanes a cycle over 64 banks, stores 16 lanes a cycle over 32 banks).
// One wavefront per SIMD x 4 waves; e = element (8-byte) index per lane.
// hipcc --offload-arch=gfx950 -O3 tools/exp/lds_b64.hip -o /tmp/lds_b64 && /tmp/lds_b64
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__device__ int pattern(int mode, int lane)
{
    switch (mode) {
    case 0: return lane;                                   // all different: conflict-free in every model
    case 1: return (lane & 15) + 32 * (lane >> 4);         // the four 16-lane groups repeat the same 16 bank pairs (mod 32 and mod 16 alike)
    case 2: return (lane & 15) + 16 * ((lane >> 4) & 1) + 64 * (lane >> 5); // each 32-lane half covers 32 different pairs
    case 3: return (lane & 7) + 32 * (lane >> 3);          // 8 pairs only: 2-way within 16 lanes, 4-way within 32
    case 4: return (lane & 31) * 2 + 64 * (lane >> 5);     // even elements only: 16 different pairs mod 16 per 16 lanes? (2e mod 16: 8 values) 
    case 5: return 32 * lane;                              // all on one bank pair: 64-way
    case 6: return (lane & 15) * 2 + (lane >> 4) * 64;     // 2e: 8 distinct mod 16 per 16 lanes (2-way), 16 distinct mod 32 per 32 lanes... 
    default: return lane;
    }
}

template <bool WRITE> __global__ void __launch_bounds__(256) k(double *out, int mode, int iters)
{
    __shared__ double x[4][2200];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = lane; i < 2200; i += 64) x[wv][i] = i;
    __syncthreads();
    const int e = pattern(mode, lane) % 2176;
    const unsigned addr = (unsigned) (size_t) &x[wv][e]; // LDS byte address
    double a[8] = {0, 0, 0, 0, 0, 0, 0, 0}, v = lane;
    for (int it = 0; it < iters; it++) {
        if (WRITE) {
            asm volatile("ds_write_b64 %0, %1\n ds_write_b64 %0, %1\n ds_write_b64 %0, %1\n ds_write_b64 %0, %1\n"
                         "ds_write_b64 %0, %1\n ds_write_b64 %0, %1\n ds_write_b64 %0, %1\n ds_write_b64 %0, %1\n"
                         "ds_write_b64 %0, %1\n ds_write_b64 %0, %1\n ds_write_b64 %0, %1\n ds_write_b64 %0, %1\n"
                         "ds_write_b64 %0, %1\n ds_write_b64 %0, %1\n ds_write_b64 %0, %1\n ds_write_b64 %0, %1\n s_waitcnt lgkmcnt(0)"
                         : : "v"(addr), "v"(v) : "memory");
        } else {
            asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8\n ds_read_b64 %2, %8\n ds_read_b64 %3, %8\n"
                         "ds_read_b64 %4, %8\n ds_read_b64 %5, %8\n ds_read_b64 %6, %8\n ds_read_b64 %7, %8\n"
                         "ds_read_b64 %0, %8\n ds_read_b64 %1, %8\n ds_read_b64 %2, %8\n ds_read_b64 %3, %8\n"
                         "ds_read_b64 %4, %8\n ds_read_b64 %5, %8\n ds_read_b64 %6, %8\n ds_read_b64 %7, %8\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(a[4]), "=&v"(a[5]), "=&v"(a[6]), "=&v"(a[7]) : "v"(addr) : "memory");
        }
    }
    double acc = v;
    for (int i = 0; i < 8; i++) acc += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc + x[wv][(e + 1) % 2176];
}

int main()
{
    double *out;
    (void) hipMalloc(&out, 1024 * 256 * sizeof(double));
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    for (int wr = 0; wr < 2; wr++)
        for (int mode = 0; mode < 7; mode++) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; rep++) {
                (void) hipEventRecord(e0);
                if (wr) hipLaunchKernelGGL(k<true>, dim3(1024), dim3(256), 0, 0, out, mode, iters);
                else hipLaunchKernelGGL(k<false>, dim3(1024), dim3(256), 0, 0, out, mode, iters);
                (void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
                float ms; (void) hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            // four workgroups of 4 wavefronts per CU: 16 waves share the CU's LDS pipe
            printf("%s pattern %d: %7.3f ms  %6.2f LDS-pipe cycles per wave-instruction (2.4 GHz nominal)\n", wr ? "store" : "load ", mode, best,
                   best * 1e-3 * 2.4e9 / (iters * 16.0 * 16));
        }
    return 0;
}
