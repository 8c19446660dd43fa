// Experiment: latency from a device-side store to a hipStreamWaitValue32 waiter on another stream.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_long(unsigned *sig, unsigned val, int mode, unsigned long long *t_store, unsigned long long ticks)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *t_store = t0;
        if (mode == 0) __hip_atomic_store(sig, val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        else if (mode == 1) { *(volatile unsigned *) sig = val; __threadfence_system(); }
        else if (mode == 2) { __hip_atomic_exchange(sig, val, __ATOMIC_SEQ_CST, __HIP_MEMORY_SCOPE_SYSTEM); }
        else if (mode == 3) { __builtin_nontemporal_store(val, sig); __threadfence_system(); }
    }
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(100);
}
__global__ void k_after(unsigned long long *t) { if (threadIdx.x == 0) *t = __builtin_amdgcn_s_memrealtime(); }
int main()
{
    hipStream_t a, b;
    CK(hipStreamCreate(&a)); CK(hipStreamCreate(&b));
    unsigned long long *t; CK(hipHostMalloc((void **) &t, 64, 0));
    for (int kind = 0; kind < 2; kind++) {
        unsigned *sig = NULL;
        if (kind == 0) CK(hipExtMallocWithFlags((void **) &sig, 8, hipMallocSignalMemory));
        else CK(hipHostMalloc((void **) &sig, 8, hipHostMallocCoherent | hipHostMallocMapped));
        CK(hipMemset(sig, 0, 8));
        unsigned gen = 0;
        for (int mode = 0; mode < 4; mode++) {
            for (int rep = 0; rep < 2; rep++) {
                gen++;
                hipError_t e = hipStreamWaitValue32(b, sig, gen, hipStreamWaitValueGte, 0xffffffffu);
                if (e != hipSuccess) { printf("kind %d: WaitValue32 -> %s\n", kind, hipGetErrorString(e)); (void)hipGetLastError(); break; }
                hipLaunchKernelGGL(k_after, dim3(1), dim3(64), 0, b, t + 1);
                hipLaunchKernelGGL(k_long, dim3(256), dim3(64), 0, a, sig, gen, mode, t, 3000000ull /* 30 ms */);
                CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
                printf("kind %d (%s) mode %d: waiter started %.3f ms after the store\n", kind, kind ? "host coherent" : "signal memory", mode, (double) (t[1] - t[0]) / 1e5);
            }
        }
    }
    return 0;
}
