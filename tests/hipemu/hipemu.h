// TEST INFRASTRUCTURE -- a tiny single-process emulator of the HIP execution model.
//
// It lets the product's kernel sources (mp3-enc-bsd_amd/csrc/*.hip) be compiled by g++
// and executed lane by lane on the CPU so that kernel LOGIC can be checked against the
// oracle in the `-m "not gpu"` test run (this container has no GPU).  It is not a
// fallback: the product library never links it, bench.py and smoke() never load it, and
// libmp3mi.so fails loudly when no GPU is present.
//
// Model: one workgroup at a time; every thread of the workgroup is a ucontext fiber;
// __syncthreads() and the cross-lane primitives yield to the scheduler, which alternates
// the order in which it resumes lanes (forward / backward) so that a missing barrier shows
// up as a wrong answer in at least one direction.
#ifndef HIPEMU_H
#define HIPEMU_H

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint4 { unsigned x, y, z, w; };
struct __attribute__((aligned(8))) uint2 { unsigned x, y; };
static inline uint2 make_uint2(unsigned x, unsigned y) { uint2 r; r.x = x; r.y = y; return r; }
struct float2 { float x, y; };
struct __attribute__((aligned(16))) float4 { float x, y, z, w; };
static inline float2 make_float2(float x, float y) { float2 r; r.x = x; r.y = y; return r; }

typedef int hipError_t;
typedef void *hipStream_t;
typedef struct emu_event { double t; } *hipEvent_t;
enum { hipSuccess = 0, hipErrorNoDevice = 100, hipErrorInvalidValue = 1 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__

namespace hipemu {
struct ThreadCtx { dim3 tid, bid, bdim, gdim; };
extern ThreadCtx *cur;
void barrier();
unsigned long long ballot(int pred);
void exchange_put(const void *src, size_t n);           // publish this lane's value
void exchange_get(void *dst, size_t n, int src_lane);   // read another lane's value (after barrier)
void launch(dim3 grid, dim3 block, const std::function<void()> &body);
}

#define threadIdx (hipemu::cur->tid)
#define blockIdx (hipemu::cur->bid)
#define blockDim (hipemu::cur->bdim)
#define gridDim (hipemu::cur->gdim)

static inline void __syncthreads() { hipemu::barrier(); }
static inline unsigned long long __ballot(int p) { return hipemu::ballot(p); }

template <typename T> static inline T __shfl(T v, int src_lane, int width = 64)
{
    (void) width;
    T r;
    hipemu::exchange_put(&v, sizeof(T));
    hipemu::barrier();
    hipemu::exchange_get(&r, sizeof(T), src_lane & 63);
    hipemu::barrier();
    return r;
}
template <typename T> static inline T __shfl_xor(T v, int mask, int width = 64)
{
    return __shfl(v, (int) ((threadIdx.x & 63) ^ (unsigned) mask), width);
}
template <typename T> static inline T __shfl_down(T v, unsigned d, int width = 64)
{
    int l = (int) (threadIdx.x & 63) + (int) d;
    return __shfl(v, l > 63 ? (int) (threadIdx.x & 63) : l, width);
}
template <typename T> static inline T __shfl_up(T v, unsigned d, int width = 64)
{
    int l = (int) (threadIdx.x & 63) - (int) d;
    return __shfl(v, l < 0 ? (int) (threadIdx.x & 63) : l, width);
}
static inline void __threadfence_system() {}
static inline unsigned atomicAdd(unsigned *p, unsigned v) { const unsigned o = *p; *p = o + v; return o; } // fibers run one at a time
static inline unsigned long long atomicMax(unsigned long long *p, unsigned long long v) { const unsigned long long o = *p; if (v > o) *p = v; return o; }
#define __ATOMIC_RELEASE_EMU 0
#define __HIP_MEMORY_SCOPE_SYSTEM 0
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
#define __hip_atomic_load(p, order, scope) (*(p))
#define __HIP_MEMORY_SCOPE_AGENT 0
static inline int __builtin_amdgcn_readfirstlane(int v) { return __shfl(v, 0); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long) v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned) v) : 32; }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long) v) : 64; }

// ---- host runtime shims (memory is plain host memory) ----
static inline hipError_t hipMalloc(void **p, size_t n) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : 2; }
template <typename T> static inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void **) p, n); }
#define hipHostMallocMapped 2u
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { return hipMalloc(p, n); }
static inline hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned) { *d = h; return hipSuccess; }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = 0) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t = 0)
{
    for (size_t r = 0; r < h; r++) memcpy((char *) d + r * dp, (const char *) s + r * sp, w);
    return hipSuccess;
}
static inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = 0) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = 0; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = 0; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
enum { hipStreamDefault = 0, hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
static inline hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { *least = 0; *greatest = 0; return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = 0; return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline const char *hipGetErrorString(hipError_t) { return "hipemu"; }
double hipemu_now();
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new emu_event{0}; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new emu_event{0}; return hipSuccess; }
enum { hipDeviceAttributeCanUseStreamWaitValue = 1, hipMallocSignalMemory = 2, hipStreamWaitValueGte = 0, hipDeviceAttributeMultiprocessorCount = 3 };
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
struct hipDeviceProp_t { int multiProcessorCount; };
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { p->multiProcessorCount = 1; return hipSuccess; }
static inline hipError_t hipDeviceGetAttribute(int *v, int, int) { *v = 1; return hipSuccess; }
static inline hipError_t hipExtMallocWithFlags(void **p, size_t n, unsigned) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : 2; }
static inline hipError_t hipStreamWaitValue32(hipStream_t, void *, unsigned, unsigned, unsigned) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = 0) { e->t = hipemu_now(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float) ((b->t - a->t) * 1e3); return hipSuccess; }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipemu::launch((grid), (block), [&]() { kernel(__VA_ARGS__); })

#endif
