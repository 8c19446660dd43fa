// Device-side self-test entry: evaluates the dmath.h routines (and the hardware f64/f32
// sqrt/divide the kernels rely on) on the GPU so tests can prove they agree bit for bit with
// the host build of the same header.  Host buffers in, host buffers out.
#include "mp3mi_host.h"
#include "mp3mi.h"
#include "dmath.h"

__global__ void k_debug_dmath(int fn, const double *__restrict__ x, const double *__restrict__ y,
                              double *__restrict__ out, size_t n)
{
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (fn >= 20) { // wave-reduction self-tests: n must be a multiple of 64, every lane active
        const int v = (int) x[i];
        int r = 0;
        if (fn == 20) r = wave_sum_i32(v);
        else if (fn == 21) r = wave_max_i32(v);
        else if (fn == 22) r = wave_readlane_i32(v, (int) y[i & ~(size_t) 63]);
        out[i] = (double) r;
        return;
    }
    if (i >= n) return;
    double r = 0.0, s, c;
    switch (fn) {
    case 0: r = dm_log(x[i]); break;
    case 1: r = dm_exp(x[i]); break;
    case 2: r = dm_sin(x[i]); break;
    case 3: r = dm_cos(x[i]); break;
    case 4: r = dm_atan2(x[i], y[i]); break;
    case 5: r = __builtin_sqrt(x[i]); break;
    case 6: r = x[i] / y[i]; break;
    case 7: r = (double) __builtin_sqrtf((float) x[i]); break;
    case 8: r = (double) ((float) x[i] / (float) y[i]); break;
    case 9: dm_sincos(x[i], &s, &c); r = s; break;
    case 10: dm_sincos(x[i], &s, &c); r = c; break;
    case 11: r = x[i] * y[i] + 1.0; break; /* must NOT be contracted to fma */
    case 12: r = (double) ((float) x[i] * (float) y[i] + 1.0f); break;
    default: break;
    }
    out[i] = r;
}

extern "C" int mp3mi_debug_dmath(int fn, const double *x, const double *y, double *out, size_t n)
{
    double *dx = NULL, *dy = NULL, *dout = NULL;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return MP3MI_ERR_NO_DEVICE;
    int rc = MP3MI_ERR_HIP;
    if (hipMalloc((void **) &dx, n * 8) == hipSuccess && hipMalloc((void **) &dy, n * 8) == hipSuccess &&
        hipMalloc((void **) &dout, n * 8) == hipSuccess &&
        hipMemcpy(dx, x, n * 8, hipMemcpyHostToDevice) == hipSuccess &&
        hipMemcpy(dy, y ? y : x, n * 8, hipMemcpyHostToDevice) == hipSuccess) {
        hipLaunchKernelGGL(k_debug_dmath, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, 0, fn, dx, dy, dout, n);
        if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(out, dout, n * 8, hipMemcpyDeviceToHost) == hipSuccess)
            rc = MP3MI_OK;
    }
    if (dx) hipFree(dx);
    if (dy) hipFree(dy);
    if (dout) hipFree(dout);
    return rc;
}


// ---- error of the quantiser's first-tier arithmetic on this device ----
// what = 0: y34 = sqrt_raw(a * sqrt_raw(a)) against a^(3/4) for EVERY float a in [1, 4) (2^24 arguments: all
// mantissas at both exponent parities; scaling a by 4 scales every intermediate by a power of two, so other
// exponents repeat these errors exactly); what = 1: exp2_raw(-0.1875f * q) for the 801 step sizes
// q = MP3MI_STEP_MIN .. MP3MI_STEP_MIN + 800 the kernel can ask for; what = 2: exp2_raw on 2^24 evenly spaced
// arguments of [-80, 80].  Reference values in double (sqrt correctly rounded; 2^x = dm_exp(x ln 2), good to 1e-14).
// Each thread folds its relative errors into max_err[what] (non-negative doubles order like their bits).
#if defined(MP3MI_EMU)
extern "C" int mp3mi_debug_fastmath_bounds(double out[3]) { (void) out; return MP3MI_ERR_NO_DEVICE; } // a statement about the hardware
extern "C" int mp3mi_debug_pknorm_bound(double out[3]) { (void) out; return MP3MI_ERR_NO_DEVICE; }
#else
// ---- what v_cvt_pknorm_u16_f32 returns (the quantiser's rounding step, k_loop.hip: loop_quant_pair) ----
// For EVERY float a of [2^-31, 2048.5 / 65535] (218 M arguments) and the floats around both clamps: the result n against
// a * 65535 in double (exact: a 24-bit by a 16-bit factor).  out[0] = max |n - a * 65535| (0.5: a correctly rounded product; the
// quantiser's proof needs <= 0.5), out[1] = arguments whose successor converts to a SMALLER n (0: monotone), out[2] = arguments
// on which the two halves of the instruction or the clamps (a < 0 -> 0, a > 1 -> 65535) misbehave (0).
__global__ void __launch_bounds__(256) k_debug_pknorm(unsigned first_bits, unsigned count, unsigned long long *__restrict__ res)
{
    double w = 0.0;
    unsigned nm = 0, bad = 0;
    for (unsigned k = blockIdx.x * 256u + threadIdx.x; k < count; k += gridDim.x * 256u) {
        const float a = __builtin_bit_cast(float, first_bits + k), b = __builtin_bit_cast(float, first_bits + k + 1);
        const unsigned r = LOOP_PKNORM_U16(a, b), r2 = LOOP_PKNORM_U16(b, a);
        const double d = (double) (r & 0xffffu) - (double) a * 65535.0;
        w = __builtin_fabs(d) > w ? __builtin_fabs(d) : w;
        nm += (r >> 16) < (r & 0xffffu) ? 1u : 0u;
        bad += (r2 >> 16) != (r & 0xffffu) || (r2 & 0xffffu) != (r >> 16) ? 1u : 0u; // both halves convert alike
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        bad += LOOP_PKNORM_U16(-1.0f, 2.0f) != 0xffff0000u ? 1u : 0u;
        bad += LOOP_PKNORM_U16(-0.0f, 1.0f) != 0xffff0000u ? 1u : 0u;
        bad += LOOP_PKNORM_U16(-1e-30f, 1.0000001f) != 0xffff0000u ? 1u : 0u;
    }
    atomicMax(&res[0], __builtin_bit_cast(unsigned long long, w));
    if (nm) atomicAdd(&res[1], (unsigned long long) nm);
    if (bad) atomicAdd(&res[2], (unsigned long long) bad);
}
extern "C" int mp3mi_debug_pknorm_bound(double out[3])
{
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return MP3MI_ERR_NO_DEVICE;
    unsigned long long *d = NULL, h[3];
    int rc = MP3MI_ERR_HIP;
    if (hipMalloc((void **) &d, sizeof(h)) == hipSuccess && hipMemset(d, 0, sizeof(h)) == hipSuccess) {
        const unsigned b0 = 0x30000000u /* 2^-31 */, b1 = __builtin_bit_cast(unsigned, 2048.5f / 65535.0f);
        hipLaunchKernelGGL(k_debug_pknorm, dim3(8192), dim3(256), 0, 0, b0, b1 - b0 + 1u, d);
        if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
            out[0] = __builtin_bit_cast(double, h[0]);
            out[1] = (double) h[1];
            out[2] = (double) h[2];
            rc = MP3MI_OK;
        }
    }
    if (d) hipFree(d);
    return rc;
}

__global__ void k_debug_fastmath(int what, unsigned long long *__restrict__ max_err)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    double err = 0.0;
    if (what == 0) {
        if (i >= (1u << 24)) return;
        const float a = __builtin_bit_cast(float, 0x3f800000u + i); // [1, 4)
        const float y = LOOP_FAST_SQRTF(a * LOOP_FAST_SQRTF(a));
        const double ad = (double) a, ref = __builtin_sqrt(ad * __builtin_sqrt(ad));
        err = __builtin_fabs((double) y - ref) / ref;
    } else {
        float x;
        if (what == 1) {
            if (i >= (unsigned) MP3MI_STEP_N) return;
            x = -0.1875f * (float) (MP3MI_STEP_MIN + (int) i);
        } else {
            if (i >= (1u << 24)) return;
            x = -80.0f + (float) i * (160.0f / 16777216.0f);
        }
        const float y = LOOP_FAST_EXP2F(x);
        const double ref = dm_exp((double) x * 0.6931471805599453);
        err = __builtin_fabs((double) y - ref) / ref;
    }
    atomicMax(&max_err[what], __builtin_bit_cast(unsigned long long, err));
}

extern "C" int mp3mi_debug_fastmath_bounds(double out[3])
{
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return MP3MI_ERR_NO_DEVICE;
    unsigned long long *d = NULL;
    int rc = MP3MI_ERR_HIP;
    if (hipMalloc((void **) &d, 3 * sizeof(unsigned long long)) == hipSuccess && hipMemset(d, 0, 3 * sizeof(unsigned long long)) == hipSuccess) {
        hipLaunchKernelGGL(k_debug_fastmath, dim3(1u << 16), dim3(256), 0, 0, 0, d);
        hipLaunchKernelGGL(k_debug_fastmath, dim3((MP3MI_STEP_N + 255) / 256), dim3(256), 0, 0, 1, d);
        hipLaunchKernelGGL(k_debug_fastmath, dim3(1u << 16), dim3(256), 0, 0, 2, d);
        if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(out, d, 3 * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess) rc = MP3MI_OK;
    }
    if (d) hipFree(d);
    return rc;
}
#endif
