// CPU build of the product's deterministic math layer for unit tests (no HIP).
// Exposes each function over plain arrays so python can compare against glibc and mpmath.
#include <cstddef>
#include "../mp3-enc-bsd_amd/csrc/dmath.h"
extern "C" {
void t_dm_log(const double *x, double *y, size_t n) { for (size_t i = 0; i < n; i++) y[i] = dm_log(x[i]); }
void t_dm_log_fast(const double *x, double *y, size_t n) { for (size_t i = 0; i < n; i++) y[i] = dm_log_fast(x[i]); }
void t_dm_exp_fast(const double *x, double *y, size_t n) { for (size_t i = 0; i < n; i++) y[i] = dm_exp_fast(x[i]); }
void t_dm_exp(const double *x, double *y, size_t n) { for (size_t i = 0; i < n; i++) y[i] = dm_exp(x[i]); }
void t_dm_sin(const double *x, double *y, size_t n) { for (size_t i = 0; i < n; i++) y[i] = dm_sin(x[i]); }
void t_dm_cos(const double *x, double *y, size_t n) { for (size_t i = 0; i < n; i++) y[i] = dm_cos(x[i]); }
void t_dm_atan2_fast(const double *a, const double *b, double *y, size_t n) { for (size_t i = 0; i < n; i++) y[i] = dm_atan2_fast(a[i], b[i]); }
void t_dm_sin_fast(const double *x, double *y, size_t n) { for (size_t i = 0; i < n; i++) { double c; dm_sincos_fast(x[i], &y[i], &c); } }
void t_dm_cos_fast(const double *x, double *y, size_t n) { for (size_t i = 0; i < n; i++) { double s; dm_sincos_fast(x[i], &s, &y[i]); } }
void t_dm_cos_only_fast(const double *x, double *y, size_t n) { for (size_t i = 0; i < n; i++) y[i] = dm_cos_fast(x[i]); }
void t_dm_sin_fast_rel(const double *x, double *y, size_t n) { for (size_t i = 0; i < n; i++) { int nm; y[i] = dm_sin_fast_rel(x[i], &nm); if (nm) y[i] = __builtin_nan(""); } }
int t_dm_float_rounding_safe(double v) { return dm_float_rounding_safe(v); }
void t_dm_atan2(const double *a, const double *b, double *y, size_t n) { for (size_t i = 0; i < n; i++) y[i] = dm_atan2(a[i], b[i]); }
}
