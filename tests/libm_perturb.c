/* TEST INFRASTRUCTURE -- an LD_PRELOAD shim that makes the host's libm "another libm": sin, cos, exp, pow, log, atan2
 * and their float forms return the real result moved by one ulp.  tests/test_table_pins.py runs an encode under it:
 * the product's init tables come out of csrc/tables_blob.bin and its run-time transcendentals out of csrc/dmath.h, so
 * nothing may change.  (Build: gcc -shared -fPIC -O2 -o libm_perturb.so libm_perturb.c -ldl -lm) */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <math.h>

#define WRAP1(name)                                                     \
    double name(double x)                                               \
    {                                                                   \
        static double (*real)(double);                                  \
        if (!real) real = (double (*)(double)) dlsym(RTLD_NEXT, #name); \
        double r = real(x);                                             \
        return (r == 0.0 || r != r || isinf(r)) ? r : nextafter(r, INFINITY); \
    }
#define WRAP2(name)                                                              \
    double name(double x, double y)                                              \
    {                                                                            \
        static double (*real)(double, double);                                   \
        if (!real) real = (double (*)(double, double)) dlsym(RTLD_NEXT, #name);  \
        double r = real(x, y);                                                   \
        return (r == 0.0 || r != r || isinf(r)) ? r : nextafter(r, INFINITY);    \
    }
WRAP1(sin)
WRAP1(cos)
WRAP1(exp)
WRAP1(log)
WRAP2(pow)
WRAP2(atan2)
