/* Deterministic double-precision elementary functions for the device kernels.
 *
 * Why: the reference calls glibc's log/exp/sin/cos/atan2 on data-dependent
 * arguments (src/l3psy.c:500-509,536-545,617,627,643,712; src/subs.c:65-121;
 * src/loop.c:384,390-391,633-667).  ROCm's device libm rounds differently from
 * glibc in a large share of calls, which would leak into the bitstream.  These
 * routines use only IEEE-754 +,-,*,/ and fma (identical on gfx950 and on the
 * host), evaluate in double-double and round once, so the result is the
 * correctly rounded one except with probability of order 1e-5 per call -- glibc
 * 2.35 itself is within 1 ulp and agrees with correct rounding in >99.9 % of calls.
 *
 * The header compiles three ways: by hipcc for the kernels, by g++ for the CPU
 * unit tests of this layer (tests/test_dmath.py), and by g++ under the test-only
 * wave emulator.  No libm call is made anywhere in it.
 */
#ifndef MP3MI_DMATH_H
#define MP3MI_DMATH_H

#if defined(__HIPCC__)
#define DM_FN static __device__ __forceinline__
#define DM_TABLE static __device__ const
#else
#define DM_FN static inline
#define DM_TABLE static const
#endif

#include "dmath_tables.h"

typedef struct { double hi, lo; } dm_dd;

DM_FN double dm_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
DM_FN double dm_fabs(double a) { return __builtin_fabs(a); }

DM_FN dm_dd dm_two_sum(double a, double b)
{
    dm_dd r;
    double bb;
    r.hi = a + b;
    bb = r.hi - a;
    r.lo = (a - (r.hi - bb)) + (b - bb);
    return r;
}

DM_FN dm_dd dm_fast_two_sum(double a, double b) /* |a| >= |b| or a == 0 */
{
    dm_dd r;
    r.hi = a + b;
    r.lo = b - (r.hi - a);
    return r;
}

DM_FN dm_dd dm_two_prod(double a, double b)
{
    dm_dd r;
    r.hi = a * b;
    r.lo = dm_fma(a, b, -r.hi);
    return r;
}

DM_FN dm_dd dm_add_dd(dm_dd a, dm_dd b)
{
    dm_dd s = dm_two_sum(a.hi, b.hi), t = dm_two_sum(a.lo, b.lo);
    s.lo += t.hi;
    s = dm_fast_two_sum(s.hi, s.lo);
    s.lo += t.lo;
    return dm_fast_two_sum(s.hi, s.lo);
}

DM_FN dm_dd dm_add_dd_d(dm_dd a, double b)
{
    dm_dd s = dm_two_sum(a.hi, b);
    s.lo += a.lo;
    return dm_fast_two_sum(s.hi, s.lo);
}

DM_FN dm_dd dm_mul_dd(dm_dd a, dm_dd b)
{
    dm_dd p = dm_two_prod(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    return dm_fast_two_sum(p.hi, p.lo);
}

DM_FN dm_dd dm_mul_dd_d(dm_dd a, double b)
{
    dm_dd p = dm_two_prod(a.hi, b);
    p.lo = dm_fma(a.lo, b, p.lo);
    return dm_fast_two_sum(p.hi, p.lo);
}

DM_FN dm_dd dm_div_dd(dm_dd n, dm_dd d)
{
    double q1 = n.hi / d.hi, q2;
    dm_dd p = dm_mul_dd_d(d, q1), r;
    r = dm_two_sum(n.hi, -p.hi);
    r.lo += n.lo - p.lo;
    q2 = (r.hi + r.lo) / d.hi;
    return dm_fast_two_sum(q1, q2);
}

DM_FN long long dm_bits(double x)
{
    union { double d; long long i; } u;
    u.d = x;
    return u.i;
}

DM_FN double dm_from_bits(long long i)
{
    union { double d; long long i; } u;
    u.i = i;
    return u.d;
}

/* ------------------------------------------------------------------ log */
DM_FN double dm_log(double x)
{
    long long ix = dm_bits(x);
    int e, i;
    double m, c, t, q;
    dm_dd r, sq, acc, p;
    if (ix <= 0 || ix >= 0x7ff0000000000000LL) {
        if ((ix << 1) == 0) return -__builtin_inf();
        if (ix < 0) return __builtin_nan("");
        return x; /* +inf or nan */
    }
    e = 0;
    if (ix < 0x0010000000000000LL) { /* subnormal */
        x *= 0x1p54;
        ix = dm_bits(x);
        e = -54;
    }
    e += (int) (ix >> 52) - 1023;
    i = (int) ((ix >> 45) & 127);
    m = dm_from_bits((ix & 0x000fffffffffffffLL) | 0x3ff0000000000000LL);
    if (i >= DM_LOG_SPLIT) { m *= 0.5; e += 1; }
    c = DM_LOG_RCP[i];
    /* r = m*c - 1 exactly as a double-double, |r| <= 2^-7 */
    p = dm_two_prod(m, c);
    r = dm_fast_two_sum(p.hi - 1.0, p.lo);
    /* log1p(r) = r - r^2/2 + r^3/3 - ... - r^10/10 */
    t = r.hi;
    q = -0.1;
    q = dm_fma(q, t, 0x1.c71c71c71c71cp-4);   /*  1/9 */
    q = dm_fma(q, t, -0.125);
    q = dm_fma(q, t, 0x1.2492492492492p-3);   /*  1/7 */
    q = dm_fma(q, t, -0x1.5555555555555p-3);  /* -1/6 */
    q = dm_fma(q, t, 0.2);
    q = dm_fma(q, t, -0.25);
    q = dm_fma(q, t, 0x1.5555555555555p-2);   /*  1/3 */
    q = q * t * t * t;
    sq = dm_two_prod(t, t);
    sq.lo = dm_fma(2.0 * t, r.lo, sq.lo);
    acc.hi = -0.5 * sq.hi;
    acc.lo = -0.5 * sq.lo + q;
    acc = dm_add_dd(r, dm_fast_two_sum(acc.hi, acc.lo));
    /* + (-log c) + e*ln2 */
    p.hi = DM_LOG_NLOGC[i][0];
    p.lo = DM_LOG_NLOGC[i][1];
    acc = dm_add_dd(p, acc);
    if (e != 0) {
        double ed = (double) e;
        dm_dd l2 = dm_two_prod(ed, DM_LN2_MID);
        l2.lo = dm_fma(ed, DM_LN2_LO, l2.lo);
        l2 = dm_add_dd(dm_fast_two_sum(ed * DM_LN2_HI, l2.hi), dm_fast_two_sum(l2.lo, 0.0));
        acc = dm_add_dd(l2, acc);
    }
    return acc.hi + acc.lo;
}

/* Plain double-precision log of a positive, finite, normal x: same argument reduction and tables
 * as dm_log, everything after it in double.  Absolute error below 2^-50 * max(1, |log x|)
 * (tests/test_dmath_host.py measures < 1.5 ulp over 2^20 arguments).  NOT correctly rounded:
 * only for callers that can tell when their decision does not depend on the last bits and fall
 * back to dm_log otherwise (k_prep.hip). */
DM_FN double dm_log_fast(double x)
{
    const long long ix = dm_bits(x);
    int e = (int) (ix >> 52) - 1023;
    const int i = (int) ((ix >> 45) & 127);
    double m = dm_from_bits((ix & 0x000fffffffffffffLL) | 0x3ff0000000000000LL);
    double r, q, ed;
    if (i >= DM_LOG_SPLIT) { m *= 0.5; e += 1; }
    r = dm_fma(m, DM_LOG_RCP[i], -1.0);     /* |r| <= 2^-7, at most one rounding */
    q = -0.125;
    q = dm_fma(q, r, 0x1.2492492492492p-3);   /*  1/7 */
    q = dm_fma(q, r, -0x1.5555555555555p-3);  /* -1/6 */
    q = dm_fma(q, r, 0.2);
    q = dm_fma(q, r, -0.25);
    q = dm_fma(q, r, 0x1.5555555555555p-2);   /*  1/3 */
    q = dm_fma(q, r, -0.5);
    q = q * r * r;                            /* log1p(r) - r, truncated after r^8/8 (< 2^-59) */
    ed = (double) e;
    q = q + dm_fma(ed, DM_LN2_MID, DM_LOG_NLOGC[i][1]);
    q = q + r;
    return dm_fma(ed, DM_LN2_HI, DM_LOG_NLOGC[i][0]) + q;   /* e*LN2_HI is exact: 36-bit constant */
}

/* ------------------------------------------------------------------ exp */
DM_FN double dm_exp(double x)
{
    double kd, t, q, scale;
    long long k;
    int j, qe;
    dm_dd r, p, er, res, T;
    if (x != x) return x;
    if (x > 709.78) return __builtin_inf();
    if (x < -745.2) return 0.0;
    kd = __builtin_rint(x * DM_INV_L64);
    k = (long long) kd;
    /* r = x - k*ln2/64 as a double-double */
    r = dm_two_sum(x, -kd * DM_L64_HI);     /* product exact: 36-bit constant, |k| < 2^17 */
    p = dm_two_prod(-kd, DM_L64_MID);
    p.lo = dm_fma(-kd, DM_L64_LO, p.lo);
    r = dm_add_dd(r, p);
    t = r.hi;
    /* e^r - 1 - r = r^2/2 + r^3/6 + ... + r^9/9!  (|r| <= 0.0055) */
    q = 0x1.71de3a556c734p-19;              /* 1/9! */
    q = dm_fma(q, t, 0x1.a01a01a01a01ap-16); /* 1/8! */
    q = dm_fma(q, t, 0x1.a01a01a01a01ap-13); /* 1/7! */
    q = dm_fma(q, t, 0x1.6c16c16c16c17p-10); /* 1/6! */
    q = dm_fma(q, t, 0x1.1111111111111p-7);  /* 1/5! */
    q = dm_fma(q, t, 0x1.5555555555555p-5);  /* 1/4! */
    q = dm_fma(q, t, 0x1.5555555555555p-3);  /* 1/3! */
    q = q * t * t * t;
    p = dm_two_prod(t, t);
    p.lo = dm_fma(2.0 * t, r.lo, p.lo);
    er.hi = 0.5 * p.hi;
    er.lo = 0.5 * p.lo + q;
    er = dm_add_dd(r, dm_fast_two_sum(er.hi, er.lo)); /* e^r - 1 */
    j = (int) (k & 63);
    qe = (int) ((k - j) / 64);
    T.hi = DM_EXP2_64[j][0];
    T.lo = DM_EXP2_64[j][1];
    res = dm_add_dd(T, dm_mul_dd(T, er));
    /* scale by 2^qe; results below the normal range take a second rounding, which no
       caller can reach (arguments stay within [-200, 200]) */
    if (qe > -1000 && qe < 1000) {
        scale = dm_from_bits((long long) (qe + 1023) << 52);
        return (res.hi + res.lo) * scale;
    }
    if (qe >= 1000) {
        scale = dm_from_bits((long long) (qe - 600 + 1023) << 52);
        return ((res.hi + res.lo) * 0x1p600) * scale;
    }
    scale = dm_from_bits((long long) (qe + 600 + 1023) << 52);
    return ((res.hi + res.lo) * 0x1p-600) * scale;
}

/* Plain double-precision exp for |x| < 700: the same reduction and table as dm_exp, everything after it
 * in double.  |error| < 2^-50 relative (tests/test_dmath_host.py).  NOT correctly rounded: for callers
 * that round a product with the result to float behind dm_float_rounding_safe_ulps (k_psy.hip). */
DM_FN double dm_exp_fast(double x)
{
    const double kd = __builtin_rint(x * DM_INV_L64);
    const long long k = (long long) kd;
    const int j = (int) (k & 63), qe = (int) ((k - j) / 64);
    /* r = x - k*ln2/64: the first product is exact (36-bit constant, |k| < 2^17) */
    const double r = dm_fma(-kd, DM_L64_MID, dm_fma(-kd, DM_L64_HI, x));
    double q = 0x1.6c16c16c16c17p-10;         /* 1/6! */
    q = dm_fma(q, r, 0x1.1111111111111p-7);  /* 1/5! */
    q = dm_fma(q, r, 0x1.5555555555555p-5);  /* 1/4! */
    q = dm_fma(q, r, 0x1.5555555555555p-3);  /* 1/3! */
    q = dm_fma(q, r, 0.5);
    q = dm_fma(q * r, r, r);                  /* e^r - 1, |r| <= 0.0055: truncation < 2^-60 */
    return dm_fma(DM_EXP2_64[j][0], q, DM_EXP2_64[j][1] + DM_EXP2_64[j][0]) * dm_from_bits((long long) (qe + 1023) << 52);
}

/* ------------------------------------------------------------- sin / cos */
/* reduce x to r in [-pi/4, pi/4] (double-double) and quadrant n; |x| < 2^20 */
DM_FN int dm_rem_pio2(double x, dm_dd *r)
{
    double kd = __builtin_rint(x * DM_2_OVER_PI);
    dm_dd a, p;
    a = dm_two_sum(x, -kd * DM_PIO2_1);  /* exact products: 30-bit parts, |k| < 2^21 */
    a = dm_add_dd_d(a, -kd * DM_PIO2_2);
    a = dm_add_dd_d(a, -kd * DM_PIO2_3);
    p = dm_two_prod(-kd, DM_PIO2_4);
    p.lo = dm_fma(-kd, DM_PIO2_5, p.lo);
    *r = dm_add_dd(a, p);
    return (int) ((long long) kd & 3);
}

/* sin and cos of a reduced double-double argument, both as double-double */
DM_FN void dm_sincos_kernel(dm_dd r, dm_dd *s, dm_dd *c)
{
    double a = dm_fabs(r.hi), jd, d, d2, ps, pc;
    int j, neg = r.hi < 0;
    dm_dd dd, sd, cd, S, C, t1, t2;
    if (neg) { r.hi = -r.hi; r.lo = -r.lo; }
    jd = __builtin_rint(a * 64.0);
    j = (int) jd;
    dd = dm_fast_two_sum(r.hi - jd * 0.015625, r.lo); /* d = r - j/64, |d| <= 1/128 */
    d = dd.hi;
    d2 = d * d;
    /* sin d = d + d^3*ps,  cos d = 1 - d^2/2 + d^4*pc */
    ps = -0x1.71de3a556c734p-19;                 /* -1/9! */
    ps = dm_fma(ps, d2, 0x1.a01a01a01a01ap-13);  /*  1/7! */
    ps = dm_fma(ps, d2, -0x1.1111111111111p-7);  /* -1/5! */
    ps = dm_fma(ps, d2, 0x1.5555555555555p-3);   /*  1/3! ... sign fixed below */
    ps = -ps;
    pc = 0x1.27e4fb7789f5cp-22;                  /*  1/10! */
    pc = dm_fma(pc, d2, -0x1.a01a01a01a01ap-16); /* -1/8! */
    pc = dm_fma(pc, d2, 0x1.6c16c16c16c17p-10);  /*  1/6! */
    pc = dm_fma(pc, d2, -0x1.5555555555555p-5);  /* -1/4! */
    pc = -pc;
    sd = dm_add_dd_d(dd, ps * d2 * d);
    {
        dm_dd sq = dm_two_prod(d, d);
        sq.lo = dm_fma(2.0 * d, dd.lo, sq.lo);
        cd.hi = -0.5 * sq.hi;
        cd.lo = -0.5 * sq.lo + pc * d2 * d2;
        cd = dm_fast_two_sum(cd.hi, cd.lo); /* cos d - 1 */
    }
    S.hi = DM_SIN_J64[j][0]; S.lo = DM_SIN_J64[j][1];
    C.hi = DM_COS_J64[j][0]; C.lo = DM_COS_J64[j][1];
    /* sin(t+d) = S + (S*(cos d - 1) + C*sin d);  cos(t+d) = C + (C*(cos d - 1) - S*sin d) */
    t1 = dm_mul_dd(S, cd);
    t2 = dm_mul_dd(C, sd);
    *s = dm_add_dd(S, dm_add_dd(t1, t2));
    t1 = dm_mul_dd(C, cd);
    t2 = dm_mul_dd(S, sd);
    t2.hi = -t2.hi; t2.lo = -t2.lo;
    *c = dm_add_dd(C, dm_add_dd(t1, t2));
    if (neg) { s->hi = -s->hi; s->lo = -s->lo; }
}

DM_FN double dm_sin(double x)
{
    dm_dd r, s, c;
    int n;
    if (!(dm_fabs(x) < 0x1p20)) return x - x; /* out of the supported range: nan */
    if (dm_fabs(x) < 0x1p-27) return x;
    n = dm_rem_pio2(x, &r);
    dm_sincos_kernel(r, &s, &c);
    switch (n) {
    case 0: return s.hi + s.lo;
    case 1: return c.hi + c.lo;
    case 2: return -(s.hi + s.lo);
    default: return -(c.hi + c.lo);
    }
}

/* sin and cos of the same argument with one reduction; each equals dm_sin / dm_cos bit for bit */
DM_FN void dm_sincos(double x, double *sn, double *cs)
{
    dm_dd r, s, c;
    double sv, cv;
    int n;
    if (!(dm_fabs(x) < 0x1p20)) { *sn = x - x; *cs = x - x; return; }
    n = dm_rem_pio2(x, &r);
    dm_sincos_kernel(r, &s, &c);
    sv = s.hi + s.lo;
    cv = c.hi + c.lo;
    switch (n) {
    case 0: *sn = sv; *cs = cv; break;
    case 1: *sn = cv; *cs = -sv; break;
    case 2: *sn = -sv; *cs = -cv; break;
    default: *sn = -cv; *cs = sv; break;
    }
    if (dm_fabs(x) < 0x1p-27) *sn = x;
}

/* Plain double-precision sin and cos of the same argument, |x| < 2^10: Cody-Waite reduction by the same three
 * parts of pi/2 (the first product and difference exact, one rounding after it) and the classic degree-13 / degree-14
 * polynomials on [-pi/4, pi/4].  |error| < 2^-51 absolute on each (tests/test_dmath_host.py).  NOT correctly
 * rounded: for callers that can tell when the last bits of the result do not matter (k_cw's first tier, whose
 * unpredictability only reaches a bit through the float-rounded partition sums of k_part). */
DM_FN void dm_sincos_fast(double x, double *sn, double *cs)
{
    const double kd = __builtin_rint(x * DM_2_OVER_PI);
    const int n = (int) ((long long) kd & 3);
    double r = dm_fma(-kd, DM_PIO2_1, x);      /* exact: 30-bit part, |k| < 2^10, and the difference cancels */
    double z, ps, pc, sv, cv;
    r = dm_fma(-kd, DM_PIO2_2, r);
    r = dm_fma(-kd, DM_PIO2_3, r);
    z = r * r;
    ps = 0x1.5d93a5acfd57cp-33;                  /*  1.58969099521155010221e-10 */
    ps = dm_fma(ps, z, -0x1.ae5e68a2b9cebp-26);  /* -2.50507602534068634195e-08 */
    ps = dm_fma(ps, z, 0x1.71de357b1fe7dp-19);   /*  2.75573137070700676789e-06 */
    ps = dm_fma(ps, z, -0x1.a01a019c161d5p-13);  /* -1.98412698298579493134e-04 */
    ps = dm_fma(ps, z, 0x1.111111110f8a6p-7);    /*  8.33333333332248946124e-03 */
    ps = dm_fma(ps, z, -0x1.5555555555549p-3);   /* -1.66666666666666324348e-01 */
    sv = dm_fma(ps * z, r, r);
    pc = -0x1.8fae9be8838d4p-37;                 /* -1.13596475577881948265e-11 */
    pc = dm_fma(pc, z, 0x1.1ee9ebdb4b1c4p-29);   /*  2.08757232129817482790e-09 */
    pc = dm_fma(pc, z, -0x1.27e4f809c52adp-22);  /* -2.75573143513906633035e-07 */
    pc = dm_fma(pc, z, 0x1.a01a019cb1590p-16);   /*  2.48015872894767294178e-05 */
    pc = dm_fma(pc, z, -0x1.6c16c16c15177p-10);  /* -1.38888888888741095749e-03 */
    pc = dm_fma(pc, z, 0x1.555555555554cp-5);    /*  4.16666666666666019037e-02 */
    cv = dm_fma(pc * z, z, dm_fma(-0.5, z, 1.0));
    switch (n) {
    case 0: *sn = sv; *cs = cv; break;
    case 1: *sn = cv; *cs = -sv; break;
    case 2: *sn = -sv; *cs = -cv; break;
    default: *sn = -cv; *cs = sv; break;
    }
}

/* Plain double-precision cosine alone, |x| < 2^10: reduction by multiples of PI (the same three parts of pi/2, taken
 * twice) to |r| <= pi/2 and ONE even polynomial there -- the Taylor series to r^22 (the next term is below 2^-63) --, sign
 * by the parity of the multiple.  |error| < 2^-51 absolute (tests/test_dmath_host.py).  About half the instructions of
 * dm_sincos_fast: for callers that need no sine (k12_psy's first tier, k_l12.hip).  NOT correctly rounded. */
DM_FN double dm_cos_fast(double x)
{
    const double kd = __builtin_rint(x * (0.5 * DM_2_OVER_PI)); /* multiples of pi */
    const double k2 = kd + kd;
    double r = dm_fma(-k2, DM_PIO2_1, x);      /* exact, as in dm_sincos_fast */
    double z, p;
    r = dm_fma(-k2, DM_PIO2_2, r);
    r = dm_fma(-k2, DM_PIO2_3, r);
    z = r * r;
    p = 0x1.0ce396db7f853p-70;                  /*  1/22! */
    p = dm_fma(p, z, -0x1.e542ba4020225p-62);   /* -1/20! */
    p = dm_fma(p, z, 0x1.6827863b97d97p-53);    /*  1/18! */
    p = dm_fma(p, z, -0x1.ae7f3e733b81fp-45);   /* -1/16! */
    p = dm_fma(p, z, 0x1.93974a8c07c9dp-37);    /*  1/14! */
    p = dm_fma(p, z, -0x1.1eed8eff8d898p-29);   /* -1/12! */
    p = dm_fma(p, z, 0x1.27e4fb7789f5cp-22);    /*  1/10! */
    p = dm_fma(p, z, -0x1.a01a01a01a01ap-16);   /* -1/8! */
    p = dm_fma(p, z, 0x1.6c16c16c16c17p-10);    /*  1/6! */
    p = dm_fma(p, z, -0x1.5555555555555p-5);    /* -1/4! */
    p = dm_fma(p, z, 0.5);
    p = dm_fma(-p, z, 1.0);
    return dm_from_bits(dm_bits(p) ^ (long long) ((unsigned long long) ((long long) kd & 1) << 63));
}

/* Plain double-precision sine with a small RELATIVE error, |x| <= 8: reduction by multiples of PI (three 30-bit parts of
 * pi/2 taken twice: exact products, pi to 2^-90) to |r| <= pi/2 and the odd Taylor series to r^23 there; sign by the parity of
 * the multiple.  Relative error below 2^-49 (tests/test_dmath_host.py) UNLESS the argument lies within 2^-30 of a non-zero
 * multiple of pi, where what the three parts leave out shows: then *near_multiple is set and the caller must not rely on the
 * value.  For k12_psy's first tier (k_l12.hip), which needs sin((phi - phi') / 2) of a line that is predicted almost
 * perfectly -- a tiny sine -- as accurately as any other.  NOT correctly rounded. */
DM_FN double dm_sin_fast_rel(double x, int *near_multiple)
{
    const double kd = __builtin_rint(x * (0.5 * DM_2_OVER_PI)); /* multiples of pi */
    const double k2 = kd + kd;
    double r = dm_fma(-k2, DM_PIO2_1, x);      /* exact, as in dm_sincos_fast */
    double z, q, sv;
    r = dm_fma(-k2, DM_PIO2_2, r);
    r = dm_fma(-k2, DM_PIO2_3, r);
    *near_multiple = kd != 0.0 && dm_fabs(r) < 0x1p-30;
    z = r * r;
    q = -0x1.761b41316381ap-75;                 /* -1/23! */
    q = dm_fma(q, z, 0x1.71b8ef6dcf572p-66);    /*  1/21! */
    q = dm_fma(q, z, -0x1.2f49b46814157p-57);   /* -1/19! */
    q = dm_fma(q, z, 0x1.952c77030ad4ap-49);    /*  1/17! */
    q = dm_fma(q, z, -0x1.ae7f3e733b81fp-41);   /* -1/15! */
    q = dm_fma(q, z, 0x1.6124613a86d09p-33);    /*  1/13! */
    q = dm_fma(q, z, -0x1.ae64567f544e4p-26);   /* -1/11! */
    q = dm_fma(q, z, 0x1.71de3a556c734p-19);    /*  1/9! */
    q = dm_fma(q, z, -0x1.a01a01a01a01ap-13);   /* -1/7! */
    q = dm_fma(q, z, 0x1.1111111111111p-7);     /*  1/5! */
    q = dm_fma(q, z, -0x1.5555555555555p-3);    /* -1/3! */
    sv = dm_fma(r * z, q, r);
    return dm_from_bits(dm_bits(sv) ^ (long long) ((unsigned long long) ((long long) kd & 1) << 63));
}

DM_FN double dm_cos(double x)
{
    dm_dd r, s, c;
    int n;
    if (!(dm_fabs(x) < 0x1p20)) return x - x;
    n = dm_rem_pio2(x, &r);
    dm_sincos_kernel(r, &s, &c);
    switch (n) {
    case 0: return c.hi + c.lo;
    case 1: return -(s.hi + s.lo);
    case 2: return -(c.hi + c.lo);
    default: return s.hi + s.lo;
    }
}

/* ---------------------------------------------------------------- atan2 */
DM_FN double dm_atan2(double y, double x)
{
    double ay = dm_fabs(y), ax = dm_fabs(x), a, b, tq, jd, cj, u, u2, pu, res;
    int ysign = dm_bits(y) < 0, xsign = dm_bits(x) < 0, swap, j;
    dm_dd num, den, uu, v, p;
    if (x != x || y != y) return x + y;
    if (ay == 0.0) { /* +-0 or +-pi */
        res = xsign ? DM_PI_HI : 0.0;
        return ysign ? -res : res;
    }
    if (ax == 0.0) return ysign ? -DM_PIO2_HI : DM_PIO2_HI;
    if (ax == __builtin_inf() || ay == __builtin_inf()) {
        if (ax == ay) res = xsign ? 3.0 * (DM_PI_HI / 4.0) : DM_PI_HI / 4.0;
        else if (ay == __builtin_inf()) res = DM_PIO2_HI;
        else res = xsign ? DM_PI_HI : 0.0;
        return ysign ? -res : res;
    }
    swap = ay > ax;
    a = swap ? ax : ay; /* a <= b */
    b = swap ? ay : ax;
    tq = a / b;
    if (tq < 0x1p-60) { /* atan(t) = t to far below 1 ulp; also keeps a, b in range below */
        v.hi = tq;
        v.lo = 0.0;
        if (!swap && !xsign) return ysign ? -tq : tq; /* includes gradual underflow of tq */
    } else {
        /* scale so that products below cannot overflow or lose bits */
        long long eb;
        double sc;
        if (b >= 0x1p1000) { a *= 0x1p-100; b *= 0x1p-100; }
        eb = (dm_bits(b) >> 52) & 0x7ff;
        sc = dm_from_bits((long long) (2046 - eb) << 52);
        a *= sc;
        b *= sc;
        jd = __builtin_rint(tq * 64.0);
        j = (int) jd;
        cj = jd * 0.015625;
        /* u = (a - c b) / (b + c a), |u| <= ~1/128 */
        p = dm_two_prod(cj, b);
        num = dm_two_sum(a, -p.hi);
        num.lo -= p.lo;
        num = dm_fast_two_sum(num.hi, num.lo);
        p = dm_two_prod(cj, a);
        den = dm_two_sum(b, p.hi);
        den.lo += p.lo;
        den = dm_fast_two_sum(den.hi, den.lo);
        uu = dm_div_dd(num, den);
        u = uu.hi;
        u2 = u * u;
        /* atan u = u - u^3/3 + u^5/5 - u^7/7 + u^9/9 */
        pu = 0x1.c71c71c71c71cp-4;
        pu = dm_fma(pu, u2, -0x1.2492492492492p-3);
        pu = dm_fma(pu, u2, 0.2);
        pu = dm_fma(pu, u2, -0x1.5555555555555p-2);
        v = dm_add_dd_d(uu, pu * u2 * u);
        p.hi = DM_ATAN_J64[j][0];
        p.lo = DM_ATAN_J64[j][1];
        v = dm_add_dd(p, v);
    }
    if (swap) {
        p.hi = DM_PIO2_HI; p.lo = DM_PIO2_LO;
        v.hi = -v.hi; v.lo = -v.lo;
        v = dm_add_dd(p, v);
    }
    if (xsign) {
        p.hi = DM_PI_HI; p.lo = DM_PI_LO;
        v.hi = -v.hi; v.lo = -v.lo;
        v = dm_add_dd(p, v);
    }
    res = v.hi + v.lo;
    return ysign ? -res : res;
}

/* Plain double-precision atan2 for finite, non-zero x and y: the same reduction (a/b <= 1, the
 * nearest j/64, atan of the small remainder) with everything in double.  |error| < 2^-50 |result|
 * (tests/test_dmath_host.py).  NOT correctly rounded: for callers that round the result to float and
 * can tell when that rounding does not depend on the last bits (dm_float_rounding_safe). */
DM_FN double dm_atan2_fast(double y, double x)
{
    const double ay = dm_fabs(y), ax = dm_fabs(x);
    const int ysign = dm_bits(y) < 0, xsign = dm_bits(x) < 0, swap = ay > ax;
    double a = swap ? ax : ay, b = swap ? ay : ax; /* a <= b */
    const double tq = a / b;
    double v;
    if (tq < 0x1p-30) v = tq; /* atan t = t (1 - t^2/3 ...): below 2^-60 relative */
    else {
        const double jd = __builtin_rint(tq * 64.0), cj = jd * 0.015625;
        const int j = (int) jd;
        const long long eb = (dm_bits(b) >> 52) & 0x7ff; /* scale b to [1, 2): no overflow or underflow below */
        const double sc = dm_from_bits((long long) (2046 - eb) << 52);
        double u, u2, pu;
        a *= sc;
        b *= sc;
        u = dm_fma(-cj, b, a) / dm_fma(cj, a, b); /* |u| <= ~1/128 */
        u2 = u * u;
        pu = 0x1.c71c71c71c71cp-4;
        pu = dm_fma(pu, u2, -0x1.2492492492492p-3);
        pu = dm_fma(pu, u2, 0.2);
        pu = dm_fma(pu, u2, -0x1.5555555555555p-2);
        v = DM_ATAN_J64[j][0] + (DM_ATAN_J64[j][1] + dm_fma(pu * u2, u, u));
    }
    if (swap) v = DM_PIO2_HI + (DM_PIO2_LO - v);
    if (xsign) v = DM_PI_HI + (DM_PI_LO - v);
    return ysign ? -v : v;
}

/* Does (float) v depend on more than the leading bits of v?  false when v, known to within
 * 2^-46 |v|, lies that close to the midpoint of two floats (or outside the floats' normal range). */
DM_FN int dm_float_rounding_safe_ulps(double v, long long ulps) /* ulps of v (2^-52 relative) that v may be off by */
{
    const long long b = dm_bits(v) & 0x7fffffffffffffffLL;
    const long long d = (b & 0x1fffffffLL) - 0x10000000LL;
    if (b < 0x3810000000000000LL || b >= 0x47f0000000000000LL) return 0; /* |v| < 2^-126 or >= 2^128 (or nan) */
    return (d < 0 ? -d : d) > ulps;
}
DM_FN int dm_float_rounding_safe(double v) { return dm_float_rounding_safe_ulps(v, 64); }
/* distance of v from the nearest midpoint of two floats, in ulps of v (diagnostics) */
DM_FN long long dm_float_midpoint_distance_ulps(double v)
{
    const long long d = (dm_bits(v) & 0x1fffffffLL) - 0x10000000LL;
    return d < 0 ? -d : d;
}

#endif
