/*
 * rt_math.h - the three transcendental functions of the path tracer's sampling code, written with IEEE-754 basic
 * operations only (+, -, *, /, fma, integer bit moves), so that they give the same bits on the GPU and on any CPU.
 *
 * The reference calls libm: cos / sin / log in the Box-Muller camera jitter (raytracer.cpp:1467-1473) and cos / sin in
 * random_cos (raytracer.cpp:116-118). A GPU has no access to the host's libm, and two libms differ in the last bit of
 * some results; a last-bit difference in a direction can move a ray across a silhouette. With these functions on both
 * sides the device code and the CPU checker walk the same paths, and what remains to be explained is the difference
 * between these functions and the host's libm, which a CPU-only test can do.
 *
 * Accuracy (measured against 80-bit long double on 10^6 arguments each, tests/test_portable_math.py): below 1 ulp.
 * Compile without floating-point contraction (-ffp-contract=off): every fma here is explicit.
 */
#ifndef RT_MATH_H
#define RT_MATH_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define RTM_FN __host__ __device__ __forceinline__
#else
#define RTM_FN static inline
#endif

namespace rtm {

RTM_FN double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
RTM_FN uint64_t bits_of(double x) {
    uint64_t u;
    memcpy(&u, &x, 8);
    return u;
}
RTM_FN double from_bits(uint64_t u) {
    double x;
    memcpy(&x, &u, 8);
    return x;
}

/* sin and cos of x for 0 <= x < 7 (the callers pass 2 pi u with u in [0, 1)); NaN for anything else.
 * Reduction: n = nearest integer to x * 2/pi (0 .. 4), r = x - n pi/2 with pi/2 in three pieces of 33 + 33 + 53 bits:
 * n * (first piece) is exact and so is its subtraction from x (both within a factor of two), the second and third
 * pieces give r as a head r0 and a tail r1 with an error below 1e-36. Then the Taylor polynomials on |r| <= pi/4 up to
 * r^17 (sine) and r^18 (cosine): truncation below 2^-60 relative. */
RTM_FN void sincos_0_7(double x_in, double *sn, double *cs) {
    /* no branches (an argument outside the range is computed as 0 and replaced at the end): two calls in a row are then
     * one straight piece of code whose two chains of dependent operations the compiler can issue interleaved */
    const bool bad = !(x_in >= 0.0 && x_in < 7.0);
    const double x = bad ? 0.0 : x_in;
    const int n = (int)(x * 0.63661977236758138243 + 0.5); /* x * 2/pi is below 4.46 */
    const double fn = (double)n;
    const double p1 = 1.57079632673412561417e+00;  /* 0x3FF921FB54400000: the first 33 bits of pi/2 */
    const double p2 = 6.07710050630396597660e-11;  /* 0x3DD0B4611A600000: the next 33 bits */
    const double p2t = 2.02226624879595063154e-21; /* 0x3BA3198A2E037073: pi/2 - p1 - p2 */
    const double z0 = x - fn * p1; /* exact */
    double w = fn * p2;
    const double r = z0 - w;
    w = fn * p2t - ((z0 - r) - w);
    const double r0 = r - w;
    const double r1 = (r - r0) - w;
    const double z = r0 * r0;
    /* sin(r0 + r1) = r0 + r0 z S(z) + r1 (1 - z / 2), S(z) = -1/3! + z/5! - ... */
    double s = 1.0 / 355687428096000.0;                 /* 1/17! */
    s = fma_(s, z, -1.0 / 1307674368000.0);             /* 1/15! */
    s = fma_(s, z, 1.0 / 6227020800.0);                 /* 1/13! */
    s = fma_(s, z, -1.0 / 39916800.0);                  /* 1/11! */
    s = fma_(s, z, 1.0 / 362880.0);                     /* 1/9! */
    s = fma_(s, z, -1.0 / 5040.0);                      /* 1/7! */
    s = fma_(s, z, 1.0 / 120.0);                        /* 1/5! */
    s = fma_(s, z, -1.0 / 6.0);                         /* 1/3! */
    const double sr = r0 + fma_(r0 * z, s, r1 * fma_(-0.5, z, 1.0));
    /* cos(r0 + r1) = 1 - z/2 + z^2 C(z) - r0 r1, C(z) = 1/4! - z/6! + ...; the leading terms as w = 1 - z/2 (rounded)
     * plus the part of z/2 that the rounding dropped */
    double c = -1.0 / 6402373705728000.0;               /* 1/18! */
    c = fma_(c, z, 1.0 / 20922789888000.0);             /* 1/16! */
    c = fma_(c, z, -1.0 / 87178291200.0);               /* 1/14! */
    c = fma_(c, z, 1.0 / 479001600.0);                  /* 1/12! */
    c = fma_(c, z, -1.0 / 3628800.0);                   /* 1/10! */
    c = fma_(c, z, 1.0 / 40320.0);                      /* 1/8! */
    c = fma_(c, z, -1.0 / 720.0);                       /* 1/6! */
    c = fma_(c, z, 1.0 / 24.0);                         /* 1/4! */
    const double hz = 0.5 * z;
    const double w1 = 1.0 - hz;
    const double cr = w1 + (((1.0 - w1) - hz) + fma_(z * z, c, -(r0 * r1)));
    /* quadrant: x = r + n pi/2 */
    const int q = n & 3;
    const double s_out = (q & 1) ? cr : sr;
    const double c_out = (q & 1) ? sr : cr;
    const double s_fin = (q & 2) ? -s_out : s_out;
    const double c_fin = (q == 1 || q == 2) ? -c_out : c_out;
    *sn = bad ? __builtin_nan("") : s_fin;
    *cs = bad ? __builtin_nan("") : c_fin;
}

/* Natural logarithm of a positive normal double (the callers pass a uniform draw u = k 2^-53 in [0, 1)). log(+0) and
 * log(-0) are -infinity as in libm; negative, subnormal... are not used by the callers: negative and NaN give NaN,
 * +infinity gives +infinity, subnormals are treated as 2^-1022 times their mantissa field read as a normal number's
 * (wrong, documented, unreachable).
 * x = 2^e m with m in (sqrt(1/2), sqrt(2)]; f = m - 1 (exact); s = f / (2 + f); log m = 2 atanh s = 2 s + s R with
 * R = z (2/3 + 2 z/5 + 2 z^2/7 + ...), z = s^2 <= 0.0295, ten terms (truncation below 2^-60); evaluated in the form
 * log m = f - (f^2/2 - s (f^2/2 + R)), which keeps the division's rounding error out of the leading term; e log 2 is added
 * in two pieces, the first with 11 trailing zero bits so that e * (first piece) is exact. */
RTM_FN double log_pos(double x) {
    /* no branches, for the same reason as above: the special arguments are replaced at the end */
    const uint64_t bx = bits_of(x);
    const bool is_zero = (bx << 1) == 0, is_nan = (bx >> 63) || x != x, is_inf = bx == 0x7FF0000000000000ull;
    const int e0 = (int)((bx >> 52) & 0x7FF) - 1023;
    const double m0 = from_bits((bx & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull); /* [1, 2) */
    const bool upper = m0 > 1.41421356237309514547;
    const double m = upper ? m0 * 0.5 : m0;
    const int e = upper ? e0 + 1 : e0;
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s;
    double q = 2.0 / 21.0;
    q = fma_(q, z, 2.0 / 19.0);
    q = fma_(q, z, 2.0 / 17.0);
    q = fma_(q, z, 2.0 / 15.0);
    q = fma_(q, z, 2.0 / 13.0);
    q = fma_(q, z, 2.0 / 11.0);
    q = fma_(q, z, 2.0 / 9.0);
    q = fma_(q, z, 2.0 / 7.0);
    q = fma_(q, z, 2.0 / 5.0);
    q = fma_(q, z, 2.0 / 3.0);
    const double R = z * q;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)e;
    const double ln2_hi = 0.6931471805598903 /* 0x1.62e42fefa3800p-1 */, ln2_lo = 5.497923018708371e-14 /* 0x1.ef35793c76730p-45 */;
    const double r = dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    return is_zero ? -__builtin_huge_val() : (is_nan ? __builtin_nan("") : (is_inf ? x : r));
}

} /* namespace rtm */

#endif /* RT_MATH_H */
