// ce_math.h -- scalar fp64 log / exp with LDS-resident coefficient tables (included inside an anonymous namespace by ce_common.h;
// plain C++ apart from the qualifiers and two builtins, so tests/test_expcone_host.py exercises it on the host)
#pragma once
#ifndef CE_MATH_HOST
#define CE_FREXP_EXP(x) __builtin_amdgcn_frexp_exp(x)
#define CE_FREXP_MANT(x) __builtin_amdgcn_frexp_mant(x)
#define CE_BITS_TO_DOUBLE(b) __longlong_as_double((long long)(b))
#endif

// ------------------------------------------------------------------------------------------------
// log / exp for the adaptive-scale update (once per check interval, on wave-uniform values), with the polynomial
// coefficients read from an LDS table instead of instruction literals: the compiler's exp / log expansions carry ~20 fp64
// literals which loop-invariant code motion parks in VGPRs across the iteration loop (and spills: 124 B of scratch per lane in
// round 1, i.e. 130 MB of scratch writes per launch).  Algorithms: fdlibm's e_log.c / e_exp.c (< 1 ulp); sqrt(exp(x)) callers use
// exp(x / 2).  tab: CE_MATH_TAB doubles, filled by ce_math_table_init.
constexpr int CE_MATH_TAB = 18;
__device__ __forceinline__ void ce_math_table_init(double *tab, int tid) {
    // 0..6 Lg1..Lg7, 7 ln2_hi, 8 ln2_lo, 9..13 P1..P5, 14 1/ln2, 15 sqrt(1/2), 16 2^54, 17 unused
    const unsigned long long bits[CE_MATH_TAB] = {
        0x3FE5555555555593ull, 0x3FD999999997FA04ull, 0x3FD2492494229359ull, 0x3FCC71C51D8E78AFull, 0x3FC7466496CB03DEull,
        0x3FC39A09D078C69Full, 0x3FC2F112DF3E5244ull, 0x3FE62E42FEE00000ull, 0x3DEA39EF35793C76ull,
        0x3FC555555555553Eull, 0xBF66C16C16BEBD93ull, 0x3F11566AAF25DE2Cull, 0xBEBBBD41C5D26BF1ull, 0x3E66376972BEA4D0ull,
        0x3FF71547652B82FEull, 0x3FE6A09E667F3BCDull, 0x4350000000000000ull, 0ull};
    if (tid < CE_MATH_TAB) tab[tid] = CE_BITS_TO_DOUBLE(bits[tid]);
}
__device__ __forceinline__ double ce_log(double x, const double *tab) {      // x > 0, finite
    int e = 0;
    if (x < 2.2250738585072014e-308) { x *= tab[16]; e = -54; }              // subnormal
    e += CE_FREXP_EXP(x);
    double mnt = CE_FREXP_MANT(x);                               // [0.5, 1)
    if (mnt < tab[15]) { mnt += mnt; e -= 1; }                                 // [sqrt(1/2), sqrt(2))
    const double f = mnt - 1.0, s = f / (2.0 + f), z = s * s, w = z * z;
    const double t1 = w * (tab[1] + w * (tab[3] + w * tab[5]));
    const double t2 = z * (tab[0] + w * (tab[2] + w * (tab[4] + w * tab[6])));
    const double R = t2 + t1, hfsq = 0.5 * f * f, dk = (double)e;
    return dk * tab[7] - ((hfsq - (s * (hfsq + R) + dk * tab[8])) - f);
}
__device__ __forceinline__ double ce_exp(double x, const double *tab) {      // |x| < 700
    const double k = rint(x * tab[14]);
    const double hi = x - k * tab[7], lo = k * tab[8], r = hi - lo, t = r * r;
    const double c = r - t * (tab[9] + t * (tab[10] + t * (tab[11] + t * (tab[12] + t * tab[13]))));
    const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    return ldexp(y, (int)k);
}

