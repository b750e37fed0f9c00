// ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle/hlsl.h).
//
// Deterministic reference results for the five transcendental instructions of gfx950 the numerics contract uses (v_rcp_f32, v_sqrt_f32,
// v_rsq_f32, v_exp_f32, v_log_f32). The oracle reproduces an instruction bit for bit as "reference result + a deviation of a few ulps", the
// deviation coming from a per-mantissa table measured on the device (tools/hw_tables.hip, which includes THIS header, so the table and its
// consumer can never disagree about the reference). The references are therefore not required to be correctly rounded -- they are (up to
// astronomically rare cases) -- but to be a pure function of their argument on every machine: only IEEE-754 double operations (+ - * /
// sqrt, std::fma, which are correctly rounded everywhere) in a fixed order, no libm transcendentals.
#pragma once

#include <cmath>
#include <cstdint>

namespace hwref {

inline float RefRcp(float x) { return (float)(1.0 / (double)x); }
inline float RefSqrt(float x) { return (float)std::sqrt((double)x); }
inline float RefRsq(float x) { return (float)(1.0 / std::sqrt((double)x)); }

// 2^t in double: t = i + f with f in [-0.5, 0.5], Taylor series of exp(f * ln 2) to degree 14 (truncation < 1e-17), Horner with fma
inline double Exp2Double(double t) {
    const double i = std::floor(t + 0.5);
    const double y = (t - i) * 0.693147180559945309417232121458;
    double p = 1.0 / 87178291200.0; // 1 / 14!
    p = std::fma(p, y, 1.0 / 6227020800.0);
    p = std::fma(p, y, 1.0 / 479001600.0);
    p = std::fma(p, y, 1.0 / 39916800.0);
    p = std::fma(p, y, 1.0 / 3628800.0);
    p = std::fma(p, y, 1.0 / 362880.0);
    p = std::fma(p, y, 1.0 / 40320.0);
    p = std::fma(p, y, 1.0 / 5040.0);
    p = std::fma(p, y, 1.0 / 720.0);
    p = std::fma(p, y, 1.0 / 120.0);
    p = std::fma(p, y, 1.0 / 24.0);
    p = std::fma(p, y, 1.0 / 6.0);
    p = std::fma(p, y, 0.5);
    p = std::fma(p, y, 1.0);
    p = std::fma(p, y, 1.0);
    return std::ldexp(p, (int)i);
}
inline float RefExp2(float t) { return (float)Exp2Double((double)t); }

// log2(x) in double for x > 0: x = m * 2^e with m in [sqrt(1/2), sqrt(2)), ln m = 2 atanh(s), s = (m - 1) / (m + 1), |s| <= 0.1716,
// odd series to s^25 (truncation < 1e-17 relative)
inline double Log2Double(double x) {
    int e;
    double m = std::frexp(x, &e); // [0.5, 1)
    if (m < 0.707106781186547524400844362105) {
        m *= 2.0;
        e -= 1;
    }
    const double s = (m - 1.0) / (m + 1.0);
    const double z = s * s;
    double p = 1.0 / 25.0;
    p = std::fma(p, z, 1.0 / 23.0);
    p = std::fma(p, z, 1.0 / 21.0);
    p = std::fma(p, z, 1.0 / 19.0);
    p = std::fma(p, z, 1.0 / 17.0);
    p = std::fma(p, z, 1.0 / 15.0);
    p = std::fma(p, z, 1.0 / 13.0);
    p = std::fma(p, z, 1.0 / 11.0);
    p = std::fma(p, z, 1.0 / 9.0);
    p = std::fma(p, z, 1.0 / 7.0);
    p = std::fma(p, z, 1.0 / 5.0);
    p = std::fma(p, z, 1.0 / 3.0);
    p = std::fma(p, z, 1.0);
    const double lnm = 2.0 * s * p;
    return std::fma(lnm, 1.44269504088896340735992468100, (double)e);
}
inline float RefLog2(float x) { return (float)Log2Double((double)x); }

} // namespace hwref
