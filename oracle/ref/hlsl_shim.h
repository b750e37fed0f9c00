// ORACLE/_ref -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// The HLSL vocabulary the NRD compute shaders are written in, as C++: vectors with swizzles, matrices, the intrinsic functions, typed textures and samplers.
// With it the reference's OWN shader text (/root/reference/Shaders/Source/*.cs.hlsl and everything they include) is compiled for the host and becomes
// oracle/_ref/libnrdref.so (recipe: oracle/ref/Makefile + hlsl2cpp.py); nothing of the reference is copied into this repository. That library is what pins
// the hand-written restatement (oracle/*.cpp) -- and through it the HIP kernels -- to the reference: tests/test_ref_parity.py runs every pass of a
// frame through both on identical inputs.
//
// Arithmetic: plain IEEE-754 binary32 -- no contraction (-ffp-contract=off), true division, correctly rounded sqrt, libm exp2f / log2f / atanf / ... .
// That is "the HLSL math on an IEEE machine"; a GPU's rcp / exp2 / log2 are approximations of it.
//
// Texture-unit semantics (D3D11 functional spec): Load / operator[] outside the resource return 0, stores outside are dropped, SampleLevel with a
// point sampler = the texel that contains uv, with a linear sampler = the fp32 bilinear blend of the 2x2 footprint around uv * size - 0.5, Gather* =
// that footprint's texels in (0,1) (1,1) (1,0) (0,0) order, clamp addressing. Typed stores convert like oracle/tex.h (fp16 round to nearest even,
// UNORM floor(x * max + 0.5)): hlsl_rt.cpp calls the very same codecs, so a difference between the two oracles is never a storage difference.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <type_traits>

namespace hlsl {

typedef uint32_t uint;

template <class T, int N> struct vec;
template <class T, int N, int... I> struct swz;

// ------------------------------------------------------------------------------------------------ traits
template <class X> struct traits { static constexpr int n = 0; typedef void S; };
template <> struct traits<float> { static constexpr int n = 1; typedef float S; static constexpr bool scalar = true; };
template <> struct traits<double> { static constexpr int n = 1; typedef float S; static constexpr bool scalar = true; }; // (an unsuffixed literal that escaped)
template <> struct traits<int> { static constexpr int n = 1; typedef int S; static constexpr bool scalar = true; };
template <> struct traits<uint> { static constexpr int n = 1; typedef uint S; static constexpr bool scalar = true; };
template <> struct traits<bool> { static constexpr int n = 1; typedef bool S; static constexpr bool scalar = true; };
template <> struct traits<long> { static constexpr int n = 1; typedef int S; static constexpr bool scalar = true; };
template <> struct traits<unsigned long> { static constexpr int n = 1; typedef uint S; static constexpr bool scalar = true; };
template <class T, int N> struct traits<vec<T, N>> { static constexpr int n = N; typedef T S; static constexpr bool scalar = false; };
template <class T, int N, int... I> struct traits<swz<T, N, I...>> { static constexpr int n = (int)sizeof...(I); typedef T S; static constexpr bool scalar = false; };

template <class X> constexpr bool is_hlsl = traits<std::decay_t<X>>::n > 0;
template <class X> constexpr int n_of = traits<std::decay_t<X>>::n;
template <class X> using s_of = typename traits<std::decay_t<X>>::S;
template <class X> constexpr bool is_vec = is_hlsl<X> && !std::is_arithmetic<std::decay_t<X>>::value;

// arithmetic promotion of element types: float wins, then uint, then int (bool counts as int)
template <class A, class B> struct promote { typedef std::conditional_t<std::is_same<A, float>::value || std::is_same<B, float>::value, float, std::conditional_t<std::is_same<A, uint>::value || std::is_same<B, uint>::value, uint, int>> type; };
template <class A, class B> using promote_t = typename promote<A, B>::type;
template <class S, int N> using res_t = std::conditional_t<N == 1, S, vec<S, N>>;

// component access with scalar broadcast
template <class X, std::enable_if_t<std::is_arithmetic<X>::value, int> = 0> inline X comp(const X& x, int) { return x; }
template <class T, int N> inline T comp(const vec<T, N>& v, int i) { return v.d[N == 1 ? 0 : i]; }
template <class T, int N, int... I> inline T comp(const swz<T, N, I...>& s, int i) { return s.get(i); }

// size of the result of a component-wise operation: scalars broadcast, vectors must agree (HLSL would truncate with a warning: refuse instead)
template <class A, class B> constexpr int common_n() {
    constexpr int a = n_of<A>, b = n_of<B>;
    static_assert(a == b || a == 1 || b == 1, "component-wise operation on vectors of different sizes");
    return a > b ? a : b;
}

// ------------------------------------------------------------------------------------------------ swizzles
template <class T, int N, int... I> struct swz {
    T d[N];
    static constexpr int K = (int)sizeof...(I);
    T get(int k) const {
        constexpr int idx[] = {I...};
        return d[idx[k]];
    }
    void set(const vec<T, K>& v) {
        constexpr int idx[] = {I...};
        for (int k = 0; k < K; k++)
            d[idx[k]] = v.d[k];
    }
    template <class A, std::enable_if_t<is_hlsl<A>, int> = 0> swz& operator=(const A& a) {
        set(vec<T, K>(a));
        return *this;
    }
    swz& operator=(const swz& o) {
        set(vec<T, K>(o));
        return *this;
    }
    template <class A> swz& operator+=(const A& a) { return *this = vec<T, K>(*this) + a; }
    template <class A> swz& operator-=(const A& a) { return *this = vec<T, K>(*this) - a; }
    template <class A> swz& operator*=(const A& a) { return *this = vec<T, K>(*this) * a; }
    template <class A> swz& operator/=(const A& a) { return *this = vec<T, K>(*this) / a; }
    template <class A> swz& operator&=(const A& a) { return *this = vec<T, K>(*this) & a; }
    template <class A> swz& operator|=(const A& a) { return *this = vec<T, K>(*this) | a; }
    template <class A> swz& operator>>=(const A& a) { return *this = vec<T, K>(*this) >> a; }
    template <class A> swz& operator<<=(const A& a) { return *this = vec<T, K>(*this) << a; }
};

#define HLSL_SWZ2(T, N, a, b, A, B) swz<T, N, A, B> a##b;
#define HLSL_SWZ3(T, N, a, b, c, A, B, C) swz<T, N, A, B, C> a##b##c;
#define HLSL_SWZ4(T, N, a, b, c, e, A, B, C, E) swz<T, N, A, B, C, E> a##b##c##e;

// all 2-, 3- and 4-component swizzles over the first 2 / 3 / 4 components
#define HLSL_SW2_OF2(T, N, M) M(T, N, x, x, 0, 0) M(T, N, x, y, 0, 1) M(T, N, y, x, 1, 0) M(T, N, y, y, 1, 1)
#define HLSL_SW2_ADD3(T, N, M) M(T, N, x, z, 0, 2) M(T, N, y, z, 1, 2) M(T, N, z, x, 2, 0) M(T, N, z, y, 2, 1) M(T, N, z, z, 2, 2)
#define HLSL_SW2_ADD4(T, N, M) M(T, N, x, w, 0, 3) M(T, N, y, w, 1, 3) M(T, N, z, w, 2, 3) M(T, N, w, x, 3, 0) M(T, N, w, y, 3, 1) M(T, N, w, z, 3, 2) M(T, N, w, w, 3, 3)

// (the 3- and 4-component swizzle members are generated: see hlsl_swizzles.inc, written by hlsl2cpp.py --swizzles)

template <class T, int N> struct vec_base;

template <class T> struct vec<T, 1> {
    union {
        T d[1];
        T x;
        T r;
    };
    vec() : d{T(0)} {}
    template <class A, std::enable_if_t<is_hlsl<A>, int> = 0> explicit vec(const A& a) { d[0] = (T)comp(a, 0); }
    operator T() const { return x; }
};

#define HLSL_VEC_COMMON(N)                                                                                                                          \
    vec() {                                                                                                                                         \
        for (int i = 0; i < N; i++)                                                                                                                 \
            d[i] = T(0);                                                                                                                            \
    }                                                                                                                                               \
    vec(const vec& o) {                                                                                                                             \
        for (int i = 0; i < N; i++)                                                                                                                 \
            d[i] = o.d[i];                                                                                                                          \
    }                                                                                                                                               \
    vec& operator=(const vec& o) {                                                                                                                  \
        for (int i = 0; i < N; i++)                                                                                                                 \
            d[i] = o.d[i];                                                                                                                          \
        return *this;                                                                                                                               \
    }                                                                                                                                               \
    /* implicit: a scalar (broadcast) or a vector of the same size (element conversion) */                                                         \
    template <class A, std::enable_if_t<is_hlsl<A> && (n_of<A> == 1 || n_of<A> == N), int> = 0> vec(const A& a) {                                    \
        for (int i = 0; i < N; i++)                                                                                                                 \
            d[i] = (T)comp(a, i);                                                                                                                   \
    }                                                                                                                                               \
    /* explicit: truncation of a longer vector, ( float3 )v4 */                                                                                    \
    template <class A, std::enable_if_t<is_hlsl<A> && (n_of<A> > N), int> = 0> explicit vec(const A& a) {                                            \
        for (int i = 0; i < N; i++)                                                                                                                 \
            d[i] = (T)comp(a, i);                                                                                                                   \
    }                                                                                                                                               \
    /* float4( v.xyz, 1 ), float3( a, b, c ), float4( uv, zw ) ... */                                                                              \
    template <class A0, class A1, class... A, std::enable_if_t<is_hlsl<A0> && is_hlsl<A1> && (is_hlsl<A> && ...) && (n_of<A0> + n_of<A1> + (n_of<A> + ... + 0)) == N, int> = 0> \
    vec(const A0& a0, const A1& a1, const A&... a) {                                                                                               \
        int k = 0;                                                                                                                                  \
        append(a0, k);                                                                                                                              \
        append(a1, k);                                                                                                                              \
        (append(a, k), ...);                                                                                                                        \
    }                                                                                                                                               \
    template <class A> void append(const A& a, int& k) {                                                                                            \
        for (int i = 0; i < n_of<A>; i++)                                                                                                           \
            d[k++] = (T)comp(a, i);                                                                                                                 \
    }                                                                                                                                               \
    T& operator[](int i) { return d[i]; }                                                                                                           \
    const T& operator[](int i) const { return d[i]; }                                                                                               \
    template <class A> vec& operator+=(const A& a) { return *this = vec(*this + a); }                                                                \
    template <class A> vec& operator-=(const A& a) { return *this = vec(*this - a); }                                                                \
    template <class A> vec& operator*=(const A& a) { return *this = vec(*this * a); }                                                                \
    template <class A> vec& operator/=(const A& a) { return *this = vec(*this / a); }                                                                \
    template <class A> vec& operator&=(const A& a) { return *this = vec(*this & a); }                                                                \
    template <class A> vec& operator|=(const A& a) { return *this = vec(*this | a); }                                                                \
    template <class A> vec& operator>>=(const A& a) { return *this = vec(*this >> a); }                                                              \
    template <class A> vec& operator<<=(const A& a) { return *this = vec(*this << a); }

template <class T> struct vec<T, 2> {
    union {
        T d[2];
        struct {
            T x, y;
        };
        struct {
            T r, g;
        };
        HLSL_SW2_OF2(T, 2, HLSL_SWZ2)
#define HLSL_SWIZZLES_2
#include "hlsl_swizzles.inc"
#undef HLSL_SWIZZLES_2
    };
    HLSL_VEC_COMMON(2)
};
template <class T> struct vec<T, 3> {
    union {
        T d[3];
        struct {
            T x, y, z;
        };
        struct {
            T r, g, b;
        };
        HLSL_SW2_OF2(T, 3, HLSL_SWZ2)
        HLSL_SW2_ADD3(T, 3, HLSL_SWZ2)
#define HLSL_SWIZZLES_3
#include "hlsl_swizzles.inc"
#undef HLSL_SWIZZLES_3
    };
    HLSL_VEC_COMMON(3)
};
template <class T> struct vec<T, 4> {
    union {
        T d[4];
        struct {
            T x, y, z, w;
        };
        struct {
            T r, g, b, a;
        };
        HLSL_SW2_OF2(T, 4, HLSL_SWZ2)
        HLSL_SW2_ADD3(T, 4, HLSL_SWZ2)
        HLSL_SW2_ADD4(T, 4, HLSL_SWZ2)
#define HLSL_SWIZZLES_4
#include "hlsl_swizzles.inc"
#undef HLSL_SWIZZLES_4
    };
    HLSL_VEC_COMMON(4)
};

typedef vec<float, 1> float1;
typedef vec<float, 2> float2;
typedef vec<float, 3> float3;
typedef vec<float, 4> float4;
typedef vec<int, 2> int2;
typedef vec<int, 3> int3;
typedef vec<int, 4> int4;
typedef vec<uint, 2> uint2;
typedef vec<uint, 3> uint3;
typedef vec<uint, 4> uint4;
typedef vec<bool, 2> bool2;
typedef vec<bool, 3> bool3;
typedef vec<bool, 4> bool4;

// ------------------------------------------------------------------------------------------------ operators
#define HLSL_ENABLE_BIN(A, B) std::enable_if_t<(is_vec<A> || is_vec<B>) && is_hlsl<A> && is_hlsl<B>, int> = 0

#define HLSL_ARITH(op)                                                                                              \
    template <class A, class B, HLSL_ENABLE_BIN(A, B)> inline auto operator op(const A& a, const B& b) {            \
        typedef promote_t<s_of<A>, s_of<B>> S;                                                                      \
        constexpr int N = common_n<A, B>();                                                                         \
        vec<S, N> r;                                                                                                \
        for (int i = 0; i < N; i++)                                                                                 \
            r.d[i] = S(S(comp(a, i)) op S(comp(b, i)));                                                             \
        return res_t<S, N>(r);                                                                                      \
    }
HLSL_ARITH(+) HLSL_ARITH(-) HLSL_ARITH(*) HLSL_ARITH(/)

#define HLSL_INTOP(op)                                                                                              \
    template <class A, class B, HLSL_ENABLE_BIN(A, B)> inline auto operator op(const A& a, const B& b) {            \
        typedef promote_t<s_of<A>, s_of<B>> S;                                                                      \
        static_assert(!std::is_same<S, float>::value, "integer operator on floats");                               \
        constexpr int N = common_n<A, B>();                                                                         \
        vec<S, N> r;                                                                                                \
        for (int i = 0; i < N; i++)                                                                                 \
            r.d[i] = S(S(comp(a, i)) op S(comp(b, i)));                                                             \
        return res_t<S, N>(r);                                                                                      \
    }
HLSL_INTOP(&) HLSL_INTOP(|) HLSL_INTOP(^) HLSL_INTOP(%)
// shifts keep the type of the left operand
#define HLSL_SHIFT(op)                                                                                              \
    template <class A, class B, HLSL_ENABLE_BIN(A, B)> inline auto operator op(const A& a, const B& b) {            \
        typedef promote_t<s_of<A>, s_of<A>> S;                                                                      \
        constexpr int N = common_n<A, B>();                                                                         \
        vec<S, N> r;                                                                                                \
        for (int i = 0; i < N; i++)                                                                                 \
            r.d[i] = S(S(comp(a, i)) op(int) comp(b, i));                                                           \
        return res_t<S, N>(r);                                                                                      \
    }
HLSL_SHIFT(>>) HLSL_SHIFT(<<)

#define HLSL_CMP(op)                                                                                                \
    template <class A, class B, HLSL_ENABLE_BIN(A, B)> inline auto operator op(const A& a, const B& b) {            \
        typedef promote_t<s_of<A>, s_of<B>> S;                                                                      \
        constexpr int N = common_n<A, B>();                                                                         \
        vec<bool, N> r;                                                                                             \
        for (int i = 0; i < N; i++)                                                                                 \
            r.d[i] = S(comp(a, i)) op S(comp(b, i));                                                                \
        return res_t<bool, N>(r);                                                                                   \
    }
HLSL_CMP(<) HLSL_CMP(>) HLSL_CMP(<=) HLSL_CMP(>=) HLSL_CMP(==) HLSL_CMP(!=)

#define HLSL_LOGIC(op)                                                                                              \
    template <class A, class B, HLSL_ENABLE_BIN(A, B)> inline auto operator op(const A& a, const B& b) {            \
        constexpr int N = common_n<A, B>();                                                                         \
        vec<bool, N> r;                                                                                             \
        for (int i = 0; i < N; i++)                                                                                 \
            r.d[i] = (comp(a, i) != 0) op(comp(b, i) != 0);                                                         \
        return res_t<bool, N>(r);                                                                                   \
    }
HLSL_LOGIC(&&) HLSL_LOGIC(||)

template <class A, std::enable_if_t<is_vec<A>, int> = 0> inline auto operator-(const A& a) {
    constexpr int N = n_of<A>;
    vec<s_of<A>, N> r;
    for (int i = 0; i < N; i++)
        r.d[i] = -comp(a, i);
    return res_t<s_of<A>, N>(r);
}
template <class A, std::enable_if_t<is_vec<A>, int> = 0> inline auto operator+(const A& a) { return vec<s_of<A>, n_of<A>>(a); }
template <class A, std::enable_if_t<is_vec<A>, int> = 0> inline auto operator!(const A& a) {
    constexpr int N = n_of<A>;
    vec<bool, N> r;
    for (int i = 0; i < N; i++)
        r.d[i] = !(comp(a, i) != 0);
    return res_t<bool, N>(r);
}
template <class A, std::enable_if_t<is_vec<A>, int> = 0> inline auto operator~(const A& a) {
    constexpr int N = n_of<A>;
    vec<s_of<A>, N> r;
    for (int i = 0; i < N; i++)
        r.d[i] = ~comp(a, i);
    return res_t<s_of<A>, N>(r);
}

// ------------------------------------------------------------------------------------------------ scalar kernels of the intrinsics (IEEE binary32)
namespace k {
inline float f(float x) { return x; }
inline float f(int x) { return (float)x; }
inline float f(uint x) { return (float)x; }
inline float f(bool x) { return x ? 1.0f : 0.0f; }
inline float f(double x) { return (float)x; }
inline float min_(float a, float b) { return fminf(a, b); } // a NaN operand loses (D3D min / max)
inline float max_(float a, float b) { return fmaxf(a, b); }
inline int min_(int a, int b) { return a < b ? a : b; }
inline int max_(int a, int b) { return a > b ? a : b; }
inline uint min_(uint a, uint b) { return a < b ? a : b; }
inline uint max_(uint a, uint b) { return a > b ? a : b; }
inline float abs_(float a) { return fabsf(a); }
inline int abs_(int a) { return a < 0 ? -a : a; }
inline uint abs_(uint a) { return a; }
inline float saturate_(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); } // NaN -> 0
inline float rcp_(float x) { return 1.0f / x; }
inline float rsqrt_(float x) { return 1.0f / sqrtf(x); }
inline float frac_(float x) { return x - floorf(x); }
inline float pow_(float x, float y) { return x <= 0.0f ? 0.0f : exp2f(y * log2f(x)); } // = exp2( y * log2( x ) ), what the compilers emit; 0 for x <= 0 (oracle/hlsl.h)
inline float sign_(float x) { return x > 0.0f ? 1.0f : x < 0.0f ? -1.0f : 0.0f; }
inline float step_(float e, float x) { return x >= e ? 1.0f : 0.0f; }
inline float lerp_(float a, float b, float t) { return a + (b - a) * t; }
inline float round_(float x) { return nearbyintf(x); } // round half to even
} // namespace k

// unary, float-valued (integers are converted first)
#define HLSL_UNARY_F(name, expr)                                                                      \
    template <class A, std::enable_if_t<is_hlsl<A>, int> = 0> inline auto name(const A& a) {         \
        constexpr int N = n_of<A>;                                                                    \
        vec<float, N> r;                                                                              \
        for (int i = 0; i < N; i++) {                                                                 \
            const float x = k::f(comp(a, i));                                                         \
            r.d[i] = (expr);                                                                          \
        }                                                                                             \
        return res_t<float, N>(r);                                                                    \
    }
HLSL_UNARY_F(saturate, k::saturate_(x))
HLSL_UNARY_F(floor, floorf(x))
HLSL_UNARY_F(ceil, ceilf(x))
HLSL_UNARY_F(trunc, truncf(x))
HLSL_UNARY_F(round, k::round_(x))
HLSL_UNARY_F(frac, k::frac_(x))
HLSL_UNARY_F(sqrt, sqrtf(x))
HLSL_UNARY_F(rsqrt, k::rsqrt_(x))
HLSL_UNARY_F(rcp, k::rcp_(x))
HLSL_UNARY_F(exp, expf(x))
HLSL_UNARY_F(exp2, exp2f(x))
HLSL_UNARY_F(log, logf(x))
HLSL_UNARY_F(log2, log2f(x))
HLSL_UNARY_F(sin, sinf(x))
HLSL_UNARY_F(cos, cosf(x))
HLSL_UNARY_F(tan, tanf(x))
HLSL_UNARY_F(acos, acosf(x))
HLSL_UNARY_F(asin, asinf(x))
HLSL_UNARY_F(atan, atanf(x))
HLSL_UNARY_F(sign, k::sign_(x))
HLSL_UNARY_F(radians, x * 0.01745329251994329577f)
HLSL_UNARY_F(degrees, x * 57.2957795130823208768f)

// unary, type-preserving
template <class A, std::enable_if_t<is_hlsl<A>, int> = 0> inline auto abs(const A& a) {
    constexpr int N = n_of<A>;
    typedef promote_t<s_of<A>, s_of<A>> S;
    vec<S, N> r;
    for (int i = 0; i < N; i++)
        r.d[i] = k::abs_(S(comp(a, i)));
    return res_t<S, N>(r);
}
template <class A, std::enable_if_t<is_hlsl<A>, int> = 0> inline auto isnan(const A& a) {
    constexpr int N = n_of<A>;
    vec<bool, N> r;
    for (int i = 0; i < N; i++)
        r.d[i] = std::isnan(k::f(comp(a, i)));
    return res_t<bool, N>(r);
}
template <class A, std::enable_if_t<is_hlsl<A>, int> = 0> inline auto isinf(const A& a) {
    constexpr int N = n_of<A>;
    vec<bool, N> r;
    for (int i = 0; i < N; i++)
        r.d[i] = std::isinf(k::f(comp(a, i)));
    return res_t<bool, N>(r);
}
template <class A, std::enable_if_t<is_hlsl<A>, int> = 0> inline bool any(const A& a) {
    for (int i = 0; i < n_of<A>; i++)
        if (comp(a, i) != 0)
            return true;
    return false;
}
template <class A, std::enable_if_t<is_hlsl<A>, int> = 0> inline bool all(const A& a) {
    for (int i = 0; i < n_of<A>; i++)
        if (!(comp(a, i) != 0))
            return false;
    return true;
}

// binary / ternary with promotion
#define HLSL_BINARY(name, expr)                                                                                       \
    template <class A, class B, std::enable_if_t<is_hlsl<A> && is_hlsl<B>, int> = 0> inline auto name(const A& a, const B& b) { \
        typedef promote_t<s_of<A>, s_of<B>> S;                                                                        \
        constexpr int N = common_n<A, B>();                                                                           \
        vec<S, N> r;                                                                                                  \
        for (int i = 0; i < N; i++) {                                                                                 \
            const S x = S(comp(a, i)), y = S(comp(b, i));                                                             \
            r.d[i] = (expr);                                                                                          \
        }                                                                                                             \
        return res_t<S, N>(r);                                                                                        \
    }
HLSL_BINARY(min, k::min_(x, y))
HLSL_BINARY(max, k::max_(x, y))
#define HLSL_BINARY_F(name, expr)                                                                                     \
    template <class A, class B, std::enable_if_t<is_hlsl<A> && is_hlsl<B>, int> = 0> inline auto name(const A& a, const B& b) { \
        constexpr int N = common_n<A, B>();                                                                           \
        vec<float, N> r;                                                                                              \
        for (int i = 0; i < N; i++) {                                                                                 \
            const float x = k::f(comp(a, i)), y = k::f(comp(b, i));                                                   \
            r.d[i] = (expr);                                                                                          \
        }                                                                                                             \
        return res_t<float, N>(r);                                                                                    \
    }
HLSL_BINARY_F(pow, k::pow_(x, y))
HLSL_BINARY_F(step, k::step_(x, y))
HLSL_BINARY_F(fmod, fmodf(x, y))
HLSL_BINARY_F(atan2, atan2f(x, y))
HLSL_BINARY_F(ldexp, ldexpf(x, (int)y))

template <class A, class B, class C, std::enable_if_t<is_hlsl<A> && is_hlsl<B> && is_hlsl<C>, int> = 0> inline auto lerp(const A& a, const B& b, const C& t) {
    constexpr int N = common_n<vec<float, common_n<A, B>()>, C>();
    vec<float, N> r;
    for (int i = 0; i < N; i++)
        r.d[i] = k::lerp_(k::f(comp(a, i)), k::f(comp(b, i)), k::f(comp(t, i)));
    return res_t<float, N>(r);
}
template <class A, class B, class C, std::enable_if_t<is_hlsl<A> && is_hlsl<B> && is_hlsl<C>, int> = 0> inline auto clamp(const A& a, const B& lo, const C& hi) {
    typedef promote_t<promote_t<s_of<A>, s_of<B>>, s_of<C>> S;
    constexpr int N = common_n<vec<S, common_n<A, B>()>, C>();
    vec<S, N> r;
    for (int i = 0; i < N; i++)
        r.d[i] = k::min_(k::max_(S(comp(a, i)), S(comp(lo, i))), S(comp(hi, i)));
    return res_t<S, N>(r);
}
template <class A, class B, class C, std::enable_if_t<is_hlsl<A> && is_hlsl<B> && is_hlsl<C>, int> = 0> inline auto mad(const A& a, const B& b, const C& c) { return a * b + c; }

// geometric
template <class A, class B, std::enable_if_t<is_hlsl<A> && is_hlsl<B>, int> = 0> inline auto dot(const A& a, const B& b) {
    typedef promote_t<s_of<A>, s_of<B>> S;
    constexpr int N = common_n<A, B>();
    S r = S(comp(a, 0)) * S(comp(b, 0));
    for (int i = 1; i < N; i++)
        r = r + S(comp(a, i)) * S(comp(b, i));
    return r;
}
template <class A, std::enable_if_t<is_hlsl<A>, int> = 0> inline float length(const A& a) { return sqrtf(dot(a, a)); }
template <class A, class B, std::enable_if_t<is_hlsl<A> && is_hlsl<B>, int> = 0> inline float distance(const A& a, const B& b) { return length(a - b); }
template <class A, std::enable_if_t<is_vec<A>, int> = 0> inline auto normalize(const A& a) { return a * k::rsqrt_(dot(a, a)); }
inline float3 cross(const float3& a, const float3& b) { return float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
template <class A, class B, std::enable_if_t<is_vec<A> && is_vec<B>, int> = 0> inline auto reflect(const A& i, const B& n) { return i - 2.0f * n * dot(i, n); }

// bit casts and integer intrinsics
inline uint asuint(float x) {
    uint u;
    memcpy(&u, &x, 4);
    return u;
}
inline uint asuint(uint x) { return x; }
inline uint asuint(int x) { return (uint)x; }
inline float asfloat(uint u) {
    float x;
    memcpy(&x, &u, 4);
    return x;
}
inline float asfloat(int u) { return asfloat((uint)u); }
inline float asfloat(float x) { return x; }
inline int asint(float x) { return (int)asuint(x); }
template <class T, int N> inline vec<uint, N> asuint(const vec<T, N>& v) {
    vec<uint, N> r;
    for (int i = 0; i < N; i++)
        r.d[i] = asuint(v.d[i]);
    return r;
}
template <class T, int N> inline vec<float, N> asfloat(const vec<T, N>& v) {
    vec<float, N> r;
    for (int i = 0; i < N; i++)
        r.d[i] = asfloat(v.d[i]);
    return r;
}
inline uint countbits(uint x) { return (uint)__builtin_popcount(x); }
inline uint reversebits(uint x) {
    uint r = 0;
    for (int i = 0; i < 32; i++)
        r |= ((x >> i) & 1u) << (31 - i);
    return r;
}
inline uint firstbithigh(uint x) { return x ? 31u - (uint)__builtin_clz(x) : 0xFFFFFFFFu; }
inline uint firstbitlow(uint x) { return x ? (uint)__builtin_ctz(x) : 0xFFFFFFFFu; }
uint f32tof16(float f);  // hlsl_rt.cpp: the codec of oracle/tex.h (round to nearest even, denormals kept)
float f16tof32(uint h);
template <int N> inline vec<uint, N> f32tof16(const vec<float, N>& v) {
    vec<uint, N> r;
    for (int i = 0; i < N; i++)
        r.d[i] = f32tof16(v.d[i]);
    return r;
}
template <int N> inline vec<float, N> f16tof32(const vec<uint, N>& v) {
    vec<float, N> r;
    for (int i = 0; i < N; i++)
        r.d[i] = f16tof32(v.d[i]);
    return r;
}

// `s.x` where s may be a scalar (REBLUR_FAST_TYPE, REBLUR_TYPE in the occlusion family ...): hlsl2cpp.py rewrites IDENT.x to swz_x1( IDENT ) -- an lvalue for lvalues
template <class A, std::enable_if_t<std::is_arithmetic<A>::value, int> = 0> inline A& swz_x1(A& a) { return a; }
template <class A, std::enable_if_t<std::is_arithmetic<A>::value, int> = 0> inline A swz_x1(const A& a) { return a; }
template <class T, int N> inline T& swz_x1(vec<T, N>& v) { return v.d[0]; }
template <class T, int N> inline T swz_x1(const vec<T, N>& v) { return v.d[0]; }
template <class S> inline auto swz_x1(const S& s) -> decltype(s.x) { return s.x; } // anything else that has a member x
// `s.xx`, `1.0.xxx` on a scalar (rewritten to these calls by hlsl2cpp.py; vectors go through their swizzle members)
template <class A, std::enable_if_t<is_hlsl<A>, int> = 0> inline auto swz_x2(const A& a) { return vec<s_of<A>, 2>(comp(a, 0), comp(a, 0)); }
template <class A, std::enable_if_t<is_hlsl<A>, int> = 0> inline auto swz_x3(const A& a) { return vec<s_of<A>, 3>(comp(a, 0), comp(a, 0), comp(a, 0)); }
template <class A, std::enable_if_t<is_hlsl<A>, int> = 0> inline auto swz_x4(const A& a) { return vec<s_of<A>, 4>(comp(a, 0), comp(a, 0), comp(a, 0), comp(a, 0)); }

// ------------------------------------------------------------------------------------------------ matrices (row vectors r[i]; M[i] is row i as in HLSL)
template <int R, int C> struct mat {
    vec<float, C> r[R];
    mat() {}
    template <class... A, std::enable_if_t<sizeof...(A) == R && (is_vec<A> && ...), int> = 0> mat(const A&... rows) {
        int i = 0;
        ((r[i++] = vec<float, C>(rows)), ...);
    }
    template <class... A, std::enable_if_t<sizeof...(A) == R * C && (std::is_arithmetic<A>::value && ...), int> = 0> mat(const A&... e) {
        const float v[] = {(float)e...};
        for (int i = 0; i < R; i++)
            for (int j = 0; j < C; j++)
                r[i].d[j] = v[i * C + j];
    }
    template <int R2, int C2, std::enable_if_t<(R2 > R || C2 > C) && R2 >= R && C2 >= C, int> = 0> explicit mat(const mat<R2, C2>& m) { // ( float3x3 )M
        for (int i = 0; i < R; i++)
            for (int j = 0; j < C; j++)
                r[i].d[j] = m.r[i].d[j];
    }
    vec<float, C>& operator[](int i) { return r[i]; }
    const vec<float, C>& operator[](int i) const { return r[i]; }
};
typedef mat<2, 2> float2x2;
typedef mat<2, 3> float2x3;
typedef mat<3, 3> float3x3;
typedef mat<3, 4> float3x4;
typedef mat<4, 4> float4x4;

template <int R, int C, class V, std::enable_if_t<is_vec<V> && n_of<V> == C, int> = 0> inline vec<float, R> mul(const mat<R, C>& m, const V& v) { // M * column vector
    vec<float, R> o;
    const vec<float, C> x(v);
    for (int i = 0; i < R; i++)
        o.d[i] = dot(m.r[i], x);
    return o;
}
template <int R, int C, class V, std::enable_if_t<is_vec<V> && n_of<V> == R, int> = 0> inline vec<float, C> mul(const V& v, const mat<R, C>& m) { // row vector * M
    vec<float, C> o;
    const vec<float, R> x(v);
    for (int j = 0; j < C; j++) {
        float s = x.d[0] * m.r[0].d[j];
        for (int i = 1; i < R; i++)
            s = s + x.d[i] * m.r[i].d[j];
        o.d[j] = s;
    }
    return o;
}
template <int R, int K, int C> inline mat<R, C> mul(const mat<R, K>& a, const mat<K, C>& b) {
    mat<R, C> o;
    for (int i = 0; i < R; i++)
        for (int j = 0; j < C; j++) {
            float s = a.r[i].d[0] * b.r[0].d[j];
            for (int k2 = 1; k2 < K; k2++)
                s = s + a.r[i].d[k2] * b.r[k2].d[j];
            o.r[i].d[j] = s;
        }
    return o;
}
template <int R, int C> inline mat<C, R> transpose(const mat<R, C>& m) {
    mat<C, R> o;
    for (int i = 0; i < R; i++)
        for (int j = 0; j < C; j++)
            o.r[j].d[i] = m.r[i].d[j];
    return o;
}

} // namespace hlsl

// ================================================================================================ runtime: planes, textures, dispatch
namespace hlsl_rt {

struct Plane { // = OraclePlane of oracle/oracle_api.cpp
    void* data;
    uint32_t rowPitchBytes;
    uint32_t format;
    uint16_t width, height;
};

// texel codecs of oracle/tex.h behind plain functions (hlsl_rt.cpp)
void Fetch(const Plane& p, int x, int y, float out[4]); // no bounds check
uint32_t FetchUint(const Plane& p, int x, int y);
void Store(const Plane& p, int x, int y, const float v[4]); // dropped outside
void StoreUint(const Plane& p, int x, int y, uint32_t v);
bool IsUintFormat(uint32_t format);

struct ThreadIds {
    hlsl::uint3 groupThreadId, groupId, dispatchThreadId;
    hlsl::uint groupIndex;
};

void Barrier(); // GroupMemoryBarrierWithGroupSync: yields this thread's fiber until every live thread of the group has arrived

enum CbKind { CB_SCALAR, CB_VEC2, CB_VEC3, CB_VEC4, CB_MAT4 };
struct ConstantReg {
    void* ptr;
    CbKind kind;
};
struct ResourceReg {
    Plane* plane;
    bool output;
    int index;
    const char* name;
};
struct ShaderTable; // per translation unit
ShaderTable* NewTable();
void SetGroupSharedClear(ShaderTable* t, void (*clear)());
void AddConstant(ShaderTable* t, void* ptr, CbKind kind);
void AddResource(ShaderTable* t, Plane* plane, bool output, int index, const char* name);
void RegisterShader(ShaderTable* t, const char* fileName, int groupX, int groupY, bool usesBarrier, void (*thunk)(const ThreadIds&));

template <class T> struct cb_kind;
template <> struct cb_kind<float> { static constexpr CbKind v = CB_SCALAR; };
template <> struct cb_kind<int> { static constexpr CbKind v = CB_SCALAR; };
template <> struct cb_kind<hlsl::uint> { static constexpr CbKind v = CB_SCALAR; };
template <class T> struct cb_kind<hlsl::vec<T, 2>> { static constexpr CbKind v = CB_VEC2; };
template <class T> struct cb_kind<hlsl::vec<T, 3>> { static constexpr CbKind v = CB_VEC3; };
template <class T> struct cb_kind<hlsl::vec<T, 4>> { static constexpr CbKind v = CB_VEC4; };
template <> struct cb_kind<hlsl::float4x4> { static constexpr CbKind v = CB_MAT4; };

struct ConstantAdder {
    template <class T> ConstantAdder(ShaderTable* t, T* p) { AddConstant(t, p, cb_kind<T>::v); }
};
struct ResourceAdder {
    ResourceAdder(ShaderTable* t, Plane* p, bool output, int index, const char* name) { AddResource(t, p, output, index, name); }
};
struct ShaderAdder {
    ShaderAdder(ShaderTable* t, const char* fileName, int gx, int gy, bool barrier, void (*thunk)(const ThreadIds&)) { RegisterShader(t, fileName, gx, gy, barrier, thunk); }
};

} // namespace hlsl_rt

namespace hlsl {

enum SamplerState { gNearestClamp = 0, gLinearClamp = 1 };

template <class T> struct texel_traits;
template <> struct texel_traits<float> { typedef float1 ret; static constexpr bool isUint = false; };
template <> struct texel_traits<float2> { typedef float2 ret; static constexpr bool isUint = false; };
template <> struct texel_traits<float3> { typedef float3 ret; static constexpr bool isUint = false; };
template <> struct texel_traits<float4> { typedef float4 ret; static constexpr bool isUint = false; };
template <> struct texel_traits<uint> { typedef vec<uint, 1> ret; static constexpr bool isUint = true; };
template <> struct texel_traits<uint2> { typedef uint2 ret; static constexpr bool isUint = true; };
template <> struct texel_traits<uint4> { typedef uint4 ret; static constexpr bool isUint = true; };

template <class T> struct TextureBase {
    hlsl_rt::Plane p;
    typedef typename texel_traits<T>::ret R;
    static constexpr int NC = n_of<R>;
    int W() const { return p.width; }
    int H() const { return p.height; }
    bool In(int x, int y) const { return (unsigned)x < p.width && (unsigned)y < p.height; }
    R FetchRaw(int x, int y) const {
        R r;
        if (texel_traits<T>::isUint) {
            r.d[0] = (s_of<R>)hlsl_rt::FetchUint(p, x, y); // single-channel uint formats only
        } else {
            float v[4];
            hlsl_rt::Fetch(p, x, y, v);
            for (int i = 0; i < NC; i++)
                r.d[i] = (s_of<R>)v[i];
        }
        return r;
    }
    R LoadTexel(int x, int y) const { return In(x, y) ? FetchRaw(x, y) : R(); }
    R FetchClamped(int x, int y) const { return FetchRaw(x < 0 ? 0 : x >= W() ? W() - 1 : x, y < 0 ? 0 : y >= H() ? H() - 1 : y); }
    void GetDimensions(float& w, float& h) const { w = (float)W(), h = (float)H(); }
    void GetDimensions(uint& w, uint& h) const { w = (uint)W(), h = (uint)H(); }
};

template <class T> struct Texture2D : TextureBase<T> {
    typedef typename TextureBase<T>::R R;
    template <class P, std::enable_if_t<is_vec<P> && n_of<P> == 2, int> = 0> R operator[](const P& pos) const { return this->LoadTexel((int)comp(pos, 0), (int)comp(pos, 1)); }
    template <class P, std::enable_if_t<is_vec<P> && n_of<P> == 3, int> = 0> R Load(const P& pos) const { return this->LoadTexel((int)comp(pos, 0), (int)comp(pos, 1)); }
    template <class P, class O, std::enable_if_t<is_vec<P> && n_of<P> == 3 && is_vec<O>, int> = 0> R Load(const P& pos, const O& off) const {
        return this->LoadTexel((int)comp(pos, 0) + (int)comp(off, 0), (int)comp(pos, 1) + (int)comp(off, 1));
    }
    // SampleLevel( sampler, uv, lod [, offset] ): single mip
    template <class U> R SampleLevel(SamplerState s, const U& uvIn, float, int2 off = int2(0, 0)) const {
        const float2 uv(uvIn);
        const float fx = uv.x * (float)this->W(), fy = uv.y * (float)this->H();
        if (s == gNearestClamp)
            return this->FetchClamped((int)floorf(fx) + off.x, (int)floorf(fy) + off.y);
        const float tx = fx - 0.5f, ty = fy - 0.5f;
        const float x0f = floorf(tx), y0f = floorf(ty);
        const float wx = tx - x0f, wy = ty - y0f;
        const int x0 = (int)x0f + off.x, y0 = (int)y0f + off.y;
        const vec<float, TextureBase<T>::NC> s00(this->FetchClamped(x0, y0)), s10(this->FetchClamped(x0 + 1, y0)), s01(this->FetchClamped(x0, y0 + 1)), s11(this->FetchClamped(x0 + 1, y0 + 1));
        const float w00 = (1.0f - wx) * (1.0f - wy), w10 = wx * (1.0f - wy), w01 = (1.0f - wx) * wy, w11 = wx * wy;
        return R(vec<float, TextureBase<T>::NC>(s00 * w00 + s10 * w10 + s01 * w01 + s11 * w11));
    }
    template <int CH, class U> vec<s_of<R>, 4> GatherChannel(const U& uvIn, int2 off) const {
        const float2 uv(uvIn);
        const int x0 = (int)floorf(uv.x * (float)this->W() - 0.5f) + off.x, y0 = (int)floorf(uv.y * (float)this->H() - 0.5f) + off.y;
        vec<s_of<R>, 4> r; // (0,1) (1,1) (1,0) (0,0)
        r.x = this->FetchClamped(x0, y0 + 1).d[CH];
        r.y = this->FetchClamped(x0 + 1, y0 + 1).d[CH];
        r.z = this->FetchClamped(x0 + 1, y0).d[CH];
        r.w = this->FetchClamped(x0, y0).d[CH];
        return r;
    }
    template <class U> auto GatherRed(SamplerState, const U& uv, int2 off = int2(0, 0)) const { return GatherChannel<0>(uv, off); }
    template <class U> auto GatherGreen(SamplerState, const U& uv, int2 off = int2(0, 0)) const { return GatherChannel < TextureBase<T>::NC >= 2 ? 1 : 0 > (uv, off); }
    template <class U> auto GatherBlue(SamplerState, const U& uv, int2 off = int2(0, 0)) const { return GatherChannel < TextureBase<T>::NC >= 3 ? 2 : 0 > (uv, off); }
    template <class U> auto GatherAlpha(SamplerState, const U& uv, int2 off = int2(0, 0)) const { return GatherChannel < TextureBase<T>::NC >= 4 ? 3 : 0 > (uv, off); }
};

template <class T> struct RWRef { // gOut[ pos ]: a store when assigned to, the texel's value otherwise
    const TextureBase<T>* t;
    int x, y;
    typedef typename TextureBase<T>::R R;
    operator R() const { return t->LoadTexel(x, y); }
    template <class V, std::enable_if_t<is_hlsl<V>, int> = 0> const RWRef& operator=(const V& v) const {
        if (texel_traits<T>::isUint) {
            hlsl_rt::StoreUint(t->p, x, y, (uint32_t)comp(v, 0));
        } else {
            float f[4] = {0, 0, 0, 0};
            for (int i = 0; i < TextureBase<T>::NC; i++)
                f[i] = k::f(comp(v, n_of<V> == 1 ? 0 : i));
            hlsl_rt::Store(t->p, x, y, f);
        }
        return *this;
    }
    const RWRef& operator=(const RWRef& o) const { return *this = R(o); }
};
template <class T> struct traits<RWRef<T>> : traits<typename texel_traits<T>::ret> {};
template <class T> inline auto comp(const RWRef<T>& r, int i) { return typename TextureBase<T>::R(r).d[TextureBase<T>::NC == 1 ? 0 : i]; }

template <class T> struct RWTexture2D : TextureBase<T> {
    template <class P, std::enable_if_t<is_vec<P> && n_of<P> == 2, int> = 0> RWRef<T> operator[](const P& pos) const { return RWRef<T>{this, (int)comp(pos, 0), (int)comp(pos, 1)}; }
};

inline void GroupMemoryBarrierWithGroupSync() { hlsl_rt::Barrier(); }
// The *_ClassifyTiles shaders order their group-shared atomics with GroupMemoryBarrier() alone (32 threads = one lock-step warp on the hardware they were
// written for): a thread that ran to completion before the others have started would read half-built counters. Here it is a full group barrier.
inline void GroupMemoryBarrier() { hlsl_rt::Barrier(); }
// group threads are fibers of one OS thread: the atomics on group-shared memory are plain read-modify-writes
template <class A, class B> inline void InterlockedAdd(A& dest, const B& v) { dest = (A)(dest + (A)v); }
template <class A, class B, class C> inline void InterlockedAdd(A& dest, const B& v, C& original) {
    original = (C)dest;
    dest = (A)(dest + (A)v);
}
template <class A, class B> inline void InterlockedMax(A& dest, const B& v) { dest = dest > (A)v ? dest : (A)v; }
template <class A, class B> inline void InterlockedMin(A& dest, const B& v) { dest = dest < (A)v ? dest : (A)v; }
template <class A, class B> inline void InterlockedOr(A& dest, const B& v) { dest = (A)(dest | (A)v); }

} // namespace hlsl

// SRV or UAV is decided by the resource's TYPE, not by the macro that declared it: reference Shaders/Resources/REFERENCE_Copy.resources.hlsli:19 declares its
// input texture with NRD_OUTPUT( Texture2D<float4>, gIn_Input, t, 0 ) -- harmless for the shader compilers (both macros expand to `type name : register( t0 )`)
namespace hlsl_rt {
template <class T> struct is_uav { static constexpr bool value = false; };
template <class T> struct is_uav<hlsl::RWTexture2D<T>> { static constexpr bool value = true; };
} // namespace hlsl_rt

// what the binding macros of the prelude (oracle/ref/prelude.hlsli) leave in the preprocessed shader text
#define HLSL_CONSTANT(type, name) \
    static type name;             \
    static hlsl_rt::ConstantAdder name##_reg(hlsl_table(), &name);
#define HLSL_INPUT(type, name, index) \
    static type name;                 \
    static hlsl_rt::ResourceAdder name##_reg(hlsl_table(), &name.p, hlsl_rt::is_uav<type>::value, index, #name);
#define HLSL_OUTPUT(type, name, index) \
    static type name;                  \
    static hlsl_rt::ResourceAdder name##_reg(hlsl_table(), &name.p, hlsl_rt::is_uav<type>::value, index, #name);
