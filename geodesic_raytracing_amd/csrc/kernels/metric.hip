// ------------------------------------------------------------------------------------------------
// math used by the generated expressions.  Everything generated is evaluated inside namespace gm,
// so unqualified sin/cos/... bind to these fp32 versions.
namespace gm {

#if defined(GR_FAST_TRIG)
// hardware v_sin_f32 / v_cos_f32: ~1e-6 absolute error (poor relative accuracy next to the zeros), opt-in only
__device__ __forceinline__ float sin(float x) { return __sinf(x); }
__device__ __forceinline__ float cos(float x) { return __cosf(x); }
#elif defined(GR_LIBM_TRIG)
__device__ __forceinline__ float sin(float x) { return ::sinf(x); }
__device__ __forceinline__ float cos(float x) { return ::cosf(x); }
#else
// sin and cos of the same angle share one Cody-Waite reduction and both minimax polynomials (the common
// sub-expressions of the two inlined calls merge), ~1 ulp for |x| < 8192; larger arguments take the libm path.
// The metric expressions evaluate sin(theta) and cos(theta) together every Verlet step, where the two separate libm
// calls (each with its own large-argument branch) were ~25 % of the step's instructions.
struct sincos_pair { float s, c; };
// POISON_LARGE: an argument outside the polynomial's range (|x| >= 8192) returns NaN for both - two full-rate instructions, no
// compare, no branch, no flag to carry: x * 4.154e34 overflows exactly then, and fma(inf, 0, r) is NaN while fma(finite, 0, r) is r
template <bool POISON_LARGE = false>
__device__ __forceinline__ sincos_pair sincos_reduced(float x) {
#pragma clang fp reassociate(off)
    // nearest multiple of pi/2 by the 1.5 * 2^23 trick: the rounded quotient lands in the low mantissa bits of t (so the
    // quadrant needs no v_cvt_i32_f32) and j = t - magic is exact; both are full-rate ops where v_rndne_f32 and
    // v_cvt_i32_f32 issue at half rate.  Valid for |x| < 2^21 (callers only trust the result below 8192).  Re-association
    // (allowed by the build flags elsewhere) is off in this function so that the two constants are not cancelled.
    float t = __builtin_fmaf(x, 0.636619772367581343f, 12582912.f);
    float j = t - 12582912.f;
    unsigned int q = __builtin_bit_cast(unsigned int, t);
    float r = __builtin_fmaf(-j, 1.57079637050628662109375f, x);   // pi/2 = hi + lo, fma keeps the product exact
    r = __builtin_fmaf(-j, -4.37113900018624283e-8f, r);
    if (POISON_LARGE) r = __builtin_fmaf(x * 4.1539e34f, 0.f, r);
    float r2 = r * r;
    float sp = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f), r2 * r, r);
    float cp = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f),
                              r2 * r2, __builtin_fmaf(-0.5f, r2, 1.0f));
    float s = (q & 1) ? cp : sp;
    float c = (q & 1) ? sp : cp;
    // quadrant signs applied to the sign bit directly (xor is full rate, compare + select are not)
    s = __builtin_bit_cast(float, __builtin_bit_cast(unsigned int, s) ^ ((q & 2u) << 30));
    c = __builtin_bit_cast(float, __builtin_bit_cast(unsigned int, c) ^ (((q + 1u) & 2u) << 30));
    return {s, c};
}
// sin^2, cos^2 and sin cos of one angle (the code generator's gr_sin2 / gr_cos2 / gr_sincos, csrc/sym.cpp lower_for_device - all a
// Boyer-Lindquist chart ever asks of its polar angle).  All three have period pi and none of them cares which of sin, cos carries a
// sign, so the angle is reduced by multiples of PI to [-pi/2, pi/2] and that is the end of it: no quadrant, no "which polynomial is
// the sine", no sign bits (round 5 reduced by pi/2 and paid v_and + v_cmp + two v_cndmask + v_lshlrev + v_xor = 21 issue cycles per
// attempt for them; the two polynomials for the wider interval cost one fma each more: 4.5 cycles).  Minimax fits for |r| <= pi/2
// (tools/ubench/fit_sincos.py): |sin error| <= 1.3e-7, |cos error| <= 7.5e-8 (1.1 / 0.6 ulp of 1) in fp32 with fmas - the sine keeps
// its RELATIVE accuracy at the poles of the chart (r -> 0: sin r = r (1 + ...)), where 1 / sin^2 amplifies it; the cosine's zero at the
// equator gets absolute accuracy.
struct sincos_products_t { float s2, c2, sc; };
template <bool POISON_LARGE = false>
__device__ __forceinline__ sincos_products_t sincos_products(float x) {
#pragma clang fp reassociate(off)
    float t = __builtin_fmaf(x, 0.318309886183790672f, 12582912.f);   // nearest multiple of pi by the 1.5 * 2^23 trick (sincos_reduced)
    float j = t - 12582912.f;
    float r = __builtin_fmaf(-j, 3.1415927410125732421875f, x);       // pi = hi + lo, the fma keeps the product exact
    r = __builtin_fmaf(-j, -8.74227800037248566e-8f, r);
    if (POISON_LARGE) r = __builtin_fmaf(x * 4.1539e34f, 0.f, r);
    float r2 = r * r;
    float sp = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(__builtin_fmaf(2.599902700239909e-06f, r2, -0.00019806546333711594f), r2, 0.008333016186952591f), r2,
                                             -0.16666656732559204f), r2 * r, r);
    float cp = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(__builtin_fmaf(-2.6192478230768756e-07f, r2, 2.4769240553723648e-05f), r2, -0.0013888567918911576f), r2,
                                             0.041666656732559204f), r2 * r2, __builtin_fmaf(-0.5f, r2, 1.0f));
    sincos_products_t p;
    p.s2 = sp * sp;
    p.c2 = cp * cp;
    p.sc = sp * cp;
    return p;
}
// the polynomial is evaluated unconditionally (so sin and cos of one angle stay in one basic block and share it);
// the libm call only overrides the result in the never-in-practice large-argument case
__device__ __forceinline__ float sin(float x) {
    float s = sincos_reduced(x).s;
    if (__builtin_expect(!(__builtin_fabsf(x) < 8192.f), 0)) s = ::sinf(x);
    return s;
}
__device__ __forceinline__ float cos(float x) {
    float c = sincos_reduced(x).c;
    if (__builtin_expect(!(__builtin_fabsf(x) < 8192.f), 0)) c = ::cosf(x);
    return c;
}
#endif
__device__ __forceinline__ float tan(float x) { return ::tanf(x); }
__device__ __forceinline__ float asin(float x) { return ::asinf(x); }
__device__ __forceinline__ float acos(float x) { return ::acosf(x); }
__device__ __forceinline__ float atan(float x) { return ::atanf(x); }
__device__ __forceinline__ float atan2(float y, float x) { return ::atan2f(y, x); }
__device__ __forceinline__ float exp(float x) { return ::expf(x); }
__device__ __forceinline__ float log(float x) { return ::logf(x); }
__device__ __forceinline__ float sqrt(float x) { return __builtin_sqrtf(x); }
__device__ __forceinline__ float fabs(float x) { return __builtin_fabsf(x); }
__device__ __forceinline__ float sinh(float x) { return ::sinhf(x); }
__device__ __forceinline__ float cosh(float x) { return ::coshf(x); }
// tanh x = 1 - 2 / (e^{2x} + 1) on the hardware's exp2 and reciprocal: 5 instructions where the library routine (range split,
// polynomial, selects) compiles to ~40 - the Alcubierre acceleration calls it twice per attempt, a quarter of its loop.  Absolute
// error <= 1.2e-7 everywhere (the subtraction from 1 costs relative accuracy near 0, where the warp-drive shape function only
// uses differences of tanh of O(1) arguments); saturates to +-1, NaN stays NaN.  -DGR_LIBM_TANH: the library routine.
// Its RELATIVE error is unbounded next to 0 (tanh x for |x| < 1e-7 comes out 0 or one quantum), so it is used only where the code
// generator has seen that nothing depends on that: -DGR_TANH_IN_SUMS_ONLY, emitted (metric_codegen.cpp) when every tanh of the
// metric meets only sums, differences and products with other tanh values, constants and $cfg-only factors - the shape functions
// of the warp drives.  A script that writes tanh(x) / x or scales a tanh by a coordinate gets the library routine.
__device__ __forceinline__ float tanh(float x) {
#if defined(GR_LIBM_TANH) || !defined(GR_TANH_IN_SUMS_ONLY)
    return ::tanhf(x);
#else
    return 1.f - 2.f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x * 2.88539008177792681472f) + 1.f);
#endif
}
// outside the Verlet loop (and in builds with another trigonometry flavour): the products from sin and cos as they are there
__device__ __forceinline__ float gr_sin2(float x) { const float s = sin(x); return s * s; }
__device__ __forceinline__ float gr_cos2(float x) { const float c = cos(x); return c * c; }
__device__ __forceinline__ float gr_sincos(float x) { return sin(x) * cos(x); }
// device-only forms the code generator lowers to (csrc/sym.cpp lower_for_device; GR_DEVICE_ACCEL*): bare v_exp_f32 and v_rsq_f32
__device__ __forceinline__ float gr_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float gr_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
// ... and its quotients: the device's rendering writes a / b as gr_div(a, b) and 1 / b as gr_rcp(b).  By default they ARE the operator
// (under the build's relaxed arithmetic: a * v_rcp_f32(b)).  A program whose argument string carries -DGR_REFINED_RECIPROCALS evaluates
// them inside the Verlet loop as the correctly rounded quotient instead (geodesic_acceleration_with, below): v_rcp_f32 is the correctly
// rounded reciprocal for 89.3 % of operands and one ulp off for the rest, one Newton step r + r (1 - b r) makes it the correctly rounded one
// for all of 2^26 random operands, and q + r (a - b q) does the same for the quotient q = a r (71 % -> 100 %; tools/ubench/
// reciprocal_refinement.hip, profiles/r06_refined_reciprocals.txt) - what the reference's x86 build divides with, for 2 and 5 full-rate
// instructions instead of the ten of the compiler's IEEE sequence.  v_sqrt_f32 needs nothing: it is correctly rounded on gfx950.
template <class A, class B> __device__ __forceinline__ auto gr_div(A a, B b) -> decltype(a / b) { return a / b; }
template <class B> __device__ __forceinline__ auto gr_rcp(B b) -> decltype(1.0f / b) { return 1.0f / b; }
__device__ __forceinline__ float pow(float x, float y) { return ::powf(x, y); }
__device__ __forceinline__ float fmod(float x, float y) { return ::fmodf(x, y); }
__device__ __forceinline__ float fmin(float x, float y) { return __builtin_fminf(x, y); }
__device__ __forceinline__ float fmax(float x, float y) { return __builtin_fmaxf(x, y); }
__device__ __forceinline__ float sign(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

// --- $cfg-only temporaries in double precision ------------------------------------------------------
// GR_CFG_TEMPORARIES (an extension of the macro set, metric_codegen.cpp): the named sub-expressions that depend on nothing but the
// $cfg parameters, every leaf wrapped into cfgf - a double behind float's interface - so that the generated text, written for
// float, evaluates them in double; GR_POS_TEMPORARIES is TEMPORARIES0 with those entries replaced by cfg_value(cpvN).  The build's
// relaxed fp32 arithmetic (v_rcp_f32, v_sqrt_f32, approximate functions) is right for the Verlet loop and wrong for a parameter
// expression that cancels (the cubic root of the double-Kerr solution as its spins go to 0: a tenth of the dynamic program's frame).
// These values are wave-uniform and loop-invariant; a string without the extension (the reference's own generator) falls back on
// TEMPORARIES0.  float -> cfgf is explicit and cfgf -> float implicit, so that "c ? cfgf : float" has one common type (float).
struct cfgf {
    double v;
    __device__ __forceinline__ cfgf() {}
    __device__ __forceinline__ explicit cfgf(double x) : v(x) {}
    __device__ __forceinline__ operator float() const { return (float)v; }
};
__device__ __forceinline__ float cfg_value(cfgf a) { return (float)a.v; }
#define GR_CFGF_BINARY(op)                                                                        \
    __device__ __forceinline__ cfgf operator op(cfgf a, cfgf b) { return cfgf(a.v op b.v); }      \
    __device__ __forceinline__ cfgf operator op(cfgf a, float b) { return cfgf(a.v op (double)b); } \
    __device__ __forceinline__ cfgf operator op(float a, cfgf b) { return cfgf((double)a op b.v); }
GR_CFGF_BINARY(+) GR_CFGF_BINARY(-) GR_CFGF_BINARY(*) GR_CFGF_BINARY(/)
#undef GR_CFGF_BINARY
__device__ __forceinline__ cfgf operator-(cfgf a) { return cfgf(-a.v); }
#define GR_CFGF_FN1(name) __device__ __forceinline__ cfgf name(cfgf a) { return cfgf(::name(a.v)); }
GR_CFGF_FN1(sin) GR_CFGF_FN1(cos) GR_CFGF_FN1(tan) GR_CFGF_FN1(asin) GR_CFGF_FN1(acos) GR_CFGF_FN1(atan) GR_CFGF_FN1(exp) GR_CFGF_FN1(log)
GR_CFGF_FN1(sqrt) GR_CFGF_FN1(fabs) GR_CFGF_FN1(sinh) GR_CFGF_FN1(cosh) GR_CFGF_FN1(tanh)
#undef GR_CFGF_FN1
#define GR_CFGF_FN2(name)                                                                              \
    __device__ __forceinline__ cfgf name(cfgf a, cfgf b) { return cfgf(::name(a.v, b.v)); }            \
    __device__ __forceinline__ cfgf name(cfgf a, float b) { return cfgf(::name(a.v, (double)b)); }     \
    __device__ __forceinline__ cfgf name(float a, cfgf b) { return cfgf(::name((double)a, b.v)); }
GR_CFGF_FN2(atan2) GR_CFGF_FN2(pow) GR_CFGF_FN2(fmod) GR_CFGF_FN2(fmin) GR_CFGF_FN2(fmax)
#undef GR_CFGF_FN2
__device__ __forceinline__ cfgf sign(cfgf x) { return cfgf(x.v > 0. ? 1. : (x.v < 0. ? -1. : 0.)); }
#if defined(GR_CFG_TEMPORARIES) && defined(GR_POS_TEMPORARIES) && !defined(GR_NO_CFG_TEMPORARIES)
#define GR_DECLARE_TEMPORARIES(T) gm::cfgf GR_CFG_TEMPORARIES; T GR_POS_TEMPORARIES;
#else
#define GR_DECLARE_TEMPORARIES(T) T TEMPORARIES0;
#endif

// --- generated-expression hosts (cl.cl:969-1355, 3377-3387) ---------------------------------------

#define GR_POSITION_VARS(p) \
    const float v1 = (p).x; const float v2 = (p).y; const float v3 = (p).z; const float v4 = (p).w; \
    const float rs = RS_IMPL; const float c = C_IMPL; (void)v1; (void)v2; (void)v3; (void)v4; (void)rs; (void)c;

// g_metric_out has 4 (diagonal) or 16 entries
__device__ __forceinline__ void metric_at(float4 pos, float* g, cfg_t cfg) {
    GR_POSITION_VARS(pos)
    GR_DECLARE_TEMPORARIES(float)
#ifndef GENERIC_BIG_METRIC
    g[0] = F1_I; g[1] = F2_I; g[2] = F3_I; g[3] = F4_I;
#else
    g[0] = F1_I; g[1] = F2_I; g[2] = F3_I; g[3] = F4_I;
    g[4] = g[1]; g[5] = F6_I; g[6] = F7_I; g[7] = F8_I;
    g[8] = g[2]; g[9] = g[6]; g[10] = F11_I; g[11] = F12_I;
    g[12] = g[3]; g[13] = g[7]; g[14] = g[11]; g[15] = F16_I;
#endif
}

// always the full symmetric 4x4
__device__ __forceinline__ void metric_big_at(float4 pos, float* g, cfg_t cfg) {
#ifndef GENERIC_BIG_METRIC
    float d[4];
    metric_at(pos, d, cfg);
    for (int i = 0; i < 16; i++) g[i] = 0.f;
    g[0] = d[0]; g[5] = d[1]; g[10] = d[2]; g[15] = d[3];
#else
    metric_at(pos, g, cfg);
#endif
}

// d g_ij / d v_k as dg[k*16 + i*4 + j] (calculate_partial_derivatives_generic(_big), cl.cl:987-1015, 1053-1199).
// Only the geodesic-camera kernels (parallel transport) use it; the ray kernels never need the partials at run time.
__device__ void partials_big_at(float4 pos, float* dg, cfg_t cfg) {
    GR_POSITION_VARS(pos)
    GR_DECLARE_TEMPORARIES(float)
    for (int i = 0; i < 64; i++) dg[i] = 0.f;
#ifndef GENERIC_BIG_METRIC
    const float p[16] = {F1_P, F2_P, F3_P, F4_P, F5_P, F6_P, F7_P, F8_P, F9_P, F10_P, F11_P, F12_P, F13_P, F14_P, F15_P, F16_P};
    for (int var = 0; var < 4; var++)
        for (int wrt = 0; wrt < 4; wrt++) dg[wrt * 16 + var * 4 + var] = p[var * 4 + wrt];
#else
    const float p[64] = {F1_P, F2_P, F3_P, F4_P, 0, F6_P, F7_P, F8_P, 0, 0, F11_P, F12_P, 0, 0, 0, F16_P,
                         F17_P, F18_P, F19_P, F20_P, 0, F22_P, F23_P, F24_P, 0, 0, F27_P, F28_P, 0, 0, 0, F32_P,
                         F33_P, F34_P, F35_P, F36_P, 0, F38_P, F39_P, F40_P, 0, 0, F43_P, F44_P, 0, 0, 0, F48_P,
                         F49_P, F50_P, F51_P, F52_P, 0, F54_P, F55_P, F56_P, 0, 0, F59_P, F60_P, 0, 0, 0, F64_P};
    for (int k = 0; k < 4; k++)
        for (int i = 0; i < 4; i++)
            for (int j = i; j < 4; j++) {
                dg[k * 16 + i * 4 + j] = p[k * 16 + i * 4 + j];
                dg[k * 16 + j * 4 + i] = p[k * 16 + i * 4 + j];
            }
#endif
}

// get_coordinate_period, cl.cl:1338-1355
__device__ __forceinline__ float4 coordinate_period(cfg_t cfg) {
#ifdef HAS_COORDINATE_PERIODICITY
    const float4 zero = make_float4(0, 0, 0, 0);
    GR_POSITION_VARS(zero)
    return make_float4(COORDINATE_PERIODICITY1, COORDINATE_PERIODICITY2, COORDINATE_PERIODICITY3, COORDINATE_PERIODICITY4);
#else
    return make_float4(0, 0, 0, 0);
#endif
}

// closed-form geodesic acceleration (GEO_ACCELn; step_verlet cl.cl:3279-3309)
#if defined(GR_FAST_TRIG) || defined(GR_LIBM_TRIG)
#define GR_ACCEL_TRIG(LIBM)
#else
// Inside the Verlet loop the expressions' sin / cos are the bare polynomial, which answers an argument outside its range with
// NaN.  The step controller then treats the attempt like any other that ends in a degenerate velocity - the ray leaves the fast
// loop - and whoever left that way has the attempt redone by a loop that calls libm (integrate_pingpong): a genuinely
// degenerate step is found degenerate again, a large argument - never seen in practice - is integrated on.  libm's argument
// reduction, two copies of it per evaluation, stays out of the loop body every ray runs and out of its register budget, and
// the fast loop carries no flag for it.
#define GR_ACCEL_TRIG(LIBM)                                                                              \
    auto sin = [&](float x) -> float { return LIBM ? ::sinf(x) : sincos_reduced<true>(x).s; };           \
    auto cos = [&](float x) -> float { return LIBM ? ::cosf(x) : sincos_reduced<true>(x).c; };           \
    auto gr_sin2 = [&](float x) -> float { if (LIBM) { const float s = ::sinf(x); return s * s; } return sincos_products<true>(x).s2; };             \
    auto gr_cos2 = [&](float x) -> float { if (LIBM) { const float c = ::cosf(x); return c * c; } return sincos_products<true>(x).c2; };             \
    auto gr_sincos = [&](float x) -> float { if (LIBM) return ::sinf(x) * ::cosf(x); return sincos_products<true>(x).sc; };                          \
    (void)sin; (void)cos; (void)gr_sin2; (void)gr_cos2; (void)gr_sincos;
#endif
template <bool LIBM>
__device__ __forceinline__ float4 geodesic_acceleration_with(float4 pos, float4 vel, cfg_t cfg) {
#ifdef GENERIC_CONSTANT_THETA
    pos.z = GR_PIf / 2;
    vel.z = 0.f;
#endif
    GR_POSITION_VARS(pos)
    const float iv1 = vel.x; const float iv2 = vel.y; const float iv3 = vel.z; const float iv4 = vel.w;
    (void)iv1; (void)iv2; (void)iv3; (void)iv4;
    GR_ACCEL_TRIG(LIBM)
#ifdef GR_REFINED_RECIPROCALS
    auto gr_rcp = [&](float b) -> float { const float r = __builtin_amdgcn_rcpf(b); return __builtin_fmaf(__builtin_fmaf(-b, r, 1.0f), r, r); };
    auto gr_div = [&](float n, float b) -> float { const float r = gr_rcp(b), q = n * r; return __builtin_fmaf(__builtin_fmaf(-b, q, n), r, q); };
    // ... and the reciprocal root of the lowering (x / sqrt(s) -> x gr_rsqrt(s)) as the refined reciprocal of v_sqrt_f32, which is correctly rounded
    auto gr_rsqrt = [&](float x) -> float { return gr_rcp(__builtin_sqrtf(x)); };
    (void)gr_rcp; (void)gr_div; (void)gr_rsqrt;
#endif
    float4 a;
#if defined(GR_DEVICE_ACCEL0) && !defined(GR_NO_DEVICE_LOWERING)
    // the Verlet loop's own form of the same expressions (metric_codegen.cpp: GR_DEVICE_ACCEL*)
    float GR_DEVICE_TEMPORARIES;
    a.x = GR_DEVICE_ACCEL0;
    a.y = GR_DEVICE_ACCEL1;
#ifndef GENERIC_CONSTANT_THETA
    a.z = GR_DEVICE_ACCEL2;
#else
    a.z = 0.f;
#endif
    a.w = GR_DEVICE_ACCEL3;
#else
    GR_DECLARE_TEMPORARIES(float)
    a.x = GEO_ACCEL0;
    a.y = GEO_ACCEL1;
#ifndef GENERIC_CONSTANT_THETA
    a.z = GEO_ACCEL2;
#else
    a.z = 0.f;
#endif
    a.w = GEO_ACCEL3;
#endif
    return a;
}
// everywhere outside the Verlet loop (ray set-up, geodesic paths): gm::sin / gm::cos with their own large-argument branches
__device__ __forceinline__ float4 geodesic_acceleration(float4 pos, float4 vel, cfg_t cfg) {
#ifdef GENERIC_CONSTANT_THETA
    pos.z = GR_PIf / 2;
    vel.z = 0.f;
#endif
    GR_POSITION_VARS(pos)
    const float iv1 = vel.x; const float iv2 = vel.y; const float iv3 = vel.z; const float iv4 = vel.w;
    (void)iv1; (void)iv2; (void)iv3; (void)iv4;
    GR_DECLARE_TEMPORARIES(float)
    float4 a;
    a.x = GEO_ACCEL0;
    a.y = GEO_ACCEL1;
#ifndef GENERIC_CONSTANT_THETA
    a.z = GEO_ACCEL2;
#else
    a.z = 0.f;
#endif
    a.w = GEO_ACCEL3;
    return a;
}

__device__ __forceinline__ float4 generic_to_spherical(float4 in, cfg_t cfg) {
    GR_POSITION_VARS(in)
    return make_float4(TO_COORD1, TO_COORD2, TO_COORD3, TO_COORD4);
}

__device__ __forceinline__ float4 generic_velocity_to_spherical_velocity(float4 in, float4 inv, cfg_t cfg) {
    GR_POSITION_VARS(in)
    const float dv1 = inv.x; const float dv2 = inv.y; const float dv3 = inv.z; const float dv4 = inv.w;
    (void)dv1; (void)dv2; (void)dv3; (void)dv4;
    return make_float4(TO_DCOORD1, TO_DCOORD2, TO_DCOORD3, TO_DCOORD4);
}

__device__ __forceinline__ float4 spherical_to_generic(float4 in, cfg_t cfg) {
    GR_POSITION_VARS(in)
    return make_float4(FROM_COORD1, FROM_COORD2, FROM_COORD3, FROM_COORD4);
}

__device__ __forceinline__ float4 spherical_velocity_to_generic_velocity(float4 in, float4 inv, cfg_t cfg) {
    GR_POSITION_VARS(in)
    const float dv1 = inv.x; const float dv2 = inv.y; const float dv3 = inv.z; const float dv4 = inv.w;
    (void)dv1; (void)dv2; (void)dv3; (void)dv4;
    return make_float4(FROM_DCOORD1, FROM_DCOORD2, FROM_DCOORD3, FROM_DCOORD4);
}

#ifdef GR_POLAR_R_SQUARED
// TO_COORD2 squared, straight from the metric's own coordinates (metric_codegen.cpp: emitted when TO_COORD2 is a square root)
__device__ __forceinline__ float polar_radius_squared(float4 in, cfg_t cfg) {
    GR_POSITION_VARS(in)
    return GR_POLAR_R_SQUARED;
}
#endif

__device__ __forceinline__ float distance_to_object(float4 polar, cfg_t cfg) {
    GR_POSITION_VARS(polar)
    return DISTANCE_FUNC;
}
// The same distance straight from the metric's own coordinates.  The host composes DISTANCE_FUNC with TO_COORDn symbolically and
// emits GR_DISTANCE_OF_GENERIC when the coordinate round trip cancels completely (csrc/sym.cpp cancel_round_trip: Alcubierre's
// origin function takes polar coordinates back to Cartesian ones - two atan2, two sin / cos pairs per Verlet attempt, for a
// distance that is one square root of the position).  A macro string without it (the reference generator's) takes the long way.
__device__ __forceinline__ float distance_to_object_from(float4 position, float4 polar, cfg_t cfg) {
#ifdef GR_DISTANCE_OF_GENERIC
    GR_POSITION_VARS(position)
    (void)polar;
    return GR_DISTANCE_OF_GENERIC;
#else
    (void)position;
    return distance_to_object(polar, cfg);
#endif
}

#ifdef GR_DISTANCE_SQUARED_OF_GENERIC
__device__ __forceinline__ float distance_squared_from(float4 position, cfg_t cfg) {
    GR_POSITION_VARS(position)
    return GR_DISTANCE_SQUARED_OF_GENERIC;
}
#endif

#ifdef GR_TWO_RAYS_PER_LANE
// --- the same hosts for two rays per lane (gr_trace_pair): every variable of the generated expressions is a pair of
// floats, one per ray, so their multiplies, adds and fmas become v_pk_mul/add/fma_f32 with nothing to shuffle ------------
typedef float pairf __attribute__((ext_vector_type(2)));
typedef unsigned int pairu __attribute__((ext_vector_type(2)));
struct pair4 { pairf x, y, z, w; };

__device__ __forceinline__ pairf pfma(pairf a, pairf b, pairf c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ pairf splat(float x) { pairf r; r.x = x; r.y = x; return r; }
struct sincos_pairf { pairf s, c; };
// sincos_reduced (above), both rays at once; the quadrant select stays per ray (v_cndmask has no packed form)
__device__ __forceinline__ sincos_pairf sincos_reduced(pairf x) {
#pragma clang fp reassociate(off)
    pairf t = pfma(x, splat(0.636619772367581343f), splat(12582912.f));
    pairf j = t - splat(12582912.f);
    pairu q = __builtin_bit_cast(pairu, t);
    pairf r = pfma(-j, splat(1.57079637050628662109375f), x);
    r = pfma(-j, splat(-4.37113900018624283e-8f), r);
    pairf r2 = r * r;
    pairf sp = pfma(pfma(pfma(splat(-1.9515295891e-4f), r2, splat(8.3321608736e-3f)), r2, splat(-1.6666654611e-1f)), r2 * r, r);
    pairf cp = pfma(pfma(pfma(splat(2.443315711809948e-5f), r2, splat(-1.388731625493765e-3f)), r2, splat(4.166664568298827e-2f)),
                    r2 * r2, pfma(splat(-0.5f), r2, splat(1.0f)));
    pairf s, c;
    s.x = (q.x & 1) ? cp.x : sp.x; s.y = (q.y & 1) ? cp.y : sp.y;
    c.x = (q.x & 1) ? sp.x : cp.x; c.y = (q.y & 1) ? sp.y : cp.y;
    pairu two; two.x = 2u; two.y = 2u;
    pairu one; one.x = 1u; one.y = 1u;
    pairu sh; sh.x = 30u; sh.y = 30u;
    s = __builtin_bit_cast(pairf, __builtin_bit_cast(pairu, s) ^ ((q & two) << sh));
    c = __builtin_bit_cast(pairf, __builtin_bit_cast(pairu, c) ^ (((q + one) & two) << sh));
    return {s, c};
}
// the never-in-practice large-argument case goes through one out-of-line libm call per value: inlined (as in the one-ray
// kernel) its four copies cost the pair kernel scalar-register spills
__device__ __attribute__((noinline)) float sin_large(float x) { return ::sinf(x); }
__device__ __attribute__((noinline)) float cos_large(float x) { return ::cosf(x); }
__device__ __forceinline__ bool large_finite(float x) { return __builtin_fabsf(x) >= 8192.f && __builtin_fabsf(x) <= 3.402823466e+38f; }
__device__ __forceinline__ pairf sin(pairf x) {
#if defined(GR_FAST_TRIG) || defined(GR_LIBM_TRIG)
    pairf s; s.x = gm::sin(x.x); s.y = gm::sin(x.y);
#else
    pairf s = sincos_reduced(x).s;
    // per half: a ray's value must not depend on what its lane partner holds (a partner frozen at a NaN/Inf final state keeps
    // being evaluated).  NaN/Inf need no libm either: the polynomial already returns NaN for them.
    if (__builtin_expect(large_finite(x.x), 0)) s.x = sin_large(x.x);
    if (__builtin_expect(large_finite(x.y), 0)) s.y = sin_large(x.y);
#endif
    return s;
}
__device__ __forceinline__ pairf cos(pairf x) {
#if defined(GR_FAST_TRIG) || defined(GR_LIBM_TRIG)
    pairf c; c.x = gm::cos(x.x); c.y = gm::cos(x.y);
#else
    pairf c = sincos_reduced(x).c;
    if (__builtin_expect(large_finite(x.x), 0)) c.x = cos_large(x.x);
    if (__builtin_expect(large_finite(x.y), 0)) c.y = cos_large(x.y);
#endif
    return c;
}
// everything else the expressions may call: once per ray
#define GR_PAIR_FN1(name) \
    __device__ __forceinline__ pairf name(pairf x) { pairf r; r.x = gm::name(x.x); r.y = gm::name(x.y); return r; }
#define GR_PAIR_FN2(name) \
    __device__ __forceinline__ pairf name(pairf x, pairf y) { pairf r; r.x = gm::name(x.x, y.x); r.y = gm::name(x.y, y.y); return r; } \
    __device__ __forceinline__ pairf name(pairf x, float y) { pairf r; r.x = gm::name(x.x, y); r.y = gm::name(x.y, y); return r; }     \
    __device__ __forceinline__ pairf name(float x, pairf y) { pairf r; r.x = gm::name(x, y.x); r.y = gm::name(x, y.y); return r; }
GR_PAIR_FN1(tan) GR_PAIR_FN1(asin) GR_PAIR_FN1(acos) GR_PAIR_FN1(atan) GR_PAIR_FN1(exp) GR_PAIR_FN1(log) GR_PAIR_FN1(sqrt)
GR_PAIR_FN1(fabs) GR_PAIR_FN1(sinh) GR_PAIR_FN1(cosh) GR_PAIR_FN1(tanh) GR_PAIR_FN1(sign)
GR_PAIR_FN2(atan2) GR_PAIR_FN2(pow) GR_PAIR_FN2(fmod) GR_PAIR_FN2(fmin) GR_PAIR_FN2(fmax)
#undef GR_PAIR_FN1
#undef GR_PAIR_FN2

#define GR_POSITION_VARS_PAIR(p) \
    const pairf v1 = (p).x; const pairf v2 = (p).y; const pairf v3 = (p).z; const pairf v4 = (p).w; \
    const float rs = RS_IMPL; const float c = C_IMPL; (void)v1; (void)v2; (void)v3; (void)v4; (void)rs; (void)c;

__device__ __forceinline__ pair4 geodesic_acceleration(pair4 pos, pair4 vel, cfg_t cfg) {
#ifdef GENERIC_CONSTANT_THETA
    pos.z = splat(GR_PIf / 2);
    vel.z = splat(0.f);
#endif
    GR_POSITION_VARS_PAIR(pos)
    const pairf iv1 = vel.x; const pairf iv2 = vel.y; const pairf iv3 = vel.z; const pairf iv4 = vel.w;
    (void)iv1; (void)iv2; (void)iv3; (void)iv4;
    GR_DECLARE_TEMPORARIES(pairf)
    pair4 a;
    a.x = GEO_ACCEL0;
    a.y = GEO_ACCEL1;
#ifndef GENERIC_CONSTANT_THETA
    a.z = GEO_ACCEL2;
#else
    a.z = splat(0.f);
#endif
    a.w = GEO_ACCEL3;
    return a;
}
__device__ __forceinline__ pair4 generic_to_spherical(pair4 in, cfg_t cfg) {
    GR_POSITION_VARS_PAIR(in)
    pair4 r;
    r.x = TO_COORD1; r.y = TO_COORD2; r.z = TO_COORD3; r.w = TO_COORD4;
    return r;
}
__device__ __forceinline__ pairf distance_to_object(pair4 polar, cfg_t cfg) {
    GR_POSITION_VARS_PAIR(polar)
    pairf d = DISTANCE_FUNC;
    return d;
}
// as distance_to_object_from above: every integrator takes its step-size distance from the same expression
__device__ __forceinline__ pairf distance_to_object_from(pair4 position, pair4 polar, cfg_t cfg) {
#ifdef GR_DISTANCE_OF_GENERIC
    GR_POSITION_VARS_PAIR(position)
    (void)polar;
    pairf d = GR_DISTANCE_OF_GENERIC;
    return d;
#else
    (void)position;
    return distance_to_object(polar, cfg);
#endif
}
#endif  // GR_TWO_RAYS_PER_LANE

}  // namespace gm

