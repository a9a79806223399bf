// geodesic_kernels.hip — gfx950 (CDNA4, wave64) kernels of the per-pixel geodesic ray pipeline.
//
// This translation unit is compiled at run time (hiprtc, --offload-arch=gfx950) once per metric,
// specialised by the same `-D` macro set the reference feeds to its OpenCL program
// (producer metric.hpp:725-959, consumer cl.cl): F*_I, F*_P, TO_/FROM_(D)COORDn, GEO_ACCELn,
// TEMPORARIES0, DISTANCE_FUNC, W_Vn, DYNVARS, feature macros, behaviour flags.
//
// Kernel <-> reference map (semantics, argument order):
//   gr_cart_to_generic          cart_to_generic_kernel    cl.cl:6018-6034
//   gr_init_basis_vectors       init_basis_vectors        cl.cl:2483-2507 (calculate_tetrads 2288-2439)
//   gr_clear_termination_buffer clear_termination_buffer  cl.cl:4997-5006
//   gr_init_rays_generic        init_rays_generic         cl.cl:3143-3251
//   gr_do_generic_rays          do_generic_rays           cl.cl:3954-4247 (step_verlet 3273-3346,
//                                                         calculate_ds_error 3431-3456)
//   gr_calculate_singularities  calculate_singularities   cl.cl:5008-5020
//   gr_calculate_render_data    calculate_render_data     cl.cl:5135-5213
//   gr_handle_adaptive_sampling handle_adaptive_sampling  cl.cl:5223-5345
//   gr_render                   render                    cl.cl:5453-5846 (read_mipmap 5421-5449)
//   gr_trace_fused              (no counterpart) init -> integrate -> render-data in one launch,
//                               ray state never leaves registers; persistent tile-waves
//   gr_trace_pair               (no counterpart) gr_trace_fused with two rays per lane in packed fp32 (integrate_pair);
//                               only built for programs whose loop expressions instantiate on float pairs
//                               (GR_TWO_RAYS_PER_LANE, decided by the host: capi.cpp pair_kernel_applies)
//   gr_trace_compact            (no counterpart) gr_trace_fused with ray compaction (resumable integrator)
//   gr_prepass_fused            (no counterpart) the W/16 x H/16 prepass as one launch -> termination flags
//   gr_boost_tetrad             boost_tetrad              cl.cl:2441-2481
//   gr_init_inertial_ray        init_inertial_ray         cl.cl:3117-3141
//   gr_get_geodesic_path        get_geodesic_path         cl.cl:4735-4940
//   gr_parallel_transport_quantity  parallel_transport_quantity  cl.cl:2569-2620
//   gr_handle_interpolating_geodesic  handle_interpolating_geodesic  cl.cl:2738-2872
//
// MI355X design notes
//   * one ray per lane, one wave64 per 64 consecutive ray slots; ray slots are laid out in 8x8
//     pixel tiles (GR_TILE) so a wave integrates an angularly compact bundle: step counts inside
//     a wave stay close and the lock-step loop wastes few lanes;
//   * the integrator state (position, velocity, acceleration, step, flags = 16 VGPRs) and every
//     metric temporary live in registers; cfg / feature values are wave-uniform kernel-argument
//     loads (SGPRs);
//   * accept / reject of an adaptive step is a per-lane select - both outcomes ran the same
//     step_verlet, so rejection costs no divergence; a wave leaves the loop on a ballot of
//     finished lanes;
//   * the Verlet loop is written against the issue rates measured on this GPU (tools/ubench/valu_rate.hip): full rate
//     for fma/mul/add/mov/bit ops, half rate for compares, selects, min/max, conversions, quarter rate for
//     rcp/rsq/sqrt - see degenerate_accumulate, acceleration_to_precision, sincos_reduced;
//   * no MFMA: the work is a 4x4 per-ray ODE, bound by fp32 VALU issue, not by HBM or matrix rate.
//
// No double-precision arithmetic on the hot path; the few double expressions of the reference's
// texture-space code (M_PI literals, cl.cl:3598-3610, 5272) are mirrored where they change results.

#define GR_PI 3.14159265358979323846
#define GR_PIf 3.14159274101257324f

struct lightray {
    float4 position;
    float4 velocity;
    float4 initial_quat;
    float4 acceleration;
    float ku_uobsu;
    float running_dlambda_dnew;
    int terminated;
    int sx;
    int sy;
};   // 96 bytes (render_state.hpp:8-19)

struct render_data {
    float2 tex_coord;
    float z_shift;
    int sx;
    int sy;
    int terminated;
    int side;
};   // 32 bytes (render_state.hpp:21-29)

struct dynamic_config {
#ifdef DYNVARS
    float DYNVARS;
#else
    float gr_unused;
#endif
};

struct dynamic_feature_config {
#ifdef DYNAMIC_FLOAT_FEATURES
    float DYNAMIC_FLOAT_FEATURES;
#endif
#ifdef DYNAMIC_BOOL_FEATURES
    int DYNAMIC_BOOL_FEATURES;
#endif
#if !defined(DYNAMIC_FLOAT_FEATURES) && !defined(DYNAMIC_BOOL_FEATURES)
    int gr_unused;
#endif
};

#ifdef KERNEL_IS_STATIC
#define GET_FEATURE(name, dfg) FEATURE_##name
#else
#define GET_FEATURE(name, dfg) ((dfg)->name)
#endif

#if defined(GENERIC_CONSTANT_THETA)
#define IS_CONSTANT_THETA
#endif

#ifndef GR_TILE
#define GR_TILE 8
#endif
// render_data.terminated of a pixel that gr_adaptive_refine wants traced (the reference's values are 0, 1, 2)
#define GR_PENDING (-1)
#define GR_TILE_CLASSES 16
#define GR_TILE_ORDER_HEADER (2 * GR_TILE_CLASSES)   // words in front of gr_order_tiles' list: class counts, class cursors
#ifndef GR_TILE_COST_REACH
#define GR_TILE_COST_REACH 1      // cells either side of the tile centre's whose rays' costs count for the tile's class
#endif
#ifndef GR_TILE_CLASS_STEPS
#define GR_TILE_CLASS_STEPS 1     // cost classes of gr_order_tiles per octave of attempts (finer ones measured no better)
#endif
#ifndef GR_AGING_PRIORITY
#define GR_AGING_PRIORITY 0      // experiment: waves that have integrated one tile for long rise in issue priority (measured: no effect)
#endif
#define GR_SKIP_CHUNK 32          // tiles of the last class per ticket

// minimum resident waves per SIMD the integrator kernels are register-allocated for (512 VGPRs / N waves each).
// 1 = no cap: the allocator takes what the metric's expressions need and occupancy follows (substituted Kerr: 92 VGPRs
// in the persistent fused kernel -> 5 waves/SIMD, which already saturates the VALU; the complex-valued double-Kerr
// metric: 186-370 VGPRs -> 1-2 waves/SIMD but no spills).  Measured on MI355X: forcing 6-8 waves on Kerr does not make it
// faster; capping double Kerr at 128 VGPRs costs 5.3x.
#ifndef GR_TRACE_WAVES
#define GR_TRACE_WAVES 1
#endif

// the same for gr_trace_fused alone: the host rebuilds a program with this set when that buys the kernel occupancy without
// spilling in its loop (capi.cpp compile_code_object)
#ifndef GR_FUSED_WAVES
#define GR_FUSED_WAVES GR_TRACE_WAVES
#endif

typedef const dynamic_config* __restrict__ cfg_t;
typedef const dynamic_feature_config* __restrict__ dfg_t;

// The integrator kernels copy the (wave-uniform) $cfg and feature blocks into registers once: the generated
// expressions say `cfg->NAME` inside the Verlet loop, and re-reading them through the pointer costs a scalar load
// plus an lgkmcnt wait per step.
#define GR_PARAMETERS_IN_REGISTERS                                   \
    const dynamic_config gr_cfg_registers = *cfg_in;                 \
    const dynamic_feature_config gr_dfg_registers = *dfg_in;         \
    const dynamic_config* const cfg = &gr_cfg_registers;             \
    const dynamic_feature_config* const dfg = &gr_dfg_registers;

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
#ifdef GR_PROBE_SOURCE_BREAKS
    if (POISON_LARGE) __builtin_amdgcn_s_setprio(0);
#endif
    return {s, c};
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
__device__ __forceinline__ float tanh(float x) { return ::tanhf(x); }
__device__ __forceinline__ float pow(float x, float y) { return ::powf(x, y); }
__device__ __forceinline__ float fmod(float x, float y) { return ::fmodf(x, y); }
__device__ __forceinline__ float fmin(float x, float y) { return __builtin_fminf(x, y); }
__device__ __forceinline__ float fmax(float x, float y) { return __builtin_fmaxf(x, y); }
__device__ __forceinline__ float sign(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

// --- generated-expression hosts (cl.cl:969-1355, 3377-3387) ---------------------------------------

#define GR_POSITION_VARS(p) \
    const float v1 = (p).x; const float v2 = (p).y; const float v3 = (p).z; const float v4 = (p).w; \
    const float rs = RS_IMPL; const float c = C_IMPL; (void)v1; (void)v2; (void)v3; (void)v4; (void)rs; (void)c;

// g_metric_out has 4 (diagonal) or 16 entries
__device__ __forceinline__ void metric_at(float4 pos, float* g, cfg_t cfg) {
    GR_POSITION_VARS(pos)
    float TEMPORARIES0;
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
    float TEMPORARIES0;
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
    (void)sin; (void)cos;
#endif
#ifdef GR_PROBE_SOURCE_BREAKS   // experiment: a scalar instruction the scheduler cannot move at the seams of the acceleration
#define GR_ISSUE_BREAK __builtin_amdgcn_s_setprio(0);
#else
#define GR_ISSUE_BREAK
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
    GR_ISSUE_BREAK
    float TEMPORARIES0;
    GR_ISSUE_BREAK
    float4 a;
    a.x = GEO_ACCEL0;
    GR_ISSUE_BREAK
    a.y = GEO_ACCEL1;
    GR_ISSUE_BREAK
#ifndef GENERIC_CONSTANT_THETA
    a.z = GEO_ACCEL2;
    GR_ISSUE_BREAK
#else
    a.z = 0.f;
#endif
    a.w = GEO_ACCEL3;
    GR_ISSUE_BREAK
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
    float TEMPORARIES0;
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

__device__ __forceinline__ float distance_to_object(float4 polar, cfg_t cfg) {
    GR_POSITION_VARS(polar)
    return DISTANCE_FUNC;
}

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
    pairf TEMPORARIES0;
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
#endif  // GR_TWO_RAYS_PER_LANE

}  // namespace gm

// ------------------------------------------------------------------------------------------------
// small vector helpers

__device__ __forceinline__ float3 f3(float x, float y, float z) { return make_float3(x, y, z); }
__device__ __forceinline__ float4 f4(float x, float y, float z, float w) { return make_float4(x, y, z, w); }
__device__ __forceinline__ float3 yzw(float4 v) { return make_float3(v.y, v.z, v.w); }
__device__ __forceinline__ float4 f4(float x, float3 v) { return make_float4(x, v.x, v.y, v.z); }
__device__ __forceinline__ float3 operator+(float3 a, float3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ float3 operator-(float3 a, float3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ float3 operator-(float3 a) { return f3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float3 operator*(float3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float3 operator*(float s, float3 a) { return f3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float3 operator/(float3 a, float s) { return f3(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ float4 operator+(float4 a, float4 b) { return f4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 operator-(float4 a, float4 b) { return f4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 operator-(float4 a) { return f4(-a.x, -a.y, -a.z, -a.w); }
__device__ __forceinline__ float4 operator*(float4 a, float s) { return f4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float4 operator*(float s, float4 a) { return f4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float4 operator/(float4 a, float s) { return f4(a.x / s, a.y / s, a.z / s, a.w / s); }
__device__ __forceinline__ float dot3(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ float3 cross3(float3 a, float3 b) {
    return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float length3(float3 a) { return __builtin_sqrtf(dot3(a, a)); }
__device__ __forceinline__ float3 normalize3(float3 a) { return a / length3(a); }
__device__ __forceinline__ float4 normalize4(float4 a) { return a / __builtin_sqrtf(dot4(a, a)); }
__device__ __forceinline__ float fsign(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }
__device__ __forceinline__ float clampf(float v, float lo, float hi) { return __builtin_fminf(__builtin_fmaxf(v, lo), hi); }
__device__ __forceinline__ float mixf(float a, float b, float t) { return a + (b - a) * t; }
__device__ __forceinline__ bool degenerate(float x) { return !(__builtin_fabsf(x) <= 3.402823466e+38f); }   // NaN or Inf
__device__ __forceinline__ bool degenerate4(float4 v) { return degenerate(v.x) || degenerate(v.y) || degenerate(v.z) || degenerate(v.w); }
// x * 0 is 0 for finite x and NaN for +-inf / NaN, so the accumulated sum is NaN exactly when a component is degenerate.
// Inside the Verlet loop this replaces four half-rate v_cmp_class per vector (plus the mask plumbing) by four full-rate
// v_fma and one compare for all vectors together.  (No -ffinite-math-only: the compiler may not fold x * 0.)
__device__ __forceinline__ float degenerate_accumulate(float4 v, float acc) {
    return __builtin_fmaf(v.x, 0.f, __builtin_fmaf(v.y, 0.f, __builtin_fmaf(v.z, 0.f, __builtin_fmaf(v.w, 0.f, acc))));
}
__device__ __forceinline__ float get4(float4 v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
__device__ __forceinline__ void swap4(float4& a, float4& b) { float4 t = a; a = b; b = t; }

// cl.cl:103-140, 185-204
__device__ __forceinline__ float3 cartesian_to_polar(float3 in) {
    float r = length3(in);
    return f3(r, acosf(in.z / r), atan2f(in.y, in.x));
}

__device__ __forceinline__ float3 polar_to_cartesian(float3 in) {
    float st = sinf(in.y), ct = cosf(in.y), sp = sinf(in.z), cp = cosf(in.z);
    return f3(in.x * st * cp, in.x * st * sp, in.x * ct);
}

__device__ __forceinline__ float3 cartesian_velocity_to_polar_velocity(float3 p, float3 v) {
    float r = length3(p);
    float repeated_eq = r * __builtin_sqrtf(1 - (p.z * p.z / (r * r)));
    float rdot = (p.x * v.x + p.y * v.y + p.z * v.z) / r;
    float tdot = ((p.z * rdot) / (r * repeated_eq)) - v.z / repeated_eq;
    float pdot = (p.x * v.y - p.y * v.x) / (p.x * p.x + p.y * p.y);
    return f3(rdot, tdot, pdot);
}

__device__ __forceinline__ float3 spherical_velocity_to_cartesian_velocity(float3 p, float3 dp) {
    float r = p.x, dr = dp.x, x = p.y, dx = dp.y, y = p.z, dy = dp.z;
    float sx = sinf(x), cx = cosf(x), sy = sinf(y), cy = cosf(y);
    float v1 = -r * sx * sy * dy + r * cx * cy * dx + sx * cy * dr;
    float v2 = sx * sy * dr + r * sx * cy * dy + r * cx * sy * dx;
    float v3 = cx * dr - r * sx * dx;
    return f3(v1, v2, v3);
}

// cl.cl:176-191
__device__ __forceinline__ float3 rot_quat_norm(float3 point, float4 q) {
    float3 qv = f3(q.x, q.y, q.z);
    float3 t = 2.f * cross3(qv, point);
    return point + q.w * t + cross3(qv, t);
}
__device__ __forceinline__ float3 rot_quat(float3 point, float4 q) { return rot_quat_norm(point, normalize4(q)); }

// ------------------------------------------------------------------------------------------------
// metric algebra on the full 4x4 (cl.cl:830-907)

__device__ __forceinline__ float4 lower_index_big(float4 v, const float* g) {
    return f4(g[0] * v.x + g[1] * v.y + g[2] * v.z + g[3] * v.w,
              g[4] * v.x + g[5] * v.y + g[6] * v.z + g[7] * v.w,
              g[8] * v.x + g[9] * v.y + g[10] * v.z + g[11] * v.w,
              g[12] * v.x + g[13] * v.y + g[14] * v.z + g[15] * v.w);
}
__device__ __forceinline__ float dot_big(float4 u, float4 v, const float* g) { return dot4(lower_index_big(u, g), v); }

// general 4x4 inverse by cofactors (role of matrix_inverse, cl.cl:560-683)
__device__ void matrix_inverse4(const float* m, float* out) {
    float s0 = m[0] * m[5] - m[4] * m[1];
    float s1 = m[0] * m[6] - m[4] * m[2];
    float s2 = m[0] * m[7] - m[4] * m[3];
    float s3 = m[1] * m[6] - m[5] * m[2];
    float s4 = m[1] * m[7] - m[5] * m[3];
    float s5 = m[2] * m[7] - m[6] * m[3];
    float c5 = m[10] * m[15] - m[14] * m[11];
    float c4 = m[9] * m[15] - m[13] * m[11];
    float c3 = m[9] * m[14] - m[13] * m[10];
    float c2 = m[8] * m[15] - m[12] * m[11];
    float c1 = m[8] * m[14] - m[12] * m[10];
    float c0 = m[8] * m[13] - m[12] * m[9];
    float det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
    float id = 1.0f / det;
    out[0] = (m[5] * c5 - m[6] * c4 + m[7] * c3) * id;
    out[1] = (-m[1] * c5 + m[2] * c4 - m[3] * c3) * id;
    out[2] = (m[13] * s5 - m[14] * s4 + m[15] * s3) * id;
    out[3] = (-m[9] * s5 + m[10] * s4 - m[11] * s3) * id;
    out[4] = (-m[4] * c5 + m[6] * c2 - m[7] * c1) * id;
    out[5] = (m[0] * c5 - m[2] * c2 + m[3] * c1) * id;
    out[6] = (-m[12] * s5 + m[14] * s2 - m[15] * s1) * id;
    out[7] = (m[8] * s5 - m[10] * s2 + m[11] * s1) * id;
    out[8] = (m[4] * c4 - m[5] * c2 + m[7] * c0) * id;
    out[9] = (-m[0] * c4 + m[1] * c2 - m[3] * c0) * id;
    out[10] = (m[12] * s4 - m[13] * s2 + m[15] * s0) * id;
    out[11] = (-m[8] * s4 + m[9] * s2 - m[11] * s0) * id;
    out[12] = (-m[4] * c3 + m[5] * c1 - m[6] * c0) * id;
    out[13] = (m[0] * c3 - m[1] * c1 + m[2] * c0) * id;
    out[14] = (-m[12] * s3 + m[13] * s1 - m[14] * s0) * id;
    out[15] = (m[8] * s3 - m[9] * s1 + m[10] * s0) * id;
}

// ------------------------------------------------------------------------------------------------
// tetrads (cl.cl:1647-1861, 2072-2114, 2210-2224, 2288-2439)

struct tetrad {
    float4 e[4];
};

__device__ __forceinline__ float4 gram_project(float4 u, float4 v, const float* g) {
    return (dot_big(u, v, g) / dot_big(u, u, g)) * u;
}

__device__ __forceinline__ float4 normalise_metric(float4 v, const float* g) {
    return v / __builtin_sqrtf(__builtin_fabsf(dot_big(v, v, g)));
}

// returns the timelike slot found; fills `out` (cl.cl:1761-1850)
__device__ int frame_basis_with_swap(const float* g, int index_swap, tetrad& out) {
    float4 arr[4] = {f4(1, 0, 0, 0), f4(0, 1, 0, 0), f4(0, 0, 1, 0), f4(0, 0, 0, 1)};
    float lengths[4] = {g[0], g[5], g[10], g[15]};
    {
        float4 t = arr[0]; arr[0] = arr[index_swap]; arr[index_swap] = t;
        float l = lengths[0]; lengths[0] = lengths[index_swap]; lengths[index_swap] = l;
    }
    int indices[4] = {0, 1, 2, 3};
    int first_nonzero = -1;
    const float eps = 0.00001f;
    for (int i = 0; i < 4; i++) {
        if (!(__builtin_fabsf(lengths[i]) <= eps)) { first_nonzero = i; break; }
    }
    if (first_nonzero == -1) first_nonzero = 0;
    if (first_nonzero != 0) {
        float4 t = arr[0]; arr[0] = arr[first_nonzero]; arr[first_nonzero] = t;
        int q = indices[0]; indices[0] = indices[first_nonzero]; indices[first_nonzero] = q;
    }
    // Gram-Schmidt in the metric (cl.cl:1647-1675)
    float4 u1 = arr[0];
    float4 u2 = arr[1];
    u2 = u2 - gram_project(u1, u2, g);
    float4 u3 = arr[2];
    u3 = u3 - gram_project(u1, u3, g);
    u3 = u3 - gram_project(u2, u3, g);
    float4 u4 = arr[3];
    u4 = u4 - gram_project(u1, u4, g);
    u4 = u4 - gram_project(u2, u4, g);
    u4 = u4 - gram_project(u3, u4, g);
    float4 res[4] = {normalise_metric(u1, g), normalise_metric(u2, g), normalise_metric(u3, g), normalise_metric(u4, g)};
    float4 sorted[4];
    for (int i = 0; i < 4; i++) {
        int old_index = indices[i];
        for (int k = 0; k < 4; k++)
            if (k == old_index) sorted[k] = res[i];
    }
    // which leg is timelike: most negative e_a.e_a (cl.cl:1713-1758)
    int lowest = -1;
    float lowest_value = 0;
    for (int i = 0; i < 4; i++) {
        float d = 0;
        for (int k = 0; k < 4; k++)
            if (k == i) d = dot_big(sorted[k], sorted[k], g);
        if (d < lowest_value) { lowest = i; lowest_value = d; }
    }
    int which = lowest != -1 ? lowest : 0;
    if (which > 0) {
        for (int k = 1; k < 4; k++)
            if (k == which) swap4(sorted[0], sorted[k]);
    }
    for (int i = 0; i < 4; i++) out.e[i] = sorted[i];
    return which;
}

__device__ void frame_basis(const float* g, tetrad& out) {
    int t = frame_basis_with_swap(g, 0, out);
    if (t == 0) return;
    frame_basis_with_swap(g, t, out);
}

// 3-d Gram-Schmidt (cl.cl:1549-1566)
__device__ __forceinline__ float3 project3(float3 u, float3 v) { return (dot3(u, v) / dot3(u, u)) * u; }

__device__ void calculate_tetrads(float4 at_metric, float3 basis_speed, tetrad& out, cfg_t cfg, int should_orient) {
    float4 polar_camera = gm::generic_to_spherical(at_metric, cfg);
    if (degenerate4(at_metric)) {
        out.e[0] = f4(1, 0, 0, 0); out.e[1] = f4(0, 1, 0, 0); out.e[2] = f4(0, 0, 1, 0); out.e[3] = f4(0, 0, 0, 1);
        return;
    }
    float g[16];
    gm::metric_big_at(at_metric, g, cfg);
    tetrad t;
    frame_basis(g, t);
    float4 e0 = t.e[0], e1 = t.e[1], e2 = t.e[2], e3 = t.e[3];

    if (should_orient) {
        // align the spatial legs with the global cartesian axes, y first (cl.cl:2329-2412)
        float3 apolar = yzw(polar_camera);
        apolar.x = __builtin_fabsf(apolar.x);
        float3 cart_camera = polar_to_cartesian(apolar);

        float m[16] = {e0.x, e1.x, e2.x, e3.x, e0.y, e1.y, e2.y, e3.y, e0.z, e1.z, e2.z, e3.z, e0.w, e1.w, e2.w, e3.w};
        float inv[16];
        matrix_inverse4(m, inv);
        float4 lo0 = f4(inv[0], inv[1], inv[2], inv[3]);
        float4 lo1 = f4(inv[4], inv[5], inv[6], inv[7]);
        float4 lo2 = f4(inv[8], inv[9], inv[10], inv[11]);
        float4 lo3 = f4(inv[12], inv[13], inv[14], inv[15]);

        float3 sx = cartesian_velocity_to_polar_velocity(cart_camera, f3(1, 0, 0));
        float3 sy = cartesian_velocity_to_polar_velocity(cart_camera, f3(0, 1, 0));
        float3 sz = cartesian_velocity_to_polar_velocity(cart_camera, f3(0, 0, 1));
        if (polar_camera.y < 0) { sx.x = -sx.x; sy.x = -sy.x; sz.x = -sz.x; }

        float4 gx = gm::spherical_velocity_to_generic_velocity(polar_camera, f4(0, sx), cfg);
        float4 gy = gm::spherical_velocity_to_generic_velocity(polar_camera, f4(0, sy), cfg);
        float4 gz = gm::spherical_velocity_to_generic_velocity(polar_camera, f4(0, sz), cfg);

        // coordinate -> tetrad components; order y, x, z
        float4 tE1 = f4(dot4(lo0, gy), dot4(lo1, gy), dot4(lo2, gy), dot4(lo3, gy));
        float4 tE2 = f4(dot4(lo0, gx), dot4(lo1, gx), dot4(lo2, gx), dot4(lo3, gx));
        float4 tE3 = f4(dot4(lo0, gz), dot4(lo1, gz), dot4(lo2, gz), dot4(lo3, gz));

        float3 u1 = yzw(tE1), u2 = yzw(tE2), u3 = yzw(tE3);
        u2 = u2 - project3(u1, u2);
        u3 = u3 - project3(u1, u3);
        u3 = u3 - project3(u2, u3);
        u1 = normalize3(u1); u2 = normalize3(u2); u3 = normalize3(u3);

        // x_basis = second, y_basis = first, z_basis = third; back to coordinates with the original legs
        float4 x_out = u2.x * e1 + u2.y * e2 + u2.z * e3;
        float4 y_out = u1.x * e1 + u1.y * e2 + u1.z * e3;
        float4 z_out = u3.x * e1 + u3.y * e2 + u3.z * e3;
        e1 = x_out; e2 = y_out; e3 = z_out;
    }

    {
        // boost into the observer's frame (cl.cl:2414-2433, 1919-1972, 2210-2224)
        float v2 = dot3(basis_speed, basis_speed);
        float Y = 1 / __builtin_sqrtf(1 - v2);
        float4 observer_velocity = Y * e0 + (Y * basis_speed.x) * e1 + (Y * basis_speed.y) * e2 + (Y * basis_speed.z) * e3;

        float4 lT4 = lower_index_big(e0, g);
        float4 lu4 = lower_index_big(observer_velocity, g);
        float T[4] = {e0.x, e0.y, e0.z, e0.w};
        float lT[4] = {lT4.x, lT4.y, lT4.z, lT4.w};
        float uo[4] = {observer_velocity.x, observer_velocity.y, observer_velocity.z, observer_velocity.w};
        float luo[4] = {lu4.x, lu4.y, lu4.z, lu4.w};
        float lorentz_factor = -dot4(lT4, observer_velocity);
        float L[16];
        for (int u = 0; u < 4; u++)
            for (int v = 0; v < 4; v++)
                L[u * 4 + v] = (u == v ? 1.f : 0.f) + (1 / (1 + lorentz_factor)) * (T[u] + uo[u]) * (lT[v] + luo[v]) - 2 * uo[u] * lT[v];
        e0 = observer_velocity;
        e1 = f4(dot4(f4(L[0], L[1], L[2], L[3]), e1), dot4(f4(L[4], L[5], L[6], L[7]), e1), dot4(f4(L[8], L[9], L[10], L[11]), e1), dot4(f4(L[12], L[13], L[14], L[15]), e1));
        e2 = f4(dot4(f4(L[0], L[1], L[2], L[3]), e2), dot4(f4(L[4], L[5], L[6], L[7]), e2), dot4(f4(L[8], L[9], L[10], L[11]), e2), dot4(f4(L[12], L[13], L[14], L[15]), e2));
        e3 = f4(dot4(f4(L[0], L[1], L[2], L[3]), e3), dot4(f4(L[4], L[5], L[6], L[7]), e3), dot4(f4(L[8], L[9], L[10], L[11]), e3), dot4(f4(L[12], L[13], L[14], L[15]), e3));
    }
    out.e[0] = e0; out.e[1] = e1; out.e[2] = e2; out.e[3] = e3;
}

// ------------------------------------------------------------------------------------------------
// ray set-up (cl.cl:2015-2059, 2949-3065)

__device__ __forceinline__ float3 pixel_direction(int cx, int cy, float width, float height, float4 camera_quat, dfg_t dfg) {
    float fov = GET_FEATURE(field_of_view, dfg);
    float fov_rad = (fov / 360.f) * 2 * GR_PIf;
    float f_stop = (width / 2) / tanf(fov_rad / 2);
    float3 dir = normalize3(f3(cx - width / 2, cy - height / 2, f_stop));
    return rot_quat(dir, camera_quat);
}

#ifdef GENERIC_CONSTANT_THETA
__device__ float4 theta_adjustment_quat(float3 pixel_dir, float4 polar_camera, float angle_sign) {
    if (length3(pixel_dir) < 0.00001f) pixel_dir = f3(0, 1, 0);
    float3 apolar = yzw(polar_camera);
    apolar.x = __builtin_fabsf(apolar.x);
    float3 cam = polar_to_cartesian(apolar);
    float3 bx = normalize3(pixel_dir);
    float3 by = normalize3(-cam);
    bx = normalize3(normalize3(bx - dot3(bx, by) * by));
    float3 plane_n = -normalize3(cross3(bx, by));
    float angle_to_flat = acosf(dot3(plane_n, f3(0, 0, 1)));
    float3 axis = normalize3(cross3(plane_n, f3(0, 0, 1)));
    float angle = angle_to_flat * angle_sign;
    float s = sinf(angle / 2);
    return normalize4(f4(axis.x * s, axis.y * s, axis.z * s, cosf(angle / 2)));
}
#endif

// rotates the ray into the equatorial plane for spherically symmetric metrics; identity otherwise
__device__ __forceinline__ void correct_lightray(float4& position, float4& velocity, float4& inverse_quat, cfg_t cfg) {
    inverse_quat = f4(0, 0, 0, 1);
#ifdef GENERIC_CONSTANT_THETA
    float4 polar_pos = gm::generic_to_spherical(position, cfg);
    float4 pos_sph = polar_pos;
    float4 vel_sph = gm::generic_velocity_to_spherical_velocity(position, velocity, cfg);
    float sgn = fsign(pos_sph.y);
    pos_sph.y = __builtin_fabsf(pos_sph.y);
    float3 pos_cart = polar_to_cartesian(yzw(pos_sph));
    float3 vel_cart = spherical_velocity_to_cartesian_velocity(yzw(pos_sph), yzw(vel_sph));
    float4 quat = theta_adjustment_quat(vel_cart, polar_pos, 1);
    inverse_quat = theta_adjustment_quat(vel_cart, polar_pos, -1);
    pos_cart = rot_quat(pos_cart, quat);
    vel_cart = rot_quat(vel_cart, quat);
    float3 next_pos = cartesian_to_polar(pos_cart);
    float3 next_vel = cartesian_velocity_to_polar_velocity(pos_cart, vel_cart);
    if (sgn < 0) next_pos.x = -next_pos.x;
    position = gm::spherical_to_generic(f4(pos_sph.x, next_pos), cfg);
    velocity = gm::spherical_velocity_to_generic_velocity(f4(pos_sph.x, next_pos), f4(vel_sph.x, next_vel), cfg);
#endif
}

// full initial state of one primary ray (geodesic_to_render_ray, cl.cl:3000-3065).  The initial
// acceleration is the same -Gamma v v the reference contracts numerically from F*_P
// (cl.cl:738-797, 1443-1537); here it is evaluated through the closed form GEO_ACCELn.
__device__ __forceinline__ lightray make_render_ray(int cx, int cy, float4 position, float4 velocity, float4 observer_velocity, cfg_t cfg) {
    lightray ray;
    correct_lightray(position, velocity, ray.initial_quat, cfg);
#ifdef IS_CONSTANT_THETA
    position.z = GR_PIf / 2;
    velocity.z = 0;
#endif
    ray.position = position;
    ray.velocity = velocity;
    ray.acceleration = gm::geodesic_acceleration(position, velocity, cfg);
    ray.running_dlambda_dnew = 1;
    ray.terminated = 0;
    {
        float g[16];
        gm::metric_big_at(position, g, cfg);
        ray.ku_uobsu = dot4(velocity, lower_index_big(observer_velocity, g));
    }
    ray.sx = cx;
    ray.sy = cy;
    return ray;
}

__device__ __forceinline__ lightray make_pixel_ray(int cx, int cy, int width, int height, float4 camera, float4 camera_quat,
                                                   float4 e0, float4 e1, float4 e2, float4 e3, int flip, cfg_t cfg, dfg_t dfg) {
    float3 dir = normalize3(pixel_direction(cx, cy, (float)width, (float)height, camera_quat, dfg));
#ifndef FORWARD_GEODESIC_PATH
    float4 pixel_t = -e0;
#else
    float4 pixel_t = e0;
#endif
    if (flip) pixel_t = -pixel_t;
    float4 velocity = dir.x * e1 + dir.y * e2 + dir.z * e3 + pixel_t;
    return make_render_ray(cx, cy, camera, velocity, e0, cfg);
}

// ray slot -> pixel.  Linear (reference order, cl.cl:3159-3160) or GR_TILE x GR_TILE tiles so that the
// 64 lanes of a wave own one compact pixel block.
__device__ __forceinline__ bool slot_to_pixel(int id, int width, int height, int tiled, int& cx, int& cy) {
    if (!tiled) {
        cx = id % width;
        cy = id / width;
        return id < width * height;
    }
    const int T = GR_TILE;
    int tiles_x = (width + T - 1) / T;
    int tile = id / (T * T);
    int in = id % (T * T);
    cx = (tile % tiles_x) * T + in % T;
    cy = (tile / tiles_x) * T + in / T;
    return cx < width && cy < height;
}

// (float)a / b rounded as IEEE division does.  The kernels are built with approximate fp32 division (v_rcp_f32); this
// quotient feeds round() to pick prepass cells (cl.cl:3217-3221), where a 1-ulp difference moves the stencil.
__device__ __forceinline__ float exact_ratio(int a, int b) {
#pragma float_control(precise, on)
    return (float)((double)a / (double)b);
}

__device__ __forceinline__ int early_terminate(int x, int y, int w, int h, const int* __restrict__ term) {
    if (x < 0 || y < 0 || x > w - 1 || y > h - 1) return 0;
    return term[y * w + x] == 1;
}
// the 5-point stencil of init_rays_generic (cl.cl:3213-3232) with all five cells read at once - clamped coordinates, the verdict
// of a cell outside the grid discarded afterwards - instead of a chain of conditional loads: a skipped tile is nothing but these
// loads' latency (measured 45 us per skipped tile with the chain, when few other waves are left to hide it)
__device__ __forceinline__ bool early_terminate_stencil(int lx, int ly, int w, int h, const int* __restrict__ term) {
    const int x0 = min(max(lx - 1, 0), w - 1), x1 = min(max(lx, 0), w - 1), x2 = min(max(lx + 1, 0), w - 1);
    const int y0 = min(max(ly - 1, 0), h - 1), y1 = min(max(ly, 0), h - 1), y2 = min(max(ly + 1, 0), h - 1);
    const int left = term[y1 * w + x0], centre = term[y1 * w + x1], right = term[y1 * w + x2], up = term[y0 * w + x1], down = term[y2 * w + x1];
    const bool inside = lx - 1 >= 0 && lx + 1 <= w - 1 && ly - 1 >= 0 && ly + 1 <= h - 1;   // any cell outside: not skipped
    return inside & (left == 1) & (centre == 1) & (right == 1) & (up == 1) & (down == 1);
}

// The same verdict while the prepass is still running in the SAME launch (gr_trace_fused with prepass_tickets > 0: the first
// tickets of the persistent launch are the prepass cells, 64 to a wave; the buffer was filled with GR_CELL_UNKNOWN before the
// launch).  A lane reads its five cells with device-scope loads until none of them is unknown; the wave sleeps between rounds.
// This cannot hang: tickets are handed out in order, so every prepass ticket is held by a running wave before the first tile
// ticket is drawn, and prepass waves wait for nothing.
#define GR_CELL_UNKNOWN (-1)
__device__ __forceinline__ bool early_terminate_stencil_when_known(int lx, int ly, int w, int h, const int* term) {
    const int x0 = min(max(lx - 1, 0), w - 1), x1 = min(max(lx, 0), w - 1), x2 = min(max(lx + 1, 0), w - 1);
    const int y0 = min(max(ly - 1, 0), h - 1), y1 = min(max(ly, 0), h - 1), y2 = min(max(ly + 1, 0), h - 1);
    int left, centre, right, up, down;
    for (;;) {
        left = __hip_atomic_load(term + y1 * w + x0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        centre = __hip_atomic_load(term + y1 * w + x1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        right = __hip_atomic_load(term + y1 * w + x2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        up = __hip_atomic_load(term + y0 * w + x1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        down = __hip_atomic_load(term + y2 * w + x1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool unknown = (left | centre | right | up | down) < 0;   // GR_CELL_UNKNOWN is the only negative value
        if (__builtin_amdgcn_ballot_w64(unknown) == 0) break;
        __builtin_amdgcn_s_sleep(64);   // ~4 k cycles: a prepass ray takes 10^5..10^6
    }
    const bool inside = lx - 1 >= 0 && lx + 1 <= w - 1 && ly - 1 >= 0 && ly + 1 <= h - 1;
    return inside & (left == 1) & (centre == 1) & (right == 1) & (up == 1) & (down == 1);
}

// ------------------------------------------------------------------------------------------------
// the integrator (cl.cl:3273-3346, 3400-3456, 3954-4247)

#ifdef ADAPTIVE_PRECISION
#define GR_W_MAX ((float)((W_V1 > W_V2 ? W_V1 : W_V2) > (W_V3 > W_V4 ? W_V3 : W_V4) ? (W_V1 > W_V2 ? W_V1 : W_V2) : (W_V3 > W_V4 ? W_V3 : W_V4)))

// returns diff, writes the unclamped step suggestion (acceleration_to_precision)
__device__ __forceinline__ float acceleration_to_precision(float4 acc, float max_acceleration, float& next_ds) {
    float4 wa = f4(acc.x * (float)(W_V1), acc.y * (float)(W_V2), acc.z * (float)(W_V3), acc.w * (float)(W_V4));
    float current = __builtin_sqrtf(dot4(wa, wa)) * 0.01f;
    current /= GR_W_MAX;
    const float scale = 65536.f;                 // I_HATE_COMPUTERS
    float err = max_acceleration;
    float diff = current * scale;
    float floor_diff = err * scale / 1e10f;      // pow(max_timestep = 100000, 2)
    if (diff < floor_diff) diff = floor_diff;
    next_ds = __builtin_sqrtf(err * scale) * __builtin_amdgcn_rsqf(diff);   // sqrt((err * scale) / diff); first factor is loop-invariant
    return diff;
}
#endif

enum { DS_NONE = 0, DS_SKIP = 1, DS_RETURN = 2 };

struct ray_state {
    float4 position, velocity, acceleration;
    float next_ds;
    float running_dlambda_dnew;
    float f_in_x;
    // progress of a ray that is integrated in several visits (ray compaction): accepted steps, attempts
    int steps;
    unsigned int tries;
};

// outcome of integrating one ray
enum { RAY_LOST = 0, RAY_TERMINATED = 1 };

// Integrates until termination.  Returns RAY_TERMINATED when the ray reached the outer boundary (or the
// SINGULAR terminator) - position/velocity/running_dlambda_dnew are then final - and RAY_LOST on any
// early return of the reference (singularity guards, NaN, step cap), where nothing is written back.
//
// RESUMABLE (ray compaction): the loop also stops - `paused` - as soon as fewer than keep_lanes lanes of the wave are still
// iterating, with everything it carries between iterations saved in `s`, so that the caller can hand the idle lanes new rays
// and call again; integrate_begin prepares `s` for the first visit.  The arithmetic of a ray does not depend on the visits.
__device__ __forceinline__ void integrate_begin(ray_state& s, dfg_t dfg) {
    s.f_in_x = __builtin_fabsf(s.velocity.x);
    s.next_ds = 0.00001f;
#ifdef ADAPTIVE_PRECISION
    (void)acceleration_to_precision(s.acceleration, GET_FEATURE(max_acceleration_change, dfg), s.next_ds);
#endif
    s.running_dlambda_dnew = 1;
    s.steps = 0;
    s.tries = 0;
}

template <bool RESUMABLE>
__device__ __forceinline__ int integrate_core(ray_state& s, cfg_t cfg, dfg_t dfg, unsigned int* attempts, int keep_lanes, bool& paused) {
    float4 position = s.position, velocity = s.velocity, acceleration = s.acceleration;
    const float f_in_x = RESUMABLE ? s.f_in_x : __builtin_fabsf(velocity.x);
#ifdef IS_CONSTANT_THETA
    position.z = GR_PIf / 2; velocity.z = 0; acceleration.z = 0;
#endif
    float next_ds = 0.00001f;
#ifdef ADAPTIVE_PRECISION
    const float max_accel = GET_FEATURE(max_acceleration_change, dfg);
    const float min_step = GET_FEATURE(min_step, dfg);
    if (RESUMABLE) next_ds = s.next_ds;
    else (void)acceleration_to_precision(acceleration, max_accel, next_ds);
#endif
    const float subambient_precision = 0.5f;
    const float ambient_precision = 0.2f;
    const float new_max = GET_FEATURE(max_precision_radius, dfg);
    const float new_min = 3;
    const float universe = GET_FEATURE(universe_size, dfg);
    const bool reparam = GET_FEATURE(reparameterisation, dfg) != 0;
    float running = RESUMABLE ? s.running_dlambda_dnew : 1.f;
    const int loop_limit = 4096 * 4;
    unsigned int tries = RESUMABLE ? s.tries : 0u;
    int result = RAY_LOST;
    paused = false;

    int i = RESUMABLE ? s.steps : 0;
    // The reference leaves the loop at its top in four ways (step cap, cylindrical singularity, runaway guard: lost; boundary
    // reached: terminated; cl.cl:3990-4060).  Here they are one combined exit - one exec-mask update per attempt instead of
    // four (scalar instructions are ~10 % of the loop's time on this GPU) - and the outcome is read off the final state after the
    // loop.  The other exits (singularity detection, degenerate values) leave a state on which these tests say "lost" as well:
    // a pre-step state that has just passed them, or NaN/Inf, which no comparison accepts.
    auto stop_lost = [&](float4 pos, float4 vel, float4 acc, float run, int steps) {
        bool lost = steps >= loop_limit;
#ifdef HAS_CYLINDRICAL_SINGULARITY
        lost |= pos.y < CYLINDRICAL_TERMINATOR;
#endif
#ifndef UNCONDITIONALLY_NONSINGULAR
        lost |= __builtin_fabsf(vel.x / run) > 1000 + f_in_x && __builtin_fabsf(acc.x / run) > 100;
#endif
        (void)pos; (void)vel; (void)acc; (void)run;
        return lost;
    };
    auto stop_terminated = [&](float4 polar) {
        bool t = __builtin_fabsf(polar.y) >= universe;
#ifdef SINGULAR
        t |= __builtin_fabsf(polar.y) < SINGULAR_TERMINATOR;
#endif
        return t;
    };
    for (;;) {
#ifdef IS_CONSTANT_THETA
        position.z = GR_PIf / 2; velocity.z = 0; acceleration.z = 0;
#endif
        float4 polar = gm::generic_to_spherical(position, cfg);
#ifdef IS_CONSTANT_THETA
        polar.z = GR_PIf / 2;
#endif
        float r_value = gm::distance_to_object(polar, cfg);
        float ar = __builtin_fabsf(r_value);
        float ds = mixf(ambient_precision, subambient_precision, (clampf(ar, new_min, new_max) - new_min) / (new_max - new_min));
#ifdef ADAPTIVE_PRECISION
        ds = next_ds;
#endif
        if (ar < new_max) ds = __builtin_fminf(ds, ambient_precision);
        else ds = 0.1f * (ar - new_max) + ambient_precision;

        if (stop_lost(position, velocity, acceleration, running, i) | stop_terminated(polar)) break;

        // velocity Verlet (step_verlet)
        tries++;
#if defined(GR_PROBE_VALU) || defined(GR_PROBE_SALU)
        // bottleneck probes (tools/README.md): extra independent instructions per attempt; does the frame time follow?
        {
#ifdef GR_PROBE_VALU
            float probe = ds;
#pragma unroll
            for (int q = 0; q < GR_PROBE_VALU; q++) asm volatile("v_mul_f32 %0, 0x3f8ccccd, %0" : "+v"(probe));
            asm volatile("" ::"v"(probe));
#endif
#ifdef GR_PROBE_SALU
            int sprobe = 1;
#pragma unroll
            for (int q = 0; q < GR_PROBE_SALU; q++) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sprobe) : : "scc");
            asm volatile("" ::"s"(sprobe));
#endif
        }
#endif
        const float half_ds = 0.5f * ds, half_ds2 = half_ds * ds;
        float4 next_position = position + velocity * ds + acceleration * half_ds2;
        float4 half_velocity = velocity + acceleration * ds;
#ifdef GR_PROBE_NO_ACCEL
        float4 next_acceleration = f4(acceleration.x * 0.999f, acceleration.y * 0.999f + half_velocity.x * 1e-3f, acceleration.z * 0.999f, acceleration.w * 0.999f + next_position.y * 1e-5f);
#else
        float4 next_acceleration = gm::geodesic_acceleration(next_position, half_velocity, cfg);
#endif
        float4 next_velocity = velocity + (acceleration + next_acceleration) * half_ds;
        float K = 1;
        if (reparam) {
            float md = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(next_velocity.x), __builtin_fabsf(next_velocity.y)),
                                       __builtin_fmaxf(__builtin_fabsf(next_velocity.z), __builtin_fabsf(next_velocity.w)));
            K = 1 / md;
            next_velocity = next_velocity * K;
            next_acceleration = next_acceleration * K * K;
        }
        running *= K;

        bool accept = true;
#ifdef ADAPTIVE_PRECISION
        if (ar < new_max) {
            // calculate_ds_error
            float suggested;
            float diff = acceleration_to_precision(next_acceleration, max_accel, suggested);
            // 0.99 * ds * clamp(suggested / ds, 0.3, 2) with ds > 0, without forming the quotient
            float nds = clampf(0.99f * suggested, (0.99f * 0.3f) * ds, (0.99f * 2.f) * ds);
            nds = __builtin_fmaxf(nds, min_step);
            next_ds = nds;
#ifdef SINGULARITY_DETECTION
            if (nds == min_step && (diff / 65536.f) > max_accel * 10000) break;
#endif
            accept = !(nds < ds / 1.95f);   // back-step: retry from the same state with the smaller step
        }
#endif
        if (accept) {
            position = next_position;
            velocity = next_velocity;
            acceleration = next_acceleration;
            i++;
            // IS_DEGENERATE of position, velocity, acceleration (cl.cl:4235-4244).  Without reparameterisation a
            // degenerate acceleration always makes the velocity computed from it degenerate, so two vectors suffice.
            float poison = degenerate_accumulate(position, degenerate_accumulate(velocity, 0.f));
            if (reparam) poison = degenerate_accumulate(acceleration, poison);
            if (!(poison == 0.f)) break;
        }
        if (RESUMABLE) {
            // lanes still in the loop = the exec mask; fewer than keep_lanes of them: leave and let the caller refill the wave
            if (__builtin_popcountll(__builtin_amdgcn_ballot_w64(true)) < keep_lanes) { paused = true; break; }
        }
    }
    if (!paused) {
        // why the loop was left (see above): the tests of the loop top on the final state, in the reference's order
        float4 polar = gm::generic_to_spherical(position, cfg);
#ifdef IS_CONSTANT_THETA
        polar.z = GR_PIf / 2;
#endif
        // ... and a state that is degenerate anywhere is the reference's plain `return` (cl.cl:4235-4244, terminated stays 0) even
        // when its position happens to lie beyond the boundary: a finite position can come with a NaN/Inf velocity or acceleration
        const bool finite = degenerate_accumulate(position, degenerate_accumulate(velocity, degenerate_accumulate(acceleration, 0.f))) == 0.f;
        if (!stop_lost(position, velocity, acceleration, running, i) && stop_terminated(polar) && finite) result = RAY_TERMINATED;
    }
    if (RESUMABLE) { s.next_ds = next_ds; s.steps = i; s.tries = tries; }
    s.position = position;
    s.velocity = velocity;
    s.acceleration = acceleration;
    s.running_dlambda_dnew = running;
    if (attempts) *attempts = tries;
    return result;
}


// ---- the integrator as it runs by default ---------------------------------------------------------------------------------
// The same algorithm written for the machine's costs.  Measured facts it is built on (MI355X, 4K Kerr): one more VALU
// instruction per attempt costs ~0.35 % of the frame, a scalar one ~0.15 %; a step is REJECTED once in ~35 000 attempts
// (oracle count, 192x108 Kerr: 357 of 12.7 M), so everything is arranged for the accepting path and a rejection may be slow.
//   * two attempts per trip, the state ping-ponging between two register sets: the accepted state is written where the next
//     attempt reads it, and only a rejection copies (the one-attempt loop paid 7 v_mov / v_xor per attempt to move the new
//     state into the loop-carried registers);
//   * the controller works on the squared, scaled error q = (|W a| 0.01 / Wmax 65536)^2: one multiply, one max against the
//     squared floor, suggestion = sqrt(err 65536) q^(-1/4) (v_sqrt, v_rsq), clamp by v_med3; the singularity test
//     (cl.cl:3446-3449) is one compare of q in the hot path, its second condition only behind it;
//   * IS_DEGENERATE (cl.cl:4235-4244) on the new velocity alone inside the loop - a non-finite acceleration makes the velocity
//     computed from it non-finite in the same step, a non-finite position needs a non-finite velocity first - and on all three
//     vectors once after the loop, where the outcome is decided;
//   * the step cap (cl.cl:3974: 16384 accepted steps) is the borrow of the subtraction that counts the steps down; the
//     attempt count the profiling launches ask for follows from it after the loop;
__device__ __forceinline__ float min_f32(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float max_f32(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// dst = src as instructions the compiler can neither turn into selects nor move out of the (rarely executed) block they are in
__device__ __forceinline__ void overwrite(float4& dst, float4 src) {
    asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                 : "+v"(dst.x), "+v"(dst.y), "+v"(dst.z), "+v"(dst.w) : "v"(src.x), "v"(src.y), "v"(src.z), "v"(src.w));
}

// the same with a wave-uniform first operand (a feature value, a literal): no VGPR is spent on it
__device__ __forceinline__ float min_f32_uniform(float uniform, float x) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "s"(uniform), "v"(x)); return r; }
__device__ __forceinline__ float max_f32_uniform(float uniform, float x) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "s"(uniform), "v"(x)); return r; }

#ifdef ADAPTIVE_PRECISION
struct step_controller {
    float floor_q, root, singular_q, min_step;
    __device__ __forceinline__ step_controller(float max_acceleration, float min_step_in) {
        const float scale = 65536.f;
        const float floor_diff = max_acceleration * scale / 1e10f;
        floor_q = floor_diff * floor_diff;
        root = __builtin_sqrtf(max_acceleration * scale);
        const float singular = max_acceleration * 10000 * scale;
        singular_q = singular * singular;
        min_step = min_step_in;
    }
    // diff^2 of acceleration_to_precision (cl.cl:3400-3429), floored
    __device__ __forceinline__ float error_q(float4 acc) const {
        const float k = 0.01f * 65536.f / GR_W_MAX;
        const float wx = acc.x * (float)(W_V1), wy = acc.y * (float)(W_V2), wz = acc.z * (float)(W_V3), ww = acc.w * (float)(W_V4);
        const float d2 = __builtin_fmaf(wx, wx, __builtin_fmaf(wy, wy, __builtin_fmaf(wz, wz, ww * ww)));
        return max_f32_uniform(floor_q, d2 * (k * k));   // a NaN error takes the floor: the step is then accepted and its velocity is caught as degenerate
    }
    __device__ __forceinline__ float suggestion(float q) const { return root * __builtin_amdgcn_rsqf(__builtin_sqrtf(q)); }
};
#endif

template <bool LIBM> struct trig_flavour { static constexpr bool value = LIBM; };

template <bool RESUMABLE>
__device__ __forceinline__ int integrate_pingpong(ray_state& s, cfg_t cfg, dfg_t dfg, unsigned int* attempts, int keep_lanes, bool& paused) {
    float4 p0 = s.position, v0 = s.velocity, a0 = s.acceleration;
    float4 p1 = p0, v1 = v0, a1 = a0;
    const float f_in_x = RESUMABLE ? s.f_in_x : __builtin_fabsf(v0.x);
    float next_ds = 0.00001f;
#ifdef ADAPTIVE_PRECISION
    const step_controller controller(GET_FEATURE(max_acceleration_change, dfg), GET_FEATURE(min_step, dfg));
    if (RESUMABLE) next_ds = s.next_ds;
    else {
        float4 a = a0;
#ifdef IS_CONSTANT_THETA
        a.z = 0;
#endif
        next_ds = controller.suggestion(controller.error_q(a));
    }
#endif
    const float subambient_precision = 0.5f;
    const float ambient_precision = 0.2f;
    const float new_max = GET_FEATURE(max_precision_radius, dfg);
    const float new_min = 3;
    const float universe = GET_FEATURE(universe_size, dfg);
    const bool reparam = GET_FEATURE(reparameterisation, dfg) != 0;
    float running = RESUMABLE ? s.running_dlambda_dnew : 1.f;
    const int loop_limit = 4096 * 4;
    // accepted steps the ray may still take (cl.cl:3974: 16384 in all).  Every attempt() entered takes one - the borrow of that very
    // subtraction is the step-cap test - and a rejection (rare) gives it back.
    const unsigned int budget_before = (unsigned int)(loop_limit - (RESUMABLE ? s.steps : 0));
    unsigned int budget = budget_before;
    unsigned int rejections = 0;
    paused = false;

    auto stop_lost = [&](float4 pos, float4 vel, float4 acc, float run) {
        bool lost = false;
#ifdef HAS_CYLINDRICAL_SINGULARITY
        lost |= pos.y < CYLINDRICAL_TERMINATOR;
#endif
#ifndef UNCONDITIONALLY_NONSINGULAR
        lost |= __builtin_fabsf(vel.x / run) > 1000 + f_in_x && __builtin_fabsf(acc.x / run) > 100;
#endif
        (void)pos; (void)vel; (void)acc; (void)run;
        return lost;
    };
    auto stop_terminated = [&](float4 polar) {
        bool t = __builtin_fabsf(polar.y) >= universe;
#ifdef SINGULAR
        t |= __builtin_fabsf(polar.y) < SINGULAR_TERMINATOR;
#endif
        return t;
    };
    const float far_offset = ambient_precision - 0.1f * new_max;
    // One Verlet attempt from (p, v, a): the state the next attempt starts from goes to (po, vo, ao) - the new state, or the old
    // one again after a rejection.  Returns true when the loop is to be left: the ray is done and (p, v, a) is its final state
    // (what the loop carries - step suggestion, budget - is then that of the abandoned attempt); or pause_wave was raised
    // (RESUMABLE): the step was taken, the wave wants new rays, the state is (po, vo, ao).
    bool pause_wave = false;
    auto attempt = [&](auto libm, float4 position, float4 velocity, float4 acceleration, float4& po, float4& vo, float4& ao, float& ds_used,
                       float& running_before) -> bool {
        // the for-loop condition of the reference, then its loop-top exits (cl.cl:3974, 4086-4130)
        if (__builtin_expect(__builtin_usub_overflow(budget, 1u, &budget), 0)) return true;
#ifdef IS_CONSTANT_THETA
        position.z = GR_PIf / 2; velocity.z = 0; acceleration.z = 0;
#endif
        float4 polar = gm::generic_to_spherical(position, cfg);
#ifdef IS_CONSTANT_THETA
        polar.z = GR_PIf / 2;
#endif
        const float ar = __builtin_fabsf(gm::distance_to_object(polar, cfg));
        const bool inside = ar < new_max;
#ifdef ADAPTIVE_PRECISION
        const float near_ds = min_f32_uniform(ambient_precision, next_ds);
#else
        const float near_ds = min_f32_uniform(ambient_precision, mixf(ambient_precision, subambient_precision, (clampf(ar, new_min, new_max) - new_min) / (new_max - new_min)));
#endif
        const float far_ds = __builtin_fmaf(0.1f, ar, far_offset);   // 0.1 (|r| - max_precision_radius) + ambient
        const float ds = inside ? near_ds : far_ds;
        ds_used = ds;
        running_before = running;
        if (stop_lost(position, velocity, acceleration, running) | stop_terminated(polar)) return true;
#if defined(GR_PROBE_VALU) || defined(GR_PROBE_SALU)
        // bottleneck probes (tools/README.md): extra independent instructions per attempt; does the frame time follow?
        {
#ifdef GR_PROBE_VALU
            float probe = ds;
#pragma unroll
            for (int q = 0; q < GR_PROBE_VALU; q++) asm volatile("v_mul_f32 %0, 0x3f8ccccd, %0" : "+v"(probe));
            asm volatile("" ::"v"(probe));
#endif
#ifdef GR_PROBE_SALU
            int sprobe = 1;
#pragma unroll
            for (int q = 0; q < GR_PROBE_SALU; q++) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sprobe) : : "scc");
            asm volatile("" ::"s"(sprobe));
#endif
        }
#endif
        // velocity Verlet (step_verlet, cl.cl:3273-3346) in the reference's operation order.  (Through the half-kicked velocity
        // h = v + a ds/2 - x' = x + h ds, v~ = h + a ds/2, v' = h + a' ds/2 - it is 16 fmas instead of 20 operations and the same
        // algebra, but its roundings are not the reference's: measured against the golden pixels the RMSE of the Kerr cases went
        // from 7e-6 to 3.7e-5 and pixels off by > 1e-3 from 0-1 to 3-6 per fixture.  Parity first: 4 instructions.)
        const float half_ds = 0.5f * ds, half_ds2 = half_ds * ds;
        const float4 next_position = position + velocity * ds + acceleration * half_ds2;
        const float4 predicted = velocity + acceleration * ds;
#ifdef GR_PROBE_NO_ACCEL
        float4 next_acceleration = f4(acceleration.x * 0.999f, acceleration.y * 0.999f + predicted.x * 1e-3f, acceleration.z * 0.999f, acceleration.w * 0.999f + next_position.y * 1e-5f);
#else
        float4 next_acceleration = gm::geodesic_acceleration_with<decltype(libm)::value>(next_position, predicted, cfg);
#endif
        // (The acceleration above is 135 and more vector instructions in a row.  A wave that issues such a stretch back to back
        // leaves the SIMD's vector port idle part of the time; the host's pass over the compiled code - csrc/codeobject.cpp,
        // break_vector_runs - puts an s_nop after every 8th vector instruction of a run, which is worth 25 % here.)
        float4 next_velocity = velocity + (acceleration + next_acceleration) * half_ds;
        if (reparam) {
            const float md = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(next_velocity.x), __builtin_fabsf(next_velocity.y)),
                                             __builtin_fmaxf(__builtin_fabsf(next_velocity.z), __builtin_fabsf(next_velocity.w)));
            const float K = 1 / md;
            next_velocity = next_velocity * K;
            next_acceleration = next_acceleration * K * K;
            running *= K;   // also on an attempt that is then rejected, as the reference does (cl.cl:4152-4154)
        }

        bool accept = true, dead = false;
#ifdef ADAPTIVE_PRECISION
        if (inside) {
            // calculate_ds_error (cl.cl:3431-3456)
            const float q = controller.error_q(next_acceleration);
            // 0.99 * ds * clamp(suggested / ds, 0.3, 2) with ds > 0, without forming the quotient
            float nds = __builtin_amdgcn_fmed3f(0.99f * controller.suggestion(q), (0.99f * 0.3f) * ds, (0.99f * 2.f) * ds);
            nds = max_f32_uniform(controller.min_step, nds);
            next_ds = nds;
#ifdef SINGULARITY_DETECTION
            dead = (q > controller.singular_q) & (nds == controller.min_step);   // DS_RETURN: lost
#endif
            accept = !(nds < ds * (1 / 1.95f));   // back-step: retry from the same state with the smaller step
        }
#endif
        // IS_DEGENERATE on the accepted velocity: a non-finite sum <=> a non-finite component (finite components cannot
        // overflow the sum below ~1e38).  A rejected attempt is not tested (the reference `continue`s before its test: an
        // overshoot into a singularity is retried with the smaller step).
        const float poison = (next_velocity.x + next_velocity.y) + (next_velocity.z + next_velocity.w);
        dead |= accept & !(__builtin_fabsf(poison) <= 3.402823466e+38f);
        // The new state is written where the next attempt reads it; a rejection (once in ~35 000 attempts) puts the old state
        // back over it.  The copies are opaque to the compiler on purpose: as plain assignments it turns the two outcomes into
        // twelve selects per attempt.
        po = next_position;
        vo = next_velocity;
        ao = next_acceleration;
        if (__builtin_expect(!accept, 0)) {
            overwrite(po, position);
            overwrite(vo, velocity);
            overwrite(ao, acceleration);
            asm volatile("v_add_u32 %0, 1, %0\n\tv_add_u32 %1, 1, %1" : "+v"(rejections), "+v"(budget));   // the step it did not take
        }
        // the rare exit: the state the ray is left in, (p, v, a), has just passed the loop-top tests
        if (__builtin_expect(dead, 0)) return true;
        if (RESUMABLE) {
            // lanes still in the loop = the exec mask; fewer than keep_lanes of them: leave and let the caller refill the wave
            if (__builtin_popcountll(__builtin_amdgcn_ballot_w64(true)) < keep_lanes) { pause_wave = true; return true; }
        }
        return false;
    };

    // The state a ray leaves the fast loop in stays in register set 0 (below): no registers of its own.  (Earlier forms: left to
    // the compiler, "whichever set the ray was in when it left" becomes twelve running copies per attempt; in twelve registers of
    // its own it cost a wave per SIMD or, held to six waves, 60 bytes of scratch per lane and 0.1-0.2 GB of scratch traffic per 4K
    // launch; parked in LDS by hand - 14 KB per workgroup - it measured 6 % slower than the spill.)
    float exit_ds = 0, exit_running = 1;
    {
        const trig_flavour<false> polynomial;
        for (;;) {
#if GR_AGING_PRIORITY
            // A wave that has been on its tile for long gets issue priority over its neighbours on the SIMD.  A launch (and, for a
            // device's share of a split frame, the frame) ends when its slowest wave ends, and the slowest waves hold the tiles
            // with rays next to the shadow's edge: up to ~5 700 attempts, 7 ms when six waves share the SIMD evenly, 2.6 ms for a
            // wave that need not wait.  The total work is unchanged - the short tiles next to it take a little longer each.
            // Every 256 trips the wave looks at the step budget of its first live lane (uniform up to the rare rejections).
            if (!RESUMABLE) {
                const unsigned int left = (unsigned int)__builtin_amdgcn_readfirstlane((int)budget);
                if (__builtin_expect((left & 511u) < 2u, 0)) {
                    const unsigned int done = (unsigned int)loop_limit - left;
                    if (done >= 3072u) __builtin_amdgcn_s_setprio(3);
                    else if (done >= 1536u) __builtin_amdgcn_s_setprio(2);
                    else if (done >= 768u) __builtin_amdgcn_s_setprio(1);
                }
            }
#endif
            float4 p1, v1, a1;
            float ds_used, running_before;
            // A ray that leaves is left in set 0: lanes that have left are masked off for the rest of the loop, so set 0 keeps their
            // state with no registers of its own; a ray that leaves from set 1 is copied over first (opaque copies, once per ray).
            if (attempt(polynomial, p0, v0, a0, p1, v1, a1, ds_used, running_before)) {
                if (RESUMABLE && pause_wave) { overwrite(p0, p1); overwrite(v0, v1); overwrite(a0, a1); }
                asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "+v"(exit_ds), "+v"(exit_running) : "v"(ds_used), "v"(running_before));
                break;
            }
            if (attempt(polynomial, p1, v1, a1, p0, v0, a0, ds_used, running_before)) {
                if (!(RESUMABLE && pause_wave)) { overwrite(p0, p1); overwrite(v0, v1); overwrite(a0, a1); }
                asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "+v"(exit_ds), "+v"(exit_running) : "v"(ds_used), "v"(running_before));
                break;
            }
        }
    }
    float4 position = p0, velocity = v0, acceleration = a0;
#ifdef IS_CONSTANT_THETA
    position.z = GR_PIf / 2; velocity.z = 0; acceleration.z = 0;
#endif
    // Why the loop was left is read off what it leaves behind (flags set inside it would have to be carried through it per lane):
    // an exhausted step budget has wrapped around; otherwise the loop-top tests on the final state, in the reference's order
    // (cl.cl:3990-4130) - every other exit leaves a state that has just passed them.
    bool capped, lost_at_top, terminated_at_top;
    auto classify = [&]() {
        capped = budget == 0xffffffffu;
        float4 polar = gm::generic_to_spherical(position, cfg);
#ifdef IS_CONSTANT_THETA
        polar.z = GR_PIf / 2;
#endif
        lost_at_top = stop_lost(position, velocity, acceleration, running);
        terminated_at_top = stop_terminated(polar);
    };
    classify();
#if !defined(GR_FAST_TRIG) && !defined(GR_LIBM_TRIG) && !defined(GR_PROBE_NO_SLOW_TRIG)
    if (__builtin_expect(!capped && !pause_wave && !(lost_at_top | terminated_at_top), 0)) {
        // Left at the bottom of an attempt: degenerate for good, or a sin / cos argument outside the polynomial's range.  The
        // attempt is done again, and the ray integrated on if it was the latter, one attempt per trip with sin / cos from libm.
        // What the loop carries is put back to what it was before the abandoned attempt.
        const float ds_used = exit_ds;
        float4 polar = gm::generic_to_spherical(position, cfg);
        if (__builtin_fabsf(gm::distance_to_object(polar, cfg)) < new_max) next_ds = ds_used;   // min(next_ds, ambient) gives ds_used again
        running = exit_running;
        budget++;
        const trig_flavour<true> precise;
        for (;;) {
            float4 np, nv, na;
            float unused_ds, unused_running;
            const bool leave = attempt(precise, position, velocity, acceleration, np, nv, na, unused_ds, unused_running);
            if (leave && !pause_wave) break;
            position = np; velocity = nv; acceleration = na;
            if (leave) break;
        }
        classify();
    }
#endif
    paused = RESUMABLE && pause_wave;
    const bool left_at_top = !capped && !paused && (lost_at_top | terminated_at_top);
    int result = RAY_LOST;
    if (!paused) {
        // a state that is degenerate anywhere is the reference's plain `return` (cl.cl:4235-4244, terminated stays 0) even when its
        // position happens to lie beyond the boundary
        const bool finite = degenerate_accumulate(position, degenerate_accumulate(velocity, degenerate_accumulate(acceleration, 0.f))) == 0.f;
        if (!capped && !lost_at_top && terminated_at_top && finite) result = RAY_TERMINATED;
    }
    // every attempt() entered took one step off the budget; the entry that found the ray finished (or the budget empty) made none
    const unsigned int taken = capped ? budget_before : budget_before - budget - (left_at_top ? 1u : 0u);
    if (RESUMABLE) { s.next_ds = next_ds; s.steps += (int)taken; s.tries += taken + rejections; }
    s.position = position;
    s.velocity = velocity;
    s.acceleration = acceleration;
    s.running_dlambda_dnew = running;
    if (attempts) *attempts = RESUMABLE ? s.tries : taken + rejections;
    return result;
}

#ifdef GR_INTEGRATOR_V1   // the one-attempt-per-trip loop of round 1, kept for A/B measurements
__device__ __forceinline__ int integrate_ray(ray_state& s, cfg_t cfg, dfg_t dfg, unsigned int* attempts) {
    bool paused;
    return integrate_core<false>(s, cfg, dfg, attempts, 0, paused);
}
#else
__device__ __forceinline__ int integrate_ray(ray_state& s, cfg_t cfg, dfg_t dfg, unsigned int* attempts) {
    bool paused;
    return integrate_pingpong<false>(s, cfg, dfg, attempts, 0, paused);
}
#endif


#ifdef GR_TWO_RAYS_PER_LANE
// ---- two rays per lane ---------------------------------------------------------------------------
// integrate_core with every per-ray float held as a pair (ray 0 in the low, ray 1 in the high half of a 64-bit register
// pair).  Measured on MI355X (tools/ubench/accel_rate.hip: the substituted Kerr acceleration + Verlet update alone): packed
// instructions get through two rays' arithmetic in less issue time than two plain ones, 374 -> 462 G ray-steps/s.  Three
// quarters of the instructions of a Verlet attempt are such multiplies and fmas, so one lane stepping two rays gets through more
// attempts per cycle.  What has no packed form (compares, selects, rcp/rsq/sqrt, the commit of an accepted step) is done per
// half.  The arithmetic of a ray is instruction for instruction that of integrate_core; a ray that has left the loop keeps
// its state (the commit is per ray) while its partner goes on.
using gm::pairf;
using gm::pair4;
using gm::splat;
__device__ __forceinline__ pair4 operator+(pair4 a, pair4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
__device__ __forceinline__ pair4 operator*(pair4 a, pairf s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
template <int H> __device__ __forceinline__ float4 half_of(pair4 v) {
    return H == 0 ? f4(v.x.x, v.y.x, v.z.x, v.w.x) : f4(v.x.y, v.y.y, v.z.y, v.w.y);
}
__device__ __forceinline__ pair4 pair_of(float4 a, float4 b) {
    pair4 r;
    r.x.x = a.x; r.x.y = b.x; r.y.x = a.y; r.y.y = b.y; r.z.x = a.z; r.z.y = b.z; r.w.x = a.w; r.w.y = b.w;
    return r;
}

#ifdef ADAPTIVE_PRECISION
__device__ __forceinline__ pairf acceleration_to_precision(pair4 acc, float max_acceleration, pairf& next_ds) {
    pair4 wa = {acc.x * (float)(W_V1), acc.y * (float)(W_V2), acc.z * (float)(W_V3), acc.w * (float)(W_V4)};
    pairf d2 = wa.x * wa.x + wa.y * wa.y + wa.z * wa.z + wa.w * wa.w;
    pairf current;
    current.x = __builtin_sqrtf(d2.x); current.y = __builtin_sqrtf(d2.y);
    current = current * 0.01f;
    current = current / GR_W_MAX;
    const float scale = 65536.f;
    float err = max_acceleration;
    pairf diff = current * scale;
    float floor_diff = err * scale / 1e10f;
    if (diff.x < floor_diff) diff.x = floor_diff;
    if (diff.y < floor_diff) diff.y = floor_diff;
    const float root = __builtin_sqrtf(err * scale);
    next_ds.x = root * __builtin_amdgcn_rsqf(diff.x);
    next_ds.y = root * __builtin_amdgcn_rsqf(diff.y);
    return diff;
}
#endif

// in: the initial states of the two rays (an inactive ray carries a copy of its partner's so that its half computes on
// benign numbers); out: final position, velocity, running_dlambda_dnew, outcome and attempts per ray
__device__ __forceinline__ void integrate_pair(pair4& position_io, pair4& velocity_io, pair4 acceleration, pairf& running_out,
                                               bool active0, bool active1, cfg_t cfg, dfg_t dfg, int& result0, int& result1,
                                               unsigned int& tries0, unsigned int& tries1) {
    pair4 position = position_io, velocity = velocity_io;
    pairf f_in_x;
    f_in_x.x = __builtin_fabsf(velocity.x.x); f_in_x.y = __builtin_fabsf(velocity.x.y);
#ifdef IS_CONSTANT_THETA
    position.z = splat(GR_PIf / 2); velocity.z = splat(0.f); acceleration.z = splat(0.f);
#endif
    pairf next_ds = splat(0.00001f);
#ifdef ADAPTIVE_PRECISION
    const float max_accel = GET_FEATURE(max_acceleration_change, dfg);
    const float min_step = GET_FEATURE(min_step, dfg);
    (void)acceleration_to_precision(acceleration, max_accel, next_ds);
#endif
    const float subambient_precision = 0.5f;
    const float ambient_precision = 0.2f;
    const float new_max = GET_FEATURE(max_precision_radius, dfg);
    const float new_min = 3;
    const float universe = GET_FEATURE(universe_size, dfg);
    const bool reparam = GET_FEATURE(reparameterisation, dfg) != 0;
    pairf running = splat(1.f);
    const int loop_limit = 4096 * 4;
    unsigned int t0 = 0, t1 = 0;
    int i0 = 0, i1 = 0;
    bool alive0 = active0, alive1 = active1;

    // the loop-top tests of integrate_core on one ray's numbers
    auto stop_lost = [&](float pos_y, float vel_x_over_run, float acc_x_over_run, float fin, int steps) {
        bool lost = steps >= loop_limit;
#ifdef HAS_CYLINDRICAL_SINGULARITY
        lost |= pos_y < CYLINDRICAL_TERMINATOR;
#endif
#ifndef UNCONDITIONALLY_NONSINGULAR
        lost |= __builtin_fabsf(vel_x_over_run) > 1000 + fin && __builtin_fabsf(acc_x_over_run) > 100;
#endif
        (void)pos_y; (void)vel_x_over_run; (void)acc_x_over_run; (void)fin;
        return lost;
    };
    auto stop_terminated = [&](float polar_y) {
        bool t = __builtin_fabsf(polar_y) >= universe;
#ifdef SINGULAR
        t |= __builtin_fabsf(polar_y) < SINGULAR_TERMINATOR;
#endif
        return t;
    };
    for (;;) {
#ifdef IS_CONSTANT_THETA
        position.z = splat(GR_PIf / 2); velocity.z = splat(0.f); acceleration.z = splat(0.f);
#endif
        pair4 polar = gm::generic_to_spherical(position, cfg);
#ifdef IS_CONSTANT_THETA
        polar.z = splat(GR_PIf / 2);
#endif
        pairf r_value = gm::distance_to_object(polar, cfg);
        pairf ar;
        ar.x = __builtin_fabsf(r_value.x); ar.y = __builtin_fabsf(r_value.y);
        pairf ds;
#ifdef ADAPTIVE_PRECISION
        ds = next_ds;
#else
        ds.x = mixf(ambient_precision, subambient_precision, (clampf(ar.x, new_min, new_max) - new_min) / (new_max - new_min));
        ds.y = mixf(ambient_precision, subambient_precision, (clampf(ar.y, new_min, new_max) - new_min) / (new_max - new_min));
#endif
        if (ar.x < new_max) ds.x = __builtin_fminf(ds.x, ambient_precision);
        else ds.x = 0.1f * (ar.x - new_max) + ambient_precision;
        if (ar.y < new_max) ds.y = __builtin_fminf(ds.y, ambient_precision);
        else ds.y = 0.1f * (ar.y - new_max) + ambient_precision;

#ifndef UNCONDITIONALLY_NONSINGULAR
        const pairf vq = velocity.x / running, aq = acceleration.x / running;
#else
        const pairf vq = splat(0.f), aq = splat(0.f);
#endif
        alive0 = alive0 && !(stop_lost(position.y.x, vq.x, aq.x, f_in_x.x, i0) | stop_terminated(polar.y.x));
        alive1 = alive1 && !(stop_lost(position.y.y, vq.y, aq.y, f_in_x.y, i1) | stop_terminated(polar.y.y));
        if (!(alive0 | alive1)) break;
        t0 += alive0 ? 1u : 0u;
        t1 += alive1 ? 1u : 0u;

        // velocity Verlet (step_verlet), both rays
        const pairf half_ds = ds * 0.5f, half_ds2 = half_ds * ds;
        pair4 next_position = position + velocity * ds + acceleration * half_ds2;
        pair4 half_velocity = velocity + acceleration * ds;
#ifdef GR_PROBE_NO_ACCEL   // experiment: the loop without the metric's acceleration (results meaningless, cycles per attempt only)
        pair4 next_acceleration = {acceleration.x * 0.999f, acceleration.y * 0.999f + half_velocity.x * 1e-3f, acceleration.z * 0.999f, acceleration.w * 0.999f + next_position.y * 1e-5f};
#else
        pair4 next_acceleration = gm::geodesic_acceleration(next_position, half_velocity, cfg);
#endif
        pair4 next_velocity = velocity + (acceleration + next_acceleration) * half_ds;
        if (reparam) {
            pairf K;
            K.x = 1 / __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(next_velocity.x.x), __builtin_fabsf(next_velocity.y.x)),
                                      __builtin_fmaxf(__builtin_fabsf(next_velocity.z.x), __builtin_fabsf(next_velocity.w.x)));
            K.y = 1 / __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(next_velocity.x.y), __builtin_fabsf(next_velocity.y.y)),
                                      __builtin_fmaxf(__builtin_fabsf(next_velocity.z.y), __builtin_fabsf(next_velocity.w.y)));
            next_velocity = next_velocity * K;
            next_acceleration = next_acceleration * K * K;
            if (alive0) running.x *= K.x;
            if (alive1) running.y *= K.y;
        }

        bool accept0 = true, accept1 = true;
#ifdef ADAPTIVE_PRECISION
        {
            // calculate_ds_error for both rays in one straight line: every step below is two independent instructions (or
            // one packed one), so the serial chain sqrt -> rsq -> clamp -> compare at the end of an attempt is walked once for
            // the two rays, not once per ray, and nothing in it changes the exec mask
            pairf suggested;
            pairf diff = acceleration_to_precision(next_acceleration, max_accel, suggested);
            const pairf want = suggested * 0.99f, lo = ds * (0.99f * 0.3f), hi = ds * (0.99f * 2.f), back = ds / 1.95f;
            pairf nds;
            nds.x = __builtin_fmaxf(clampf(want.x, lo.x, hi.x), min_step);
            nds.y = __builtin_fmaxf(clampf(want.y, lo.y, hi.y), min_step);
            const bool inside0 = ar.x < new_max, inside1 = ar.y < new_max;
            next_ds.x = inside0 ? nds.x : next_ds.x;
            next_ds.y = inside1 ? nds.y : next_ds.y;
#ifdef SINGULARITY_DETECTION
            const pairf dq = diff / 65536.f;
            alive0 = alive0 & !(inside0 & (nds.x == min_step) & (dq.x > max_accel * 10000));
            alive1 = alive1 & !(inside1 & (nds.y == min_step) & (dq.y > max_accel * 10000));
#endif
            accept0 = !inside0 | !(nds.x < back.x);   // back-step: retry from the same state with the smaller step
            accept1 = !inside1 | !(nds.y < back.y);
            (void)diff;
        }
#endif
        if (alive0 && accept0) {
            position.x.x = next_position.x.x; position.y.x = next_position.y.x; position.z.x = next_position.z.x; position.w.x = next_position.w.x;
            velocity.x.x = next_velocity.x.x; velocity.y.x = next_velocity.y.x; velocity.z.x = next_velocity.z.x; velocity.w.x = next_velocity.w.x;
            acceleration.x.x = next_acceleration.x.x; acceleration.y.x = next_acceleration.y.x; acceleration.z.x = next_acceleration.z.x; acceleration.w.x = next_acceleration.w.x;
            i0++;
            float poison = degenerate_accumulate(half_of<0>(position), degenerate_accumulate(half_of<0>(velocity), 0.f));
            if (reparam) poison = degenerate_accumulate(half_of<0>(acceleration), poison);
            if (!(poison == 0.f)) alive0 = false;
        }
        if (alive1 && accept1) {
            position.x.y = next_position.x.y; position.y.y = next_position.y.y; position.z.y = next_position.z.y; position.w.y = next_position.w.y;
            velocity.x.y = next_velocity.x.y; velocity.y.y = next_velocity.y.y; velocity.z.y = next_velocity.z.y; velocity.w.y = next_velocity.w.y;
            acceleration.x.y = next_acceleration.x.y; acceleration.y.y = next_acceleration.y.y; acceleration.z.y = next_acceleration.z.y; acceleration.w.y = next_acceleration.w.y;
            i1++;
            float poison = degenerate_accumulate(half_of<1>(position), degenerate_accumulate(half_of<1>(velocity), 0.f));
            if (reparam) poison = degenerate_accumulate(half_of<1>(acceleration), poison);
            if (!(poison == 0.f)) alive1 = false;
        }
    }
    // why each ray left the loop: the loop-top tests on its final state (see integrate_core)
    {
        pair4 polar = gm::generic_to_spherical(position, cfg);
#ifndef UNCONDITIONALLY_NONSINGULAR
        const pairf vq = velocity.x / running, aq = acceleration.x / running;
#else
        const pairf vq = splat(0.f), aq = splat(0.f);
#endif
        const bool finite0 = degenerate_accumulate(half_of<0>(position), degenerate_accumulate(half_of<0>(velocity), degenerate_accumulate(half_of<0>(acceleration), 0.f))) == 0.f;
        const bool finite1 = degenerate_accumulate(half_of<1>(position), degenerate_accumulate(half_of<1>(velocity), degenerate_accumulate(half_of<1>(acceleration), 0.f))) == 0.f;
        result0 = (!stop_lost(position.y.x, vq.x, aq.x, f_in_x.x, i0) && stop_terminated(polar.y.x) && finite0) ? RAY_TERMINATED : RAY_LOST;
        result1 = (!stop_lost(position.y.y, vq.y, aq.y, f_in_x.y, i1) && stop_terminated(polar.y.y) && finite1) ? RAY_TERMINATED : RAY_LOST;
    }
    position_io = position;
    velocity_io = velocity;
    running_out = running;
    tries0 = t0;
    tries1 = t1;
}
#endif  // GR_TWO_RAYS_PER_LANE

// ------------------------------------------------------------------------------------------------
// final position -> sky coordinates (cl.cl:211-263, 5024-5100)

__device__ __forceinline__ float3 fix_ray_position_cart(float3 pos, float3 vel, float radius) {
    vel = normalize3(vel);
    float b = 2 * dot3(vel, pos);
    float c = dot3(pos, pos) - radius * radius;
    float discrim = b * b - 4 * c;
    if (discrim < 0) return pos;
    float sq = __builtin_sqrtf(discrim);
    float t0 = (-b - sq) / 2;
    float t1 = (-b + sq) / 2;
    float t = __builtin_fabsf(t0) < __builtin_fabsf(t1) ? t0 : t1;
    return pos + t * vel;
}

__device__ __forceinline__ float3 fix_ray_position(float3 polar_pos, float3 polar_vel, float radius) {
    float sgn = fsign(polar_pos.x);
    float3 cpos = polar_pos;
    cpos.x = __builtin_fabsf(cpos.x);
    polar_vel.x *= sgn;
    float3 cart_vel = spherical_velocity_to_cartesian_velocity(cpos, polar_vel);
    float3 cart_pos = polar_to_cartesian(cpos);
    float3 fixed = cartesian_to_polar(fix_ray_position_cart(cart_pos, cart_vel, radius));
#ifdef IS_CONSTANT_THETA
    fixed.y = GR_PIf / 2;
#endif
    fixed.x *= sgn;
    return fixed;
}

__device__ __forceinline__ float4 intersection_position(float4 ray_position, float4 ray_velocity, float4 initial_quat, cfg_t cfg, dfg_t dfg) {
    float4 position = gm::generic_to_spherical(ray_position, cfg);
    float4 velocity = gm::generic_velocity_to_spherical_velocity(ray_position, ray_velocity, cfg);
#ifdef IS_CONSTANT_THETA
    position.z = GR_PIf / 2;
    velocity.z = 0;
#endif
    const float universe = GET_FEATURE(universe_size, dfg);
    if (__builtin_fabsf(position.y) >= universe) {
        float3 p = fix_ray_position(yzw(position), yzw(velocity), universe);
        position = f4(position.x, p);
    }
#if defined(SINGULAR) && defined(TRAVERSABLE_EVENT_HORIZON)
    if (__builtin_fabsf(position.y) < SINGULAR_TERMINATOR) {
        float3 p = fix_ray_position(yzw(position), yzw(velocity), SINGULAR_TERMINATOR);
        position = f4(position.x, p);
    }
#endif
    float3 npolar = yzw(position);
#ifdef GENERIC_CONSTANT_THETA
    npolar = cartesian_to_polar(rot_quat(polar_to_cartesian(yzw(position)), initial_quat));
#endif
    (void)initial_quat;
    return f4(position.x, npolar);
}

__device__ __forceinline__ float2 angle_to_tex(float theta, float phi) {
    float thetaf = fmodf(theta, 2 * GR_PIf);
    float phif = phi;
    if (thetaf >= GR_PIf) { phif += GR_PIf; thetaf -= GR_PIf; }
    phif = fmodf(phif, 2 * GR_PIf);
    return make_float2(phif / (2 * GR_PIf) + 0.5f, thetaf / GR_PIf);
}

__device__ __forceinline__ float2 tex_to_angle(float2 tex) {
    return make_float2((tex.x - 0.5f) * (2 * GR_PIf), tex.y * GR_PIf);
}

// render_data of one finished ray (body of calculate_render_data, cl.cl:5146-5212)
__device__ __forceinline__ render_data make_render_data(float4 position, float4 velocity, float4 initial_quat, float ku_uobsu,
                                                        float running, int terminated, int sx, int sy, cfg_t cfg, dfg_t dfg,
                                                        bool need_redshift) {
    render_data dat;
    dat.terminated = terminated;
    dat.sx = sx;
    dat.sy = sy;
    dat.z_shift = 0;
    dat.tex_coord = make_float2(0, 0);
    dat.side = 1;
    if (terminated != 1) return dat;

    float4 ipos = intersection_position(position, velocity, initial_quat, cfg, dfg);
    float4 generic_velocity = velocity / running;
    dat.side = gm::generic_to_spherical(position, cfg).y < 0 ? 0 : 1;
#if !defined(TRAVERSABLE_EVENT_HORIZON)
    if (__builtin_fabsf(ipos.y) <= 1) return dat;
#endif
    if (need_redshift) {
        tetrad t;
        calculate_tetrads(position, f3(0, 0, 0), t, cfg, 0);
        float g[16];
        gm::metric_big_at(position, g, cfg);
        float4 obvs_low = lower_index_big(t.e[0], g);
        float z_shift = (dot4(generic_velocity, obvs_low) / ku_uobsu) - 1;
        dat.z_shift = __builtin_fmaxf(z_shift, -0.999f);
    }
    dat.tex_coord = angle_to_tex(ipos.z, ipos.w);
    return dat;
}

// ================================================================================================
// kernels

extern "C" __global__ void gr_cart_to_generic(const float4* __restrict__ position_cart_in, float4* __restrict__ position_generic_out,
                                              int count, float flip, cfg_t cfg) {
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= count) return;
    float4 in = position_cart_in[id];
    float3 polar = cartesian_to_polar(yzw(in));
    if (flip > 0) polar.x = -polar.x;
    position_generic_out[id] = gm::spherical_to_generic(f4(in.x, polar), cfg);
}

extern "C" __global__ void gr_init_basis_vectors(const float4* __restrict__ generic_in, int count, float speed_x, float speed_y, float speed_z,
                                                 float4* __restrict__ e0_out, float4* __restrict__ e1_out,
                                                 float4* __restrict__ e2_out, float4* __restrict__ e3_out, cfg_t cfg) {
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= count) return;
    tetrad t;
    calculate_tetrads(generic_in[id], f3(speed_x, speed_y, speed_z), t, cfg, 1);
    e0_out[id] = t.e[0];
    e1_out[id] = t.e[1];
    e2_out[id] = t.e[2];
    e3_out[id] = t.e[3];
}

extern "C" __global__ void gr_clear_termination_buffer(int* __restrict__ termination_buffer, int width, int height) {
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= width * height) return;
    termination_buffer[id] = 1;
}

// `tiled` is an extension over the reference signature: 0 = reference slot order (slot = cy*width+cx),
// 1 = 8x8 tile order (slot count is then rounded up to whole tiles; out-of-image slots get terminated = 2).
extern "C" __global__ void gr_init_rays_generic(const float4* __restrict__ g_generic_camera_in, const float4* __restrict__ g_camera_quat,
                                                lightray* __restrict__ metric_rays, int* __restrict__ metric_ray_count,
                                                int width, int height, const int* __restrict__ termination_buffer,
                                                int prepass_width, int prepass_height, int flip_geodesic_direction,
                                                const float4* __restrict__ e0, const float4* __restrict__ e1,
                                                const float4* __restrict__ e2, const float4* __restrict__ e3,
                                                cfg_t cfg, dfg_t dfg, int i_am_prepass, int tiled) {
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    int cx, cy;
    const int T = GR_TILE;
    int slots = tiled ? ((width + T - 1) / T) * ((height + T - 1) / T) * T * T : width * height;
    if (id >= slots) return;
    bool inside = slot_to_pixel(id, width, height, tiled, cx, cy);

    bool full = i_am_prepass || !GET_FEATURE(adaptive_sampling, dfg) || GET_FEATURE(use_triangle_rendering, dfg);
    if (id == 0) *metric_ray_count = full ? slots : (height * width) / 4;

    if (!inside) {
        lightray dead;
        dead.position = dead.velocity = dead.acceleration = f4(0, 0, 0, 0);
        dead.initial_quat = f4(0, 0, 0, 1);
        dead.ku_uobsu = 1; dead.running_dlambda_dnew = 1; dead.terminated = 2; dead.sx = -1; dead.sy = -1;
        metric_rays[id] = dead;
        return;
    }

    lightray ray = make_pixel_ray(cx, cy, width, height, *g_generic_camera_in, *g_camera_quat, *e0, *e1, *e2, *e3,
                                  flip_geodesic_direction, cfg, dfg);

    // prepass stencil (cl.cl:3213-3232)
    if (prepass_width != width && prepass_height != height) {
        float fx = exact_ratio(cx, width);
        float fy = exact_ratio(cy, height);
        int lx = (int)roundf(fx * prepass_width);
        int ly = (int)roundf(fy * prepass_height);
        if (early_terminate(lx - 1, ly, prepass_width, prepass_height, termination_buffer) &&
            early_terminate(lx, ly, prepass_width, prepass_height, termination_buffer) &&
            early_terminate(lx + 1, ly, prepass_width, prepass_height, termination_buffer) &&
            early_terminate(lx, ly - 1, prepass_width, prepass_height, termination_buffer) &&
            early_terminate(lx, ly + 1, prepass_width, prepass_height, termination_buffer)) {
            ray.terminated = 2;
        }
    }

    if (full) {
        metric_rays[id] = ray;
    } else {
        if ((cx % 2) != 0 || (cy % 2) != 0) return;
        metric_rays[(cy / 2) * (width / 2) + cx / 2] = ray;
    }
}

extern "C" __global__ void __launch_bounds__(64, GR_TRACE_WAVES)
gr_do_generic_rays(lightray* __restrict__ generic_rays_in, const int* __restrict__ generic_count_in,
                   int* __restrict__ ray_time_min, int* __restrict__ ray_time_max,
                   cfg_t cfg_in, dfg_t dfg_in, int width, int height, int mouse_x, int mouse_y,
                   float4* __restrict__ ray_write, int* __restrict__ ray_write_counts, int max_write,
                   unsigned long long* __restrict__ attempt_counter) {
    GR_PARAMETERS_IN_REGISTERS
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= *generic_count_in) return;
    if (ray_write_counts) ray_write_counts[id] = 0;
    lightray* ray = &generic_rays_in[id];
    if (ray->terminated == 2) return;

    ray_state s;
    s.position = ray->position;
    s.velocity = ray->velocity;
    s.acceleration = ray->acceleration;
    unsigned int tries = 0;
    int res = integrate_ray(s, cfg, dfg, &tries);
    if (res == RAY_TERMINATED) {
        ray->position = s.position;
        ray->velocity = s.velocity;
        ray->running_dlambda_dnew = s.running_dlambda_dnew;
        ray->terminated = 1;
    }
    if (attempt_counter) atomicAdd(attempt_counter, (unsigned long long)tries);   // one add per wave after compiler coalescing
}

extern "C" __global__ void gr_calculate_singularities(const lightray* __restrict__ finished_rays, const int* __restrict__ finished_count,
                                                      int* __restrict__ termination_buffer, int width, int height) {
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= *finished_count) return;
    int sx = id % width;
    int sy = id / width;
    termination_buffer[sy * width + sx] = !finished_rays[id].terminated;
}

extern "C" __global__ void gr_calculate_render_data(const lightray* __restrict__ rays_in, const int* __restrict__ rays_in_count,
                                                    render_data* __restrict__ rdata, int* __restrict__ rdata_count,
                                                    int width, int height, cfg_t cfg, dfg_t dfg) {
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= *rays_in_count) return;
    if (gid == 0) *rdata_count = width * height;
    const lightray* ray = &rays_in[gid];
    int sx = ray->sx, sy = ray->sy;
    if (sx < 0 || sy < 0 || sx >= width || sy >= height) return;   // padding slots of the tiled layout
    render_data dat = make_render_data(ray->position, ray->velocity, ray->initial_quat, ray->ku_uobsu, ray->running_dlambda_dnew,
                                       ray->terminated, sx, sy, cfg, dfg, true);
    rdata[sy * width + sx] = dat;
}

#ifdef GR_COUNT_WAVE_SLOTS
// the wave's longest trip count x slots_per_iteration, returned in one lane (0 in the others)
__device__ __forceinline__ unsigned int wave_slots(unsigned int trips, unsigned int slots_per_iteration) {
    unsigned int longest = trips;
    for (int o = 32; o > 0; o >>= 1) {
        unsigned int other = (unsigned int)__shfl_xor((int)longest, o, 64);
        longest = other > longest ? other : longest;
    }
    const unsigned long long active = __builtin_amdgcn_ballot_w64(true);
    const int first = __builtin_ctzll(active);
    return (int)(threadIdx.x % 64) == first ? longest * slots_per_iteration : 0u;
}
#endif

// init -> integrate -> render-data for one pixel per lane, 8x8 tiles, nothing but the 32-byte result is stored.
// `wave` numbers the tile-waves of this device: image rows are dealt to devices in blocks of `block_rows` rows (block-cyclic:
// global block gb belongs to device gb % strip_count); a block is tiles_x * block_rows/8 tile-waves, and, when the image is
// split, 64x1 "halo" waves tracing the row just below it, which the texture filter of the block's last row reads
// (cl.cl:5509-5520).  strip_count == 1: one block covering the whole image.
// Adaptive sampling on a split frame (halo of two rows).  A device that owns the row blocks strip_rank, strip_rank +
// strip_count, ... needs the block decisions of the pixel-block rows y (even) with r0 <= y <= r0 + B for each of its blocks [r0, r0 + B)
// - the row r0 + B is the halo row under the block that the texture filter reads - and for those the lattice rows y - 2 ... y + 2.
__device__ __forceinline__ bool own_block_within(int y, int margin, int height, int block_rows, int strip_rank, int strip_count) {
    // is there an own block b (b % strip_count == strip_rank, b * B < height) with b * B - margin <= y <= (b + 1) * B + margin ?
    const int last = (y + margin) / block_rows;
    for (int b = last; b >= 0 && (b + 1) * block_rows + margin >= y; b--)
        if (b % strip_count == strip_rank && b * block_rows < height) return true;
    return false;
}

// Shading inside the trace launch.  Of the 64 pixels of a tile, the 49 that are not in its last column or row have both neighbours
// the texture filter looks at (the pixel to the right and the pixel below, cl.cl:5509-5546) in the same wave: their sky coordinates
// come over by ds_bpermute and the wave writes the finished float4 pixels itself, straight from the registers the render-data
// record was built in.  The 15 pixels of the last column and row need records other waves write; gr_render shades those in a
// second, small launch (seams_only).  out == NULL: no shading here (gr_render does all of it).  Compiled into programs whose
// argument string carries -DGR_TILE_SHADING (gr_program_has_tile_shading); measured slower than the separate pass, DESIGN.md 4.
struct trace_shading {
    float4* out;
    const uchar4* bg1_texels;
    const uchar4* bg2_texels;
    int bg_width, bg_height, bg_levels, most_probes, compact_out;
};
__device__ __attribute__((noinline)) float4 shade_pixel_in_tile(const render_data& self, float2 beside, float2 below, const trace_shading& shading, dfg_t dfg);

__device__ __forceinline__ void trace_tile(int wave, int lane, const float4* __restrict__ camera, const float4* __restrict__ camera_quat,
                                           render_data* __restrict__ rdata, int width, int height, int block_rows, int strip_rank,
                                           int strip_count, const int* __restrict__ termination_buffer, int prepass_width,
                                           int prepass_height, const float4* __restrict__ e0, const float4* __restrict__ e1,
                                           const float4* __restrict__ e2, const float4* __restrict__ e3, cfg_t cfg, dfg_t dfg,
                                           unsigned long long* __restrict__ attempt_counter, int lattice, int pending_only,
                                           const trace_shading& shading, bool known_skipped, int cell_wave, bool cells_in_flight) {
    // cell_wave >= 0: this "tile" is 64 cells of the low-resolution prepass (prepass_cell, below) traced by the launch itself:
    // the ray of cell (cx, cy) of the prepass grid, and its verdict goes to the termination buffer instead of a record.
    // cells_in_flight: the launch has such waves, so a tile waits for the cells its pixels look at.
    // Adaptive sampling on the fused path (cl.cl:3234-3250, 5223-5345): lattice = 2 traces the pixels (2x, 2y) only - the tiles
    // then cover the half-resolution grid - and pending_only = 1 traces the pixels gr_adaptive_refine marked (terminated ==
    // GR_PENDING) and leaves every other record alone.  On a split frame (strip_count > 1) the lattice launch traces the lattice
    // rows this device's decisions read, the second launch the marked pixels of its own rows and halo rows.
    const int image_width = width, image_height = height;
    const int device_block_rows = block_rows, device_rank = strip_rank, device_count = strip_count;
    // the lattice launch walks the tiles of the whole half-resolution grid whoever owns the rows; a device of a split frame
    // traces the lattice rows its blocks' decisions read and leaves the others alone (below)
    if (lattice == 2) { width /= 2; height /= 2; block_rows = ((height + 7) / 8) * 8; strip_rank = 0; strip_count = 1; }
    const int T = GR_TILE;
    const int tiles_x = (width + T - 1) / T;
    const int tile_rows = block_rows / T;
    const int halo_waves = strip_count > 1 ? (width + 63) / 64 : 0;
    const int waves_per_block = tiles_x * tile_rows + halo_waves;
    const int local_block = wave / waves_per_block;
    const int within = wave % waves_per_block;
    const int r0 = (local_block * strip_count + strip_rank) * block_rows;
    int cx, cy;
    int ray_grid_width = image_width, ray_grid_height = image_height;   // the grid the ray's direction is a pixel of
    if (cell_wave >= 0) {
        const int cell = cell_wave * 64 + lane;
        if (cell >= prepass_width * prepass_height) return;
        cx = cell % prepass_width;
        cy = cell / prepass_width;
        ray_grid_width = prepass_width; ray_grid_height = prepass_height;
        width = image_width; height = image_height;
    } else {
    if (within < tiles_x * tile_rows) {
        cx = (within % tiles_x) * T + lane % T;
        cy = r0 + (within / tiles_x) * T + lane / T;
        if (cy >= r0 + block_rows) return;
    } else {
        cx = (within - tiles_x * tile_rows) * 64 + lane;
        cy = r0 + block_rows;
    }
    if (cx >= width || cy >= height) return;
    cx *= lattice; cy *= lattice;
    width = image_width; height = image_height;
    if (lattice == 2 && device_count > 1 && !own_block_within(cy, 2, height, device_block_rows, device_rank, device_count)) return;
    if (pending_only && rdata[cy * width + cx].terminated != GR_PENDING) return;
    }

    // the prepass verdict first: a skipped pixel (58 % of the 4K Kerr frame) needs no ray at all
    // known_skipped: a tile of gr_order_tiles' last class - the 5x5 cells around it are all in the shadow, and the stencil of every
    // one of its pixels lies inside those (a pixel rounds to a cell at most one from the tile centre's) - needs no look-up at all
    int terminated = known_skipped ? 2 : 0;
    if (cell_wave < 0 && !known_skipped && !pending_only && termination_buffer && prepass_width != width && prepass_height != height) {
        float fx = exact_ratio(cx, width);
        float fy = exact_ratio(cy, height);
        int lx = (int)roundf(fx * prepass_width);
        int ly = (int)roundf(fy * prepass_height);
        const bool skip = cells_in_flight ? early_terminate_stencil_when_known(lx, ly, prepass_width, prepass_height, termination_buffer)
                                          : early_terminate_stencil(lx, ly, prepass_width, prepass_height, termination_buffer);
        if (skip) terminated = 2;
    }
    render_data dat;
    unsigned int tries = 0;
    if (terminated == 2) {
        dat.tex_coord = make_float2(0, 0);
        dat.z_shift = 0;
        dat.sx = cx;
        dat.sy = cy;
        dat.terminated = 2;
        dat.side = 1;
    } else {
        // camera and tetrad are re-read (scalar loads) for every tile: 24 wave-uniform values held across the integrator
        // loop would spill scalar registers
        lightray ray = make_pixel_ray(cx, cy, ray_grid_width, ray_grid_height, *camera, *camera_quat, *e0, *e1, *e2, *e3, 0, cfg, dfg);
        ray_state s;
        s.position = ray.position;
        s.velocity = ray.velocity;
        s.acceleration = ray.acceleration;
        s.running_dlambda_dnew = 1;
        int res = integrate_ray(s, cfg, dfg, &tries);
        if (cell_wave >= 0) {
            // calculate_singularities (cl.cl:5008-5020): 1 = the ray did not reach the boundary.  Device scope: tiles on other
            // XCDs (each with an L2 of its own) are polling for it.
            __hip_atomic_store(const_cast<int*>(termination_buffer) + cy * prepass_width + cx, res == RAY_TERMINATED ? 0 : 1, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            return;   // (the prepass rays' attempts are not counted: gr_render_state_attempts is the frame's pixels', as with the prepass launched on its own)
        }
        if (res == RAY_TERMINATED) terminated = 1;
        else { s.position = ray.position; s.velocity = ray.velocity; s.running_dlambda_dnew = 1; }
        dat = make_render_data(s.position, s.velocity, ray.initial_quat, ray.ku_uobsu, s.running_dlambda_dnew, terminated, cx, cy, cfg,
                               dfg, GET_FEATURE(redshift, dfg) != 0);
    }
    rdata[cy * width + cx] = dat;
#ifdef GR_TILE_SHADING   // programs built with -DGR_TILE_SHADING only: carried along unused, the call's spills add 0.12 GB of scratch traffic per 4K launch
    if (shading.out && lattice == 1 && !pending_only && within < tiles_x * tile_rows) {
        // every lane of the tile that holds a pixel hands its sky coordinates to the lanes left of and above it
        const float2 beside = make_float2(__int_as_float(__builtin_amdgcn_ds_bpermute((lane + 1) * 4, __float_as_int(dat.tex_coord.x))),
                                          __int_as_float(__builtin_amdgcn_ds_bpermute((lane + 1) * 4, __float_as_int(dat.tex_coord.y))));
        const float2 below = make_float2(__int_as_float(__builtin_amdgcn_ds_bpermute((lane + T) * 4, __float_as_int(dat.tex_coord.x))),
                                         __int_as_float(__builtin_amdgcn_ds_bpermute((lane + T) * 4, __float_as_int(dat.tex_coord.y))));
        if (lane % T < T - 1 && lane / T < T - 1 && cx < width - 1 && cy < height - 1) {
            long long out_index = (long long)cy * width + cx;
            if (shading.compact_out) {   // the device's blocks back to back (gr_render's compact_out)
                const int block = cy / block_rows;
                out_index = ((long long)(block / strip_count) * block_rows + (cy - block * block_rows)) * width + cx;
            }
            shading.out[out_index] = shade_pixel_in_tile(dat, beside, below, shading, dfg);
        }
    }
#endif
#ifdef GR_COUNT_WAVE_SLOTS
    // experiment (tools/README.md): count the lane slots the wave spent in the Verlet loop - 64 x the trip count of its longest
    // ray - instead of the attempts; attempts / slots = the fraction of the lanes doing useful work
    tries = wave_slots(tries, 64u);
#endif
    if (attempt_counter) atomicAdd(attempt_counter, (unsigned long long)tries);
}

// workgroup size of the fused trace kernel: 4 tile-waves, one per SIMD of a CU (capi.cpp launches with the same number)
#ifndef GR_TRACE_BLOCK
#define GR_TRACE_BLOCK 256
#endif
// Two scheduling modes.  tile_counter == NULL: wave w of the launch traces tile w (grid = all tiles).  tile_counter != NULL:
// persistent waves - the launch only fills the machine and every wave keeps drawing the next tile from the device-side
// counter until total_waves are handed out, so a SIMD slot never idles between the end of a short tile (prepass-skipped
// tiles finish in a few hundred cycles) and the dispatcher's next workgroup.
extern "C" __global__ void __launch_bounds__(GR_TRACE_BLOCK, GR_FUSED_WAVES)
gr_trace_fused(const float4* __restrict__ g_generic_camera_in, const float4* __restrict__ g_camera_quat,
               render_data* __restrict__ rdata, int width, int height, int block_rows, int strip_rank, int strip_count,
               const int* __restrict__ termination_buffer, int prepass_width, int prepass_height,
               const float4* __restrict__ e0, const float4* __restrict__ e1, const float4* __restrict__ e2, const float4* __restrict__ e3,
               cfg_t cfg_in, dfg_t dfg_in, unsigned long long* __restrict__ attempt_counter, unsigned int* __restrict__ tile_counter,
               int total_waves, int lattice, int pending_only, const unsigned int* __restrict__ tile_order, trace_shading shading,
               int prepass_tickets) {
    // prepass_tickets > 0 (persistent launches in image order only): the first prepass_tickets tickets are the waves of the
    // low-resolution prepass, then come the tiles, which wait for the cells they look at (trace_tile).  A frame whose camera was not
    // known in advance then pays the prepass's single-ray latency once per cell wave alongside the first tiles instead of as a
    // launch of its own in front of the trace.
    GR_PARAMETERS_IN_REGISTERS
    const int lane = threadIdx.x % 64;
    // profiling launches (attempt_counter != NULL) also measure the shader clock they ran at: every wave adds its lifetime in
    // shader cycles (s_memtime) and in ticks of the constant 100 MHz reference clock (s_memrealtime) to attempt_counter[1], [2]
    unsigned long long born_cycles = 0, born_ticks = 0;
    if (attempt_counter) { born_cycles = __builtin_amdgcn_s_memtime(); born_ticks = __builtin_amdgcn_s_memrealtime(); }
    // one call site for both modes: the two schedules must run the very same instructions per pixel (strip renders are
    // compared bit for bit with whole-frame renders)
    int wave = blockIdx.x * (GR_TRACE_BLOCK / 64) + threadIdx.x / 64;
    // With gr_order_tiles' list a ticket is one tile of the classes that trace, or GR_SKIP_CHUNK tiles of the last class (all
    // pixels skipped by the prepass: a store each).  The tickets come from ONE counter, which the memory system serves at about
    // 10 ns a ticket whoever asks - nothing next to a tile's 0.1-1 ms of tracing, but the 75 000 skipped tiles of the 4K Kerr
    // frame, handed out back to back at the end of the list, would add 0.7 ms of pure ticket traffic to the launch.
    int held = 0, cursor = 0;   // tiles this wave still holds from its last ticket, and where in the list they start
    bool known_skipped = false;  // the ticket was a chunk of the last class
    const int tickets_total = total_waves + (prepass_tickets > 0 ? prepass_tickets : 0);   // (no tile order with prepass tickets)
    const int singles = (tile_counter && tile_order) ? total_waves - (int)tile_order[GR_TILE_CLASSES - 1] : tickets_total;
    for (;;) {
        if (tile_counter) {
            if (held == 0) {
                unsigned int ticket = 0;
                if (lane == 0) ticket = atomicAdd(tile_counter, 1u);
                cursor = (int)__builtin_amdgcn_readfirstlane(ticket);
                held = 1;
                known_skipped = tile_order && cursor >= singles;
                if (cursor >= singles) {
                    cursor = singles + (cursor - singles) * GR_SKIP_CHUNK;
                    held = tickets_total - cursor < GR_SKIP_CHUNK ? tickets_total - cursor : GR_SKIP_CHUNK;
                }
                if (held <= 0) break;
            }
            wave = tile_order ? (int)tile_order[GR_TILE_ORDER_HEADER + cursor] : cursor;
            cursor++;
            held--;
        }
        int cell_wave = -1;
        if (prepass_tickets > 0) {
            if (wave < prepass_tickets) { cell_wave = wave; wave = 0; }
            else wave -= prepass_tickets;
        }
        if (wave >= total_waves) break;
        // Launder the camera / tetrad pointers once per tile: otherwise everything in the ray set-up that depends only on
        // them is hoisted out of the tile loop and held in registers across the integrator (94 instead of 64 VGPRs, i.e.
        // 5 instead of 8 waves per SIMD).  Re-reading 96 bytes through the scalar cache per tile is free by comparison.
        asm volatile("" : "+s"(g_generic_camera_in), "+s"(g_camera_quat), "+s"(e0), "+s"(e1), "+s"(e2), "+s"(e3));
#if GR_AGING_PRIORITY
        __builtin_amdgcn_s_setprio(0);   // a new tile starts young
#endif
#ifdef GR_PROBE_LIFE_HISTOGRAM
        const unsigned long long tile_began = attempt_counter ? __builtin_amdgcn_s_memrealtime() : 0ull;
#endif
        trace_tile(wave, lane, g_generic_camera_in, g_camera_quat, rdata, width, height, block_rows, strip_rank, strip_count,
                   termination_buffer, prepass_width, prepass_height, e0, e1, e2, e3, cfg, dfg, attempt_counter, lattice, pending_only, shading,
                   known_skipped && lattice == 1 && !pending_only, cell_wave, prepass_tickets > 0);
#ifdef GR_PROBE_LIFE_HISTOGRAM   // per class of gr_order_tiles: tiles, summed and longest duration (10 ns ticks) in words 128..151 of the block
        if (attempt_counter && tile_order && tile_counter && lane == 0) {
            const unsigned long long took = __builtin_amdgcn_s_memrealtime() - tile_began;
            const unsigned int cls = tile_order[GR_TILE_ORDER_HEADER + total_waves + wave] % GR_TILE_CLASSES;
            atomicAdd(attempt_counter + 128 + cls * 3, 1ull);
            atomicAdd(attempt_counter + 129 + cls * 3, took);
            atomicMax(attempt_counter + 130 + cls * 3, took);
        }
#endif
#ifdef GR_TRACE_SINGLE_TILE   // experiment: one tile per wave only (launch with GR_TRACE_PERSISTENT=0)
        break;
#else
        if (!tile_counter) break;
#endif
    }
    if (attempt_counter && lane == 0) {
        atomicAdd(attempt_counter + 1, (unsigned long long)__builtin_amdgcn_s_memtime() - born_cycles);
        atomicAdd(attempt_counter + 2, (unsigned long long)__builtin_amdgcn_s_memrealtime() - born_ticks);
        atomicAdd(attempt_counter + 3, 1ull);
#ifdef GR_PROBE_LIFE_HISTOGRAM   // experiment (the frame state's 1 KiB counter block only): wave lifetimes in 0.125 ms bins
        unsigned long long bin = ((unsigned long long)__builtin_amdgcn_s_memrealtime() - born_ticks) / 12500ull;
        atomicAdd(attempt_counter + 8 + (bin < 119ull ? bin : 119ull), 1ull);
#endif
    }
}

// ---- ray compaction ------------------------------------------------------------------------------
// slot t of a device's work list = lane t % 64 of tile-wave t / 64 (the mapping of trace_tile); false for padding slots
__device__ __forceinline__ bool trace_slot_to_pixel(unsigned int slot, int width, int height, int block_rows, int strip_rank, int strip_count,
                                                    int& cx, int& cy) {
    const int T = GR_TILE;
    const int wave = (int)(slot / 64u), lane = (int)(slot % 64u);
    const int tiles_x = (width + T - 1) / T;
    const int tile_rows = block_rows / T;
    const int halo_waves = strip_count > 1 ? (width + 63) / 64 : 0;
    const int waves_per_block = tiles_x * tile_rows + halo_waves;
    const int local_block = wave / waves_per_block;
    const int within = wave % waves_per_block;
    const int r0 = (local_block * strip_count + strip_rank) * block_rows;
    if (within < tiles_x * tile_rows) {
        cx = (within % tiles_x) * T + lane % T;
        cy = r0 + (within / tiles_x) * T + lane / T;
        if (cy >= r0 + block_rows) return false;
    } else {
        cx = (within - tiles_x * tile_rows) * 64 + lane;
        cy = r0 + block_rows;
    }
    return cx < width && cy < height;
}

__device__ __forceinline__ bool prepass_skips_pixel(int cx, int cy, int width, int height, const int* __restrict__ termination_buffer,
                                                    int prepass_width, int prepass_height) {
    if (!termination_buffer || prepass_width == width || prepass_height == height) return false;
    float fx = exact_ratio(cx, width);
    float fy = exact_ratio(cy, height);
    int lx = (int)roundf(fx * prepass_width);
    int ly = (int)roundf(fy * prepass_height);
    return early_terminate_stencil(lx, ly, prepass_width, prepass_height, termination_buffer);
}

#ifdef GR_TWO_RAYS_PER_LANE
// gr_trace_fused with two rays per lane (integrate_pair): a wave takes the tile-waves 2k and 2k+1 of trace_tile's numbering -
// two horizontally adjacent 8x8 tiles - and lane l owns pixel l of each.  Same arguments, same records written.
__device__ __forceinline__ void trace_tile_pair(int pair_wave, int lane, const float4* __restrict__ camera, const float4* __restrict__ camera_quat,
                                                render_data* __restrict__ rdata, int width, int height, int block_rows, int strip_rank,
                                                int strip_count, const int* __restrict__ termination_buffer, int prepass_width,
                                                int prepass_height, const float4* __restrict__ e0, const float4* __restrict__ e1,
                                                const float4* __restrict__ e2, const float4* __restrict__ e3, cfg_t cfg, dfg_t dfg,
                                                unsigned long long* __restrict__ attempt_counter, int total_waves) {
    int cx0 = 0, cy0 = 0, cx1 = 0, cy1 = 0;
    const bool has0 = trace_slot_to_pixel((unsigned)(2 * pair_wave) * 64u + (unsigned)lane, width, height, block_rows, strip_rank, strip_count, cx0, cy0);
    const bool has1 = 2 * pair_wave + 1 < total_waves &&
                      trace_slot_to_pixel((unsigned)(2 * pair_wave + 1) * 64u + (unsigned)lane, width, height, block_rows, strip_rank, strip_count, cx1, cy1);
    const bool live0 = has0 && !prepass_skips_pixel(cx0, cy0, width, height, termination_buffer, prepass_width, prepass_height);
    const bool live1 = has1 && !prepass_skips_pixel(cx1, cy1, width, height, termination_buffer, prepass_width, prepass_height);
    render_data dat0, dat1;
    dat0.tex_coord = make_float2(0, 0); dat0.z_shift = 0; dat0.sx = cx0; dat0.sy = cy0; dat0.terminated = 2; dat0.side = 1;
    dat1.tex_coord = make_float2(0, 0); dat1.z_shift = 0; dat1.sx = cx1; dat1.sy = cy1; dat1.terminated = 2; dat1.side = 1;
    unsigned int tries0 = 0, tries1 = 0;
    if (live0 | live1) {
        // a lane with one ray only steps that ray in both halves
        const int ax = live0 ? cx0 : cx1, ay = live0 ? cy0 : cy1, bx = live1 ? cx1 : cx0, by = live1 ? cy1 : cy0;
        lightray ray0 = make_pixel_ray(ax, ay, width, height, *camera, *camera_quat, *e0, *e1, *e2, *e3, 0, cfg, dfg);
        lightray ray1 = make_pixel_ray(bx, by, width, height, *camera, *camera_quat, *e0, *e1, *e2, *e3, 0, cfg, dfg);
        pair4 position = pair_of(ray0.position, ray1.position), velocity = pair_of(ray0.velocity, ray1.velocity);
        pairf running;
        int res0, res1;
        integrate_pair(position, velocity, pair_of(ray0.acceleration, ray1.acceleration), running, live0, live1, cfg, dfg, res0, res1,
                       tries0, tries1);
        const bool need_redshift = GET_FEATURE(redshift, dfg) != 0;
        if (live0) dat0 = make_render_data(half_of<0>(position), half_of<0>(velocity), ray0.initial_quat, ray0.ku_uobsu, running.x,
                                           res0 == RAY_TERMINATED ? 1 : 0, cx0, cy0, cfg, dfg, need_redshift);
        if (live1) dat1 = make_render_data(half_of<1>(position), half_of<1>(velocity), ray1.initial_quat, ray1.ku_uobsu, running.y,
                                           res1 == RAY_TERMINATED ? 1 : 0, cx1, cy1, cfg, dfg, need_redshift);
    }
    if (has0) rdata[cy0 * width + cx0] = dat0;
    if (has1) rdata[cy1 * width + cx1] = dat1;
#ifdef GR_COUNT_WAVE_SLOTS
    tries0 = wave_slots(tries0 > tries1 ? tries0 : tries1, 128u);
    tries1 = 0;
#endif
    if (attempt_counter && (has0 | has1)) atomicAdd(attempt_counter, (unsigned long long)tries0 + (unsigned long long)tries1);
}

extern "C" __global__ void __launch_bounds__(GR_TRACE_BLOCK, GR_TRACE_WAVES)
gr_trace_pair(const float4* __restrict__ g_generic_camera_in, const float4* __restrict__ g_camera_quat,
              render_data* __restrict__ rdata, int width, int height, int block_rows, int strip_rank, int strip_count,
              const int* __restrict__ termination_buffer, int prepass_width, int prepass_height,
              const float4* __restrict__ e0, const float4* __restrict__ e1, const float4* __restrict__ e2, const float4* __restrict__ e3,
              cfg_t cfg_in, dfg_t dfg_in, unsigned long long* __restrict__ attempt_counter, unsigned int* __restrict__ tile_counter,
              int total_waves) {
    GR_PARAMETERS_IN_REGISTERS
    const int lane = threadIdx.x % 64;
    const int pair_waves = (total_waves + 1) / 2;
    int wave = blockIdx.x * (GR_TRACE_BLOCK / 64) + threadIdx.x / 64;
    for (;;) {
        if (tile_counter) {
            unsigned int ticket = 0;
            if (lane == 0) ticket = atomicAdd(tile_counter, 1u);
            wave = (int)__builtin_amdgcn_readfirstlane(ticket);
        }
        if (wave >= pair_waves) break;
        asm volatile("" : "+s"(g_generic_camera_in), "+s"(g_camera_quat), "+s"(e0), "+s"(e1), "+s"(e2), "+s"(e3));
        trace_tile_pair(wave, lane, g_generic_camera_in, g_camera_quat, rdata, width, height, block_rows, strip_rank, strip_count,
                        termination_buffer, prepass_width, prepass_height, e0, e1, e2, e3, cfg, dfg, attempt_counter, total_waves);
        if (!tile_counter) break;
    }
}
#endif  // GR_TWO_RAYS_PER_LANE

// gr_trace_fused with ray compaction: a persistent wave keeps one ray per lane and, as soon as fewer than keep_lanes of them
// are still integrating (wave-level ballot inside the Verlet loop), finishes the rays that ended, draws as many new pixels
// from the device-side slot counter as it has idle lanes and sets those rays up, then resumes the loop.  The arithmetic of
// a ray does not depend on which lane or in how many visits it is integrated, so the frame equals gr_trace_fused's up to
// what the compiler contracts differently in two kernels; what changes is how many lanes of the 64 do useful work when
// neighbouring rays need very different numbers of steps.  Measured on MI355X (4K Kerr): it does not pay for the workloads
// of BASELINE.json - 8x8 tiles already keep 97 % (a = 0.45) and 94 % (the a = 0.9 naked singularity) of the lanes busy,
// and the visits cost more than the idle lanes (7.1 -> 9.6 ms at keep_lanes 16..48) - so the frame driver leaves it off
// unless asked (gr_frame_options.ray_compaction).
extern "C" __global__ void __launch_bounds__(GR_TRACE_BLOCK, GR_TRACE_WAVES)
gr_trace_compact(const float4* __restrict__ g_generic_camera_in, const float4* __restrict__ g_camera_quat,
                 render_data* __restrict__ rdata, int width, int height, int block_rows, int strip_rank, int strip_count,
                 const int* __restrict__ termination_buffer, int prepass_width, int prepass_height,
                 const float4* __restrict__ e0, const float4* __restrict__ e1, const float4* __restrict__ e2, const float4* __restrict__ e3,
                 cfg_t cfg_in, dfg_t dfg_in, unsigned long long* __restrict__ attempt_counter, unsigned int* __restrict__ slot_counter,
                 unsigned int total_slots, int keep_lanes) {
    GR_PARAMETERS_IN_REGISTERS
    const bool need_redshift = GET_FEATURE(redshift, dfg) != 0;
    // per-lane ray: pixel, what render-data needs from the set-up, integrator progress
    int cx = 0, cy = 0;
    float4 start_position = f4(0, 0, 0, 0), start_velocity = f4(0, 0, 0, 0), initial_quat = f4(0, 0, 0, 1);
    float ku_uobsu = 1;
    ray_state s;
    s.position = s.velocity = s.acceleration = f4(0, 0, 0, 0);
    s.next_ds = 0; s.running_dlambda_dnew = 1; s.f_in_x = 0; s.steps = 0; s.tries = 0;
    bool has_ray = false;     // this lane holds a ray
    bool integrating = false; // ... that has not ended yet
    int outcome = RAY_LOST;
    bool exhausted = false;   // wave-uniform: the slot counter ran past the work list

    for (;;) {
        // 1. rays that ended: render-data record, lane becomes idle
        if (has_ray && !integrating) {
            int terminated = 0;
            float4 p = start_position, v = start_velocity;
            float running = 1;
            if (outcome == RAY_TERMINATED) { terminated = 1; p = s.position; v = s.velocity; running = s.running_dlambda_dnew; }
            rdata[cy * width + cx] = make_render_data(p, v, initial_quat, ku_uobsu, running, terminated, cx, cy, cfg, dfg, need_redshift);
            if (attempt_counter) atomicAdd(attempt_counter, (unsigned long long)s.tries);
            has_ray = false;
        }
        // 2. refill idle lanes (skipped and padding slots use up tickets without giving work, hence the loop)
        while (!exhausted) {
            const unsigned long long idle = __builtin_amdgcn_ballot_w64(!has_ray);
            const int n_idle = __builtin_popcountll(idle);
            if (n_idle == 0 || (n_idle <= 64 - keep_lanes && n_idle != 64)) break;   // enough rays on board
            unsigned int base = 0;
            if (threadIdx.x % 64 == 0) base = atomicAdd(slot_counter, (unsigned int)n_idle);
            base = __builtin_amdgcn_readfirstlane(base);
            if (base >= total_slots) { exhausted = true; break; }
            if (base + (unsigned int)n_idle >= total_slots) exhausted = true;
            const int my_rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(idle >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)idle, 0u));
            const unsigned int slot = base + (unsigned int)my_rank;
            if (!has_ray && slot < total_slots && trace_slot_to_pixel(slot, width, height, block_rows, strip_rank, strip_count, cx, cy)) {
                if (prepass_skips_pixel(cx, cy, width, height, termination_buffer, prepass_width, prepass_height)) {
                    render_data dat;
                    dat.tex_coord = make_float2(0, 0);
                    dat.z_shift = 0;
                    dat.sx = cx;
                    dat.sy = cy;
                    dat.terminated = 2;
                    dat.side = 1;
                    rdata[cy * width + cx] = dat;
                } else {
                    lightray ray = make_pixel_ray(cx, cy, width, height, *g_generic_camera_in, *g_camera_quat, *e0, *e1, *e2, *e3, 0, cfg, dfg);
                    start_position = ray.position;
                    start_velocity = ray.velocity;
                    initial_quat = ray.initial_quat;
                    ku_uobsu = ray.ku_uobsu;
                    s.position = ray.position;
                    s.velocity = ray.velocity;
                    s.acceleration = ray.acceleration;
                    integrate_begin(s, dfg);
                    has_ray = true;
                    integrating = true;
                }
            }
        }
        if (__builtin_amdgcn_ballot_w64(has_ray) == 0) break;   // nothing on board and nothing left to draw
        // 3. integrate until fewer than keep_lanes rays are still going (all of them to the end once the list is exhausted)
        if (integrating) {
            bool paused = false;
#ifdef GR_INTEGRATOR_V1
            outcome = integrate_core<true>(s, cfg, dfg, nullptr, exhausted ? 1 : keep_lanes, paused);
#else
            outcome = integrate_pingpong<true>(s, cfg, dfg, nullptr, exhausted ? 1 : keep_lanes, paused);
#endif
            integrating = paused;
        }
    }
}

// termination flags of the low-resolution prepass, straight from a fused trace (role of
// clear_termination_buffer + init_rays_generic(prepass) + do_generic_rays + calculate_singularities).
// When the image is split over devices (strip_count > 1) a device only traces the cells its own rows can look at: a pixel
// row cy reads the cell rows round(cy * ph / H) - 1 .. + 1 (init_rays_generic's 5-point stencil, cl.cl:3213-3232), so
// cell row cp matters to this device only if one of its blocks (or the halo row under it) intersects the pixel rows
// that map to cp - 1 .. cp + 1.  The prepass is otherwise replicated work: 11 % of a device's frame at 8 devices.
__device__ __forceinline__ void prepass_cell(int id, float4 camera, float4 camera_quat, float4 e0, float4 e1, float4 e2, float4 e3,
                                             int* __restrict__ termination_buffer, int prepass_width, int prepass_height, cfg_t cfg, dfg_t dfg,
                                             int image_height, int block_rows, int strip_rank, int strip_count,
                                             unsigned int* __restrict__ cell_attempts, int row_margin) {
    if (id >= prepass_width * prepass_height) return;
    int cx = id % prepass_width, cy = id / prepass_width;
    if (strip_count > 1) {
        // pixel rows whose stencil can touch cell row cy: round(y * ph / H) in [cy - 1, cy + 1], one row of slack either side for
        // the float rounding of that quotient (tests/test_distributed_cpu.py checks the rule by brute force)
        // row_margin: pixel rows beyond its blocks and halo rows the device also looks from (adaptive sampling: 2, the lattice rows
        // its block decisions read)
        long long lo = ((long long)(2 * cy - 3) * image_height) / (2 * prepass_height) - 1 - row_margin;
        long long hi = ((long long)(2 * cy + 3) * image_height + 2 * prepass_height - 1) / (2 * prepass_height) + 1 + row_margin;
        if (lo < 0) lo = 0;
        if (hi > image_height - 1) hi = image_height - 1;
        // blocks b (rows b*B .. (b+1)*B inclusive of the halo row) that meet [lo, hi]: (b+1)*B >= lo and b*B <= hi
        long long b_lo = (lo - 1) / block_rows, b_hi = hi / block_rows;
        if (b_lo < 0) b_lo = 0;
        long long first = b_lo + (((long long)strip_rank - b_lo) % strip_count + strip_count) % strip_count;   // first own block >= b_lo
        if (first > b_hi) return;
    }
    lightray ray = make_pixel_ray(cx, cy, prepass_width, prepass_height, camera, camera_quat, e0, e1, e2, e3, 0, cfg, dfg);
    ray_state s;
    s.position = ray.position;
    s.velocity = ray.velocity;
    s.acceleration = ray.acceleration;
    unsigned int tries = 0;
    int res = integrate_ray(s, cfg, dfg, &tries);
    termination_buffer[id] = res == RAY_TERMINATED ? 0 : 1;
    if (cell_attempts) cell_attempts[id] = tries;   // what the ray cost: gr_order_tiles' estimate for the tiles around the cell
}

extern "C" __global__ void __launch_bounds__(64, GR_TRACE_WAVES)
gr_prepass_fused(const float4* __restrict__ g_generic_camera_in, const float4* __restrict__ g_camera_quat,
                 int* __restrict__ termination_buffer, int prepass_width, int prepass_height,
                 const float4* __restrict__ e0, const float4* __restrict__ e1, const float4* __restrict__ e2, const float4* __restrict__ e3,
                 cfg_t cfg_in, dfg_t dfg_in, int image_height, int block_rows, int strip_rank, int strip_count,
                 unsigned int* __restrict__ cell_attempts, int row_margin) {
    GR_PARAMETERS_IN_REGISTERS
    prepass_cell(blockIdx.x * blockDim.x + threadIdx.x, *g_generic_camera_in, *g_camera_quat, *e0, *e1, *e2, *e3, termination_buffer,
                 prepass_width, prepass_height, cfg, dfg, image_height, block_rows, strip_rank, strip_count, cell_attempts, row_margin);
}

// cart_to_generic_kernel + init_basis_vectors + the prepass in ONE launch (the reference: three of its launches and the prepass
// sequence, main.cpp:2311, 2329, 2380-2436).  The camera's metric coordinates and tetrad - one lane's worth of work, ~900
// instructions - are computed by every lane of the launch for itself (same inputs, same instructions, same values), lane 0 of the
// launch also stores them for gr_trace_fused.  What this buys is the launch chain of a frame: two single-lane kernels with their
// queue latencies sat in front of every prepass (1.3 ms on average on the look-ahead stream under load, round-1 profile), which
// is what a device's share of a frame costs altogether once the frame is split eight ways.  prepass_width * prepass_height may
// be 0 (metrics without a prepass): the launch is then the camera set-up alone.
extern "C" __global__ void __launch_bounds__(64, GR_TRACE_WAVES)
gr_camera_prepass(const float4* __restrict__ position_cart_in, float flip, float speed_x, float speed_y, float speed_z,
                  float4* __restrict__ position_generic_out, float4* __restrict__ e0_out, float4* __restrict__ e1_out,
                  float4* __restrict__ e2_out, float4* __restrict__ e3_out, const float4* __restrict__ g_camera_quat,
                  int* __restrict__ termination_buffer, int prepass_width, int prepass_height, cfg_t cfg_in, dfg_t dfg_in,
                  int image_height, int block_rows, int strip_rank, int strip_count, unsigned int* __restrict__ cell_attempts,
                  int row_margin) {
    GR_PARAMETERS_IN_REGISTERS
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    const float4 in = *position_cart_in;
    float3 polar = cartesian_to_polar(yzw(in));
    if (flip > 0) polar.x = -polar.x;
    const float4 camera = gm::spherical_to_generic(f4(in.x, polar), cfg);
    tetrad t;
    calculate_tetrads(camera, f3(speed_x, speed_y, speed_z), t, cfg, 1);
    if (id == 0) {
        *position_generic_out = camera;
        *e0_out = t.e[0];
        *e1_out = t.e[1];
        *e2_out = t.e[2];
        *e3_out = t.e[3];
    }
    prepass_cell(id, camera, *g_camera_quat, t.e[0], t.e[1], t.e[2], t.e[3], termination_buffer, prepass_width, prepass_height, cfg, dfg,
                 image_height, block_rows, strip_rank, strip_count, cell_attempts, row_margin);
}

// ---- the order the persistent trace hands its tiles out in -----------------------------------------
// A persistent launch ends when its slowest wave ends, and a wave that draws a long tile late ends late: with the tiles handed
// out in image order the 4K Kerr launch spent its last 1.4 of 6.3 ms draining (tickets gone at 4.9 ms; 6 waves share a SIMD,
// so an average traced tile of ~500 attempts takes 0.6 ms and the tiles on the shadow's edge several times that).  The prepass
// has already traced one ray per 16x16 pixels: what those rays cost is a fair estimate of what the tiles around them will
// cost, so the tiles are handed out longest first - 16 classes: tiles that straddle the shadow's edge first of all, then by the
// most expensive ray among the cells around the tile's centre, an octave of attempts per class, tiles no pixel of which needs a
// ray (a store per pixel) last.  That is for a device's share of a split frame, where a wave slot gets one or two tiles and which
// comes last decides when the launch ends; on a whole 4K frame (nine traced tiles per slot) image order measured 2 % faster -
// the longest tiles take 7 ms when six of them share a SIMD from the start, 3.5 ms next to short tiles that keep restarting.  Scheduling only: which wave traces a tile and when has no influence on its rays.
// Two launches over the device's tiles: phase 0 counts the classes, phase 1 deals every tile a place in its class's range
// (order within a class: as the atomics fall, i.e. roughly image order).  list[0..15] counts, [16..31] cursors, then the tiles,
// then the tiles' classes (scratch between the two phases).
__device__ __forceinline__ int tile_cost_class(int tile, int width, int height, int block_rows, int strip_rank, int strip_count,
                                               const int* __restrict__ termination_buffer, const unsigned int* __restrict__ cell_attempts,
                                               int prepass_width, int prepass_height) {
    int cx = 0, cy = 0;
    if (!trace_slot_to_pixel((unsigned)tile * 64u + 36u, width, height, block_rows, strip_rank, strip_count, cx, cy) &&
        !trace_slot_to_pixel((unsigned)tile * 64u, width, height, block_rows, strip_rank, strip_count, cx, cy))
        return GR_TILE_CLASSES - 1;   // padding: nothing to trace
    const int lx = (int)roundf(exact_ratio(cx, width) * prepass_width), ly = (int)roundf(exact_ratio(cy, height) * prepass_height);
    // the halo pieces of a split frame are 64 pixels of one row, not a tile: four cells wide, so the promise of the last class
    // (below) cannot be made for them
    const int tiles_in_block = ((width + GR_TILE - 1) / GR_TILE) * (block_rows / GR_TILE);
    const bool halo_piece = strip_count > 1 && tile % (tiles_in_block + (width + 63) / 64) >= tiles_in_block;
    // a pixel's stencil reaches one cell beyond the cell it rounds to, and the pixels of a tile round to cells up to one away
    // from the centre's: shadow flags over 5x5 cells (the last class promises that no pixel of the tile needs a ray), costs over
    // the cells within GR_TILE_COST_REACH
    int in_shadow = 0;
    unsigned int dearest = 0;
    // every cell is read, at clamped coordinates, whether it counts or not: 50 independent loads in flight instead of a chain of
    // conditional ones (the kernel is nothing but their latency)
#pragma unroll
    for (int dy = -2; dy <= 2; dy++)
#pragma unroll
        for (int dx = -2; dx <= 2; dx++) {
            const int x = min(max(lx + dx, 0), prepass_width - 1), y = min(max(ly + dy, 0), prepass_height - 1);
            const bool inside = x == lx + dx && y == ly + dy;   // outside the grid: never "skip" (early_terminate)
            const int flag = termination_buffer[y * prepass_width + x];
            in_shadow += (inside && flag == 1) ? 1 : 0;
            const unsigned int a = cell_attempts[y * prepass_width + x];
            const bool near = dx >= -GR_TILE_COST_REACH && dx <= GR_TILE_COST_REACH && dy >= -GR_TILE_COST_REACH && dy <= GR_TILE_COST_REACH;
            dearest = (inside && near && a > dearest) ? a : dearest;
        }
    if (in_shadow == 25 && !halo_piece) return GR_TILE_CLASSES - 1;
    if (in_shadow > 0) return 0;
    // classes 1 .. 14 by the dearest ray, GR_TILE_CLASS_STEPS classes per octave of attempts, dearest first, < 256 (128) last
    const int steps = (int)((float)GR_TILE_CLASS_STEPS * __log2f((float)(dearest > 128u ? dearest : 128u) * (1.f / 128.f)));
    return GR_TILE_CLASSES - 2 - (steps > 13 ? 13 : steps);
}

extern "C" __global__ void __launch_bounds__(1024)
gr_order_tiles(const int* __restrict__ termination_buffer, const unsigned int* __restrict__ cell_attempts, int prepass_width,
               int prepass_height, int width, int height, int block_rows, int strip_rank, int strip_count, int total_tiles,
               unsigned int* __restrict__ list, int phase) {
    // one atomic per class and WORKGROUP on the device-wide counters: they are single addresses that every XCD contends for
    // (~50 ns an atomic; per wave the 2 000 waves of a 4K frame spent 0.1 ms on them), the waves of a workgroup meet in LDS
    __shared__ unsigned int group_count[GR_TILE_CLASSES], group_base[GR_TILE_CLASSES];
    if (threadIdx.x < GR_TILE_CLASSES) group_count[threadIdx.x] = 0;
    __syncthreads();
    const int tile = blockIdx.x * blockDim.x + threadIdx.x;
    // phase 0 works the classes out and leaves them behind the list for phase 1
    unsigned int* classes = list + GR_TILE_ORDER_HEADER + total_tiles;
    int cls = -1;
    if (tile < total_tiles) {
        if (phase == 0) {
            cls = tile_cost_class(tile, width, height, block_rows, strip_rank, strip_count, termination_buffer, cell_attempts, prepass_width,
                                  prepass_height);
            classes[tile] = (unsigned int)cls;
        } else {
            cls = (int)classes[tile];
        }
    }
    const int lane = threadIdx.x % 64;
    unsigned int place = 0;   // of this tile among its workgroup's tiles of the same class
    for (int c = 0; c < GR_TILE_CLASSES; c++) {
        const unsigned long long members = __builtin_amdgcn_ballot_w64(cls == c);
        if (members) {
            const int leader = __builtin_ctzll(members);
            unsigned int wave_base = 0;
            if (lane == leader) wave_base = atomicAdd(&group_count[c], (unsigned int)__builtin_popcountll(members));
            wave_base = __builtin_amdgcn_readlane(wave_base, leader);
            if (cls == c) place = wave_base + (unsigned int)__builtin_popcountll(members & ((1ull << lane) - 1ull));
        }
    }
    __syncthreads();
    if (threadIdx.x < GR_TILE_CLASSES && group_count[threadIdx.x])
        group_base[threadIdx.x] = atomicAdd(list + (phase == 0 ? 0 : GR_TILE_CLASSES) + threadIdx.x, group_count[threadIdx.x]);
    __syncthreads();
    if (phase == 1 && cls >= 0) {
        unsigned int first = 0;   // where the class's range starts
        for (int c = 0; c < cls; c++) first += list[c];
        list[GR_TILE_ORDER_HEADER + first + group_base[cls] + place] = (unsigned int)tile;
    }
}

// ------------------------------------------------------------------------------------------------
// adaptive sampling (cl.cl:5215-5345)

__device__ __forceinline__ float angle_between_angles(float2 a1, float2 a2) {
    float3 v1 = polar_to_cartesian(f3(1.f, a1.x, a1.y));
    float3 v2 = polar_to_cartesian(f3(1.f, a2.x, a2.y));
    return acosf(clampf(dot3(v1, v2), -1.f, 1.f));
}

__device__ __forceinline__ render_data interpolate_render_data(render_data r1, render_data r2) {
    float2 a1 = tex_to_angle(r1.tex_coord);
    float2 a2 = tex_to_angle(r2.tex_coord);
    float3 v1 = polar_to_cartesian(f3(1.f, a1.y, a1.x));
    float3 v2 = polar_to_cartesian(f3(1.f, a2.y, a2.x));
    float3 vc = (v1 + v2) / 2.f;
    float3 fangle = cartesian_to_polar(vc);
    render_data out;
    out.tex_coord = angle_to_tex(fangle.y, fangle.z);
    out.z_shift = (r1.z_shift + r2.z_shift) / 2.f;
    out.terminated = r1.terminated;
    out.sx = (r1.sx + r2.sx) / 2;
    out.sy = (r1.sy + r2.sy) / 2;
    out.side = (r1.side + r2.side) / 2;
    return out;
}

// handle_adaptive_sampling on the fused path: the half-resolution records are already render_data (gr_trace_fused, lattice 2), so
// the decision is taken on them - the sky angles come back out of the texture coordinates instead of out of 96-byte ray records -
// and a block that needs its three other pixels marks them GR_PENDING in place for the second fused launch (pending_only) instead
// of appending rays to a list: the second launch then walks the same 8x8 tiles, neighbouring rays stay in one wave, no atomics
// order the work.  Same tests as the reference (cl.cl:5242-5282): boundary blocks always refine, differing termination flags
// refine, otherwise the angular error across the block against the per-pixel angle times the threshold.
extern "C" __global__ void gr_adaptive_refine(render_data* __restrict__ rdat, int* __restrict__ pending_count, int width, int height,
                                              dfg_t dfg, int block_rows, int strip_rank, int strip_count) {
    const int sx = blockIdx.x * blockDim.x + threadIdx.x;
    const int sy = blockIdx.y * blockDim.y + threadIdx.y;
    const int hw = width / 2, hh = height / 2;
    if (sx >= hw || sy >= hh) return;
    const int lsx = 2 * sx, lsy = 2 * sy;
    // split frame: only the pixel blocks whose rows this device shades or reads as a halo row (their lattice neighbours were traced)
    if (strip_count > 1 && !own_block_within(lsy, 0, height, block_rows, strip_rank, strip_count)) return;
    auto at = [&](int x, int y) -> render_data& { return rdat[y * width + x]; };
    bool refine = true;
    if (sx != 0 && sx != hw - 1 && sy != 0 && sy != hh - 1) {
        const render_data centre = at(lsx, lsy), left = at(lsx - 2, lsy), right = at(lsx + 2, lsy), up = at(lsx, lsy - 2), down = at(lsx, lsy + 2);
        const int down_right_flag = at(lsx + 2, lsy + 2).terminated;
        const float2 la = tex_to_angle(left.tex_coord), ra = tex_to_angle(right.tex_coord), ua = tex_to_angle(up.tex_coord), da = tex_to_angle(down.tex_coord);
        // tex_to_angle gives (phi, theta); the reference compares (theta, phi) pairs
        const float x_error = __builtin_fabsf(angle_between_angles(make_float2(la.y, la.x), make_float2(ra.y, ra.x)));
        const float y_error = __builtin_fabsf(angle_between_angles(make_float2(da.y, da.x), make_float2(ua.y, ua.x)));
        const float relative_angular_error = (float)((double)(((x_error + x_error + y_error + y_error) / 4.f) / 2) * GR_PI);
        const float fov = GET_FEATURE(field_of_view, dfg);
        const float per_pixel = (float)((double)(fov * 2) * GR_PI / (double)360.f) / width;
        refine = relative_angular_error >= per_pixel * GET_FEATURE(adaptive_sampling_threshold, dfg);
        const int ct = centre.terminated;
        if (ct != left.terminated || ct != right.terminated || ct != up.terminated || ct != down.terminated || ct != down_right_flag) refine = true;
    }
    if (refine) {
        at(lsx + 1, lsy).terminated = GR_PENDING;
        at(lsx, lsy + 1).terminated = GR_PENDING;
        at(lsx + 1, lsy + 1).terminated = GR_PENDING;
        if (pending_count) atomicAdd(pending_count, 3);
    } else {
        const render_data c = at(lsx, lsy);
        at(lsx + 1, lsy) = interpolate_render_data(c, at(lsx + 2, lsy));
        at(lsx, lsy + 1) = interpolate_render_data(c, at(lsx, lsy + 2));
        at(lsx + 1, lsy + 1) = interpolate_render_data(c, at(lsx + 2, lsy + 2));
    }
}

extern "C" __global__ void gr_handle_adaptive_sampling(const lightray* __restrict__ rays_in, const int* __restrict__ rays_in_count,
                                                       render_data* __restrict__ rdat, int* __restrict__ rdata_count,
                                                       lightray* __restrict__ unprocessed_rays_out, int* __restrict__ unprocessed_rays_out_count,
                                                       const float4* __restrict__ g_generic_camera_in, const float4* __restrict__ g_camera_quat,
                                                       const float4* __restrict__ e0, const float4* __restrict__ e1,
                                                       const float4* __restrict__ e2, const float4* __restrict__ e3,
                                                       int width, int height, cfg_t cfg, dfg_t dfg) {
    int sx = blockIdx.x * blockDim.x + threadIdx.x;
    int sy = blockIdx.y * blockDim.y + threadIdx.y;
    int hw = width / 2, hh = height / 2;
    if (sx >= hw || sy >= hh) return;

    bool should_sample = true;
    if (sx != 0 && sx != hw - 1 && sy != 0 && sy != hh - 1) {
        const lightray* centre = &rays_in[sy * hw + sx];
        const lightray* left = &rays_in[sy * hw + sx - 1];
        const lightray* right = &rays_in[sy * hw + sx + 1];
        const lightray* up = &rays_in[(sy - 1) * hw + sx];
        const lightray* down = &rays_in[(sy + 1) * hw + sx];
        const lightray* down_right = &rays_in[(sy + 1) * hw + sx + 1];

        float4 lpos = intersection_position(left->position, left->velocity, left->initial_quat, cfg, dfg);
        float4 rpos = intersection_position(right->position, right->velocity, right->initial_quat, cfg, dfg);
        float4 upos = intersection_position(up->position, up->velocity, up->initial_quat, cfg, dfg);
        float4 dpos = intersection_position(down->position, down->velocity, down->initial_quat, cfg, dfg);

        float x_error = __builtin_fabsf(angle_between_angles(make_float2(lpos.z, lpos.w), make_float2(rpos.z, rpos.w)));
        float y_error = __builtin_fabsf(angle_between_angles(make_float2(dpos.z, dpos.w), make_float2(upos.z, upos.w)));
        // the reference's expression is ((xe.x+xe.y+ye.x+ye.y)/4.f)/2*M_PI with both lanes of each float2 equal (cl.cl:5272)
        float relative_angular_error = (float)((double)(((x_error + x_error + y_error + y_error) / 4.f) / 2) * GR_PI);
        float fov = GET_FEATURE(field_of_view, dfg);
        float fov_angle_pi = (float)((double)(fov * 2) * GR_PI / (double)360.f);
        float per_pixel = fov_angle_pi / width;
        should_sample = relative_angular_error >= per_pixel * GET_FEATURE(adaptive_sampling_threshold, dfg);
        int ct = centre->terminated;
        if (ct != left->terminated || ct != right->terminated || ct != up->terminated || ct != down->terminated || ct != down_right->terminated)
            should_sample = true;
    }

    if (should_sample) {
        int base_sx = sx * 2, base_sy = sy * 2;
        int px[3] = {base_sx + 1, base_sx, base_sx + 1};
        int py[3] = {base_sy, base_sy + 1, base_sy + 1};
        int root_id = atomicAdd(unprocessed_rays_out_count, 3);
        for (int i = 0; i < 3; i++) {
            unprocessed_rays_out[root_id + i] = make_pixel_ray(px[i], py[i], width, height, *g_generic_camera_in, *g_camera_quat,
                                                               *e0, *e1, *e2, *e3, 0, cfg, dfg);
        }
    } else {
        int lsx = rays_in[sy * hw + sx].sx;
        int lsy = rays_in[sy * hw + sx].sy;
        render_data cdata = rdat[lsy * width + lsx];
        render_data rdata_ = rdat[lsy * width + lsx + 2];
        render_data ddata = rdat[(lsy + 2) * width + lsx];
        render_data drdata = rdat[(lsy + 2) * width + lsx + 2];
        rdat[lsy * width + lsx + 1] = interpolate_render_data(cdata, rdata_);
        rdat[(lsy + 1) * width + lsx] = interpolate_render_data(cdata, ddata);
        rdat[(lsy + 1) * width + lsx + 1] = interpolate_render_data(cdata, drdata);
    }
}

// ------------------------------------------------------------------------------------------------
// Shading: what a pixel's ray sees of the sky (the reference: render, cl.cl:5453-5846, with read_mipmap 5421-5449, the colour
// helpers 326-350 and 5366-5413, circular_diff 3598-3610).  The pixel's footprint on the sky texture is an ellipse (from the
// texture-coordinate differences to the neighbouring pixels); it is integrated by a few Gaussian-weighted trilinear probes along
// its long axis (an EWA approximation), each at the mip level of its short axis.  Written here as three pieces - the sky sampler,
// the footprint, the probe integration - around the arithmetic the golden pixels pin (tests/test_gpu_parity.py::test_render_pixels).

// The sky as the reference stores it (graphics_settings.cpp:152-212): `levels` RGBA8 slices of the full size, slice L holding mip L
// in its top-left 2^-L corner with the edge texels replicated, so that a lookup at coordinates scaled by 2^-L never leaves the mip.
// Sampling is what OpenCL's NORMALIZED | REPEAT | LINEAR sampler does on a 2D array (OpenCL 1.2, 8.2 and 8.4): wrap, texel
// centres at +0.5, the slice picked by rounding.
struct sky_sampler {
    const uchar4* __restrict__ texels;   // [levels][height][width]
    int width, height, levels;

    __device__ __forceinline__ float4 texel(int x, int y, int slice) const {
        const uchar4 t = texels[((size_t)slice * height + y) * width + x];
        const float unorm = 1.f / 255.f;
        return f4(t.x * unorm, t.y * unorm, t.z * unorm, t.w * unorm);
    }
    __device__ float4 bilinear(float u, float v, float slice_f) const {
        int slice = (int)rintf(slice_f);
        slice = slice < 0 ? 0 : (slice > levels - 1 ? levels - 1 : slice);
        const float x = (u - floorf(u)) * width - 0.5f, y = (v - floorf(v)) * height - 0.5f;
        const float x_floor = floorf(x), y_floor = floorf(y);
        int x0 = (int)x_floor, y0 = (int)y_floor, x1 = x0 + 1, y1 = y0 + 1;
        if (x0 < 0) x0 += width;
        if (x1 > width - 1) x1 -= width;
        if (y0 < 0) y0 += height;
        if (y1 > height - 1) y1 -= height;
        const float wx = x - x_floor, wy = y - y_floor;
        return ((1 - wx) * (1 - wy)) * texel(x0, y0, slice) + (wx * (1 - wy)) * texel(x1, y0, slice) +
               ((1 - wx) * wy) * texel(x0, y1, slice) + (wx * wy) * texel(x1, y1, slice);
    }
    // between the two mips around `lod`
    __device__ float4 trilinear(float2 uv, float lod) const {
        lod = __builtin_fmaxf(lod, 0.f);
        uv.x = fmodf(uv.x, 1.f);
        uv.y = fmodf(uv.y, 1.f);
        const float fine = floorf(lod), coarse = ceilf(lod);
        const float fine_scale = exp2f(fine), coarse_scale = exp2f(coarse);
        const float4 a = bilinear(uv.x / fine_scale, uv.y / fine_scale, fine);
        const float4 b = bilinear(uv.x / coarse_scale, uv.y / coarse_scale, coarse);
        return a + (b - a) * (lod - fine);
    }
};

// Footprint of a pixel in texels: the ellipse  A u^2 + B u v + C v^2 = 1  spanned by the texture-space images of the pixel's two
// edges, each padded by one texel (the "+ 1" that keeps a vanishing footprint from collapsing), reduced to its axes.
struct sky_footprint {
    float long_radius, short_radius, angle;
};
__device__ __forceinline__ sky_footprint pixel_footprint(float2 along_x, float2 along_y) {
    const float raw_a = along_x.y * along_x.y + along_y.y * along_y.y + 1;
    const float raw_b = -2 * (along_x.x * along_x.y + along_y.x * along_y.y);
    const float raw_c = along_x.x * along_x.x + along_y.x * along_y.x + 1;
    const float norm = raw_a * raw_c - raw_b * raw_b / 4;
    const float a = raw_a / norm, b = raw_b / norm, c = raw_c / norm;
    const float spread = __builtin_sqrtf((a - c) * (a - c) + b * b);
    sky_footprint f;
    f.long_radius = 1.f / __builtin_sqrtf((a + c - spread) / 2);
    f.short_radius = 1.f / __builtin_sqrtf((a + c + spread) / 2);
    f.angle = atan2f(b, (a - c) / 2);
    f.long_radius = __builtin_fmaxf(f.long_radius, 1.f);
    f.short_radius = __builtin_fmaxf(f.short_radius, 1.f);
    f.long_radius = __builtin_fmaxf(f.long_radius, f.short_radius);
    return f;
}

// The footprint integrated over the sky: 2 (long / short) - 1 probes, capped at `most_probes` (the short axis then grows to keep
// the long one covered), spaced along the long axis, Gaussian weights exp(-2 d^2) in units of the long radius, each probe a
// trilinear lookup at the mip level of the short axis.
__device__ float4 integrate_footprint(const sky_sampler& sky, float2 centre, sky_footprint f, int most_probes) {
    const float wanted = 2 * (f.long_radius / f.short_radius) - 1;
    int probes = (int)floorf(wanted + 0.5f);
    probes = probes < most_probes ? probes : most_probes;
    if (probes < wanted) f.short_radius = 2 * f.long_radius / (probes + 1);
    float lod = log2f(f.short_radius);
    const int coarsest = sky.levels - 1;
    if (lod > coarsest) { lod = coarsest; probes = 1; }
    if (probes <= 1) {
        if (probes < 1) lod = coarsest;
        return sky.trilinear(centre, lod);
    }
    const float span = 2 * (f.long_radius - f.short_radius);
    const float step_u = cosf(f.angle) * span / (probes - 1), step_v = sinf(f.angle) * span / (probes - 1);
    const float step_u_norm = step_u / sky.width, step_v_norm = step_v / sky.height;
    const float step2 = (step_u * step_u + step_v * step_v) / (f.long_radius * f.long_radius);
    // probe k sits at (2k - (probes - 1)) half steps from the centre; an even count starts one half step further out on the
    // low side (the reference's startN, cl.cl:5641-5650)
    int half_steps = (probes % 2) == 1 ? -2 * ((probes - 1) / 2) : -2 * (probes / 2) - 1;
    float4 sum = f4(0, 0, 0, 0);
    float weight_sum = 0;
    for (int k = 0; k < probes; k++, half_steps += 2) {
        const float weight = expf(-2.f * ((half_steps * half_steps / 4.f) * step2));
        const float offset = half_steps / 2.f;
        sum = sum + weight * sky.trilinear(make_float2(centre.x + offset * step_u_norm, centre.y + offset * step_v_norm), lod);
        weight_sum += weight;
    }
    return sum / weight_sum;
}

// colour (cl.cl:326-350, 5366-5413)
__device__ __forceinline__ float srgb_to_linear(float v) { return v < 0.04045f ? v / 12.92f : powf((v + 0.055f) / 1.055f, 2.4f); }
__device__ __forceinline__ float linear_to_srgb(float v) { return v <= 0.0031308f ? v * 12.92f : 1.055f * powf(v, 1.0f / 2.4f) - 0.055f; }
__device__ __forceinline__ float3 srgb_to_linear(float3 c) { return f3(srgb_to_linear(c.x), srgb_to_linear(c.y), srgb_to_linear(c.z)); }
__device__ __forceinline__ float3 linear_to_srgb(float3 c) { return f3(linear_to_srgb(c.x), linear_to_srgb(c.y), linear_to_srgb(c.z)); }
__device__ __forceinline__ float luminous_energy(float3 v) { return v.x * 0.2125f + v.y * 0.7154f + v.z * 0.0721f; }
__device__ __forceinline__ float3 saturate3(float3 v) { return f3(clampf(v.x, 0.f, 1.f), clampf(v.y, 0.f, 1.f), clampf(v.z, 0.f, 1.f)); }
__device__ __forceinline__ float3 blend3(float3 a, float3 b, float t) { return a + (b - a) * t; }

// A linear colour seen at redshift z: its luminance scaled as the cube of the wavelength ratio (555 nm reference), then tinted
// towards red (z > 0) or blue (z < 0) by tanh of the shift; what a blue shift pushes out of gamut is handed to red and green.
__device__ float3 apply_redshift(float3 linear, float z, dfg_t dfg) {
    const float light_speed = 299792458;
    const float reference_wavelength = 555 / light_speed;
    const float seen_wavelength = reference_wavelength / (z + 1);
    const float luminance = 0.2126f * linear.x + 0.7152f * linear.y + 0.0722f * linear.z;
    const float shifted_luminance = clampf(powf(seen_wavelength, 3.f) * luminance / powf(reference_wavelength, 3.f), 0.f, 1.f);
    if ((double)luminance > 0.00001) linear = saturate3((shifted_luminance / luminance) * linear);
    const float energy = luminous_energy(linear);
    const float3 pure_red = f3(1 / 0.2125f, 0.f, 0.f), pure_green = f3(0, (float)(1 / 0.7154), 0.f), pure_blue = f3(0.f, 0.f, (float)(1 / 0.0721));
    float3 tinted;
    if (z > 0) {
        tinted = blend3(linear, energy * pure_red, tanhf(z));
    } else {
        tinted = blend3(linear, energy * pure_blue, tanhf((1 / (1 + z)) - 1));
        if (!GET_FEATURE(use_old_redshift, dfg)) {
            const float lost = luminous_energy(tinted) - luminous_energy(saturate3(tinted));
            tinted.x += lost * (pure_red.x + pure_green.x);
            tinted.y += lost * (pure_red.y + pure_green.y);
        }
    }
    return saturate3(saturate3(tinted));
}

// b - a for texture coordinates that wrap with period 1, through the angle they stand for (mixed double / float as the
// reference evaluates it, cl.cl:3598-3604)
__device__ __forceinline__ float wrapped_difference(float a, float b) {
    const float angle_a = (float)((double)a * (2 * GR_PI / (double)1.f));
    const float angle_b = (float)((double)b * (2 * GR_PI / (double)1.f));
    const float d = angle_b - angle_a;
    return (float)((double)(1.f * atan2f(sinf(d), cosf(d))) / (2 * GR_PI));
}

// One pixel: `self` is its record, `beside` / `below` the texture coordinates of its horizontal / vertical neighbour - the next
// pixel, or the previous one at the last column / row (`beside_is_previous`, `below_is_previous`), cl.cl:5509-5546.
__device__ float4 shade_pixel(const render_data& self, float2 beside, bool beside_is_previous, float2 below, bool below_is_previous,
                              const sky_sampler& near_sky, const sky_sampler& far_sky, int most_probes, dfg_t dfg) {
    if (self.terminated != 1) return f4(0, 0, 0, 1);
    const sky_sampler& sky = self.side >= 1 ? near_sky : far_sky;   // which side of a wormhole the ray ended on
    const float shrink = 1.3f;   // the reference's filter bias
    float2 along_x = make_float2(wrapped_difference(self.tex_coord.x, beside.x) / shrink, wrapped_difference(self.tex_coord.y, beside.y) / shrink);
    float2 along_y = make_float2(wrapped_difference(self.tex_coord.x, below.x) / shrink, wrapped_difference(self.tex_coord.y, below.y) / shrink);
    if (beside_is_previous) { along_x.x = -along_x.x; along_x.y = -along_x.y; }
    if (below_is_previous) { along_y.x = -along_y.x; along_y.y = -along_y.y; }
    along_x.x *= sky.width; along_y.x *= sky.width;
    along_x.y *= sky.height; along_y.y *= sky.height;
    float4 colour = integrate_footprint(sky, self.tex_coord, pixel_footprint(along_x, along_y), most_probes);
    float3 rgb = f3(colour.x, colour.y, colour.z);
    if (GET_FEATURE(redshift, dfg)) {
        rgb = apply_redshift(srgb_to_linear(rgb), self.z_shift, dfg);
#ifndef LINEAR_FRAMEBUFFER
        rgb = linear_to_srgb(rgb);
#endif
    } else {
#ifdef LINEAR_FRAMEBUFFER
        rgb = srgb_to_linear(rgb);
#endif
    }
    return f4(rgb.x, rgb.y, rgb.z, colour.w);
}

__device__ __attribute__((noinline)) float4 shade_pixel_in_tile(const render_data& self, float2 beside, float2 below, const trace_shading& shading, dfg_t dfg) {
    const sky_sampler near_sky{shading.bg1_texels, shading.bg_width, shading.bg_height, shading.bg_levels},
                      far_sky{shading.bg2_texels, shading.bg_width, shading.bg_height, shading.bg_levels};
    return shade_pixel(self, beside, false, below, false, near_sky, far_sky, shading.most_probes, dfg);
}

// The launch of the reference (num_pixels = block_pixels = width * height, strip_rank 0, strip_count 1, compact_out 0) shades the
// whole image; the extension shades one device's row blocks of a split image: work item gid is pixel `off` of local block `lb`,
// the global block being lb * strip_count + strip_rank, and with compact_out the device's blocks are written back to back.
extern "C" __global__ void gr_render(const render_data* __restrict__ rdata, const int* __restrict__ rdata_count, float4* __restrict__ out,
                                     const uchar4* __restrict__ bg1_texels, const uchar4* __restrict__ bg2_texels,
                                     int bg_width, int bg_height, int bg_levels,
                                     int width, int height, int most_probes, cfg_t cfg, dfg_t dfg,
                                     int num_pixels, int block_pixels, int strip_rank, int strip_count, int compact_out, int seams_only) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= num_pixels) return;
    int lb, off;
    if (seams_only) {
        // the pixels a fused trace launch left (trace_shading): last column and last row of every 8x8 tile, 15 work items per tile,
        // tiles numbered block by block as the trace numbers them.  Records are indexed by pixel; width is a multiple of 8.
        const int tiles_x = width / GR_TILE, tiles_per_block = tiles_x * (block_pixels / width / GR_TILE);
        const int tile = gid / 15, k = gid - tile * 15;
        lb = tile / tiles_per_block;
        const int within = tile - lb * tiles_per_block;
        const int x = (within % tiles_x) * GR_TILE + (k < GR_TILE ? GR_TILE - 1 : k - GR_TILE);
        const int y = (within / tiles_x) * GR_TILE + (k < GR_TILE ? k : GR_TILE - 1);
        off = y * width + x;
    } else {
        lb = gid / block_pixels;
        off = gid - lb * block_pixels;
    }
    const int id = (lb * strip_count + strip_rank) * block_pixels + off;
    if (id >= *rdata_count || id >= width * height) return;
    const render_data self = rdata[id];
    const int px = self.sx, py = self.sy;
    const int out_index = compact_out ? lb * block_pixels + off : py * width + px;
    const bool last_column = px == width - 1, last_row = py == height - 1;
    float2 beside = make_float2(0, 0), below = make_float2(0, 0);
    if (self.terminated == 1) {
        beside = rdata[py * width + px + (last_column ? -1 : 1)].tex_coord;
        below = rdata[(py + (last_row ? -1 : 1)) * width + px].tex_coord;
    }
    const sky_sampler near_sky{bg1_texels, bg_width, bg_height, bg_levels}, far_sky{bg2_texels, bg_width, bg_height, bg_levels};
    out[out_index] = shade_pixel(self, beside, last_column, below, last_row, near_sky, far_sky, most_probes, dfg);
    (void)cfg;
}

// ================================================================================================
// camera riding a timelike geodesic: boost_tetrad cl.cl:2441-2481, init_inertial_ray :3117-3141,
// get_geodesic_path :4735-4940, parallel_transport_quantity :2569-2620, handle_interpolating_geodesic :2738-2872.
// One lane per observer (the reference launches {1}/{1} for the camera, {N}/{256} for objects); buffers are
// "step-major": element k of observer id lives at [k * count + id].

__device__ __forceinline__ float4 timelike_vector(float3 speed, float4 e0, float4 e1, float4 e2, float4 e3) {
    float v2 = dot3(speed, speed);
    float Y = 1 / __builtin_sqrtf(1 - v2);
    return Y * e0 + (Y * speed.x) * e1 + (Y * speed.y) * e2 + (Y * speed.z) * e3;
}

extern "C" __global__ void gr_boost_tetrad(const float4* __restrict__ generic_in, int count, const float4* __restrict__ basis_speed,
                                           float4* __restrict__ e0_io, float4* __restrict__ e1_io, float4* __restrict__ e2_io,
                                           float4* __restrict__ e3_io, cfg_t cfg) {
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= count) return;
    float4 e0 = e0_io[id], e1 = e1_io[id], e2 = e2_io[id], e3 = e3_io[id];
    float4 sp = basis_speed[id];
    float4 observer = timelike_vector(f3(sp.x, sp.y, sp.z), e0, e1, e2, e3);
    float g[16];
    gm::metric_big_at(generic_in[id], g, cfg);
    // calculate_lorentz_boost(_big), cl.cl:1919-1972
    float4 lT4 = lower_index_big(e0, g), lu4 = lower_index_big(observer, g);
    float T[4] = {e0.x, e0.y, e0.z, e0.w}, lT[4] = {lT4.x, lT4.y, lT4.z, lT4.w};
    float uo[4] = {observer.x, observer.y, observer.z, observer.w}, luo[4] = {lu4.x, lu4.y, lu4.z, lu4.w};
    float gamma = -dot4(lT4, observer);
    float L[16];
    for (int u = 0; u < 4; u++)
        for (int v = 0; v < 4; v++)
            L[u * 4 + v] = (u == v ? 1.f : 0.f) + (1 / (1 + gamma)) * (T[u] + uo[u]) * (lT[v] + luo[v]) - 2 * uo[u] * lT[v];
    float4 rows[4] = {f4(L[0], L[1], L[2], L[3]), f4(L[4], L[5], L[6], L[7]), f4(L[8], L[9], L[10], L[11]), f4(L[12], L[13], L[14], L[15])};
    e0_io[id] = observer;
    e1_io[id] = f4(dot4(rows[0], e1), dot4(rows[1], e1), dot4(rows[2], e1), dot4(rows[3], e1));
    e2_io[id] = f4(dot4(rows[0], e2), dot4(rows[1], e2), dot4(rows[2], e2), dot4(rows[3], e2));
    e3_io[id] = f4(dot4(rows[0], e3), dot4(rows[1], e3), dot4(rows[2], e3), dot4(rows[3], e3));
}

extern "C" __global__ void gr_init_inertial_ray(const float4* __restrict__ generic_position_in, int ray_count,
                                                lightray* __restrict__ metric_rays, int* __restrict__ metric_ray_count,
                                                const float4* __restrict__ e0, const float4* __restrict__ e1,
                                                const float4* __restrict__ e2, const float4* __restrict__ e3,
                                                const float4* __restrict__ basis_speed, cfg_t cfg) {
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= ray_count) return;
    float4 sp = basis_speed[id];
    float4 velocity = timelike_vector(f3(sp.x, sp.y, sp.z), e0[id], e1[id], e2[id], e3[id]);
    // geodesic_to_trace_ray, cl.cl:3066-3115
    lightray ray = make_render_ray(0, 0, generic_position_in[id], velocity, e0[id], cfg);
    ray.ku_uobsu = 1;
    metric_rays[id] = ray;
    if (id == 0) *metric_ray_count = ray_count;
}

// circular_diff / periodic_diff, cl.cl:3598-3630
__device__ __forceinline__ float circular_diff_period(float f1, float f2, float period) {
    float g1 = (float)((double)f1 * (2 * GR_PI / (double)period));
    float g2 = (float)((double)f2 * (2 * GR_PI / (double)period));
    float d = g2 - g1;
    return (float)((double)(period * atan2f(sinf(d), cosf(d))) / (2 * GR_PI));
}
__device__ __forceinline__ float4 periodic_diff(float4 in1, float4 in2, float4 periods) {
    float4 ret = in1 - in2;
    if (periods.x != 0) ret.x = circular_diff_period(in2.x, in1.x, periods.x);
    if (periods.y != 0) ret.y = circular_diff_period(in2.y, in1.y, periods.y);
    if (periods.z != 0) ret.z = circular_diff_period(in2.z, in1.z, periods.z);
    if (periods.w != 0) ret.w = circular_diff_period(in2.w, in1.w, periods.w);
    return ret;
}

extern "C" __global__ void gr_get_geodesic_path(const lightray* __restrict__ generic_rays_in, float4* __restrict__ positions_out,
                                                float4* __restrict__ velocities_out, float* __restrict__ ds_out,
                                                const int* __restrict__ generic_count_in, int max_path_length, cfg_t cfg, dfg_t dfg,
                                                int* __restrict__ count_out) {
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= *generic_count_in) return;
    const lightray* ray = &generic_rays_in[id];
    float4 position = ray->position, velocity = ray->velocity, acceleration = ray->acceleration;
    const float4 quat = ray->initial_quat;
    const float f_in_x = __builtin_fabsf(velocity.x);
#ifdef IS_CONSTANT_THETA
    position.z = GR_PIf / 2; velocity.z = 0; acceleration.z = 0;
#endif
    float next_ds = 0.00001f;
#ifdef ADAPTIVE_PRECISION
    const float max_accel = __builtin_fminf(0.00001000f, GET_FEATURE(max_acceleration_change, dfg));
    const float min_step = GET_FEATURE(min_step, dfg);
    (void)acceleration_to_precision(acceleration, max_accel, next_ds);
#endif
    const float subambient_precision = 0.5f, ambient_precision = 0.2f;
    const float new_max = GET_FEATURE(max_precision_radius, dfg), new_min = 3;
    const float universe = GET_FEATURE(universe_size, dfg);
    const bool reparam = GET_FEATURE(reparameterisation, dfg) != 0;
    const int stride = *generic_count_in;
    int bufc = 0;
    const float4 periods = gm::coordinate_period(cfg);
    float4 last_pos_generic = f4(0, 0, 0, 0);
    float running = 1;
    (void)quat; (void)periods; (void)last_pos_generic;

    for (int i = 0; i < max_path_length; i++) {
#ifdef IS_CONSTANT_THETA
        position.z = GR_PIf / 2; velocity.z = 0; acceleration.z = 0;
#endif
        float4 polar = gm::generic_to_spherical(position, cfg);
#ifdef IS_CONSTANT_THETA
        polar.z = GR_PIf / 2;
#endif
        float ar = __builtin_fabsf(gm::distance_to_object(polar, cfg));
        float ds = mixf(ambient_precision, subambient_precision, (clampf(ar, new_min, new_max) - new_min) / (new_max - new_min));
#ifdef ADAPTIVE_PRECISION
        ds = next_ds;
#endif
        if (ar < new_max) ds = __builtin_fminf(ds, ambient_precision);
        else ds = (float)(0.1 * (double)(ar - new_max) + (double)ambient_precision);   // unsuffixed 0.1 here (cl.cl:4815)

        bool should_break = __builtin_fabsf(polar.y) >= universe;
#ifdef SINGULAR
        should_break |= __builtin_fabsf(polar.y) < SINGULAR_TERMINATOR;
#endif
        float4 next_position = position + velocity * ds + (0.5f * acceleration) * (ds * ds);
        float4 half_velocity = velocity + acceleration * ds;
        float4 next_acceleration = gm::geodesic_acceleration(next_position, half_velocity, cfg);
        float4 next_velocity = velocity + (0.5f * (acceleration + next_acceleration)) * ds;
        float K = 1;
        if (reparam) {
            float md = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(next_velocity.x), __builtin_fabsf(next_velocity.y)),
                                       __builtin_fmaxf(__builtin_fabsf(next_velocity.z), __builtin_fabsf(next_velocity.w)));
            K = 1 / md;
            next_velocity = next_velocity * K;
            next_acceleration = next_acceleration * K * K;
        }
        const float old_dlambda = running;
        running *= K;
#ifdef ADAPTIVE_PRECISION
        if (ar < new_max) {
            float suggested;
            float diff = acceleration_to_precision(next_acceleration, max_accel, suggested);
            float nds = 0.99f * ds * clampf(suggested / ds, 0.3f, 2.f);
            nds = __builtin_fmaxf(nds, min_step);
            next_ds = nds;
#ifdef SINGULARITY_DETECTION
            if (nds == min_step && (diff / 65536.f) > max_accel * 10000) should_break = true;
#endif
            if (nds < ds / 1.95f) continue;   // here a rejected step consumes a loop iteration (cl.cl:4849-4850)
        }
#endif
#ifndef UNCONDITIONALLY_NONSINGULAR
        if (__builtin_fabsf(velocity.x / running) > 1000 + f_in_x && __builtin_fabsf(acceleration.x / running) > 100) should_break = true;
#endif
        float4 generic_position_out = position;
        float4 generic_velocity_out = velocity / old_dlambda;
#ifdef GENERIC_CONSTANT_THETA
        {   // undo the equatorial-plane rotation (cl.cl:4864-4902)
            float4 pos_sph = gm::generic_to_spherical(position, cfg);
            float4 vel_sph = gm::generic_velocity_to_spherical_velocity(position, velocity / old_dlambda, cfg);
            float sgn = fsign(pos_sph.y);
            pos_sph.y = __builtin_fabsf(pos_sph.y);
            float3 pos_cart = rot_quat(polar_to_cartesian(yzw(pos_sph)), quat);
            float3 vel_cart = rot_quat(spherical_velocity_to_cartesian_velocity(yzw(pos_sph), yzw(vel_sph)), quat);
            float3 next_pos_sph = cartesian_to_polar(pos_cart);
            float3 next_vel_sph = cartesian_velocity_to_polar_velocity(pos_cart, vel_cart);
            if (sgn < 0) next_pos_sph.x = -next_pos_sph.x;
            float4 next_pos_generic = gm::spherical_to_generic(f4(pos_sph.x, next_pos_sph), cfg);
            float4 next_vel_generic = gm::spherical_velocity_to_generic_velocity(f4(pos_sph.x, next_pos_sph), f4(vel_sph.x, next_vel_sph), cfg);
            if (i != 0) next_pos_generic = periodic_diff(next_pos_generic, last_pos_generic, periods) + last_pos_generic;
            last_pos_generic = next_pos_generic;
            generic_position_out = next_pos_generic;
            generic_velocity_out = next_vel_generic;
        }
#endif
        if (degenerate4(next_position) || degenerate4(next_velocity) || degenerate4(next_acceleration)) break;
        position = next_position;
        velocity = next_velocity;
        acceleration = next_acceleration;
        positions_out[bufc * stride + id] = generic_position_out;
        if (velocities_out) velocities_out[bufc * stride + id] = generic_velocity_out;
        if (ds_out) ds_out[bufc * stride + id] = ds * old_dlambda;
        bufc++;
        if (should_break) break;
    }
    count_out[id] = bufc;
}

// parallel_transport_get_velocity, cl.cl:2164-2207: dX^a/dlambda = -Gamma^a_bs X^b Y^s, Gamma contracted from g and dg
__device__ float4 parallel_transport_velocity(float4 X, float4 position, float4 velocity, cfg_t cfg) {
    float g[16], dg[64], ginv[16];
    gm::metric_big_at(position, g, cfg);
    gm::partials_big_at(position, dg, cfg);
    matrix_inverse4(g, ginv);
    float Xa[4] = {X.x, X.y, X.z, X.w}, Ya[4] = {velocity.x, velocity.y, velocity.z, velocity.w};
    float out[4];
    for (int a = 0; a < 4; a++) {
        float sum = 0;
        for (int b = 0; b < 4; b++)
            for (int s = 0; s < 4; s++) {
                float gam = 0;
                for (int m = 0; m < 4; m++)
                    gam += ginv[a * 4 + m] * (dg[s * 16 + m * 4 + b] + dg[b * 16 + m * 4 + s] - dg[m * 16 + b * 4 + s]);
                sum += 0.5f * gam * Xa[b] * Ya[s];
            }
        out[a] = -sum;
    }
    return f4(out[0], out[1], out[2], out[3]);
}

extern "C" __global__ void gr_parallel_transport_quantity(const float4* __restrict__ geodesic_path, const float4* __restrict__ geodesic_velocity,
                                                          const float* __restrict__ ds_in, const float4* __restrict__ quantity,
                                                          const int* __restrict__ count_in, int count, float4* __restrict__ quantity_out, cfg_t cfg) {
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= count) return;
    int cnt = count_in[id];
    if (cnt == 0) return;
    const int stride = count;
    float4 current = quantity[id];
    quantity_out[id] = current;
    if (cnt == 1) return;
    for (int kk = 0; kk < cnt - 1; kk++) {
        int cur = kk * stride + id, nxt = (kk + 1) * stride + id;
        float ds = ds_in[cur];
        // second-order (Heun) step of the transport equation
        float4 f_x = parallel_transport_velocity(current, geodesic_path[cur], geodesic_velocity[cur], cfg);
        float4 predictor = current + f_x * ds;
        float4 next = current + (0.5f * ds) * (f_x + parallel_transport_velocity(predictor, geodesic_path[nxt], geodesic_velocity[nxt], cfg));
        quantity_out[cur] = current;
        current = next;
    }
    quantity_out[(cnt - 1) * stride + id] = current;
}

__device__ __forceinline__ float4 mix4(float4 a, float4 b, float t) { return a + (b - a) * t; }

extern "C" __global__ void gr_handle_interpolating_geodesic(const float4* __restrict__ geodesic_path, const float4* __restrict__ geodesic_velocity,
                                                            const float* __restrict__ ds_in, float4* __restrict__ camera_generic_out,
                                                            const float4* __restrict__ t_e0, const float4* __restrict__ t_e1,
                                                            const float4* __restrict__ t_e2, const float4* __restrict__ t_e3,
                                                            float4* __restrict__ e0_out, float4* __restrict__ e1_out,
                                                            float4* __restrict__ e2_out, float4* __restrict__ e3_out, float target_time,
                                                            const int* __restrict__ count_in, int parallel_transport_observer,
                                                            const float4* __restrict__ basis_speed, float4* __restrict__ interpolated_velocity,
                                                            cfg_t cfg) {
    if (blockIdx.x * blockDim.x + threadIdx.x != 0) return;
    int cnt = *count_in;
    if (cnt == 0) return;
    const float3 speed = f3(basis_speed->x, basis_speed->y, basis_speed->z);
    auto store = [&](const tetrad& t) { *e0_out = t.e[0]; *e1_out = t.e[1]; *e2_out = t.e[2]; *e3_out = t.e[3]; };
    auto store_from = [&](int i) { *e0_out = t_e0[i]; *e1_out = t_e1[i]; *e2_out = t_e2[i]; *e3_out = t_e3[i]; };
    if (!parallel_transport_observer) {
        tetrad t;
        calculate_tetrads(geodesic_path[0], speed, t, cfg, 1);
        store(t);
    } else {
        store_from(0);
    }
    float proper_time = 0;
    *camera_generic_out = geodesic_path[0];
    *interpolated_velocity = geodesic_velocity[0];
    if (cnt == 1) return;
    for (int i = 0; i < cnt - 1; i++) {
        float next_proper_time = proper_time + ds_in[i];
        if ((target_time >= proper_time && target_time < next_proper_time) || target_time < proper_time) {
            float dx = (target_time - proper_time) / (next_proper_time - proper_time);
            if (target_time < proper_time) dx = 0;
            float4 fin = mix4(geodesic_path[i], geodesic_path[i + 1], dx);
            *camera_generic_out = fin;
            tetrad t;
            t.e[0] = mix4(t_e0[i], t_e0[i + 1], dx);
            t.e[1] = mix4(t_e1[i], t_e1[i + 1], dx);
            t.e[2] = mix4(t_e2[i], t_e2[i + 1], dx);
            t.e[3] = mix4(t_e3[i], t_e3[i + 1], dx);
            if (!parallel_transport_observer) calculate_tetrads(fin, speed, t, cfg, 1);
            *interpolated_velocity = mix4(geodesic_velocity[i], geodesic_velocity[i + 1], dx);
            store(t);
            return;
        }
        proper_time = next_proper_time;
    }
    *camera_generic_out = geodesic_path[cnt - 1];
    *interpolated_velocity = geodesic_velocity[cnt - 1];
    if (!parallel_transport_observer) {
        tetrad t;
        calculate_tetrads(geodesic_path[cnt - 1], speed, t, cfg, 1);
        store(t);
    } else {
        store_from(cnt - 1);
    }
}
