// restate.cpp — TEST INFRASTRUCTURE ONLY: the CPU oracle.  Never linked into, imported by or called
// from the product (libgeodesic_hip.so / the HIP kernels); only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may build and run it.
//
// A plain C++ restatement of the reference's per-pixel geodesic pipeline (cl.cl), specialised at
// compile time by the same `-D` macro set the reference feeds its OpenCL compiler
// (metric.hpp:725-959): oracle/build_restate.py runs `g++ @macros.rsp restate.cpp`.  Every function
// cites the reference lines it follows.  It mirrors the reference's structure (array-of-structs
// rays, Christoffel symbols contracted numerically from F*_P at ray set-up, one function per
// kernel) rather than the HIP kernels' fused/register design, so agreement between the two is a
// real check.  The oracle is held to tests/golden/*.npz, which were produced by the reference's own
// cl.cl compiled for x86-64 (oracle/build_ref.py, tests/golden/make_golden.py).
//
// PARITY UNPINNED, formally: the reference holds no golden vectors for this path, and that build of
// cl.cl needs two things this repository wrote - the -D macro strings (the reference's generator
// depends on un-vendored submodules) and the OpenCL built-in library oracle/ref_shim.cpp.  A build
// with stand-ins pins nothing by the rules of this exercise.  What backs the fixtures beyond that:
// tests/test_oracle.py compiles the same cl.cl with macro strings derived independently by sympy
// (tools/sympy_macros.py) and requires the same frames; EXPERIMENTS.md C.4.
//
// Exports the same ref_* driver functions as oracle/ref_shim.cpp, so oracle/refpipe.py drives both.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace {

const float PIf = 3.14159274101257324f;
const double PId = 3.14159265358979323846;

struct v3 { float x, y, z; };
struct v4 { float x, y, z, w; };
inline v3 operator+(v3 a, v3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline v3 operator-(v3 a, v3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline v3 operator-(v3 a) { return {-a.x, -a.y, -a.z}; }
inline v3 operator*(v3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline v3 operator*(float s, v3 a) { return {a.x * s, a.y * s, a.z * s}; }
inline v3 operator/(v3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline v4 operator+(v4 a, v4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline v4 operator-(v4 a, v4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
inline v4 operator-(v4 a) { return {-a.x, -a.y, -a.z, -a.w}; }
inline v4 operator*(v4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
inline v4 operator*(float s, v4 a) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
inline v4 operator*(v4 a, v4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
inline v4 operator/(v4 a, float s) { return {a.x / s, a.y / s, a.z / s, a.w / s}; }
inline float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float dot(v4 a, v4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
inline v3 cross(v3 a, v3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float length(v3 a) { return std::sqrt(dot(a, a)); }
inline v3 normalize(v3 a) { return a / length(a); }
inline v4 normalize(v4 a) { return a / std::sqrt(dot(a, a)); }
inline v3 yzw(v4 a) { return {a.y, a.z, a.w}; }
inline v4 mk4(float x, v3 v) { return {x, v.x, v.y, v.z}; }
inline float signf(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }
inline float clampf(float v, float lo, float hi) { return std::fmin(std::fmax(v, lo), hi); }
inline float mixf(float a, float b, float t) { return a + (b - a) * t; }
inline bool degenerate(float x) { return std::isnan(x) || !std::isfinite(x); }   // IS_DEGENERATE, cl.cl:68
inline bool degenerate(v4 v) { return degenerate(v.x) || degenerate(v.y) || degenerate(v.z) || degenerate(v.w); }

// cl.cl:813-824
struct lightray {
    v4 position, velocity, initial_quat, acceleration;
    float ku_uobsu, running_dlambda_dnew;
    int terminated, sx, sy;
    int pad[3];
};
static_assert(sizeof(lightray) == 96, "lightray layout");

// cl.cl:5066-5074
struct render_data {
    float tex_x, tex_y;
    float z_shift;
    int sx, sy, terminated, side;
    int pad;
};
static_assert(sizeof(render_data) == 32, "render_data layout");

// cl.cl:940-956
struct dynamic_config {
#ifdef DYNVARS
    float DYNVARS;
#else
    float unused;
#endif
};
struct dynamic_feature_config {
#ifdef DYNAMIC_FLOAT_FEATURES
    float DYNAMIC_FLOAT_FEATURES;
#endif
#ifdef DYNAMIC_BOOL_FEATURES
    int DYNAMIC_BOOL_FEATURES;
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
typedef const dynamic_config* cfg_t;
typedef const dynamic_feature_config* dfg_t;

// ---- generated expressions -----------------------------------------------------------------------
namespace gen {
inline float sin(float x) { return std::sin(x); }
inline float cos(float x) { return std::cos(x); }
inline float tan(float x) { return std::tan(x); }
inline float asin(float x) { return std::asin(x); }
inline float acos(float x) { return std::acos(x); }
inline float atan(float x) { return std::atan(x); }
inline float atan2(float y, float x) { return std::atan2(y, x); }
inline float exp(float x) { return std::exp(x); }
inline float log(float x) { return std::log(x); }
inline float sqrt(float x) { return std::sqrt(x); }
inline float fabs(float x) { return std::fabs(x); }
inline float sinh(float x) { return std::sinh(x); }
inline float cosh(float x) { return std::cosh(x); }
inline float tanh(float x) { return std::tanh(x); }
inline float pow(float x, float y) { return std::pow(x, y); }
inline float fmod(float x, float y) { return std::fmod(x, y); }
inline float fmin(float x, float y) { return std::fmin(x, y); }
inline float fmax(float x, float y) { return std::fmax(x, y); }
inline float sign(float x) { return signf(x); }

#define POSITION_VARS(p)                                                                          \
    const float v1 = (p).x, v2 = (p).y, v3 = (p).z, v4 = (p).w;                                     \
    const float rs = RS_IMPL, c = C_IMPL;                                                          \
    (void)v1; (void)v2; (void)v3; (void)v4; (void)rs; (void)c;

// calculate_metric_generic(_big), cl.cl:969-985, 1023-1051: result always widened to 4x4
void metric_big(::v4 pos, float g[16], cfg_t cfg) {
    POSITION_VARS(pos)
    float TEMPORARIES0;
    for (int i = 0; i < 16; i++) g[i] = 0;
#ifndef GENERIC_BIG_METRIC
    g[0] = F1_I; g[5] = F2_I; g[10] = F3_I; g[15] = F4_I;
#else
    g[0] = F1_I; g[1] = F2_I; g[2] = F3_I; g[3] = F4_I;
    g[4] = g[1]; g[5] = F6_I; g[6] = F7_I; g[7] = F8_I;
    g[8] = g[2]; g[9] = g[6]; g[10] = F11_I; g[11] = F12_I;
    g[12] = g[3]; g[13] = g[7]; g[14] = g[11]; g[15] = F16_I;
#endif
}

// calculate_partial_derivatives_generic(_big), cl.cl:987-1015, 1053-1199; dg[k*16 + i*4 + j] = d g_ij / d v_k
void partials_big(::v4 pos, float dg[64], cfg_t cfg) {
    POSITION_VARS(pos)
    float TEMPORARIES0;
    for (int i = 0; i < 64; i++) dg[i] = 0;
#ifndef GENERIC_BIG_METRIC
    // diagonal layout g_partials[var*4 + wrt] (cl.cl:987-1015)
    const float p[16] = {F1_P, F2_P, F3_P, F4_P, F5_P, F6_P, F7_P, F8_P, F9_P, F10_P, F11_P, F12_P, F13_P, F14_P, F15_P, F16_P};
    for (int var = 0; var < 4; var++)
        for (int wrt = 0; wrt < 4; wrt++) dg[wrt * 16 + var * 4 + var] = p[var * 4 + wrt];
#else
    // upper triangle only, mirrored (cl.cl:1102-1197)
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

// GEO_ACCELn inside step_verlet, cl.cl:3279-3309
::v4 geo_accel(::v4 pos, ::v4 vel, cfg_t cfg) {
    float v1 = pos.x, v2 = pos.y, v3 = pos.z, v4 = pos.w;
    float iv1 = vel.x, iv2 = vel.y, iv3 = vel.z, iv4 = vel.w;
    const float rs = RS_IMPL, c = C_IMPL;
    (void)rs; (void)c; (void)v1; (void)v2; (void)v3; (void)v4; (void)iv1; (void)iv2; (void)iv3; (void)iv4;
    float TEMPORARIES0;
#ifdef GENERIC_CONSTANT_THETA
    v3 = (float)(PId / 2);
    iv3 = 0;
#endif
    ::v4 a;
    a.x = GEO_ACCEL0;
    a.y = GEO_ACCEL1;
#ifndef GENERIC_CONSTANT_THETA
    a.z = GEO_ACCEL2;
#else
    a.z = 0;
#endif
    a.w = GEO_ACCEL3;
    return a;
}

::v4 to_spherical(::v4 in, cfg_t cfg) { POSITION_VARS(in) return {TO_COORD1, TO_COORD2, TO_COORD3, TO_COORD4}; }             // cl.cl:1202-1215
::v4 from_spherical(::v4 in, cfg_t cfg) { POSITION_VARS(in) return {FROM_COORD1, FROM_COORD2, FROM_COORD3, FROM_COORD4}; }   // cl.cl:1237-1250
::v4 velocity_to_spherical(::v4 in, ::v4 d, cfg_t cfg) {                                                                        // cl.cl:1217-1235
    POSITION_VARS(in)
    const float dv1 = d.x, dv2 = d.y, dv3 = d.z, dv4 = d.w;
    (void)dv1; (void)dv2; (void)dv3; (void)dv4;
    return {TO_DCOORD1, TO_DCOORD2, TO_DCOORD3, TO_DCOORD4};
}
::v4 velocity_from_spherical(::v4 in, ::v4 d, cfg_t cfg) {                                                                      // cl.cl:1252-1270
    POSITION_VARS(in)
    const float dv1 = d.x, dv2 = d.y, dv3 = d.z, dv4 = d.w;
    (void)dv1; (void)dv2; (void)dv3; (void)dv4;
    return {FROM_DCOORD1, FROM_DCOORD2, FROM_DCOORD3, FROM_DCOORD4};
}
float distance_to_object(::v4 polar, cfg_t cfg) { POSITION_VARS(polar) return DISTANCE_FUNC; }                                  // cl.cl:3377-3387
}  // namespace gen

// ---- coordinate helpers (cl.cl:103-140, 185-204) ---------------------------------------------------
v3 cartesian_to_polar(v3 in) {
    float r = length(in);
    return {r, std::acos(in.z / r), std::atan2(in.y, in.x)};
}
v3 polar_to_cartesian(v3 in) {
    return {in.x * std::sin(in.y) * std::cos(in.z), in.x * std::sin(in.y) * std::sin(in.z), in.x * std::cos(in.y)};
}
v3 cartesian_velocity_to_polar_velocity(v3 p, v3 v) {
    float r = length(p);
    float repeated_eq = r * std::sqrt(1 - (p.z * p.z / (r * r)));
    float rdot = (p.x * v.x + p.y * v.y + p.z * v.z) / r;
    float tdot = ((p.z * rdot) / (r * repeated_eq)) - v.z / repeated_eq;
    float pdot = (p.x * v.y - p.y * v.x) / (p.x * p.x + p.y * p.y);
    return {rdot, tdot, pdot};
}
v3 spherical_velocity_to_cartesian_velocity(v3 p, v3 dp) {
    float r = p.x, dr = dp.x, x = p.y, dx = dp.y, y = p.z, dy = dp.z;
    float v1 = -r * std::sin(x) * std::sin(y) * dy + r * std::cos(x) * std::cos(y) * dx + std::sin(x) * std::cos(y) * dr;
    float v2 = std::sin(x) * std::sin(y) * dr + r * std::sin(x) * std::cos(y) * dy + r * std::cos(x) * std::sin(y) * dx;
    float v3_ = std::cos(x) * dr - r * std::sin(x) * dx;
    return {v1, v2, v3_};
}
// rot_quat, cl.cl:176-183
v3 rot_quat(v3 point, v4 quat) {
    quat = normalize(quat);
    v3 q{quat.x, quat.y, quat.z};
    v3 t = 2.f * cross(q, point);
    return point + quat.w * t + cross(q, t);
}

// ---- metric algebra (cl.cl:469-558, 830-907) ---------------------------------------------------------
v4 lower_index(v4 v, const float g[16]) {
    return {g[0] * v.x + g[1] * v.y + g[2] * v.z + g[3] * v.w, g[4] * v.x + g[5] * v.y + g[6] * v.z + g[7] * v.w,
            g[8] * v.x + g[9] * v.y + g[10] * v.z + g[11] * v.w, g[12] * v.x + g[13] * v.y + g[14] * v.z + g[15] * v.w};
}
float dot_metric(v4 u, v4 v, const float g[16]) { return dot(lower_index(u, g), v); }

// inverse of a general 4x4 via Gauss-Jordan with partial pivoting (role of metric_inverse / matrix_inverse, cl.cl:469-683)
void inverse4(const float m[16], float out[16]) {
    double a[4][8];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) { a[i][j] = m[i * 4 + j]; a[i][4 + j] = i == j ? 1.0 : 0.0; }
    for (int col = 0; col < 4; col++) {
        int piv = col;
        for (int r = col + 1; r < 4; r++)
            if (std::fabs(a[r][col]) > std::fabs(a[piv][col])) piv = r;
        for (int j = 0; j < 8; j++) std::swap(a[col][j], a[piv][j]);
        double d = a[col][col];
        for (int j = 0; j < 8; j++) a[col][j] /= d;
        for (int r = 0; r < 4; r++) {
            if (r == col) continue;
            double f = a[r][col];
            for (int j = 0; j < 8; j++) a[r][j] -= f * a[col][j];
        }
    }
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) out[i * 4 + j] = (float)a[i][4 + j];
}

// calculate_acceleration(_big), cl.cl:738-797, 1443-1537: Christoffel symbols from dg, then -Gamma v v
v4 acceleration_from_partials(v4 vel, const float g[16], const float dg[64]) {
#ifdef IS_CONSTANT_THETA
    vel.z = 0;
#endif
    float ginv[16];
    inverse4(g, ginv);
    float v[4] = {vel.x, vel.y, vel.z, vel.w};
    float res[4];
    for (int i = 0; i < 4; i++) {
        float sum_i = 0;
        for (int k = 0; k < 4; k++)
            for (int l = 0; l < 4; l++) {
                float sum = 0;
                for (int m = 0; m < 4; m++) {
                    sum += ginv[i * 4 + m] * dg[l * 16 + m * 4 + k];
                    sum += ginv[i * 4 + m] * dg[k * 16 + m * 4 + l];
                    sum -= ginv[i * 4 + m] * dg[m * 16 + k * 4 + l];
                }
                sum_i += v[k] * v[l] * (0.5f * sum);
            }
        res[i] = -sum_i;
    }
    v4 acc{res[0], res[1], res[2], res[3]};
#ifdef IS_CONSTANT_THETA
    acc.z = 0;
#endif
    return acc;
}

// ---- tetrads (cl.cl:1647-1861, 2072-2114, 2210-2224, 2288-2439) ---------------------------------------
v4 gram_proj(v4 u, v4 v, const float g[16]) { return (dot_metric(u, v, g) / dot_metric(u, u, g)) * u; }
v4 normalise_metric(v4 v, const float g[16]) { return v / std::sqrt(std::fabs(dot_metric(v, v, g))); }

int frame_basis_with_swap(const float g[16], int index_swap, v4 out[4]) {
    v4 arr[4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    float lengths[4] = {g[0], g[5], g[10], g[15]};
    std::swap(arr[0], arr[index_swap]);
    std::swap(lengths[0], lengths[index_swap]);
    int indices[4] = {0, 1, 2, 3};
    int first_nonzero = -1;
    for (int i = 0; i < 4; i++)
        if (!(std::fabs(lengths[i] - 0.f) <= 0.00001f)) { first_nonzero = i; break; }
    if (first_nonzero == -1) first_nonzero = 0;
    if (first_nonzero != 0) { std::swap(arr[0], arr[first_nonzero]); std::swap(indices[0], indices[first_nonzero]); }
    v4 u1 = arr[0];
    v4 u2 = arr[1]; u2 = u2 - gram_proj(u1, u2, g);
    v4 u3 = arr[2]; u3 = u3 - gram_proj(u1, u3, g); u3 = u3 - gram_proj(u2, u3, g);
    v4 u4 = arr[3]; u4 = u4 - gram_proj(u1, u4, g); u4 = u4 - gram_proj(u2, u4, g); u4 = u4 - gram_proj(u3, u4, g);
    v4 res[4] = {normalise_metric(u1, g), normalise_metric(u2, g), normalise_metric(u3, g), normalise_metric(u4, g)};
    v4 sorted[4];
    for (int i = 0; i < 4; i++) sorted[indices[i]] = res[i];
    int lowest = -1;
    float lowest_value = 0;
    for (int i = 0; i < 4; i++) {
        float d = dot_metric(sorted[i], sorted[i], g);
        if (d < lowest_value) { lowest = i; lowest_value = d; }
    }
    int which = lowest != -1 ? lowest : 0;
    if (which > 0) std::swap(sorted[0], sorted[which]);
    for (int i = 0; i < 4; i++) out[i] = sorted[i];
    return which;
}

void frame_basis(const float g[16], v4 out[4]) {
    int t = frame_basis_with_swap(g, 0, out);
    if (t == 0) return;
    frame_basis_with_swap(g, t, out);
}

v3 project3(v3 u, v3 v) { return (dot(u, v) / dot(u, u)) * u; }

void calculate_tetrads(v4 at_metric, v3 speed, v4 e[4], cfg_t cfg, int should_orient) {
    v4 polar_camera = gen::to_spherical(at_metric, cfg);
    if (degenerate(at_metric)) {
        e[0] = {1, 0, 0, 0}; e[1] = {0, 1, 0, 0}; e[2] = {0, 0, 1, 0}; e[3] = {0, 0, 0, 1};
        return;
    }
    float g[16];
    gen::metric_big(at_metric, g, cfg);
    v4 b[4];
    frame_basis(g, b);
    v4 e0 = b[0], e1 = b[1], e2 = b[2], e3 = b[3];
    if (should_orient) {
        v3 apolar = yzw(polar_camera);
        apolar.x = std::fabs(apolar.x);
        v3 cart_camera = polar_to_cartesian(apolar);
        float m[16] = {e0.x, e1.x, e2.x, e3.x, e0.y, e1.y, e2.y, e3.y, e0.z, e1.z, e2.z, e3.z, e0.w, e1.w, e2.w, e3.w};
        float inv[16];
        inverse4(m, inv);
        v4 lo[4];
        for (int i = 0; i < 4; i++) lo[i] = {inv[i * 4 + 0], inv[i * 4 + 1], inv[i * 4 + 2], inv[i * 4 + 3]};
        v3 sx = cartesian_velocity_to_polar_velocity(cart_camera, {1, 0, 0});
        v3 sy = cartesian_velocity_to_polar_velocity(cart_camera, {0, 1, 0});
        v3 sz = cartesian_velocity_to_polar_velocity(cart_camera, {0, 0, 1});
        if (polar_camera.y < 0) { sx.x = -sx.x; sy.x = -sy.x; sz.x = -sz.x; }
        v4 gx = gen::velocity_from_spherical(polar_camera, mk4(0, sx), cfg);
        v4 gy = gen::velocity_from_spherical(polar_camera, mk4(0, sy), cfg);
        v4 gz = gen::velocity_from_spherical(polar_camera, mk4(0, sz), cfg);
        auto to_tetrad = [&](v4 v) { return v4{dot(lo[0], v), dot(lo[1], v), dot(lo[2], v), dot(lo[3], v)}; };
        v4 tE1 = to_tetrad(gy), tE2 = to_tetrad(gx), tE3 = to_tetrad(gz);   // y first (cl.cl:2378)
        v3 u1 = yzw(tE1), u2 = yzw(tE2), u3 = yzw(tE3);
        u2 = u2 - project3(u1, u2);
        u3 = u3 - project3(u1, u3);
        u3 = u3 - project3(u2, u3);
        u1 = normalize(u1); u2 = normalize(u2); u3 = normalize(u3);
        v4 x_out = u2.x * e1 + u2.y * e2 + u2.z * e3;
        v4 y_out = u1.x * e1 + u1.y * e2 + u1.z * e3;
        v4 z_out = u3.x * e1 + u3.y * e2 + u3.z * e3;
        e1 = x_out; e2 = y_out; e3 = z_out;
    }
    {
        float vv = dot(speed, speed);
        float Y = 1 / std::sqrt(1 - vv);
        v4 obs = Y * e0 + (Y * speed.x) * e1 + (Y * speed.y) * e2 + (Y * speed.z) * e3;
        v4 lT4 = lower_index(e0, g), lu4 = lower_index(obs, g);
        float T[4] = {e0.x, e0.y, e0.z, e0.w}, lT[4] = {lT4.x, lT4.y, lT4.z, lT4.w};
        float uo[4] = {obs.x, obs.y, obs.z, obs.w}, luo[4] = {lu4.x, lu4.y, lu4.z, lu4.w};
        float gamma = -dot(lT4, obs);
        float L[16];
        for (int u = 0; u < 4; u++)
            for (int v = 0; v < 4; v++)
                L[u * 4 + v] = (u == v ? 1.f : 0.f) + (1 / (1 + gamma)) * (T[u] + uo[u]) * (lT[v] + luo[v]) - 2 * uo[u] * lT[v];
        auto contract = [&](v4 v) {
            return v4{L[0] * v.x + L[1] * v.y + L[2] * v.z + L[3] * v.w, L[4] * v.x + L[5] * v.y + L[6] * v.z + L[7] * v.w,
                      L[8] * v.x + L[9] * v.y + L[10] * v.z + L[11] * v.w, L[12] * v.x + L[13] * v.y + L[14] * v.z + L[15] * v.w};
        };
        e0 = obs; e1 = contract(e1); e2 = contract(e2); e3 = contract(e3);
    }
    e[0] = e0; e[1] = e1; e[2] = e2; e[3] = e3;
}

// ---- ray set-up (cl.cl:2015-2059, 2949-3065, 3143-3251) ---------------------------------------------
v3 pixel_direction(int cx, int cy, float width, float height, v4 camera_quat, dfg_t dfg) {
    float fov = GET_FEATURE(field_of_view, dfg);
    float fov_rad = (fov / 360.f) * 2 * PIf;
    float f_stop = (width / 2) / std::tan(fov_rad / 2);
    v3 d = normalize(v3{cx - width / 2, cy - height / 2, f_stop});
    return rot_quat(d, camera_quat);
}

v4 theta_adjustment_quat(v3 pixel_dir, v4 polar_camera, float angle_sign) {
    if (length(pixel_dir) < 0.00001f) pixel_dir = {0, 1, 0};
    v3 apolar = yzw(polar_camera);
    apolar.x = std::fabs(apolar.x);
    v3 cam = polar_to_cartesian(apolar);
    v3 bx = normalize(pixel_dir);
    v3 by = normalize(-cam);
    bx = normalize(normalize(bx - dot(bx, by) * by));
    v3 plane_n = -normalize(cross(bx, by));
    float angle_to_flat = std::acos(dot(plane_n, v3{0, 0, 1}));
    v3 axis = normalize(cross(plane_n, v3{0, 0, 1}));
    float angle = angle_to_flat * angle_sign;
    float s = std::sin(angle / 2);
    return normalize(v4{axis.x * s, axis.y * s, axis.z * s, std::cos(angle / 2)});
}

lightray render_ray(int cx, int cy, v4 position, v4 velocity, v4 observer_velocity, cfg_t cfg) {
    v4 inverse_quat{0, 0, 0, 1};
#ifdef GENERIC_CONSTANT_THETA
    {   // correct_lightray, cl.cl:2949-2997
        v4 polar_pos = gen::to_spherical(position, cfg);
        v4 pos_sph = polar_pos;
        v4 vel_sph = gen::velocity_to_spherical(position, velocity, cfg);
        float sgn = signf(pos_sph.y);
        pos_sph.y = std::fabs(pos_sph.y);
        v3 pos_cart = polar_to_cartesian(yzw(pos_sph));
        v3 vel_cart = spherical_velocity_to_cartesian_velocity(yzw(pos_sph), yzw(vel_sph));
        v4 quat = theta_adjustment_quat(vel_cart, polar_pos, 1);
        inverse_quat = theta_adjustment_quat(vel_cart, polar_pos, -1);
        pos_cart = rot_quat(pos_cart, quat);
        vel_cart = rot_quat(vel_cart, quat);
        v3 next_pos = cartesian_to_polar(pos_cart);
        v3 next_vel = cartesian_velocity_to_polar_velocity(pos_cart, vel_cart);
        if (sgn < 0) next_pos.x = -next_pos.x;
        position = gen::from_spherical(mk4(pos_sph.x, next_pos), cfg);
        velocity = gen::velocity_from_spherical(mk4(pos_sph.x, next_pos), mk4(vel_sph.x, next_vel), cfg);
    }
#endif
#ifdef IS_CONSTANT_THETA
    position.z = PIf / 2;
    velocity.z = 0;
#endif
    float g[16], dg[64];
    gen::metric_big(position, g, cfg);
    gen::partials_big(position, dg, cfg);
    lightray ray;
    std::memset(&ray, 0, sizeof(ray));
    ray.position = position;
    ray.velocity = velocity;
    ray.acceleration = acceleration_from_partials(velocity, g, dg);
    ray.initial_quat = inverse_quat;
    ray.running_dlambda_dnew = 1;
    ray.terminated = 0;
    ray.ku_uobsu = dot(velocity, lower_index(observer_velocity, g));
    ray.sx = cx;
    ray.sy = cy;
    return ray;
}

lightray pixel_ray(int cx, int cy, int width, int height, v4 camera, v4 quat, const v4 e[4], int flip, cfg_t cfg, dfg_t dfg) {
    v3 dir = normalize(pixel_direction(cx, cy, (float)width, (float)height, quat, dfg));
#ifndef FORWARD_GEODESIC_PATH
    v4 pixel_t = -e[0];
#else
    v4 pixel_t = e[0];
#endif
    if (flip) pixel_t = -pixel_t;
    v4 velocity = dir.x * e[1] + dir.y * e[2] + dir.z * e[3] + pixel_t;
    return render_ray(cx, cy, camera, velocity, e[0], cfg);
}

int should_early_terminate(int x, int y, int w, int h, const int* term) {
    if (x < 0 || y < 0 || x > w - 1 || y > h - 1) return 0;
    return term[y * w + x] == 1;
}

// ---- integrator (cl.cl:3273-3346, 3400-3456, 3954-4247) ----------------------------------------------
#ifdef ADAPTIVE_PRECISION
float acceleration_to_precision(v4 acc, float max_acceleration, float* next_ds_out) {
    float divisor = (float)std::max(std::max(W_V1, W_V2), std::max(W_V3, W_V4));
    v4 w{(float)(W_V1), (float)(W_V2), (float)(W_V3), (float)(W_V4)};
    v4 aw = acc * w;
    float current = std::sqrt(dot(aw, aw)) * 0.01f;
    current /= divisor;
    const float big = 256 * 256;
    float err = max_acceleration;
    float diff = current * big;
    float lowest = err * big / std::pow(100000.f, 2.f);
    if (diff < lowest) diff = lowest;
    *next_ds_out = std::sqrt((err * big) / diff);
    return diff;
}
#endif

// returns true when the ray terminated (terminated = 1 written)
bool trace_ray(lightray* ray, cfg_t cfg, dfg_t dfg, uint64_t* attempts) {
    v4 position = ray->position, velocity = ray->velocity, acceleration = ray->acceleration;
    float f_in_x = std::fabs(velocity.x);
#ifdef IS_CONSTANT_THETA
    position.z = PIf / 2; velocity.z = 0; acceleration.z = 0;
#endif
    float next_ds = 0.00001f;
#ifdef ADAPTIVE_PRECISION
    (void)acceleration_to_precision(acceleration, GET_FEATURE(max_acceleration_change, dfg), &next_ds);
#endif
    const float subambient = 0.5f, ambient = 0.2f;
    float running = 1;
    for (int i = 0; i < 4096 * 4; i++) {
#ifdef IS_CONSTANT_THETA
        position.z = PIf / 2; velocity.z = 0; acceleration.z = 0;
#endif
        float new_max = GET_FEATURE(max_precision_radius, dfg);
        float new_min = 3;
        v4 polar = gen::to_spherical(position, cfg);
#ifdef IS_CONSTANT_THETA
        polar.z = PIf / 2;
#endif
        float r_value = gen::distance_to_object(polar, cfg);
        float ds = mixf(ambient, subambient, (clampf(std::fabs(r_value), new_min, new_max) - new_min) / (new_max - new_min));
#ifdef ADAPTIVE_PRECISION
        ds = next_ds;
#endif
        if (std::fabs(r_value) < new_max) ds = std::fmin(ds, ambient);
        else ds = 0.1f * (std::fabs(r_value) - new_max) + ambient;
        bool should_terminate = std::fabs(polar.y) >= GET_FEATURE(universe_size, dfg);
#ifdef SINGULAR
        should_terminate |= std::fabs(polar.y) < SINGULAR_TERMINATOR;
#endif
#ifdef HAS_CYLINDRICAL_SINGULARITY
        if (position.y < CYLINDRICAL_TERMINATOR) return false;
#endif
#ifndef UNCONDITIONALLY_NONSINGULAR
        if (std::fabs(velocity.x / running) > 1000 + f_in_x && std::fabs(acceleration.x / running) > 100) return false;
#endif
        if (should_terminate) {
            ray->position = position;
            ray->velocity = velocity;
            ray->running_dlambda_dnew = running;
            ray->terminated = 1;
            return true;
        }
        // step_verlet
        if (attempts) (*attempts)++;
        v4 next_position = position + velocity * ds + 0.5f * acceleration * ds * ds;
        v4 half_velocity = velocity + acceleration * ds;
        v4 next_acceleration = gen::geo_accel(next_position, half_velocity, cfg);
        v4 next_velocity = velocity + 0.5f * (acceleration + next_acceleration) * ds;
        float K = 1 / std::fmax(std::fmax(std::fabs(next_velocity.x), std::fabs(next_velocity.y)),
                                std::fmax(std::fabs(next_velocity.z), std::fabs(next_velocity.w)));
        if (!GET_FEATURE(reparameterisation, dfg)) K = 1;
        next_velocity = next_velocity * K;
        next_acceleration = next_acceleration * K * K;
        running *= K;
#ifdef ADAPTIVE_PRECISION
        if (std::fabs(r_value) < new_max) {
            // calculate_ds_error
            float max_accel = GET_FEATURE(max_acceleration_change, dfg);
            float suggested = 0;
            float diff = acceleration_to_precision(next_acceleration, max_accel, &suggested);
            float nds = 0.99f * ds * clampf(suggested / ds, 0.3f, 2.f);
            float min_step = GET_FEATURE(min_step, dfg);
            nds = std::fmax(nds, min_step);
            next_ds = nds;
#ifdef SINGULARITY_DETECTION
            if (nds == min_step && (diff / (256 * 256)) > max_accel * 10000) return false;
#endif
            if (nds < ds / 1.95f) { i--; continue; }
        }
#endif
        position = next_position;
        velocity = next_velocity;
        acceleration = next_acceleration;
        if (degenerate(position) || degenerate(velocity) || degenerate(acceleration)) return false;
    }
    return false;
}

// ---- the same integrator in float64 -------------------------------------------------------------------
// Not a reference kernel: the reference's discrete algorithm (same steps, same controller, same thresholds, the same float
// parameters) evaluated in double precision, as a yardstick for rays on which two fp32 builds disagree.  Along a ray that grazes
// the polar axis of a Boyer-Lindquist chart every last-place difference is amplified (d phi / d lambda ~ 1 / sin^2 theta): the
// number of pixels in which one fp32 build differs from another then measures how many roundings they do not share, not an
// error of either - what each of them is off from this evaluation does (tests/test_gpu_parity.py, tools/polar_probe.py).
namespace gen64 {
inline double sin(double x) { return std::sin(x); }
inline double cos(double x) { return std::cos(x); }
inline double tan(double x) { return std::tan(x); }
inline double asin(double x) { return std::asin(x); }
inline double acos(double x) { return std::acos(x); }
inline double atan(double x) { return std::atan(x); }
inline double atan2(double y, double x) { return std::atan2(y, x); }
inline double exp(double x) { return std::exp(x); }
inline double log(double x) { return std::log(x); }
inline double sqrt(double x) { return std::sqrt(x); }
inline double fabs(double x) { return std::fabs(x); }
inline double sinh(double x) { return std::sinh(x); }
inline double cosh(double x) { return std::cosh(x); }
inline double tanh(double x) { return std::tanh(x); }
inline double pow(double x, double y) { return std::pow(x, y); }
inline double fmod(double x, double y) { return std::fmod(x, y); }
inline double fmin(double x, double y) { return std::fmin(x, y); }
inline double fmax(double x, double y) { return std::fmax(x, y); }
inline double sign(double x) { return x > 0 ? 1.0 : (x < 0 ? -1.0 : 0.0); }
struct d4 { double x, y, z, w; };
inline d4 operator+(d4 a, d4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline d4 operator*(d4 a, double s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
#define POSITION_VARS64(p)                                                                        \
    const double v1 = (p).x, v2 = (p).y, v3 = (p).z, v4 = (p).w;                                    \
    const double rs = RS_IMPL, c = C_IMPL;                                                         \
    (void)v1; (void)v2; (void)v3; (void)v4; (void)rs; (void)c;
d4 geo_accel(d4 pos, d4 vel, cfg_t cfg) {
#ifdef GENERIC_CONSTANT_THETA
    pos.z = (double)(PIf / 2);
    vel.z = 0;
#endif
    POSITION_VARS64(pos)
    const double iv1 = vel.x, iv2 = vel.y, iv3 = vel.z, iv4 = vel.w;
    (void)iv1; (void)iv2; (void)iv3; (void)iv4;
    double TEMPORARIES0;
    d4 a;
    a.x = GEO_ACCEL0;
    a.y = GEO_ACCEL1;
#ifndef GENERIC_CONSTANT_THETA
    a.z = GEO_ACCEL2;
#else
    a.z = 0;
#endif
    a.w = GEO_ACCEL3;
    return a;
}
d4 to_spherical(d4 in, cfg_t cfg) { POSITION_VARS64(in) return {TO_COORD1, TO_COORD2, TO_COORD3, TO_COORD4}; }
double distance_to_object(d4 polar, cfg_t cfg) { POSITION_VARS64(polar) return DISTANCE_FUNC; }
d4 from_spherical(d4 in, cfg_t cfg) { POSITION_VARS64(in) return {FROM_COORD1, FROM_COORD2, FROM_COORD3, FROM_COORD4}; }
d4 velocity_to_spherical(d4 in, d4 d, cfg_t cfg) {
    POSITION_VARS64(in)
    const double dv1 = d.x, dv2 = d.y, dv3 = d.z, dv4 = d.w;
    (void)dv1; (void)dv2; (void)dv3; (void)dv4;
    return {TO_DCOORD1, TO_DCOORD2, TO_DCOORD3, TO_DCOORD4};
}
d4 velocity_from_spherical(d4 in, d4 d, cfg_t cfg) {
    POSITION_VARS64(in)
    const double dv1 = d.x, dv2 = d.y, dv3 = d.z, dv4 = d.w;
    (void)dv1; (void)dv2; (void)dv3; (void)dv4;
    return {FROM_DCOORD1, FROM_DCOORD2, FROM_DCOORD3, FROM_DCOORD4};
}
}  // namespace gen64

// trace_ray in float64; returns the reference's `terminated` (1 reached the boundary, 0 otherwise) and the final position
int trace_ray_f64(const lightray& ray, double position_out[4], cfg_t cfg, dfg_t dfg) {
    using gen64::d4;
    auto widen = [](v4 v) { return d4{v.x, v.y, v.z, v.w}; };
    d4 position = widen(ray.position), velocity = widen(ray.velocity), acceleration = widen(ray.acceleration);
    const double f_in_x = std::fabs(velocity.x);
    const double half_pi = (double)(PIf / 2);
#ifdef ADAPTIVE_PRECISION
    const double max_accel = GET_FEATURE(max_acceleration_change, dfg), min_step = GET_FEATURE(min_step, dfg);
    auto precision = [&](d4 acc, double* next_ds_out) {
        const double divisor = (double)std::max(std::max(W_V1, W_V2), std::max(W_V3, W_V4));
        const double ax = acc.x * (double)(W_V1), ay = acc.y * (double)(W_V2), az = acc.z * (double)(W_V3), aw = acc.w * (double)(W_V4);
        double current = std::sqrt(ax * ax + ay * ay + az * az + aw * aw) * (double)0.01f / divisor;
        const double big = 256 * 256;
        double diff = current * big;
        const double lowest = max_accel * big / 1e10;
        if (diff < lowest) diff = lowest;
        *next_ds_out = std::sqrt((max_accel * big) / diff);
        return diff;
    };
#endif
#ifdef IS_CONSTANT_THETA
    position.z = half_pi; velocity.z = 0; acceleration.z = 0;
#endif
    double next_ds = (double)0.00001f;
#ifdef ADAPTIVE_PRECISION
    (void)precision(acceleration, &next_ds);
#endif
    const double subambient = 0.5, ambient = (double)0.2f;
    double running = 1;
    const double new_max = GET_FEATURE(max_precision_radius, dfg), new_min = 3, universe = GET_FEATURE(universe_size, dfg);
    for (int i = 0; i < 4096 * 4; i++) {
#ifdef IS_CONSTANT_THETA
        position.z = half_pi; velocity.z = 0; acceleration.z = 0;
#endif
        d4 polar = gen64::to_spherical(position, cfg);
#ifdef IS_CONSTANT_THETA
        polar.z = half_pi;
#endif
        const double ar = std::fabs(gen64::distance_to_object(polar, cfg));
        double ds = ambient + (subambient - ambient) * ((std::fmin(std::fmax(ar, new_min), new_max) - new_min) / (new_max - new_min));
#ifdef ADAPTIVE_PRECISION
        ds = next_ds;
#endif
        if (ar < new_max) ds = std::fmin(ds, ambient);
        else ds = (double)0.1f * (ar - new_max) + ambient;
        bool should_terminate = std::fabs(polar.y) >= universe;
#ifdef SINGULAR
        should_terminate |= std::fabs(polar.y) < SINGULAR_TERMINATOR;
#endif
#ifdef HAS_CYLINDRICAL_SINGULARITY
        if (position.y < CYLINDRICAL_TERMINATOR) return 0;
#endif
#ifndef UNCONDITIONALLY_NONSINGULAR
        if (std::fabs(velocity.x / running) > 1000 + f_in_x && std::fabs(acceleration.x / running) > 100) return 0;
#endif
        if (should_terminate) {
            position_out[0] = position.x; position_out[1] = position.y; position_out[2] = position.z; position_out[3] = position.w;
            return 1;
        }
        d4 next_position = position + velocity * ds + acceleration * (0.5 * ds * ds);
        d4 half_velocity = velocity + acceleration * ds;
        d4 next_acceleration = gen64::geo_accel(next_position, half_velocity, cfg);
        d4 next_velocity = velocity + (acceleration + next_acceleration) * (0.5 * ds);
        double K = 1 / std::fmax(std::fmax(std::fabs(next_velocity.x), std::fabs(next_velocity.y)),
                                 std::fmax(std::fabs(next_velocity.z), std::fabs(next_velocity.w)));
        if (!GET_FEATURE(reparameterisation, dfg)) K = 1;
        next_velocity = next_velocity * K;
        next_acceleration = next_acceleration * (K * K);
        running *= K;
#ifdef ADAPTIVE_PRECISION
        if (ar < new_max) {
            double suggested = 0;
            const double diff = precision(next_acceleration, &suggested);
            double nds = (double)0.99f * ds * std::fmin(std::fmax(suggested / ds, (double)0.3f), 2.0);
            nds = std::fmax(nds, min_step);
            next_ds = nds;
#ifdef SINGULARITY_DETECTION
            if (nds == min_step && (diff / (256 * 256)) > max_accel * 10000) return 0;
#endif
            if (nds < ds / (double)1.95f) { i--; continue; }
        }
#endif
        position = next_position;
        velocity = next_velocity;
        acceleration = next_acceleration;
        auto bad = [](d4 v) { return !std::isfinite(v.x) || !std::isfinite(v.y) || !std::isfinite(v.z) || !std::isfinite(v.w); };
        if (bad(position) || bad(velocity) || bad(acceleration)) return 0;
    }
    return 0;
}

// ---- render data (cl.cl:211-263, 5024-5100, 5135-5213) -------------------------------------------------
v3 fix_ray_position_cart(v3 pos, v3 vel, float radius) {
    vel = normalize(vel);
    float b = 2 * dot(vel, pos);
    float c = dot(pos, pos) - radius * radius;
    float discrim = b * b - 4 * c;
    if (discrim < 0) return pos;
    float t0 = (-b - std::sqrt(discrim)) / 2, t1 = (-b + std::sqrt(discrim)) / 2;
    float t = std::fabs(t0) < std::fabs(t1) ? t0 : t1;
    return pos + t * vel;
}

v3 fix_ray_position(v3 polar_pos, v3 polar_vel, float radius) {
    float sgn = signf(polar_pos.x);
    v3 cpos = polar_pos;
    cpos.x = std::fabs(cpos.x);
    polar_vel.x *= sgn;
    v3 cart_vel = spherical_velocity_to_cartesian_velocity(cpos, polar_vel);
    v3 cart_pos = polar_to_cartesian(cpos);
    v3 fixed = cartesian_to_polar(fix_ray_position_cart(cart_pos, cart_vel, radius));
#ifdef IS_CONSTANT_THETA
    fixed.y = PIf / 2;
#endif
    fixed.x *= sgn;
    return fixed;
}

v4 intersection_position(const lightray& ray, cfg_t cfg, dfg_t dfg) {
    v4 position = gen::to_spherical(ray.position, cfg);
    v4 velocity = gen::velocity_to_spherical(ray.position, ray.velocity, cfg);
#ifdef IS_CONSTANT_THETA
    position.z = PIf / 2;
    velocity.z = 0;
#endif
    float universe = GET_FEATURE(universe_size, dfg);
    if (std::fabs(position.y) >= universe) position = mk4(position.x, fix_ray_position(yzw(position), yzw(velocity), universe));
#if defined(SINGULAR) && defined(TRAVERSABLE_EVENT_HORIZON)
    if (std::fabs(position.y) < SINGULAR_TERMINATOR) position = mk4(position.x, fix_ray_position(yzw(position), yzw(velocity), SINGULAR_TERMINATOR));
#endif
    v3 npolar = yzw(position);
#ifdef GENERIC_CONSTANT_THETA
    npolar = cartesian_to_polar(rot_quat(polar_to_cartesian(yzw(position)), ray.initial_quat));
#endif
    return mk4(position.x, npolar);
}

void angle_to_tex(float theta, float phi, float* tx, float* ty) {
    float thetaf = std::fmod(theta, 2 * PIf);
    float phif = phi;
    if (thetaf >= PIf) { phif += PIf; thetaf -= PIf; }
    phif = std::fmod(phif, 2 * PIf);
    *tx = phif / (2 * PIf) + 0.5f;
    *ty = thetaf / PIf;
}

render_data make_render_data(const lightray& ray, cfg_t cfg, dfg_t dfg) {
    render_data dat;
    std::memset(&dat, 0, sizeof(dat));
    dat.terminated = ray.terminated;
    dat.sx = ray.sx;
    dat.sy = ray.sy;
    dat.side = 1;
    if (ray.terminated != 1) return dat;
    v4 position = intersection_position(ray, cfg, dfg);
    v4 generic_velocity = ray.velocity / ray.running_dlambda_dnew;
    dat.side = gen::to_spherical(ray.position, cfg).y < 0 ? 0 : 1;
#if !defined(TRAVERSABLE_EVENT_HORIZON)
    if (std::fabs(position.y) <= 1) return dat;
#endif
    v4 fe[4];
    calculate_tetrads(ray.position, v3{0, 0, 0}, fe, cfg, 0);
    float g[16];
    gen::metric_big(ray.position, g, cfg);
    v4 obvs_low = lower_index(fe[0], g);
    float z_shift = (dot(generic_velocity, obvs_low) / ray.ku_uobsu) - 1;
    dat.z_shift = std::fmax(z_shift, -0.999f);
    angle_to_tex(position.z, position.w, &dat.tex_x, &dat.tex_y);
    return dat;
}

// ---- shading (cl.cl:326-350, 3598-3610, 5366-5449, 5453-5846) -------------------------------------------
struct image {
    int width, height, levels;
    const uint8_t* texels;
};

void texel(const image& im, int x, int y, int layer, float out[4]) {
    const uint8_t* p = im.texels + (((size_t)layer * im.height + y) * im.width + x) * 4;
    for (int c = 0; c < 4; c++) out[c] = p[c] / 255.0f;
}

// read_imagef, normalized | repeat | linear (OpenCL 1.2 specification 8.2)
void sample(const image& im, float s, float t, float layer_f, float out[4]) {
    int layer = (int)std::rint(layer_f);
    layer = layer < 0 ? 0 : (layer > im.levels - 1 ? im.levels - 1 : layer);
    float u = (s - std::floor(s)) * im.width, v = (t - std::floor(t)) * im.height;
    int i0 = (int)std::floor(u - 0.5f), j0 = (int)std::floor(v - 0.5f);
    int i1 = i0 + 1, j1 = j0 + 1;
    if (i0 < 0) i0 += im.width;
    if (i1 > im.width - 1) i1 -= im.width;
    if (j0 < 0) j0 += im.height;
    if (j1 > im.height - 1) j1 -= im.height;
    // memory-safe whatever the coordinates, as an image read is (a NaN coordinate converts to INT_MIN here; the weights are NaN then)
    i0 = i0 < 0 ? 0 : (i0 > im.width - 1 ? im.width - 1 : i0);
    i1 = i1 < 0 ? 0 : (i1 > im.width - 1 ? im.width - 1 : i1);
    j0 = j0 < 0 ? 0 : (j0 > im.height - 1 ? im.height - 1 : j0);
    j1 = j1 < 0 ? 0 : (j1 > im.height - 1 ? im.height - 1 : j1);
    float a = (u - 0.5f) - std::floor(u - 0.5f), b = (v - 0.5f) - std::floor(v - 0.5f);
    float t00[4], t10[4], t01[4], t11[4];
    texel(im, i0, j0, layer, t00); texel(im, i1, j0, layer, t10); texel(im, i0, j1, layer, t01); texel(im, i1, j1, layer, t11);
    for (int c = 0; c < 4; c++) out[c] = (1 - a) * (1 - b) * t00[c] + a * (1 - b) * t10[c] + (1 - a) * b * t01[c] + a * b * t11[c];
}

void read_mipmap(const image& bg1, const image& bg2, int side, float px, float py, float lod, float out[4]) {
    lod = std::fmax(lod, 0.f);
    px = std::fmod(px, 1.f);
    py = std::fmod(py, 1.f);
    float lo = std::floor(lod), hi = std::ceil(lod);
    float dlo = std::pow(2.f, lo), dhi = std::pow(2.f, hi);
    float w = lod - lo;
    const image& bg = side >= 1 ? bg1 : bg2;
    float a[4], b[4];
    sample(bg, px / dlo, py / dlo, lo, a);
    sample(bg, px / dhi, py / dhi, hi, b);
    for (int c = 0; c < 4; c++) out[c] = a[c] + (b[c] - a[c]) * w;
}

float srgb_to_lin1(float in) { return in < 0.04045f ? in / 12.92f : std::pow((in + 0.055f) / 1.055f, 2.4f); }
float lin_to_srgb1(float in) { return in <= 0.0031308f ? in * 12.92f : 1.055f * std::pow(in, 1.0f / 2.4f) - 0.055f; }
float energy_of(v3 v) { return v.x * 0.2125f + v.y * 0.7154f + v.z * 0.0721f; }
v3 clamp3(v3 v, float lo, float hi) { return {clampf(v.x, lo, hi), clampf(v.y, lo, hi), clampf(v.z, lo, hi)}; }
v3 mix3(v3 a, v3 b, float t) { return a + (b - a) * t; }

v3 redshift(v3 v, float z, dfg_t dfg) {
    float radiant = energy_of(v);
    v3 red{1 / 0.2125f, 0.f, 0.f};
    v3 green{0, (float)(1 / 0.7154), 0.f};
    v3 blue{0.f, 0.f, (float)(1 / 0.0721)};
    v3 result;
    if (z > 0) {
        result = mix3(v, radiant * red, std::tanh(z));
    } else {
        float iv1pz = (1 / (1 + z)) - 1;
        v3 col = mix3(v, radiant * blue, std::tanh(iv1pz));
        if (!GET_FEATURE(use_old_redshift, dfg)) {
            float final_energy = energy_of(clamp3(col, 0.f, 1.f));
            float remaining = energy_of(col) - final_energy;
            col.x += remaining * (red.x + green.x);
            col.y += remaining * (red.y + green.y);
        }
        result = col;
    }
    return clamp3(result, 0.f, 1.f);
}

float circular_diff(float f1, float f2) {   // period 1, mixed double/float as cl.cl:3598-3604
    float g1 = (float)((double)f1 * (2 * PId / (double)1.f));
    float g2 = (float)((double)f2 * (2 * PId / (double)1.f));
    return (float)((double)(1.f * std::atan2(std::sin(g2 - g1), std::cos(g2 - g1))) / (2 * PId));
}

void shade_pixel(const render_data* rdata, int id, float* out, const image& bg1, const image& bg2, int width, int height, int maxProbes, dfg_t dfg) {
    render_data rdat = rdata[id];
    int sx = rdat.sx, sy = rdat.sy, side = rdat.side;
    float* o = out + ((size_t)sy * width + sx) * 4;
    if (rdat.terminated != 1) { o[0] = 0; o[1] = 0; o[2] = 0; o[3] = 1; return; }
    float sxf = rdat.tex_x, syf = rdat.tex_y;
    int dx = sx == width - 1 ? -1 : 1, dy = sy == height - 1 ? -1 : 1;
    const render_data& tl = rdata[sy * width + sx];
    const render_data& tr = rdata[sy * width + sx + dx];
    const render_data& bl = rdata[(sy + dy) * width + sx];
    const float bias = 1.3f;
    float dxu = circular_diff(tl.tex_x, tr.tex_x) / bias, dxv = circular_diff(tl.tex_y, tr.tex_y) / bias;
    float dyu = circular_diff(tl.tex_x, bl.tex_x) / bias, dyv = circular_diff(tl.tex_y, bl.tex_y) / bias;
    if (dx == -1) { dxu = -dxu; dxv = -dxv; }
    if (dy == -1) { dyu = -dyu; dyv = -dyv; }
    dxu *= bg1.width; dyu *= bg1.width; dxv *= bg1.height; dyv *= bg1.height;
    float dv_dx = dxv, dv_dy = dyv, du_dx = dxu, du_dy = dyu;
    float Ann = dv_dx * dv_dx + dv_dy * dv_dy + 1;
    float Bnn = -2 * (du_dx * dv_dx + du_dy * dv_dy);
    float Cnn = du_dx * du_dx + du_dy * du_dy + 1;
    float F = Ann * Cnn - Bnn * Bnn / 4;
    float A = Ann / F, B = Bnn / F, C = Cnn / F;
    float root = std::sqrt((A - C) * (A - C) + B * B);
    float a_prime = (A + C - root) / 2, c_prime = (A + C + root) / 2;
    float major = 1 / std::sqrt(a_prime), minor = 1 / std::sqrt(c_prime);
    float theta = std::atan2(B, (A - C) / 2);
    major = std::fmax(major, 1.f);
    minor = std::fmax(minor, 1.f);
    major = std::fmax(major, minor);
    float fProbes = 2 * (major / minor) - 1;
    int iProbes = (int)std::floor(fProbes + 0.5f);
    iProbes = std::min(iProbes, maxProbes);
    if (iProbes < fProbes) minor = 2 * major / (iProbes + 1);
    float lod = std::log2(minor);
    int maxLod = bg1.levels - 1;
    if (lod > maxLod) { lod = maxLod; iProbes = 1; }
    float end[4] = {0, 0, 0, 0};
    if (iProbes <= 1) {
        if (iProbes < 1) lod = maxLod;
        read_mipmap(bg1, bg2, side, sxf, syf, lod, end);
    } else {
        float line = 2 * (major - minor);
        float du = std::cos(theta) * line / (iProbes - 1), dv = std::sin(theta) * line / (iProbes - 1);
        float total[4] = {0, 0, 0, 0};
        float accumulated = 0;
        int currentN = (iProbes % 2) == 1 ? -2 * ((iProbes - 1) / 2) : -2 * (iProbes / 2) - 1;
        float sU = du / bg1.width, sV = dv / bg1.height;
        for (int cnt = 0; cnt < iProbes; cnt++) {
            float d_2 = (currentN * currentN / 4.f) * (du * du + dv * dv) / (major * major);
            float weight = std::exp(-2.f * d_2);
            float f[4];
            read_mipmap(bg1, bg2, side, sxf + (currentN / 2.f) * sU, syf + (currentN / 2.f) * sV, lod, f);
            for (int c = 0; c < 4; c++) total[c] += weight * f[c];
            accumulated += weight;
            currentN += 2;
        }
        for (int c = 0; c < 4; c++) end[c] = total[c] / accumulated;
    }
    if (GET_FEATURE(redshift, dfg)) {
        float z = rdat.z_shift;
        v3 lin{srgb_to_lin1(end[0]), srgb_to_lin1(end[1]), srgb_to_lin1(end[2])};
        const float sol = 299792458;
        float test_wavelength = 555 / sol;
        float local_wavelength = test_wavelength / (z + 1);
        float lum = 0.2126f * lin.x + 0.7152f * lin.y + 0.0722f * lin.z;
        float new_lum = std::pow(local_wavelength, 3.f) * lum / std::pow(test_wavelength, 3.f);
        new_lum = clampf(new_lum, 0.f, 1.f);
        if ((double)lum > 0.00001) lin = clamp3((new_lum / lum) * lin, 0.f, 1.f);
        lin = clamp3(redshift(lin, z, dfg), 0.f, 1.f);
#ifndef LINEAR_FRAMEBUFFER
        lin = {lin_to_srgb1(lin.x), lin_to_srgb1(lin.y), lin_to_srgb1(lin.z)};
#endif
        end[0] = lin.x; end[1] = lin.y; end[2] = lin.z;
    }
#ifdef LINEAR_FRAMEBUFFER
    if (!GET_FEATURE(redshift, dfg)) { end[0] = srgb_to_lin1(end[0]); end[1] = srgb_to_lin1(end[1]); end[2] = srgb_to_lin1(end[2]); }
#endif
    for (int c = 0; c < 4; c++) o[c] = end[c];
}

// ---- adaptive sampling (cl.cl:5102-5133, 5215-5345) ---------------------------------------------------
float angle_between(float t1, float p1, float t2, float p2) {
    v3 a = polar_to_cartesian({1.f, t1, p1}), b = polar_to_cartesian({1.f, t2, p2});
    return std::acos(clampf(dot(a, b), -1.f, 1.f));
}

render_data interpolate(render_data r1, render_data r2) {
    float a1x = (r1.tex_x - 0.5f) * (2 * PIf), a1y = r1.tex_y * PIf;
    float a2x = (r2.tex_x - 0.5f) * (2 * PIf), a2y = r2.tex_y * PIf;
    v3 p = polar_to_cartesian({1.f, a1y, a1x}), q = polar_to_cartesian({1.f, a2y, a2x});
    v3 fangle = cartesian_to_polar((p + q) / 2.f);
    render_data out;
    std::memset(&out, 0, sizeof(out));
    angle_to_tex(fangle.y, fangle.z, &out.tex_x, &out.tex_y);
    out.z_shift = (r1.z_shift + r2.z_shift) / 2.f;
    out.terminated = r1.terminated;
    out.sx = (r1.sx + r2.sx) / 2;
    out.sy = (r1.sy + r2.sy) / 2;
    out.side = (r1.side + r2.side) / 2;
    return out;
}

template <typename F>
void parallel_for(long n, int nthreads, F&& f) {
    if (nthreads <= 1 || n < 256) { for (long i = 0; i < n; i++) f(i); return; }
    std::vector<std::thread> pool;
    for (int t = 0; t < nthreads; t++)
        pool.emplace_back([&, t]() {
            for (long base = (long)t * 64; base < n; base += (long)nthreads * 64)
                for (long i = base; i < base + 64 && i < n; i++) f(i);
        });
    for (auto& th : pool) th.join();
}

}  // namespace

extern "C" {

// cart_to_generic_kernel, cl.cl:6018-6034
void ref_cart_to_generic(const float* pos_cart, float* pos_generic, float flip, const void* cfg) {
    v3 polar = cartesian_to_polar({pos_cart[1], pos_cart[2], pos_cart[3]});
    if (flip > 0) polar.x = -polar.x;
    v4 g = gen::from_spherical(mk4(pos_cart[0], polar), (cfg_t)cfg);
    pos_generic[0] = g.x; pos_generic[1] = g.y; pos_generic[2] = g.z; pos_generic[3] = g.w;
}

// init_basis_vectors, cl.cl:2483-2507
void ref_init_basis_vectors(const float* generic, const float* speed, float* e0, float* e1, float* e2, float* e3, const void* cfg) {
    v4 e[4];
    calculate_tetrads({generic[0], generic[1], generic[2], generic[3]}, {speed[0], speed[1], speed[2]}, e, (cfg_t)cfg, 1);
    float* outs[4] = {e0, e1, e2, e3};
    for (int i = 0; i < 4; i++) { outs[i][0] = e[i].x; outs[i][1] = e[i].y; outs[i][2] = e[i].z; outs[i][3] = e[i].w; }
}

// clear_termination_buffer, cl.cl:4997-5006
void ref_clear_termination_buffer(int* term, int w, int h) { for (int i = 0; i < w * h; i++) term[i] = 1; }

// init_rays_generic, cl.cl:3143-3251
void ref_init_rays_generic(const float* cam, const float* quat, void* rays_v, int* count, int w, int h, const int* term, int pw, int ph,
                           int flip, const float* e0, const float* e1, const float* e2, const float* e3, const void* cfg_v,
                           const void* dfg_v, int i_am_prepass, int nthreads) {
    lightray* rays = (lightray*)rays_v;
    cfg_t cfg = (cfg_t)cfg_v;
    dfg_t dfg = (dfg_t)dfg_v;
    v4 camera{cam[0], cam[1], cam[2], cam[3]}, q{quat[0], quat[1], quat[2], quat[3]};
    v4 e[4] = {{e0[0], e0[1], e0[2], e0[3]}, {e1[0], e1[1], e1[2], e1[3]}, {e2[0], e2[1], e2[2], e2[3]}, {e3[0], e3[1], e3[2], e3[3]}};
    bool full = i_am_prepass || !GET_FEATURE(adaptive_sampling, dfg) || GET_FEATURE(use_triangle_rendering, dfg);
    *count = full ? h * w : (h * w) / 4;
    parallel_for((long)w * h, nthreads, [&](long id) {
        int cx = (int)(id % w), cy = (int)(id / w);
        lightray ray = pixel_ray(cx, cy, w, h, camera, q, e, flip, cfg, dfg);
        if (pw != w && ph != h) {
            float fx = (float)cx / w, fy = (float)cy / h;
            int lx = (int)std::round(fx * pw), ly = (int)std::round(fy * ph);
            if (should_early_terminate(lx - 1, ly, pw, ph, term) && should_early_terminate(lx, ly, pw, ph, term) &&
                should_early_terminate(lx + 1, ly, pw, ph, term) && should_early_terminate(lx, ly - 1, pw, ph, term) &&
                should_early_terminate(lx, ly + 1, pw, ph, term))
                ray.terminated = 2;
        }
        if (full) rays[id] = ray;
        else if ((cx % 2) == 0 && (cy % 2) == 0) rays[(cy / 2) * (w / 2) + cx / 2] = ray;
    });
}

static uint64_t g_attempts = 0;
uint64_t ref_last_attempts() { return g_attempts; }

// do_generic_rays, cl.cl:3954-4247
void ref_do_generic_rays(void* rays_v, const int* count, int n_items, const void* cfg, const void* dfg, int w, int h,
                         int* ray_write_counts, int nthreads) {
    lightray* rays = (lightray*)rays_v;
    long n = std::min<long>(n_items, *count);
    std::vector<uint64_t> per_thread(nthreads > 0 ? nthreads : 1, 0);
    if (nthreads <= 1 || n < 256) {
        for (long id = 0; id < n; id++) {
            if (ray_write_counts) ray_write_counts[id] = 0;
            if (rays[id].terminated == 2) continue;
            trace_ray(&rays[id], (cfg_t)cfg, (dfg_t)dfg, &per_thread[0]);
        }
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nthreads; t++)
            pool.emplace_back([&, t]() {
                // (the attempts are counted in a local and stored once: eight-byte counters of neighbouring threads share a cache line,
                // and an increment per attempt through it made 8 threads barely faster than one - round 5, tools/cpu_calibration.py)
                uint64_t mine = 0;
                for (long base = (long)t * 64; base < n; base += (long)nthreads * 64)
                    for (long id = base; id < base + 64 && id < n; id++) {
                        if (ray_write_counts) ray_write_counts[id] = 0;
                        if (rays[id].terminated == 2) continue;
                        trace_ray(&rays[id], (cfg_t)cfg, (dfg_t)dfg, &mine);
                    }
                per_thread[t] = mine;
            });
        for (auto& th : pool) th.join();
    }
    g_attempts = 0;
    for (auto a : per_thread) g_attempts += a;
}

// Step attempts of every ray of `rays_v` (traced on copies; a ray skipped by the prepass counts 0).  Not a reference kernel: the
// parity tests use it to tell ordinary rays from the ones that linger near a photon orbit (positions are
// held to 1e-3 for rays with fewer than twice the median number of attempts).
void ref_attempts_per_ray(const void* rays_v, const int* count, int n_items, const void* cfg, const void* dfg, int* attempts_out,
                          int nthreads) {
    const lightray* rays = (const lightray*)rays_v;
    long n = std::min<long>(n_items, *count);
    if (nthreads < 1) nthreads = 1;
    std::vector<std::thread> pool;
    for (int t = 0; t < nthreads; t++)
        pool.emplace_back([&, t]() {
            for (long id = t; id < n; id += nthreads) {
                attempts_out[id] = 0;
                if (rays[id].terminated == 2) continue;
                lightray copy = rays[id];
                uint64_t a = 0;
                trace_ray(&copy, (cfg_t)cfg, (dfg_t)dfg, &a);
                attempts_out[id] = (int)a;
            }
        });
    for (auto& th : pool) th.join();
}

// Final positions (float64, 4 per ray) and termination flags of every ray of `rays_v` integrated in float64 (trace_ray_f64).
void ref_trace_f64(const void* rays_v, const int* count, int n_items, const void* cfg, const void* dfg, double* positions_out,
                   int* terminated_out, int nthreads) {
    const lightray* rays = (const lightray*)rays_v;
    long n = std::min<long>(n_items, *count);
    if (nthreads < 1) nthreads = 1;
    std::vector<std::thread> pool;
    for (int t = 0; t < nthreads; t++)
        pool.emplace_back([&, t]() {
            for (long id = t; id < n; id += nthreads) {
                for (int k = 0; k < 4; k++) positions_out[4 * id + k] = 0;
                terminated_out[id] = rays[id].terminated == 2 ? 2 : trace_ray_f64(rays[id], positions_out + 4 * id, (cfg_t)cfg, (dfg_t)dfg);
            }
        });
    for (auto& th : pool) th.join();
}

// calculate_singularities, cl.cl:5008-5020
void ref_calculate_singularities(const void* rays_v, const int* count, int n_items, int* term, int w, int h) {
    const lightray* rays = (const lightray*)rays_v;
    for (int id = 0; id < n_items && id < *count; id++) term[(id / w) * w + (id % w)] = !rays[id].terminated;
}

// calculate_render_data, cl.cl:5135-5213
void ref_calculate_render_data(const void* rays_v, const int* count, int n_items, void* rdata_v, int* rcount, int w, int h,
                               const void* cfg, const void* dfg, int nthreads) {
    const lightray* rays = (const lightray*)rays_v;
    render_data* rdata = (render_data*)rdata_v;
    long n = std::min<long>(n_items, *count);
    *rcount = w * h;
    parallel_for(n, nthreads, [&](long gid) {
        render_data d = make_render_data(rays[gid], (cfg_t)cfg, (dfg_t)dfg);
        rdata[d.sy * w + d.sx] = d;
    });
}

// handle_adaptive_sampling, cl.cl:5223-5345
void ref_handle_adaptive_sampling(const void* rays_v, const int* count, void* rdata_v, int* rcount, void* new_rays_v, int* new_count,
                                  float* cam, float* quat, const float* e0, const float* e1, const float* e2, const float* e3, int width,
                                  int height, const void* cfg_v, const void* dfg_v) {
    const lightray* rays = (const lightray*)rays_v;
    render_data* rdat = (render_data*)rdata_v;
    lightray* out = (lightray*)new_rays_v;
    cfg_t cfg = (cfg_t)cfg_v;
    dfg_t dfg = (dfg_t)dfg_v;
    v4 camera{cam[0], cam[1], cam[2], cam[3]}, q{quat[0], quat[1], quat[2], quat[3]};
    v4 e[4] = {{e0[0], e0[1], e0[2], e0[3]}, {e1[0], e1[1], e1[2], e1[3]}, {e2[0], e2[1], e2[2], e2[3]}, {e3[0], e3[1], e3[2], e3[3]}};
    int hw = width / 2, hh = height / 2;
    for (int sy = 0; sy < hh; sy++)
        for (int sx = 0; sx < hw; sx++) {
            bool should_sample = true;
            if (sx != 0 && sx != hw - 1 && sy != 0 && sy != hh - 1) {
                const lightray& centre = rays[sy * hw + sx];
                const lightray& left = rays[sy * hw + sx - 1];
                const lightray& right = rays[sy * hw + sx + 1];
                const lightray& up = rays[(sy - 1) * hw + sx];
                const lightray& down = rays[(sy + 1) * hw + sx];
                const lightray& down_right = rays[(sy + 1) * hw + sx + 1];
                v4 lpos = intersection_position(left, cfg, dfg), rpos = intersection_position(right, cfg, dfg);
                v4 upos = intersection_position(up, cfg, dfg), dpos = intersection_position(down, cfg, dfg);
                float x_error = std::fabs(angle_between(lpos.z, lpos.w, rpos.z, rpos.w));
                float y_error = std::fabs(angle_between(dpos.z, dpos.w, upos.z, upos.w));
                float relative = (float)((double)(((x_error + x_error + y_error + y_error) / 4.f) / 2) * PId);   // cl.cl:5272
                float fov = GET_FEATURE(field_of_view, dfg);
                float fov_angle = (float)((double)(fov * 2) * PId / (double)360.f);
                float per_pixel = fov_angle / width;
                should_sample = relative >= per_pixel * GET_FEATURE(adaptive_sampling_threshold, dfg);
                int ct = centre.terminated;
                if (ct != left.terminated || ct != right.terminated || ct != up.terminated || ct != down.terminated || ct != down_right.terminated)
                    should_sample = true;
            }
            if (should_sample) {
                int bx = sx * 2, by = sy * 2;
                int px[3] = {bx + 1, bx, bx + 1}, py[3] = {by, by + 1, by + 1};
                int root = *new_count;
                *new_count += 3;
                for (int i = 0; i < 3; i++) out[root + i] = pixel_ray(px[i], py[i], width, height, camera, q, e, 0, cfg, dfg);
            } else {
                int lsx = rays[sy * hw + sx].sx, lsy = rays[sy * hw + sx].sy;
                render_data c = rdat[lsy * width + lsx], r = rdat[lsy * width + lsx + 2];
                render_data d = rdat[(lsy + 2) * width + lsx], dr = rdat[(lsy + 2) * width + lsx + 2];
                rdat[lsy * width + lsx + 1] = interpolate(c, r);
                rdat[(lsy + 1) * width + lsx] = interpolate(c, d);
                rdat[(lsy + 1) * width + lsx + 1] = interpolate(c, dr);
            }
        }
}

// render, cl.cl:5453-5846
void ref_render(const void* rdata_v, const int* count, int n_items, float* out, const uint8_t* bg1, const uint8_t* bg2, int bgw, int bgh,
                int levels, int w, int h, int max_probes, const void* cfg, const void* dfg, int nthreads) {
    image b1{bgw, bgh, levels, bg1}, b2{bgw, bgh, levels, bg2};
    long n = std::min<long>(n_items, *count);
    parallel_for(n, nthreads, [&](long id) { shade_pixel((const render_data*)rdata_v, (int)id, out, b1, b2, w, h, max_probes, (dfg_t)dfg); });
}

// ---- camera on a timelike geodesic (single observer) ------------------------------------------------
// get_timelike_vector, cl.cl:1974-1990
static v4 timelike_vector(v3 speed, const v4 e[4]) {
    float Y = 1 / std::sqrt(1 - dot(speed, speed));
    return Y * e[0] + (Y * speed.x) * e[1] + (Y * speed.y) * e[2] + (Y * speed.z) * e[3];
}
static inline v4 ld4(const float* p) { return {p[0], p[1], p[2], p[3]}; }
static inline void st4(float* p, v4 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w; }

// boost_tetrad, cl.cl:2441-2481 (calculate_lorentz_boost_big :1919-1972)
void ref_boost_tetrad(float* generic, float* speed4, float* e0, float* e1, float* e2, float* e3, const void* cfg_v) {
    cfg_t cfg = (cfg_t)cfg_v;
    v4 e[4] = {ld4(e0), ld4(e1), ld4(e2), ld4(e3)};
    v4 observer = timelike_vector({speed4[0], speed4[1], speed4[2]}, e);
    float g[16];
    gen::metric_big(ld4(generic), g, cfg);
    v4 lT4 = lower_index(e[0], g), lu4 = lower_index(observer, g);
    float T[4] = {e[0].x, e[0].y, e[0].z, e[0].w}, lT[4] = {lT4.x, lT4.y, lT4.z, lT4.w};
    float uo[4] = {observer.x, observer.y, observer.z, observer.w}, luo[4] = {lu4.x, lu4.y, lu4.z, lu4.w};
    float gamma = -dot(lT4, observer);
    float L[16];
    for (int u = 0; u < 4; u++)
        for (int v = 0; v < 4; v++)
            L[u * 4 + v] = (u == v ? 1.f : 0.f) + (1 / (1 + gamma)) * (T[u] + uo[u]) * (lT[v] + luo[v]) - 2 * uo[u] * lT[v];
    auto apply = [&](v4 x) {
        float xi[4] = {x.x, x.y, x.z, x.w}, o[4];
        for (int u = 0; u < 4; u++) o[u] = L[u * 4 + 0] * xi[0] + L[u * 4 + 1] * xi[1] + L[u * 4 + 2] * xi[2] + L[u * 4 + 3] * xi[3];
        return v4{o[0], o[1], o[2], o[3]};
    };
    st4(e0, observer); st4(e1, apply(e[1])); st4(e2, apply(e[2])); st4(e3, apply(e[3]));
}

// init_inertial_ray, cl.cl:3117-3141 (geodesic_to_trace_ray :3066-3115)
void ref_init_inertial_ray(float* generic, void* rays_v, int* count, float* e0, float* e1, float* e2, float* e3, float* speed4,
                           const void* cfg_v) {
    cfg_t cfg = (cfg_t)cfg_v;
    v4 e[4] = {ld4(e0), ld4(e1), ld4(e2), ld4(e3)};
    v4 velocity = timelike_vector({speed4[0], speed4[1], speed4[2]}, e);
    lightray ray = render_ray(0, 0, ld4(generic), velocity, e[0], cfg);
    ray.ku_uobsu = 1;
    ((lightray*)rays_v)[0] = ray;
    *count = 1;
}

// circular_diff / periodic_diff, cl.cl:3598-3630
static float circular_diff_period(float f1, float f2, float period) {
    float g1 = (float)((double)f1 * (2 * PId / (double)period));
    float g2 = (float)((double)f2 * (2 * PId / (double)period));
    float d = g2 - g1;
    return (float)((double)(period * std::atan2(std::sin(d), std::cos(d))) / (2 * PId));
}
static v4 periodic_diff(v4 in1, v4 in2, v4 periods) {
    v4 ret = in1 - in2;
    if (periods.x != 0) ret.x = circular_diff_period(in2.x, in1.x, periods.x);
    if (periods.y != 0) ret.y = circular_diff_period(in2.y, in1.y, periods.y);
    if (periods.z != 0) ret.z = circular_diff_period(in2.z, in1.z, periods.z);
    if (periods.w != 0) ret.w = circular_diff_period(in2.w, in1.w, periods.w);
    return ret;
}
// get_coordinate_period, cl.cl:1338-1355
static v4 coordinate_period(cfg_t cfg) {
#ifdef HAS_COORDINATE_PERIODICITY
    v4 zero{0, 0, 0, 0};
    POSITION_VARS(zero)
    return {(float)(COORDINATE_PERIODICITY1), (float)(COORDINATE_PERIODICITY2), (float)(COORDINATE_PERIODICITY3),
            (float)(COORDINATE_PERIODICITY4)};
#else
    (void)cfg;
    return {0, 0, 0, 0};
#endif
}

// get_geodesic_path, cl.cl:4735-4940
void ref_get_geodesic_path(void* rays_v, float* positions, float* velocities, float* ds_out, int* count_in, int max_len,
                           const void* cfg_v, const void* dfg_v, int* count_out) {
    cfg_t cfg = (cfg_t)cfg_v;
    dfg_t dfg = (dfg_t)dfg_v;
    if (*count_in < 1) return;
    const lightray* ray = (const lightray*)rays_v;
    v4 position = ray->position, velocity = ray->velocity, acceleration = ray->acceleration;
    const v4 quat = ray->initial_quat;
    const float f_in_x = std::fabs(velocity.x);
#ifdef IS_CONSTANT_THETA
    position.z = PIf / 2; velocity.z = 0; acceleration.z = 0;
#endif
    float next_ds = 0.00001f;
#ifdef ADAPTIVE_PRECISION
    const float max_accel = std::fmin(0.00001000f, GET_FEATURE(max_acceleration_change, dfg));
    const float min_step = GET_FEATURE(min_step, dfg);
    (void)acceleration_to_precision(acceleration, max_accel, &next_ds);
#endif
    const float subambient = 0.5f, ambient = 0.2f;
    const float new_max = GET_FEATURE(max_precision_radius, dfg), new_min = 3;
    int bufc = 0;
    const v4 periods = coordinate_period(cfg);
    v4 last_pos_generic{0, 0, 0, 0};
    float running = 1;
    (void)quat; (void)periods; (void)last_pos_generic;
    for (int i = 0; i < max_len; i++) {
#ifdef IS_CONSTANT_THETA
        position.z = PIf / 2; velocity.z = 0; acceleration.z = 0;
#endif
        v4 polar = gen::to_spherical(position, cfg);
#ifdef IS_CONSTANT_THETA
        polar.z = PIf / 2;
#endif
        float ar = std::fabs(gen::distance_to_object(polar, cfg));
        float ds = mixf(ambient, subambient, (clampf(ar, new_min, new_max) - new_min) / (new_max - new_min));
#ifdef ADAPTIVE_PRECISION
        ds = next_ds;
#endif
        if (ar < new_max) ds = std::fmin(ds, ambient);
        else ds = (float)(0.1 * (double)(ar - new_max) + (double)ambient);   // double-typed literal at cl.cl:4815
        bool should_break = std::fabs(polar.y) >= GET_FEATURE(universe_size, dfg);
#ifdef SINGULAR
        should_break |= std::fabs(polar.y) < SINGULAR_TERMINATOR;
#endif
        v4 next_position = position + velocity * ds + 0.5f * acceleration * ds * ds;
        v4 half_velocity = velocity + acceleration * ds;
        v4 next_acceleration = gen::geo_accel(next_position, half_velocity, cfg);
        v4 next_velocity = velocity + 0.5f * (acceleration + next_acceleration) * ds;
        float K = 1;
        if (GET_FEATURE(reparameterisation, dfg)) {
            K = 1 / std::fmax(std::fmax(std::fabs(next_velocity.x), std::fabs(next_velocity.y)),
                              std::fmax(std::fabs(next_velocity.z), std::fabs(next_velocity.w)));
            next_velocity = next_velocity * K;
            next_acceleration = next_acceleration * K * K;
        }
        const float old_dlambda = running;
        running *= K;
#ifdef ADAPTIVE_PRECISION
        if (ar < new_max) {
            float suggested = 0;
            float diff = acceleration_to_precision(next_acceleration, max_accel, &suggested);
            float nds = 0.99f * ds * clampf(suggested / ds, 0.3f, 2.f);
            nds = std::fmax(nds, min_step);
            next_ds = nds;
#ifdef SINGULARITY_DETECTION
            if (nds == min_step && (diff / (256 * 256)) > max_accel * 10000) should_break = true;
#endif
            if (nds < ds / 1.95f) continue;   // the rejected step still consumes an iteration here (cl.cl:4849-4850)
        }
#endif
#ifndef UNCONDITIONALLY_NONSINGULAR
        if (std::fabs(velocity.x / running) > 1000 + f_in_x && std::fabs(acceleration.x / running) > 100) should_break = true;
#endif
        v4 pos_out = position, vel_out = velocity / old_dlambda;
#ifdef GENERIC_CONSTANT_THETA
        {   // undo the equatorial rotation, cl.cl:4864-4902
            v4 pos_sph = gen::to_spherical(position, cfg);
            v4 vel_sph = gen::velocity_to_spherical(position, velocity / old_dlambda, cfg);
            float sgn = signf(pos_sph.y);
            pos_sph.y = std::fabs(pos_sph.y);
            v3 pos_cart = rot_quat(polar_to_cartesian(yzw(pos_sph)), quat);
            v3 vel_cart = rot_quat(spherical_velocity_to_cartesian_velocity(yzw(pos_sph), yzw(vel_sph)), quat);
            v3 npos = cartesian_to_polar(pos_cart);
            v3 nvel = cartesian_velocity_to_polar_velocity(pos_cart, vel_cart);
            if (sgn < 0) npos.x = -npos.x;
            v4 next_pos_generic = gen::from_spherical(mk4(pos_sph.x, npos), cfg);
            v4 next_vel_generic = gen::velocity_from_spherical(mk4(pos_sph.x, npos), mk4(vel_sph.x, nvel), cfg);
            if (i != 0) next_pos_generic = periodic_diff(next_pos_generic, last_pos_generic, periods) + last_pos_generic;
            last_pos_generic = next_pos_generic;
            pos_out = next_pos_generic;
            vel_out = next_vel_generic;
        }
#endif
        if (degenerate(next_position) || degenerate(next_velocity) || degenerate(next_acceleration)) break;
        position = next_position;
        velocity = next_velocity;
        acceleration = next_acceleration;
        st4(positions + 4 * bufc, pos_out);
        if (velocities) st4(velocities + 4 * bufc, vel_out);
        if (ds_out) ds_out[bufc] = ds * old_dlambda;
        bufc++;
        if (should_break) break;
    }
    count_out[0] = bufc;
}

// get_geodesic_path in float64: NOT a reference kernel - the reference's discrete algorithm (same steps, controller, thresholds, the same
// float parameters and initial ray) evaluated in double precision, the yardstick for camera paths on which two fp32 builds disagree
// (a sample that lands next to a horizon, free fall at velocity 1e3, a path past the polar axis: tests/golden/paths/soak_*).  Same
// role as trace_ray_f64 for rays.  positions / velocities: 4 doubles per sample.
void ref_get_geodesic_path_f64(void* rays_v, double* positions, double* velocities, double* ds_out, int* count_in, int max_len,
                               const void* cfg_v, const void* dfg_v, int* count_out) {
    using gen64::d4;
    cfg_t cfg = (cfg_t)cfg_v;
    dfg_t dfg = (dfg_t)dfg_v;
    count_out[0] = 0;
    if (*count_in < 1) return;
    const lightray* ray = (const lightray*)rays_v;
    auto widen = [](v4 v) { return d4{v.x, v.y, v.z, v.w}; };
    d4 position = widen(ray->position), velocity = widen(ray->velocity), acceleration = widen(ray->acceleration);
    const d4 quat = widen(ray->initial_quat);
    const double f_in_x = std::fabs(velocity.x);
    const double half_pi = (double)(PIf / 2);
    (void)f_in_x; (void)quat;
#ifdef IS_CONSTANT_THETA
    position.z = half_pi; velocity.z = 0; acceleration.z = 0;
#endif
    double next_ds = (double)0.00001f;
#ifdef ADAPTIVE_PRECISION
    const double max_accel = std::fmin((double)0.00001000f, (double)GET_FEATURE(max_acceleration_change, dfg));
    const double min_step = GET_FEATURE(min_step, dfg);
    auto precision = [&](d4 acc, double* out) {
        const double divisor = (double)std::max(std::max(W_V1, W_V2), std::max(W_V3, W_V4));
        const double ax = acc.x * (double)(W_V1), ay = acc.y * (double)(W_V2), az = acc.z * (double)(W_V3), aw = acc.w * (double)(W_V4);
        const double current = std::sqrt(ax * ax + ay * ay + az * az + aw * aw) * (double)0.01f / divisor;
        const double big = 256 * 256;
        double diff = current * big;
        const double lowest = max_accel * big / 1e10;
        if (diff < lowest) diff = lowest;
        *out = std::sqrt((max_accel * big) / diff);
        return diff;
    };
    (void)precision(acceleration, &next_ds);
#endif
    const double subambient = 0.5, ambient = (double)0.2f;
    const double new_max = GET_FEATURE(max_precision_radius, dfg), new_min = 3;
    int bufc = 0;
    const v4 periods_f = coordinate_period(cfg);
    const double periods[4] = {periods_f.x, periods_f.y, periods_f.z, periods_f.w};
    d4 last_generic{0, 0, 0, 0};
    double running = 1;
    (void)periods; (void)last_generic;
    auto degenerate64 = [](d4 v) { return !std::isfinite(v.x) || !std::isfinite(v.y) || !std::isfinite(v.z) || !std::isfinite(v.w); };
    for (int i = 0; i < max_len; i++) {
#ifdef IS_CONSTANT_THETA
        position.z = half_pi; velocity.z = 0; acceleration.z = 0;
#endif
        d4 polar = gen64::to_spherical(position, cfg);
#ifdef IS_CONSTANT_THETA
        polar.z = half_pi;
#endif
        const double ar = std::fabs(gen64::distance_to_object(polar, cfg));
        double ds = ambient + (subambient - ambient) * ((std::fmin(std::fmax(ar, new_min), new_max) - new_min) / (new_max - new_min));
#ifdef ADAPTIVE_PRECISION
        ds = next_ds;
#endif
        if (ar < new_max) ds = std::fmin(ds, ambient);
        else ds = 0.1 * (ar - new_max) + ambient;
        bool should_break = std::fabs(polar.y) >= (double)GET_FEATURE(universe_size, dfg);
#ifdef SINGULAR
        should_break |= std::fabs(polar.y) < SINGULAR_TERMINATOR;
#endif
        d4 next_position = position + velocity * ds + acceleration * (0.5 * ds * ds);
        d4 half_velocity = velocity + acceleration * ds;
        d4 next_acceleration = gen64::geo_accel(next_position, half_velocity, cfg);
        d4 next_velocity = velocity + (acceleration + next_acceleration) * (0.5 * ds);
        double K = 1;
        if (GET_FEATURE(reparameterisation, dfg)) {
            K = 1 / std::fmax(std::fmax(std::fabs(next_velocity.x), std::fabs(next_velocity.y)),
                              std::fmax(std::fabs(next_velocity.z), std::fabs(next_velocity.w)));
            next_velocity = next_velocity * K;
            next_acceleration = next_acceleration * (K * K);
        }
        const double old_dlambda = running;
        running *= K;
#ifdef ADAPTIVE_PRECISION
        if (ar < new_max) {
            double suggested = 0;
            const double diff = precision(next_acceleration, &suggested);
            double nds = (double)0.99f * ds * std::fmin(std::fmax(suggested / ds, (double)0.3f), 2.0);
            nds = std::fmax(nds, min_step);
            next_ds = nds;
#ifdef SINGULARITY_DETECTION
            if (nds == min_step && (diff / (256 * 256)) > max_accel * 10000) should_break = true;
#endif
            (void)diff;
            if (nds < ds / (double)1.95f) continue;
        }
#endif
#ifndef UNCONDITIONALLY_NONSINGULAR
        if (std::fabs(velocity.x / running) > 1000 + f_in_x && std::fabs(acceleration.x / running) > 100) should_break = true;
#endif
        d4 pos_out = position, vel_out = velocity * (1 / old_dlambda);
#ifdef GENERIC_CONSTANT_THETA
        {   // the sample rotated back out of the equatorial plane, in double (cl.cl:4864-4902)
            d4 pos_sph = gen64::to_spherical(position, cfg);
            d4 vel_sph = gen64::velocity_to_spherical(position, velocity * (1 / old_dlambda), cfg);
            const double sgn = pos_sph.y > 0 ? 1.0 : (pos_sph.y < 0 ? -1.0 : 0.0);
            pos_sph.y = std::fabs(pos_sph.y);
            const double qn = std::sqrt(quat.x * quat.x + quat.y * quat.y + quat.z * quat.z + quat.w * quat.w);
            const double qx = quat.x / qn, qy = quat.y / qn, qz = quat.z / qn, qw = quat.w / qn;
            auto rotate = [&](double x, double y, double z, double out[3]) {   // rot_quat, cl.cl:176-183
                const double tx = 2 * (qy * z - qz * y), ty = 2 * (qz * x - qx * z), tz = 2 * (qx * y - qy * x);
                out[0] = x + qw * tx + (qy * tz - qz * ty);
                out[1] = y + qw * ty + (qz * tx - qx * tz);
                out[2] = z + qw * tz + (qx * ty - qy * tx);
            };
            const double r = pos_sph.y, th = pos_sph.z, ph = pos_sph.w, dr = vel_sph.y, dth = vel_sph.z, dph = vel_sph.w;
            const double cart[3] = {r * std::sin(th) * std::cos(ph), r * std::sin(th) * std::sin(ph), r * std::cos(th)};
            const double cvel[3] = {-r * std::sin(th) * std::sin(ph) * dph + r * std::cos(th) * std::cos(ph) * dth + std::sin(th) * std::cos(ph) * dr,
                                    std::sin(th) * std::sin(ph) * dr + r * std::sin(th) * std::cos(ph) * dph + r * std::cos(th) * std::sin(ph) * dth,
                                    std::cos(th) * dr - r * std::sin(th) * dth};
            double p[3], v[3];
            rotate(cart[0], cart[1], cart[2], p);
            rotate(cvel[0], cvel[1], cvel[2], v);
            const double nr = std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
            double npos[3] = {nr, std::acos(p[2] / nr), std::atan2(p[1], p[0])};
            const double repeated = nr * std::sqrt(1 - (p[2] * p[2] / (nr * nr)));
            const double rdot = (p[0] * v[0] + p[1] * v[1] + p[2] * v[2]) / nr;
            const double nvel[3] = {rdot, ((p[2] * rdot) / (nr * repeated)) - v[2] / repeated, (p[0] * v[1] - p[1] * v[0]) / (p[0] * p[0] + p[1] * p[1])};
            if (sgn < 0) npos[0] = -npos[0];
            d4 next_generic = gen64::from_spherical(d4{pos_sph.x, npos[0], npos[1], npos[2]}, cfg);
            d4 next_vel_generic = gen64::velocity_from_spherical(d4{pos_sph.x, npos[0], npos[1], npos[2]}, d4{vel_sph.x, nvel[0], nvel[1], nvel[2]}, cfg);
            if (i != 0) {
                double* g = &next_generic.x;
                const double* l = &last_generic.x;
                for (int c = 0; c < 4; c++)
                    if (periods[c] != 0) {
                        const double d = (g[c] - l[c]) * (2 * PId / periods[c]);
                        g[c] = l[c] + periods[c] * std::atan2(std::sin(d), std::cos(d)) / (2 * PId);
                    }
            }
            last_generic = next_generic;
            pos_out = next_generic;
            vel_out = next_vel_generic;
        }
#endif
        if (degenerate64(next_position) || degenerate64(next_velocity) || degenerate64(next_acceleration)) break;
        position = next_position;
        velocity = next_velocity;
        acceleration = next_acceleration;
        positions[4 * bufc + 0] = pos_out.x; positions[4 * bufc + 1] = pos_out.y; positions[4 * bufc + 2] = pos_out.z; positions[4 * bufc + 3] = pos_out.w;
        if (velocities) { velocities[4 * bufc + 0] = vel_out.x; velocities[4 * bufc + 1] = vel_out.y; velocities[4 * bufc + 2] = vel_out.z; velocities[4 * bufc + 3] = vel_out.w; }
        if (ds_out) ds_out[bufc] = ds * old_dlambda;
        bufc++;
        if (should_break) break;
    }
    count_out[0] = bufc;
}

// parallel_transport_get_velocity, cl.cl:2164-2207
static v4 parallel_transport_velocity(v4 X, v4 position, v4 velocity, cfg_t cfg) {
    float g[16], dg[64], ginv[16];
    gen::metric_big(position, g, cfg);
    gen::partials_big(position, dg, cfg);
    inverse4(g, ginv);
    float Xa[4] = {X.x, X.y, X.z, X.w}, Ya[4] = {velocity.x, velocity.y, velocity.z, velocity.w}, out[4];
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
    return {out[0], out[1], out[2], out[3]};
}

// parallel_transport_quantity, cl.cl:2569-2620
void ref_parallel_transport_quantity(float* path, float* vel, float* ds_in, float* quantity, int* count_in, float* out, const void* cfg_v) {
    cfg_t cfg = (cfg_t)cfg_v;
    int cnt = count_in[0];
    if (cnt == 0) return;
    v4 current = ld4(quantity);
    st4(out, current);
    if (cnt == 1) return;
    for (int k = 0; k < cnt - 1; k++) {
        float ds = ds_in[k];
        v4 f_x = parallel_transport_velocity(current, ld4(path + 4 * k), ld4(vel + 4 * k), cfg);
        v4 predictor = current + f_x * ds;
        v4 next = current + (0.5f * ds) * (f_x + parallel_transport_velocity(predictor, ld4(path + 4 * (k + 1)), ld4(vel + 4 * (k + 1)), cfg));
        st4(out + 4 * k, current);
        current = next;
    }
    st4(out + 4 * (cnt - 1), current);
}

// handle_interpolating_geodesic, cl.cl:2738-2872
void ref_handle_interpolating_geodesic(const float* path, const float* vel, const float* ds_in, float* cam_out, const float* te0,
                                       const float* te1, const float* te2, const float* te3, float* e0, float* e1, float* e2, float* e3,
                                       float target_time, const int* count_in, int parallel_transport_observer, const float* speed4,
                                       float* interp_vel, const void* cfg_v) {
    cfg_t cfg = (cfg_t)cfg_v;
    int cnt = *count_in;
    if (cnt == 0) return;
    v3 speed{speed4[0], speed4[1], speed4[2]};
    auto store = [&](const v4 t[4]) { st4(e0, t[0]); st4(e1, t[1]); st4(e2, t[2]); st4(e3, t[3]); };
    auto stored = [&](int i) { v4 t[4] = {ld4(te0 + 4 * i), ld4(te1 + 4 * i), ld4(te2 + 4 * i), ld4(te3 + 4 * i)}; store(t); };
    auto mix = [](v4 a, v4 b, float t) { return a + (b - a) * t; };
    if (!parallel_transport_observer) {
        v4 t[4];
        calculate_tetrads(ld4(path), speed, t, cfg, 1);
        store(t);
    } else {
        stored(0);
    }
    float proper_time = 0;
    st4(cam_out, ld4(path));
    st4(interp_vel, ld4(vel));
    if (cnt == 1) return;
    for (int i = 0; i < cnt - 1; i++) {
        float next_proper_time = proper_time + ds_in[i];
        if ((target_time >= proper_time && target_time < next_proper_time) || target_time < proper_time) {
            float dx = (target_time - proper_time) / (next_proper_time - proper_time);
            if (target_time < proper_time) dx = 0;
            v4 fin = mix(ld4(path + 4 * i), ld4(path + 4 * (i + 1)), dx);
            st4(cam_out, fin);
            v4 t[4] = {mix(ld4(te0 + 4 * i), ld4(te0 + 4 * (i + 1)), dx), mix(ld4(te1 + 4 * i), ld4(te1 + 4 * (i + 1)), dx),
                       mix(ld4(te2 + 4 * i), ld4(te2 + 4 * (i + 1)), dx), mix(ld4(te3 + 4 * i), ld4(te3 + 4 * (i + 1)), dx)};
            if (!parallel_transport_observer) calculate_tetrads(fin, speed, t, cfg, 1);
            st4(interp_vel, mix(ld4(vel + 4 * i), ld4(vel + 4 * (i + 1)), dx));
            store(t);
            return;
        }
        proper_time = next_proper_time;
    }
    st4(cam_out, ld4(path + 4 * (cnt - 1)));
    st4(interp_vel, ld4(vel + 4 * (cnt - 1)));
    if (!parallel_transport_observer) {
        v4 t[4];
        calculate_tetrads(ld4(path + 4 * (cnt - 1)), speed, t, cfg, 1);
        store(t);
    } else {
        stored(cnt - 1);
    }
}

}  // extern "C"
