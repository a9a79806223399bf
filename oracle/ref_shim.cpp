// ref_shim.cpp — TEST INFRASTRUCTURE ONLY (never linked into or called by the product).
//
// Lets the reference's own device source (/root/reference/cl.cl, compiled as OpenCL C for x86-64 by
// oracle/build_ref.py, object kept under oracle/_ref/) run on the CPU of this container:
//   * defines the OpenCL C built-ins that object leaves undefined, under the exact Itanium-mangled
//     names OpenCL C gives them, implemented per the OpenCL 1.2 specification on top of libm
//     (native_* / fast_* are implemented exactly, i.e. at least as accurately as any GPU run);
//   * implements image2d_array_t / image2d_t / sampler_t as plain structs
//     (read_imagef: normalized coords, CLK_ADDRESS_REPEAT, CLK_FILTER_LINEAR, spec section 8.2);
//   * exports ref_* drivers that set the work-item ids and call each kernel entry point once per
//     work-item (kernel signatures: cl.cl:6019, 2484-2487, 4998, 3144-3152, 3955-3964, 5009,
//     5136-5139, 5224-5230, 5454-5457).
// No reference source text is reproduced here; the kernels are linked from the object file.
//
// Must be compiled with the same clang that compiled cl.cl (ext_vector_type ABI).
#include <stddef.h>
#include <stdint.h>

#include <thread>
#include <vector>

extern "C" {
float sinf(float); float cosf(float); float tanf(float); float asinf(float); float acosf(float); float atanf(float);
float atan2f(float, float); float expf(float); float exp2f(float); float logf(float); float log2f(float); float log10f(float);
float powf(float, float); float sqrtf(float); float tanhf(float); float sinhf(float); float coshf(float); float fabsf(float);
float floorf(float); float ceilf(float); float roundf(float); float truncf(float); float fmodf(float, float);
float copysignf(float, float); float rintf(float); float fminf(float, float); float fmaxf(float, float);
double pow(double, double); double sqrt(double); double sin(double); double cos(double); double atan2(double, double);
double fabs(double); double floor(double);
}

typedef float float2 __attribute__((ext_vector_type(2)));
typedef float float3 __attribute__((ext_vector_type(3)));
typedef float float4 __attribute__((ext_vector_type(4)));
typedef int int2 __attribute__((ext_vector_type(2)));
typedef int int4 __attribute__((ext_vector_type(4)));

#define OCL(name) __asm__(name)

// ---- work-item functions -------------------------------------------------------------------------
static thread_local size_t t_gid[3], t_lid[3], t_grp[3], t_lsz[3] = {1, 1, 1}, t_ngrp[3] = {1, 1, 1};
size_t ocl_get_global_id(unsigned d) OCL("_Z13get_global_idj");
size_t ocl_get_global_id(unsigned d) { return d < 3 ? t_gid[d] : 0; }
size_t ocl_get_local_id(unsigned d) OCL("_Z12get_local_idj");
size_t ocl_get_local_id(unsigned d) { return d < 3 ? t_lid[d] : 0; }
size_t ocl_get_group_id(unsigned d) OCL("_Z12get_group_idj");
size_t ocl_get_group_id(unsigned d) { return d < 3 ? t_grp[d] : 0; }
size_t ocl_get_local_size(unsigned d) OCL("_Z14get_local_sizej");
size_t ocl_get_local_size(unsigned d) { return d < 3 ? t_lsz[d] : 1; }
size_t ocl_get_num_groups(unsigned d) OCL("_Z14get_num_groupsj");
size_t ocl_get_num_groups(unsigned d) { return d < 3 ? t_ngrp[d] : 1; }
void ocl_barrier(unsigned) OCL("_Z7barrierj");
void ocl_barrier(unsigned) {}

int ocl_atomic_add(volatile int* p, int v) OCL("_Z10atomic_addPU8CLglobalVii");
int ocl_atomic_add(volatile int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
int ocl_atomic_inc(volatile int* p) OCL("_Z10atomic_incPU8CLglobalVi");
int ocl_atomic_inc(volatile int* p) { return __atomic_fetch_add(p, 1, __ATOMIC_SEQ_CST); }
int ocl_atomic_min(volatile int* p, int v) OCL("_Z10atomic_minPU8CLglobalVii");
int ocl_atomic_min(volatile int* p, int v) {
    int old = *p;
    while (v < old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return old;
}
int ocl_atomic_max(volatile int* p, int v) OCL("_Z10atomic_maxPU8CLglobalVii");
int ocl_atomic_max(volatile int* p, int v) {
    int old = *p;
    while (v > old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return old;
}

// ---- scalar math ---------------------------------------------------------------------------------
#define UNARY(ocl, mangled, impl) float ocl(float x) OCL(mangled); float ocl(float x) { return impl(x); }
UNARY(o_sin, "_Z3sinf", sinf) UNARY(o_cos, "_Z3cosf", cosf) UNARY(o_tan, "_Z3tanf", tanf)
UNARY(o_asin, "_Z4asinf", asinf) UNARY(o_acos, "_Z4acosf", acosf) UNARY(o_atan, "_Z4atanf", atanf)
UNARY(o_exp, "_Z3expf", expf) UNARY(o_exp2, "_Z4exp2f", exp2f) UNARY(o_log, "_Z3logf", logf)
UNARY(o_log2, "_Z4log2f", log2f) UNARY(o_log10, "_Z5log10f", log10f) UNARY(o_sqrt, "_Z4sqrtf", sqrtf)
UNARY(o_tanh, "_Z4tanhf", tanhf) UNARY(o_sinh, "_Z4sinhf", sinhf) UNARY(o_cosh, "_Z4coshf", coshf)
UNARY(o_fabs, "_Z4fabsf", fabsf) UNARY(o_floor, "_Z5floorf", floorf) UNARY(o_ceil, "_Z4ceilf", ceilf)
UNARY(o_round, "_Z5roundf", roundf) UNARY(o_trunc, "_Z5truncf", truncf)
UNARY(o_nsin, "_Z10native_sinf", sinf) UNARY(o_ncos, "_Z10native_cosf", cosf) UNARY(o_nexp, "_Z10native_expf", expf)
UNARY(o_nsqrt, "_Z11native_sqrtf", sqrtf)
static inline float rsqrt_impl(float x) { return 1.0f / sqrtf(x); }
UNARY(o_nrsqrt, "_Z12native_rsqrtf", rsqrt_impl)
static inline float sign_impl(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }   // sign(NaN) = 0, sign(0) = 0
UNARY(o_sign, "_Z4signf", sign_impl)

#define BINARY(ocl, mangled, impl) float ocl(float x, float y) OCL(mangled); float ocl(float x, float y) { return impl(x, y); }
BINARY(o_atan2, "_Z5atan2ff", atan2f) BINARY(o_pow, "_Z3powff", powf) BINARY(o_fmod, "_Z4fmodff", fmodf)
BINARY(o_copysign, "_Z8copysignff", copysignf) BINARY(o_fmin, "_Z4fminff", fminf) BINARY(o_fmax, "_Z4fmaxff", fmaxf)
static inline float min_impl(float x, float y) { return y < x ? y : x; }
static inline float max_impl(float x, float y) { return x < y ? y : x; }
BINARY(o_min, "_Z3minff", min_impl) BINARY(o_max, "_Z3maxff", max_impl)
double o_powd(double x, double y) OCL("_Z3powdd");
double o_powd(double x, double y) { return pow(x, y); }
int o_mini(int x, int y) OCL("_Z3minii");
int o_mini(int x, int y) { return y < x ? y : x; }
int o_maxi(int x, int y) OCL("_Z3maxii");
int o_maxi(int x, int y) { return x < y ? y : x; }
float o_clamp(float x, float lo, float hi) OCL("_Z5clampfff");
float o_clamp(float x, float lo, float hi) { return min_impl(max_impl(x, lo), hi); }
int o_clampi(int x, int lo, int hi) OCL("_Z5clampiii");
int o_clampi(int x, int lo, int hi) { return o_mini(o_maxi(x, lo), hi); }
float o_mix(float a, float b, float t) OCL("_Z3mixfff");
float o_mix(float a, float b, float t) { return a + (b - a) * t; }
int o_isnan(float x) OCL("_Z5isnanf");
int o_isnan(float x) { return x != x; }
int o_isfinite(float x) OCL("_Z8isfinitef");
int o_isfinite(float x) { return fabsf(x) <= 3.402823466e+38f; }

// ---- vector math ---------------------------------------------------------------------------------
float o_dot2(float2 a, float2 b) OCL("_Z3dotDv2_fS_");
float o_dot2(float2 a, float2 b) { return a.x * b.x + a.y * b.y; }
float o_dot3(float3 a, float3 b) OCL("_Z3dotDv3_fS_");
float o_dot3(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
float o_dot4(float4 a, float4 b) OCL("_Z3dotDv4_fS_");
float o_dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
float3 o_cross(float3 a, float3 b) OCL("_Z5crossDv3_fS_");
float3 o_cross(float3 a, float3 b) { return (float3){a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
float o_length3(float3 a) OCL("_Z6lengthDv3_f");
float o_length3(float3 a) { return sqrtf(o_dot3(a, a)); }
float o_flength3(float3 a) OCL("_Z11fast_lengthDv3_f");
float o_flength3(float3 a) { return sqrtf(o_dot3(a, a)); }
float o_flength4(float4 a) OCL("_Z11fast_lengthDv4_f");
float o_flength4(float4 a) { return sqrtf(o_dot4(a, a)); }
float3 o_normalize3(float3 a) OCL("_Z9normalizeDv3_f");
float3 o_normalize3(float3 a) { return a / o_length3(a); }
float2 o_fnormalize2(float2 a) OCL("_Z14fast_normalizeDv2_f");
float2 o_fnormalize2(float2 a) { return a / sqrtf(o_dot2(a, a)); }
float3 o_fnormalize3(float3 a) OCL("_Z14fast_normalizeDv3_f");
float3 o_fnormalize3(float3 a) { return a / o_length3(a); }
float4 o_fnormalize4(float4 a) OCL("_Z14fast_normalizeDv4_f");
float4 o_fnormalize4(float4 a) { return a / sqrtf(o_dot4(a, a)); }
float4 o_min4(float4 a, float4 b) OCL("_Z3minDv4_fS_");
float4 o_min4(float4 a, float4 b) { return (float4){min_impl(a.x, b.x), min_impl(a.y, b.y), min_impl(a.z, b.z), min_impl(a.w, b.w)}; }
float4 o_max4(float4 a, float4 b) OCL("_Z3maxDv4_fS_");
float4 o_max4(float4 a, float4 b) { return (float4){max_impl(a.x, b.x), max_impl(a.y, b.y), max_impl(a.z, b.z), max_impl(a.w, b.w)}; }
float3 o_mix3(float3 a, float3 b, float t) OCL("_Z3mixDv3_fS_f");
float3 o_mix3(float3 a, float3 b, float t) { return a + (b - a) * t; }
float4 o_mix4(float4 a, float4 b, float t) OCL("_Z3mixDv4_fS_f");
float4 o_mix4(float4 a, float4 b, float t) { return a + (b - a) * t; }
float2 o_fabs2(float2 a) OCL("_Z4fabsDv2_f");
float2 o_fabs2(float2 a) { return (float2){fabsf(a.x), fabsf(a.y)}; }
float3 o_fabs3(float3 a) OCL("_Z4fabsDv3_f");
float3 o_fabs3(float3 a) { return (float3){fabsf(a.x), fabsf(a.y), fabsf(a.z)}; }
float2 o_fmod2(float2 a, float2 b) OCL("_Z4fmodDv2_fS_");
float2 o_fmod2(float2 a, float2 b) { return (float2){fmodf(a.x, b.x), fmodf(a.y, b.y)}; }
float3 o_clamp3(float3 a, float lo, float hi) OCL("_Z5clampDv3_fff");
float3 o_clamp3(float3 a, float lo, float hi) { return (float3){o_clamp(a.x, lo, hi), o_clamp(a.y, lo, hi), o_clamp(a.z, lo, hi)}; }
// vector relationals return -1 / 0 per lane; any/all test the sign bit
int4 o_isnan4(float4 a) OCL("_Z5isnanDv4_f");
int4 o_isnan4(float4 a) { return (int4){-(a.x != a.x), -(a.y != a.y), -(a.z != a.z), -(a.w != a.w)}; }
int4 o_isfinite4(float4 a) OCL("_Z8isfiniteDv4_f");
int4 o_isfinite4(float4 a) { return (int4){-o_isfinite(a.x), -o_isfinite(a.y), -o_isfinite(a.z), -o_isfinite(a.w)}; }
int o_all4(int4 a) OCL("_Z3allDv4_i");
int o_all4(int4 a) { return (a.x < 0) && (a.y < 0) && (a.z < 0) && (a.w < 0); }
int o_any4(int4 a) OCL("_Z3anyDv4_i");
int o_any4(int4 a) { return (a.x < 0) || (a.y < 0) || (a.z < 0) || (a.w < 0); }

// ---- images --------------------------------------------------------------------------------------
struct ref_image {
    int width, height, levels;
    const uint8_t* texels;   // read-only array image: [levels][height][width][4] UNORM8
    float* out;              // write-only 2d image: [height][width][4] float
};

void* o_sampler(int v) OCL("__translate_sampler_initializer");
void* o_sampler(int v) { return (void*)(intptr_t)v; }

static inline float4 texel(const ref_image* im, int x, int y, int layer) {
    const uint8_t* p = im->texels + (((size_t)layer * im->height + y) * im->width + x) * 4;
    return (float4){p[0] / 255.0f, p[1] / 255.0f, p[2] / 255.0f, p[3] / 255.0f};
}

float4 o_read_imagef(ref_image* im, void* sampler, float4 c) OCL("_Z11read_imagef20ocl_image2d_array_ro11ocl_samplerDv4_f");
float4 o_read_imagef(ref_image* im, void*, float4 c) {
    int layer = (int)rintf(c.z);
    layer = layer < 0 ? 0 : (layer > im->levels - 1 ? im->levels - 1 : layer);
    float u = (c.x - floorf(c.x)) * im->width;
    float v = (c.y - floorf(c.y)) * im->height;
    int i0 = (int)floorf(u - 0.5f), j0 = (int)floorf(v - 0.5f);
    int i1 = i0 + 1, j1 = j0 + 1;
    if (i0 < 0) i0 += im->width;
    if (i1 > im->width - 1) i1 -= im->width;
    if (j0 < 0) j0 += im->height;
    if (j1 > im->height - 1) j1 -= im->height;
    // an image read is memory-safe whatever the coordinates (the spec guarantees that much and leaves the value open): a NaN or infinite
    // coordinate - a ray that ended in a coordinate singularity with terminated = 1 - converts to INT_MIN on x86; any texel will do, the
    // weights below are NaN then and so is the result (tests/fuzz_refscripts.py: kerr_rational_polynomial)
    i0 = i0 < 0 ? 0 : (i0 > im->width - 1 ? im->width - 1 : i0);
    i1 = i1 < 0 ? 0 : (i1 > im->width - 1 ? im->width - 1 : i1);
    j0 = j0 < 0 ? 0 : (j0 > im->height - 1 ? im->height - 1 : j0);
    j1 = j1 < 0 ? 0 : (j1 > im->height - 1 ? im->height - 1 : j1);
    float a = (u - 0.5f) - floorf(u - 0.5f);
    float b = (v - 0.5f) - floorf(v - 0.5f);
    return (1 - a) * (1 - b) * texel(im, i0, j0, layer) + a * (1 - b) * texel(im, i1, j0, layer) +
           (1 - a) * b * texel(im, i0, j1, layer) + a * b * texel(im, i1, j1, layer);
}

void o_write_imagef(ref_image* im, int2 p, float4 v) OCL("_Z12write_imagef14ocl_image2d_woDv2_iDv4_f");
void o_write_imagef(ref_image* im, int2 p, float4 v) {
    if (p.x < 0 || p.y < 0 || p.x >= im->width || p.y >= im->height) return;
    float* o = im->out + ((size_t)p.y * im->width + p.x) * 4;
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
int o_img_w_wo(ref_image* im) OCL("_Z15get_image_width14ocl_image2d_wo");
int o_img_w_wo(ref_image* im) { return im->width; }
int o_img_h_wo(ref_image* im) OCL("_Z16get_image_height14ocl_image2d_wo");
int o_img_h_wo(ref_image* im) { return im->height; }
int o_img_w_arr(ref_image* im) OCL("_Z15get_image_width20ocl_image2d_array_ro");
int o_img_w_arr(ref_image* im) { return im->width; }
int o_img_h_arr(ref_image* im) OCL("_Z16get_image_height20ocl_image2d_array_ro");
int o_img_h_arr(ref_image* im) { return im->height; }
int o_img_layers(ref_image* im) OCL("_Z20get_image_array_size20ocl_image2d_array_ro");
int o_img_layers(ref_image* im) { return im->levels; }

// ---- kernel entry points of the reference object ---------------------------------------------------
extern "C" {
void cart_to_generic_kernel(const float4*, float4*, int, float, const void*);
void init_basis_vectors(const float4*, int, float3, float4*, float4*, float4*, float4*, const void*);
void clear_termination_buffer(int*, int, int);
void init_rays_generic(const float4*, const float4*, void*, int*, int, int, const int*, int, int, int, const float4*,
                       const float4*, const float4*, const float4*, const void*, const void*, int);
void do_generic_rays(void*, const int*, int*, int*, const void*, const void*, int, int, int, int, float4*, int*, int);
void calculate_singularities(const void*, const int*, int*, int, int);
void calculate_render_data(const void*, const int*, void*, int*, int, int, const void*, const void*);
void handle_adaptive_sampling(const void*, const int*, void*, int*, void*, int*, float4*, float4*, const float4*, const float4*,
                              const float4*, const float4*, int, int, const void*, const void*);
void render(const void*, const int*, ref_image*, ref_image*, ref_image*, int, int, int, const void*, const void*);
void boost_tetrad(float4*, int, float3*, float4*, float4*, float4*, float4*, const void*);
void init_inertial_ray(float4*, int, void*, int*, float4*, float4*, float4*, float4*, float3*, const void*);
void get_geodesic_path(void*, float4*, float4*, float*, int*, int, const void*, const void*, int*);
void parallel_transport_quantity(float4*, float4*, float*, float4*, int*, int, float4*, const void*);
void handle_interpolating_geodesic(const float4*, const float4*, const float*, float4*, const float4*, const float4*, const float4*,
                                   const float4*, float4*, float4*, float4*, float4*, float, const int*, int, const float3*, float4*,
                                   const void*);
}

template <typename F>
static void run_items(long n, int nthreads, F&& f) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads == 1 || n < 256) {
        for (long i = 0; i < n; i++) { t_gid[0] = (size_t)i; t_gid[1] = 0; f(); }
        return;
    }
    std::vector<std::thread> pool;
    for (int t = 0; t < nthreads; t++) {
        pool.emplace_back([&, t]() {
            // interleaved blocks of 64 items keep the per-thread load even
            for (long base = (long)t * 64; base < n; base += (long)nthreads * 64)
                for (long i = base; i < base + 64 && i < n; i++) { t_gid[0] = (size_t)i; t_gid[1] = 0; f(); }
        });
    }
    for (auto& th : pool) th.join();
}

extern "C" {

void ref_cart_to_generic(const float* pos_cart, float* pos_generic, float flip, const void* cfg) {
    t_gid[0] = 0;
    cart_to_generic_kernel((const float4*)pos_cart, (float4*)pos_generic, 1, flip, cfg);
}

void ref_init_basis_vectors(const float* generic, const float* speed, float* e0, float* e1, float* e2, float* e3, const void* cfg) {
    t_gid[0] = 0;
    float3 s = (float3){speed[0], speed[1], speed[2]};
    init_basis_vectors((const float4*)generic, 1, s, (float4*)e0, (float4*)e1, (float4*)e2, (float4*)e3, cfg);
}

void ref_clear_termination_buffer(int* term, int w, int h) {
    run_items((long)w * h, 1, [&]() { clear_termination_buffer(term, w, h); });
}

void ref_init_rays_generic(const float* cam, const float* quat, void* rays, int* count, int w, int h, const int* term, int pw,
                           int ph, int flip, const float* e0, const float* e1, const float* e2, const float* e3,
                           const void* cfg, const void* dfg, int i_am_prepass, int nthreads) {
    run_items((long)w * h, nthreads, [&]() {
        init_rays_generic((const float4*)cam, (const float4*)quat, rays, count, w, h, term, pw, ph, flip, (const float4*)e0,
                          (const float4*)e1, (const float4*)e2, (const float4*)e3, cfg, dfg, i_am_prepass);
    });
}

void ref_do_generic_rays(void* rays, const int* count, int n_items, const void* cfg, const void* dfg, int w, int h,
                         int* ray_write_counts, int nthreads) {
    int tmin = 0, tmax = 0;
    run_items(n_items, nthreads, [&]() { do_generic_rays(rays, count, &tmin, &tmax, cfg, dfg, w, h, 0, 0, nullptr, ray_write_counts, 0); });
}

void ref_calculate_singularities(const void* rays, const int* count, int n_items, int* term, int w, int h) {
    run_items(n_items, 1, [&]() { calculate_singularities(rays, count, term, w, h); });
}

void ref_calculate_render_data(const void* rays, const int* count, int n_items, void* rdata, int* rcount, int w, int h,
                               const void* cfg, const void* dfg, int nthreads) {
    run_items(n_items, nthreads, [&]() { calculate_render_data(rays, count, rdata, rcount, w, h, cfg, dfg); });
}

void ref_handle_adaptive_sampling(const void* rays, const int* count, void* rdata, int* rcount, void* new_rays, int* new_count,
                                  float* cam, float* quat, const float* e0, const float* e1, const float* e2, const float* e3,
                                  int w, int h, const void* cfg, const void* dfg) {
    for (int y = 0; y < h / 2; y++)
        for (int x = 0; x < w / 2; x++) {
            t_gid[0] = (size_t)x;
            t_gid[1] = (size_t)y;
            handle_adaptive_sampling(rays, count, rdata, rcount, new_rays, new_count, (float4*)cam, (float4*)quat, (const float4*)e0,
                                     (const float4*)e1, (const float4*)e2, (const float4*)e3, w, h, cfg, dfg);
        }
}

void ref_render(const void* rdata, const int* count, int n_items, float* out, const uint8_t* bg1, const uint8_t* bg2, int bgw,
                int bgh, int levels, int w, int h, int max_probes, const void* cfg, const void* dfg, int nthreads) {
    ref_image o{w, h, 1, nullptr, out};
    ref_image b1{bgw, bgh, levels, bg1, nullptr};
    ref_image b2{bgw, bgh, levels, bg2, nullptr};
    run_items(n_items, nthreads, [&]() { render(rdata, count, &o, &b1, &b2, w, h, max_probes, cfg, dfg); });
}

// camera on a timelike geodesic: single observer (count = 1), cl.cl:2441-2481, 3117-3141, 4735-4940, 2569-2620, 2738-2872
void ref_boost_tetrad(float* generic, float* speed4, float* e0, float* e1, float* e2, float* e3, const void* cfg) {
    t_gid[0] = 0;
    boost_tetrad((float4*)generic, 1, (float3*)speed4, (float4*)e0, (float4*)e1, (float4*)e2, (float4*)e3, cfg);
}
void ref_init_inertial_ray(float* generic, void* rays, int* count, float* e0, float* e1, float* e2, float* e3, float* speed4, const void* cfg) {
    t_gid[0] = 0;
    init_inertial_ray((float4*)generic, 1, rays, count, (float4*)e0, (float4*)e1, (float4*)e2, (float4*)e3, (float3*)speed4, cfg);
}
void ref_get_geodesic_path(void* rays, float* positions, float* velocities, float* ds, int* count_in, int max_len, const void* cfg,
                           const void* dfg, int* count_out) {
    t_gid[0] = 0;
    get_geodesic_path(rays, (float4*)positions, (float4*)velocities, ds, count_in, max_len, cfg, dfg, count_out);
}
void ref_parallel_transport_quantity(float* path, float* vel, float* ds, float* quantity, int* count_in, float* out, const void* cfg) {
    t_gid[0] = 0;
    parallel_transport_quantity((float4*)path, (float4*)vel, ds, (float4*)quantity, count_in, 1, (float4*)out, cfg);
}
void ref_handle_interpolating_geodesic(const float* path, const float* vel, const float* ds, float* cam_out, const float* te0,
                                       const float* te1, const float* te2, const float* te3, float* e0, float* e1, float* e2, float* e3,
                                       float target_time, const int* count_in, int parallel_transport_observer, const float* speed4,
                                       float* interp_vel, const void* cfg) {
    t_gid[0] = 0;
    handle_interpolating_geodesic((const float4*)path, (const float4*)vel, ds, (float4*)cam_out, (const float4*)te0, (const float4*)te1,
                                  (const float4*)te2, (const float4*)te3, (float4*)e0, (float4*)e1, (float4*)e2, (float4*)e3, target_time,
                                  count_in, parallel_transport_observer, (const float3*)speed4, (float4*)interp_vel, cfg);
}

}  // extern "C"
