// two rays per lane (packed fp32) against one ray per lane, on the substituted Kerr acceleration + a Verlet-like update
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#ifndef ITERS_SCALE
#define ITERS_SCALE 1
#endif
#include "macros.h"
typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));

namespace gm {
template <typename R> struct bits;
template <> struct bits<float> { typedef unsigned int type; };
template <> struct bits<v2f> { typedef v2u type; };
template <typename R> __device__ __forceinline__ R fma_(R a, R b, R c);
template <> __device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
template <> __device__ __forceinline__ v2f fma_(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
template <typename R> struct sc { R s, c; };
template <typename R> __device__ __forceinline__ R pick(typename bits<R>::type odd, R a, R b);
template <> __device__ __forceinline__ float pick(unsigned int odd, float a, float b) { return odd ? a : b; }
template <> __device__ __forceinline__ v2f pick(v2u odd, v2f a, v2f b) { v2f r; r.x = odd.x ? a.x : b.x; r.y = odd.y ? a.y : b.y; return r; }
template <typename R> __device__ __forceinline__ sc<R> sincos_reduced(R x) {
#pragma clang fp reassociate(off)
    typedef typename bits<R>::type U;
    R t = fma_<R>(x, R(0.636619772367581343f), R(12582912.f));
    R j = t - R(12582912.f);
    U q = __builtin_bit_cast(U, t);
    R r = fma_<R>(-j, R(1.57079637050628662109375f), x);
    r = fma_<R>(-j, R(-4.37113900018624283e-8f), r);
    R r2 = r * r;
    R sp = fma_<R>(fma_<R>(fma_<R>(R(-1.9515295891e-4f), r2, R(8.3321608736e-3f)), r2, R(-1.6666654611e-1f)), r2 * r, r);
    R cp = fma_<R>(fma_<R>(fma_<R>(R(2.443315711809948e-5f), r2, R(-1.388731625493765e-3f)), r2, R(4.166664568298827e-2f)), r2 * r2, fma_<R>(R(-0.5f), r2, R(1.0f)));
    R s = pick<R>(q & U(1), cp, sp);
    R c = pick<R>(q & U(1), sp, cp);
    s = __builtin_bit_cast(R, __builtin_bit_cast(U, s) ^ ((q & U(2)) << U(30)));
    c = __builtin_bit_cast(R, __builtin_bit_cast(U, c) ^ (((q + U(1)) & U(2)) << U(30)));
    return {s, c};
}
__device__ __forceinline__ float sin(float x) { return sincos_reduced<float>(x).s; }
__device__ __forceinline__ float cos(float x) { return sincos_reduced<float>(x).c; }
__device__ __forceinline__ v2f sin(v2f x) { return sincos_reduced<v2f>(x).s; }
__device__ __forceinline__ v2f cos(v2f x) { return sincos_reduced<v2f>(x).c; }

template <typename R> __device__ __forceinline__ void accel(R v1, R v2, R v3, R v4, R iv1, R iv2, R iv3, R iv4, R& a0, R& a1, R& a2, R& a3) {
    R TEMPORARIES0;
    a0 = GEO_ACCEL0; a1 = GEO_ACCEL1; a2 = GEO_ACCEL2; a3 = GEO_ACCEL3;
}
}

__device__ unsigned long long g_clk[4];
template <typename R> __global__ void __launch_bounds__(256) run(R* out, int iters, float ds) {
    unsigned long long c0 = clock64(), w0 = wall_clock64();
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    R p1 = R(0.f), p2 = R(4.f) + R(1e-3f * (id & 63)), p3 = R(1.2f), p4 = R(0.3f);
    R u1 = R(1.1f), u2 = R(-0.5f), u3 = R(0.05f), u4 = R(0.03f);
    R a1 = R(0.f), a2 = R(0.f), a3 = R(0.f), a4 = R(0.f);
    for (int i = 0; i < iters; i++) {
        R h = R(ds);
        R n1 = p1 + h * (u1 + R(0.5f) * h * a1), n2 = p2 + h * (u2 + R(0.5f) * h * a2), n3 = p3 + h * (u3 + R(0.5f) * h * a3), n4 = p4 + h * (u4 + R(0.5f) * h * a4);
        R w1 = u1 + h * a1, w2 = u2 + h * a2, w3 = u3 + h * a3, w4 = u4 + h * a4;
        R b1, b2, b3, b4;
        gm::accel<R>(n1, n2, n3, n4, w1, w2, w3, w4, b1, b2, b3, b4);
        u1 += R(0.5f) * h * (a1 + b1); u2 += R(0.5f) * h * (a2 + b2); u3 += R(0.5f) * h * (a3 + b3); u4 += R(0.5f) * h * (a4 + b4);
        p1 = n1; p2 = n2; p3 = n3; p4 = n4; a1 = b1; a2 = b2; a3 = b3; a4 = b4;
    }
    out[id] = p1 + p2 + p3 + p4 + u1 + u2 + u3 + u4;
    if (id == 0) { g_clk[0] = clock64() - c0; g_clk[1] = wall_clock64() - w0; }
}

static int g_lds = 0;
template <typename R> double bench(const char* name, int rays_per_lane) {
    int blocks = 256 * 8, threads = 256, iters = 4096 * ITERS_SCALE;
    R* out; hipMalloc(&out, sizeof(R) * blocks * threads);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)run<R>, hipFuncAttributeMaxDynamicSharedMemorySize, g_lds);
    run<R><<<blocks, threads, g_lds>>>(out, 16, 1e-4f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    run<R><<<blocks, threads, g_lds>>>(out, iters, 1e-4f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double attempts = double(blocks) * threads * rays_per_lane * iters;
    unsigned long long clk[4]; hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof(clk));
    printf("%-10s %8.3f ms  %8.2f G ray-steps/s   shader clock %.0f MHz (wall 100 MHz)\n", name, ms, attempts / ms * 1e-6, double(clk[0]) / double(clk[1]) * 100.0);
    R h[4]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("   check %g\n", double(((float*)h)[0]));
    hipFree(out);
    return ms;
}
int main(int argc, char** argv) { if (argc > 1) g_lds = atoi(argv[1]); printf("dynamic LDS %d bytes per workgroup\n", g_lds); for (int k = 0; k < 3; k++) { bench<float>("scalar", 1); bench<v2f>("packed2", 2); } return 0; }
