// reciprocal_refinement.hip - VERDICT r05 item 9: would ONE Newton step on the Verlet loop's v_rcp_f32 / v_sqrt_f32 / v_rsq_f32 give the
// quotients and roots the reference's x86 build computes (correctly rounded IEEE division and square root)?  Counts, over 2^26 random
// operands per function, how often each form differs from the correctly rounded result, and by how many ulps.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/reciprocal_refinement.hip -o tools/ubench/reciprocal_refinement   (build container)
//   tools/ubench/reciprocal_refinement                                                                             (GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>

__device__ __forceinline__ uint32_t pcg(uint32_t& s) { s = s * 747796405u + 2891336453u; uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u; return (w >> 22u) ^ w; }
__device__ __forceinline__ float operand(uint32_t& s) {   // magnitudes 2^-8 .. 2^8, random mantissa, random sign: what the loop's denominators look like
    const uint32_t m = pcg(s) & 0x7fffffu, e = 119u + (pcg(s) % 17u), sg = pcg(s) & 0x80000000u;
    return __uint_as_float(sg | (e << 23) | m);
}
__device__ __forceinline__ int ulps(float a, float b) {
    int ia = __float_as_int(a), ib = __float_as_int(b);
    if (ia < 0) ia = 0x80000000 - ia;
    if (ib < 0) ib = 0x80000000 - ib;
    int d = ia - ib;
    return d < 0 ? -d : d;
}

// counts[f][0..3]: differs from the correctly rounded result by 0, 1, 2, >= 3 ulps, for f = 0 rcp, 1 rcp + Newton, 2 sqrt, 3 sqrt + Newton, 4 rsq, 5 rsq + Newton,
// 6 a * rcp(b) (the loop's division), 7 a * refined rcp(b), 8 a * refined rcp(b) + one residual correction (fma(fma(-b, q, a), r, q))
__global__ void count(unsigned long long* counts, int per_thread) {
    uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    unsigned int local[9][4] = {};
    for (int i = 0; i < per_thread; i++) {
        const float x = operand(s), a = operand(s);
        const float ax = fabsf(x);
        // correctly rounded references (IEEE): __frcp_rn / __fsqrt_rn / __fdiv_rn; 1/sqrt has no single IEEE operation: double, rounded once
        const float rcp_rn = __frcp_rn(x), sqrt_rn = __fsqrt_rn(ax), rsq_rn = (float)(1.0 / sqrt((double)ax)), div_rn = __fdiv_rn(a, x);
        const float r0 = __builtin_amdgcn_rcpf(x);
        const float r1 = fmaf(fmaf(-x, r0, 1.0f), r0, r0);                       // Newton: r + r (1 - x r)
        const float q0 = __builtin_amdgcn_sqrtf(ax);
        const float h = 0.5f * __builtin_amdgcn_rcpf(q0);
        const float q1 = fmaf(fmaf(-q0, q0, ax), h, q0);                          // q + (x - q^2) / (2 q)
        const float s0 = __builtin_amdgcn_rsqf(ax);
        const float s1 = fmaf(fmaf(-ax * s0, s0, 1.0f), 0.5f * s0, s0);           // s + s (1 - x s^2) / 2
        const float d0 = a * r0, d1 = a * r1;
        const float d2 = fmaf(fmaf(-x, d1, a), r1, d1);
        const float got[9] = {r0, r1, q0, q1, s0, s1, d0, d1, d2};
        const float want[9] = {rcp_rn, rcp_rn, sqrt_rn, sqrt_rn, rsq_rn, rsq_rn, div_rn, div_rn, div_rn};
        for (int f = 0; f < 9; f++) { int u = ulps(got[f], want[f]); local[f][u > 3 ? 3 : u]++; }
    }
    for (int f = 0; f < 9; f++) for (int k = 0; k < 4; k++) atomicAdd(&counts[f * 4 + k], (unsigned long long)local[f][k]);
}

int main() {
    unsigned long long* d; unsigned long long h[36] = {};
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) { std::printf("no device\n"); return 1; }
    hipMemset(d, 0, sizeof(h));
    const int blocks = 1024, threads = 256, per = 256;   // 2^26 operand pairs
    hipLaunchKernelGGL(count, dim3(blocks), dim3(threads), 0, 0, d, per);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[9] = {"v_rcp_f32(x)", "v_rcp_f32 + one Newton step", "v_sqrt_f32(x)", "v_sqrt_f32 + one Newton step", "v_rsq_f32(x)", "v_rsq_f32 + one Newton step",
                            "a * v_rcp_f32(b)  [the loop's quotient]", "a * refined rcp(b)", "a * refined rcp(b), residual-corrected"};
    const double n = (double)blocks * threads * per;
    std::printf("# %.0f random operands per form, magnitudes 2^-8..2^8; against the correctly rounded IEEE result (division, square root; 1/sqrt: double rounded once)\n", n);
    std::printf("%-44s %10s %10s %10s %10s\n", "form", "exact", "1 ulp off", "2 ulp", ">= 3 ulp");
    for (int f = 0; f < 9; f++) std::printf("%-44s %9.4f%% %9.4f%% %9.4f%% %9.4f%%\n", names[f], 100 * h[f * 4] / n, 100 * h[f * 4 + 1] / n, 100 * h[f * 4 + 2] / n, 100 * h[f * 4 + 3] / n);
    return 0;
}
