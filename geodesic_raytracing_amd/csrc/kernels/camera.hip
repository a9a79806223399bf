// ------------------------------------------------------------------------------------------------
// camera.hip - what runs ONCE PER FRAME on one lane: the camera's metric coordinates and its tetrad (and, behind this part,
// geodesic_camera.hip: the camera riding a timelike geodesic).  These kernels are built into a code object of their own - the
// "set-up module" (capi.cpp: program, probes, metric, setup, camera, geodesic_camera) - with IEEE arithmetic: correctly rounded
// divide and square root, library sin / cos / acos / atan2, no reciprocal or approximate-function arithmetic, no re-association,
// no contraction.  The per-ray kernels keep OpenCL's relaxed arithmetic (-cl-unsafe-math-optimizations, metric_manager.hpp:70),
// which is what the Verlet loop needs; here it costs nothing and it is what makes a frame agree with the reference when the
// camera sits next to an axis of its chart: cartesian_velocity_to_polar_velocity forms r sqrt(1 - z^2 / r^2), which a camera 0.9
// degrees off the polar axis evaluates as 1 - 0.99975 - half an ulp of 1 is 1.2e-4 of the result, v_rcp_f32's one-to-two ulp
// several times that - and the whole view turned by 1e-5 rad: masked pixel RMSE 1.2e-4 for flat space (soak 44/171,
// tests/golden/soak/), every later stage agreeing with the reference to 3e-6 when fed the reference's tetrad.

#ifdef ADAPTIVE_PRECISION
#define GR_W_MAX ((float)((W_V1 > W_V2 ? W_V1 : W_V2) > (W_V3 > W_V4 ? W_V3 : W_V4) ? (W_V1 > W_V2 ? W_V1 : W_V2) : (W_V3 > W_V4 ? W_V3 : W_V4)))
// acceleration_to_precision (cl.cl:3400-3429) as written there (the ray kernels' version lives in integrator.hip)
__device__ __forceinline__ float acceleration_to_precision(float4 acc, float max_acceleration, float& next_ds) {
    float4 wa = f4(acc.x * (float)(W_V1), acc.y * (float)(W_V2), acc.z * (float)(W_V3), acc.w * (float)(W_V4));
    float current = __builtin_sqrtf(dot4(wa, wa)) * 0.01f;
    current /= GR_W_MAX;
    const float scale = 65536.f;
    float err = max_acceleration;
    float diff = current * scale;
    float floor_diff = err * scale / 1e10f;
    if (diff < floor_diff) diff = floor_diff;
    next_ds = __builtin_sqrtf((err * scale) / diff);
    return diff;
}
#endif

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

// gr_cart_to_generic + gr_init_basis_vectors for one camera in one launch (the frame driver's form: gr_camera_prepass launches this
// and then the prepass grid)
extern "C" __global__ void gr_camera_setup(const float4* __restrict__ position_cart_in, float flip, float speed_x, float speed_y, float speed_z,
                                           float4* __restrict__ position_generic_out, float4* __restrict__ e0_out, float4* __restrict__ e1_out,
                                           float4* __restrict__ e2_out, float4* __restrict__ e3_out, cfg_t cfg) {
    if (blockIdx.x * blockDim.x + threadIdx.x != 0) return;
    const float4 in = *position_cart_in;
    float3 polar = cartesian_to_polar(yzw(in));
    if (flip > 0) polar.x = -polar.x;
    const float4 camera = gm::spherical_to_generic(f4(in.x, polar), cfg);
    tetrad t;
    calculate_tetrads(camera, f3(speed_x, speed_y, speed_z), t, cfg, 1);
    *position_generic_out = camera;
    *e0_out = t.e[0];
    *e1_out = t.e[1];
    *e2_out = t.e[2];
    *e3_out = t.e[3];
}
