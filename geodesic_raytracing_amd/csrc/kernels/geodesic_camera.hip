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
            // The rotation turns angles; the first two coordinates of such a chart (X, Y, theta, phi with a metric that does not
            // depend on the angles) go through to-spherical and back unchanged - up to rounding, and that is the catch: ingoing
            // Eddington-Finkelstein time is t + r*, so the round trip of dX/dlambda adds and subtracts (dr/dlambda) / (1 - rs / r), which at
            // r = 1.0002 rs is 6 000 against a result of 2.  With exact division (the reference's x86 build) that costs four digits
            // of seven; with v_rcp_f32 it cost all of them for the step that lands there (tests/fuzz_paths.py: dX/dlambda = -0.67 for
            // 2.05).  They are taken as they are.
            generic_velocity_out.x = velocity.x / old_dlambda;
            generic_velocity_out.y = velocity.y / old_dlambda;
            // (the position's too: the same chart's time is t + r + rs ln(r / rs - 1), whose round trip is as ill-conditioned there)
            generic_position_out.x = position.x;
            generic_position_out.y = position.y;
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
