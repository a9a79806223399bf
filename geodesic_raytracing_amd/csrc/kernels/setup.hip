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
    // (every index into these small arrays is a literal after unrolling - "the k that equals the index" instead of the index itself -
    // so that they live in registers: indexed by a register they were 64 bytes of scratch per lane in every kernel that computes a
    // tetrad per ray, the fused trace with redshift among them)
#pragma unroll
    for (int k = 1; k < 4; k++)
        if (k == index_swap) {
            swap4(arr[0], arr[k]);
            const float l = lengths[0]; lengths[0] = lengths[k]; lengths[k] = l;
        }
    int indices[4] = {0, 1, 2, 3};
    int first_nonzero = -1;
    const float eps = 0.00001f;
#pragma unroll
    for (int i = 3; i >= 0; i--)
        if (!(__builtin_fabsf(lengths[i]) <= eps)) first_nonzero = i;   // the first one: the last assignment of a descending scan
    if (first_nonzero == -1) first_nonzero = 0;
#pragma unroll
    for (int k = 1; k < 4; k++)
        if (k == first_nonzero) {
            swap4(arr[0], arr[k]);
            const int q = indices[0]; indices[0] = indices[k]; indices[k] = q;
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
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int old_index = indices[i];
#pragma unroll
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
    // The rotation turns angles: the chart's first two coordinates and their rates come back from the round trip through spherical
    // coordinates as they went in, up to rounding - which in a chart whose time mixes in r* (ingoing Eddington-Finkelstein) is the
    // difference of two terms ~ 1 / (1 - rs / r) near the horizon (geodesic_camera.hip has the measured case).  They are kept.
    const float first = position.x, second = position.y, first_rate = velocity.x, second_rate = velocity.y;
    position = gm::spherical_to_generic(f4(pos_sph.x, next_pos), cfg);
    velocity = gm::spherical_velocity_to_generic_velocity(f4(pos_sph.x, next_pos), f4(vel_sph.x, next_vel), cfg);
    position.x = first; position.y = second;
    velocity.x = first_rate; velocity.y = second_rate;
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

