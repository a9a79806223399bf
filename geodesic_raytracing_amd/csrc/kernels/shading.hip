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
        // fmod(x, 1) is x - trunc(x), and a quotient by 2^n a product with 2^-n: the same values as the reference's fmod / exp2 /
        // divide without the library routines (two probes a pixel at the least, eight at the most, each through here)
        uv.x = uv.x - __builtin_truncf(uv.x);
        uv.y = uv.y - __builtin_truncf(uv.y);
        const float fine = floorf(lod), coarse = ceilf(lod);
        const float fine_scale = __builtin_ldexpf(1.f, -(int)fine), coarse_scale = __builtin_ldexpf(1.f, -(int)coarse);
        const float4 a = bilinear(uv.x * fine_scale, uv.y * fine_scale, fine);
        const float4 b = bilinear(uv.x * coarse_scale, uv.y * coarse_scale, coarse);
        return a + (b - a) * (lod - fine);
    }
};

// Footprint of a pixel in texels: the ellipse  A u^2 + B u v + C v^2 = 1  spanned by the texture-space images of the pixel's two
// edges, each padded by one texel (the "+ 1" that keeps a vanishing footprint from collapsing), reduced to its axes.
struct sky_footprint {
    float long_radius, short_radius;
    float cos_angle, sin_angle;   // of the reference's angle = atan2(b, (a - c) / 2): the direction the probes are spaced along
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
    {   // cos and sin of atan2(y, x) are x / hypot and y / hypot: no angle is ever needed (atan2f + cosf + sinf were a tenth of the pass)
        const float x = (a - c) / 2, y = b;
        const float hyp = __builtin_sqrtf(x * x + y * y);
        f.cos_angle = hyp > 0.f ? x / hyp : 1.f;   // atan2(0, 0) = 0
        f.sin_angle = hyp > 0.f ? y / hyp : 0.f;
    }
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
    float lod = __builtin_amdgcn_logf(f.short_radius);   // v_log_f32 is log2; short_radius >= 1
    const int coarsest = sky.levels - 1;
    if (lod > coarsest) { lod = coarsest; probes = 1; }
    if (probes <= 1) {
        if (probes < 1) lod = coarsest;
        return sky.trilinear(centre, lod);
    }
    const float span = 2 * (f.long_radius - f.short_radius);
    const float step_u = f.cos_angle * span / (probes - 1), step_v = f.sin_angle * span / (probes - 1);
    const float step_u_norm = step_u / sky.width, step_v_norm = step_v / sky.height;
    const float step2 = (step_u * step_u + step_v * step_v) / (f.long_radius * f.long_radius);
    // probe k sits at (2k - (probes - 1)) half steps from the centre; an even count starts one half step further out on the
    // low side (the reference's startN, cl.cl:5641-5650)
    int half_steps = (probes % 2) == 1 ? -2 * ((probes - 1) / 2) : -2 * (probes / 2) - 1;
    float4 sum = f4(0, 0, 0, 0);
    float weight_sum = 0;
    for (int k = 0; k < probes; k++, half_steps += 2) {
        const float weight = __builtin_amdgcn_exp2f(-2.88539008177792681472f * ((half_steps * half_steps / 4.f) * step2));   // exp(-2 d^2) on v_exp_f32
        const float offset = half_steps / 2.f;
        sum = sum + weight * sky.trilinear(make_float2(centre.x + offset * step_u_norm, centre.y + offset * step_v_norm), lod);
        weight_sum += weight;
    }
    return sum / weight_sum;
}

// colour (cl.cl:326-350, 5366-5413)
// x^y for x > 0 on v_log_f32 / v_exp_f32: the colour curves take powers of values in (0, 1] with |log2 x| < 9, where the error of this
// form is below 2e-7 of the result (the library's pow: ~100 instructions, three to six of them a pixel)
__device__ __forceinline__ float colour_pow(float x, float y) { return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)); }
__device__ __forceinline__ float srgb_to_linear(float v) { return v < 0.04045f ? v / 12.92f : colour_pow((v + 0.055f) / 1.055f, 2.4f); }
__device__ __forceinline__ float linear_to_srgb(float v) { return v <= 0.0031308f ? v * 12.92f : 1.055f * colour_pow(v, 1.0f / 2.4f) - 0.055f; }
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
    // (seen / reference)^3 as the reference writes it, pow(seen, 3) * luminance / pow(reference, 3), with the cubes multiplied out
    const float shifted_luminance = clampf((seen_wavelength * seen_wavelength * seen_wavelength) * luminance /
                                           (reference_wavelength * reference_wavelength * reference_wavelength), 0.f, 1.f);
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
    // atan2(sin d, cos d) is d brought into (-pi, pi]: for the differences between neighbouring pixels that is d itself, across the
    // texture's seam one turn is taken off.  (The library's sin, cos and atan2 here - four differences a pixel - were 400 of the
    // pass's ~1 000 vector instructions per pixel.)
    // One value takes neither road: |d| = float(pi) exactly - a record in the equatorial plane (v = 0.5) next to a black one (0, 0), i.e.
    // the shadow's edge on the middle row of a frame with an even number of rows.  float(pi) lies 8.7e-8 beyond pi, so sin d has the
    // sign opposite to d's and atan2(sin d, cos d) is the float below pi with the OTHER sign: the footprint's long axis is mirrored
    // (found by the round-5 fixture kerr_max_probes_16, one pixel off by 2e-2; invisible with up to 8 probes on the odd-rowed fixtures).
    const float two_pi = 6.283185307179586f, pi_f = 3.14159274101257324f, below_pi = 3.14159250259399414f;
    float wrapped = __builtin_fabsf(d) <= pi_f ? d : d - two_pi * __builtin_rintf(d / two_pi);
    if (__builtin_fabsf(d) == pi_f) wrapped = d > 0 ? -below_pi : below_pi;
    return (float)((double)(1.f * wrapped) / (2 * GR_PI));
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
#if GR_FRAME_KERNELS
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
#endif  // GR_FRAME_KERNELS

