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
        // The emitter's 4-velocity is the timelike leg of calculate_tetrads(position, no speed, no orientation) (cl.cl:5188-5208) - the only
        // leg the redshift reads.  Where the t axis of the chart is timelike at the ray's end (g_tt < -eps: everywhere but inside an
        // ergosphere or a superluminal warp wall) that leg is decided before any Gram-Schmidt: d/dt is the first non-null coordinate
        // vector, so it leads unswapped, and normalised it is the one leg of negative norm, so it is slot 0 - e0 = d/dt / sqrt|g_tt|, the
        // value the full construction returns (same functions, same operands).  The other three legs - six projections, three
        // normalisations, four norms: most of what an Alcubierre ray of 43 attempts costs outside its loop - are built only for a ray
        // that ends where g_tt >= -eps, or at a degenerate position.  (round 6)
        float g[16];
        gm::metric_big_at(position, g, cfg);
        float4 e0;
        if (!degenerate4(position) && g[0] < -0.00001f) e0 = normalise_metric(f4(1, 0, 0, 0), g);
        else {
            tetrad t;
            calculate_tetrads(position, f3(0, 0, 0), t, cfg, 0);
            e0 = t.e[0];
        }
        float4 obvs_low = lower_index_big(e0, g);
        float z_shift = (dot4(generic_velocity, obvs_low) / ku_uobsu) - 1;
        dat.z_shift = __builtin_fmaxf(z_shift, -0.999f);
    }
    dat.tex_coord = angle_to_tex(ipos.z, ipos.w);
    return dat;
}

// ================================================================================================
// kernels

#if GR_OTHER_KERNELS   // the reference-shaped sequence's kernels: not launched by a fused frame (program.hip: GR_BUILD_FRAME_PATH / GR_BUILD_REST)
extern "C" __global__ void gr_clear_termination_buffer(int* __restrict__ termination_buffer, int width, int height) {
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= width * height) return;
    termination_buffer[id] = 1;
}

// `tiled` is an extension over the reference signature: 0 = reference slot order (slot = cy*width+cx),
// 1 = 8x8 tile order (slot count is then rounded up to whole tiles; out-of-image slots get terminated = 2).
#define GR_INIT_BLOCK 256   // workgroup size of gr_init_rays_generic (capi.cpp launches with the same number)
extern "C" __global__ void __launch_bounds__(GR_INIT_BLOCK) gr_init_rays_generic(const float4* __restrict__ g_generic_camera_in, const float4* __restrict__ g_camera_quat,
                                                lightray* __restrict__ metric_rays, int* __restrict__ metric_ray_count,
                                                int width, int height, const int* __restrict__ termination_buffer,
                                                int prepass_width, int prepass_height, int flip_geodesic_direction,
                                                const float4* __restrict__ e0, const float4* __restrict__ e1,
                                                const float4* __restrict__ e2, const float4* __restrict__ e3,
                                                cfg_t cfg, dfg_t dfg, int i_am_prepass, int tiled) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    int cx = 0, cy = 0;
    const int T = GR_TILE;
    const int slots = tiled ? ((width + T - 1) / T) * ((height + T - 1) / T) * T * T : width * height;
    const bool full = i_am_prepass || !GET_FEATURE(adaptive_sampling, dfg) || GET_FEATURE(use_triangle_rendering, dfg);
    if (id == 0) *metric_ray_count = full ? slots : (height * width) / 4;

    lightray ray;
    bool quarter = false;   // adaptive sampling: the ray of a pixel (2x, 2y), to slot (y, x) of the quarter-size list
    if (id < slots) {
        if (!slot_to_pixel(id, width, height, tiled, cx, cy)) {
            ray.position = ray.velocity = ray.acceleration = f4(0, 0, 0, 0);
            ray.initial_quat = f4(0, 0, 0, 1);
            ray.ku_uobsu = 1; ray.running_dlambda_dnew = 1; ray.terminated = 2; ray.sx = -1; ray.sy = -1;
        } else {
            ray = make_pixel_ray(cx, cy, width, height, *g_generic_camera_in, *g_camera_quat, *e0, *e1, *e2, *e3, flip_geodesic_direction, cfg, dfg);
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
            quarter = !full && (cx % 2) == 0 && (cy % 2) == 0;
        }
    }
    if (!full) {
        if (quarter) metric_rays[(cy / 2) * (width / 2) + cx / 2] = ray;
        return;
    }
    // The records of a wave are 64 x 96 consecutive bytes.  Stored lane by lane they are six store instructions of 64 pieces of 16 bytes
    // 96 bytes apart - 0.34 ms for the 796 MB of a 4K frame, 2.3 TB/s; passed through LDS and stored as consecutive 16-byte pieces -
    // 1 KB per instruction - the same bytes go out at the rate the memory takes them.  (Wave-private staging: no workgroup barrier.)
    __shared__ float4 staging[GR_INIT_BLOCK / 64][64 * 6];
    const int lane = threadIdx.x % 64;
    float4* mine = staging[threadIdx.x / 64];
    if (id < slots) {
        mine[lane * 6 + 0] = ray.position;
        mine[lane * 6 + 1] = ray.velocity;
        mine[lane * 6 + 2] = ray.initial_quat;
        mine[lane * 6 + 3] = ray.acceleration;
        mine[lane * 6 + 4] = f4(ray.ku_uobsu, ray.running_dlambda_dnew, __int_as_float(ray.terminated), __int_as_float(ray.sx));
        mine[lane * 6 + 5] = f4(__int_as_float(ray.sy), 0, 0, 0);   // (the struct's padding: zeros)
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int first = id - lane;
    const int pieces = (slots - first < 64 ? slots - first : 64) * 6;
    float4* out = reinterpret_cast<float4*>(metric_rays + first);
#pragma unroll
    for (int k = 0; k < 6; k++)
        if (k * 64 + lane < pieces) out[k * 64 + lane] = mine[k * 64 + lane];
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

// gr_do_generic_rays for ray records in 8x8-tile slot order (gr_init_rays_generic's `tiled`), scheduled: as many waves as the device
// holds, each drawing tiles - 64 consecutive records - from a ticket counter, in the order of `tile_order` when there is one (the frame
// driver sorts the tiles dearest first by what they cost in the frame before: gr_sort_tiles_by_cost), and leaving what each tile cost -
// the attempts of its longest ray - in `tile_cost`.  Scheduling only: every record is what gr_do_generic_rays writes.  (Round 5: the
// reference-shaped sequence ran its trace in slot order, one workgroup per tile - a launch on its own then drains for a sixth of its time.)
#ifndef GR_SCHEDULED_WAVES
#define GR_SCHEDULED_WAVES GR_TRACE_WAVES
#endif
extern "C" __global__ void __launch_bounds__(64, GR_SCHEDULED_WAVES)
gr_do_generic_rays_scheduled(lightray* __restrict__ generic_rays_in, const int* __restrict__ generic_count_in, cfg_t cfg_in, dfg_t dfg_in,
                             unsigned long long* __restrict__ attempt_counter, unsigned int* __restrict__ tickets, int tile_count,
                             const unsigned int* __restrict__ tile_order, unsigned int* __restrict__ tile_cost) {
    GR_PARAMETERS_IN_REGISTERS
    const int count = *generic_count_in;
    unsigned long long attempts_of_wave = 0;
    for (;;) {
        unsigned int ticket = 0;
        if (threadIdx.x == 0) ticket = atomicAdd(tickets, 1u);
        ticket = __builtin_amdgcn_readfirstlane(ticket);
        if (ticket >= (unsigned int)tile_count) break;
        const unsigned int tile = tile_order ? tile_order[ticket] : ticket;
        const int id = (int)(tile * 64u + threadIdx.x);
        unsigned int tries = 0;
        if (id < count) {
            lightray* ray = &generic_rays_in[id];
            if (ray->terminated != 2) {
                ray_state s;
                s.position = ray->position;
                s.velocity = ray->velocity;
                s.acceleration = ray->acceleration;
                const int res = integrate_ray(s, cfg, dfg, &tries);
                if (res == RAY_TERMINATED) {
                    ray->position = s.position;
                    ray->velocity = s.velocity;
                    ray->running_dlambda_dnew = s.running_dlambda_dnew;
                    ray->terminated = 1;
                }
            }
        }
        attempts_of_wave += tries;
        if (tile_cost) {
            unsigned int longest = tries;
#pragma unroll
            for (int offset = 32; offset > 0; offset >>= 1) longest = max(longest, (unsigned int)__shfl_xor((int)longest, offset, 64));
            if (threadIdx.x == 0) tile_cost[tile] = longest;
        }
    }
    if (attempt_counter) atomicAdd(attempt_counter, attempts_of_wave);   // (one add per wave after the compiler's coalescing)
}

// The tiles of a frame sorted dearest first by `cost` (one word per tile, tiles_x to a row: what gr_do_generic_rays_scheduled left in the
// frame before): a tile takes the largest cost among itself and its eight neighbours - the long rays are filaments a pixel or two wide,
// and the picture moves a little from frame to frame -, classes of half an octave, dearest class first, any order inside a class.
// Two launches over `work` = 64 counts, 64 cursors (both zeroed by the caller), then a class per tile: count, then place.  (As ONE
// workgroup walking 130 000 tiles it took 0.4 ms - a tenth of the launch it was to shorten.)
__device__ __forceinline__ int tile_cost_class_of(const unsigned int* __restrict__ cost, int tile, int tiles_x, int tiles_y) {
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    unsigned int c = 0;
    for (int dy = -1; dy <= 1; dy++)
        for (int dx = -1; dx <= 1; dx++) {
            const int x = tx + dx, y = ty + dy;
            if (x >= 0 && y >= 0 && x < tiles_x && y < tiles_y) c = max(c, cost[y * tiles_x + x]);
        }
    if (c == 0) return 63;                                       // nothing traced around it: last
    const int octave = 31 - __clz((int)c);
    const int half = octave > 0 ? (int)((c >> (octave - 1)) & 1u) : 0;
    const int k = 2 * octave + half;                             // 0 .. 63, dear = large
    return k >= 62 ? 0 : 62 - k;                                 // dearest class first
}
extern "C" __global__ void __launch_bounds__(256)
gr_sort_tiles_count(const unsigned int* __restrict__ cost, int tiles_x, int tiles_y, unsigned int* __restrict__ work) {
    __shared__ unsigned int counts[64];
    if (threadIdx.x < 64) counts[threadIdx.x] = 0;
    __syncthreads();
    const int tile = blockIdx.x * blockDim.x + threadIdx.x;
    if (tile < tiles_x * tiles_y) {
        const int k = tile_cost_class_of(cost, tile, tiles_x, tiles_y);
        work[128 + tile] = (unsigned int)k;
        atomicAdd(&counts[k], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 64 && counts[threadIdx.x]) atomicAdd(&work[threadIdx.x], counts[threadIdx.x]);
}
extern "C" __global__ void __launch_bounds__(256)
gr_sort_tiles_place(int tile_count, unsigned int* __restrict__ work, unsigned int* __restrict__ order) {
    // a workgroup reserves one range per class for its 256 tiles (one global atomic per class and workgroup: a cursor bumped once per
    // tile - 130 000 atomics on a handful of addresses, most of them on the class of the untraced tiles - took 0.86 ms)
    __shared__ unsigned int base[64], mine[64], taken[64];
    if (threadIdx.x < 64) { mine[threadIdx.x] = 0; taken[threadIdx.x] = 0; }
    __syncthreads();
    const int tile = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned int k = tile < tile_count ? work[128 + tile] : 0u;
    if (tile < tile_count) atomicAdd(&mine[k], 1u);
    __syncthreads();
    if (threadIdx.x < 64) {
        unsigned int first = 0;
        for (unsigned int c = 0; c < threadIdx.x; c++) first += work[c];
        base[threadIdx.x] = first + (mine[threadIdx.x] ? atomicAdd(&work[64 + threadIdx.x], mine[threadIdx.x]) : 0u);
    }
    __syncthreads();
    if (tile < tile_count) order[base[k] + atomicAdd(&taken[k], 1u)] = (unsigned int)tile;
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
#endif  // GR_OTHER_KERNELS


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

// Does a device of a split frame (strip_count > 1) need the prepass cells of row cy?  A pixel row y reads the cell rows
// round(y * ph / H) - 1 .. + 1 (init_rays_generic's 5-point stencil, cl.cl:3213-3232), so cell row cy matters only if one of the
// device's blocks (or the halo row under it) meets the pixel rows that map to cy - 1 .. cy + 1 - one row of slack either side
// for the float rounding of that quotient (tests/test_distributed_cpu.py checks the rule by brute force).  row_margin: pixel rows
// beyond its blocks and halo rows the device also looks from (adaptive sampling: 2, the lattice rows its block decisions read).
__device__ __forceinline__ bool cell_row_matters(int cy, int prepass_height, int image_height, int block_rows, int strip_rank, int strip_count,
                                                 int row_margin) {
    if (strip_count <= 1) return true;
    long long lo = ((long long)(2 * cy - 3) * image_height) / (2 * prepass_height) - 1 - row_margin;
    long long hi = ((long long)(2 * cy + 3) * image_height + 2 * prepass_height - 1) / (2 * prepass_height) + 1 + row_margin;
    if (lo < 0) lo = 0;
    if (hi > image_height - 1) hi = image_height - 1;
    // blocks b (rows b*B .. (b+1)*B inclusive of the halo row) that meet [lo, hi]: (b+1)*B >= lo and b*B <= hi
    long long b_lo = (lo - 1) / block_rows, b_hi = hi / block_rows;
    if (b_lo < 0) b_lo = 0;
    const long long first = b_lo + (((long long)strip_rank - b_lo) % strip_count + strip_count) % strip_count;   // first own block >= b_lo
    return first <= b_hi;
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

// ---- parking ----------------------------------------------------------------------------------------
// gr_trace_fused_parking (PARKING below): a tile-wave that is down to its last few rays after a long time - a naked singularity's
// frame (BASELINE.json configs[2] read literally) has rays of 10 000 attempts scattered one or two to a tile, and a wave issues every
// instruction for 64 lanes whether 2 or 64 of them hold a ray: 9 % of that frame's issue slots - writes the state of those rays to a
// lot in device memory and draws its next tile; whoever draws next (any wave of the launch) first looks at the lot and, when 64 parked
// rays have come together, takes those instead of a tile and goes on with them (and may park again under the same rule).  Nothing
// waits for anything: a wave that finds no tile ticket left takes whatever the lot holds, and leaves when it holds nothing - the wave
// that parks last looks last, so nothing stays behind.  The arithmetic of a ray does not know where it is integrated: the loop's
// carried state (step suggestion, reparameterisation factor, step and attempt counts) travels with the ray.
//   records  GR_PARKED_FLOAT4 float4 per ray: position, velocity, acceleration, initial quaternion, (step suggestion, dlambda/dnew,
//            |v.x| at the start, k.u of the observer), (accepted steps, attempts, pixel x | y << 16, the tile-wave the ray came from)
//   words    [0] groups reserved | [1] records reserved (one 64-bit atomic), [2] groups handed out | [3] records handed out (one
//            64-bit compare-and-swap), [4] groups the lot had no room for (their wave went on itself), [5] waves that took parked rays,
//            [GR_LOT_HEADER + g] group g: 0 until its records are written, then 1 << 31 | rays << 24 | first record
// A group is what one wave parked at once (1 .. lanes - 1 rays).
struct parking_lot {
    float4* records;
    unsigned int* words;
    int lanes, trips;   // hand over when fewer than `lanes` rays are left after `trips` trips (two attempts each) of a visit; lanes 0: never
    int slots, groups;  // capacity
};
#define GR_PARKED_FLOAT4 6
#define GR_LOT_REFUSED 4
#define GR_LOT_WAVES 5
#define GR_LOT_HEADER 16

// The lot's records are written and read as device-scope accesses (the sc1 flavour of the load / store: coherent between the XCDs' L2s
// without any cache maintenance).  A release / acquire fence pair would be the textbook way and costs 100 ms a frame here: on this chip
// it is a write-back and an invalidation of an XCD's whole L2, for every parking event (measured: a = 0.9 at 4K, 15.6 -> 31.5 ms).
__device__ __forceinline__ void lot_store(float4* at, float4 v) {
    unsigned long long* words = reinterpret_cast<unsigned long long*>(at);
    __hip_atomic_store(words, (unsigned long long)__float_as_uint(v.x) | ((unsigned long long)__float_as_uint(v.y) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(words + 1, (unsigned long long)__float_as_uint(v.z) | ((unsigned long long)__float_as_uint(v.w) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float4 lot_load(const float4* at) {
    unsigned long long* words = reinterpret_cast<unsigned long long*>(const_cast<float4*>(at));
    const unsigned long long lo = __hip_atomic_load(words, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long hi = __hip_atomic_load(words + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return f4(__uint_as_float((unsigned int)lo), __uint_as_float((unsigned int)(lo >> 32)), __uint_as_float((unsigned int)hi), __uint_as_float((unsigned int)(hi >> 32)));
}

__device__ __forceinline__ bool park_rays(const parking_lot& lot, int lane, bool parked, const ray_state& s, float4 initial_quat, float ku_uobsu,
                                          int cx, int cy, int home_wave) {
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(parked);
    const unsigned int n = (unsigned int)__builtin_popcountll(mask);
    const bool leader = lane == (int)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true));
    unsigned long long old = 0;
    if (leader) old = atomicAdd(reinterpret_cast<unsigned long long*>(lot.words), ((unsigned long long)n << 32) | 1ull);
    const unsigned int g = __builtin_amdgcn_readfirstlane((unsigned int)old), base = __builtin_amdgcn_readfirstlane((unsigned int)(old >> 32));
    if (g >= (unsigned int)lot.groups || base + n > (unsigned int)lot.slots) {
        // no room: the group stays empty (whoever walks the groups steps over it) and the wave goes on with its rays itself
        if (leader) {
            if (g < (unsigned int)lot.groups) __hip_atomic_store(lot.words + GR_LOT_HEADER + g, 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicAdd(lot.words + GR_LOT_REFUSED, 1u);
        }
        return false;
    }
    if (parked) {
        const unsigned int rank = __builtin_amdgcn_mbcnt_hi((unsigned int)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mask, 0u));
        float4* record = lot.records + GR_PARKED_FLOAT4 * (size_t)(base + rank);
        lot_store(record + 0, s.position);
        lot_store(record + 1, s.velocity);
        lot_store(record + 2, s.acceleration);
        lot_store(record + 3, initial_quat);
        lot_store(record + 4, f4(s.next_ds, s.running_dlambda_dnew, s.f_in_x, ku_uobsu));
        lot_store(record + 5, f4(__int_as_float(s.steps), __int_as_float((int)s.tries), __int_as_float(cx | (cy << 16)), __int_as_float(home_wave)));
    }
    // the records have arrived before the word that announces them is sent
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (leader) __hip_atomic_store(lot.words + GR_LOT_HEADER + g, 0x80000000u | (n << 24) | base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

// Called by a whole wave.  Takes the groups at the front of the lot, as many as fill 64 lanes - or fewer when `anything` (no tile ticket
// is left) - and returns how many rays that is (0: nothing taken); record = this lane's, or -1.
__device__ __forceinline__ int claim_parked(const parking_lot& lot, int lane, bool anything, int& record) {
    record = -1;
    for (int round = 0; round < 8; round++) {
        // (one address for the whole wave, but read past the scalar cache: said to be uniform by hand)
        const unsigned long long reserved_now = __hip_atomic_load(reinterpret_cast<unsigned long long*>(lot.words), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long handed_now = __hip_atomic_load(reinterpret_cast<unsigned long long*>(lot.words + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int reserved_groups = __builtin_amdgcn_readfirstlane((unsigned int)reserved_now), reserved_rays = __builtin_amdgcn_readfirstlane((unsigned int)(reserved_now >> 32));
        const unsigned int g0 = __builtin_amdgcn_readfirstlane((unsigned int)handed_now), handed_rays = __builtin_amdgcn_readfirstlane((unsigned int)(handed_now >> 32));
        const unsigned long long handed = (unsigned long long)g0 | ((unsigned long long)handed_rays << 32);
        const unsigned int groups = min(reserved_groups, (unsigned int)lot.groups);
        if (g0 >= groups) return 0;
        // (cheap first look while tiles are left: not even 64 rays in the lot)
        if (!anything && reserved_rays - handed_rays < 64u) return 0;
        unsigned int info = 0;
        if ((unsigned int)lane < groups - g0) info = __hip_atomic_load(lot.words + GR_LOT_HEADER + g0 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long unwritten = __builtin_amdgcn_ballot_w64((info >> 31) == 0u);
        const int leading = unwritten ? (int)__builtin_ctzll(unwritten) : 64;
        if (leading == 0) return 0;   // the group in front is being written this very moment: its wave will look again after it
        const unsigned int n = lane < leading ? (info >> 24) & 63u : 0u;
        unsigned int sum = n;   // running sum of the groups' rays
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned int before = (unsigned int)__shfl_up((int)sum, d);
            if (lane >= d) sum += before;
        }
        const int G = __builtin_popcountll(__builtin_amdgcn_ballot_w64(lane < leading && sum <= 64u));   // (>= 1: a group is < 64 rays)
        const unsigned int total = __builtin_amdgcn_readfirstlane((unsigned int)__shfl((int)sum, G - 1));   // (uniform, and known to be)
        if (!anything && G == leading && total < 64u) return 0;   // more will come
        bool won = false;
        if (lane == 0)
            won = atomicCAS(reinterpret_cast<unsigned long long*>(lot.words + 2), handed,
                            (unsigned long long)(g0 + (unsigned int)G) | ((unsigned long long)(handed_rays + total) << 32)) == handed;
        if (!__builtin_amdgcn_readfirstlane((int)won)) continue;   // another wave was quicker: look again
        if (total == 0u) continue;                                  // nothing but groups that found no room
        asm volatile("" ::: "memory");   // (the records are read after the words: device-scope loads, issued in order)
        // lane j takes ray j: the first group whose running sum exceeds j
        int lo = 0, hi = G - 1;
#pragma unroll
        for (int it = 0; it < 6; it++) {
            const int mid = (lo + hi) >> 1;
            const unsigned int sum_mid = (unsigned int)__shfl((int)sum, mid);
            if (lo < hi) { if (sum_mid > (unsigned int)lane) hi = mid; else lo = mid + 1; }
        }
        const unsigned int info_l = (unsigned int)__shfl((int)info, lo), sum_l = (unsigned int)__shfl((int)sum, lo);
        const unsigned int first = sum_l - ((info_l >> 24) & 63u);
        if ((unsigned int)lane < total) record = (int)((info_l & 0xffffffu) + ((unsigned int)lane - first));
        if (lane == 0) atomicAdd(lot.words + GR_LOT_WAVES, 1u);
        return (int)total;
    }
    return 0;
}

// LATTICE_RAYS: the instantiation of gr_trace_fused_lattice, which also leaves its rays' end states behind (below).  A kernel of its
// own so that gr_trace_fused stays exactly the code it was: as a run-time option the three stores cost the headline kernel its
// seventh wave per SIMD (72 VGPRs + 48 B -> 80 + 36 B at six).
template <bool LATTICE_RAYS, bool PARKING = false>
__device__ __forceinline__ void trace_tile(int wave, int lane, const float4* __restrict__ camera, const float4* __restrict__ camera_quat,
                                           render_data* __restrict__ rdata, int width, int height, int block_rows, int strip_rank,
                                           int strip_count, const int* __restrict__ termination_buffer, int prepass_width,
                                           int prepass_height, const float4* __restrict__ e0, const float4* __restrict__ e1,
                                           const float4* __restrict__ e2, const float4* __restrict__ e3, cfg_t cfg, dfg_t dfg,
                                           unsigned long long* __restrict__ attempt_counter, int lattice, int pending_only,
                                           const trace_shading& shading, bool known_skipped, int cell_wave, bool cells_in_flight,
                                           unsigned int* __restrict__ tile_cost, float4* __restrict__ lattice_rays,
                                           const parking_lot* lot = nullptr, int record = -1, bool from_lot = false, bool speculative = false,
                                           int guess_wave = -1, unsigned int* __restrict__ guessed = nullptr) {
    // guess_wave >= 0 (the lattice launch of adaptive sampling; `guessed`: gr_guessed_bytes): this "tile" is 64 pixels of the list of
    // pixels the frame before had to trace in its second launch and found dear - traced here, beside the lattice, before anybody knows
    // whether this frame's decisions will ask for them; their records wait in `guessed` for gr_apply_guessed.
    // speculative (wave-uniform; a launch with cells in flight only): the tile does not wait for the prepass cells its pixels look at -
    // it traces every pixel at once and looks the verdicts up when its rays have ended; a pixel the prepass skips gets the skipped
    // record then, as if its ray had never been traced (trace_fused_body says which tiles: the ones on the launch's critical path).
    // PARKING with from_lot (wave-uniform): this "tile" is up to 64 parked rays (claim_parked), lane by lane the record it was dealt.
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
    if (PARKING && from_lot) {
        if (record < 0) return;
        const float4 where = lot_load(lot->records + GR_PARKED_FLOAT4 * (size_t)record + 5);
        cx = __float_as_int(where.z) & 0xffff;
        cy = (int)((unsigned int)__float_as_int(where.z) >> 16);
        wave = __float_as_int(where.w);   // (per lane here: only the ray's cost, at its end, is filed under it)
    } else if (LATTICE_RAYS && guess_wave >= 0) {
        const unsigned int entry = (unsigned int)guess_wave * 64u + (unsigned int)lane;
        if (entry >= min(guessed[0], (unsigned int)GR_GUESSED_CAPACITY)) return;
        const unsigned int pixel = guessed[GR_GUESSED_HEADER + entry];
        cx = (int)(pixel % (unsigned int)image_width);
        cy = (int)(pixel / (unsigned int)image_width);
        width = image_width; height = image_height;
    } else if (cell_wave >= 0) {
        // a cell wave is 8 x 8 cells, not 64 of a row: a wave publishes its verdicts when its longest ray has ended, the long rays lie along
        // the shadow's edge, and a curve crosses far fewer blocks than rows - the other waves' cells are known early (GR_CELL_BLOCK=0: rows)
#if GR_CELL_BLOCK
        const int blocks_x = (prepass_width + 7) / 8;
        cx = (cell_wave % blocks_x) * 8 + lane % 8;
        cy = (cell_wave / blocks_x) * 8 + lane / 8;
        if (cx >= prepass_width || cy >= prepass_height) return;
#else
        const int cell = cell_wave * 64 + lane;
        if (cell >= prepass_width * prepass_height) return;
        cx = cell % prepass_width;
        cy = cell / prepass_width;
#endif
        // a device of a split frame traces the cells its rows look at; the others stay unknown and nobody asks for them
        if (!cell_row_matters(cy, prepass_height, image_height, device_block_rows, device_rank, device_count, 0)) return;
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
    const bool looks_at_cells = cell_wave < 0 && !(LATTICE_RAYS && guess_wave >= 0) && !known_skipped && !pending_only && !(PARKING && from_lot) && termination_buffer && prepass_width != width &&
                                prepass_height != height;
    if (looks_at_cells && !speculative) {
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
    bool parked_here = false;   // PARKING: this lane's ray went to the lot - whoever ends it writes its record
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
        lightray ray;
        ray_state s;
        int res = RAY_LOST;
        if (PARKING) {
            unsigned int tries_before = 0;
#ifdef GR_PROBE_LOT   // (-DGR_PROBE_LOT: how long the waves that took parked rays were at it - counter words 4..6: ticks, attempts of each visit's longest stay, visits)
            const unsigned long long visit_began = (attempt_counter && from_lot) ? __builtin_amdgcn_s_memrealtime() : 0ull;
#endif
            if (from_lot) {
                const float4* parked = lot->records + GR_PARKED_FLOAT4 * (size_t)record;
                s.position = lot_load(parked + 0);
                s.velocity = lot_load(parked + 1);
                s.acceleration = lot_load(parked + 2);
                ray.initial_quat = lot_load(parked + 3);
                const float4 carried = lot_load(parked + 4), counts = lot_load(parked + 5);
                s.next_ds = carried.x; s.running_dlambda_dnew = carried.y; s.f_in_x = carried.z; ray.ku_uobsu = carried.w;
                s.steps = __float_as_int(counts.x); s.tries = (unsigned int)__float_as_int(counts.y);
                tries_before = s.tries;
                ray.position = ray.velocity = f4(0, 0, 0, 0);   // (where the ray started: nothing in a record depends on it, make_render_data)
            } else {
                ray = make_pixel_ray(cx, cy, ray_grid_width, ray_grid_height, *camera, *camera_quat, *e0, *e1, *e2, *e3, 0, cfg, dfg);
                s.position = ray.position;
                s.velocity = ray.velocity;
                s.acceleration = ray.acceleration;
                s.running_dlambda_dnew = 1;
            }
            // one call site for a fresh ray, a parked one, and a ray the lot had no room for (it goes on here, never to be parked again)
            bool going = true, resumed = from_lot;
#ifndef GR_PARK_AGAIN
#define GR_PARK_AGAIN 1   // 0: rays that were parked once are integrated to their ends by the wave that took them
#endif
            int keep_lanes = (cell_wave >= 0 || (from_lot && !GR_PARK_AGAIN)) ? 0 : lot->lanes;
            for (;;) {
                bool paused = false;
                if (going) res = integrate_pingpong<false, true>(s, cfg, dfg, &tries, keep_lanes, paused, resumed, (unsigned int)lot->trips);
                going = going && paused;
                if (__builtin_amdgcn_ballot_w64(going) == 0) break;
                if (park_rays(*lot, lane, going, s, ray.initial_quat, ray.ku_uobsu, cx, cy, wave)) { parked_here = going; break; }
                keep_lanes = 0;
                resumed = true;
            }
            // a ray that ends here after a stay in the lot: what it cost in all is its tile's cost for the next frame's order (the tile's wave
            // only saw it to the lot) - one atomic per such ray
            if (from_lot && tile_cost && !parked_here) atomicMax(tile_cost + wave, tries);
            tries -= tries_before;   // (the attempts made here: what the counters below add up)
#ifdef GR_PROBE_LOT
            if (attempt_counter && from_lot) {
                unsigned long long lanes_here = __builtin_amdgcn_ballot_w64(true);
                const bool leader = lane == (int)__builtin_ctzll(lanes_here);
                unsigned int longest = 0;
                while (lanes_here) {
                    const int l = (int)__builtin_ctzll(lanes_here);
                    const unsigned int v = (unsigned int)__builtin_amdgcn_readlane((int)tries, l);
                    longest = v > longest ? v : longest;
                    lanes_here &= lanes_here - 1;
                }
                if (leader) {
                    atomicAdd(attempt_counter + 4, (unsigned long long)__builtin_amdgcn_s_memrealtime() - visit_began);
                    atomicAdd(attempt_counter + 5, (unsigned long long)longest);
                    atomicAdd(attempt_counter + 6, 1ull);
                }
            }
#endif
        } else {
            ray = make_pixel_ray(cx, cy, ray_grid_width, ray_grid_height, *camera, *camera_quat, *e0, *e1, *e2, *e3, 0, cfg, dfg);
            s.position = ray.position;
            s.velocity = ray.velocity;
            s.acceleration = ray.acceleration;
            s.running_dlambda_dnew = 1;
            res = integrate_ray(s, cfg, dfg, &tries);
        }
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
        if (LATTICE_RAYS && guess_wave >= 0) {
            // (what gr_trace_pending would have written, had the pixel been on its list already: gr_apply_guessed hands it over if it is)
            const unsigned int entry = (unsigned int)guess_wave * 64u + (unsigned int)lane;
            guessed[GR_GUESSED_HEADER + GR_GUESSED_CAPACITY + entry] = tries;
            reinterpret_cast<render_data*>(guessed + GR_GUESSED_HEADER + 2 * GR_GUESSED_CAPACITY)[entry] = dat;
            return;
        }
        if (looks_at_cells && speculative) {
            // the verdict, after the fact: by now the cells' rays have usually ended too (they are as long as this tile's, and began with it)
            const int lx = (int)roundf(exact_ratio(cx, width) * prepass_width), ly = (int)roundf(exact_ratio(cy, height) * prepass_height);
            if (early_terminate_stencil_when_known(lx, ly, prepass_width, prepass_height, termination_buffer)) {
                dat.tex_coord = make_float2(0, 0);
                dat.z_shift = 0;
                dat.sx = cx;
                dat.sy = cy;
                dat.terminated = 2;
                dat.side = 1;
                tries = 0;   // (what the frame's counters and the next frame's order see: the attempts of the rays the frame needed)
            }
        }
        // The lattice launch of adaptive sampling leaves where its rays ended - position, velocity, the quaternion of the rotated frame -
        // for gr_adaptive_refine, which needs what the reference reads off its ray records (get_intersection_position of every
        // neighbour, also of rays whose record stays black: cl.cl:5260-5268).  Three stores of values that are live anyway: working the
        // sky angles out here gave the intersection's products a second use, the compiler then formed other fmas, and the records of
        // every launch came out rounded differently (Schwarzschild 1000 x 500 against gr_trace_compact, pixels apart by > 1e-4:
        // 3e-5 of the frame -> 1.3e-3); doing it on laundered copies cost the headline kernel a wave per SIMD.
        if (LATTICE_RAYS && lattice_rays && lattice == 2 && dat.terminated != 2) {   // (2: a speculative tile's pixel that the prepass skips after all - as if never traced)
            float4* record = lattice_rays + 3 * ((size_t)(cy / 2) * (image_width / 2) + cx / 2);
            record[0] = s.position;
            record[1] = s.velocity;
            record[2] = ray.initial_quat;
        }
    }
    // ... and, behind the end states, what every lattice ray cost (its attempts; 0 for a pixel the prepass skipped): the estimate
    // gr_adaptive_refine orders the pixels of the second launch by
    if (LATTICE_RAYS && lattice_rays && lattice == 2 && cell_wave < 0) {
        const size_t lattice_pixels = (size_t)(image_width / 2) * (image_height / 2);
        reinterpret_cast<unsigned int*>(lattice_rays + 3 * lattice_pixels)[(size_t)(cy / 2) * (image_width / 2) + cx / 2] = tries;
    }
    if (!(PARKING && parked_here)) rdata[cy * width + cx] = dat;
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
    GR_PROBE_WAVE_SLOTS(tries, 64u)
    // One add per tile-wave after the compiler's wave reduction - into one of GR_ATTEMPT_COUNTERS words chosen by the workgroup,
    // not into one word: a same-address atomic per tile is a second ticket counter (an 8K Alcubierre frame, 518 400 short tiles,
    // measured 6.5 ms counted into one word and 3.7 ms uncounted).  The host adds the words up (gr_render_state_attempts).
    if (attempt_counter) atomicAdd(attempt_counter + GR_ATTEMPT_COUNTERS_AT + (blockIdx.x % GR_ATTEMPT_COUNTERS), (unsigned long long)tries);
    // what the tile cost - the attempts of its longest ray, which is how long the wave was busy - for the next frame's order
    // (gr_order_tiles_by_history): again one atomic per tile-wave after the compiler's wave reduction, every tile to a word of its own
    // (a tile that parked rays: how long its wave was busy with it, which is what the order is for)
    if (tile_cost && !(PARKING && from_lot)) atomicMax(tile_cost + __builtin_amdgcn_readfirstlane(wave), tries);   // (uniform by construction; said so for the reduction)
}

// workgroup size of the fused trace kernel: 4 tile-waves, one per SIMD of a CU (capi.cpp launches with the same number)
#ifndef GR_TRACE_BLOCK
#define GR_TRACE_BLOCK 256
#endif
// Two scheduling modes.  tile_counter == NULL: wave w of the launch traces tile w (grid = all tiles).  tile_counter != NULL:
// persistent waves - the launch only fills the machine and every wave keeps drawing the next tile from the device-side
// counter until total_waves are handed out, so a SIMD slot never idles between the end of a short tile (prepass-skipped
// tiles finish in a few hundred cycles) and the dispatcher's next workgroup.
template <bool LATTICE_RAYS, bool PARKING = false>
__device__ __forceinline__ void trace_fused_body(const float4* __restrict__ g_generic_camera_in, const float4* __restrict__ g_camera_quat,
               render_data* __restrict__ rdata, int width, int height, int block_rows, int strip_rank, int strip_count,
               const int* __restrict__ termination_buffer, int prepass_width, int prepass_height,
               const float4* __restrict__ e0, const float4* __restrict__ e1, const float4* __restrict__ e2, const float4* __restrict__ e3,
               cfg_t cfg_in, dfg_t dfg_in, unsigned long long* __restrict__ attempt_counter, unsigned int* __restrict__ tile_counter,
               int total_waves, int lattice, int pending_only, const unsigned int* __restrict__ tile_order, trace_shading shading,
               int prepass_tickets, int ticket_tiles, unsigned int* __restrict__ tile_cost, int last_class_is_skipped,
               float4* __restrict__ lattice_rays, parking_lot lot = parking_lot(), unsigned int* __restrict__ guessed = nullptr) {
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
    // ticket space: the prepass's cell waves, then the tiles - in the list's order if there is one
    // ... then (the lattice launch of adaptive sampling, `guessed`) the waves of the pixels guessed for the second launch (trace_tile), then the tiles
    const int guess_tickets = (LATTICE_RAYS && guessed && tile_counter) ? (int)((min(guessed[0], (unsigned int)GR_GUESSED_CAPACITY) + 63u) / 64u) : 0;
    const int cell_tickets = (prepass_tickets > 0 ? prepass_tickets : 0) + guess_tickets;   // (everything in front of the tiles)
    const int tickets_total = total_waves + cell_tickets;
    const int singles = (tile_counter && tile_order) ? tickets_total - (int)tile_order[GR_TILE_CLASSES - 1] : tickets_total;
    // the list says itself whether its last class is a promise (program.hip GR_LIST_BY_PREPASS)
    const bool list_promises = tile_order && ((last_class_is_skipped & 1) || tile_order[GR_TILE_ORDER_HEADER + total_waves] == GR_LIST_BY_PREPASS);
    bool tickets_gone = false;   // PARKING: no tile left to draw, what the lot holds is all there is
    // Speculative tiles (the upper bits of last_class_is_skipped: how many of the list's classes, dearest first): with the prepass's cells
    // traced by this launch and the tiles in the order of the frame before's costs, the launch's critical path is the longest cell ray
    // followed by the longest tile, which looks at that cell.  The tiles of the first classes - a few hundred of a 4K Kerr frame's 130 000 -
    // are not held up: they trace all their pixels from t = 0 and take the verdicts afterwards (trace_tile).  What that wastes are the
    // rays of their pixels the prepass would have skipped, in lanes that would have idled.
    int speculative_tiles = 0;
    if (!PARKING && tile_counter && tile_order && prepass_tickets > 0 && !list_promises)
        for (int c = 0; c < (last_class_is_skipped >> 8) && c < GR_TILE_CLASSES - 2; c++) speculative_tiles += (int)tile_order[c];
    bool speculative = false;
    for (;;) {
        // (the pointers laundered - below - at the top of the loop here: one value on every way back to it)
        if (PARKING) asm volatile("" : "+s"(g_generic_camera_in), "+s"(g_camera_quat), "+s"(e0), "+s"(e1), "+s"(e2), "+s"(e3));
        // PARKING (persistent launches only): before a ticket is drawn, a look at the lot - 64 parked rays are the longest work there is
        bool from_lot = false;
        int record = -1;
        if (PARKING && held == 0) {
            from_lot = __builtin_amdgcn_readfirstlane(claim_parked(lot, lane, tickets_gone, record)) > 0;
            if (!from_lot && tickets_gone) break;
        }
        int cell_wave = -1, guess_wave = -1;
        if (PARKING && from_lot) { wave = 0; known_skipped = false; }
        else {
        if (tile_counter) {
            if (held == 0) {
                unsigned int ticket = 0;
                if (lane == 0) ticket = atomicAdd(tile_counter, 1u);
                const int drawn = (int)__builtin_amdgcn_readfirstlane(ticket);
                // a ticket is ticket_tiles consecutive entries of the tiles that trace (1 unless the launch has far more tiles than
                // waves: the one counter serves ~10^8 tickets a second, and the 518 400 short tiles of an 8K Alcubierre frame
                // were waiting for it more than they traced), then GR_SKIP_CHUNK entries of the last class
                const int single_tickets = (singles + ticket_tiles - 1) / ticket_tiles;
                known_skipped = list_promises && drawn >= single_tickets;
                if (drawn < single_tickets) {
                    cursor = drawn * ticket_tiles;
                    held = singles - cursor < ticket_tiles ? singles - cursor : ticket_tiles;
                } else {
                    cursor = singles + (drawn - single_tickets) * GR_SKIP_CHUNK;
                    held = tickets_total - cursor < GR_SKIP_CHUNK ? tickets_total - cursor : GR_SKIP_CHUNK;
                }
                if (held <= 0) {
                    if (PARKING) { tickets_gone = true; held = 0; continue; }
                    break;
                }
            }
            wave = (tile_order && cursor >= cell_tickets) ? cell_tickets + (int)tile_order[GR_TILE_ORDER_HEADER + cursor - cell_tickets] : cursor;
            speculative = cursor >= cell_tickets && cursor - cell_tickets < speculative_tiles;
            cursor++;
            held--;
        }
        if (cell_tickets > 0) {
            if (wave < prepass_tickets) { cell_wave = wave; wave = 0; }
            else if (wave < cell_tickets) { guess_wave = wave - (prepass_tickets > 0 ? prepass_tickets : 0); wave = 0; }
            else wave -= cell_tickets;
        }
        if (wave >= total_waves) break;
        }
        // Launder the camera / tetrad pointers once per tile: otherwise everything in the ray set-up that depends only on
        // them is hoisted out of the tile loop and held in registers across the integrator (94 instead of 64 VGPRs, i.e.
        // 5 instead of 8 waves per SIMD).  Re-reading 96 bytes through the scalar cache per tile is free by comparison.
        if (!PARKING) asm volatile("" : "+s"(g_generic_camera_in), "+s"(g_camera_quat), "+s"(e0), "+s"(e1), "+s"(e2), "+s"(e3));
        GR_PROBE_TILE_BEGAN
        trace_tile<LATTICE_RAYS, PARKING>(wave, lane, g_generic_camera_in, g_camera_quat, rdata, width, height, block_rows, strip_rank, strip_count,
                   termination_buffer, prepass_width, prepass_height, e0, e1, e2, e3, cfg, dfg, attempt_counter, lattice, pending_only, shading,
                   known_skipped && lattice == 1 && !pending_only, cell_wave, prepass_tickets > 0, tile_cost, lattice_rays, &lot, record, from_lot,
                   speculative && !pending_only, guess_wave, guessed);
        GR_PROBE_TILE_ENDED
        if (!tile_counter) break;
    }
    if (attempt_counter && lane == 0) {
        atomicAdd(attempt_counter + 1, (unsigned long long)__builtin_amdgcn_s_memtime() - born_cycles);
        atomicAdd(attempt_counter + 2, (unsigned long long)__builtin_amdgcn_s_memrealtime() - born_ticks);
        atomicAdd(attempt_counter + 3, 1ull);
        GR_PROBE_WAVE_ENDED
    }
}

#if GR_FRAME_KERNELS
extern "C" __global__ void __launch_bounds__(GR_TRACE_BLOCK, GR_FUSED_WAVES)
gr_trace_fused(const float4* __restrict__ g_generic_camera_in, const float4* __restrict__ g_camera_quat,
               render_data* __restrict__ rdata, int width, int height, int block_rows, int strip_rank, int strip_count,
               const int* __restrict__ termination_buffer, int prepass_width, int prepass_height,
               const float4* __restrict__ e0, const float4* __restrict__ e1, const float4* __restrict__ e2, const float4* __restrict__ e3,
               cfg_t cfg_in, dfg_t dfg_in, unsigned long long* __restrict__ attempt_counter, unsigned int* __restrict__ tile_counter,
               int total_waves, int lattice, int pending_only, const unsigned int* __restrict__ tile_order, trace_shading shading,
               int prepass_tickets, int ticket_tiles, unsigned int* __restrict__ tile_cost, int last_class_is_skipped,
               float4* __restrict__ lattice_rays) {
    trace_fused_body<false>(g_generic_camera_in, g_camera_quat, rdata, width, height, block_rows, strip_rank, strip_count, termination_buffer, prepass_width, prepass_height, e0, e1, e2, e3, cfg_in, dfg_in, attempt_counter, tile_counter, total_waves, lattice, pending_only, tile_order, shading, prepass_tickets, ticket_tiles, tile_cost, last_class_is_skipped, lattice_rays);
}
#endif  // GR_FRAME_KERNELS

// the lattice launch of adaptive sampling (lattice = 2 with lattice_rays): the same tiles, tickets and integrator
#if GR_ADAPTIVE_KERNELS
extern "C" __global__ void __launch_bounds__(GR_TRACE_BLOCK, GR_FUSED_WAVES)
gr_trace_fused_lattice(const float4* __restrict__ g_generic_camera_in, const float4* __restrict__ g_camera_quat,
               render_data* __restrict__ rdata, int width, int height, int block_rows, int strip_rank, int strip_count,
               const int* __restrict__ termination_buffer, int prepass_width, int prepass_height,
               const float4* __restrict__ e0, const float4* __restrict__ e1, const float4* __restrict__ e2, const float4* __restrict__ e3,
               cfg_t cfg_in, dfg_t dfg_in, unsigned long long* __restrict__ attempt_counter, unsigned int* __restrict__ tile_counter,
               int total_waves, int lattice, int pending_only, const unsigned int* __restrict__ tile_order, trace_shading shading,
               int prepass_tickets, int ticket_tiles, unsigned int* __restrict__ tile_cost, int last_class_is_skipped,
               float4* __restrict__ lattice_rays, parking_lot unused_lot, unsigned int* __restrict__ guessed) {
    trace_fused_body<true>(g_generic_camera_in, g_camera_quat, rdata, width, height, block_rows, strip_rank, strip_count, termination_buffer, prepass_width, prepass_height, e0, e1, e2, e3, cfg_in, dfg_in, attempt_counter, tile_counter, total_waves, lattice, pending_only, tile_order, shading, prepass_tickets, ticket_tiles, tile_cost, last_class_is_skipped, lattice_rays, parking_lot(), guessed);
}
#endif  // GR_ADAPTIVE_KERNELS

// gr_trace_fused with parking (above): persistent launches of whole-image or strip frames, every pixel (no lattice, no list).
// Programs built with -DGR_PARKING in their argument string only (gr_program_has_parking).
#if defined(GR_PARKING) && GR_FRAME_KERNELS
extern "C" __global__ void __launch_bounds__(GR_TRACE_BLOCK, GR_FUSED_WAVES)
gr_trace_fused_parking(const float4* __restrict__ g_generic_camera_in, const float4* __restrict__ g_camera_quat,
               render_data* __restrict__ rdata, int width, int height, int block_rows, int strip_rank, int strip_count,
               const int* __restrict__ termination_buffer, int prepass_width, int prepass_height,
               const float4* __restrict__ e0, const float4* __restrict__ e1, const float4* __restrict__ e2, const float4* __restrict__ e3,
               cfg_t cfg_in, dfg_t dfg_in, unsigned long long* __restrict__ attempt_counter, unsigned int* __restrict__ tile_counter,
               int total_waves, int lattice, int pending_only, const unsigned int* __restrict__ tile_order, trace_shading shading,
               int prepass_tickets, int ticket_tiles, unsigned int* __restrict__ tile_cost, int last_class_is_skipped,
               float4* __restrict__ lattice_rays, parking_lot lot) {
    trace_fused_body<false, true>(g_generic_camera_in, g_camera_quat, rdata, width, height, block_rows, strip_rank, strip_count, termination_buffer, prepass_width, prepass_height, e0, e1, e2, e3, cfg_in, dfg_in, attempt_counter, tile_counter, total_waves, 1, 0, tile_order, shading, prepass_tickets, ticket_tiles, tile_cost, last_class_is_skipped, nullptr, lot);
}
#endif

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
    GR_PROBE_WAVE_SLOTS_PAIR(tries0, tries1)
    if (attempt_counter && (has0 | has1)) atomicAdd(attempt_counter, (unsigned long long)tries0 + (unsigned long long)tries1);
}

#if GR_FRAME_KERNELS
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
#endif  // GR_FRAME_KERNELS
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
#if GR_OTHER_KERNELS
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
            outcome = integrate_pingpong<true>(s, cfg, dfg, nullptr, exhausted ? 1 : keep_lanes, paused);
            integrating = paused;
        }
    }
}
#endif  // GR_OTHER_KERNELS

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
    if (!cell_row_matters(cy, prepass_height, image_height, block_rows, strip_rank, strip_count, row_margin)) return;
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

#if GR_FRAME_KERNELS
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
#endif  // GR_FRAME_KERNELS

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

// The other estimate: what the tiles cost in the frame before (tile_history: the attempts of each tile's longest ray, left by that
// frame's gr_trace_fused).  A camera that moves a little per frame - any interactive one - sees nearly the same costs again, and
// they are exact where the prepass's one ray per 16x16 pixels only samples: the rays that circle the hole many times before they
// leave are filaments a pixel or two wide, and the tile one of them crosses is ten times its neighbours.  The tile's estimate is
// the largest cost among itself and the tiles within `reach` of it (the filament may have moved on by a tile or two).  Classes: an octave of
// attempts each, dearest first; the last class - nothing traced in the tile or within three times the reach - is handed out in chunks like
// the prepass order's, but as a guess, not a promise: its tiles look their pixels up like any other.
__device__ __forceinline__ int tile_history_class(int tile, int width, int block_rows, int strip_count, const unsigned int* __restrict__ tile_history,
                                                  int reach, int shift_x, int shift_y) {
    const int tiles_x = (width + GR_TILE - 1) / GR_TILE, tile_rows = block_rows / GR_TILE;
    const int per_block = tiles_x * tile_rows + (strip_count > 1 ? (width + 63) / 64 : 0);
    const int block = tile / per_block, within = tile % per_block;
    // (shift_x, shift_y): how many tiles the picture has moved since the history was recorded (the caller's estimate) - this tile
    // shows what the tile that far back showed then
    unsigned int dearest = (shift_x | shift_y) ? 0u : tile_history[tile];
    unsigned int around = dearest;   // ... and within three times the reach: the guard ring of the last class
    if (within < tiles_x * tile_rows) {
        const int tx = within % tiles_x - shift_x, ty = within / tiles_x - shift_y;
        for (int dy = -3 * reach; dy <= 3 * reach; dy++)
            for (int dx = -3 * reach; dx <= 3 * reach; dx++) {
                const int x = min(max(tx + dx, 0), tiles_x - 1), y = min(max(ty + dy, 0), tile_rows - 1);
                const unsigned int c = tile_history[block * per_block + y * tiles_x + x];
                around = c > around ? c : around;
                const bool near = dx >= -reach && dx <= reach && dy >= -reach && dy <= reach;
                dearest = (near && c > dearest) ? c : dearest;
            }
    }
    // The last class goes out 32 tiles to a ticket, so a tile that turns out to need tracing after all must not be in it: only
    // tiles with nothing traced within three times the reach - 48 px, and a history the picture has moved further from is not
    // followed at all (gr_render_frame) - so that chunks of 32 dear tiles to one wave, at the very end, cannot happen (a fast camera
    // measured 6.8 -> 9.8 ms with the last class one reach wide, a faster one 27 -> 85 ms with two).  The ring in between: the
    // cheapest single class.
    if (dearest == 0 && around != 0) return GR_TILE_CLASSES - 2;
    if (dearest == 0) return GR_TILE_CLASSES - 1;
    const int octave = 31 - __builtin_clz(dearest);   // 16384 attempts (the step cap) = 14
    return GR_TILE_CLASSES - 2 - (octave > GR_TILE_CLASSES - 2 ? GR_TILE_CLASSES - 2 : octave);
}

#if GR_FRAME_KERNELS
extern "C" __global__ void __launch_bounds__(1024)
gr_order_tiles(const int* __restrict__ termination_buffer, const unsigned int* __restrict__ cell_attempts, int prepass_width,
               int prepass_height, int width, int height, int block_rows, int strip_rank, int strip_count, int total_tiles,
               unsigned int* __restrict__ list, int phase, const unsigned int* __restrict__ tile_history, int history_reach,
               int history_shift_x, int history_shift_y) {
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
            cls = tile_history ? tile_history_class(tile, width, block_rows, strip_count, tile_history, history_reach, history_shift_x, history_shift_y)
                               : tile_cost_class(tile, width, height, block_rows, strip_rank, strip_count, termination_buffer, cell_attempts,
                                                 prepass_width, prepass_height);
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
    if (phase == 1 && tile == 0) classes[0] = tile_history ? GR_LIST_BY_HISTORY : GR_LIST_BY_PREPASS;   // (tile 0's own class was read above)
}
#endif  // GR_FRAME_KERNELS

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
// The pixels of the second launch as a LIST (pending_list != NULL; gr_adaptive_refine_list runs the kernel twice): phase 0 decides,
// marks and counts the marked pixels by what their rays are expected to cost - the dearest of the four lattice rays around the block,
// an octave of attempts per class (the lattice launch leaves the attempts behind its rays' end states) - phase 1 deals every marked
// pixel a place in its class's range, dearest class first.  gr_trace_pending then traces the list 64 entries to a wave: every lane has
// a ray (a tile of the image has 48 at best and most have a handful), the rays of a wave are neighbours of one cost class, and the
// longest rays of the frame - the refined pixels are the long ones: 12 % of a 4K Kerr frame's pixels, 59 % of what the adaptive
// frame traces - start first instead of wherever the image has them.  A quarter of an octave per class: the 64 rays of a wave run
// in lock step until the longest is done, and with whole octaves a wave's rays differed by up to 2x (the launch traced 191 M
// attempts a millisecond where a launch over 8x8 tiles of neighbours traces 270).  list[0..63] entries per class, [64..127] cursors,
// [128..] pixels (y * width + x).
#define GR_PENDING_CLASSES 64
#define GR_PENDING_HEADER (2 * GR_PENDING_CLASSES)
// class of a cost in attempts: dearest first, four to the octave (the two bits below the leading one)
__device__ __forceinline__ int pending_class(unsigned int cost) {
    if (cost < 4u) return GR_PENDING_CLASSES - 1;
    const int octave = 31 - __builtin_clz(cost);
    int fine = 4 * octave + (int)((cost >> (octave - 2)) & 3u);   // 8 .. 127
    if (fine > 4 * 15 + 3) fine = 4 * 15 + 3;                      // (the step cap is 16 384 = octave 14)
    return GR_PENDING_CLASSES - 1 - fine;
}
#if GR_ADAPTIVE_KERNELS
extern "C" __global__ void gr_adaptive_refine(render_data* __restrict__ rdat, int* __restrict__ pending_count, int width, int height,
                                              dfg_t dfg, int block_rows, int strip_rank, int strip_count,
                                              const float4* __restrict__ lattice_rays, cfg_t cfg, unsigned int* __restrict__ pending_list, int phase,
                                              const unsigned int* __restrict__ block_cost_before) {
    __shared__ unsigned int group_count[GR_PENDING_CLASSES], group_base[GR_PENDING_CLASSES];
    const int thread = threadIdx.y * blockDim.x + threadIdx.x;
    if (pending_list) {
        if (thread < GR_PENDING_CLASSES) group_count[thread] = 0;
        __syncthreads();
    }
    const int sx = blockIdx.x * blockDim.x + threadIdx.x;
    const int sy = blockIdx.y * blockDim.y + threadIdx.y;
    const int hw = width / 2, hh = height / 2;
    const int lsx = 2 * sx, lsy = 2 * sy;
    // split frame: only the pixel blocks whose rows this device shades or reads as a halo row (their lattice neighbours were traced)
    bool mine = sx < hw && sy < hh && !(strip_count > 1 && !own_block_within(lsy, 0, height, block_rows, strip_rank, strip_count));
    auto at = [&](int x, int y) -> render_data& { return rdat[y * width + x]; };
    bool refine = mine;
    if (pending_list && phase == 1) refine = mine && at(lsx + 1, lsy).terminated == GR_PENDING;
    if (!pending_list && !mine) return;
    if (mine && !(pending_list && phase == 1) && sx != 0 && sx != hw - 1 && sy != 0 && sy != hh - 1) {
        const render_data centre = at(lsx, lsy), left = at(lsx - 2, lsy), right = at(lsx + 2, lsy), up = at(lsx, lsy - 2), down = at(lsx, lsy + 2);
        const int down_right_flag = at(lsx + 2, lsy + 2).terminated;
        // (theta, phi) where each neighbour's ray meets the sky: the reference's get_intersection_position of the ray as the lattice
        // launch left it (lattice_rays) - also for rays whose record is black, whose texture coordinates say nothing: decided on
        // those, a frame with thin black features interpolated where the reference refines and refined where it interpolates, up
        // to a tenth of its pixels (found by the adaptive soak).  Only neighbours that reached the sky are asked: any other flag
        // differs from the centre's or - where all six are alike - the rays' untouched initial directions are as smooth as the
        // constant used here.  Without the buffer: back out of the texture coordinates (tex_to_angle gives (phi, theta)).
        auto sky = [&](const render_data& r, int x, int y) -> float2 {
            if (lattice_rays) {
                if (r.terminated != 1) return make_float2(0, 0);
                const float4* record = lattice_rays + 3 * ((size_t)(y / 2) * hw + x / 2);
                const float4 meets = intersection_position(record[0], record[1], record[2], cfg, dfg);
                return make_float2(meets.z, meets.w);
            }
            const float2 a = tex_to_angle(r.tex_coord);
            return make_float2(a.y, a.x);
        };
        const float2 la = sky(left, lsx - 2, lsy), ra = sky(right, lsx + 2, lsy), ua = sky(up, lsx, lsy - 2), da = sky(down, lsx, lsy + 2);
        const float x_error = __builtin_fabsf(angle_between_angles(la, ra));
        const float y_error = __builtin_fabsf(angle_between_angles(da, ua));
        const float relative_angular_error = (float)((double)(((x_error + x_error + y_error + y_error) / 4.f) / 2) * GR_PI);
        const float fov = GET_FEATURE(field_of_view, dfg);
        const float per_pixel = (float)((double)(fov * 2) * GR_PI / (double)360.f) / width;
        refine = relative_angular_error >= per_pixel * GET_FEATURE(adaptive_sampling_threshold, dfg);
        const int ct = centre.terminated;
        if (ct != left.terminated || ct != right.terminated || ct != up.terminated || ct != down.terminated || ct != down_right_flag) refine = true;
    }
    if (!(pending_list && phase == 1) && mine) {
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
    if (!pending_list) return;
    // the block's cost class: the dearest lattice ray at its four corners, an octave per class, dearest first
    int cls = -1;
    if (refine) {
        unsigned int dearest = 1;
        if (lattice_rays) {
            const unsigned int* cost = reinterpret_cast<const unsigned int*>(lattice_rays + 3 * (size_t)hw * hh);
            for (int dy = 0; dy < 2; dy++)
                for (int dx = 0; dx < 2; dx++)
                    if (sx + dx < hw && sy + dy < hh) { const unsigned int c = cost[(size_t)(sy + dy) * hw + sx + dx]; dearest = c > dearest ? c : dearest; }
        }
        // ... or what the block's own rays cost in the frame before (gr_trace_pending leaves the dearest of the three; the caller
        // passes it while the picture has moved little): the long rays of the shadow's edge are filaments a pixel or two wide, and the
        // lattice rays either side of one say nothing about it
        if (block_cost_before) { const unsigned int c = block_cost_before[(size_t)sy * hw + sx]; dearest = c > dearest ? c : dearest; }
        cls = pending_class(dearest);
    }
    // one atomic per class and workgroup on the list's counters (as gr_order_tiles): a block's place among its workgroup's blocks
    const int lane = thread % 64;
    unsigned int place = 0;
    // (the classes present in the wave, one after the other: a wave of neighbouring blocks holds a handful of the 64)
    for (unsigned long long todo = __builtin_amdgcn_ballot_w64(cls >= 0); todo;) {
        const int c = __builtin_amdgcn_readlane(cls, __builtin_ctzll(todo));
        const unsigned long long members = __builtin_amdgcn_ballot_w64(cls == c);
        const int leader = __builtin_ctzll(members);
        unsigned int wave_base = 0;
        if (lane == leader) wave_base = atomicAdd(&group_count[c], 3u * (unsigned int)__builtin_popcountll(members));
        wave_base = __builtin_amdgcn_readlane(wave_base, leader);
        if (cls == c) place = wave_base + 3u * (unsigned int)__builtin_popcountll(members & ((1ull << lane) - 1ull));
        todo &= ~members;
    }
    __syncthreads();
    if (thread < GR_PENDING_CLASSES && group_count[thread])
        group_base[thread] = atomicAdd(pending_list + (phase == 0 ? 0 : GR_PENDING_CLASSES) + thread, group_count[thread]);
    __syncthreads();
    if (phase == 1 && cls >= 0) {
        unsigned int first = 0;
        for (int c = 0; c < cls; c++) first += pending_list[c];
        unsigned int* entry = pending_list + GR_PENDING_HEADER + first + group_base[cls] + place;
        entry[0] = (unsigned int)(lsy * width + lsx + 1);
        entry[1] = (unsigned int)((lsy + 1) * width + lsx);
        entry[2] = (unsigned int)((lsy + 1) * width + lsx + 1);
    }
}
#endif  // GR_ADAPTIVE_KERNELS

// The second launch of adaptive sampling over gr_adaptive_refine's list: persistent waves, a ticket is 64 consecutive entries, every
// lane traces the pixel of its entry exactly as a tile-wave of gr_trace_fused would (make_pixel_ray, integrate_ray, make_render_data)
// and writes its record.  No prepass look-ups: a marked pixel is traced whatever the prepass said about its cell, as in the reference,
// whose second do_generic_rays runs over the rays handle_adaptive_sampling appended.
#if GR_ADAPTIVE_KERNELS
extern "C" __global__ void __launch_bounds__(GR_TRACE_BLOCK, GR_FUSED_WAVES)
gr_trace_pending(const float4* __restrict__ g_generic_camera_in, const float4* __restrict__ g_camera_quat, render_data* __restrict__ rdata,
                 int width, int height, const float4* __restrict__ e0, const float4* __restrict__ e1, const float4* __restrict__ e2,
                 const float4* __restrict__ e3, cfg_t cfg_in, dfg_t dfg_in, unsigned long long* __restrict__ attempt_counter,
                 unsigned int* __restrict__ ticket_counter, const unsigned int* __restrict__ pending_list, unsigned int* __restrict__ block_cost,
                 unsigned int* __restrict__ guessed_next) {
    // guessed_next (may be NULL; its count zeroed by the caller): the pixels that cost GR_GUESSED_ATTEMPTS attempts or more are left there for the
    // next frame's lattice launch to trace ahead (trace_tile's guess waves); a pixel gr_apply_guessed has served already is passed over
    // block_cost (may be NULL; zeroed by the caller): one word per 2x2 block of the image, left holding what the dearest of the block's
    // rays cost - the order of the next frame's list (gr_adaptive_refine's block_cost_before)
    GR_PARAMETERS_IN_REGISTERS
    const int lane = threadIdx.x % 64;
    unsigned int total = 0;
    for (int c = 0; c < GR_PENDING_CLASSES; c++) total += pending_list[c];
    for (;;) {
        unsigned int ticket = 0;
        if (lane == 0) ticket = atomicAdd(ticket_counter, 1u);
        const unsigned int first = (unsigned int)__builtin_amdgcn_readfirstlane(ticket) * 64u;
        if (first >= total) break;
        asm volatile("" : "+s"(g_generic_camera_in), "+s"(g_camera_quat), "+s"(e0), "+s"(e1), "+s"(e2), "+s"(e3));   // (as gr_trace_fused: nothing of the set-up hoisted over the loop)
        unsigned int tries = 0;
        const unsigned int pixel = first + lane < total ? pending_list[GR_PENDING_HEADER + first + lane] : 0u;
        if (first + lane < total && (!guessed_next || rdata[pixel].terminated == GR_PENDING)) {
            const int cx = (int)(pixel % (unsigned int)width), cy = (int)(pixel / (unsigned int)width);
            lightray ray = make_pixel_ray(cx, cy, width, height, *g_generic_camera_in, *g_camera_quat, *e0, *e1, *e2, *e3, 0, cfg, dfg);
            ray_state s;
            s.position = ray.position;
            s.velocity = ray.velocity;
            s.acceleration = ray.acceleration;
            s.running_dlambda_dnew = 1;
            int terminated = 0;
            if (integrate_ray(s, cfg, dfg, &tries) == RAY_TERMINATED) terminated = 1;
            else { s.position = ray.position; s.velocity = ray.velocity; s.running_dlambda_dnew = 1; }
            rdata[cy * width + cx] = make_render_data(s.position, s.velocity, ray.initial_quat, ray.ku_uobsu, s.running_dlambda_dnew, terminated,
                                                      cx, cy, cfg, dfg, GET_FEATURE(redshift, dfg) != 0);
            if (block_cost) atomicMax(block_cost + (size_t)(cy / 2) * (width / 2) + cx / 2, tries);
            if (guessed_next && tries >= GR_GUESSED_ATTEMPTS) {
                const unsigned int slot = atomicAdd(guessed_next, 1u);
                if (slot < GR_GUESSED_CAPACITY) guessed_next[GR_GUESSED_HEADER + slot] = pixel;
            }
        }
        if (attempt_counter) atomicAdd(attempt_counter + GR_ATTEMPT_COUNTERS_AT + (blockIdx.x % GR_ATTEMPT_COUNTERS), (unsigned long long)tries);
    }
}

// The guesses of the lattice launch (trace_tile's guess waves) against this frame's decisions: a guessed pixel that gr_adaptive_refine marked
// GR_PENDING gets the record traced ahead for it - the ray of a pixel is the ray of a pixel, whoever traces it and when - with its attempts
// counted and its cost left where gr_trace_pending would have left them; a guess the decisions did not ask for is dropped.  One lane per guess.
extern "C" __global__ void gr_apply_guessed(render_data* __restrict__ rdata, int width, const unsigned int* __restrict__ guessed,
                                            unsigned int* __restrict__ guessed_next, unsigned int* __restrict__ block_cost,
                                            unsigned long long* __restrict__ attempt_counter) {
    const unsigned int entry = blockIdx.x * blockDim.x + threadIdx.x;
    if (entry >= min(guessed[0], (unsigned int)GR_GUESSED_CAPACITY)) return;
    const unsigned int pixel = guessed[GR_GUESSED_HEADER + entry];
    if (rdata[pixel].terminated != GR_PENDING) return;
    const unsigned int tries = guessed[GR_GUESSED_HEADER + GR_GUESSED_CAPACITY + entry];
    rdata[pixel] = reinterpret_cast<const render_data*>(guessed + GR_GUESSED_HEADER + 2 * GR_GUESSED_CAPACITY)[entry];
    const int cx = (int)(pixel % (unsigned int)width), cy = (int)(pixel / (unsigned int)width);
    if (block_cost) atomicMax(block_cost + (size_t)(cy / 2) * (width / 2) + cx / 2, tries);
    if (attempt_counter) atomicAdd(attempt_counter + GR_ATTEMPT_COUNTERS_AT + (blockIdx.x % GR_ATTEMPT_COUNTERS), (unsigned long long)tries);
    if (guessed_next && tries >= GR_GUESSED_ATTEMPTS) {
        const unsigned int slot = atomicAdd(guessed_next, 1u);
        if (slot < GR_GUESSED_CAPACITY) guessed_next[GR_GUESSED_HEADER + slot] = pixel;
    }
}
#endif  // GR_ADAPTIVE_KERNELS

#if GR_OTHER_KERNELS
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
#endif  // GR_OTHER_KERNELS

