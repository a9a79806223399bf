/* geodesic_hip.h — C ABI of libgeodesic_hip.so, the MI355X (gfx950) implementation of the per-pixel
 * geodesic ray pipeline of 20k/geodesic_raytracing.
 *
 * The reference has no C function API for this path; its boundary is
 *   (1) the compile-time macro string  metrics::build_argument_string            metric.hpp:725-959
 *       + dynamic_feature_config::generate_{dynamic,static}_argument_string       dynamic_feature_config.cpp:122-180
 *       consumed by cl::build_program_with_cache({"cl.cl"}, ..., argument_string) metric_manager.hpp:88-108
 *   (2) launches by kernel name: cl::command_queue::exec(name, args, global, local)
 *       main.cpp:203, 2311, 2329, 2396, 2422, 2435, 2461, 2475, 2498, 2509, 2525
 * Every entry point below replaces one of those; the replaced reference interface is cited at each
 * declaration.  All device buffers are plain device pointers owned by the caller (the reference's
 * cl::buffer ownership, render_state.hpp:172-196); kernels never allocate.  `stream` is a
 * hipStream_t passed as void* (NULL = default stream); launches are asynchronous on it, like the
 * reference's single in-order queue (main.cpp:1460).
 *
 * Every function returns 0 on success and a negative gr_status otherwise; gr_last_error() returns
 * a thread-local message (the reference surfaces build errors through toolkit logging and
 * script/JSON problems as std::runtime_error; kernels themselves report nothing).
 */
#ifndef GEODESIC_HIP_H
#define GEODESIC_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum gr_status {
    GR_OK = 0,
    GR_ERROR_INVALID_ARGUMENT = -1,
    GR_ERROR_SCRIPT = -2,      /* metric script / JSON problem (std::runtime_error in the reference) */
    GR_ERROR_COMPILE = -3,     /* device program build failure */
    GR_ERROR_DEVICE = -4,      /* HIP runtime failure (no device, launch error, ...) */
    GR_ERROR_BUFFER_TOO_SMALL = -5
} gr_status;

const char* gr_last_error(void);

/* ---- data layouts shared with the device -------------------------------------------------- */

/* struct lightray, cl.cl:813-824 / render_state.hpp:8-19; 96 bytes */
typedef struct gr_lightray {
    float position[4];
    float velocity[4];
    float initial_quat[4];
    float acceleration[4];
    float ku_uobsu;
    float running_dlambda_dnew;
    int terminated;            /* 0 = lost/absorbed, 1 = reached a boundary, 2 = skipped by the prepass */
    int sx, sy;
    int pad_[3];
} gr_lightray;

/* struct render_data, cl.cl:5066-5074 / render_state.hpp:21-29; 32 bytes */
typedef struct gr_render_data {
    float tex_coord[2];
    float z_shift;
    int sx, sy;
    int terminated;
    int side;
    int pad_;
} gr_render_data;

/* struct dynamic_feature_config as packed by dynamic_feature_config::alloc_and_write_gpu_buffer
 * (dynamic_feature_config.cpp:182-237): floats in alphabetical order, then bools as int. 48 bytes.
 * Defaults: main.cpp:1123-1158. */
typedef struct gr_features {
    float adaptive_sampling_threshold;
    float field_of_view;
    float max_acceleration_change;
    float max_precision_radius;
    float min_step;
    float ray_skip;
    float universe_size;
    int adaptive_sampling;
    int redshift;
    int reparameterisation;
    int use_old_redshift;
    int use_triangle_rendering;
} gr_features;

void gr_features_default(gr_features* out);

/* ---- host side: metric -> macro string ----------------------------------------------------- */

typedef struct gr_metric gr_metric;

/* per-metric settings that steer the frame driver (metrics::metric_config, metric.hpp:330-357) */
typedef struct gr_metric_info {
    int is_big;                 /* GENERIC_BIG_METRIC */
    int is_constant_theta;      /* GENERIC_CONSTANT_THETA */
    int use_prepass;
    int adaptive_precision;
    float max_acceleration_change;
    int num_dynamic_vars;       /* $cfg.NAME parameters */
    int accel_ops;              /* DAG op count of GEO_ACCEL0..3 (VALU roofline accounting) */
    int accel_transcendentals;
    int coord_ops;              /* TO_COORDn + DISTANCE_FUNC */
} gr_metric_info;

/* One of the built-in metrics: "minkowski", "schwarzschild", "kerr_boyer", "alcubierre". */
int gr_metric_builtin(const char* name, gr_metric** out);

/* Loads <scripts_dir>/<name>.json (+ one level of inherit_settings) and the scripts it names, exactly
 * as content_manager.cpp:9-112 does; the script dialect is the reference's (js_interop.cpp:665-959). */
int gr_metric_load_script(const char* scripts_dir, const char* name, gr_metric** out);

/* A metric that is only its frame-driver settings - what gr_render_frame reads off a metric: gr_metric_info (prepass, tolerance) and
 * the $cfg names and defaults - for a caller that already holds the argument strings (a program cache of its own, a fixture):
 * gr_metric_argument_string fails on it.  var_names may be NULL. */
int gr_metric_from_info(const gr_metric_info* info, const char* const* var_names, const float* var_defaults, gr_metric** out);

void gr_metric_destroy(gr_metric* m);
int gr_metric_get_info(const gr_metric* m, gr_metric_info* out);
const char* gr_metric_dynamic_var_name(const gr_metric* m, int index);
float gr_metric_dynamic_var_default(const gr_metric* m, int index);

/* metrics::build_argument_string (metric.hpp:725-959).
 *   is_static = 0: "dynamic" program - expressions read cfg->NAME, features are read from the
 *                  feature struct (KERNEL_IS_DYNAMIC);
 *   is_static = 1: "substituted" program - cfg_values (NULL = defaults) and `features` are baked in
 *                  as literals (KERNEL_IS_STATIC), metric_manager.hpp:153-166.
 * Writes a NUL-terminated string; *needed receives the required capacity including the NUL.
 * buffer = NULL with capacity = 0 is a size query and returns GR_OK. */
int gr_metric_argument_string(const gr_metric* m, const gr_features* features, int is_static,
                              const float* cfg_values, int num_cfg_values,
                              char* buffer, size_t capacity, size_t* needed);

/* gr_metric_info's operation counts for the substituted program of these parameter values (NULL = defaults): parameters that
 * make parts of a metric vanish - real rod lengths in the complex-valued double-Kerr family - shrink the DAG a good deal. */
int gr_metric_substituted_op_counts(const gr_metric* m, const float* cfg_values, int num_cfg_values, int* accel_ops,
                                    int* accel_transcendentals, int* coord_ops);

/* ---- device program ------------------------------------------------------------------------- */

typedef struct gr_program gr_program;

/* cl::build_program_with_cache({"cl.cl"}, argument_string) (metric_manager.hpp:88-108): compiles the
 * ray kernels for gfx950 specialised by `argument_string` (the macro set above; unknown macros such
 * as CART_TO_POLn / FIX_LIGHTn / METRIC_TIME_G00 are accepted and ignored) and loads them on HIP
 * device `device`.  Code objects are cached on disk keyed by a hash of source + arguments. */
int gr_program_create(const char* argument_string, int device, gr_program** out);

/* Compile only (no device needed): fills the on-disk cache; used by the build step. */
int gr_program_precompile(const char* argument_string);

/* Background build + swap of the "substituted" program (metric_manager.hpp:153-166 builds it asynchronously,
 * check_substitution :172-219 swaps it in once is_built()).  create_async starts the compile on a worker thread and
 * returns at once; poll returns 1 and a loaded program when it is ready, 0 while pending, < 0 on a build error. */
typedef struct gr_program_future gr_program_future;
int gr_program_create_async(const char* argument_string, int device, gr_program_future** out);
int gr_program_future_poll(gr_program_future* f, gr_program** out);
void gr_program_future_destroy(gr_program_future* f);

void gr_program_destroy(gr_program* p);

/* metric_manager (metric_manager.hpp:19-219) as an object: which program to launch this frame.  gr_program_manager_create builds
 * the dynamic program of `m` (blocking, as the reference does for a newly selected metric) and starts the substituted build for
 * `features` / `cfg_values` (NULL = the metric's defaults) on a worker thread.  gr_program_manager_current is check_substitution:
 * called once per frame, it swaps the substituted program in as soon as its build has finished (wait != 0: waits for it) and hands
 * back the program to launch - owned by the manager, valid until the manager is destroyed or two later updates have retired it.
 * gr_program_manager_update is the soft recompile: values that differ from the current ones put the dynamic program back at
 * once, abandon a pending build and start the new one; equal values change nothing.  The metric must outlive the manager.
 * One thread at a time. */
typedef struct gr_program_manager gr_program_manager;
int gr_program_manager_create(const gr_metric* m, int device, const gr_features* features, const float* cfg_values, int num_cfg_values,
                              gr_program_manager** out);
int gr_program_manager_update(gr_program_manager* pm, const gr_features* features, const float* cfg_values, int num_cfg_values);
int gr_program_manager_current(gr_program_manager* pm, int wait, gr_program** program, int* is_substituted);
gr_program* gr_program_manager_dynamic(gr_program_manager* pm);
void gr_program_manager_destroy(gr_program_manager* pm);

/* registers / scratch of a kernel as recorded in the code object (0 if unknown) */
int gr_program_kernel_info(const gr_program* p, const char* kernel_name, int* vgprs, int* sgprs, int* scratch_bytes);

/* ---- launchers: one per reference kernel, reference argument order -------------------------- */

/* cart_to_generic_kernel, cl.cl:6018-6034; launched {1}/{1} at main.cpp:2311 */
int gr_cart_to_generic(gr_program* p, void* stream, const void* position_cart_in, void* position_generic_out,
                       int count, float flip, const void* cfg);

/* init_basis_vectors, cl.cl:2483-2507; main.cpp:2329.  cartesian_basis_speed is a float3 there. */
int gr_init_basis_vectors(gr_program* p, void* stream, const void* generic_in, int count,
                          const float cartesian_basis_speed[3],
                          void* e0_out, void* e1_out, void* e2_out, void* e3_out, const void* cfg);

/* clear_termination_buffer, cl.cl:4997-5006; main.cpp:2396 */
int gr_clear_termination_buffer(gr_program* p, void* stream, void* termination_buffer, int width, int height);

/* init_rays_generic, cl.cl:3143-3251; main.cpp:2422 (prepass) and :2461.
 * `tiled` (extension, pass 0 for reference behaviour): lay ray slots out in 8x8 pixel tiles; the ray
 * buffer must then hold gr_tiled_slot_count(width, height) rays. */
int gr_init_rays_generic(gr_program* p, void* stream, const void* camera_generic, const void* camera_quat,
                         void* rays, void* ray_count, int width, int height,
                         const void* termination_buffer, int prepass_width, int prepass_height,
                         int flip_geodesic_direction,
                         const void* e0, const void* e1, const void* e2, const void* e3,
                         const void* cfg, const void* dfg, int i_am_prepass, int tiled);
int gr_tiled_slot_count(int width, int height);

/* do_generic_rays, cl.cl:3954-4247; execute_kernel main.cpp:139-205.  `num_rays` sizes the grid
 * (the reference launches width*height work-items); the device-side count is still honoured.
 * ray_time_min/max, ray_write, mouse_x/y exist for signature parity (triangle path: unused here).
 * attempt_counter (extension, may be NULL): device uint64 accumulating Verlet step attempts. */
int gr_do_generic_rays(gr_program* p, void* stream, void* rays, const void* ray_count, int num_rays,
                       void* ray_time_min, void* ray_time_max, const void* cfg, const void* dfg,
                       int width, int height, int mouse_x, int mouse_y,
                       void* ray_write, void* ray_write_counts, int max_write, void* attempt_counter);

/* calculate_singularities, cl.cl:5008-5020; main.cpp:2435 */
int gr_calculate_singularities(gr_program* p, void* stream, const void* finished_rays, const void* finished_count,
                               int num_rays, void* termination_buffer, int width, int height);

/* calculate_render_data, cl.cl:5135-5213; main.cpp:2475, 2509 */
int gr_calculate_render_data(gr_program* p, void* stream, const void* rays, const void* ray_count, int num_rays,
                             void* render_data, void* render_data_count, int width, int height,
                             const void* cfg, const void* dfg);

/* handle_adaptive_sampling, cl.cl:5223-5345; main.cpp:2498 */
int gr_handle_adaptive_sampling(gr_program* p, void* stream, const void* rays, const void* ray_count,
                                void* render_data, void* render_data_count,
                                void* new_rays, void* new_ray_count,
                                const void* camera_generic, const void* camera_quat,
                                const void* e0, const void* e1, const void* e2, const void* e3,
                                int width, int height, const void* cfg, const void* dfg);

/* render, cl.cl:5453-5846; main.cpp:2525.  The image2d_t output becomes a float4[width*height] buffer,
 * each image2d_array_t background becomes RGBA8 texels [levels][bg_height][bg_width] laid out as
 * load_mipped_image does (graphics_settings.cpp:152-212). */
int gr_render(gr_program* p, void* stream, const void* render_data, const void* render_data_count, int num_pixels,
              void* out_rgba_f32, const void* background1, const void* background2,
              int bg_width, int bg_height, int bg_levels,
              int width, int height, int max_probes, const void* cfg, const void* dfg);

/* render restricted to this device's row blocks (block-cyclic rows, see gr_trace_fused); render_data must be
 * pixel-indexed.  compact_out = 1 writes the device's blocks back to back (block i of this device at
 * out + i*block_rows*width float4), which is the layout the multi-GPU gather ships. */
int gr_render_strips(gr_program* p, void* stream, const void* render_data, void* out_rgba_f32,
                     const void* background1, const void* background2, int bg_width, int bg_height, int bg_levels,
                     int width, int height, int block_rows, int strip_rank, int strip_count, int compact_out,
                     int max_probes, const void* cfg, const void* dfg);
/* number of row blocks device `strip_rank` owns */
int gr_strip_local_blocks(int height, int block_rows, int strip_rank, int strip_count);

/* ---- camera riding a timelike geodesic (the snapshot sequence of main.cpp:2675-2760 and the
 *      per-frame interpolation of main.cpp:2265-2297).  Path buffers are step-major: element k of observer id is at
 *      [k*count + id].  basis_speed buffers hold float4 per observer (the reference's float3 has the same 16-byte stride). */

/* boost_tetrad, cl.cl:2441-2481; main.cpp:2700 */
int gr_boost_tetrad(gr_program* p, void* stream, const void* generic_in, int count, const void* basis_speed,
                    void* e0_io, void* e1_io, void* e2_io, void* e3_io, const void* cfg);
/* init_inertial_ray, cl.cl:3117-3141; main.cpp:2722 */
int gr_init_inertial_ray(gr_program* p, void* stream, const void* generic_position_in, int ray_count, void* rays, void* ray_count_out,
                         const void* e0, const void* e1, const void* e2, const void* e3, const void* basis_speed, const void* cfg);
/* get_geodesic_path, cl.cl:4735-4940; main.cpp:2742.  velocities_out / ds_out may be NULL. */
int gr_get_geodesic_path(gr_program* p, void* stream, const void* rays, int num_rays, void* positions_out, void* velocities_out,
                         void* ds_out, const void* ray_count, int max_path_length, const void* cfg, const void* dfg, void* count_out);
/* parallel_transport_quantity, cl.cl:2569-2620; main.cpp:2758 (once per tetrad leg) */
int gr_parallel_transport_quantity(gr_program* p, void* stream, const void* geodesic_path, const void* geodesic_velocity,
                                   const void* ds_in, const void* quantity, const void* count_in, int count, void* quantity_out,
                                   const void* cfg);
/* handle_interpolating_geodesic, cl.cl:2738-2872; main.cpp:2293: camera position + tetrad at proper time target_time */
int gr_handle_interpolating_geodesic(gr_program* p, void* stream, const void* geodesic_path, const void* geodesic_velocity,
                                     const void* ds_in, void* camera_generic_out, const void* t_e0, const void* t_e1,
                                     const void* t_e2, const void* t_e3, void* e0_out, void* e1_out, void* e2_out, void* e3_out,
                                     float target_time, const void* count_in, int parallel_transport_observer,
                                     const void* basis_speed, void* interpolated_velocity, const void* cfg);

/* ---- fused MI355X path (no reference counterpart) ------------------------------------------- */

/* Prepass termination flags from one fused trace at prepass resolution (replaces the sequence
 * clear_termination_buffer / init_rays_generic / do_generic_rays / calculate_singularities,
 * main.cpp:2387-2436). */
int gr_prepass_fused(gr_program* p, void* stream, const void* camera_generic, const void* camera_quat,
                     void* termination_buffer, int prepass_width, int prepass_height,
                     const void* e0, const void* e1, const void* e2, const void* e3,
                     const void* cfg, const void* dfg);
/* the same for a device that owns only the row blocks strip_rank, strip_rank + strip_count, ... of an image of
 * image_height rows (see gr_trace_fused): cells none of its rows can look at are not traced and keep their old value.
 * cell_attempts (unsigned[prepass_width * prepass_height], may be NULL): the step attempts each cell's ray took, the cost
 * estimate gr_order_tiles works from.  row_margin: how many pixel rows beyond its blocks and their halo rows the device also
 * traces from (0; adaptive sampling on a split frame: 2) */
int gr_prepass_fused_strips(gr_program* p, void* stream, const void* camera_generic, const void* camera_quat,
                            void* termination_buffer, int prepass_width, int prepass_height,
                            const void* e0, const void* e1, const void* e2, const void* e3, const void* cfg, const void* dfg,
                            int image_height, int block_rows, int strip_rank, int strip_count, void* cell_attempts, int row_margin);

/* gr_cart_to_generic + gr_init_basis_vectors + gr_prepass_fused_strips in one launch: the camera's metric coordinates and tetrad
 * are computed from the Cartesian camera inside the launch (and stored to position_generic_out / e*_out for gr_trace_fused), then
 * the prepass cells are traced.  prepass_width * prepass_height may be 0: camera set-up only.  Removes two single-lane launches
 * from every frame's chain. */
int gr_camera_prepass(gr_program* p, void* stream, const void* position_cart, float flip, const float basis_speed[3],
                      void* position_generic_out, void* e0_out, void* e1_out, void* e2_out, void* e3_out, const void* camera_quat,
                      void* termination_buffer, int prepass_width, int prepass_height, const void* cfg, const void* dfg,
                      int image_height, int block_rows, int strip_rank, int strip_count, void* cell_attempts, int row_margin);

/* The order in which a persistent gr_trace_fused launch hands out its tiles: longest first, as estimated from what the prepass
 * rays around each tile cost (cell_attempts of gr_prepass_fused_strips / gr_camera_prepass), tiles on the shadow's edge before
 * everything, tiles no pixel of which needs a ray last (gr_trace_fused_launch then writes their records without looking anything up).  A launch lasts as long as its slowest wave, and a long tile drawn late is what
 * makes a wave slow; which wave traces a tile has no influence on the tile's pixels.  tile_order: gr_tile_order_bytes(...) bytes,
 * written by two small launches on `stream`; pass it to gr_trace_fused_ordered with the same image and strip description. */
long long gr_tile_order_bytes(int width, int height, int block_rows, int strip_rank, int strip_count);
int gr_order_tiles(gr_program* p, void* stream, const void* termination_buffer, const void* cell_attempts, int prepass_width,
                   int prepass_height, int width, int height, int block_rows, int strip_rank, int strip_count, void* tile_order);
/* The same list from what the tiles cost in an earlier frame of the same size and strip description (tile_history: the tile_cost a
 * gr_trace_fused_launch left): exact where the prepass rays sample - the long rays near the photon orbits are filaments a pixel or
 * two wide - as long as the camera moves little between the two frames (a tile takes the largest cost among itself and its eight
 * neighbours).  Needs no prepass, so it combines with inline_prepass.  Pass the list with tile_order_by_history = 1.
 * shift_x, shift_y: how far the picture has moved since, in tiles of 8 pixels (0, 0 if unknown): a tile takes the costs of the
 * tiles that far back.  gr_render_frame estimates it from where the two cameras see the coordinate origin. */
int gr_order_tiles_by_history(gr_program* p, void* stream, const void* tile_history, int width, int height, int block_rows,
                              int strip_rank, int strip_count, void* tile_order, int shift_x, int shift_y);

/* Counter block of the fused trace launchers (their `attempt_counter`; NULL = count nothing): GR_COUNTER_WORDS uint64 words on
 * the device, zeroed by the caller.  [0] attempts of the pair / compaction kernels, [1] summed wave lifetimes in shader cycles,
 * [2] the same in ticks of the 100 MHz reference clock, [3] waves, [8..255] probe builds only, [256..511] gr_trace_fused's attempts
 * spread over 256 words by workgroup (one same-address atomic per tile would serialise a frame of many short tiles).  The total
 * is [0] + sum [256..511]; gr_render_state_attempts does that for a frame's own block. */
#define GR_COUNTER_WORDS 512

/* init -> integrate -> render-data in one launch; writes only render_data[sy*width+sx] (32 B per pixel).
 * Rows are dealt to devices block-cyclically: global block b (block_rows rows, multiple of 8) belongs to device
 * b % strip_count; each block additionally traces the one row below it (texture-filter halo).  strip_count <= 1
 * traces the whole image.  termination_buffer may be NULL (no prepass). */
int gr_trace_fused(gr_program* p, void* stream, const void* camera_generic, const void* camera_quat,
                   void* render_data, int width, int height, int block_rows, int strip_rank, int strip_count,
                   const void* termination_buffer, int prepass_width, int prepass_height,
                   const void* e0, const void* e1, const void* e2, const void* e3,
                   const void* cfg, const void* dfg, void* attempt_counter);
/* gr_trace_fused with everything that only schedules it or rides along, by name:
 *   tile_order      gr_order_tiles' list (NULL: image order)
 *   waves_per_simd  1..8: at most that many persistent waves per SIMD (0: as many as the kernel's registers allow)
 *   lattice, pending_only   the two launches of adaptive sampling (gr_trace_fused_adaptive), also on a split frame
 *   shading.out     not NULL: the launch also SHADES the pixels whose two filter neighbours lie in the same 8x8 tile - 49 of every
 *                   64 - from the registers their records were built in (the neighbours' sky coordinates come over by
 *                   ds_bpermute) and writes them to shading.out as gr_render would (compact_out as in gr_render_strips);
 *                   gr_render_seams then shades the last column and row of every tile from the records.  Width and height
 *                   must be multiples of 8, and the program must have been built with -DGR_TILE_SHADING appended to its argument
 *                   string (gr_program_has_tile_shading).  The records are written either way. */
typedef struct gr_trace_shading {
    void* out;                       /* float4 per pixel; NULL = no shading in the trace launch */
    const void* background1;
    const void* background2;
    int bg_width, bg_height, bg_levels, max_probes, compact_out;
} gr_trace_shading;
typedef struct gr_trace_fused_args {
    const void* camera_generic;
    const void* camera_quat;
    void* render_data;
    int width, height, block_rows, strip_rank, strip_count;
    const void* termination_buffer;
    int prepass_width, prepass_height;
    const void *e0, *e1, *e2, *e3, *cfg, *dfg;
    void* attempt_counter;
    const void* tile_order;
    int waves_per_simd;
    gr_trace_shading shading;
    int lattice;        /* 0 or 1: every pixel; 2: the pixels (2x, 2y) only (first launch of adaptive sampling) */
    int pending_only;   /* 1: only the pixels gr_adaptive_refine marked (second launch of adaptive sampling) */
    int inline_prepass; /* 1: the launch traces the prepass grid itself - its cells are the first tickets of the persistent launch, 64
                         * to a wave, and a tile waits for the cells its pixels look at (device-scope flags in termination_buffer,
                         * which must be writable and is reset by the call).  Image order only (no tile_order), lattice 1; a device's
                         * share of a split frame traces the cells its rows look at and leaves the others unknown.  Camera and tetrad must be on the device already (gr_camera_prepass with a 0 x 0
                         * grid).  Records and flags are those of the two-launch sequence.  With tile_order_by_history the tiles
                         * behind the cell waves follow tile_order. */
    void* tile_cost;    /* not NULL: unsigned[number of the device's tiles = (gr_tile_order_bytes - 128) / 8]; the launch leaves what
                         * each tile cost there (the attempts of its longest ray) for gr_order_tiles_by_history.  Every pixel, one ray per lane. */
    int tile_order_by_history;   /* 1: tile_order is gr_order_tiles_by_history's list (its last class is looked up like any tile) */
    void* lattice_rays;          /* lattice = 2: where the launch leaves its rays' end states for gr_adaptive_refine (see there); may be NULL */
} gr_trace_fused_args;
int gr_trace_fused_launch(gr_program* p, void* stream, const gr_trace_fused_args* args);
/* waves of gr_trace_fused the program's device holds at once (what a persistent launch fills it with): a frame of many more tiles than
 * that has a short tail whatever the order of its tiles */
long long gr_trace_fused_wave_slots(gr_program* p);
/* the pixels such a launch leaves to shade: same arguments as gr_render_strips (strip_count <= 1: the whole image) */
int gr_render_seams(gr_program* p, void* stream, const void* render_data, void* out_rgba_f32,
                    const void* background1, const void* background2, int bg_width, int bg_height, int bg_levels,
                    int width, int height, int block_rows, int strip_rank, int strip_count, int compact_out,
                    int max_probes, const void* cfg, const void* dfg);

/* Adaptive sampling on the fused path (the reference: init_rays_generic's packing cl.cl:3234-3250 + handle_adaptive_sampling
 * cl.cl:5223-5345 + a second do_generic_rays / calculate_render_data).  gr_trace_fused_adaptive is gr_trace_fused on a whole image
 * with lattice = 2: only the pixels (2x, 2y) are traced; or with pending_only = 1: only the pixels whose record says terminated ==
 * -1 are traced, every other record is left alone.  gr_adaptive_refine decides per 2x2 block from the lattice records (boundary
 * blocks and blocks whose termination flags differ always refine, otherwise the angular error across the block against the
 * per-pixel angle times adaptive_sampling_threshold): a block to refine gets its three other records marked -1, any other block
 * gets them interpolated; pending_count (device int, may be NULL) accumulates 3 per refined block. */
int gr_trace_fused_adaptive(gr_program* p, void* stream, const void* camera_generic, const void* camera_quat, void* render_data,
                            int width, int height, const void* termination_buffer, int prepass_width, int prepass_height,
                            const void* e0, const void* e1, const void* e2, const void* e3, const void* cfg, const void* dfg,
                            void* attempt_counter, int lattice, int pending_only, void* lattice_rays);
int gr_adaptive_refine(gr_program* p, void* stream, void* render_data, void* pending_count, int width, int height, const void* dfg,
                       const void* lattice_rays, const void* cfg);
/* The second launch as a list (what gr_render_frame does): gr_adaptive_refine_list decides and marks as gr_adaptive_refine_strips does
 * and leaves the marked pixels in pending_list (gr_pending_list_bytes), ordered by what their rays are expected to cost - the dearest of
 * the four lattice rays around the block, an octave of attempts per class, dearest first; gr_trace_pending traces the list 64 entries to
 * a wave (waves_per_simd as in gr_trace_fused_args; 0 = as many as fit).  Every lane of every wave has a ray, the rays of a wave are
 * neighbours of one cost class, and the longest rays of the frame start first.  Records equal those of the pending_only launch to
 * rounding (another kernel around the same device functions). */
size_t gr_pending_list_bytes(int width, int height);
size_t gr_lattice_rays_bytes(int width, int height);
int gr_adaptive_refine_list(gr_program* p, void* stream, void* render_data, void* pending_count, int width, int height, const void* dfg,
                            int block_rows, int strip_rank, int strip_count, const void* lattice_rays, const void* cfg, void* pending_list);
int gr_trace_pending(gr_program* p, void* stream, const void* camera_generic, const void* camera_quat, void* render_data, int width, int height,
                     const void* e0, const void* e1, const void* e2, const void* e3, const void* cfg, const void* dfg, void* attempt_counter,
                     const void* pending_list, int waves_per_simd);
/* lattice_rays: gr_lattice_rays_bytes(width, height) bytes - 3 x float4 per lattice pixel, and behind those one unsigned per lattice
 * pixel: the attempts its ray took (the cost estimate of gr_adaptive_refine_list) - written by the lattice launch (lattice = 2) and
 * read by gr_adaptive_refine: where every lattice ray ended (position, velocity, the quaternion of its rotated frame) - what the
 * reference's decision reads off its ray records through get_intersection_position, also for rays whose render-data record is black
 * (they ended inside r = 1; their texture coordinates are 0, 0; cl.cl:5260-5268).  NULL on both: the decision falls back on the
 * records' texture coordinates, which differs from the reference's around black features.  cfg: the metric's dynamic variables
 * (needed with lattice_rays). */
/* the same on a device's share of a split frame: only the 2x2 pixel blocks whose rows the device shades or reads as a halo row are
 * decided.  The lattice launch before it (gr_trace_fused_launch with lattice = 2 and the strip description) traces the lattice
 * rows those decisions read - two rows of halo either side of a block - and the launch after it (pending_only = 1, same strip
 * description) the marked pixels of the device's rows; the rows equal those of the whole frame sampled adaptively. */
int gr_adaptive_refine_strips(gr_program* p, void* stream, void* render_data, void* pending_count, int width, int height,
                              const void* dfg, int block_rows, int strip_rank, int strip_count, const void* lattice_rays, const void* cfg);

/* gr_trace_fused with two rays per lane: a wave takes two neighbouring 8x8 tiles and every lane integrates one pixel of each,
 * all per-ray arithmetic in packed fp32 (v_pk_fma/mul/add_f32: one instruction, two rays).  Same arguments, same records;
 * each ray's arithmetic is that of gr_trace_fused (results agree to what the compiler contracts differently).  A program has
 * the kernel when its Verlet-loop expressions instantiate on float pairs (no comparison/select forms, moderate size) and it
 * steps without the adaptive controller - there it is 1.3-1.5x faster; with the controller it is slower and only built on
 * request (GR_TRACE_PAIR_BUILD=1).  gr_program_has_trace_pair tells; GR_ERROR_INVALID_ARGUMENT when it is missing. */
int gr_trace_pair(gr_program* p, void* stream, const void* camera_generic, const void* camera_quat,
                  void* render_data, int width, int height, int block_rows, int strip_rank, int strip_count,
                  const void* termination_buffer, int prepass_width, int prepass_height,
                  const void* e0, const void* e1, const void* e2, const void* e3,
                  const void* cfg, const void* dfg, void* attempt_counter);
int gr_program_has_trace_pair(const gr_program* p);   /* 1 / 0 */
/* 1 when the program's argument string (or GR_EXTRA_FLAGS) carried -DGR_TILE_SHADING: its gr_trace_fused can shade (gr_trace_shading) */
int gr_program_has_tile_shading(const gr_program* p);
/* Process-unique identity of a program object (never reused, unlike its address); 0 for NULL. */
unsigned long long gr_program_serial(const gr_program* p);
/* What the program's code object was built from - kernel source, every compile option, hiprtc version, the setting of the pass
 * over the compiled code - as 16 hex digits (the name of its cache file), followed by what came out: "-v<VGPRs>s<scratch bytes>"
 * of gr_trace_fused as loaded (the build-time occupancy rule can go either way for one set of inputs).  Measurements that belong
 * to one build (hardware counters under profiles/) carry it, so that a reader can tell whether they still describe the kernel
 * that runs. */
const char* gr_program_build_key(const gr_program* p);

/* gr_trace_fused with ray compaction (north_star: "wave-level ballots for step-acceptance and ray compaction"): persistent
 * waves hold one ray per lane; whenever fewer than keep_lanes (1..64) of a wave's rays are still integrating, the finished
 * ones are written out and the idle lanes draw new pixels from a device-side counter.  Every ray is integrated exactly as
 * in gr_trace_fused (results agree to rounding); it can only pay when neighbouring rays need very different numbers of
 * steps - not the case for the BASELINE workloads, where 8x8 tiles keep >= 94 % of the lanes busy. */
int gr_trace_compact(gr_program* p, void* stream, const void* camera_generic, const void* camera_quat,
                     void* render_data, int width, int height, int block_rows, int strip_rank, int strip_count,
                     const void* termination_buffer, int prepass_width, int prepass_height,
                     const void* e0, const void* e1, const void* e2, const void* e3,
                     const void* cfg, const void* dfg, void* attempt_counter, int keep_lanes);

/* ---- frame driver (the enqueue sequence of main.cpp:2244-2526) ------------------------------- */

typedef struct gr_render_state gr_render_state;   /* render_state.hpp:97-197: all per-frame device buffers */

/* camera, main.cpp:664-673: Cartesian (t,x,y,z) position, orientation quaternion (x,y,z,w) */
typedef struct gr_camera gr_camera;
struct gr_camera {
    float position[4];
    float quat[4];
    float basis_speed[3];   /* cartesian_basis_speed, main.cpp:2320-2327 */
    float flip;             /* flip_sign > 0 puts the camera on the far side (negative r) */
};
void gr_camera_default(gr_camera* out);   /* pos (0,0,-4,0), axis-angle (1,0,0,-pi/2) */

enum { GR_MODE_REFERENCE = 0,   /* one launch per reference kernel, 96-byte ray records in HBM */
       GR_MODE_FUSED = 1 };     /* gr_prepass_fused + gr_trace_fused + gr_render */

/* Snapshot of the camera's own timelike geodesic, resident on the device: the buffers of main.cpp:1232-1242
 * (geodesic_trace / vel / ds / count) and the four parallel-transported tetrad legs. */
typedef struct gr_geodesic_camera gr_geodesic_camera;

typedef struct gr_frame_options {
    int mode;              /* GR_MODE_* */
    int tiled;             /* reference mode only: 8x8-tile ray order (ignored when adaptive sampling is on) */
    int use_prepass;       /* -1: per metric config (metric_cfg.use_prepass), 0 / 1 force it off / on for this frame;
                            * -2: per metric config and - whole frames on the fused path - only while it pays: when a frame's prepass
                            * lets the trace skip fewer than 2 % of the pixels (cells with their whole 5-point stencil marked), the
                            * next 30 frames of this render state go without one, then it is tried again
                            * (gr_render_state_prepass_policy reports).  A pixel the prepass would have skipped is then traced on its
                            * own, which in chaotic regions (naked singularities) need not come out black as the reference's does. */
    int max_probes;        /* anisotropy, graphics_settings.hpp:34 (8) */
    int strip_rank;        /* fused mode, multi-GPU: image rows are dealt in blocks of block_rows rows,          */
    int strip_count;       /*   global block b belongs to device b % strip_count (1 = whole image on this device) */
    int block_rows;        /*   multiple of 8                                                                      */
    int compact_out;       /*   1: write this device's blocks back to back into out (gather layout)               */
    int time_kernels;      /* 1: record HIP events around every stage of this frame (gr_render_state_stage_ms);
                            * 2: log one event pair per trace launch until gr_render_state_trace_log collects them */
    int count_attempts;    /* accumulate Verlet step attempts (gr_render_state_attempts) */
    const struct gr_camera* next_camera;   /* fused mode, optional: the camera of the NEXT gr_render_frame call.  Its tetrad and
                            * prepass are then computed on a second stream while this frame traces, and the next call
                            * (same program / cfg / features / camera) skips them.  NULL = no look-ahead. */
    const gr_geodesic_camera* geodesic;   /* camera_on_geodesic (main.cpp:2264-2293): position and tetrad come from this snapshot
                            * at proper time geodesic_time instead of camera->position; camera->quat still orients the view */
    float geodesic_time;          /* current_geodesic_time of this frame */
    float next_geodesic_time;     /* ... of the next frame, used with next_camera (look-ahead) */
    int parallel_transport_observer;   /* 1 (default, main.cpp:1259): interpolate the transported tetrads; 0: rebuild them */
    int ray_compaction;    /* fused mode: 0 = gr_trace_fused (one tile per wave at a time), 1..64 = gr_trace_compact with this
                            * keep_lanes; -1 = library default */
    const struct gr_camera* next_camera2;   /* optional: the camera of the call AFTER next_camera's.  Two prepasses are then in
                            * flight on two streams, which hides their latency even when a frame traces faster than one
                            * prepass runs (row-split frames on several GPUs). */
    float next_geodesic_time2;
    int next_strip_rank;   /* strip_rank of the next_camera / next_camera2 frames when a device's share of the image rotates from */
    int next_strip_rank2;  /*   frame to frame (load balance over ranks); -1 = the same as this frame's */
    int rays_per_lane;     /* fused mode without compaction: 1 = gr_trace_fused, 2 = gr_trace_pair (error if the program lacks it),
                            * 0 = library default: 2 where the program has the pair kernel, else 1 (GR_TRACE_RAYS_PER_LANE=1|2 overrides) */
    int fused_shading;     /* fused mode: 1 = the trace launch shades the 49 of every 64 pixels whose filter neighbours are in the same
                            * tile and gr_render_seams the rest (needs a program built with -DGR_TILE_SHADING, width and height
                            * multiples of 8, one ray per lane, no compaction, no adaptive sampling; an error otherwise);
                            * 0 or -1 (library default) = gr_render shades every pixel.  Off by
                            * default on measurement: 4K Kerr, three frames in flight, 1 723 against 1 743 Mrays/s - the shading
                            * arithmetic moves into the trace launch, and the separate pass was already hidden behind the next
                            * frame's trace */
    int inline_prepass;    /* fused mode, a frame whose prepass was not computed ahead (next_camera): trace the prepass grid inside
                            * the trace launch (gr_trace_fused_args.inline_prepass; one ray per lane, no adaptive sampling, no
                            * compaction).  -1 (default): on whole frames that do not order their tiles; 1: also on a device's share
                            * of a split frame (which then is not ordered - measured slower there); 0: the prepass as a launch of
                            * its own in front */
    int trace_waves_per_simd;   /* fused mode: persistent waves per SIMD a trace launch takes, 1..8; 0 = as many as fit (best for
                            * one frame at a time).  With three or more frames in flight on streams of their own, 4 measured
                            * 2-3 % faster than all: the launches then share the device instead of queueing for it. */
    int tile_history;      /* fused mode, one ray per lane: 1 = hand the tiles of this frame out dearest first by what they cost in this
                            * render state's previous frame (gr_order_tiles_by_history, shifted by how far the camera has moved the
                            * picture since; the first frame, one of another strip description, one whose camera has moved the picture
                            * by more than 48 px or rides a geodesic goes in image order); every such frame records its tiles' costs.
                            * Scheduling only: the pixels do not depend on it.  0 = no; -1 = library default: whole frames of at most
                            * 32 tiles per wave slot that find no frame of ANOTHER stream still running on the device when they are
                            * submitted (frames in flight on several streams fill each other's tails, and the order measured slower
                            * there; frames queued on one stream run one after the other and do follow it) */
} gr_frame_options;
void gr_frame_options_default(gr_frame_options* out);
/* What tile_history did with this render state's frames so far: how many recorded their tiles' costs, how many of those followed
 * the costs of the frame before, and the shift (in tiles) the last one that did applied.  Any pointer may be NULL. */
int gr_render_state_tile_history(gr_render_state* s, unsigned long long* frames_recorded, unsigned long long* frames_followed,
                                 int last_shift[2]);
/* The two estimates tile_history works with (host arithmetic, no device).  gr_camera_origin_on_screen: the pixel at which the
 * camera sees the coordinate origin as if space were flat - the inverse of the kernels' pixel -> direction map (cl.cl:2015-2059) -
 * 1 and pixel_out[0..1] = (x, y), or 0 when the origin is behind the camera or the camera sits on it.  gr_picture_motion: an upper
 * estimate of how many pixels the picture moves between two cameras (angle between the orientations + parallax of the origin, at the
 * focal length); 1e9 when flip or observer speed differ. */
int gr_camera_origin_on_screen(const gr_camera* camera, float field_of_view, int width, int height, float pixel_out[2]);
float gr_picture_motion(const gr_camera* from, const gr_camera* to, float field_of_view, int width);

int gr_render_state_create(int device, int width, int height, gr_render_state** out);
void gr_render_state_destroy(gr_render_state* s);

/* Renders one frame into out_rgba_f32 (float4[width*height], device memory; in fused strip mode only
 * rows [row_begin,row_end) are written).  cfg_values = the $cfg parameters (NULL = metric defaults). */
int gr_render_frame(gr_render_state* s, gr_program* p, const gr_metric* m, void* stream,
                    const gr_camera* camera, const gr_features* features,
                    const float* cfg_values, int num_cfg_values,
                    const void* background1, const void* background2, int bg_width, int bg_height, int bg_levels,
                    void* out_rgba_f32, const gr_frame_options* options);

/* Camera on a timelike geodesic: object form of main.cpp:2675-2760.  gr_geodesic_camera_snapshot launches
 * cart_to_generic, init_basis_vectors (camera->basis_speed), boost_tetrad, init_inertial_ray, get_geodesic_path and four
 * parallel_transport_quantity on `stream`, then synchronises once to report the number of samples and the proper time the
 * path covers.  geodesic_basis_speed is g_geodesic_basis_speed (main.cpp:2253-2261), |v| < 1. */
int gr_geodesic_camera_create(int device, int max_path_length, gr_geodesic_camera** out);
void gr_geodesic_camera_destroy(gr_geodesic_camera* g);
int gr_geodesic_camera_snapshot(gr_geodesic_camera* g, gr_program* p, const gr_metric* m, void* stream, const gr_camera* camera,
                                const float geodesic_basis_speed[3], const gr_features* features, const float* cfg_values,
                                int num_cfg_values, int* steps_out, float* proper_time_out);
/* handle_interpolating_geodesic + read-back (the reference's geodesic_q / camera_q async reads, main.cpp:2295-2296):
 * generic camera position, tetrad (4 rows of 4) and 4-velocity at `proper_time`; any output may be NULL */
int gr_geodesic_camera_interpolate(gr_geodesic_camera* g, gr_program* p, void* stream, float proper_time,
                                   int parallel_transport_observer, float camera_generic_out[4], float tetrad_out[16],
                                   float velocity_out[4]);
enum { GR_GEOBUF_PATH = 0, GR_GEOBUF_VELOCITY = 1, GR_GEOBUF_DS = 2, GR_GEOBUF_COUNT = 3, GR_GEOBUF_TRANSPORTED0 = 4,
       GR_GEOBUF_TRANSPORTED1 = 5, GR_GEOBUF_TRANSPORTED2 = 6, GR_GEOBUF_TRANSPORTED3 = 7 };
void* gr_geodesic_camera_buffer(gr_geodesic_camera* g, int which);

/* what the prepass policy (gr_frame_options.use_prepass = -2) has done with this state's frames so far, and the fraction of
 * the prepass grid the last inspected prepass made skippable (-1: none inspected yet); any output may be NULL */
int gr_render_state_prepass_policy(gr_render_state* s, unsigned long long* frames_with_prepass, unsigned long long* frames_without,
                                   float* last_marked_fraction);

/* stages for timing / buffer access */
enum { GR_STAGE_CAMERA = 0, GR_STAGE_PREPASS = 1, GR_STAGE_INIT = 2, GR_STAGE_TRACE = 3, GR_STAGE_RENDER_DATA = 4,
       GR_STAGE_ADAPTIVE = 5, GR_STAGE_RENDER = 6, GR_STAGE_COUNT = 7 };
/* elapsed milliseconds of a stage of the last timed frame (synchronises on the stage's stop event) */
int gr_render_state_stage_ms(gr_render_state* s, int stage, float* ms);
/* sum of the durations and number of the trace launches (gr_trace_fused / gr_do_generic_rays) logged with time_kernels = 2
 * since the last reset; waits for the logged launches to finish */
int gr_render_state_trace_log(gr_render_state* s, float* total_ms, int* launches, int reset);
/* total Verlet step attempts of the last frame rendered with count_attempts (synchronises the device) */
int gr_render_state_attempts(gr_render_state* s, unsigned long long* attempts);
/* average shader clock (MHz) the fused trace kernel of that frame ran at: wave lifetimes in shader cycles (s_memtime) over the
 * same lifetimes in ticks of the constant 100 MHz reference clock (s_memrealtime); 0 when the frame was not traced by
 * gr_trace_fused with count_attempts (synchronises the device) */
int gr_render_state_shader_clock(gr_render_state* s, double* mhz);
/* of the same launch: the summed lifetime of its waves in milliseconds and how many waves ran.  Over (wave slots the launch
 * held) x (launch duration) this is the share of the slots that was occupied - the rest is the launch's ramp and tail */
int gr_render_state_wave_time(gr_render_state* s, double* wave_ms, unsigned long long* waves);
/* the first count (<= 256) words of that frame's counter block as the kernels left them: [0] attempts, [1] shader cycles,
 * [2] reference-clock ticks, [3] waves, [8..255] only written by probe builds of the kernels (tools/README.md) */
int gr_render_state_counters(gr_render_state* s, unsigned long long* words, int count);

enum { GR_BUF_RAYS_IN = 0, GR_BUF_RAYS_COUNT = 1, GR_BUF_RENDER_DATA = 2, GR_BUF_TERMINATION = 3, GR_BUF_CAMERA_GENERIC = 4,
       GR_BUF_TETRAD0 = 5, GR_BUF_TETRAD1 = 6, GR_BUF_TETRAD2 = 7, GR_BUF_TETRAD3 = 8, GR_BUF_RAYS_ADAPTIVE = 9,
       GR_BUF_RAYS_ADAPTIVE_COUNT = 10, GR_BUF_CFG = 11, GR_BUF_DFG = 12, GR_BUF_CAMERA_QUAT = 13 };
/* device pointer of one of the state's buffers (NULL if not allocated) */
void* gr_render_state_buffer(gr_render_state* s, int which);
/* blocking copies for tests and tools */
int gr_device_download(int device, void* host_dst, const void* device_src, size_t bytes);
int gr_device_upload(int device, void* device_dst, const void* host_src, size_t bytes);
int gr_device_alloc(int device, size_t bytes, void** out);
int gr_device_free(int device, void* ptr);
int gr_device_synchronize(int device);
/* HIP streams of the library's own runtime (the role of the reference's cl::command_queue objects, main.cpp:1458-1461): one per
 * frame in flight.  A caller that already has hipStream_t handles from the same runtime can pass those instead. */
int gr_stream_create(int device, int high_priority, void** stream_out);
int gr_stream_synchronize(void* stream);
int gr_stream_destroy(void* stream);
int gr_device_count(int* count);

/* ---- one frame over several GPUs (the reference is single-GPU) ------------------------------------------
 * Image rows are dealt to `world` participants in blocks of block_rows rows, block-cyclically; each renders its share with
 * gr_render_frame's strip mode (own prepass cells, one halo row per block, no exchange while tracing) and the finished float4 rows
 * go to participant 0 - every block straight to its final place in its frame buffer, nothing is staged or un-permuted there.
 *   GR_TRANSPORT_RCCL: one process per GPU.  Participant 0 calls gr_tiled_unique_id and hands the 128 bytes to the others by any
 *     means; everybody then calls gr_tiled_create (collective: returns when all have).  Per frame and block: ncclSend on the
 *     owner / ncclRecv on participant 0 at the block's row offset, one group per frame, enqueued on the caller's stream.
 *     librccl is loaded at run time (dlopen), the library has no link dependency on it.
 *   GR_TRANSPORT_PEER: one process driving `count` devices (gr_tiled_create_local; devices may repeat): hipMemcpyPeerAsync per
 *     block on the owner's stream; gr_tiled_join makes participant 0's stream wait for every frame issued so far.
 *   GR_TRANSPORT_CUSTOM: the caller's point-to-point library behind a gr_transport table (gr_tiled_create_custom); RCCL's call
 *     pattern with the caller's send / recv: per frame one group, on the owner one send per block in block order, on participant
 *     0 the matching receives peer by peer.  With device < 0 no device is touched at all: gr_tiled_exchange then runs the
 *     schedule on host memory (how tests/test_distributed_cpu.py checks order and offsets for every rank of a world).
 * Frames in flight: each participant stages its rows in a ring of GR_TILED_STAGING (default 4) buffers, one per frame in flight;
 * gr_render_frame_tiled may be called for frame k+1 on another stream, with another rotation, while frame k's transfers run (a
 * frame that finds its ring slot still in use makes its stream wait for that frame's transfers).
 * gr_render_frame_tiled = gr_render_frame for this participant's share + the transfer.  `options` as for gr_render_frame (mode,
 * strip_* and compact_out are overridden; next_camera / next_strip_rank look-ahead works as there, see gr_tiled_share).
 * `rotation`: participant r renders share (r + rotation) % world - rotate with the frame number to even out shares of different
 * cost.  frame_on_root: float4[height * width] on participant 0's device; NULL elsewhere with RCCL, the same pointer for every
 * participant with peer copies. */
/*   GR_TRANSPORT_IPC: one process per participant like RCCL, but the participants may share a device (RCCL refuses that): the same
 *     group / send / receive calls in the same order, the blocks copied device to device through inter-process memory handles, the
 *     matching done in a POSIX shared-memory mailbox named after `session`.  Host-blocking at the end of a group: a rehearsal stage
 *     for boxes with fewer GPUs than ranks (tests, bench.py's dry run), not a product path.  gr_tiled_create_ipc is collective. */
typedef struct gr_tiled gr_tiled;
enum { GR_TRANSPORT_RCCL = 0, GR_TRANSPORT_PEER = 1, GR_TRANSPORT_CUSTOM = 2, GR_TRANSPORT_IPC = 3 };
/* the point-to-point calls a split frame needs (the subset of RCCL it uses); every function returns GR_OK or an error code that
 * gr_render_frame_tiled / gr_tiled_exchange hand back.  group_begin / group_end may be NULL. */
typedef struct gr_transport {
    void* user;
    int (*group_begin)(void* user);
    int (*group_end)(void* user);
    int (*send)(void* user, const void* data, size_t float_count, int peer, void* stream);
    int (*recv)(void* user, void* data, size_t float_count, int peer, void* stream);
} gr_transport;
int gr_tiled_unique_id(void* id_out_128_bytes);
int gr_tiled_create(int world, int rank, int device, const void* unique_id_128_bytes, int width, int height, int block_rows, gr_tiled** out);
int gr_tiled_create_custom(int world, int rank, int device, const gr_transport* transport, int width, int height, int block_rows, gr_tiled** out);
int gr_tiled_create_local(int count, const int* devices, int width, int height, int block_rows, gr_tiled** out_array);
int gr_tiled_create_ipc(int world, int rank, int device, const char* session, int width, int height, int block_rows, gr_tiled** out);
void gr_tiled_destroy(gr_tiled* t);
int gr_render_frame_tiled(gr_tiled* t, gr_render_state* s, gr_program* p, const gr_metric* m, void* stream, const gr_camera* camera,
                          const gr_features* features, const float* cfg_values, int num_cfg_values,
                          const void* background1, const void* background2, int bg_width, int bg_height, int bg_levels,
                          void* frame_on_root, const gr_frame_options* options, int rotation);
int gr_tiled_join(gr_tiled* root, void* stream);
/* the transfer step of gr_render_frame_tiled on its own: `staging` = this participant's compact rows (blocks back to back,
 * gr_tiled_staging_bytes; unused on participant 0), frame_on_root as there */
int gr_tiled_exchange(gr_tiled* t, const void* staging, void* frame_on_root, int rotation, void* stream);
size_t gr_tiled_staging_bytes(const gr_tiled* t);
/* the share participant t renders in a frame with this rotation (what to put into options->next_strip_rank for a look-ahead) */
int gr_tiled_share(const gr_tiled* t, int rotation);
/* rows [row_begin, row_end) of the local_block-th block of a share; returns 1, 0 for a padding block past the image, -1 on bad
 * arguments.  Pure arithmetic: global block = local_block * world + share. */
int gr_tiled_block_rows(int height, int block_rows, int world, int share, int local_block, int* row_begin, int* row_end);
int gr_tiled_block_rows_of(const gr_tiled* t, int share, int local_block, int* row_begin, int* row_end);

/* ---- host helper: background image ----------------------------------------------------------- */

/* load_mipped_image (graphics_settings.cpp:152-212): packs an RGBA8 image and its box-filtered mip
 * chain into `levels` same-size slices (mip i in the top-left corner of slice i, edge replicated).
 * Returns the number of levels; `out` needs levels*width*height*4 bytes (call with out=NULL to query). */
int gr_pack_mipped_background(const unsigned char* rgba, int width, int height, unsigned char* out);

/* ---- host helpers: PNG in/out (headless counterpart of the screenshot path main.cpp:2762-2808 and of the
 *      sf::Image background loader graphics_settings.cpp:214-243) ------------------------------------------- */

/* clamp -> linear-to-sRGB -> clamp -> 8 bit, as the reference's screenshot loop does (main.cpp:2791-2800) */
int gr_frame_to_rgba8(const float* frame_rgba_f32, int width, int height, unsigned char* out_rgba8);
int gr_write_frame_png(const char* path, const float* frame_rgba_f32, int width, int height);
int gr_write_png_rgba8(const char* path, const unsigned char* rgba, int width, int height);
/* 8-bit non-interlaced PNG -> RGBA8; call with out = NULL to query the size */
int gr_read_png_rgba8(const char* path, int* width, int* height, unsigned char* out, size_t capacity);

#ifdef __cplusplus
}
#endif
#endif
