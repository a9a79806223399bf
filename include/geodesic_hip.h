/* geodesic_hip.h - C ABI of libgeodesic_hip.so, the MI355X (gfx950) implementation of the per-pixel geodesic ray pipeline of
 * 20k/geodesic_raytracing.  The reference has no C function API for this path; its boundary is
 *   (1) the compile-time macro string: metrics::build_argument_string (metric.hpp:725-959) + dynamic_feature_config::generate_{dynamic,
 *       static}_argument_string (dynamic_feature_config.cpp:122-180), consumed by cl::build_program_with_cache (metric_manager.hpp:88-108);
 *   (2) launches by kernel name: cl::command_queue::exec(name, args, global, local), main.cpp:203, 2311, 2329, 2396 ... 2525.
 * Every entry point replaces one of those and cites it.  Device buffers are plain device pointers owned by the caller (the reference's
 * cl::buffer ownership, render_state.hpp:172-196); kernels never allocate.  `stream` is a hipStream_t passed as void* (NULL = default
 * stream); launches are asynchronous on it, like the reference's in-order queue (main.cpp:1460).  Every function returns 0 on success
 * and a negative gr_status otherwise; gr_last_error() returns a thread-local message.
 * This header is the CONTRACT: layouts, metric -> macro string, program, one launcher per reference kernel, the frame driver, one frame
 * over several GPUs, image helpers.  The fused MI355X launchers, their schedules and every measurement hook: geodesic_hip_internal.h. */
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
    float position[4], velocity[4], initial_quat[4], acceleration[4];
    float ku_uobsu, running_dlambda_dnew;
    int terminated;            /* 0 = lost/absorbed, 1 = reached a boundary, 2 = skipped by the prepass */
    int sx, sy, pad_[3];
} gr_lightray;

/* struct render_data, cl.cl:5066-5074 / render_state.hpp:21-29; 32 bytes */
typedef struct gr_render_data {
    float tex_coord[2], z_shift;
    int sx, sy, terminated, side, pad_;
} gr_render_data;

/* struct dynamic_feature_config as packed by dynamic_feature_config::alloc_and_write_gpu_buffer
 * (dynamic_feature_config.cpp:182-237): floats in alphabetical order, then bools as int. 48 bytes.
 * Defaults: main.cpp:1123-1158. */
typedef struct gr_features {
    float adaptive_sampling_threshold, field_of_view, max_acceleration_change, max_precision_radius, min_step, ray_skip, universe_size;
    int adaptive_sampling, redshift, reparameterisation, use_old_redshift, use_triangle_rendering;
} gr_features;

void gr_features_default(gr_features* out);

/* ---- host side: metric -> macro string ----------------------------------------------------- */
typedef struct gr_metric gr_metric;

/* per-metric settings that steer the frame driver (metrics::metric_config, metric.hpp:330-357) */
typedef struct gr_metric_info {
    int is_big, is_constant_theta;   /* GENERIC_BIG_METRIC, GENERIC_CONSTANT_THETA */
    int use_prepass, adaptive_precision;
    float max_acceleration_change;
    int num_dynamic_vars;            /* $cfg.NAME parameters */
    int accel_ops, accel_transcendentals, coord_ops;   /* DAG op counts of GEO_ACCEL0..3 and of TO_COORDn + DISTANCE_FUNC (roofline accounting) */
} gr_metric_info;

/* One of the built-in metrics: "minkowski", "schwarzschild", "kerr_boyer", "alcubierre". */
int gr_metric_builtin(const char* name, gr_metric** out);

/* Loads <scripts_dir>/<name>.json (+ one level of inherit_settings) and the scripts it names, exactly
 * as content_manager.cpp:9-112 does; the script dialect is the reference's (js_interop.cpp:665-959). */
int gr_metric_load_script(const char* scripts_dir, const char* name, gr_metric** out);

/* A metric that is only what gr_render_frame reads off one (gr_metric_info, the $cfg names and defaults), for a caller that already
 * holds the argument strings (a program cache, a fixture); gr_metric_argument_string fails on it.  var_names may be NULL. */
int gr_metric_from_info(const gr_metric_info* info, const char* const* var_names, const float* var_defaults, gr_metric** out);

void gr_metric_destroy(gr_metric* m);
int gr_metric_get_info(const gr_metric* m, gr_metric_info* out);
const char* gr_metric_dynamic_var_name(const gr_metric* m, int index);
float gr_metric_dynamic_var_default(const gr_metric* m, int index);

/* metrics::build_argument_string (metric.hpp:725-959).  is_static = 0: "dynamic" program - expressions read cfg->NAME, features come
 * from the feature struct (KERNEL_IS_DYNAMIC); 1: "substituted" program - cfg_values (NULL = defaults) and `features` baked in as
 * literals (KERNEL_IS_STATIC, metric_manager.hpp:153-166).  *needed = capacity required incl. the NUL; buffer = NULL, capacity = 0 asks. */
int gr_metric_argument_string(const gr_metric* m, const gr_features* features, int is_static, const float* cfg_values, int
                              num_cfg_values, char* buffer, size_t capacity, size_t* needed);

/* The metric's generated expressions evaluated on the HOST at one point, in double: the symbolic graph the macro strings are printed from,
 * interpreted - no device, no compiler (BASELINE configs[0], SURVEY.md 7.2c: "CPU evaluator of generated metric code").  For a host that
 * wants to look at a metric; not a rendering path.  position = chart coordinates (v1..v4), velocity = their d/dlambda (iv1..iv4; NULL
 * where unused), cfg_values NULL = defaults.  Writes gr_metric_evaluate_count(what) doubles: 16 g_ij row-major | 64 d g_ij / d x^k as
 * [k][i][j] | 4 accelerations -Gamma^i_jk v^j v^k (GEO_ACCEL0..3) | 4 polar coordinates of the chart point (TO_COORDn) | 4 chart
 * coordinates of the polar point in `position` (FROM_COORDn) | 1 DISTANCE_FUNC of the chart point. */
enum { GR_EVAL_METRIC_TENSOR = 0, GR_EVAL_METRIC_DERIVATIVES = 1, GR_EVAL_ACCELERATION = 2, GR_EVAL_TO_POLAR = 3, GR_EVAL_FROM_POLAR = 4,
       GR_EVAL_ORIGIN_DISTANCE = 5 };
int gr_metric_evaluate_count(int what);
int gr_metric_evaluate(const gr_metric* m, int what, const double position[4], const double velocity[4], const float* cfg_values,
                       int num_cfg_values, double* out, int out_count);

/* ---- device program ------------------------------------------------------------------------- */
typedef struct gr_program gr_program;

/* cl::build_program_with_cache({"cl.cl"}, argument_string) (metric_manager.hpp:88-108): compiles the kernels for gfx950 specialised
 * by `argument_string` (the macro set above; macros of dead device code - CART_TO_POLn, FIX_LIGHTn, METRIC_TIME_G00 - are accepted
 * and ignored; -cl-fp32-correctly-rounded-divide-sqrt, OpenCL's own switch, is honoured) and loads them on HIP device `device`.
 * Code objects are cached on disk keyed by a hash of source + arguments. */
int gr_program_create(const char* argument_string, int device, gr_program** out);
/* A program is usable as soon as the kernels a fused frame launches are there; the kernels of the reference-shaped sequence are a second
 * code object that - when it is not in the cache yet - is still being built when gr_program_create returns, and the first launcher that
 * needs one of them waits for it.  gr_program_complete waits now (and reports that build's error, if any). */
int gr_program_complete(gr_program* p);

/* Compile only (no device needed): fills the on-disk cache; used by the build step. */
int gr_program_precompile(const char* argument_string);

void gr_program_destroy(gr_program* p);

/* metric_manager (metric_manager.hpp:19-219) as an object.  _create builds the dynamic program of `m` (blocking, as the reference does
 * for a newly selected metric) and starts the substituted build for `features` / `cfg_values` (NULL = defaults) on a worker thread.
 * _current is check_substitution: once per frame, swaps the substituted program in when its build has finished (wait != 0: waits) and
 * hands back the program to launch - owned by the manager, valid until two later updates have retired it.  _update is the soft
 * recompile: other values put the dynamic program back at once and start the new build (an overtaken build is not waited for); equal
 * values change nothing.  The metric must outlive the manager.  One thread at a time. */
typedef struct gr_program_manager gr_program_manager;
int gr_program_manager_create(const gr_metric* m, int device, const gr_features* features, const float* cfg_values, int
                              num_cfg_values, gr_program_manager** out);
int gr_program_manager_update(gr_program_manager* pm, const gr_features* features, const float* cfg_values, int num_cfg_values);
int gr_program_manager_current(gr_program_manager* pm, int wait, gr_program** program, int* is_substituted);
gr_program* gr_program_manager_dynamic(gr_program_manager* pm);
void gr_program_manager_destroy(gr_program_manager* pm);

/* ---- launchers: one per reference kernel, reference argument order -------------------------- */
/* cart_to_generic_kernel, cl.cl:6018-6034; launched {1}/{1} at main.cpp:2311 */
int gr_cart_to_generic(gr_program* p, void* stream, const void* position_cart_in, void* position_generic_out, int count, float flip,
                       const void* cfg);
/* init_basis_vectors, cl.cl:2483-2507; main.cpp:2329.  cartesian_basis_speed is a float3 there. */
int gr_init_basis_vectors(gr_program* p, void* stream, const void* generic_in, int count, const float cartesian_basis_speed[3],
                          void* e0_out, void* e1_out, void* e2_out, void* e3_out, const void* cfg);
/* clear_termination_buffer, cl.cl:4997-5006; main.cpp:2396 */
int gr_clear_termination_buffer(gr_program* p, void* stream, void* termination_buffer, int width, int height);
/* init_rays_generic, cl.cl:3143-3251; main.cpp:2422 (prepass) and :2461.  `tiled` (extension, 0 = reference behaviour): ray slots
 * laid out in 8x8 pixel tiles; the ray buffer must then hold gr_tiled_slot_count(width, height) rays. */
int gr_init_rays_generic(gr_program* p, void* stream, const void* camera_generic, const void* camera_quat, void* rays, void*
                         ray_count, int width, int height, const void* termination_buffer, int prepass_width, int prepass_height,
                         int flip_geodesic_direction, const void* e0, const void* e1, const void* e2, const void* e3, const void*
                         cfg, const void* dfg, int i_am_prepass, int tiled);
int gr_tiled_slot_count(int width, int height);
/* do_generic_rays, cl.cl:3954-4247; execute_kernel main.cpp:139-205.  `num_rays` sizes the grid (the reference launches width*height
 * work-items); the device-side count is still honoured.  ray_time_min/max, ray_write, mouse_x/y: signature parity (triangle path,
 * unused).  attempt_counter (extension, may be NULL): device uint64 accumulating Verlet step attempts. */
int gr_do_generic_rays(gr_program* p, void* stream, void* rays, const void* ray_count, int num_rays, void* ray_time_min, void*
                       ray_time_max, const void* cfg, const void* dfg, int width, int height, int mouse_x, int mouse_y, void*
                       ray_write, void* ray_write_counts, int max_write, void* attempt_counter);
/* calculate_singularities, cl.cl:5008-5020; main.cpp:2435 */
int gr_calculate_singularities(gr_program* p, void* stream, const void* finished_rays, const void* finished_count, int num_rays,
                               void* termination_buffer, int width, int height);
/* calculate_render_data, cl.cl:5135-5213; main.cpp:2475, 2509 */
int gr_calculate_render_data(gr_program* p, void* stream, const void* rays, const void* ray_count, int num_rays, void* render_data,
                             void* render_data_count, int width, int height, const void* cfg, const void* dfg);
/* handle_adaptive_sampling, cl.cl:5223-5345; main.cpp:2498 */
int gr_handle_adaptive_sampling(gr_program* p, void* stream, const void* rays, const void* ray_count, void* render_data, void*
                                render_data_count, void* new_rays, void* new_ray_count, const void* camera_generic, const void*
                                camera_quat, const void* e0, const void* e1, const void* e2, const void* e3, int width, int height,
                                const void* cfg, const void* dfg);
/* render, cl.cl:5453-5846; main.cpp:2525.  The image2d_t output is a float4[width*height] buffer, each image2d_array_t background
 * RGBA8 texels [levels][bg_height][bg_width] laid out as load_mipped_image does (graphics_settings.cpp:152-212). */
int gr_render(gr_program* p, void* stream, const void* render_data, const void* render_data_count, int num_pixels, void*
              out_rgba_f32, const void* background1, const void* background2, int bg_width, int bg_height, int bg_levels, int width,
              int height, int max_probes, const void* cfg, const void* dfg);
/* ---- camera riding a timelike geodesic (the snapshot sequence of main.cpp:2675-2760 and the
 *      per-frame interpolation of main.cpp:2265-2297).  Path buffers are step-major: element k of observer id is at
 *      [k*count + id].  basis_speed buffers hold float4 per observer (the reference's float3 has the same 16-byte stride). */

/* boost_tetrad, cl.cl:2441-2481; main.cpp:2700 */
int gr_boost_tetrad(gr_program* p, void* stream, const void* generic_in, int count, const void* basis_speed, void* e0_io, void*
                    e1_io, void* e2_io, void* e3_io, const void* cfg);
/* init_inertial_ray, cl.cl:3117-3141; main.cpp:2722 */
int gr_init_inertial_ray(gr_program* p, void* stream, const void* generic_position_in, int ray_count, void* rays, void*
                         ray_count_out, const void* e0, const void* e1, const void* e2, const void* e3, const void* basis_speed,
                         const void* cfg);
/* get_geodesic_path, cl.cl:4735-4940; main.cpp:2742.  velocities_out / ds_out may be NULL. */
int gr_get_geodesic_path(gr_program* p, void* stream, const void* rays, int num_rays, void* positions_out, void* velocities_out,
                         void* ds_out, const void* ray_count, int max_path_length, const void* cfg, const void* dfg, void*
                         count_out);
/* parallel_transport_quantity, cl.cl:2569-2620; main.cpp:2758 (once per tetrad leg) */
int gr_parallel_transport_quantity(gr_program* p, void* stream, const void* geodesic_path, const void* geodesic_velocity, const
                                void* ds_in, const void* quantity, const void* count_in, int count, void* quantity_out, const void*
                                cfg);
/* handle_interpolating_geodesic, cl.cl:2738-2872; main.cpp:2293: camera position + tetrad at proper time target_time */
int gr_handle_interpolating_geodesic(gr_program* p, void* stream, const void* geodesic_path, const void* geodesic_velocity, const
                                void* ds_in, void* camera_generic_out, const void* t_e0, const void* t_e1, const void* t_e2, const
                                void* t_e3, void* e0_out, void* e1_out, void* e2_out, void* e3_out, float target_time, const void*
                                count_in, int parallel_transport_observer, const void* basis_speed, void* interpolated_velocity,
                                const void* cfg);

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
/* pos (0,0,-4,0), axis-angle (1,0,0,-pi/2) */
void gr_camera_default(gr_camera* out);

enum {
    GR_MODE_REFERENCE = 0,   /* one launch per reference kernel, 96-byte ray records in HBM */
    GR_MODE_FUSED = 1        /* gr_prepass_fused + gr_trace_fused + gr_render */
};

/* Snapshot of the camera's own timelike geodesic, resident on the device: the buffers of main.cpp:1232-1242
 * (geodesic_trace / vel / ds / count) and the four parallel-transported tetrad legs. */
typedef struct gr_geodesic_camera gr_geodesic_camera;

typedef struct gr_frame_tuning gr_frame_tuning;   /* geodesic_hip_internal.h: which fused kernel, schedule and launch size (defaults are right) */

typedef struct gr_frame_options {
    int mode;              /* GR_MODE_* */
    int tiled;             /* reference mode only: 8x8-tile ray order (ignored when adaptive sampling is on) */
    int use_prepass;       /* -1: per metric config (metric_cfg.use_prepass), 0 / 1 force it off / on for this frame; -2: per metric config
                            * and - whole frames on the fused path - only while it skips >= 2 % of the pixels (not output-neutral where
                            * the reference is not either: a skipped pixel is black by decree) */
    int max_probes;        /* anisotropy, graphics_settings.hpp:34 (8) */
    int strip_rank;        /* fused mode, multi-GPU: image rows are dealt in blocks of block_rows rows,          */
    int strip_count;       /*   global block b belongs to device b % strip_count (1 = whole image on this device) */
    int block_rows;        /*   multiple of 8                                                                      */
    int compact_out;       /*   1: write this device's blocks back to back into out (gather layout)               */
    int time_kernels;      /* 1: HIP events around every stage of this frame (gr_render_state_stage_ms); 2: one event pair per trace launch (internal header) */
    const struct gr_camera* next_camera;   /* fused mode, optional: the camera of the NEXT gr_render_frame call of this state.  Its tetrad
                            * and prepass are computed on a side stream while this frame traces.  NULL = no look-ahead. */
    const struct gr_camera* next_camera2;  /* optional: the camera of the call after that (two prepasses in flight: split frames) */
    const gr_geodesic_camera* geodesic;   /* camera_on_geodesic (main.cpp:2264-2293): position and tetrad come from this snapshot at
                            * proper time geodesic_time instead of camera->position; camera->quat still orients the view */
    float geodesic_time;   /* current_geodesic_time of this frame */
    int parallel_transport_observer;   /* 1 (default, main.cpp:1259): interpolate the transported tetrads; 0: rebuild them */
    const gr_frame_tuning* tuning;     /* NULL = library defaults; also where the look-ahead frames' strip ranks and geodesic times and the
                                        * measurement switches live (geodesic_hip_internal.h) */
} gr_frame_options;
void gr_frame_options_default(gr_frame_options* out);

int gr_render_state_create(int device, int width, int height, gr_render_state** out);
void gr_render_state_destroy(gr_render_state* s);

/* Renders one frame into out_rgba_f32 (float4[width*height], device memory; in fused strip mode only
 * rows [row_begin,row_end) are written).  cfg_values = the $cfg parameters (NULL = metric defaults). */
int gr_render_frame(gr_render_state* s, gr_program* p, const gr_metric* m, void* stream, const gr_camera* camera, const gr_features*
                    features, const float* cfg_values, int num_cfg_values, const void* background1, const void* background2, int
                    bg_width, int bg_height, int bg_levels, void* out_rgba_f32, const gr_frame_options* options);

/* stage timing of a frame rendered with options->time_kernels = 1 (the reference's -bench mode times frames on the host, main.cpp:2864-2871) */
enum { GR_STAGE_CAMERA = 0, GR_STAGE_PREPASS = 1, GR_STAGE_INIT = 2, GR_STAGE_TRACE = 3, GR_STAGE_RENDER_DATA = 4,
       GR_STAGE_ADAPTIVE = 5, GR_STAGE_RENDER = 6, GR_STAGE_COUNT = 7 };
/* elapsed milliseconds of a stage of the last timed frame (synchronises on the stage's stop event) */
int gr_render_state_stage_ms(gr_render_state* s, int stage, float* ms);

/* Camera on a timelike geodesic, object form of main.cpp:2675-2760: _snapshot launches cart_to_generic, init_basis_vectors, boost_tetrad,
 * init_inertial_ray, get_geodesic_path and four parallel_transport_quantity on `stream`, then synchronises once to report the number of
 * samples and the proper time the path covers.  geodesic_basis_speed is g_geodesic_basis_speed (main.cpp:2253-2261), |v| < 1. */
int gr_geodesic_camera_create(int device, int max_path_length, gr_geodesic_camera** out);
void gr_geodesic_camera_destroy(gr_geodesic_camera* g);
int gr_geodesic_camera_snapshot(gr_geodesic_camera* g, gr_program* p, const gr_metric* m, void* stream, const gr_camera* camera,
                                const float geodesic_basis_speed[3], const gr_features* features, const float* cfg_values, int
                                num_cfg_values, int* steps_out, float* proper_time_out);
/* handle_interpolating_geodesic + read-back (the reference's geodesic_q / camera_q async reads, main.cpp:2295-2296):
 * generic camera position, tetrad (4 rows of 4) and 4-velocity at `proper_time`; any output may be NULL */
int gr_geodesic_camera_interpolate(gr_geodesic_camera* g, gr_program* p, void* stream, float proper_time, int
                                parallel_transport_observer, float camera_generic_out[4], float tetrad_out[16], float
                                velocity_out[4]);

/* blocking copies for tests and tools */
int gr_device_download(int device, void* host_dst, const void* device_src, size_t bytes);
int gr_device_upload(int device, void* device_dst, const void* host_src, size_t bytes);
int gr_device_alloc(int device, size_t bytes, void** out);
int gr_device_free(int device, void* ptr);
int gr_device_synchronize(int device);
/* HIP streams (the reference's cl::command_queue objects, main.cpp:1458-1461): one per frame in flight; a caller's own hipStream_t works too */
int gr_stream_create(int device, int high_priority, void** stream_out);
int gr_stream_synchronize(void* stream);
int gr_stream_destroy(void* stream);
int gr_device_count(int* count);

/* ---- one frame over several GPUs (the reference is single-GPU) ------------------------------------------
 * Image rows are dealt to `world` participants in blocks of block_rows rows, block-cyclically; each renders its share with
 * gr_render_frame's strip mode (own prepass cells, one halo row per block, nothing exchanged while tracing) and the finished float4
 * rows go to participant 0, every block straight to its rows of the frame.  gr_render_frame_tiled = gr_render_frame for the share +
 * that transfer (`options` as there; mode, strip_* and compact_out are overridden).  `rotation`: participant r renders share
 * (r + rotation) % world - rotate with the frame number to even out shares of different cost.  frame_on_root: float4[height * width]
 * on participant 0's device (NULL elsewhere, except with peer copies: the same pointer for everybody).  Frames may be in flight on
 * several streams: a participant stages its rows in a ring of GR_TILED_STAGING (4) buffers.
 *   GR_TRANSPORT_RCCL    one process per GPU; participant 0 hands gr_tiled_unique_id's 128 bytes to the others, gr_tiled_create is
 *                        collective; per frame one group of ncclSend (owner) / ncclRecv (participant 0) per block.  librccl is dlopen'ed.
 *   GR_TRANSPORT_PEER    one process driving several devices (gr_tiled_create_local): hipMemcpyPeerAsync per block; gr_tiled_join
 *                        makes participant 0's stream wait for every frame issued so far.
 *   GR_TRANSPORT_CUSTOM  the caller's point-to-point library behind a gr_transport table, called in RCCL's pattern.
 *   GR_TRANSPORT_IPC     one process per participant, participants may share a device (RCCL refuses that): the same calls in the same
 *                        order over inter-process memory handles, matched in a shared-memory mailbox named after `session`; blocks
 *                        the host at the end of a group - a rehearsal stage for boxes with fewer GPUs than ranks, not a product path. */
typedef struct gr_tiled gr_tiled;
enum { GR_TRANSPORT_RCCL = 0, GR_TRANSPORT_PEER = 1, GR_TRANSPORT_CUSTOM = 2, GR_TRANSPORT_IPC = 3 };
/* the point-to-point calls a split frame needs (the subset of RCCL it uses); group_begin / group_end may be NULL */
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
                          const gr_features* features, const float* cfg_values, int num_cfg_values, const void* background1, const
                          void* background2, int bg_width, int bg_height, int bg_levels, void* frame_on_root, const
                          gr_frame_options* options, int rotation);
int gr_tiled_join(gr_tiled* root, void* stream);
/* the share participant t renders in a frame with this rotation */
int gr_tiled_share(const gr_tiled* t, int rotation);
/* A caller that cycles through n render states (n frames in flight) and passes options->next_camera says so here: the look-ahead frame of a
 * call with rotation k is then the frame of rotation k + n (next_camera2: k + 2 n), and gr_render_frame_tiled computes its prepass for the
 * share the participant will have by then.  Default 1. */
int gr_tiled_look_ahead(gr_tiled* t, int rotations);

/* ---- host helper: background image ----------------------------------------------------------- */
/* load_mipped_image (graphics_settings.cpp:152-212): an RGBA8 image and its box-filtered mip chain in `levels` same-size slices (mip i
 * in the top-left corner of slice i, edge replicated).  Returns the number of levels; out needs levels*width*height*4 bytes (NULL asks). */
int gr_pack_mipped_background(const unsigned char* rgba, int width, int height, unsigned char* out);

/* ---- host helpers: PNG in/out (the screenshot path main.cpp:2762-2808, the background loader graphics_settings.cpp:214-243) ---- */
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
