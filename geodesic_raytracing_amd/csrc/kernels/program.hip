// kernels/*.hip — gfx950 (CDNA4, wave64) kernels of the per-pixel geodesic ray pipeline.  This is the first of the parts the host
// concatenates into one translation unit (csrc/capi.cpp KERNEL_PARTS): program, probes.inc, metric, setup, integrator, trace, shading,
// geodesic_camera.
//
// This translation unit is compiled at run time (hiprtc, --offload-arch=gfx950) once per metric,
// specialised by the same `-D` macro set the reference feeds to its OpenCL program
// (producer metric.hpp:725-959, consumer cl.cl): F*_I, F*_P, TO_/FROM_(D)COORDn, GEO_ACCELn,
// TEMPORARIES0, DISTANCE_FUNC, W_Vn, DYNVARS, feature macros, behaviour flags.
//
// Kernel <-> reference map (semantics, argument order):
//   gr_cart_to_generic          cart_to_generic_kernel    cl.cl:6018-6034
//   gr_init_basis_vectors       init_basis_vectors        cl.cl:2483-2507 (calculate_tetrads 2288-2439)
//   gr_clear_termination_buffer clear_termination_buffer  cl.cl:4997-5006
//   gr_init_rays_generic        init_rays_generic         cl.cl:3143-3251
//   gr_do_generic_rays          do_generic_rays           cl.cl:3954-4247 (step_verlet 3273-3346,
//                                                         calculate_ds_error 3431-3456)
//   gr_calculate_singularities  calculate_singularities   cl.cl:5008-5020
//   gr_calculate_render_data    calculate_render_data     cl.cl:5135-5213
//   gr_handle_adaptive_sampling handle_adaptive_sampling  cl.cl:5223-5345
//   gr_render                   render                    cl.cl:5453-5846 (read_mipmap 5421-5449)
//   gr_trace_fused              (no counterpart) init -> integrate -> render-data in one launch,
//                               ray state never leaves registers; persistent tile-waves
//   gr_trace_pair               (no counterpart) gr_trace_fused with two rays per lane in packed fp32 (integrate_pair);
//                               only built for programs whose loop expressions instantiate on float pairs
//                               (GR_TWO_RAYS_PER_LANE, decided by the host: capi.cpp pair_kernel_applies)
//   gr_trace_compact            (no counterpart) gr_trace_fused with ray compaction (resumable integrator)
//   gr_prepass_fused            (no counterpart) the W/16 x H/16 prepass as one launch -> termination flags
//   gr_boost_tetrad             boost_tetrad              cl.cl:2441-2481
//   gr_init_inertial_ray        init_inertial_ray         cl.cl:3117-3141
//   gr_get_geodesic_path        get_geodesic_path         cl.cl:4735-4940
//   gr_parallel_transport_quantity  parallel_transport_quantity  cl.cl:2569-2620
//   gr_handle_interpolating_geodesic  handle_interpolating_geodesic  cl.cl:2738-2872
//
// MI355X design notes
//   * one ray per lane, one wave64 per 64 consecutive ray slots; ray slots are laid out in 8x8
//     pixel tiles (GR_TILE) so a wave integrates an angularly compact bundle: step counts inside
//     a wave stay close and the lock-step loop wastes few lanes;
//   * the integrator state (position, velocity, acceleration, step, flags = 16 VGPRs) and every
//     metric temporary live in registers; cfg / feature values are wave-uniform kernel-argument
//     loads (SGPRs);
//   * accept / reject of an adaptive step is a per-lane select - both outcomes ran the same
//     step_verlet, so rejection costs no divergence; a wave leaves the loop on a ballot of
//     finished lanes;
//   * the Verlet loop is written against the issue rates measured on this GPU (tools/ubench/valu_rate.hip): full rate
//     for fma/mul/add/mov/bit ops, half rate for compares, selects, min/max, conversions, quarter rate for
//     rcp/rsq/sqrt - see degenerate_accumulate, acceleration_to_precision, sincos_reduced;
//   * no MFMA: the work is a 4x4 per-ray ODE, bound by fp32 VALU issue, not by HBM or matrix rate.
//
// No double-precision arithmetic on the hot path; the few double expressions of the reference's
// texture-space code (M_PI literals, cl.cl:3598-3610, 5272) are mirrored where they change results.

#define GR_PI 3.14159265358979323846
#define GR_PIf 3.14159274101257324f

struct lightray {
    float4 position;
    float4 velocity;
    float4 initial_quat;
    float4 acceleration;
    float ku_uobsu;
    float running_dlambda_dnew;
    int terminated;
    int sx;
    int sy;
};   // 96 bytes (render_state.hpp:8-19)

struct render_data {
    float2 tex_coord;
    float z_shift;
    int sx;
    int sy;
    int terminated;
    int side;
};   // 32 bytes (render_state.hpp:21-29)

struct dynamic_config {
#ifdef DYNVARS
    float DYNVARS;
#else
    float gr_unused;
#endif
};

struct dynamic_feature_config {
#ifdef DYNAMIC_FLOAT_FEATURES
    float DYNAMIC_FLOAT_FEATURES;
#endif
#ifdef DYNAMIC_BOOL_FEATURES
    int DYNAMIC_BOOL_FEATURES;
#endif
#if !defined(DYNAMIC_FLOAT_FEATURES) && !defined(DYNAMIC_BOOL_FEATURES)
    int gr_unused;
#endif
};

#ifdef KERNEL_IS_STATIC
#define GET_FEATURE(name, dfg) FEATURE_##name
#else
#define GET_FEATURE(name, dfg) ((dfg)->name)
#endif

#if defined(GENERIC_CONSTANT_THETA)
#define IS_CONSTANT_THETA
#endif

// Which kernels a build of this translation unit holds (capi.cpp: compile_code_object).  A program is two code objects: the kernels a
// fused frame launches (-DGR_BUILD_FRAME_PATH: gr_trace_fused, gr_trace_pair, gr_prepass_fused, gr_order_tiles, gr_render - and the
// adaptive-sampling kernels unless the program is substituted with adaptive sampling off), built first, and the others
// (-DGR_BUILD_REST: the reference-shaped sequence, ray compaction), built behind them: after a parameter change the substituted
// program is swapped in when its frame path is there (metric_manager.hpp:172-219 waits for the whole cl::program).  Neither macro: all.
#if defined(GR_BUILD_FRAME_PATH)
#define GR_FRAME_KERNELS 1
#define GR_OTHER_KERNELS 0
#elif defined(GR_BUILD_REST)
#define GR_FRAME_KERNELS 0
#define GR_OTHER_KERNELS 1
#else
#define GR_FRAME_KERNELS 1
#define GR_OTHER_KERNELS 1
#endif
#if defined(KERNEL_IS_STATIC) && defined(FEATURE_adaptive_sampling)
#if FEATURE_adaptive_sampling
#define GR_ADAPTIVE_KERNELS GR_FRAME_KERNELS
#else
#define GR_ADAPTIVE_KERNELS GR_OTHER_KERNELS
#endif
#else
#define GR_ADAPTIVE_KERNELS GR_FRAME_KERNELS
#endif

#ifndef GR_TILE
#define GR_TILE 8
#endif
// render_data.terminated of a pixel that gr_adaptive_refine wants traced (the reference's values are 0, 1, 2)
#define GR_PENDING (-1)
#define GR_TILE_CLASSES 16
#define GR_TILE_ORDER_HEADER (2 * GR_TILE_CLASSES)   // words in front of gr_order_tiles' list: class counts, class cursors
// what kind of list it is, left by the launch that made it in the first word behind the list (the classes' scratch area, free once
// the list is dealt): gr_order_tiles' last class is a PROMISE that no pixel of its tiles needs a ray - the trace writes their records
// without looking anything up - gr_order_tiles_by_history's a guess that is looked up like any other tile
#define GR_LIST_BY_PREPASS 0x50524550u
#define GR_LIST_BY_HISTORY 0x48495354u
#ifndef GR_TILE_COST_REACH
#define GR_TILE_COST_REACH 1      // cells either side of the tile centre's whose rays' costs count for the tile's class
#endif
// The pixels guessed for the second launch of adaptive sampling (trace.hip: trace_tile's guess waves, gr_apply_guessed, gr_trace_pending): one
// buffer of GR_GUESSED_HEADER words (word 0: how many), GR_GUESSED_CAPACITY pixels, as many attempt counts, as many 32-byte records
// (capi.cpp gr_guessed_bytes).  A pixel is worth guessing if it cost GR_GUESSED_ATTEMPTS attempts or more in the frame before.
#define GR_GUESSED_HEADER 8
#define GR_GUESSED_CAPACITY 65536
#ifndef GR_GUESSED_ATTEMPTS
#define GR_GUESSED_ATTEMPTS 4096u
#endif
#ifndef GR_CELL_BLOCK
#define GR_CELL_BLOCK 1           // the prepass cells a trace launch traces itself: 1 = a wave to 8 x 8 cells, 0 = to 64 cells of a row (capi.cpp: prepass_tickets)
#endif
#ifndef GR_TILE_CLASS_STEPS
#define GR_TILE_CLASS_STEPS 1     // cost classes of gr_order_tiles per octave of attempts (finer ones measured no better)
#endif
#define GR_SKIP_CHUNK 32          // tiles of the last class per ticket
// the frame's counter block (uint64 words): [0] attempts, [1] shader cycles, [2] 100 MHz ticks, [3] waves, [8..255] probe builds,
// [256..511] the fused trace's attempts, spread over as many words as keep same-address atomics apart
#define GR_ATTEMPT_COUNTERS_AT 256
#define GR_ATTEMPT_COUNTERS 256

// minimum resident waves per SIMD the integrator kernels are register-allocated for (512 VGPRs / N waves each).
// 1 = no cap: the allocator takes what the metric's expressions need and occupancy follows (substituted Kerr: 92 VGPRs
// in the persistent fused kernel -> 5 waves/SIMD, which already saturates the VALU; the complex-valued double-Kerr
// metric: 186-370 VGPRs -> 1-2 waves/SIMD but no spills).  Measured on MI355X: forcing 6-8 waves on Kerr does not make it
// faster; capping double Kerr at 128 VGPRs costs 5.3x.
// trips of the fast Verlet loop (two attempts each) after which a wave raises its issue priority, again at twice and four times
// the count (integrator.hip); 0 = never.  Measured (tools/priority_probe.py, 4K Kerr): a = 0.9 one frame at a time 17.8 -> 16.3 ms
// with 64 ... 256 alike, 16.6 with 512; a = 0.45, frames in flight, adaptive sampling, shares of a split frame: unchanged.
#ifndef GR_PRIORITY_TRIPS
#define GR_PRIORITY_TRIPS 128
#endif
#ifndef GR_TRACE_WAVES
#define GR_TRACE_WAVES 1
#endif

// the same for gr_trace_fused alone: the host rebuilds a program with this set when that buys the kernel occupancy without
// spilling in its loop (capi.cpp compile_code_object)
#ifndef GR_FUSED_WAVES
#define GR_FUSED_WAVES GR_TRACE_WAVES
#endif

typedef const dynamic_config* __restrict__ cfg_t;
typedef const dynamic_feature_config* __restrict__ dfg_t;

// The integrator kernels copy the (wave-uniform) $cfg and feature blocks into registers once: the generated
// expressions say `cfg->NAME` inside the Verlet loop, and re-reading them through the pointer costs a scalar load
// plus an lgkmcnt wait per step.
#define GR_PARAMETERS_IN_REGISTERS                                   \
    const dynamic_config gr_cfg_registers = *cfg_in;                 \
    const dynamic_feature_config gr_dfg_registers = *dfg_in;         \
    const dynamic_config* const cfg = &gr_cfg_registers;             \
    const dynamic_feature_config* const dfg = &gr_dfg_registers;

