/* geodesic_hip_internal.h - the rest of libgeodesic_hip.so's exports: the launchers of the fused MI355X path (no reference counterpart),
 * its schedules (tile orders, tickets, the pending list of adaptive sampling), the knobs gr_render_frame picks them with
 * (gr_frame_tuning) and every measurement hook (stage timers, attempt / clock counters, build keys).  Tests, bench.py and tools/ use
 * them; the contract a maintainer binds is geodesic_hip.h.  Same conventions: 0 on success, device pointers owned by the caller. */
#ifndef GEODESIC_HIP_INTERNAL_H
#define GEODESIC_HIP_INTERNAL_H

#include "geodesic_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- what gr_render_frame chooses among (gr_frame_options.tuning; NULL = these defaults) ------------------------------------- */
struct gr_frame_tuning {
    int ray_compaction;    /* fused mode: 0 = gr_trace_fused (one tile per wave at a time), 1..64 = gr_trace_compact with this
                            * keep_lanes; -1 = library default (off: the benchmark workloads keep > 95 % of their lanes busy) */
    int rays_per_lane;     /* fused mode without compaction: 1 = gr_trace_fused, 2 = gr_trace_pair (error if the program lacks it),
                            * 0 = library default: 2 where the program has the pair kernel, else 1 (GR_TRACE_RAYS_PER_LANE=1|2 overrides) */
    int fused_shading;     /* fused mode: 1 = the trace launch shades the 49 of every 64 pixels whose filter neighbours are in the same
                            * tile and gr_render_seams the rest (needs a program built with -DGR_TILE_SHADING, width and height
                            * multiples of 8, one ray per lane, no compaction, no adaptive sampling; an error otherwise);
                            * 0 or -1 (library default) = gr_render shades every pixel (measured faster: EXPERIMENTS.md C.2) */
    int inline_prepass;    /* fused mode, a frame whose prepass was not computed ahead (next_camera): trace the prepass grid inside
                            * the trace launch (gr_trace_fused_args.inline_prepass).  -1 (default): on whole frames that do not order
                            * their tiles; 1: also on a device's share of a split frame; 0: the prepass as a launch of its own in front */
    int trace_waves_per_simd;   /* fused mode: persistent waves per SIMD a trace launch takes, 1..8; 0 = as many as fit (best for
                            * one frame at a time).  With three or more frames in flight on streams of their own, 4 measured 2-3 % faster */
    int tile_history;      /* fused mode, one ray per lane: 1 = hand the tiles of this frame out dearest first by what they cost in this
                            * render state's previous frame (gr_order_tiles_by_history, shifted by how far the camera has moved the
                            * picture since); 0 = no; -1 = library default: whole frames of at most 32 tiles per wave slot that find no
                            * frame of ANOTHER stream still running on the device when they are submitted.  Scheduling only. */
    int park_lanes;        /* fused mode, one ray per lane, every pixel: 2..64 = gr_trace_fused_parking - a tile's wave with fewer than this many
                            * rays left after park_trips trips of the Verlet loop hands them over (gr_parking_lot); 0 = no; -1 = library
                            * default (GR_PARK="lanes,trips" in the environment, else off).  Scheduling only.  The program must have been
                            * built with -DGR_PARKING appended to its argument string (gr_program_has_parking): an error if asked for
                            * explicitly without it, ignored when it only comes from the environment. */
    int park_trips;        /* ... trips (two attempts each); <= 0 = library default (512) */
    /* -- what rides with the look-ahead cameras of gr_frame_options (next_camera, next_camera2) -- */
    int next_strip_rank;   /* strip_rank of the next_camera / next_camera2 frames when a device's share rotates from frame to frame; */
    int next_strip_rank2;  /*   -1 (default) = the same as this frame's.  gr_render_frame_tiled fills both in (gr_tiled_look_ahead). */
    float next_geodesic_time, next_geodesic_time2;   /* current_geodesic_time of the look-ahead frames (gr_frame_options.geodesic) */
    /* -- measurement -- */
    int count_attempts;    /* 1: accumulate the Verlet step attempts of this frame (gr_render_state_attempts) */
    /* -- an interactive caller that does not know the next camera (round 6) -- */
    int guess_still_camera;   /* fused mode, whole frames with a prepass, no next_camera given: 1 = when this frame repeats the previous frame of
                            * this render state (as below), take "the same again" as the NEXT frame's camera - its camera set-up and prepass
                            * then run on the side stream while this frame traces, exactly as for an announced next_camera, and are used only
                            * if the next frame's key matches bit for bit.  The prepass is still computed every frame (a measurement that must
                            * not skip it uses this with reuse_still_camera = 0); it has to find wave slots beside a trace launch that fills
                            * the device, and a long one (Kerr a = 0.9: 8 ms) that loses that race ends up in front of the next frame -
                            * profiles/r06_still_camera.txt.  0 = never; -1 = library default (off: reuse_still_camera supersedes it). */
    int reuse_still_camera;   /* fused mode, whole frames with a prepass and a Cartesian camera: 1 = a frame whose camera, parameters, features
                            * and program equal, bit for bit, those of the previous frame of this render state on the same stream takes that
                            * frame's camera set-up and prepass verdicts as they stand - both are functions of exactly those inputs and
                            * still sit in the state's buffers (a pointer to them handed out by gr_render_state_buffer ends that) - and
                            * launches neither.  A viewer whose user has stopped moving renders the same camera again and again; the
                            * reference pays the prepass's single-ray latency on each of those frames (main.cpp:2384-2437).  0 = never;
                            * -1 = library default (on).  gr_render_state_prepass_reused counts the frames that did. */
    int speculative_classes;  /* a whole frame that traces its prepass inside its trace launch and hands its tiles out by the frame before's
                            * costs (tile_history): gr_trace_fused_args.speculative_classes of its launch - n classes, 0 = none, -1 = library default */
};
void gr_frame_tuning_default(gr_frame_tuning* out);

/* ---- buffers of the frame driver's objects, the background build a program manager runs (tests and tools) ---- */
enum { GR_GEOBUF_PATH = 0, GR_GEOBUF_VELOCITY = 1, GR_GEOBUF_DS = 2, GR_GEOBUF_COUNT = 3, GR_GEOBUF_TRANSPORTED0 = 4,
       GR_GEOBUF_TRANSPORTED1 = 5, GR_GEOBUF_TRANSPORTED2 = 6, GR_GEOBUF_TRANSPORTED3 = 7 };
void* gr_geodesic_camera_buffer(gr_geodesic_camera* g, int which);

enum { GR_BUF_RAYS_IN = 0, GR_BUF_RAYS_COUNT = 1, GR_BUF_RENDER_DATA = 2, GR_BUF_TERMINATION = 3, GR_BUF_CAMERA_GENERIC = 4,
       GR_BUF_TETRAD0 = 5, GR_BUF_TETRAD1 = 6, GR_BUF_TETRAD2 = 7, GR_BUF_TETRAD3 = 8, GR_BUF_RAYS_ADAPTIVE = 9,
       GR_BUF_RAYS_ADAPTIVE_COUNT = 10, GR_BUF_CFG = 11, GR_BUF_DFG = 12, GR_BUF_CAMERA_QUAT = 13 };
/* device pointer of one of the state's buffers (NULL if not allocated).  Asking for the camera set, the prepass verdicts or the parameters
 * (anything but the ray and render-data records) tells the state that they may be written from outside: the next frame then does its own
 * camera set-up and prepass whatever gr_frame_tuning.reuse_still_camera says.  That holds for the NEXT frame only - a caller that keeps such
 * a pointer and writes through it in later frames renders those with reuse_still_camera = 0. */
void* gr_render_state_buffer(gr_render_state* s, int which);

/* counters of a program manager: out[0] parameter changes taken (gr_program_manager_update), [1] substituted programs swapped in, [2] substituted
 * builds started - never more than one of them is running -, [3] 1 while a build is running whose result nobody wants any more */
void gr_program_manager_counters(const gr_program_manager* pm, unsigned long long out[4]);

/* Background build of the "substituted" program (metric_manager.hpp:153-166, swapped in by check_substitution :172-219): create_async
 * returns at once; poll returns 1 and a loaded program when it is ready, 0 while pending, < 0 on a build error. */
typedef struct gr_program_future gr_program_future;
int gr_program_create_async(const char* argument_string, int device, gr_program_future** out);
/* compile only (no device): what gr_program_create_async's worker builds before the program can be swapped in - the code object of the
 * kernels a fused frame launches and the set-up module, side by side on two threads; fills the cache (bench.py: startup.program_build_s) */
int gr_program_precompile_frame_path(const char* argument_string);
int gr_program_future_poll(gr_program_future* f, gr_program** out);
void gr_program_future_destroy(gr_program_future* f);


/* gr_metric_info's operation counts for the substituted program of these parameter values (NULL = defaults): parameters that
 * make parts of a metric vanish - real rod lengths in the complex-valued double-Kerr family - shrink the DAG a good deal. */
int gr_metric_substituted_op_counts(const gr_metric* m, const float* cfg_values, int num_cfg_values, int* accel_ops,
                                    int* accel_transcendentals, int* coord_ops);


/* registers / scratch of a kernel as recorded in the code object (0 if unknown) */
int gr_program_kernel_info(const gr_program* p, const char* kernel_name, int* vgprs, int* sgprs, int* scratch_bytes);

/* render restricted to this device's row blocks (block-cyclic rows, see gr_trace_fused); render_data must be
 * pixel-indexed.  compact_out = 1 writes the device's blocks back to back (block i of this device at
 * out + i*block_rows*width float4), which is the layout the multi-GPU gather ships. */
int gr_render_strips(gr_program* p, void* stream, const void* render_data, void* out_rgba_f32,
                     const void* background1, const void* background2, int bg_width, int bg_height, int bg_levels,
                     int width, int height, int block_rows, int strip_rank, int strip_count, int compact_out,
                     int max_probes, const void* cfg, const void* dfg);
/* number of row blocks device `strip_rank` owns */
int gr_strip_local_blocks(int height, int block_rows, int strip_rank, int strip_count);

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
/* The reference-shaped trace with the fused trace's scheduling (round 5; what gr_render_frame's reference mode launches for ray records in
 * 8x8-tile slot order): persistent waves draw tiles - 64 consecutive records - from a ticket counter, in tile_order's order (unsigned per
 * tile; NULL = slot order), and leave each tile's cost (attempts of its longest ray) in tile_cost (unsigned per tile; NULL = not).  The
 * records are gr_do_generic_rays'.  gr_sort_tiles_by_cost: the tiles dearest first by such costs (largest among a tile and its neighbours);
 * work: (128 + tiles) unsigned words of scratch. */
int gr_do_generic_rays_scheduled(gr_program* p, void* stream, void* rays, const void* ray_count, int tile_count, const void* cfg, const void* dfg,
                                 void* attempt_counter, const void* tile_order, void* tile_cost);
int gr_sort_tiles_by_cost(gr_program* p, void* stream, const void* tile_cost, int tiles_x, int tiles_y, void* tile_order, void* work);

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
/* Parking (kernels/trace.hip, gr_trace_fused_parking): a tile's wave that has fewer than `lanes` rays left after `trips` trips of the
 * Verlet loop (two attempts each) writes their state to the lot and draws its next tile; parked rays are picked up 64 to a wave by
 * whichever wave draws next.  Scheduling only: the records are those of the unparked launch bit for bit.  lanes = 0 (or a NULL
 * lot): the plain kernel.  records: gr_parking_lot_bytes(slots, groups, &words_bytes) bytes, words: words_bytes; a lot that runs
 * out of room is not an error (the wave goes on with its rays itself; words[4] counts the times). */
typedef struct gr_parking_lot {
    void* records;
    void* words;
    int lanes, trips;
    int slots, groups;   /* capacity: parked rays (<= 2^24), and groups of rays parked together (each parking event is one) */
} gr_parking_lot;
size_t gr_parking_lot_bytes(int slots, int groups, size_t* words_bytes);
/* 1 when the program's argument string (or GR_EXTRA_FLAGS) carried -DGR_PARKING: it has the gr_trace_fused_parking kernel.  A build option,
 * like -DGR_TILE_SHADING: measured on MI355X the kernel does not pay (EXPERIMENTS.md C.2), and every program would carry its compile time. */
int gr_program_has_parking(const gr_program* p);
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
    gr_parking_lot parking;      /* lanes > 0: gr_trace_fused_parking (every pixel, one ray per lane, no in-tile shading) */
    int speculative_classes;     /* inline_prepass with tile_order_by_history: the tiles of the list's first n classes (octaves of a tile's
                                  * longest ray in the frame before, 16 384 attempts = class 0) do not wait for the prepass cells their pixels
                                  * look at: they trace every pixel from the start and take the verdicts when their rays have ended - the
                                  * launch's critical path is otherwise the longest cell ray followed by the longest tile, which reads that
                                  * cell.  Records are the same.  0 = library default (5: tiles that had a ray of 1 024 attempts or more), -1 = none */
    void* guessed;               /* lattice = 2 (persistent launch): gr_guessed_bytes of device memory whose first word says how many pixels of
                                  * its list the launch is to trace ahead, beside the lattice - the pixels the frame before traced in its second
                                  * launch and found dear (gr_trace_pending leaves them) - each record into the same buffer for
                                  * gr_apply_guessed; NULL or a count of 0: none */
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
 * the four lattice rays around the block, a quarter of an octave of attempts per class, dearest first; gr_trace_pending traces the list 64
 * entries to a wave (waves_per_simd as in gr_trace_fused_args; 0 = as many as fit).  Every lane of every wave has a ray, the rays of a wave
 * are neighbours of one cost class, and the longest rays of the frame start first.  Records equal those of the pending_only launch to
 * rounding (another kernel around the same device functions).  block_cost (unsigned per 2x2 block, (width / 2) * (height / 2), zeroed by the
 * caller; may be NULL): gr_trace_pending leaves what the dearest ray of each block cost; given to the NEXT frame's gr_adaptive_refine_list
 * as block_cost_before (while the picture has moved little) it orders that list by the blocks' own rays - the lattice rays either side of
 * a filament of long rays say nothing about it. */
size_t gr_pending_list_bytes(int width, int height);
size_t gr_lattice_rays_bytes(int width, int height);
int gr_adaptive_refine_list(gr_program* p, void* stream, void* render_data, void* pending_count, int width, int height, const void* dfg,
                            int block_rows, int strip_rank, int strip_count, const void* lattice_rays, const void* cfg, void* pending_list,
                            const void* block_cost_before);
int gr_trace_pending(gr_program* p, void* stream, const void* camera_generic, const void* camera_quat, void* render_data, int width, int height,
                     const void* e0, const void* e1, const void* e2, const void* e3, const void* cfg, const void* dfg, void* attempt_counter,
                     const void* pending_list, int waves_per_simd, void* block_cost, void* guessed_next);
/* Tracing ahead what the second launch will ask for (round 6).  An adaptively sampled frame is two dependent launches, and where single rays
 * run to the step cap (an extremal hole at 1080p) each lasts as long as its longest ray: 2 x 14 ms.  gr_trace_pending leaves the pixels that
 * cost 4 096 attempts or more in guessed_next (gr_guessed_bytes; its first word zeroed by the caller; NULL: nothing kept, nothing passed
 * over); the next frame's lattice launch traces them beside its tiles (gr_trace_fused_args.guessed), gr_apply_guessed - after
 * gr_adaptive_refine_list, before gr_trace_pending - hands a guessed pixel that IS marked its record, attempts and cost (and keeps it
 * guessed for the frame after), and gr_trace_pending passes over what is no longer marked.  Records are those of a frame that guesses nothing. */
size_t gr_guessed_bytes(void);
int gr_apply_guessed(gr_program* p, void* stream, void* render_data, int width, const void* guessed, void* guessed_next, void* block_cost,
                     void* attempt_counter);
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
 * steps - not the case for the BASELINE workloads, where 8x8 tiles keep >= 94 % of the lanes busy (EXPERIMENTS.md C.2). */
int gr_trace_compact(gr_program* p, void* stream, const void* camera_generic, const void* camera_quat,
                     void* render_data, int width, int height, int block_rows, int strip_rank, int strip_count,
                     const void* termination_buffer, int prepass_width, int prepass_height,
                     const void* e0, const void* e1, const void* e2, const void* e3,
                     const void* cfg, const void* dfg, void* attempt_counter, int keep_lanes);


/* What tile_history did with this render state's frames so far: how many recorded their tiles' costs, how many of those followed
 * the costs of the frame before, and the shift (in tiles) the last one that did applied.  Any pointer may be NULL. */
int gr_render_state_tile_history(gr_render_state* s, unsigned long long* frames_recorded, unsigned long long* frames_followed,
                                 int last_shift[2]);
/* Do the expressions a program's Verlet loop evaluates (its GEO_ACCEL* / GR_DEVICE_ACCEL* macros and their temporaries) call sin / cos?  1 yes,
 * 0 no, -1 no string.  A program whose accelerations call neither is built with -DGR_ACCEL_WITHOUT_TRIG: its loop has no range-limited
 * polynomial whose NaN it would have to tell from the metric's own (kernels/integrator.hip). */
int gr_argument_string_accelerations_call_trig(const char* argument_string);
/* How many frames of this render state took the previous frame's camera set-up and prepass (gr_frame_tuning.reuse_still_camera). */
int gr_render_state_prepass_reused(gr_render_state* s, unsigned long long* frames);
/* The two estimates tile_history works with (host arithmetic, no device).  gr_camera_origin_on_screen: the pixel at which the
 * camera sees the coordinate origin as if space were flat - the inverse of the kernels' pixel -> direction map (cl.cl:2015-2059) -
 * 1 and pixel_out[0..1] = (x, y), or 0 when the origin is behind the camera or the camera sits on it.  gr_picture_motion: an upper
 * estimate of how many pixels the picture moves between two cameras (angle between the orientations + parallax of the origin, at the
 * focal length); 1e9 when flip or observer speed differ. */
int gr_camera_origin_on_screen(const gr_camera* camera, float field_of_view, int width, int height, float pixel_out[2]);
float gr_picture_motion(const gr_camera* from, const gr_camera* to, float field_of_view, int width);


/* what the prepass policy (gr_frame_options.use_prepass = -2) has done with this state's frames so far, and the fraction of
 * the prepass grid the last inspected prepass made skippable (-1: none inspected yet); any output may be NULL */
int gr_render_state_prepass_policy(gr_render_state* s, unsigned long long* frames_with_prepass, unsigned long long* frames_without,
                                   float* last_marked_fraction);

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

/* the transfer step of gr_render_frame_tiled on its own: `staging` = this participant's compact rows (blocks back to back,
 * gr_tiled_staging_bytes; unused on participant 0), frame_on_root as there */
int gr_tiled_exchange(gr_tiled* t, const void* staging, void* frame_on_root, int rotation, void* stream);
size_t gr_tiled_staging_bytes(const gr_tiled* t);
/* rows [row_begin, row_end) of the local_block-th block of a share; returns 1, 0 for a padding block past the image, -1 on bad
 * arguments.  Pure arithmetic: global block = local_block * world + share. */
int gr_tiled_block_rows(int height, int block_rows, int world, int share, int local_block, int* row_begin, int* row_end);
int gr_tiled_block_rows_of(const gr_tiled* t, int share, int local_block, int* row_begin, int* row_end);

#ifdef __cplusplus
}
#endif
#endif
