// frame.cpp — per-frame device buffers and the frame enqueue sequence.
//
// Reference counterparts: render_state.hpp:97-197 (buffers), main.cpp:2244-2526 (the sequence of
// launches on one in-order queue), execute_kernel main.cpp:139-205.  Everything is asynchronous on
// the caller's stream; the only host->device traffic per frame is camera (48 B), cfg and features.
// library default of gr_frame_options.rays_per_lane = 0 (see include/geodesic_hip.h)
#ifndef GR_DEFAULT_TILE_HISTORY
#define GR_DEFAULT_TILE_HISTORY 1
#endif
#ifndef GR_DEFAULT_RAYS_PER_LANE
#define GR_DEFAULT_RAYS_PER_LANE 2   /* where the program has gr_trace_pair (capi.cpp: pair_kernel_applies) */
#endif
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <mutex>

#include "../../include/geodesic_hip_internal.h"

extern "C" int gr_internal_fail(int code, const char* msg);   // capi.cpp

#define HIP_CHECK(expr)                                                                                  \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess)                                                                            \
            return gr_internal_fail(GR_ERROR_DEVICE, (std::string(#expr) + ": " + hipGetErrorString(_e)).c_str()); \
    } while (0)
#define GR_CHECK(expr)            \
    do {                          \
        int _rc = (expr);         \
        if (_rc != GR_OK) return _rc; \
    } while (0)

// Small host -> device uploads (camera, $cfg values, features) go through PINNED memory of the library's own: hipMemcpyAsync from
// pageable memory may read its source when the stream gets there, not when it is called - and the sources here are locals of
// gr_render_frame and the caller's structs.  With the device to itself a frame's uploads ran at once and nothing showed; eight
// processes sharing one GPU (the inter-process rehearsal of a split frame, tests/test_gpu_two_ranks.py) delayed the streams, and
// a state's first frame read its features off a dead stack frame - one share of one frame rendered with garbage parameters.
// A ring of 256-byte chunks; a chunk is reused 64 uploads later at the earliest, after the event recorded behind ITS copy on the stream
// that copy went to (round 5: one event per 16 uploads, recorded on whichever stream issued the 16th, did not cover the copies the
// caller's stream and the look-ahead slots' streams had queued in between - the host never blocks in the pipelined path and can run
// 64 uploads ahead of a delayed stream).  One lock: two threads may drive one state's look-ahead.
struct upload_ring {
    static const int CHUNK = 256, CHUNKS = 64;
    char* base = nullptr;
    hipEvent_t used[CHUNKS] = {};
    bool recorded[CHUNKS] = {};
    unsigned long long next = 0;
    std::mutex lock;
    int copy(void* dst, const void* src, size_t bytes, hipStream_t stream) {
        if (bytes > (size_t)CHUNK) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "upload_ring: too large");
        std::lock_guard<std::mutex> guard(lock);
        if (!base) {
            HIP_CHECK(hipHostMalloc((void**)&base, (size_t)CHUNK * CHUNKS, hipHostMallocDefault));
            for (auto& e : used) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        const int chunk = (int)(next % CHUNKS);
        if (recorded[chunk]) HIP_CHECK(hipEventSynchronize(used[chunk]));   // (64 uploads ago: long done, whichever stream it was on)
        memcpy(base + (size_t)chunk * CHUNK, src, bytes);
        HIP_CHECK(hipMemcpyAsync(dst, base + (size_t)chunk * CHUNK, bytes, hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipEventRecord(used[chunk], stream));
        recorded[chunk] = true;
        next++;
        return GR_OK;
    }
    void release() {
        if (base) (void)hipHostFree(base);
        for (auto& e : used) if (e) (void)hipEventDestroy(e);
        base = nullptr;
    }
};

struct gr_render_state {
    int device = 0;
    int width = 0, height = 0;
    upload_ring uploads;
    // small buffers (render_state.hpp:150-170)
    void* camera_pos_cart = nullptr;
    void* camera_quat = nullptr;
    void* camera_pos_generic = nullptr;
    void* tetrad[4] = {};
    void* rays_count_in = nullptr;
    void* rays_adaptive_count = nullptr;
    void* render_data_count = nullptr;
    void* cfg = nullptr;            // struct dynamic_config (floats in declaration order)
    void* dfg = nullptr;            // struct dynamic_feature_config
    void* attempts = nullptr;       // uint64[GR_COUNTER_WORDS]: attempts, shader cycles, 100 MHz ticks, waves (the last three: fused trace only); [8..255] probe builds; [256..511] the fused trace's attempts, spread
    // per-pixel buffers (render_state.hpp:172-196); ray records are allocated on first use
    void* rays_in = nullptr;
    void* rays_adaptive = nullptr;
    void* render_data = nullptr;
    void* termination_buffer = nullptr;
    void* tile_order = nullptr;      // the order the persistent trace hands its tiles out in (gr_order_tiles)
    size_t tile_order_bytes = 0;
    // what each tile of the last fused frame cost (gr_trace_fused_args.tile_cost) and the frame shape that goes with it: the next
    // frame's tiles are handed out dearest first by it (gr_frame_options.tile_history)
    void* lattice_rays = nullptr;   // adaptive sampling on the fused path: gr_lattice_rays_bytes (allocated on first use)
    // the pixels of the second launch traced ahead by the lattice launch (gr_apply_guessed): [0] what this frame's lattice launch traces,
    // [1] what this frame's second launch leaves for the next; swapped every frame that keeps them
    void* guessed[2] = {nullptr, nullptr};
    bool guessed_valid = false;
    void* parking_records = nullptr;   // gr_trace_fused_parking's lot (gr_parking_lot_bytes; allocated the first time a frame parks)
    void* parking_words = nullptr;
    int parking_slots = 0;
    void* pending_list = nullptr;   // ... the pixels of its second launch, dearest first: gr_pending_list_bytes
    // ... and what the rays of each 2x2 block cost in this frame / in the frame before (the two alternate): the order of the next
    // frame's list while the picture moves little
    void* block_cost = nullptr;
    void* block_cost_before = nullptr;
    bool block_cost_valid = false;
    unsigned long long block_cost_program = 0;
    gr_camera block_cost_camera{};
    void* tile_cost = nullptr;
    int tile_cost_shape[3] = {0, 0, 0};   // block_rows, strip_rank, strip_count
    bool tile_cost_valid = false;
    float tile_cost_anchor[2] = {0, 0};   // the pixel that frame's camera saw the coordinate origin at (origin_on_screen)
    bool tile_cost_anchored = false;
    gr_camera tile_cost_camera{};
    unsigned long long tile_cost_program = 0;
    // reference-shaped sequence, rays in tile slot order: what every tile cost in this state's last such frame ([1]: the one before),
    // and the tiles sorted by it (gr_do_generic_rays_scheduled, gr_sort_tiles_by_cost)
    void* ref_cost[2] = {nullptr, nullptr};
    void* ref_order = nullptr;
    void* ref_sort_work = nullptr;
    int ref_cost_tiles = 0;
    bool ref_cost_valid = false;
    gr_camera ref_cost_camera{};
    unsigned long long ref_cost_program = 0;
    unsigned long long history_recorded = 0, history_followed = 0;   // frames (gr_render_state_tile_history)
    unsigned long long prepass_reused = 0;   // frames that took the previous frame's set-up and prepass (gr_render_state_prepass_reused)
    int history_last_shift[2] = {0, 0};
    size_t ray_capacity = 0;
    hipEvent_t ev_start[GR_STAGE_COUNT] = {};
    hipEvent_t ev_stop[GR_STAGE_COUNT] = {};
    bool stage_timed[GR_STAGE_COUNT] = {};
    std::vector<float> host_cfg;
    gr_features host_features{};
    bool features_valid = false;
    // Look-ahead (gr_frame_options.next_camera / next_camera2): the camera set-up and the prepass of the next one or two
    // frames - latency-bound launches of only W/16 x H/16 rays - run on high-priority side streams into buffer sets of
    // their own while this frame traces.  A frame that finds its own request in a slot swaps that set in and skips both.
    struct camera_set {
        void* camera_pos_cart = nullptr;
        void* camera_quat = nullptr;
        void* camera_pos_generic = nullptr;
        void* tetrad[4] = {};
        void* termination_buffer = nullptr;
        void* tile_order = nullptr;   // gr_order_tiles' list for the frame's trace (the cost estimates sit behind the prepass flags)
    };
    struct prefetch_key {
        gr_camera camera{};
        std::vector<float> cfg;
        gr_features features{};
        unsigned long long program = 0;   // gr_program_serial: an address could be reused by a later program
        const void* geodesic = nullptr;
        float geodesic_time = 0;
        int transport = 0;
        int strip[3] = {0, 0, 0};   // block_rows, strip_rank, strip_count: the prepass only covers the cells these rows look at
        bool operator==(const prefetch_key& o) const {
            return memcmp(&camera, &o.camera, sizeof(camera)) == 0 && cfg == o.cfg && memcmp(&features, &o.features, sizeof(features)) == 0 &&
                   program == o.program && geodesic == o.geodesic && (!geodesic || (geodesic_time == o.geodesic_time && transport == o.transport)) &&
                   memcmp(strip, o.strip, sizeof(strip)) == 0;
        }
    };
    struct prefetch_slot {
        camera_set set;
        void* velocity = nullptr;   // interpolated 4-velocity written by handle_interpolating_geodesic (unused here)
        hipStream_t stream = nullptr;
        hipEvent_t ready = nullptr;
        bool valid = false;
        unsigned long long age = 0;
        prefetch_key key;
    };
    static const int LOOKAHEAD = 2;
    prefetch_slot pre[LOOKAHEAD];
    prefetch_key previous_key;          // of the last whole fused frame with a prepass, whose camera set-up and prepass verdicts are still
    bool previous_key_valid = false;    // in the current buffer set (gr_frame_tuning.reuse_still_camera); false once anybody was handed a buffer
    hipStream_t previous_stream = nullptr;
    // Prepass policy (use_prepass = -2, whole frames on the fused path; opt-in: see the last sentence).  The prepass pays for itself through the pixels it lets the
    // trace skip; where it skips next to nothing (Kerr with a = 0.9 in the script's units: a naked singularity, no shadow - 8.4 ms of
    // single-ray latency in front of every 4K frame, for nothing) it is left out: its flags are copied to the host after a frame
    // that ran it, read a frame or two later without waiting, and when fewer than PREPASS_MIN_SKIP of the cells have their whole
    // 5-point stencil marked (the share of the pixels the trace may skip) the next PREPASS_HOLIDAY frames go without one; then it
    // is tried again.  Not the default, because it is not quite neutral: a pixel the prepass skips is black by decree (its five cells'
    // rays were lost), and traced on its own its ray may still find a way out in a chaotic region - measured on the a = 0.9 frame:
    // the pixels that differ are among the < 2 % the prepass would have skipped (tests/test_gpu_schedule.py).
    struct prepass_policy {
        int* host_flags = nullptr;   // pinned
        size_t capacity = 0, cells = 0;
        int grid_width = 0;
        hipEvent_t copied = nullptr;
        bool in_flight = false;
        int holiday = 0;
        float last_fraction = -1.f;
        unsigned long long with_prepass = 0, without_prepass = 0;
    } policy;
    static constexpr float PREPASS_MIN_SKIP = 0.02f;
    static const int PREPASS_HOLIDAY = 30;
    hipEvent_t main_mark = nullptr;
    unsigned long long frame_counter = 0;
    unsigned long long policy_program = 0;
    // time_kernels == 2: one event pair per trace launch, kept until gr_render_state_trace_log collects them (frames of
    // several states overlap on the GPU in pipelined rendering, so "the last frame" is not a representative sample)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> trace_log;
    size_t trace_log_used = 0;

    void swap_in(camera_set& o) {
        std::swap(camera_pos_cart, o.camera_pos_cart);
        std::swap(camera_quat, o.camera_quat);
        std::swap(camera_pos_generic, o.camera_pos_generic);
        for (int i = 0; i < 4; i++) std::swap(tetrad[i], o.tetrad[i]);
        std::swap(termination_buffer, o.termination_buffer);
        std::swap(tile_order, o.tile_order);
    }
};

static const int CFG_MAX = 64;

// A snapshot of the camera's own timelike geodesic (main.cpp:1232-1242 buffers, :2675-2760 snapshot): path, velocity and
// proper-time step per sample, the four tetrad legs parallel transported along it, all resident on the device.
struct gr_geodesic_camera {
    int device = 0;
    upload_ring uploads;
    int max_path_length = 0;
    void* path = nullptr;        // float4[max]
    void* velocity = nullptr;    // float4[max]
    void* ds = nullptr;          // float[max]
    void* count = nullptr;       // int
    void* transported[4] = {};   // float4[max] each
    void* ray = nullptr;         // lightray
    void* ray_count = nullptr;   // int
    void* basis_speed = nullptr; // float4
    void* camera_generic = nullptr;
    void* tetrad[4] = {};
    void* interpolated_velocity = nullptr;
    void* cfg = nullptr;
    void* dfg = nullptr;
    int steps = 0;
    float proper_time = 0;
};

extern "C" {

// Where a camera sees the coordinate origin, in pixels, as if space were flat (the inverse of the kernels' pixel_direction).  Between
// two frames of a moving or turning camera the picture of whatever sits there - the hole, the bubble, the throat, which is where the
// dear tiles are - moves by about as much as this point does; false if the origin is behind the camera or the camera sits on it.
static bool origin_on_screen(const gr_camera& c, float fov_degrees, int width, int height, float out[2]) {   // = gr_camera_origin_on_screen
    const double px = c.position[1], py = c.position[2], pz = c.position[3];
    const double r = std::sqrt(px * px + py * py + pz * pz);
    double qx = c.quat[0], qy = c.quat[1], qz = c.quat[2], qw = c.quat[3];
    const double qn = std::sqrt(qx * qx + qy * qy + qz * qz + qw * qw);
    if (!(r > 1e-6) || !(qn > 1e-6)) return false;
    qx = -qx / qn; qy = -qy / qn; qz = -qz / qn; qw /= qn;   // the inverse rotation: world -> camera
    const double d[3] = {-px / r, -py / r, -pz / r};
    const double t[3] = {2 * (qy * d[2] - qz * d[1]), 2 * (qz * d[0] - qx * d[2]), 2 * (qx * d[1] - qy * d[0])};
    const double v[3] = {d[0] + qw * t[0] + (qy * t[2] - qz * t[1]), d[1] + qw * t[1] + (qz * t[0] - qx * t[2]),
                         d[2] + qw * t[2] + (qx * t[1] - qy * t[0])};
    if (!(v[2] > 0.05)) return false;
    const double f_stop = (width / 2.0) / std::tan(fov_degrees / 360.0 * M_PI);
    out[0] = (float)(width / 2.0 + f_stop * v[0] / v[2]);
    out[1] = (float)(height / 2.0 + f_stop * v[1] / v[2]);
    return std::isfinite(out[0]) && std::isfinite(out[1]);
}

// An upper estimate of how many pixels the picture moves between two cameras: the angle between the two orientations and the
// parallax of the origin, at the focal length.  A history the picture has moved more than 48 px away from is not followed: a wrong
// order is worse than none (a camera rolling 5 degrees a frame, 170 px at the edge: 4K Kerr 7.8 -> 13.4 ms, a = 0.9 27 -> 85 ms following
// it blindly; up to 43 px - 0.08 units sideways or 1 degree of roll a frame - it measured a gain or nothing).
// the two parts on their own: a turn of the camera moves the picture rigidly (the history's shift follows it), a step moves it by parallax
static bool picture_motion_parts(const gr_camera& a, const gr_camera& b, float fov_degrees, int width, float& turn_px, float& parallax_px) {
    double dot = 0, na = 0, nb = 0, dp = 0, r = 0;
    for (int i = 0; i < 4; i++) { dot += (double)a.quat[i] * b.quat[i]; na += (double)a.quat[i] * a.quat[i]; nb += (double)b.quat[i] * b.quat[i]; }
    for (int i = 1; i < 4; i++) { dp += ((double)a.position[i] - b.position[i]) * ((double)a.position[i] - b.position[i]); r += (double)b.position[i] * b.position[i]; }
    turn_px = parallax_px = 1e9f;
    if (!(na > 0) || !(nb > 0)) return false;
    if (a.flip != b.flip || memcmp(a.basis_speed, b.basis_speed, sizeof(a.basis_speed)) != 0) return false;
    const double c = std::min(1.0, std::fabs(dot) / std::sqrt(na * nb));
    const double f_stop = (width / 2.0) / std::tan(fov_degrees / 360.0 * M_PI);
    const double turn = 2 * std::acos(c) * f_stop, parallax = std::sqrt(dp) / std::max(std::sqrt(r), 1e-3) * f_stop;
    if (!std::isfinite(turn) || !std::isfinite(parallax)) return false;
    turn_px = (float)turn; parallax_px = (float)parallax;
    return true;
}

static float picture_motion(const gr_camera& a, const gr_camera& b, float fov_degrees, int width) {
    double dot = 0, na = 0, nb = 0, dp = 0, r = 0;
    for (int i = 0; i < 4; i++) { dot += (double)a.quat[i] * b.quat[i]; na += (double)a.quat[i] * a.quat[i]; nb += (double)b.quat[i] * b.quat[i]; }
    for (int i = 1; i < 4; i++) { dp += ((double)a.position[i] - b.position[i]) * ((double)a.position[i] - b.position[i]); r += (double)b.position[i] * b.position[i]; }
    if (!(na > 0) || !(nb > 0)) return 1e9f;
    const double c = std::min(1.0, std::fabs(dot) / std::sqrt(na * nb));
    const double f_stop = (width / 2.0) / std::tan(fov_degrees / 360.0 * M_PI);
    const double motion = (2 * std::acos(c) + std::sqrt(dp) / std::max(std::sqrt(r), 1e-3)) * f_stop;
    if (a.flip != b.flip || memcmp(a.basis_speed, b.basis_speed, sizeof(a.basis_speed)) != 0) return 1e9f;
    return std::isfinite(motion) ? (float)motion : 1e9f;
}

int gr_render_state_tile_history(gr_render_state* s, unsigned long long* frames_recorded, unsigned long long* frames_followed, int last_shift[2]) {
    if (!s) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "null argument");
    if (frames_recorded) *frames_recorded = s->history_recorded;
    if (frames_followed) *frames_followed = s->history_followed;
    if (last_shift) { last_shift[0] = s->history_last_shift[0]; last_shift[1] = s->history_last_shift[1]; }
    return GR_OK;
}

int gr_render_state_prepass_reused(gr_render_state* s, unsigned long long* frames) {
    if (!s || !frames) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "null argument");
    *frames = s->prepass_reused;
    return GR_OK;
}

int gr_camera_origin_on_screen(const gr_camera* camera, float field_of_view, int width, int height, float pixel_out[2]) {
    if (!camera || !pixel_out || width <= 0 || height <= 0) return 0;
    return origin_on_screen(*camera, field_of_view, width, height, pixel_out) ? 1 : 0;
}

float gr_picture_motion(const gr_camera* from, const gr_camera* to, float field_of_view, int width) {
    if (!from || !to || width <= 0) return 1e9f;
    return picture_motion(*from, *to, field_of_view, width);
}

// Is an earlier fused frame still on this device when the next one is submitted?  (What tile_history's default asks: a frame that
// has the device to itself ends when its last tile ends, and the order of its tiles decides when that is; frames that overlap fill
// each other's tails, and there the order measured 3-4 % slower than image order.)  The end of every fused frame is marked with an
// event; the question is whether the latest such mark has been reached.
namespace {
struct device_activity {
    std::mutex lock;
    struct mark { hipEvent_t reached = nullptr; hipStream_t stream = nullptr; bool set = false; } marks[4];
    unsigned int used = 0;
};
device_activity g_activity[64];

// Frames on the caller's own stream do not count: they run one after the other whatever the host does, so a host that submits
// its next frame while the last one is still running - on the same stream - has the device to itself per frame all the same.
bool earlier_frame_still_running(int device, hipStream_t stream) {
    if (device < 0 || device >= 64) return false;
    auto& a = g_activity[device];
    std::lock_guard<std::mutex> hold(a.lock);
    for (auto& m : a.marks) {
        if (!m.set || m.stream == stream) continue;
        const hipError_t e = hipEventQuery(m.reached);
        if (e == hipErrorNotReady) { (void)hipGetLastError(); return true; }
        m.set = false;   // reached (or unusable): nothing to ask again
    }
    return false;
}

void mark_frame_end(int device, hipStream_t stream) {
    if (device < 0 || device >= 64) return;
    auto& a = g_activity[device];
    std::lock_guard<std::mutex> hold(a.lock);
    // the stream's own slot if it has one (one mark per stream is enough: its latest), else the oldest
    auto* slot = &a.marks[a.used % 4];
    for (auto& m : a.marks)
        if (m.set && m.stream == stream) { slot = &m; break; }
    if (!slot->reached && hipEventCreateWithFlags(&slot->reached, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        slot->reached = nullptr;
        return;
    }
    if (hipEventRecord(slot->reached, stream) == hipSuccess) {
        if (slot == &a.marks[a.used % 4]) a.used++;
        slot->stream = stream;
        slot->set = true;
    } else {
        (void)hipGetLastError();
    }
}
}   // namespace

void gr_camera_default(gr_camera* c) {
    if (!c) return;
    // camera::camera(), main.cpp:669-673: rot.load_from_axis_angle({1, 0, 0, -pi/2})
    c->position[0] = 0; c->position[1] = 0; c->position[2] = -4; c->position[3] = 0;
    float half = (float)(-M_PI / 2) / 2;
    c->quat[0] = std::sin(half); c->quat[1] = 0; c->quat[2] = 0; c->quat[3] = std::cos(half);
    c->basis_speed[0] = c->basis_speed[1] = c->basis_speed[2] = 0;
    c->flip = 0;
}

int gr_render_state_prepass_policy(gr_render_state* s, unsigned long long* frames_with_prepass, unsigned long long* frames_without,
                                   float* last_marked_fraction) {
    if (!s) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "null argument");
    if (frames_with_prepass) *frames_with_prepass = s->policy.with_prepass;
    if (frames_without) *frames_without = s->policy.without_prepass;
    if (last_marked_fraction) *last_marked_fraction = s->policy.last_fraction;
    return GR_OK;
}

void gr_frame_options_default(gr_frame_options* o) {
    if (!o) return;
    o->mode = GR_MODE_FUSED;
    o->tiled = 1;
    o->use_prepass = -1;
    o->max_probes = 8;
    o->strip_rank = 0;
    o->strip_count = 1;
    o->block_rows = 16;
    o->compact_out = 0;
    o->time_kernels = 0;
    o->next_camera = nullptr;
    o->next_camera2 = nullptr;
    o->geodesic = nullptr;
    o->geodesic_time = 0;
    o->parallel_transport_observer = 1;   // main.cpp:1259
    o->tuning = nullptr;
}

void gr_frame_tuning_default(gr_frame_tuning* t) {
    if (!t) return;
    t->ray_compaction = -1;
    t->rays_per_lane = 0;
    t->fused_shading = -1;
    t->inline_prepass = -1;
    t->trace_waves_per_simd = 0;
    t->tile_history = -1;
    t->park_lanes = -1;
    t->park_trips = 0;
    t->next_strip_rank = -1;
    t->next_strip_rank2 = -1;
    t->next_geodesic_time = 0;
    t->next_geodesic_time2 = 0;
    t->count_attempts = 0;
    t->guess_still_camera = -1;
    t->reuse_still_camera = -1;
    t->speculative_classes = -1;
}

int gr_device_count(int* count) {
    if (!count) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "null argument");
    HIP_CHECK(hipGetDeviceCount(count));
    return GR_OK;
}
int gr_device_alloc(int device, size_t bytes, void** out) {
    HIP_CHECK(hipSetDevice(device));
    HIP_CHECK(hipMalloc(out, bytes ? bytes : 1));
    return GR_OK;
}
int gr_device_free(int device, void* ptr) {
    HIP_CHECK(hipSetDevice(device));
    HIP_CHECK(hipFree(ptr));
    return GR_OK;
}
int gr_device_download(int device, void* dst, const void* src, size_t bytes) {
    HIP_CHECK(hipSetDevice(device));
    HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return GR_OK;
}
int gr_device_upload(int device, void* dst, const void* src, size_t bytes) {
    HIP_CHECK(hipSetDevice(device));
    HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return GR_OK;
}
int gr_stream_create(int device, int high_priority, void** out) {
    if (!out) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "null argument");
    HIP_CHECK(hipSetDevice(device));
    hipStream_t s = nullptr;
    if (high_priority) {
        int least = 0, greatest = 0;
        HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_CHECK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, greatest));
    } else {
        HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    }
    *out = (void*)s;
    return GR_OK;
}
int gr_stream_synchronize(void* stream) {
    HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return GR_OK;
}
int gr_stream_destroy(void* stream) {
    if (stream) HIP_CHECK(hipStreamDestroy((hipStream_t)stream));
    return GR_OK;
}
int gr_device_synchronize(int device) {
    HIP_CHECK(hipSetDevice(device));
    HIP_CHECK(hipDeviceSynchronize());
    return GR_OK;
}

int gr_render_state_create(int device, int width, int height, gr_render_state** out) {
    if (!out || width <= 0 || height <= 0) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "bad render state size");
    HIP_CHECK(hipSetDevice(device));
    gr_render_state* s = new gr_render_state();
    s->device = device;
    s->width = width;
    s->height = height;
    auto alloc = [&](void** p, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(p, bytes);
        if (e == hipSuccess) e = hipMemset(*p, 0, bytes);
        return e;
    };
    hipError_t e = hipSuccess;
    auto A = [&](void** p, size_t bytes) { if (e == hipSuccess) e = alloc(p, bytes); };
    A(&s->camera_pos_cart, 16);
    A(&s->camera_quat, 16);
    A(&s->camera_pos_generic, 16);
    for (auto& t : s->tetrad) A(&t, 16);
    A(&s->rays_count_in, 4);
    A(&s->rays_adaptive_count, 4);
    A(&s->render_data_count, 4);
    A(&s->cfg, CFG_MAX * sizeof(float));
    A(&s->dfg, sizeof(gr_features));
    A(&s->attempts, GR_COUNTER_WORDS * 8);
    size_t px = (size_t)width * height;
    A(&s->render_data, px * sizeof(gr_render_data));
    A(&s->termination_buffer, px * sizeof(int));
    // two words per tile: 8x8 tiles of the whole image or, split over devices, of at most all its blocks + their halo pieces
    const size_t order_bytes = (px / 16 + 2 * (size_t)width + 8192) * sizeof(unsigned int);
    A(&s->tile_order, order_bytes);
    s->tile_order_bytes = order_bytes;
    A(&s->tile_cost, order_bytes / 2);
    for (auto& slot : s->pre) {
        A(&slot.set.camera_pos_cart, 16);
        A(&slot.set.camera_quat, 16);
        A(&slot.set.camera_pos_generic, 16);
        for (auto& t : slot.set.tetrad) A(&t, 16);
        A(&slot.set.termination_buffer, px * sizeof(int));
        A(&slot.set.tile_order, order_bytes);
        A(&slot.velocity, 16);
    }
    if (e == hipSuccess) {
        // High priority: the look-ahead prepass is a latency-bound launch of a few hundred waves that must make progress
        // while the trace kernel occupies every CU, and priority streams get hardware queues of their own (with the
        // default priority the stream can share a queue with the caller's stream once a framework - torch + RCCL - has
        // created a few streams of its own, which serialises the overlap: measured 6.66 -> 7.98 ms per 4K frame).
        int least = 0, greatest = 0;
        e = hipDeviceGetStreamPriorityRange(&least, &greatest);
        for (auto& slot : s->pre) {
            if (e == hipSuccess) e = hipStreamCreateWithPriority(&slot.stream, hipStreamNonBlocking, greatest);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&slot.ready, hipEventDisableTiming);
        }
    }
    // hipMemset returns before the device has done it, and the null stream it is ordered on does not hold back the non-blocking
    // streams frames are submitted on: a state's first frame could be overtaken by its own zeroing (seen with eight processes
    // sharing one GPU: a share of a state's first frame rendered from a zeroed camera and features)
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->main_mark, hipEventDisableTiming);
    for (int i = 0; i < GR_STAGE_COUNT && e == hipSuccess; i++) {
        e = hipEventCreate(&s->ev_start[i]);
        if (e == hipSuccess) e = hipEventCreate(&s->ev_stop[i]);
    }
    if (e != hipSuccess) {
        gr_render_state_destroy(s);
        return gr_internal_fail(GR_ERROR_DEVICE, (std::string("render state allocation: ") + hipGetErrorString(e)).c_str());
    }
    *out = s;
    return GR_OK;
}

void gr_render_state_destroy(gr_render_state* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    (void)hipDeviceSynchronize();   // (uploads from the pinned ring may still be queued)
    s->uploads.release();
    std::vector<void*> ptrs = {s->camera_pos_cart, s->camera_quat, s->camera_pos_generic, s->tetrad[0], s->tetrad[1], s->tetrad[2],
                               s->tetrad[3], s->rays_count_in, s->rays_adaptive_count, s->render_data_count, s->cfg, s->dfg,
                               s->attempts, s->rays_in, s->rays_adaptive, s->render_data, s->termination_buffer, s->tile_order,
                               s->tile_cost, s->lattice_rays, s->guessed[0], s->guessed[1], s->pending_list, s->block_cost, s->block_cost_before, s->ref_cost[0], s->ref_cost[1], s->ref_order, s->ref_sort_work, s->parking_records, s->parking_words};
    for (auto& slot : s->pre) {
        if (slot.stream) { (void)hipStreamSynchronize(slot.stream); (void)hipStreamDestroy(slot.stream); }
        if (slot.ready) (void)hipEventDestroy(slot.ready);
        ptrs.insert(ptrs.end(), {slot.set.camera_pos_cart, slot.set.camera_quat, slot.set.camera_pos_generic, slot.set.tetrad[0],
                                 slot.set.tetrad[1], slot.set.tetrad[2], slot.set.tetrad[3], slot.set.termination_buffer, slot.set.tile_order,
                                 slot.velocity});
    }
    if (s->main_mark) (void)hipEventDestroy(s->main_mark);
    if (s->policy.copied) (void)hipEventDestroy(s->policy.copied);
    if (s->policy.host_flags) (void)hipHostFree(s->policy.host_flags);
    for (auto& pr : s->trace_log) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (int i = 0; i < GR_STAGE_COUNT; i++) {
        if (s->ev_start[i]) (void)hipEventDestroy(s->ev_start[i]);
        if (s->ev_stop[i]) (void)hipEventDestroy(s->ev_stop[i]);
    }
    delete s;
}

// (for csrc/tiled.cpp: a participant checks that the state it is handed is of its frame's size)
extern "C" int gr_internal_render_state_size(const gr_render_state* s, int* width, int* height) {
    if (!s) return 0;
    if (width) *width = s->width;
    if (height) *height = s->height;
    return 1;
}

void* gr_render_state_buffer(gr_render_state* s, int which) {
    if (!s) return nullptr;
    // whoever holds a pointer to the camera set, the prepass verdicts or the parameters may write through it: the next frame does its
    // own set-up and prepass (reuse_still_camera); ray and render-data records are outputs of every frame
    if (which != GR_BUF_RAYS_IN && which != GR_BUF_RAYS_COUNT && which != GR_BUF_RENDER_DATA && which != GR_BUF_RAYS_ADAPTIVE &&
        which != GR_BUF_RAYS_ADAPTIVE_COUNT)
        s->previous_key_valid = false;
    switch (which) {
        case GR_BUF_RAYS_IN: return s->rays_in;
        case GR_BUF_RAYS_COUNT: return s->rays_count_in;
        case GR_BUF_RENDER_DATA: return s->render_data;
        case GR_BUF_TERMINATION: return s->termination_buffer;
        case GR_BUF_CAMERA_GENERIC: return s->camera_pos_generic;
        case GR_BUF_TETRAD0: return s->tetrad[0];
        case GR_BUF_TETRAD1: return s->tetrad[1];
        case GR_BUF_TETRAD2: return s->tetrad[2];
        case GR_BUF_TETRAD3: return s->tetrad[3];
        case GR_BUF_RAYS_ADAPTIVE: return s->rays_adaptive;
        case GR_BUF_RAYS_ADAPTIVE_COUNT: return s->rays_adaptive_count;
        case GR_BUF_CFG: return s->cfg;
        case GR_BUF_DFG: return s->dfg;
        case GR_BUF_CAMERA_QUAT: return s->camera_quat;
    }
    return nullptr;
}

int gr_render_state_stage_ms(gr_render_state* s, int stage, float* ms) {
    if (!s || !ms || stage < 0 || stage >= GR_STAGE_COUNT) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "bad stage");
    *ms = 0;
    if (!s->stage_timed[stage]) return GR_OK;
    HIP_CHECK(hipEventSynchronize(s->ev_stop[stage]));
    HIP_CHECK(hipEventElapsedTime(ms, s->ev_start[stage], s->ev_stop[stage]));
    return GR_OK;
}

int gr_render_state_trace_log(gr_render_state* s, float* total_ms, int* launches, int reset) {
    if (!s) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "null argument");
    HIP_CHECK(hipSetDevice(s->device));
    double sum = 0;
    for (size_t i = 0; i < s->trace_log_used; i++) {
        float ms = 0;
        HIP_CHECK(hipEventSynchronize(s->trace_log[i].second));
        HIP_CHECK(hipEventElapsedTime(&ms, s->trace_log[i].first, s->trace_log[i].second));
        sum += ms;
    }
    if (total_ms) *total_ms = (float)sum;
    if (launches) *launches = (int)s->trace_log_used;
    if (reset) s->trace_log_used = 0;
    return GR_OK;
}

int gr_render_state_shader_clock(gr_render_state* s, double* mhz) {
    if (!s || !mhz) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "null argument");
    HIP_CHECK(hipSetDevice(s->device));
    unsigned long long v[4] = {};
    HIP_CHECK(hipMemcpy(v, s->attempts, 32, hipMemcpyDeviceToHost));
    *mhz = v[2] ? 100.0 * (double)v[1] / (double)v[2] : 0.0;   // cycles per tick of the 100 MHz reference clock
    return GR_OK;
}

int gr_render_state_wave_time(gr_render_state* s, double* wave_ms, unsigned long long* waves) {
    if (!s || !wave_ms || !waves) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "null argument");
    HIP_CHECK(hipSetDevice(s->device));
    unsigned long long v[4] = {};
    HIP_CHECK(hipMemcpy(v, s->attempts, 32, hipMemcpyDeviceToHost));
    *wave_ms = (double)v[2] * 1e-5;   // ticks of 10 ns
    *waves = v[3];
    return GR_OK;
}

int gr_render_state_counters(gr_render_state* s, unsigned long long* words, int count) {
    if (!s || !words || count < 0 || count > 256) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "counters: up to 256 words");
    HIP_CHECK(hipSetDevice(s->device));
    HIP_CHECK(hipMemcpy(words, s->attempts, (size_t)count * 8, hipMemcpyDeviceToHost));
    return GR_OK;
}

int gr_render_state_attempts(gr_render_state* s, unsigned long long* attempts) {
    if (!s || !attempts) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "null argument");
    HIP_CHECK(hipSetDevice(s->device));
    std::vector<unsigned long long> words(GR_COUNTER_WORDS);
    HIP_CHECK(hipMemcpy(words.data(), s->attempts, words.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long sum = words[0];   // the reference-shaped, pair and compaction kernels count here
    for (int i = 256; i < GR_COUNTER_WORDS; i++) sum += words[i];   // gr_trace_fused: spread over 256 words (kernels/program.hip)
    *attempts = sum;
    return GR_OK;
}

static int ensure_rays(gr_render_state* s, size_t slots, bool adaptive) {
    if (s->ray_capacity < slots) {
        if (s->rays_in) (void)hipFree(s->rays_in);
        if (s->rays_adaptive) (void)hipFree(s->rays_adaptive);
        s->rays_in = s->rays_adaptive = nullptr;
        s->ray_capacity = 0;
        HIP_CHECK(hipMalloc(&s->rays_in, slots * sizeof(gr_lightray)));
        s->ray_capacity = slots;
    }
    if (adaptive && !s->rays_adaptive) HIP_CHECK(hipMalloc(&s->rays_adaptive, s->ray_capacity * sizeof(gr_lightray)));
    return GR_OK;
}

// ---- camera on a timelike geodesic ------------------------------------------------------------------
int gr_geodesic_camera_create(int device, int max_path_length, gr_geodesic_camera** out) {
    if (!out || max_path_length < 2) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "bad geodesic camera size");
    HIP_CHECK(hipSetDevice(device));
    gr_geodesic_camera* g = new gr_geodesic_camera();
    g->device = device;
    g->max_path_length = max_path_length;
    hipError_t e = hipSuccess;
    auto A = [&](void** p, size_t bytes) {
        if (e == hipSuccess) e = hipMalloc(p, bytes);
        if (e == hipSuccess) e = hipMemset(*p, 0, bytes);
    };
    size_t n = (size_t)max_path_length;
    A(&g->path, n * 16); A(&g->velocity, n * 16); A(&g->ds, n * 4); A(&g->count, 4);
    for (auto& t : g->transported) A(&t, n * 16);
    A(&g->ray, sizeof(gr_lightray)); A(&g->ray_count, 4); A(&g->basis_speed, 16); A(&g->camera_generic, 16);
    for (auto& t : g->tetrad) A(&t, 16);
    A(&g->interpolated_velocity, 16);
    A(&g->cfg, CFG_MAX * sizeof(float)); A(&g->dfg, sizeof(gr_features));
    if (e == hipSuccess) e = hipDeviceSynchronize();   // (the zeroing above must not overtake the first snapshot: see gr_render_state_create)
    if (e != hipSuccess) {
        gr_geodesic_camera_destroy(g);
        return gr_internal_fail(GR_ERROR_DEVICE, (std::string("geodesic camera allocation: ") + hipGetErrorString(e)).c_str());
    }
    *out = g;
    return GR_OK;
}

void gr_geodesic_camera_destroy(gr_geodesic_camera* g) {
    if (!g) return;
    (void)hipSetDevice(g->device);
    (void)hipDeviceSynchronize();
    g->uploads.release();
    void* ptrs[] = {g->path, g->velocity, g->ds, g->count, g->transported[0], g->transported[1], g->transported[2], g->transported[3],
                    g->ray, g->ray_count, g->basis_speed, g->camera_generic, g->tetrad[0], g->tetrad[1], g->tetrad[2], g->tetrad[3],
                    g->interpolated_velocity, g->cfg, g->dfg};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    delete g;
}

int gr_geodesic_camera_snapshot(gr_geodesic_camera* g, gr_program* p, const gr_metric* m, void* stream_v, const gr_camera* camera,
                                const float geodesic_basis_speed[3], const gr_features* features_in, const float* cfg_values,
                                int num_cfg_values, int* steps_out, float* proper_time_out) {
    if (!g || !p || !m || !camera || !geodesic_basis_speed) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "null argument");
    hipStream_t stream = (hipStream_t)stream_v;
    HIP_CHECK(hipSetDevice(g->device));
    gr_metric_info info;
    GR_CHECK(gr_metric_get_info(m, &info));
    gr_features features;
    gr_features_default(&features);
    if (features_in) features = *features_in;
    else features.max_acceleration_change = info.max_acceleration_change;
    std::vector<float> cfg(info.num_dynamic_vars > 0 ? info.num_dynamic_vars : 1, 0.f);
    for (int i = 0; i < info.num_dynamic_vars; i++)
        cfg[i] = (cfg_values && i < num_cfg_values) ? cfg_values[i] : gr_metric_dynamic_var_default(m, i);
    if ((int)cfg.size() > CFG_MAX) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "too many dynamic variables");
    float speed4[4] = {geodesic_basis_speed[0], geodesic_basis_speed[1], geodesic_basis_speed[2], 0.f};
    if (speed4[0] * speed4[0] + speed4[1] * speed4[1] + speed4[2] * speed4[2] >= 1.f)
        return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "geodesic basis speed must be below c");
    void* cart = g->interpolated_velocity;   // scratch until the first interpolation
    GR_CHECK(g->uploads.copy(g->cfg, cfg.data(), cfg.size() * sizeof(float), stream));
    GR_CHECK(g->uploads.copy(g->dfg, &features, sizeof(features), stream));
    GR_CHECK(g->uploads.copy(g->basis_speed, speed4, 16, stream));
    GR_CHECK(g->uploads.copy(cart, camera->position, 16, stream));
    HIP_CHECK(hipMemsetAsync(g->count, 0, 4, stream));
    HIP_CHECK(hipMemsetAsync(g->ray_count, 0, 4, stream));
    // main.cpp:2311-2329 (this frame's camera), then :2689-2758
    GR_CHECK(gr_cart_to_generic(p, stream, cart, g->camera_generic, 1, camera->flip, g->cfg));
    GR_CHECK(gr_init_basis_vectors(p, stream, g->camera_generic, 1, camera->basis_speed, g->tetrad[0], g->tetrad[1], g->tetrad[2],
                                   g->tetrad[3], g->cfg));
    GR_CHECK(gr_boost_tetrad(p, stream, g->camera_generic, 1, g->basis_speed, g->tetrad[0], g->tetrad[1], g->tetrad[2], g->tetrad[3],
                             g->cfg));
    GR_CHECK(gr_init_inertial_ray(p, stream, g->camera_generic, 1, g->ray, g->ray_count, g->tetrad[0], g->tetrad[1], g->tetrad[2],
                                  g->tetrad[3], g->basis_speed, g->cfg));
    GR_CHECK(gr_get_geodesic_path(p, stream, g->ray, 1, g->path, g->velocity, g->ds, g->ray_count, g->max_path_length, g->cfg, g->dfg,
                                  g->count));
    for (int i = 0; i < 4; i++)
        GR_CHECK(gr_parallel_transport_quantity(p, stream, g->path, g->velocity, g->ds, g->tetrad[i], g->count, 1, g->transported[i],
                                                g->cfg));
    HIP_CHECK(hipMemcpyAsync(&g->steps, g->count, 4, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    std::vector<float> ds((size_t)std::max(g->steps, 1));
    if (g->steps > 0) HIP_CHECK(hipMemcpy(ds.data(), g->ds, (size_t)g->steps * 4, hipMemcpyDeviceToHost));
    double total = 0;
    for (int i = 0; i + 1 < g->steps; i++) total += ds[i];   // the last sample has no successor (cl.cl:2808-2845)
    g->proper_time = (float)total;
    if (steps_out) *steps_out = g->steps;
    if (proper_time_out) *proper_time_out = g->proper_time;
    return GR_OK;
}

int gr_geodesic_camera_interpolate(gr_geodesic_camera* g, gr_program* p, void* stream_v, float proper_time,
                                   int parallel_transport_observer, float camera_generic_out[4], float tetrad_out[16],
                                   float velocity_out[4]) {
    if (!g || !p) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "null argument");
    hipStream_t stream = (hipStream_t)stream_v;
    HIP_CHECK(hipSetDevice(g->device));
    GR_CHECK(gr_handle_interpolating_geodesic(p, stream, g->path, g->velocity, g->ds, g->camera_generic, g->transported[0],
                                              g->transported[1], g->transported[2], g->transported[3], g->tetrad[0], g->tetrad[1],
                                              g->tetrad[2], g->tetrad[3], proper_time, g->count, parallel_transport_observer,
                                              g->basis_speed, g->interpolated_velocity, g->cfg));
    if (camera_generic_out) HIP_CHECK(hipMemcpyAsync(camera_generic_out, g->camera_generic, 16, hipMemcpyDeviceToHost, stream));
    if (velocity_out) HIP_CHECK(hipMemcpyAsync(velocity_out, g->interpolated_velocity, 16, hipMemcpyDeviceToHost, stream));
    if (tetrad_out)
        for (int i = 0; i < 4; i++) HIP_CHECK(hipMemcpyAsync(tetrad_out + 4 * i, g->tetrad[i], 16, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    return GR_OK;
}

void* gr_geodesic_camera_buffer(gr_geodesic_camera* g, int which) {
    if (!g) return nullptr;
    switch (which) {
        case GR_GEOBUF_PATH: return g->path;
        case GR_GEOBUF_VELOCITY: return g->velocity;
        case GR_GEOBUF_DS: return g->ds;
        case GR_GEOBUF_COUNT: return g->count;
        case GR_GEOBUF_TRANSPORTED0: return g->transported[0];
        case GR_GEOBUF_TRANSPORTED1: return g->transported[1];
        case GR_GEOBUF_TRANSPORTED2: return g->transported[2];
        case GR_GEOBUF_TRANSPORTED3: return g->transported[3];
    }
    return nullptr;
}

int gr_render_frame(gr_render_state* s, gr_program* p, const gr_metric* m, void* stream_v, const gr_camera* camera,
                    const gr_features* features_in, const float* cfg_values, int num_cfg_values, const void* bg1,
                    const void* bg2, int bg_width, int bg_height, int bg_levels, void* out, const gr_frame_options* opt_in) {
    if (!s || !p || !m || !camera) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "null argument");
    // a frame that is to be shaded needs both skies and their shape (the texture pass reads them unchecked: a NULL sky is a device fault,
    // not an error code - tests/test_gpu_lifecycle.py found it); out == NULL stops after the render-data
    if (out && (!bg1 || !bg2 || bg_width <= 0 || bg_height <= 0 || bg_levels <= 0))
        return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "gr_render_frame: an output frame needs both background textures and their width, height and levels");
    if (num_cfg_values < 0 || (num_cfg_values > 0 && !cfg_values)) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "gr_render_frame: cfg_values");
    hipStream_t stream = (hipStream_t)stream_v;
    HIP_CHECK(hipSetDevice(s->device));
    gr_frame_options opt;
    gr_frame_options_default(&opt);
    if (opt_in) opt = *opt_in;
    gr_frame_tuning tune;   // which fused kernel, schedule and launch size (geodesic_hip_internal.h; NULL = the defaults)
    gr_frame_tuning_default(&tune);
    if (opt.tuning) tune = *opt.tuning;
    gr_metric_info info;
    GR_CHECK(gr_metric_get_info(m, &info));

    gr_features features;
    gr_features_default(&features);
    if (features_in) features = *features_in;
    else features.max_acceleration_change = info.max_acceleration_change;   // metric_manager.hpp:50

    const int width = s->width, height = s->height;
    // The defaults of gr_features (adaptive_sampling on, as the reference's GUI) and of gr_frame_options (fused mode) work
    // together: a frame is sampled adaptively on the fused path (half-resolution lattice, gr_adaptive_refine, second fused
    // launch over the marked pixels; cl.cl:3234-3250, 5223-5345) - a device's share of a split frame too: it traces the lattice
    // rows its blocks' decisions read (two rows of halo either side) and its rows come out as those of the whole frame.
    bool use_prepass = opt.use_prepass < 0 ? info.use_prepass != 0 : opt.use_prepass != 0;
    bool adaptive = features.adaptive_sampling != 0 && !features.use_triangle_rendering;
    // The 2x2 blocks of handle_adaptive_sampling cover 2 (W/2) x 2 (H/2) pixels (cl.cl:5228-5236): with an odd width or height the
    // last column or row belongs to no block and would keep whatever its record held.  The fused path then traces every pixel.
    if (opt.mode == GR_MODE_FUSED && ((width | height) & 1)) adaptive = false;

    // dynamic_config: $cfg values in declaration order (metric_manager.hpp:60-66)
    std::vector<float> cfg(info.num_dynamic_vars > 0 ? info.num_dynamic_vars : 1, 0.f);
    for (int i = 0; i < info.num_dynamic_vars; i++)
        cfg[i] = (cfg_values && i < num_cfg_values) ? cfg_values[i] : gr_metric_dynamic_var_default(m, i);
    if ((int)cfg.size() > CFG_MAX) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "too many dynamic variables");
    const bool cfg_changed = cfg != s->host_cfg;
    // ... by a step of a slider (every parameter within a tenth of itself): the picture is nearly the one before, and what its tiles cost is
    // still the best estimate there is of what they cost now - orders are only orders, a wrong one costs time, never a record.  A frame of a
    // slider being dragged (the dynamic program, new parameters every frame: metric_manager.hpp:60-66) would otherwise start from image
    // order every time (4K Kerr: 6.0 ms against 5.1).
    const bool cfg_jumped = cfg_changed && [&] {
        if (cfg.size() != s->host_cfg.size()) return true;
        for (size_t i = 0; i < cfg.size(); i++) {
            const float a = cfg[i], b = s->host_cfg[i];
            if (!(std::fabs(a - b) <= 0.1f * std::max(std::max(std::fabs(a), std::fabs(b)), 0.05f))) return true;
        }
        return false;
    }();
    const bool features_changed = !s->features_valid || memcmp(&features, &s->host_features, sizeof(features)) != 0;
    // another metric, parameter set or field of view: what the last frame's tiles cost says nothing about this one's (tile_history)
    if (cfg_jumped || features_changed || gr_program_serial(p) != s->tile_cost_program) {
        s->tile_cost_valid = false;
        s->tile_cost_program = gr_program_serial(p);
    }
    const bool prepass_by_policy = opt.use_prepass == -2 && use_prepass && opt.mode == GR_MODE_FUSED && opt.strip_count <= 1;
    if (prepass_by_policy) {
        auto& pol = s->policy;
        const unsigned long long serial = gr_program_serial(p);
        if (cfg_changed || features_changed || serial != s->policy_program) {   // another metric or parameter set: start over
            pol.holiday = 0;
            pol.last_fraction = -1.f;
            if (pol.in_flight) { HIP_CHECK(hipEventSynchronize(pol.copied)); pol.in_flight = false; }
            s->policy_program = serial;
        }
        if (pol.in_flight && hipEventQuery(pol.copied) == hipSuccess) {
            // the share of the grid whose 5-point stencil is marked throughout: what init_rays_generic's test lets the trace skip
            // (a pixel is skipped when the cell it rounds to and that cell's four neighbours are all marked, cl.cl:3213-3232)
            size_t skippable = 0;
            const int pw = pol.grid_width, ph = pol.cells && pw ? (int)(pol.cells / pw) : 0;
            for (int y = 1; y + 1 < ph; y++)
                for (int x = 1; x + 1 < pw; x++) {
                    const int* c = pol.host_flags + (size_t)y * pw + x;
                    skippable += c[0] == 1 && c[-1] == 1 && c[1] == 1 && c[-pw] == 1 && c[pw] == 1;
                }
            pol.last_fraction = pol.cells ? (float)skippable / (float)pol.cells : 0.f;
            pol.in_flight = false;
            if (pol.last_fraction < gr_render_state::PREPASS_MIN_SKIP) pol.holiday = gr_render_state::PREPASS_HOLIDAY;
        } else if (pol.in_flight) {
            (void)hipGetLastError();   // hipErrorNotReady is not an error
        }
        if (pol.holiday > 0) { pol.holiday--; pol.without_prepass++; use_prepass = false; }
        else pol.with_prepass++;
    }
    if (cfg_changed || features_changed) {
        // Look-ahead prepasses read s->cfg / s->dfg on their side streams: the upload below must not overtake one that is still
        // running (it would read a mix of old and new parameters), and what the slots hold was computed for the old parameters.
        for (auto& slot : s->pre) {
            if (!slot.valid) continue;
            HIP_CHECK(hipStreamWaitEvent(stream, slot.ready, 0));
            slot.valid = false;
        }
    }
    if (cfg_changed) {
        GR_CHECK(s->uploads.copy(s->cfg, cfg.data(), cfg.size() * sizeof(float), stream));
        s->host_cfg = cfg;
    }
    if (features_changed) {
        GR_CHECK(s->uploads.copy(s->dfg, &features, sizeof(features), stream));
        s->host_features = features;
        s->features_valid = true;
    }
    int prepass_width = width / 16, prepass_height = height / 16;   // main.cpp:2380-2381
    if (prepass_width < 1 || prepass_height < 1) use_prepass = false;

    // was this frame's camera set-up + prepass already done on a side stream during an earlier frame?
    auto make_key = [&](const gr_camera* c, float time, int frame_strip_rank) {
        gr_render_state::prefetch_key k;
        k.camera = *c;
        k.cfg = cfg;
        k.features = features;
        k.program = gr_program_serial(p);
        k.geodesic = (const void*)opt.geodesic;
        k.geodesic_time = time;
        k.transport = opt.parallel_transport_observer;
        k.strip[2] = opt.strip_count > 1 ? opt.strip_count : 1;
        k.strip[1] = k.strip[2] > 1 ? frame_strip_rank : 0;
        k.strip[0] = k.strip[2] > 1 ? opt.block_rows : 0;
        return k;
    };
    s->frame_counter++;
    bool prefetched = false;
    if (opt.mode == GR_MODE_FUSED && use_prepass) {
        const auto want = make_key(camera, opt.geodesic_time, opt.strip_rank);
        gr_render_state::prefetch_slot* hit = nullptr;   // the oldest matching prefetch: it has had the most time to finish
        for (auto& slot : s->pre)
            if (slot.valid && slot.key == want && (!hit || slot.age < hit->age)) hit = &slot;
        if (hit) {
            s->swap_in(hit->set);
            HIP_CHECK(hipStreamWaitEvent(stream, hit->ready, 0));
            hit->valid = false;
            prefetched = true;
        }
    }
    // ... or is it the previous frame of this state over again - camera, parameters, features, program, bit for bit, on the same stream
    // (gr_frame_tuning.reuse_still_camera)?  Camera position, tetrad and the prepass verdicts are functions of exactly those, and they are
    // where that frame left them: a viewer whose user has stopped moving pays neither again (the reference does, every frame:
    // main.cpp:2311-2437).  Whole frames with a Cartesian camera whose tiles are not ordered by the prepass rays' costs.
    static const int tile_order_mode = [] { const char* e = getenv("GR_TILE_ORDER"); return !e ? -1 : e[0] == '0' ? 0 : 1; }();
    static const int reuse_default = [] { const char* e = getenv("GR_REUSE_STILL_CAMERA"); return !e ? 1 : e[0] != '0'; }();
    const bool repeats_previous_frame = opt.mode == GR_MODE_FUSED && use_prepass && !opt.geodesic && opt.strip_count <= 1 && s->previous_key_valid &&
                                        s->previous_key == make_key(camera, opt.geodesic_time, opt.strip_rank);
    const bool reuse_on = (tune.reuse_still_camera < 0 ? reuse_default : tune.reuse_still_camera) != 0 && tile_order_mode != 1;
    if (!prefetched && repeats_previous_frame && reuse_on && s->previous_stream == stream) {
        prefetched = true;
        s->prepass_reused++;
    }
    s->previous_key_valid = false;   // until this frame has left its own set-up and prepass behind
    if (!prefetched) {
        GR_CHECK(s->uploads.copy(s->camera_pos_cart, camera->position, 16, stream));
        GR_CHECK(s->uploads.copy(s->camera_quat, camera->quat, 16, stream));
    }

    for (int i = 0; i < GR_STAGE_COUNT; i++) s->stage_timed[i] = false;
    const bool log_trace = opt.time_kernels == 2;
    if (log_trace && s->trace_log_used == s->trace_log.size()) {
        hipEvent_t a = nullptr, b = nullptr;
        HIP_CHECK(hipEventCreate(&a));
        HIP_CHECK(hipEventCreate(&b));
        s->trace_log.emplace_back(a, b);
    }
    auto begin = [&](int st) -> int {
        if (log_trace) { if (st == GR_STAGE_TRACE) HIP_CHECK(hipEventRecord(s->trace_log[s->trace_log_used].first, stream)); }
        else if (opt.time_kernels) { HIP_CHECK(hipEventRecord(s->ev_start[st], stream)); }
        return GR_OK;
    };
    auto end = [&](int st) -> int {
        if (log_trace) { if (st == GR_STAGE_TRACE) { HIP_CHECK(hipEventRecord(s->trace_log[s->trace_log_used].second, stream)); s->trace_log_used++; } }
        else if (opt.time_kernels) { HIP_CHECK(hipEventRecord(s->ev_stop[st], stream)); s->stage_timed[st] = true; }
        return GR_OK;
    };
    void* attempts = nullptr;
    if (tune.count_attempts) {
        HIP_CHECK(hipMemsetAsync(s->attempts, 0, GR_COUNTER_WORDS * 8, stream));
        attempts = s->attempts;
    }

    // camera position and tetrad: from the cartesian camera (main.cpp:2311, 2329), or - camera on a geodesic - interpolated
    // from the snapshot at the requested proper time (main.cpp:2264-2293)
    const gr_geodesic_camera* gc = opt.geodesic;
    if (gc && gc->device != s->device) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "geodesic camera lives on another device");
    auto camera_setup = [&](hipStream_t st, void* cart, void* generic, void* const* tetrad, const gr_camera* cam, float time,
                            void* velocity_out) -> int {
        if (gc)
            return gr_handle_interpolating_geodesic(p, st, gc->path, gc->velocity, gc->ds, generic, gc->transported[0], gc->transported[1],
                                                    gc->transported[2], gc->transported[3], tetrad[0], tetrad[1], tetrad[2], tetrad[3],
                                                    time, gc->count, opt.parallel_transport_observer, gc->basis_speed, velocity_out,
                                                    s->cfg);
        GR_CHECK(gr_cart_to_generic(p, st, cart, generic, 1, cam->flip, s->cfg));
        return gr_init_basis_vectors(p, st, generic, 1, cam->basis_speed, tetrad[0], tetrad[1], tetrad[2], tetrad[3], s->cfg);
    };
    // fused mode with a Cartesian camera: camera set-up and prepass are one call (gr_camera_prepass), issued below
    const bool one_launch_setup = opt.mode == GR_MODE_FUSED && !gc;
    if (!prefetched && !one_launch_setup) {
        GR_CHECK(begin(GR_STAGE_CAMERA));
        GR_CHECK(camera_setup(stream, s->camera_pos_cart, s->camera_pos_generic, s->tetrad, camera, opt.geodesic_time,
                              gc ? gc->interpolated_velocity : nullptr));
        GR_CHECK(end(GR_STAGE_CAMERA));
    }

    if (opt.mode == GR_MODE_FUSED) {
        int strip_count = opt.strip_count > 1 ? opt.strip_count : 1;
        int strip_rank = strip_count > 1 ? opt.strip_rank : 0;
        int block_rows = strip_count > 1 ? opt.block_rows : ((height + 7) / 8) * 8;
        // The prepass rays' costs (kept behind the prepass flags in the termination buffer, which is allocated per pixel) order the
        // tiles of the trace, longest first (gr_order_tiles).
        // Default: on a device's share of a split frame (+8 % with three frames in flight, +35 % one frame at a time, one of 8
        // devices), not on a whole frame (there image order measured 2 % faster); GR_TILE_ORDER=0 never, =1 always.
        // The other source of an order: what the tiles of this state's previous frame cost (gr_order_tiles_by_history).
        static const int history_default = [] { const char* e = getenv("GR_TILE_HISTORY"); return !e ? GR_DEFAULT_TILE_HISTORY : e[0] != '0'; }();
        // Default: whole frames that find the device idle when they are submitted (they record their costs, and follow those of
        // the frame before if that one did too).
        // Not frames of more than 32 tiles per wave slot (8K Alcubierre: 72 short tiles of much the same cost; recording and sorting
        // them measured +3 % on the frame, with nothing to gain).
        // An adaptively sampled whole frame: the same for its lattice launch - the tiles of the half-resolution grid, their costs left by the
        // lattice launch of the frame before (GR_LATTICE_HISTORY=0: image order as before round 6's fifth session).  Only for the metrics with a
        // prepass - the ones with a shadow and long rays along its edge: there the launch gains by its speculative tiles when it traces
        // its own cells, and by the order alone where it is long (4K Kerr a = 0.9, prepass reused: 12.2 -> 9.1 ms); recording + sorting cost
        // the 0.4 ms frames of the metrics without a prepass 5-10 %.
        static const bool lattice_history = [] { const char* e = getenv("GR_LATTICE_HISTORY"); return !(e && e[0] == '0'); }();
        const int hist_width = adaptive ? width / 2 : width, hist_height = adaptive ? height / 2 : height;
        const int hist_block_rows = adaptive ? ((hist_height + 7) / 8) * 8 : block_rows;
        const long long tile_words = gr_tile_order_bytes(hist_width, hist_height, hist_block_rows, strip_rank, strip_count) / 8;
        const bool history_wanted = (tune.tile_history < 0 ? history_default != 0 && strip_count == 1 && tile_words <= 32 * gr_trace_fused_wave_slots(p)
                                                          : tune.tile_history != 0) && (!adaptive || (lattice_history && strip_count == 1 && use_prepass)) &&
                                    (size_t)gr_tile_order_bytes(hist_width, hist_height, hist_block_rows, strip_rank, strip_count) <= s->tile_order_bytes;
        // (the pixels traced ahead for the second launch of adaptive sampling - below - are for such lone frames too, prepass or not)
        static const bool guess_default = [] { const char* e = getenv("GR_ADAPTIVE_GUESS"); return !(e && e[0] == '0'); }();
        const bool guesses_wanted = guess_default && adaptive && strip_count == 1 && !opt.geodesic && tune.tile_history != 0 && history_default != 0;
        const bool device_busy = (history_wanted || guesses_wanted) && tune.tile_history < 0 && earlier_frame_still_running(s->device, stream);
        const bool tile_order_enabled = !history_wanted && (tile_order_mode == 1 || (tile_order_mode == -1 && strip_count > 1));
        const size_t cells = use_prepass ? (size_t)prepass_width * prepass_height : 0;
        // (a frame whose prepass rides in its trace launch - below - has no costs to order by; the frames it announces still do)
        const bool order_capable = tile_order_enabled && use_prepass && !adaptive && 2 * cells <= (size_t)width * height &&
                                   (size_t)gr_tile_order_bytes(width, height, block_rows, strip_rank, strip_count) <= s->tile_order_bytes;
        const int prepass_margin = adaptive ? 2 : 0;   // the lattice rows beyond a block that its 2x2 decisions read
        auto cost_plane = [&](void* termination_buffer) -> void* { return order_capable ? (void*)((unsigned int*)termination_buffer + cells) : nullptr; };
        // which trace kernel this frame takes (needed here already: the prepass may ride in the trace launch)
        // library default: no compaction (the benchmark workloads keep > 95 % of their lanes busy without it); experiments can
        // switch it on for every frame with GR_TRACE_COMPACT=<keep_lanes>
        static const int default_compaction = [] { const char* e = getenv("GR_TRACE_COMPACT"); int v = e ? atoi(e) : 0; return (v >= 1 && v <= 64) ? v : 0; }();
        int keep_lanes = tune.ray_compaction < 0 ? default_compaction : tune.ray_compaction;
        // What does not combine is refused, not silently dropped: ray compaction and the two-rays-per-lane kernel trace every
        // pixel (no lattice / pending-only form), in-tile shading needs every pixel's record in its own tile's wave.
        if (adaptive && tune.ray_compaction > 0)
            return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "ray_compaction > 0 with adaptive sampling: gr_trace_compact traces every pixel (switch one of them off)");
        if (adaptive && tune.rays_per_lane == 2)
            return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "rays_per_lane = 2 with adaptive sampling: gr_trace_pair traces every pixel (switch one of them off)");
        if (tune.fused_shading == 1 && (adaptive || keep_lanes > 0 || tune.rays_per_lane == 2))
            return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "fused_shading = 1 needs one ray per lane, no compaction and no adaptive sampling");
        if (adaptive) keep_lanes = 0;   // GR_TRACE_COMPACT (an experiment switch for every frame) does not apply to adaptive frames
        // two rays per lane (gr_trace_pair) where the program has that kernel, unless told otherwise
        static const int default_rays_per_lane = [] { const char* e = getenv("GR_TRACE_RAYS_PER_LANE"); int v = e ? atoi(e) : 0; return (v == 1 || v == 2) ? v : GR_DEFAULT_RAYS_PER_LANE; }();
        int rays_per_lane = tune.rays_per_lane == 1 || tune.rays_per_lane == 2 ? tune.rays_per_lane : default_rays_per_lane;
        if (rays_per_lane == 2 && !gr_program_has_trace_pair(p)) {
            if (tune.rays_per_lane == 2) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "rays_per_lane = 2: this program has no gr_trace_pair kernel");
            rays_per_lane = 1;
        }
        // The prepass inside the trace launch (gr_trace_fused_args.inline_prepass): for a frame whose prepass was not computed
        // ahead of time - an interactive caller does not know the next camera - the prepass's single-ray latency (1.1 ms at 4K
        // Kerr, 8 ms with a = 0.9) then runs alongside the first tiles instead of in front of the whole trace.
        static const int inline_default = [] { const char* e = getenv("GR_INLINE_PREPASS"); return !e ? 1 : e[0] != '0'; }();
        // By default on whole frames that do not order their tiles (the order needs the prepass rays' costs first).  A device's share
        // of a split frame can do it too (inline_prepass = 1: it traces the cells its rows look at), but does not by default, on
        // measurement - one rank of 8 / of 4, one frame at a time, 2 wave slots per SIMD: 2.41 / 3.42 ms against 2.20 / 2.80 with the
        // prepass in front and the tiles ordered by its costs (tools/strip_probe.py, STRIP_PROBE_DEPTH=0): a share is few tiles, and
        // which of them start first matters more than the prepass's latency.
        const bool inline_wanted = tune.inline_prepass < 0 ? (inline_default != 0 && strip_count == 1 && !order_capable) : tune.inline_prepass != 0;
        // (an adaptively sampled whole frame: the cells ride in front of the lattice launch's tiles)
        const bool inline_prepass = inline_wanted && !prefetched && one_launch_setup && use_prepass && (!adaptive || strip_count == 1) && keep_lanes == 0 &&
                                    rays_per_lane == 1 && prepass_width != width && prepass_height != height;
        const bool order_tiles = order_capable && !inline_prepass;
        // (the kernels that record and follow the history are gr_trace_fused's: one ray per lane, no compaction)
        const bool record_history = history_wanted && !device_busy && keep_lanes == 0 && rays_per_lane == 1;
        if (history_wanted && !record_history) s->tile_cost_valid = false;   // (what is there would be older than the last frame)
        const int shape[3] = {adaptive ? -hist_block_rows : block_rows, strip_rank, strip_count};   // (negative: the tiles of a lattice launch)
        static const float history_max_motion = [] { const char* e = getenv("GR_TILE_HISTORY_MAX_MOTION"); return e ? (float)atof(e) : 48.f; }();
        // ... or, when the camera mostly TURNED (mouse look: the picture shifts rigidly and the order shifts with it - follow_and_record_history -
        // as long as both frames see the coordinate origin), up to GR_TILE_HISTORY_MAX_TURN px of turn with at most the 48 px of parallax:
        // 1.5 degrees a frame at 4K (50 px) rendered in 6.23 ms with the history dropped and in 5.24 with it (a = 0.9: 24.0 and 21.9), 2.2 degrees
        // (73 px) in 6.36 and 5.50 (24.4 and 23.3); at 3 degrees (100 px) the a = 0.9 frame loses badly (23.9 -> 33.8 ms: the shift is the
        // picture centre's, a perspective picture moves by 1 / cos^2 more towards its edges, and beyond the guard ring of 48 px tiles
        // guessed empty are dear): 64 px.
        static const float history_max_turn = [] { const char* e = getenv("GR_TILE_HISTORY_MAX_TURN"); return e ? (float)atof(e) : 64.f; }();
        const bool history_order = record_history && s->tile_cost_valid && memcmp(shape, s->tile_cost_shape, sizeof(shape)) == 0 && !gc && [&] {
            if (picture_motion(s->tile_cost_camera, *camera, features.field_of_view, width) <= history_max_motion) return true;
            float turn = 0, parallax = 0, anchor[2];
            return picture_motion_parts(s->tile_cost_camera, *camera, features.field_of_view, width, turn, parallax) && parallax <= history_max_motion &&
                   turn <= history_max_turn && s->tile_cost_anchored && origin_on_screen(*camera, features.field_of_view, width, height, anchor);
        }();
        if (!prefetched && one_launch_setup) {
            GR_CHECK(begin(GR_STAGE_PREPASS));
            GR_CHECK(gr_camera_prepass(p, stream, s->camera_pos_cart, camera->flip, camera->basis_speed, s->camera_pos_generic, s->tetrad[0],
                                       s->tetrad[1], s->tetrad[2], s->tetrad[3], s->camera_quat, s->termination_buffer,
                                       use_prepass && !inline_prepass ? prepass_width : 0, use_prepass && !inline_prepass ? prepass_height : 0, s->cfg,
                                       s->dfg, height, block_rows, strip_rank, strip_count, cost_plane(s->termination_buffer), prepass_margin));
            if (order_tiles)
                GR_CHECK(gr_order_tiles(p, stream, s->termination_buffer, cost_plane(s->termination_buffer), prepass_width, prepass_height,
                                        width, height, block_rows, strip_rank, strip_count, s->tile_order));
            GR_CHECK(end(GR_STAGE_PREPASS));
        } else if (use_prepass && !prefetched) {
            GR_CHECK(begin(GR_STAGE_PREPASS));
            GR_CHECK(gr_prepass_fused_strips(p, stream, s->camera_pos_generic, s->camera_quat, s->termination_buffer, prepass_width,
                                             prepass_height, s->tetrad[0], s->tetrad[1], s->tetrad[2], s->tetrad[3], s->cfg, s->dfg,
                                             height, block_rows, strip_rank, strip_count, cost_plane(s->termination_buffer), prepass_margin));
            if (order_tiles)
                GR_CHECK(gr_order_tiles(p, stream, s->termination_buffer, cost_plane(s->termination_buffer), prepass_width, prepass_height,
                                        width, height, block_rows, strip_rank, strip_count, s->tile_order));
            GR_CHECK(end(GR_STAGE_PREPASS));
        }
        // look-ahead requests that are not already sitting in a slot
        struct request { const gr_camera* camera; float time; int strip_rank; };   // strip_rank < 0: this frame's
        std::vector<request> todo;
        bool claimed[gr_render_state::LOOKAHEAD] = {};   // a slot serves one request (two frames may share one camera)
        if (use_prepass) {
            request asked[2] = {{opt.next_camera, tune.next_geodesic_time, tune.next_strip_rank},
                                {opt.next_camera2, tune.next_geodesic_time2, tune.next_strip_rank2}};
            // nobody announced the next camera, and this frame is the previous one over again: the guess "the same once more"
            // (gr_frame_tuning.guess_still_camera, off by default since reuse_still_camera does without the prepass altogether) - used
            // by the next frame only if its key matches bit for bit
            static const int guess_default = [] { const char* e = getenv("GR_GUESS_STILL_CAMERA"); return !e ? 0 : e[0] != '0'; }();
            if (!opt.next_camera && !opt.next_camera2 && !gc && strip_count == 1 && (tune.guess_still_camera < 0 ? guess_default : tune.guess_still_camera) != 0 &&
                repeats_previous_frame && !reuse_on)
                asked[0] = {camera, opt.geodesic_time, -1};
            for (auto& r : asked) {
                if (!r.camera) continue;
                if (r.strip_rank < 0 || r.strip_rank >= strip_count) r.strip_rank = strip_rank;
                const auto k = make_key(r.camera, r.time, r.strip_rank);
                bool have = false;
                for (int i = 0; i < gr_render_state::LOOKAHEAD && !have; i++)
                    if (!claimed[i] && s->pre[i].valid && s->pre[i].key == k) claimed[i] = have = true;
                if (!have) todo.push_back(r);
            }
        }
        if (!todo.empty()) HIP_CHECK(hipEventRecord(s->main_mark, stream));   // everything up to here is older than the prefetches
        auto inspect_prepass = [&]() -> int {
            if (!(prepass_by_policy && use_prepass && !s->policy.in_flight)) return GR_OK;
            // this frame's prepass flags -> host, behind the prepass on the frame's stream; read when a later frame finds them there
            auto& pol = s->policy;
            if (pol.capacity < cells) {
                if (pol.host_flags) (void)hipHostFree(pol.host_flags);
                pol.host_flags = nullptr;
                HIP_CHECK(hipHostMalloc((void**)&pol.host_flags, cells * sizeof(int), hipHostMallocDefault));
                pol.capacity = cells;
            }
            if (!pol.copied) HIP_CHECK(hipEventCreateWithFlags(&pol.copied, hipEventDisableTiming));
            HIP_CHECK(hipMemcpyAsync(pol.host_flags, s->termination_buffer, cells * sizeof(int), hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipEventRecord(pol.copied, stream));
            pol.cells = cells;
            pol.grid_width = prepass_width;
            pol.in_flight = true;
            return GR_OK;
        };
        if (!inline_prepass) GR_CHECK(inspect_prepass());   // (with the prepass inside the trace launch: after it)
        // every device runs the (tiny) prepass itself; its own row blocks (+ one halo row each) are traced here
        bool shade_in_trace = false;
        GR_CHECK(begin(GR_STAGE_TRACE));
        // the order of the frame before's costs and the record of this frame's, for the launch that traces the frame's tiles
        // (gr_trace_fused on every pixel, or the lattice launch of adaptive sampling: tiles of 8 x 8 lattice pixels = 16 x 16 pixels)
        auto follow_and_record_history = [&](gr_trace_fused_args& a) -> int {
            float anchor[2] = {0, 0};
            const bool anchored = record_history && !gc && origin_on_screen(*camera, features.field_of_view, width, height, anchor);
            if (history_order) {
                // how far the picture has moved since the costs were recorded, in tiles
                int shift[2] = {0, 0};
                static const bool follow_camera = [] { const char* e = getenv("GR_TILE_HISTORY_FOLLOW"); return !(e && e[0] == '0'); }();
                if (follow_camera && anchored && s->tile_cost_anchored)
                    for (int i = 0; i < 2; i++)
                        shift[i] = (int)std::lround(std::max(-4096.f, std::min(4096.f, (anchor[i] - s->tile_cost_anchor[i]) / (adaptive ? 16.f : 8.f))));
                GR_CHECK(gr_order_tiles_by_history(p, stream, s->tile_cost, hist_width, hist_height, hist_block_rows, strip_rank, strip_count, s->tile_order,
                                                   shift[0], shift[1]));
                s->history_followed++;
                s->history_last_shift[0] = shift[0]; s->history_last_shift[1] = shift[1];
                a.tile_order = s->tile_order;
                a.tile_order_by_history = 1;
                a.speculative_classes = tune.speculative_classes < 0 ? 0 : tune.speculative_classes == 0 ? -1 : tune.speculative_classes;
            }
            if (record_history) {
                a.tile_cost = s->tile_cost;
                s->history_recorded++;
                memcpy(s->tile_cost_shape, shape, sizeof(shape));
                s->tile_cost_valid = true;
                s->tile_cost_anchored = anchored;
                s->tile_cost_anchor[0] = anchor[0]; s->tile_cost_anchor[1] = anchor[1];
                s->tile_cost_camera = *camera;
            }
            return GR_OK;
        };
        if (keep_lanes > 0)
            GR_CHECK(gr_trace_compact(p, stream, s->camera_pos_generic, s->camera_quat, s->render_data, width, height, block_rows,
                                      strip_rank, strip_count, use_prepass ? s->termination_buffer : nullptr,
                                      use_prepass ? prepass_width : width, use_prepass ? prepass_height : height, s->tetrad[0],
                                      s->tetrad[1], s->tetrad[2], s->tetrad[3], s->cfg, s->dfg, attempts, keep_lanes));
        else {
            if (adaptive) {
                // quarter of the primary rays (the pixels (2x, 2y)), then the blocks that need it refined by a second launch
                const void* term = use_prepass ? s->termination_buffer : nullptr;
                const int pw = use_prepass ? prepass_width : width, ph = use_prepass ? prepass_height : height;
                gr_trace_fused_args a{};
                a.camera_generic = s->camera_pos_generic; a.camera_quat = s->camera_quat; a.render_data = s->render_data;
                a.width = width; a.height = height; a.block_rows = block_rows; a.strip_rank = strip_rank; a.strip_count = strip_count;
                a.termination_buffer = term; a.prepass_width = pw; a.prepass_height = ph;
                a.e0 = s->tetrad[0]; a.e1 = s->tetrad[1]; a.e2 = s->tetrad[2]; a.e3 = s->tetrad[3]; a.cfg = s->cfg; a.dfg = s->dfg;
                a.attempt_counter = attempts;
                a.waves_per_simd = tune.trace_waves_per_simd;
                a.lattice = 2;
                a.inline_prepass = inline_prepass ? 1 : 0;
                if (!s->lattice_rays) HIP_CHECK(hipMalloc(&s->lattice_rays, gr_lattice_rays_bytes(width, height)));
                if (!s->pending_list) HIP_CHECK(hipMalloc(&s->pending_list, gr_pending_list_bytes(width, height)));
                a.lattice_rays = s->lattice_rays;
                GR_CHECK(follow_and_record_history(a));
                // Tracing ahead what the second launch will ask for (gr_apply_guessed): where the frame before's second launch found pixels of
                // 4 096 attempts and more (program.hip GR_GUESSED_ATTEMPTS), this frame's lattice launch traces the same pixels beside its
                // tiles.  Whole frames that find the device idle (GR_ADAPTIVE_GUESS=0: never).
                // (a guess is a PIXEL: it is right when the picture has not moved - a viewer whose user is looking, or dragging a slider -
                // and a ray traced for nothing otherwise: used up to half a pixel of motion, GR_ADAPTIVE_GUESS_MAX_MOTION)
                static const float guess_max_motion = [] { const char* e = getenv("GR_ADAPTIVE_GUESS_MAX_MOTION"); return e ? (float)atof(e) : 0.5f; }();
                const bool keep_guesses = guesses_wanted && !device_busy;
                if (keep_guesses && !s->guessed[0])
                    for (void*& g : s->guessed) { HIP_CHECK(hipMalloc(&g, gr_guessed_bytes())); HIP_CHECK(hipMemsetAsync(g, 0, 32, stream)); }
                const bool use_guesses = keep_guesses && s->guessed_valid && s->block_cost_valid && !cfg_jumped && !features_changed &&
                                         s->block_cost_program == gr_program_serial(p) &&
                                         picture_motion(s->block_cost_camera, *camera, features.field_of_view, width) <= guess_max_motion;
                if (keep_guesses && !use_guesses) HIP_CHECK(hipMemsetAsync(s->guessed[0], 0, 4, stream));   // (whatever is there is not for this picture)
                a.guessed = keep_guesses ? s->guessed[0] : nullptr;
                GR_CHECK(gr_trace_fused_launch(p, stream, &a));
                a.tile_order = nullptr; a.tile_order_by_history = 0; a.tile_cost = nullptr;   // (the second launch below is not the lattice's)
                GR_CHECK(end(GR_STAGE_TRACE));
                GR_CHECK(begin(GR_STAGE_ADAPTIVE));
                HIP_CHECK(hipMemsetAsync(s->rays_adaptive_count, 0, 4, stream));
                // the decisions, the marked pixels as a list ordered dearest first, and the second launch over that list: every lane
                // of every wave has a ray (GR_ADAPTIVE_PENDING_LIST=0: the marked pixels found by walking the image's tiles again)
                static const bool as_list = [] { const char* e = getenv("GR_ADAPTIVE_PENDING_LIST"); return !(e && e[0] == '0'); }();
                if (as_list) {
                    // the list's order: what the lattice rays around a block cost, and - while the picture has moved little since - what
                    // the block's own rays cost in this state's frame before (the long rays are filaments a pixel or two wide)
                    static const float history_max_motion = [] { const char* e = getenv("GR_ADAPTIVE_HISTORY_MAX_MOTION"); return e ? (float)atof(e) : 48.f; }();
                    const size_t image_blocks = (size_t)(width / 2) * (height / 2);
                    std::swap(s->block_cost, s->block_cost_before);
                    if (!s->block_cost) HIP_CHECK(hipMalloc(&s->block_cost, image_blocks * sizeof(unsigned int)));
                    const bool by_history = strip_count == 1 && s->block_cost_valid && s->block_cost_before && !cfg_jumped && !features_changed &&
                                            s->block_cost_program == gr_program_serial(p) && !gc &&
                                            picture_motion(s->block_cost_camera, *camera, features.field_of_view, width) <= history_max_motion;
                    HIP_CHECK(hipMemsetAsync(s->block_cost, 0, image_blocks * sizeof(unsigned int), stream));
                    GR_CHECK(gr_adaptive_refine_list(p, stream, s->render_data, s->rays_adaptive_count, width, height, s->dfg, block_rows, strip_rank,
                                                     strip_count, s->lattice_rays, s->cfg, s->pending_list, by_history ? s->block_cost_before : nullptr));
                    if (keep_guesses) {
                        HIP_CHECK(hipMemsetAsync(s->guessed[1], 0, 4, stream));
                        GR_CHECK(gr_apply_guessed(p, stream, s->render_data, width, s->guessed[0], s->guessed[1], s->block_cost, attempts));
                    }
                    GR_CHECK(gr_trace_pending(p, stream, s->camera_pos_generic, s->camera_quat, s->render_data, width, height, s->tetrad[0],
                                              s->tetrad[1], s->tetrad[2], s->tetrad[3], s->cfg, s->dfg, attempts, s->pending_list,
                                              tune.trace_waves_per_simd, s->block_cost, keep_guesses ? s->guessed[1] : nullptr));
                    if (keep_guesses) std::swap(s->guessed[0], s->guessed[1]);
                    s->guessed_valid = keep_guesses;
                    s->block_cost_valid = strip_count == 1;
                    s->block_cost_program = gr_program_serial(p);
                    s->block_cost_camera = *camera;
                } else {
                    s->block_cost_valid = false;
                    s->guessed_valid = false;
                    GR_CHECK(gr_adaptive_refine_strips(p, stream, s->render_data, s->rays_adaptive_count, width, height, s->dfg, block_rows,
                                                       strip_rank, strip_count, s->lattice_rays, s->cfg));
                    a.lattice = 1;
                    a.pending_only = 1;
                    a.inline_prepass = 0;
                    GR_CHECK(gr_trace_fused_launch(p, stream, &a));
                }
                GR_CHECK(end(GR_STAGE_ADAPTIVE));
            } else {
            if (rays_per_lane == 2)
                GR_CHECK(gr_trace_pair(p, stream, s->camera_pos_generic, s->camera_quat, s->render_data, width, height, block_rows,
                                       strip_rank, strip_count, use_prepass ? s->termination_buffer : nullptr,
                                       use_prepass ? prepass_width : width, use_prepass ? prepass_height : height, s->tetrad[0],
                                       s->tetrad[1], s->tetrad[2], s->tetrad[3], s->cfg, s->dfg, attempts));
            else {
                gr_trace_fused_args a{};
                a.camera_generic = s->camera_pos_generic; a.camera_quat = s->camera_quat; a.render_data = s->render_data;
                a.width = width; a.height = height; a.block_rows = block_rows; a.strip_rank = strip_rank; a.strip_count = strip_count;
                a.termination_buffer = use_prepass ? s->termination_buffer : nullptr;
                a.prepass_width = use_prepass ? prepass_width : width; a.prepass_height = use_prepass ? prepass_height : height;
                a.e0 = s->tetrad[0]; a.e1 = s->tetrad[1]; a.e2 = s->tetrad[2]; a.e3 = s->tetrad[3]; a.cfg = s->cfg; a.dfg = s->dfg;
                a.attempt_counter = attempts;
                a.tile_order = order_tiles ? s->tile_order : nullptr;
                GR_CHECK(follow_and_record_history(a));
                a.waves_per_simd = tune.trace_waves_per_simd;
                a.inline_prepass = inline_prepass ? 1 : 0;
                // parking (gr_trace_fused_parking): the lot is the state's, allocated the first time a frame asks for it - room for an
                // eighth of the frame's rays (a = 0.9 at 4K parks 4 % of them, re-parked ones counted again; a full lot is not an error)
                static const int park_default[2] = {[] { const char* e = getenv("GR_PARK"); return e ? atoi(e) : 0; }(),
                                                    [] { const char* e = getenv("GR_PARK"); const char* c = e ? strchr(e, ',') : nullptr; return c ? atoi(c + 1) : 0; }()};
                const int park_lanes = tune.park_lanes < 0 ? park_default[0] : tune.park_lanes;
                const int park_trips = tune.park_trips > 0 ? tune.park_trips : park_default[1] > 0 ? park_default[1] : 512;
                if (tune.park_lanes > 1 && !gr_program_has_parking(p))
                    return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "park_lanes: needs a program built with -DGR_PARKING appended to its argument string");
                if (park_lanes > 1 && tune.fused_shading != 1 && gr_program_has_parking(p)) {
                    const int slots = (int)std::min<long long>(1 << 24, std::max<long long>(65536, (long long)width * height / 8));
                    if (!s->parking_records || s->parking_slots != slots) {
                        if (s->parking_records) (void)hipFree(s->parking_records);
                        if (s->parking_words) (void)hipFree(s->parking_words);
                        s->parking_records = s->parking_words = nullptr;
                        size_t words_bytes = 0;
                        const size_t bytes = gr_parking_lot_bytes(slots, slots, &words_bytes);
                        HIP_CHECK(hipMalloc(&s->parking_records, bytes));
                        HIP_CHECK(hipMalloc(&s->parking_words, words_bytes));
                        s->parking_slots = slots;
                    }
                    a.parking.records = s->parking_records; a.parking.words = s->parking_words;
                    a.parking.lanes = std::min(park_lanes, 64); a.parking.trips = park_trips;
                    a.parking.slots = slots; a.parking.groups = slots;
                }
                // the trace shades the pixels whose filter neighbours are in their own tile; gr_render_seams below does the rest
                shade_in_trace = out && tune.fused_shading == 1 && width % 8 == 0 && height % 8 == 0 && gr_program_has_tile_shading(p);
                if (tune.fused_shading == 1 && !shade_in_trace && out)   // default: off, on measurement
                    return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT,
                                            "fused_shading = 1: needs a program built with -DGR_TILE_SHADING and width, height multiples of 8");
                if (shade_in_trace) {
                    a.shading.out = out; a.shading.background1 = bg1; a.shading.background2 = bg2; a.shading.bg_width = bg_width;
                    a.shading.bg_height = bg_height; a.shading.bg_levels = bg_levels; a.shading.max_probes = opt.max_probes;
                    a.shading.compact_out = strip_count > 1 ? opt.compact_out : 0;
                }
                GR_CHECK(gr_trace_fused_launch(p, stream, &a));
            }
            }
        }
        if (!adaptive) GR_CHECK(end(GR_STAGE_TRACE));
        if (inline_prepass) GR_CHECK(inspect_prepass());
        for (const auto& r : todo) {
            // a free slot, else the stalest one no current request claims (a camera that was announced but never came)
            gr_render_state::prefetch_slot* slot = nullptr;
            int chosen = -1;
            for (int i = 0; i < gr_render_state::LOOKAHEAD; i++)
                if (!claimed[i] && (chosen < 0 || (!s->pre[i].valid && s->pre[chosen].valid) ||
                                    (s->pre[i].valid == s->pre[chosen].valid && s->pre[i].age < s->pre[chosen].age)))
                    chosen = i;
            if (chosen >= 0) { slot = &s->pre[chosen]; claimed[chosen] = true; }
            if (!slot) break;
            // camera set-up and prepass on the slot's stream, into the slot's buffer set, while the trace runs.  The set may
            // have belonged to the previous frame (swapped out above): the stream first waits for everything the caller's
            // stream had queued before this frame's trace.
            HIP_CHECK(hipStreamWaitEvent(slot->stream, s->main_mark, 0));
            GR_CHECK(s->uploads.copy(slot->set.camera_pos_cart, r.camera->position, 16, slot->stream));
            GR_CHECK(s->uploads.copy(slot->set.camera_quat, r.camera->quat, 16, slot->stream));
            if (one_launch_setup) {
                GR_CHECK(gr_camera_prepass(p, slot->stream, slot->set.camera_pos_cart, r.camera->flip, r.camera->basis_speed,
                                           slot->set.camera_pos_generic, slot->set.tetrad[0], slot->set.tetrad[1], slot->set.tetrad[2],
                                           slot->set.tetrad[3], slot->set.camera_quat, slot->set.termination_buffer, prepass_width,
                                           prepass_height, s->cfg, s->dfg, height, block_rows, r.strip_rank, strip_count,
                                           cost_plane(slot->set.termination_buffer), prepass_margin));
            } else {
                GR_CHECK(camera_setup(slot->stream, slot->set.camera_pos_cart, slot->set.camera_pos_generic, slot->set.tetrad, r.camera, r.time,
                                      slot->velocity));
                GR_CHECK(gr_prepass_fused_strips(p, slot->stream, slot->set.camera_pos_generic, slot->set.camera_quat,
                                                 slot->set.termination_buffer, prepass_width, prepass_height, slot->set.tetrad[0],
                                                 slot->set.tetrad[1], slot->set.tetrad[2], slot->set.tetrad[3], s->cfg, s->dfg, height,
                                                 block_rows, r.strip_rank, strip_count, cost_plane(slot->set.termination_buffer), prepass_margin));
            }
            if (order_capable)   // the look-ahead frame's order too: off the frame's critical path like its prepass
                GR_CHECK(gr_order_tiles(p, slot->stream, slot->set.termination_buffer, cost_plane(slot->set.termination_buffer), prepass_width,
                                        prepass_height, width, height, block_rows, r.strip_rank, strip_count, slot->set.tile_order));
            HIP_CHECK(hipEventRecord(slot->ready, slot->stream));
            slot->valid = true;
            slot->age = s->frame_counter;
            slot->key = make_key(r.camera, r.time, r.strip_rank);
        }
        s->previous_key_valid = use_prepass && strip_count == 1 && !gc;
        if (s->previous_key_valid) { s->previous_key = make_key(camera, opt.geodesic_time, opt.strip_rank); s->previous_stream = stream; }
        if (out) {
            GR_CHECK(begin(GR_STAGE_RENDER));
            if (shade_in_trace)
                GR_CHECK(gr_render_seams(p, stream, s->render_data, out, bg1, bg2, bg_width, bg_height, bg_levels, width, height, block_rows,
                                         strip_rank, strip_count, strip_count > 1 ? opt.compact_out : 0, opt.max_probes, s->cfg, s->dfg));
            else
                GR_CHECK(gr_render_strips(p, stream, s->render_data, out, bg1, bg2, bg_width, bg_height, bg_levels, width, height,
                                          strip_count > 1 ? block_rows : height, strip_rank, strip_count,
                                          strip_count > 1 ? opt.compact_out : 0, opt.max_probes, s->cfg, s->dfg));
            GR_CHECK(end(GR_STAGE_RENDER));
        }
        if (history_wanted || guesses_wanted) mark_frame_end(s->device, stream);
        return GR_OK;
    }

    // ---- reference-shaped sequence --------------------------------------------------------------------
    int tiled = (opt.tiled && !adaptive) ? 1 : 0;
    size_t slots = tiled ? (size_t)gr_tiled_slot_count(width, height) : (size_t)width * height;
    GR_CHECK(ensure_rays(s, slots, adaptive));

    if (use_prepass) {
        GR_CHECK(begin(GR_STAGE_PREPASS));
        GR_CHECK(gr_clear_termination_buffer(p, stream, s->termination_buffer, prepass_width, prepass_height));
        HIP_CHECK(hipMemsetAsync(s->rays_count_in, 0, 4, stream));
        GR_CHECK(gr_init_rays_generic(p, stream, s->camera_pos_generic, s->camera_quat, s->rays_in, s->rays_count_in,
                                      prepass_width, prepass_height, s->termination_buffer, prepass_width, prepass_height, 0,
                                      s->tetrad[0], s->tetrad[1], s->tetrad[2], s->tetrad[3], s->cfg, s->dfg, 1, 0));
        GR_CHECK(gr_do_generic_rays(p, stream, s->rays_in, s->rays_count_in, prepass_width * prepass_height, nullptr, nullptr,
                                    s->cfg, s->dfg, width, height, 0, 0, nullptr, nullptr, 0, nullptr));
        GR_CHECK(gr_calculate_singularities(p, stream, s->rays_in, s->rays_count_in, prepass_width * prepass_height,
                                            s->termination_buffer, prepass_width, prepass_height));
        GR_CHECK(end(GR_STAGE_PREPASS));
    }
    int pw = use_prepass ? prepass_width : width, ph = use_prepass ? prepass_height : height;

    GR_CHECK(begin(GR_STAGE_INIT));
    GR_CHECK(gr_init_rays_generic(p, stream, s->camera_pos_generic, s->camera_quat, s->rays_in, s->rays_count_in, width, height,
                                  s->termination_buffer, pw, ph, 0, s->tetrad[0], s->tetrad[1], s->tetrad[2], s->tetrad[3],
                                  s->cfg, s->dfg, 0, tiled));
    GR_CHECK(end(GR_STAGE_INIT));

    GR_CHECK(begin(GR_STAGE_TRACE));
    // Rays in tile slot order are traced the way the fused kernel's tiles are: the device filled once, tiles by ticket, dearest first by
    // what they cost in this state's frame before while the picture has moved little since (round 5: in slot order, a workgroup to a tile,
    // the 4K Kerr launch took 6.1 ms against the fused trace's 4.8 - the difference was the tail).  GR_REFERENCE_SCHEDULED=0: the
    // reference's own launch shape, one work item per record.  The records are the same either way.
    static const bool scheduled_default = [] { const char* e = getenv("GR_REFERENCE_SCHEDULED"); return !(e && e[0] == '0'); }();
    if (tiled && scheduled_default) {
        const int tiles_x = (width + 7) / 8, tiles_y = (height + 7) / 8, tile_count = tiles_x * tiles_y;
        if (s->ref_cost_tiles != tile_count) {
            for (void*& b : s->ref_cost) { if (b) (void)hipFree(b); b = nullptr; }
            if (s->ref_order) (void)hipFree(s->ref_order);
            if (s->ref_sort_work) (void)hipFree(s->ref_sort_work);
            s->ref_order = s->ref_sort_work = nullptr;
            for (void*& b : s->ref_cost) HIP_CHECK(hipMalloc(&b, (size_t)tile_count * sizeof(unsigned int)));
            HIP_CHECK(hipMalloc(&s->ref_order, (size_t)tile_count * sizeof(unsigned int)));
            HIP_CHECK(hipMalloc(&s->ref_sort_work, ((size_t)tile_count + 128) * sizeof(unsigned int)));
            s->ref_cost_tiles = tile_count;
            s->ref_cost_valid = false;
        }
        static const float max_motion = [] { const char* e = getenv("GR_TILE_HISTORY_MAX_MOTION"); return e ? (float)atof(e) : 48.f; }();
        const bool follow = s->ref_cost_valid && !cfg_jumped && !features_changed && s->ref_cost_program == gr_program_serial(p) && !gc &&
                            picture_motion(s->ref_cost_camera, *camera, features.field_of_view, width) <= max_motion;
        std::swap(s->ref_cost[0], s->ref_cost[1]);
        if (follow) {
            GR_CHECK(gr_sort_tiles_by_cost(p, stream, s->ref_cost[1], tiles_x, tiles_y, s->ref_order, s->ref_sort_work));
            s->history_followed++;
        }
        s->history_recorded++;
        GR_CHECK(gr_do_generic_rays_scheduled(p, stream, s->rays_in, s->rays_count_in, tile_count, s->cfg, s->dfg, attempts,
                                              follow ? s->ref_order : nullptr, s->ref_cost[0]));
        s->ref_cost_valid = true;
        s->ref_cost_camera = *camera;
        s->ref_cost_program = gr_program_serial(p);
    } else {
        GR_CHECK(gr_do_generic_rays(p, stream, s->rays_in, s->rays_count_in, (int)slots, nullptr, nullptr, s->cfg, s->dfg, width,
                                    height, 0, 0, nullptr, nullptr, 0, attempts));
    }
    GR_CHECK(end(GR_STAGE_TRACE));

    HIP_CHECK(hipMemsetAsync(s->render_data_count, 0, 4, stream));
    GR_CHECK(begin(GR_STAGE_RENDER_DATA));
    GR_CHECK(gr_calculate_render_data(p, stream, s->rays_in, s->rays_count_in, (int)slots, s->render_data, s->render_data_count,
                                      width, height, s->cfg, s->dfg));
    GR_CHECK(end(GR_STAGE_RENDER_DATA));

    if (adaptive) {
        GR_CHECK(begin(GR_STAGE_ADAPTIVE));
        HIP_CHECK(hipMemsetAsync(s->rays_adaptive_count, 0, 4, stream));
        GR_CHECK(gr_handle_adaptive_sampling(p, stream, s->rays_in, s->rays_count_in, s->render_data, s->render_data_count,
                                             s->rays_adaptive, s->rays_adaptive_count, s->camera_pos_generic, s->camera_quat,
                                             s->tetrad[0], s->tetrad[1], s->tetrad[2], s->tetrad[3], width, height, s->cfg,
                                             s->dfg));
        GR_CHECK(gr_do_generic_rays(p, stream, s->rays_adaptive, s->rays_adaptive_count, width * height, nullptr, nullptr, s->cfg,
                                    s->dfg, width, height, 0, 0, nullptr, nullptr, 0, attempts));
        GR_CHECK(gr_calculate_render_data(p, stream, s->rays_adaptive, s->rays_adaptive_count, width * height, s->render_data,
                                          s->render_data_count, width, height, s->cfg, s->dfg));
        GR_CHECK(end(GR_STAGE_ADAPTIVE));
    }

    if (out) {
        GR_CHECK(begin(GR_STAGE_RENDER));
        GR_CHECK(gr_render(p, stream, s->render_data, s->render_data_count, width * height, out, bg1, bg2, bg_width, bg_height,
                           bg_levels, width, height, opt.max_probes, s->cfg, s->dfg));
        GR_CHECK(end(GR_STAGE_RENDER));
    }
    return GR_OK;
}

}  // extern "C"
