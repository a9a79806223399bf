// tiled.cpp - one frame over several GPUs, through the C ABI alone.
//
// The reference is single-GPU (main.cpp:2244-2526 drives one cl::command_queue).  Rays are independent, so a frame splits without
// any exchange during tracing: image rows are dealt to the participants in blocks of block_rows rows, block-cyclically (global
// block b belongs to participant b % world - the shadow, which the prepass mostly skips, is spread over everybody), each
// participant runs its own W/16 x H/16 prepass for the cells its rows look at, traces its blocks plus one halo row under each
// (the texture filter reads the pixel below, cl.cl:5509-5520) and shades them.  The only communication is the finished float4
// rows going to the root - every block straight to its final place in the root's frame, so the root neither stages nor
// un-permutes anything:
//   * GR_TRANSPORT_RCCL  one process per GPU: per block an ncclSend on the owner and an ncclRecv on the root at the block's real
//     row offset, all of a frame's in one group (direct fan-in over the xGMI links).  librccl is loaded at run time - the library
//     itself has no link dependency on it - and the communicator is built from an id the caller distributes
//     (gr_tiled_unique_id on the root; MPI, a socket, torch.distributed.broadcast - the caller's choice);
//   * GR_TRANSPORT_PEER  one process driving several devices (or several participants on one device, which is how the tests
//     exercise every offset on a one-GPU box): hipMemcpyPeerAsync per block on the owner's stream, events for the root to wait on.
// The root renders its own blocks directly into the frame (compact_out = 0).  The share a participant renders can rotate with
// the frame number (`rotation`): shares differ in cost by up to ~9 % and with frames in flight everybody then runs at the mean.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/geodesic_hip.h"

extern "C" int gr_internal_fail(int code, const char* msg);

namespace {

#define HIP_CHECK(expr)                                                                                             \
    do {                                                                                                            \
        hipError_t e_ = (expr);                                                                                     \
        if (e_ != hipSuccess) return gr_internal_fail(GR_ERROR_DEVICE, (std::string(#expr) + ": " + hipGetErrorString(e_)).c_str()); \
    } while (0)

// the part of the RCCL API used here (rccl.h: ncclUniqueId is 128 bytes, ncclFloat = 7, ncclSuccess = 0)
struct rccl_api {
    void* handle = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    void* comm_init_rank_raw = nullptr;   // ncclCommInitRank takes the 128-byte id BY VALUE: cast where it is called
};
struct unique_id { char bytes[128]; };

rccl_api* rccl() {
    static rccl_api api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names) {
            api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
        }
        if (!api.handle) return;
        api.GetUniqueId = (int (*)(void*))dlsym(api.handle, "ncclGetUniqueId");
        api.comm_init_rank_raw = dlsym(api.handle, "ncclCommInitRank");
        api.CommDestroy = (int (*)(void*))dlsym(api.handle, "ncclCommDestroy");
        api.GroupStart = (int (*)())dlsym(api.handle, "ncclGroupStart");
        api.GroupEnd = (int (*)())dlsym(api.handle, "ncclGroupEnd");
        api.Send = (int (*)(const void*, size_t, int, int, void*, hipStream_t))dlsym(api.handle, "ncclSend");
        api.Recv = (int (*)(void*, size_t, int, int, void*, hipStream_t))dlsym(api.handle, "ncclRecv");
        api.GetErrorString = (const char* (*)(int))dlsym(api.handle, "ncclGetErrorString");
    });
    const bool ok = api.handle && api.GetUniqueId && api.comm_init_rank_raw && api.CommDestroy && api.GroupStart && api.GroupEnd && api.Send &&
                    api.Recv;
    return ok ? &api : nullptr;
}

int rccl_fail(const char* what, int rc) {
    rccl_api* r = rccl();
    std::string msg = std::string(what) + " failed: " + (r && r->GetErrorString ? r->GetErrorString(rc) : "?");
    return gr_internal_fail(GR_ERROR_DEVICE, msg.c_str());
}

// participants of one process that share a frame through peer copies
struct peer_group {
    int world = 0;
    std::vector<hipEvent_t> done;     // one per participant: its blocks have reached the root's frame
    std::vector<char> pending;
};

}  // namespace

struct gr_tiled {
    int transport = GR_TRANSPORT_PEER;
    int world = 1, rank = 0, device = 0, root = 0;
    int width = 0, height = 0, block_rows = 16, blocks_per_share = 0;
    void* comm = nullptr;                    // ncclComm_t
    void* local = nullptr;                   // compact strip buffer: blocks_per_share x block_rows x width float4 (not on the root)
    std::shared_ptr<peer_group> group;       // GR_TRANSPORT_PEER
    int root_device = 0;
};

extern "C" {

int gr_tiled_unique_id(void* id_out_128_bytes) {
    if (!id_out_128_bytes) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "null argument");
    rccl_api* r = rccl();
    if (!r) return gr_internal_fail(GR_ERROR_DEVICE, "librccl could not be loaded");
    int rc = r->GetUniqueId(id_out_128_bytes);
    return rc == 0 ? GR_OK : rccl_fail("ncclGetUniqueId", rc);
}

static int tiled_common(gr_tiled* t, int world, int rank, int device, int width, int height, int block_rows) {
    if (world < 1 || rank < 0 || rank >= world || width < 1 || height < 1 || block_rows < 8 || block_rows % 8)
        return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "world >= 1, 0 <= rank < world, block_rows a positive multiple of 8");
    if (world > 1 && height > 1 && (height - 1) % block_rows == 0)
        return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "the last image row must not start a block (its filter reads the row above)");
    t->world = world; t->rank = rank; t->device = device; t->width = width; t->height = height; t->block_rows = block_rows;
    const int total_blocks = (height + block_rows - 1) / block_rows;
    t->blocks_per_share = (total_blocks + world - 1) / world;
    if (rank != t->root) {
        HIP_CHECK(hipSetDevice(device));
        HIP_CHECK(hipMalloc(&t->local, (size_t)t->blocks_per_share * block_rows * width * 16));
    }
    return GR_OK;
}

int gr_tiled_create(int world, int rank, int device, const void* unique_id_128_bytes, int width, int height, int block_rows, gr_tiled** out) {
    if (!out) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "null argument");
    auto t = std::make_unique<gr_tiled>();
    t->transport = GR_TRANSPORT_RCCL;
    int rc = tiled_common(t.get(), world, rank, device, width, height, block_rows);
    if (rc != GR_OK) { if (t->local) (void)hipFree(t->local); return rc; }
    if (world > 1) {
        if (!unique_id_128_bytes) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "world > 1 needs the id of gr_tiled_unique_id");
        rccl_api* r = rccl();
        if (!r) return gr_internal_fail(GR_ERROR_DEVICE, "librccl could not be loaded");
        HIP_CHECK(hipSetDevice(device));
        unique_id id;
        memcpy(id.bytes, unique_id_128_bytes, sizeof(id.bytes));
        auto init = (int (*)(void**, int, unique_id, int))r->comm_init_rank_raw;   // ncclCommInitRank(comm*, nranks, id BY VALUE, rank)
        int nrc = init(&t->comm, world, id, rank);
        if (nrc != 0) { if (t->local) (void)hipFree(t->local); return rccl_fail("ncclCommInitRank", nrc); }
    }
    *out = t.release();
    return GR_OK;
}

int gr_tiled_create_local(int count, const int* devices, int width, int height, int block_rows, gr_tiled** out_array) {
    if (count < 1 || !devices || !out_array) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "null argument");
    auto group = std::make_shared<peer_group>();
    group->world = count;
    group->done.assign(count, nullptr);
    group->pending.assign(count, 0);
    std::vector<std::unique_ptr<gr_tiled>> made;
    for (int r = 0; r < count; r++) {
        auto t = std::make_unique<gr_tiled>();
        t->transport = GR_TRANSPORT_PEER;
        t->group = group;
        t->root_device = devices[0];
        int rc = tiled_common(t.get(), count, r, devices[r], width, height, block_rows);
        if (rc == GR_OK && hipSetDevice(devices[r]) == hipSuccess && hipEventCreateWithFlags(&group->done[r], hipEventDisableTiming) != hipSuccess)
            rc = gr_internal_fail(GR_ERROR_DEVICE, "hipEventCreate failed");
        if (rc == GR_OK && devices[r] != devices[0]) {
            int can = 0;
            (void)hipDeviceCanAccessPeer(&can, devices[r], devices[0]);
            if (can) { hipError_t e = hipDeviceEnablePeerAccess(devices[0], 0); if (e != hipSuccess) (void)hipGetLastError(); }   // already enabled is fine
        }
        if (rc != GR_OK) {
            if (t->local) (void)hipFree(t->local);
            for (auto& m : made) if (m->local) (void)hipFree(m->local);
            for (auto e : group->done) if (e) (void)hipEventDestroy(e);
            return rc;
        }
        made.push_back(std::move(t));
    }
    for (int r = 0; r < count; r++) out_array[r] = made[r].release();
    return GR_OK;
}

void gr_tiled_destroy(gr_tiled* t) {
    if (!t) return;
    (void)hipSetDevice(t->device);
    if (t->comm) { rccl_api* r = rccl(); if (r) (void)r->CommDestroy(t->comm); }
    if (t->local) (void)hipFree(t->local);
    if (t->group && t->group->done[t->rank]) { (void)hipEventDestroy(t->group->done[t->rank]); t->group->done[t->rank] = nullptr; }
    delete t;
}

int gr_tiled_share(const gr_tiled* t, int rotation) { return t ? ((t->rank + (rotation % t->world + t->world)) % t->world) : 0; }

int gr_tiled_block_rows(int height, int block_rows, int world, int share, int local_block, int* row_begin, int* row_end) {
    if (!row_begin || !row_end || height < 1 || block_rows < 1 || world < 1 || share < 0 || share >= world || local_block < 0) return -1;
    const long long a = ((long long)local_block * world + share) * block_rows;
    if (a >= height) return 0;   // padding block of this share
    *row_begin = (int)a;
    *row_end = a + block_rows < height ? (int)a + block_rows : height;
    return 1;
}

int gr_tiled_block_rows_of(const gr_tiled* t, int share, int local_block, int* row_begin, int* row_end) {
    if (!t) return -1;
    return gr_tiled_block_rows(t->height, t->block_rows, t->world, share, local_block, row_begin, row_end);
}

int gr_render_frame_tiled(gr_tiled* t, gr_render_state* s, gr_program* p, const gr_metric* m, void* stream_v, const gr_camera* camera,
                          const gr_features* features, const float* cfg_values, int num_cfg_values, const void* bg1, const void* bg2,
                          int bg_width, int bg_height, int bg_levels, void* frame_on_root, const gr_frame_options* options, int rotation) {
    if (!t || !s || !p || !m || !camera) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "null argument");
    hipStream_t stream = (hipStream_t)stream_v;
    const bool is_root = t->rank == t->root;
    if (is_root && !frame_on_root) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "the root needs the frame buffer");
    if (t->transport == GR_TRANSPORT_PEER && !frame_on_root) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "peer transport: every participant is given the root's frame buffer");
    gr_frame_options opt;
    gr_frame_options_default(&opt);
    if (options) opt = *options;
    opt.mode = GR_MODE_FUSED;
    const int share = gr_tiled_share(t, rotation);
    if (t->world > 1) {
        opt.strip_count = t->world;
        opt.strip_rank = share;
        opt.block_rows = t->block_rows;
        opt.compact_out = is_root ? 0 : 1;
    } else {
        opt.strip_count = 1;
        opt.strip_rank = 0;
        opt.compact_out = 0;
    }
    HIP_CHECK(hipSetDevice(t->device));
    int rc = gr_render_frame(s, p, m, stream_v, camera, features, cfg_values, num_cfg_values, bg1, bg2, bg_width, bg_height, bg_levels,
                             is_root ? frame_on_root : t->local, &opt);
    if (rc != GR_OK || t->world == 1) return rc;

    const size_t row_bytes = (size_t)t->width * 16, row_floats = (size_t)t->width * 4;
    if (t->transport == GR_TRANSPORT_RCCL) {
        rccl_api* r = rccl();
        int nrc = r->GroupStart();
        if (nrc != 0) return rccl_fail("ncclGroupStart", nrc);
        if (is_root) {
            for (int peer = 0; peer < t->world; peer++) {
                if (peer == t->root) continue;
                const int peer_share = (peer + (rotation % t->world + t->world)) % t->world;
                for (int i = 0; i < t->blocks_per_share; i++) {
                    int a, b;
                    if (gr_tiled_block_rows_of(t, peer_share, i, &a, &b) != 1) continue;
                    nrc = r->Recv((char*)frame_on_root + (size_t)a * row_bytes, (size_t)(b - a) * row_floats, 7 /* ncclFloat */, peer, t->comm, stream);
                    if (nrc != 0) { (void)r->GroupEnd(); return rccl_fail("ncclRecv", nrc); }
                }
            }
        } else {
            for (int i = 0; i < t->blocks_per_share; i++) {
                int a, b;
                if (gr_tiled_block_rows_of(t, share, i, &a, &b) != 1) continue;
                nrc = r->Send((const char*)t->local + (size_t)i * t->block_rows * row_bytes, (size_t)(b - a) * row_floats, 7, t->root, t->comm, stream);
                if (nrc != 0) { (void)r->GroupEnd(); return rccl_fail("ncclSend", nrc); }
            }
        }
        nrc = r->GroupEnd();
        if (nrc != 0) return rccl_fail("ncclGroupEnd", nrc);
        return GR_OK;
    }
    // peer copies: the owner pushes its blocks into the root's frame on its own stream
    if (!is_root) {
        for (int i = 0; i < t->blocks_per_share; i++) {
            int a, b;
            if (gr_tiled_block_rows_of(t, share, i, &a, &b) != 1) continue;
            void* dst = (char*)frame_on_root + (size_t)a * row_bytes;
            const void* src = (const char*)t->local + (size_t)i * t->block_rows * row_bytes;
            if (t->device == t->root_device) HIP_CHECK(hipMemcpyAsync(dst, src, (size_t)(b - a) * row_bytes, hipMemcpyDeviceToDevice, stream));
            else HIP_CHECK(hipMemcpyPeerAsync(dst, t->root_device, src, t->device, (size_t)(b - a) * row_bytes, stream));
        }
    }
    HIP_CHECK(hipEventRecord(t->group->done[t->rank], stream));
    t->group->pending[t->rank] = 1;
    return GR_OK;
}

int gr_tiled_join(gr_tiled* root, void* stream_v) {
    if (!root) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "null argument");
    if (root->transport != GR_TRANSPORT_PEER || root->world == 1) return GR_OK;   // RCCL: the receives are ordered on the root's stream
    HIP_CHECK(hipSetDevice(root->device));
    for (int r = 0; r < root->world; r++) {
        if (!root->group->pending[r]) continue;
        HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream_v, root->group->done[r], 0));
        root->group->pending[r] = 0;
    }
    return GR_OK;
}

}  // extern "C"
