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
//   * GR_TRANSPORT_CUSTOM  the caller's own point-to-point calls (a gr_transport table: MPI, sockets - and the recording transport
//     of tests/test_distributed_cpu.py, which drives the send / receive schedule of every rank of a world on the CPU).  RCCL is
//     one such table internally.
// Frames in flight: a participant stages its compact rows in a ring of buffers, one per frame in flight (frame_slot below), so
// frame k+1 - another stream, another share - never shades into memory frame k's sends still read.
#include <dlfcn.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/geodesic_hip_internal.h"

extern "C" int gr_internal_fail(int code, const char* msg);
extern "C" int gr_internal_render_state_size(const gr_render_state* s, int* width, int* height);
struct gr_tiled;

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
    std::vector<gr_tiled*> members;   // by rank; nullptr once destroyed
};

// One frame in flight of one participant: where its compact rows are staged (not on the root, which renders in place) and the
// event after which they have left that buffer.  gr_render_frame_tiled may be called for the next frame - on another stream, with
// another share - while this frame's sends still read the buffer, so a participant owns a ring of these and a frame that reuses a
// slot first waits for the slot's event.
struct frame_slot {
    void* buffer = nullptr;
    hipEvent_t done = nullptr;   // recorded on the frame's stream after its transfers were enqueued
    bool recorded = false;
    bool pending = false;        // GR_TRANSPORT_PEER: gr_tiled_join has not waited for it yet
};

}  // namespace

struct gr_tiled {
    int transport = GR_TRANSPORT_PEER;
    int world = 1, rank = 0, device = 0, root = 0;
    int width = 0, height = 0, block_rows = 16, blocks_per_share = 0;
    void* comm = nullptr;                    // ncclComm_t (GR_TRANSPORT_RCCL)
    gr_transport link{};                     // point-to-point calls of GR_TRANSPORT_RCCL / GR_TRANSPORT_CUSTOM
    std::vector<frame_slot> ring;
    unsigned long long frames = 0;
    int look_ahead = 1;                      // gr_tiled_look_ahead: rotations between a frame and the frame its options->next_camera is for
    std::shared_ptr<peer_group> group;       // GR_TRANSPORT_PEER
    int root_device = 0;
    std::shared_ptr<void> ipc;               // GR_TRANSPORT_IPC: the ipc_link (defined below; released with the participant)

    size_t staging_bytes() const { return (size_t)blocks_per_share * block_rows * width * 16; }
    ~gr_tiled() {
        if (device >= 0) (void)hipSetDevice(device);
        if (comm) { rccl_api* r = rccl(); if (r) (void)r->CommDestroy(comm); }
        for (auto& sl : ring) {
            if (sl.buffer) (void)hipFree(sl.buffer);
            if (sl.done) (void)hipEventDestroy(sl.done);
        }
        if (group && rank < (int)group->members.size()) group->members[rank] = nullptr;
    }
};

namespace {

// RCCL as a gr_transport (user = the gr_tiled that owns the communicator)
int rccl_group_begin(void*) { int rc = rccl()->GroupStart(); return rc == 0 ? GR_OK : rccl_fail("ncclGroupStart", rc); }
int rccl_group_end(void*) { int rc = rccl()->GroupEnd(); return rc == 0 ? GR_OK : rccl_fail("ncclGroupEnd", rc); }
int rccl_send(void* user, const void* data, size_t floats, int peer, void* stream) {
    int rc = rccl()->Send(data, floats, 7 /* ncclFloat */, peer, ((gr_tiled*)user)->comm, (hipStream_t)stream);
    return rc == 0 ? GR_OK : rccl_fail("ncclSend", rc);
}
int rccl_recv(void* user, void* data, size_t floats, int peer, void* stream) {
    int rc = rccl()->Recv(data, floats, 7, peer, ((gr_tiled*)user)->comm, (hipStream_t)stream);
    return rc == 0 ? GR_OK : rccl_fail("ncclRecv", rc);
}

// ---- GR_TRANSPORT_IPC: the point-to-point calls of a split frame between PROCESSES THAT MAY SHARE A DEVICE -------------------------
// RCCL refuses two ranks of one host on one device (unless each claims a host of its own: NCCL_HOSTID, tests/test_gpu_two_ranks.py), so on a one-GPU box the N-process path of gr_render_frame_tiled - one process per rank, a
// group per frame, per block a send on the owner and the matching receive on the root, rotation, frames in flight on several
// streams - could only be exercised with one participant.  This transport keeps the call pattern and the matching rule (the n-th
// send of rank r to the root meets the root's n-th receive from r) and moves the data itself: a receive publishes where the block
// goes (the inter-process handle of the allocation + offset) in a mailbox in POSIX shared memory, the send waits for that, maps the
// handle and copies device to device on its stream; the sender's group end waits for its stream and marks the blocks delivered, the
// root's group end waits for the marks.  Host-blocking where RCCL is asynchronous - it is a rehearsal stage, not a product path -
// but every rank issues what it issues under RCCL, in the same order, across real process boundaries.
struct ipc_slot {
    std::atomic<unsigned long long> posted;   // sequence number + 1 of the receive this slot describes
    hipIpcMemHandle_t handle;
    unsigned long long allocation;            // which of the root's allocations the handle is of (it is exported once; senders map it once)
    unsigned long long offset, bytes;
    std::atomic<unsigned long long> done;     // sequence number + 1 once the block has arrived
};
const int IPC_DEPTH = 256;                    // receives of one peer that can be outstanding (a frame has blocks_per_share of them)
struct ipc_shared {
    std::atomic<unsigned int> magic;          // 0x47524950 once rank 0 has initialised the region
    std::atomic<unsigned int> arrived;        // ranks that have opened it
    unsigned int world, pad;
    ipc_slot box[1];                          // [world][IPC_DEPTH]
};
struct ipc_link {
    std::string name;
    ipc_shared* shared = nullptr;
    size_t bytes = 0;
    int world = 0, rank = 0;
    bool owner = false;
    std::vector<unsigned long long> next;                       // per peer: how many sends (on a peer) / receives (on the root) were issued
    std::vector<std::pair<int, unsigned long long>> in_group;   // (peer, sequence) of this group's calls
    hipStream_t last_stream = nullptr;
    std::map<unsigned long long, void*> mapped;                 // a sender: the root's allocations opened so far, by their number
    std::map<void*, std::pair<hipIpcMemHandle_t, unsigned long long>> exported;   // the root: its allocations exported so far, by base address
    ipc_slot& slot(int peer, unsigned long long seq) { return shared->box[(size_t)peer * IPC_DEPTH + seq % IPC_DEPTH]; }
    ~ipc_link() {
        for (auto& kv : mapped) (void)hipIpcCloseMemHandle(kv.second);
        if (shared) munmap(shared, bytes);
        if (owner) shm_unlink(name.c_str());
    }
};
double ipc_timeout_seconds() { const char* e = getenv("GR_TILED_IPC_TIMEOUT"); double v = e ? atof(e) : 60.0; return v > 0 ? v : 60.0; }
template <class F>
bool ipc_wait(F ready) {
    const auto t0 = std::chrono::steady_clock::now();
    const double limit = ipc_timeout_seconds();
    for (int spins = 0; !ready(); spins++) {
        if (spins > 1000) std::this_thread::sleep_for(std::chrono::microseconds(50));
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) return false;
    }
    return true;
}
int ipc_group_begin(void* user) { ((ipc_link*)user)->in_group.clear(); return GR_OK; }
int ipc_recv(void* user, void* data, size_t floats, int peer, void*) {
    ipc_link* l = (ipc_link*)user;
    if (peer < 0 || peer >= l->world) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "ipc transport: peer");
    void* base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange((hipDeviceptr_t*)&base, &size, (hipDeviceptr_t)data) != hipSuccess) {
        (void)hipGetLastError();
        return gr_internal_fail(GR_ERROR_DEVICE, "ipc transport: the receive buffer is not device memory of this process");
    }
    const unsigned long long seq = l->next[(size_t)peer]++;
    ipc_slot& sl = l->slot(peer, seq);
    if (seq >= (unsigned long long)IPC_DEPTH && !ipc_wait([&] { return sl.done.load() >= seq - IPC_DEPTH + 1; }))
        return gr_internal_fail(GR_ERROR_DEVICE, "ipc transport: mailbox full (a peer stopped sending)");
    auto ex = l->exported.find(base);
    if (ex == l->exported.end()) {
        hipIpcMemHandle_t h;
        HIP_CHECK(hipIpcGetMemHandle(&h, base));
        ex = l->exported.emplace(base, std::make_pair(h, (unsigned long long)l->exported.size() + 1)).first;
    }
    sl.handle = ex->second.first;
    sl.allocation = ex->second.second;
    sl.offset = (unsigned long long)((char*)data - (char*)base);
    sl.bytes = (unsigned long long)floats * 4;
    sl.posted.store(seq + 1);
    l->in_group.emplace_back(peer, seq);
    return GR_OK;
}
int ipc_send(void* user, const void* data, size_t floats, int peer, void* stream) {
    ipc_link* l = (ipc_link*)user;
    (void)peer;   // every send of a split frame goes to the root; the mailbox is the sender's own
    const unsigned long long seq = l->next[(size_t)l->rank]++;
    ipc_slot& sl = l->slot(l->rank, seq);
    if (!ipc_wait([&] { return sl.posted.load() >= seq + 1; }))
        return gr_internal_fail(GR_ERROR_DEVICE, "ipc transport: the root did not post the matching receive in time");
    if (sl.bytes != (unsigned long long)floats * 4) return gr_internal_fail(GR_ERROR_DEVICE, "ipc transport: send and receive sizes differ");
    auto it = l->mapped.find(sl.allocation);
    if (it == l->mapped.end()) {
        void* ptr = nullptr;
        HIP_CHECK(hipIpcOpenMemHandle(&ptr, sl.handle, hipIpcMemLazyEnablePeerAccess));
        it = l->mapped.emplace(sl.allocation, ptr).first;
    }
    HIP_CHECK(hipMemcpyAsync((char*)it->second + sl.offset, data, (size_t)sl.bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    l->last_stream = (hipStream_t)stream;
    l->in_group.emplace_back(l->rank, seq);
    return GR_OK;
}
int ipc_group_end(void* user) {
    ipc_link* l = (ipc_link*)user;
    if (l->rank != 0) {   // the copies of this group have left: tell the root
        if (!l->in_group.empty()) HIP_CHECK(hipStreamSynchronize(l->last_stream));
        for (auto& c : l->in_group) l->slot(c.first, c.second).done.store(c.second + 1);
    } else {
        for (auto& c : l->in_group)
            if (!ipc_wait([&] { return l->slot(c.first, c.second).done.load() >= c.second + 1; }))
                return gr_internal_fail(GR_ERROR_DEVICE, "ipc transport: a block did not arrive in time");
    }
    l->in_group.clear();
    return GR_OK;
}

int ring_size() {
    int n = 4;   // frames in flight a participant can have before a frame waits for an earlier frame's transfers
    if (const char* e = getenv("GR_TILED_STAGING")) n = atoi(e);
    return n < 1 ? 1 : n > 64 ? 64 : n;
}

int tiled_common(gr_tiled* t, int world, int rank, int device, int width, int height, int block_rows) {
    if (world < 1 || rank < 0 || rank >= world || width < 1 || height < 1 || block_rows < 8 || block_rows % 8)
        return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "world >= 1, 0 <= rank < world, block_rows a positive multiple of 8");
    if (world > 1 && height > 1 && (height - 1) % block_rows == 0)
        return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "the last image row must not start a block (its filter reads the row above)");
    t->world = world; t->rank = rank; t->device = device; t->width = width; t->height = height; t->block_rows = block_rows;
    const int total_blocks = (height + block_rows - 1) / block_rows;
    t->blocks_per_share = (total_blocks + world - 1) / world;
    t->ring.assign((size_t)ring_size(), frame_slot());
    if (device >= 0) HIP_CHECK(hipSetDevice(device));
    return GR_OK;
}

// the slot of the next frame, ready to be rendered into on `stream`
int next_slot(gr_tiled* t, hipStream_t stream, frame_slot** out) {
    frame_slot& sl = t->ring[(size_t)(t->frames++ % t->ring.size())];
    if (!sl.done) HIP_CHECK(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    if (t->rank != t->root && !sl.buffer) HIP_CHECK(hipMalloc(&sl.buffer, t->staging_bytes()));
    if (sl.recorded) HIP_CHECK(hipStreamWaitEvent(stream, sl.done, 0));   // the frame that used this slot last has left it
    *out = &sl;
    return GR_OK;
}

}  // namespace

extern "C" {

int gr_tiled_unique_id(void* id_out_128_bytes) {
    if (!id_out_128_bytes) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "null argument");
    rccl_api* r = rccl();
    if (!r) return gr_internal_fail(GR_ERROR_DEVICE, "librccl could not be loaded");
    int rc = r->GetUniqueId(id_out_128_bytes);
    return rc == 0 ? GR_OK : rccl_fail("ncclGetUniqueId", rc);
}

int gr_tiled_create(int world, int rank, int device, const void* unique_id_128_bytes, int width, int height, int block_rows, gr_tiled** out) {
    if (!out) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "null argument");
    if (device < 0) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "device");
    auto t = std::make_unique<gr_tiled>();   // ~gr_tiled releases whatever exists on every early return below
    t->transport = GR_TRANSPORT_RCCL;
    int rc = tiled_common(t.get(), world, rank, device, width, height, block_rows);
    if (rc != GR_OK) return rc;
    if (world > 1) {
        if (!unique_id_128_bytes) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "world > 1 needs the id of gr_tiled_unique_id");
        rccl_api* r = rccl();
        if (!r) return gr_internal_fail(GR_ERROR_DEVICE, "librccl could not be loaded");
        unique_id id;
        memcpy(id.bytes, unique_id_128_bytes, sizeof(id.bytes));
        auto init = (int (*)(void**, int, unique_id, int))r->comm_init_rank_raw;   // ncclCommInitRank(comm*, nranks, id BY VALUE, rank)
        int nrc = init(&t->comm, world, id, rank);
        if (nrc != 0) { t->comm = nullptr; return rccl_fail("ncclCommInitRank", nrc); }
        t->link = gr_transport{t.get(), rccl_group_begin, rccl_group_end, rccl_send, rccl_recv};
    }
    *out = t.release();
    return GR_OK;
}

int gr_tiled_create_custom(int world, int rank, int device, const gr_transport* transport, int width, int height, int block_rows, gr_tiled** out) {
    if (!out || !transport || !transport->send || !transport->recv) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "null argument");
    auto t = std::make_unique<gr_tiled>();
    t->transport = GR_TRANSPORT_CUSTOM;
    int rc = tiled_common(t.get(), world, rank, device, width, height, block_rows);
    if (rc != GR_OK) return rc;
    t->link = *transport;
    *out = t.release();
    return GR_OK;
}

int gr_tiled_create_ipc(int world, int rank, int device, const char* session, int width, int height, int block_rows, gr_tiled** out) {
    if (!out || !session || !session[0]) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "null argument");
    if (device < 0) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "device");
    auto t = std::make_unique<gr_tiled>();
    t->transport = GR_TRANSPORT_IPC;
    int rc = tiled_common(t.get(), world, rank, device, width, height, block_rows);
    if (rc != GR_OK) return rc;
    if (world > 1) {
        auto l = std::make_shared<ipc_link>();
        l->name = std::string("/grtiled_") + session;
        l->world = world; l->rank = rank;
        l->next.assign((size_t)world, 0);
        l->bytes = sizeof(ipc_shared) + sizeof(ipc_slot) * ((size_t)world * IPC_DEPTH);
        // the inode behind the session's name right now (0: no such region)
        auto name_inode = [&]() -> unsigned long long {
            const int probe = shm_open(l->name.c_str(), O_RDWR, 0600);
            if (probe < 0) return 0ull;
            struct stat st;
            const unsigned long long ino = fstat(probe, &st) == 0 ? (unsigned long long)st.st_ino : 0ull;
            close(probe);
            return ino;
        };
        if (rank == 0) {
            shm_unlink(l->name.c_str());   // a stale region of an earlier run of the same session
            const int fd = shm_open(l->name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
            if (fd < 0 || ftruncate(fd, (off_t)l->bytes) != 0) { if (fd >= 0) close(fd); return gr_internal_fail(GR_ERROR_DEVICE, "ipc transport: cannot create the shared region"); }
            l->owner = true;
            void* mem = mmap(nullptr, l->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            close(fd);
            if (mem == MAP_FAILED) return gr_internal_fail(GR_ERROR_DEVICE, "ipc transport: mmap failed");
            l->shared = (ipc_shared*)mem;
            memset(mem, 0, l->bytes);
            l->shared->world = (unsigned int)world;
            l->shared->magic.store(0x47524950u);
            // collective, like ncclCommInitRank: returns when every rank has arrived
            l->shared->arrived.fetch_add(1);
            if (!ipc_wait([&] { return l->shared->arrived.load() >= (unsigned int)world; }))
                return gr_internal_fail(GR_ERROR_DEVICE, "ipc transport: not every rank arrived in time");
        } else {
            // A rank that starts before rank 0 can find the region a crashed or finished run of the same session left behind: size, magic
            // and world all look right, and a finished run's arrival counter is already full - it would pass the barrier on the old region
            // while rank 0 unlinks it and makes a new one, and every send would then wait out its timeout (round 5).  So: a region
            // whose counter was full BEFORE this rank arrived is a stale one; and whoever waits on a region keeps watching the session's
            // name - once it points at another inode rank 0 has replaced the region, and the rank starts over on the new one.
            const auto t0 = std::chrono::steady_clock::now();
            auto out_of_time = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > ipc_timeout_seconds(); };
            for (;;) {
                if (out_of_time()) return gr_internal_fail(GR_ERROR_DEVICE, "ipc transport: rank 0's shared region did not appear in time");
                int fd = shm_open(l->name.c_str(), O_RDWR, 0600);
                struct stat st;
                if (fd < 0 || fstat(fd, &st) != 0 || (size_t)st.st_size < l->bytes) {
                    if (fd >= 0) close(fd);
                    std::this_thread::sleep_for(std::chrono::microseconds(200));
                    continue;
                }
                const unsigned long long attached = (unsigned long long)st.st_ino;
                void* mem = mmap(nullptr, l->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
                close(fd);
                if (mem == MAP_FAILED) return gr_internal_fail(GR_ERROR_DEVICE, "ipc transport: mmap failed");
                ipc_shared* region = (ipc_shared*)mem;
                bool replaced = false;
                auto wait_on_region = [&](auto ready) {   // true: ready; false: the region was replaced or time ran out
                    for (int spins = 0; !ready(); spins++) {
                        if (spins > 1000) std::this_thread::sleep_for(std::chrono::microseconds(50));
                        if (spins % 256 == 255 && name_inode() != attached) { replaced = true; return false; }
                        if (out_of_time()) return false;
                    }
                    return true;
                };
                bool joined = wait_on_region([&] { return region->magic.load() == 0x47524950u; });
                if (joined && region->world != (unsigned int)world) {
                    munmap(mem, l->bytes);
                    return gr_internal_fail(GR_ERROR_DEVICE, "ipc transport: the shared region is not rank 0's for this world");
                }
                if (joined) {
                    const unsigned int before_me = region->arrived.fetch_add(1);
                    if (before_me >= (unsigned int)world) {   // everybody of an earlier run had arrived already: not this run's region
                        joined = false;
                        (void)wait_on_region([&] { return false; });   // until rank 0 has replaced it (or time runs out)
                    } else {
                        joined = wait_on_region([&] { return region->arrived.load() >= (unsigned int)world; });
                    }
                }
                // ... and a region a CRASHED run left behind with fewer than `world` arrivals can be filled up to `world` by the early ranks of
                // this run while rank 0 is still on its way to unlink it: the barrier above then completes on the stale inode.  Once more
                // after the barrier: the name must still point at the region this rank is attached to - rank 0 makes its region
                // before it arrives, so a barrier that rank 0 took part in completes on the inode the name has.
                if (joined && name_inode() != attached) { joined = false; replaced = true; }
                if (joined) { l->shared = region; break; }
                munmap(mem, l->bytes);
                if (!replaced) return gr_internal_fail(GR_ERROR_DEVICE, "ipc transport: not every rank arrived in time");
            }
        }
        t->link = gr_transport{l.get(), ipc_group_begin, ipc_group_end, ipc_send, ipc_recv};
        t->ipc = l;
    }
    *out = t.release();
    return GR_OK;
}

int gr_tiled_create_local(int count, const int* devices, int width, int height, int block_rows, gr_tiled** out_array) {
    if (count < 1 || !devices || !out_array) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "null argument");
    auto group = std::make_shared<peer_group>();
    group->world = count;
    group->members.assign(count, nullptr);
    std::vector<std::unique_ptr<gr_tiled>> made;
    for (int r = 0; r < count; r++) {
        if (devices[r] < 0) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "device");
        auto t = std::make_unique<gr_tiled>();
        t->transport = GR_TRANSPORT_PEER;
        t->root_device = devices[0];
        int rc = tiled_common(t.get(), count, r, devices[r], width, height, block_rows);   // selects the device, or fails
        if (rc != GR_OK) return rc;
        if (devices[r] != devices[0]) {
            int can = 0;
            (void)hipDeviceCanAccessPeer(&can, devices[r], devices[0]);
            if (can) { hipError_t e = hipDeviceEnablePeerAccess(devices[0], 0); if (e != hipSuccess) (void)hipGetLastError(); }   // already enabled is fine
        }
        made.push_back(std::move(t));
    }
    for (int r = 0; r < count; r++) {
        made[r]->group = group;
        group->members[r] = made[r].get();
        out_array[r] = made[r].release();
    }
    return GR_OK;
}

void gr_tiled_destroy(gr_tiled* t) { delete t; }

int gr_tiled_share(const gr_tiled* t, int rotation) { return t ? ((t->rank + (rotation % t->world + t->world)) % t->world) : 0; }

int gr_tiled_look_ahead(gr_tiled* t, int rotations) {
    if (!t || rotations < 1) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "gr_tiled_look_ahead: a participant and a distance of at least one rotation");
    t->look_ahead = rotations;
    return GR_OK;
}

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

size_t gr_tiled_staging_bytes(const gr_tiled* t) { return t ? t->staging_bytes() : 0; }

// The transfer of one frame: every block of every other participant's share from its compact place in that participant's
// `staging` to its rows of the root's frame.  Point-to-point transports: one group per frame, on the owner one send per block in
// block order, on the root the matching receives peer by peer in the same block order (sends and receives between two ranks
// match in issue order).
int gr_tiled_exchange(gr_tiled* t, const void* staging, void* frame_on_root, int rotation, void* stream_v) {
    if (!t) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "null argument");
    if (t->world == 1) return GR_OK;
    hipStream_t stream = (hipStream_t)stream_v;
    const bool is_root = t->rank == t->root;
    if (is_root ? !frame_on_root : !staging) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "the root needs the frame, the others their staged rows");
    const int share = gr_tiled_share(t, rotation);
    const size_t row_bytes = (size_t)t->width * 16, row_floats = (size_t)t->width * 4;
    if (t->transport == GR_TRANSPORT_PEER) {
        if (!frame_on_root) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "peer transport: every participant is given the root's frame buffer");
        if (is_root) return GR_OK;
        // peer copies: the owner pushes its blocks into the root's frame on its own stream
        for (int i = 0; i < t->blocks_per_share; i++) {
            int a, b;
            if (gr_tiled_block_rows_of(t, share, i, &a, &b) != 1) continue;
            void* dst = (char*)frame_on_root + (size_t)a * row_bytes;
            const void* src = (const char*)staging + (size_t)i * t->block_rows * row_bytes;
            if (t->device == t->root_device) HIP_CHECK(hipMemcpyAsync(dst, src, (size_t)(b - a) * row_bytes, hipMemcpyDeviceToDevice, stream));
            else HIP_CHECK(hipMemcpyPeerAsync(dst, t->root_device, src, t->device, (size_t)(b - a) * row_bytes, stream));
        }
        return GR_OK;
    }
    const gr_transport& link = t->link;
    int rc = link.group_begin ? link.group_begin(link.user) : GR_OK;
    if (rc != GR_OK) return rc;
    if (is_root) {
        for (int peer = 0; peer < t->world && rc == GR_OK; peer++) {
            if (peer == t->root) continue;
            const int peer_share = (peer + (rotation % t->world + t->world)) % t->world;
            for (int i = 0; i < t->blocks_per_share && rc == GR_OK; i++) {
                int a, b;
                if (gr_tiled_block_rows_of(t, peer_share, i, &a, &b) != 1) continue;
                rc = link.recv(link.user, (char*)frame_on_root + (size_t)a * row_bytes, (size_t)(b - a) * row_floats, peer, stream_v);
            }
        }
    } else {
        for (int i = 0; i < t->blocks_per_share && rc == GR_OK; i++) {
            int a, b;
            if (gr_tiled_block_rows_of(t, share, i, &a, &b) != 1) continue;
            rc = link.send(link.user, (const char*)staging + (size_t)i * t->block_rows * row_bytes, (size_t)(b - a) * row_floats, t->root, stream_v);
        }
    }
    const int end_rc = link.group_end ? link.group_end(link.user) : GR_OK;
    return rc != GR_OK ? rc : end_rc;
}

int gr_render_frame_tiled(gr_tiled* t, gr_render_state* s, gr_program* p, const gr_metric* m, void* stream_v, const gr_camera* camera,
                          const gr_features* features, const float* cfg_values, int num_cfg_values, const void* bg1, const void* bg2,
                          int bg_width, int bg_height, int bg_levels, void* frame_on_root, const gr_frame_options* options, int rotation) {
    if (!t || !s || !p || !m || !camera) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "null argument");
    if (t->device < 0) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "this participant was created without a device (schedule tests): nothing to render with");
    {   // (the staging buffers and the blocks' places in the root's frame are those of the participant's frame size)
        int sw = 0, sh = 0;
        gr_internal_render_state_size(s, &sw, &sh);
        if (sw != t->width || sh != t->height) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "the render state is not of the size the participant was created for");
    }
    hipStream_t stream = (hipStream_t)stream_v;
    const bool is_root = t->rank == t->root;
    if (is_root && !frame_on_root) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "the root needs the frame buffer");
    if (t->transport == GR_TRANSPORT_PEER && !frame_on_root) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "peer transport: every participant is given the root's frame buffer");
    gr_frame_options opt;
    gr_frame_options_default(&opt);
    if (options) opt = *options;
    opt.mode = GR_MODE_FUSED;
    const int share = gr_tiled_share(t, rotation);
    // the shares of the look-ahead frames (options->next_camera, next_camera2): this render state's next two calls, look_ahead rotations
    // apart - unless the caller's own tuning names them
    gr_frame_tuning tune;
    gr_frame_tuning_default(&tune);
    if (opt.tuning) tune = *opt.tuning;
    if (t->world > 1 && opt.next_camera && tune.next_strip_rank < 0) tune.next_strip_rank = gr_tiled_share(t, rotation + t->look_ahead);
    if (t->world > 1 && opt.next_camera2 && tune.next_strip_rank2 < 0) tune.next_strip_rank2 = gr_tiled_share(t, rotation + 2 * t->look_ahead);
    opt.tuning = &tune;
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
    if (t->world == 1)
        return gr_render_frame(s, p, m, stream_v, camera, features, cfg_values, num_cfg_values, bg1, bg2, bg_width, bg_height, bg_levels, frame_on_root, &opt);
    frame_slot* sl = nullptr;
    int rc = next_slot(t, stream, &sl);
    if (rc != GR_OK) return rc;
    const int render_rc = gr_render_frame(s, p, m, stream_v, camera, features, cfg_values, num_cfg_values, bg1, bg2, bg_width, bg_height, bg_levels,
                                          is_root ? frame_on_root : sl->buffer, &opt);
    std::string render_error = render_rc != GR_OK ? gr_last_error() : "";
    // The transfers are issued even when this participant's render failed: the others have matching sends / receives in their
    // groups and would wait for ever.  The caller sees the render's error.
    rc = gr_tiled_exchange(t, sl->buffer, frame_on_root, rotation, stream_v);
    HIP_CHECK(hipEventRecord(sl->done, stream));
    sl->recorded = true;
    sl->pending = t->transport == GR_TRANSPORT_PEER;
    if (render_rc != GR_OK) return gr_internal_fail(render_rc, render_error.c_str());
    return rc;
}

int gr_tiled_join(gr_tiled* root, void* stream_v) {
    if (!root) return gr_internal_fail(GR_ERROR_INVALID_ARGUMENT, "null argument");
    if (root->transport != GR_TRANSPORT_PEER || root->world == 1) return GR_OK;   // point-to-point: the receives are ordered on the root's stream
    HIP_CHECK(hipSetDevice(root->device));
    for (gr_tiled* member : root->group->members) {
        if (!member) continue;
        for (auto& sl : member->ring) {
            if (!sl.pending) continue;
            HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream_v, sl.done, 0));
            sl.pending = false;
        }
    }
    return GR_OK;
}

}  // extern "C"
