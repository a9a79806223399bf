// render_tiled.cpp — one frame split over N GPUs through the C ABI alone (include/geodesic_hip.h): one process per GPU, RCCL over
// xGMI for the finished rows, no Python, no torch, no MPI.  The 128-byte communicator id travels through a file.
//
//   make -C examples
//   examples/render_tiled --spawn 8 geodesic_raytracing_amd/scripts kerr_boyer 3840 2160 frame.png [frames] [name=value ...]
//       forks one worker per GPU (rank r -> device r) and waits for them; or, started by any launcher of your own, one process each:
//   examples/render_tiled --rank R --world N --device D --id-file /tmp/id geodesic_raytracing_amd/scripts kerr_boyer 3840 2160 frame.png
//   examples/render_tiled --spawn 3 --one-device 1 ...   rehearsal on a box with one GPU: every rank on device 0, RCCL through sockets
//   examples/render_tiled --spawn 8 --self-test 1 ...    first contact with a new node: N ranks on whatever GPUs there are (rank r on
//       device r % count; fewer GPUs than ranks: RCCL through sockets as above), rank 0 compares the gathered frame with the same frame
//       rendered on its own GPU alone, and the launcher prints which RCCL transport every channel took (P2P/IPC over xGMI, SHM, NET)
//
// Every rank renders its (rotating) share of the image rows with frames in flight on streams of their own, rank 0 receives
// every block at its place in the frame (gr_render_frame_tiled), writes the last frame as a PNG and prints frames per second
// with and without the transfer (a second pass renders the same shares without sending them).
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "geodesic_hip.h"

#define CHECK(call)                                                                          \
    do {                                                                                     \
        if ((call) != GR_OK) {                                                               \
            std::fprintf(stderr, "[rank %d] %s failed: %s\n", g_rank, #call, gr_last_error()); \
            return 1;                                                                        \
        }                                                                                    \
    } while (0)

static int g_rank = 0;
static int g_self_test = 0;

// the id of gr_tiled_unique_id from rank 0 to everybody: written to a temporary name and renamed, so a reader sees all of it or nothing
static bool publish_id(const std::string& path, const unsigned char id[128]) {
    const std::string tmp = path + ".tmp";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return false;
    const bool ok = std::fwrite(id, 1, 128, f) == 128;
    std::fclose(f);
    return ok && std::rename(tmp.c_str(), path.c_str()) == 0;
}

static bool await_id(const std::string& path, unsigned char id[128], double seconds) {
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(seconds);
    while (std::chrono::steady_clock::now() < deadline) {
        if (FILE* f = std::fopen(path.c_str(), "rb")) {
            const size_t n = std::fread(id, 1, 128, f);
            std::fclose(f);
            if (n == 128) return true;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
    return false;
}

static int worker(int world, int rank, int device, const std::string& id_file, int argc, char** argv) {
    g_rank = rank;
    const char* scripts = argv[0];
    const char* name = argv[1];
    const int width = std::atoi(argv[2]), height = std::atoi(argv[3]);
    const char* out_path = argv[4];
    int frames = 12;
    int first_override = 5;
    if (argc > 5 && !std::strchr(argv[5], '=')) { frames = std::atoi(argv[5]); first_override = 6; }
    const int block_rows = 48, in_flight = 3;

    if (g_self_test) {
        // whatever GPUs this node has: rank r on device r % count; with fewer GPUs than ranks every rank claims a host of its own so
        // that RCCL accepts two of them on one device (socket transport over loopback - the exchange's code, not its speed)
        int count = 0;
        CHECK(gr_device_count(&count));
        if (count < 1) { std::fprintf(stderr, "[rank %d] no GPU\n", rank); return 1; }
        device = rank % count;
        if (count < world) {
            setenv("NCCL_HOSTID", ("render-tiled-" + std::to_string((long)getppid()) + "-rank" + std::to_string(rank)).c_str(), 1);
            setenv("NCCL_SOCKET_IFNAME", "lo", 1);
            setenv("NCCL_IB_DISABLE", "1", 1);
            setenv("NCCL_NET_GDR_LEVEL", "0", 1);
        }
        if (rank == 0) std::printf("self-test: %d ranks on %d GPU(s)%s\n", world, count, count < world ? " (several ranks to a device: RCCL over sockets)" : "");
    }
    gr_metric* metric = nullptr;
    CHECK(gr_metric_load_script(scripts, name, &metric));
    gr_metric_info info;
    CHECK(gr_metric_get_info(metric, &info));
    std::vector<float> cfg(info.num_dynamic_vars > 0 ? info.num_dynamic_vars : 1, 0.f);
    for (int i = 0; i < info.num_dynamic_vars; i++) cfg[i] = gr_metric_dynamic_var_default(metric, i);
    for (int a = first_override; a < argc; a++) {
        const char* eq = std::strchr(argv[a], '=');
        if (!eq) continue;
        const std::string key(argv[a], eq - argv[a]);
        for (int i = 0; i < info.num_dynamic_vars; i++)
            if (key == gr_metric_dynamic_var_name(metric, i)) cfg[i] = (float)std::atof(eq + 1);
    }
    gr_features features;
    gr_features_default(&features);
    features.adaptive_sampling = 0;
    features.max_acceleration_change = info.max_acceleration_change;

    size_t need = 0;
    CHECK(gr_metric_argument_string(metric, &features, 1, cfg.data(), info.num_dynamic_vars, nullptr, 0, &need));
    std::string arguments(need, '\0');
    CHECK(gr_metric_argument_string(metric, &features, 1, cfg.data(), info.num_dynamic_vars, arguments.data(), need, &need));
    gr_program* program = nullptr;
    CHECK(gr_program_create(arguments.c_str(), device, &program));

    // the communicator: rank 0 makes the id, everybody joins (collective)
    unsigned char id[128] = {};
    if (world > 1) {
        if (rank == 0) {
            CHECK(gr_tiled_unique_id(id));
            if (!publish_id(id_file, id)) { std::fprintf(stderr, "cannot write %s\n", id_file.c_str()); return 1; }
        } else if (!await_id(id_file, id, 120.0)) {
            std::fprintf(stderr, "[rank %d] no communicator id in %s after 120 s\n", rank, id_file.c_str());
            return 1;
        }
    }
    gr_tiled* tiled = nullptr;
    CHECK(gr_tiled_create(world, rank, device, world > 1 ? id : nullptr, width, height, block_rows, &tiled));
    CHECK(gr_tiled_look_ahead(tiled, in_flight));   // a render state's next frame is in_flight rotations on: its look-ahead prepass is for that share

    const int bw = 2048, bh = 1024;
    std::vector<unsigned char> sky((size_t)bw * bh * 4);
    for (int y = 0; y < bh; y++)
        for (int x = 0; x < bw; x++) {
            const bool line = (x % (bw / 36) == 0) || (y % (bh / 18) == 0);
            unsigned char* p = &sky[((size_t)y * bw + x) * 4];
            p[0] = line ? 255 : (unsigned char)(40 + 100 * x / bw);
            p[1] = line ? 255 : (unsigned char)(40 + 120 * y / bh);
            p[2] = line ? 255 : 160;
            p[3] = 255;
        }
    const int levels = gr_pack_mipped_background(sky.data(), bw, bh, nullptr);
    std::vector<unsigned char> packed((size_t)levels * bw * bh * 4);
    if (gr_pack_mipped_background(sky.data(), bw, bh, packed.data()) != levels) return 1;
    void* d_background = nullptr;
    CHECK(gr_device_alloc(device, packed.size(), &d_background));
    CHECK(gr_device_upload(device, d_background, packed.data(), packed.size()));

    // a ring of frames in flight, as the reference's ring of render_state objects (main.cpp:1463-1469): state, stream and - on
    // rank 0 - frame buffer each
    std::vector<gr_render_state*> states(in_flight, nullptr);
    std::vector<void*> streams(in_flight, nullptr), frames_on_root(in_flight, nullptr);
    for (int j = 0; j < in_flight; j++) {
        CHECK(gr_render_state_create(device, width, height, &states[j]));
        CHECK(gr_stream_create(device, 0, &streams[j]));
        if (rank == 0) CHECK(gr_device_alloc(device, (size_t)width * height * 16, &frames_on_root[j]));
    }
    gr_camera camera;
    gr_camera_default(&camera);

    auto run = [&](int count, bool transfer, double& seconds) -> int {
        for (int j = 0; j < in_flight; j++) CHECK(gr_stream_synchronize(streams[j]));
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < count; k++) {
            const int j = k % in_flight;
            gr_frame_options options;
            gr_frame_options_default(&options);
            options.next_camera = &camera;   // a batch renderer knows the next camera: its prepass runs ahead on a side stream
            if (transfer) {
                CHECK(gr_render_frame_tiled(tiled, states[j], program, metric, streams[j], &camera, &features, cfg.data(), info.num_dynamic_vars,
                                            d_background, d_background, bw, bh, levels, frames_on_root[j], &options, k));
            } else {   // the same share, rendered and left where it is
                options.strip_count = world;
                options.strip_rank = gr_tiled_share(tiled, k);
                options.block_rows = block_rows;
                options.compact_out = 0;
                static void* scratch = nullptr;
                if (!scratch) CHECK(gr_device_alloc(device, (size_t)width * height * 16, &scratch));
                CHECK(gr_render_frame(states[j], program, metric, streams[j], &camera, &features, cfg.data(), info.num_dynamic_vars, d_background,
                                      d_background, bw, bh, levels, scratch, &options));
            }
        }
        for (int j = 0; j < in_flight; j++) CHECK(gr_stream_synchronize(streams[j]));
        seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return 0;
    };
    double warm = 0, with_transfer = 0, without_transfer = 0;
    if (run(in_flight, true, warm)) return 1;                 // every state has rendered once, the communicator is up
    if (run(frames, true, with_transfer)) return 1;
    if (run(frames, false, without_transfer)) return 1;

    if (rank == 0) {
        const int last = (frames - 1) % in_flight;
        std::vector<float> frame((size_t)width * height * 4);
        CHECK(gr_device_download(device, frame.data(), frames_on_root[last], frame.size() * sizeof(float)));
        CHECK(gr_write_frame_png(out_path, frame.data(), width, height));
        std::printf("%s %dx%d over %d GPU(s), %d frames, %d in flight: %.1f frames/s with the transfer to rank 0, %.1f without; wrote %s\n", name, width,
                    height, world, frames, in_flight, frames / with_transfer, frames / without_transfer, out_path);
        if (g_self_test) {
            // the same frame on this GPU alone (the split changes who traces a row, not what is traced: the frames are equal bit for bit)
            gr_render_state* whole = nullptr;
            void* d_whole = nullptr;
            CHECK(gr_render_state_create(device, width, height, &whole));
            CHECK(gr_device_alloc(device, (size_t)width * height * 16, &d_whole));
            gr_frame_options options;
            gr_frame_options_default(&options);
            options.mode = GR_MODE_FUSED;
            CHECK(gr_render_frame(whole, program, metric, streams[0], &camera, &features, cfg.data(), info.num_dynamic_vars, d_background, d_background,
                                  bw, bh, levels, d_whole, &options));
            CHECK(gr_stream_synchronize(streams[0]));
            std::vector<float> alone((size_t)width * height * 4);
            CHECK(gr_device_download(device, alone.data(), d_whole, alone.size() * sizeof(float)));
            size_t differing = 0;
            double worst = 0;
            for (size_t i = 0; i < alone.size(); i++) {
                const double d = std::fabs((double)alone[i] - (double)frame[i]);
                if (d > 0) differing++;
                if (d > worst || d != d) worst = d != d ? 1e30 : d;
            }
            std::printf("self-test: gathered frame against the single-GPU frame: %zu of %zu values differ, largest difference %.3g -> %s\n", differing,
                        alone.size(), worst, differing == 0 ? "IDENTICAL" : worst <= 1e-3 ? "within 1e-3" : "MISMATCH");
            gr_device_free(device, d_whole);
            gr_render_state_destroy(whole);
            if (worst > 1e-3) return 1;
        }
    }
    gr_tiled_destroy(tiled);
    for (int j = 0; j < in_flight; j++) {
        gr_stream_destroy(streams[j]);
        gr_render_state_destroy(states[j]);
        if (frames_on_root[j]) gr_device_free(device, frames_on_root[j]);
    }
    gr_device_free(device, d_background);
    gr_program_destroy(program);
    gr_metric_destroy(metric);
    return 0;
}

int main(int argc, char** argv) {
    int world = 1, rank = 0, device = -1, spawn = 0, one_device = 0;
    std::string id_file;
    int a = 1;
    for (; a < argc && std::strncmp(argv[a], "--", 2) == 0; a += 2) {
        if (a + 1 >= argc) break;
        if (!std::strcmp(argv[a], "--spawn")) spawn = std::atoi(argv[a + 1]);
        else if (!std::strcmp(argv[a], "--world")) world = std::atoi(argv[a + 1]);
        else if (!std::strcmp(argv[a], "--rank")) rank = std::atoi(argv[a + 1]);
        else if (!std::strcmp(argv[a], "--device")) device = std::atoi(argv[a + 1]);
        else if (!std::strcmp(argv[a], "--id-file")) id_file = argv[a + 1];
        else if (!std::strcmp(argv[a], "--one-device")) one_device = std::atoi(argv[a + 1]);
        else if (!std::strcmp(argv[a], "--self-test")) g_self_test = std::atoi(argv[a + 1]);
        else { std::fprintf(stderr, "unknown option %s\n", argv[a]); return 2; }
    }
    if (argc - a < 5) {
        std::fprintf(stderr, "usage: %s (--spawn N [--one-device 1] | --world N --rank R [--device D] --id-file PATH) <scripts dir> <metric> <width> <height> <out.png> "
                             "[frames] [name=value ...]\n", argv[0]);
        return 2;
    }
    if (spawn > 0) {
        // one worker per GPU; the devices are not touched in this process (a forked child must not inherit an initialised runtime)
        id_file = "/tmp/gr_tiled_id." + std::to_string((long)getpid());
        std::remove(id_file.c_str());
        std::vector<pid_t> children;
        const std::string rccl_log = "/tmp/gr_tiled_rccl." + std::to_string((long)getpid());
        for (int r = 0; r < spawn; r++) {
            const pid_t pid = fork();
            if (pid == 0) {
                if (g_self_test) {   // RCCL says which transport every channel took; the launcher reads it back below
                    setenv("NCCL_DEBUG", "INFO", 1);
                    setenv("NCCL_DEBUG_SUBSYS", "INIT,P2P,NET,SHM", 1);
                    setenv("NCCL_DEBUG_FILE", (rccl_log + ".rank" + std::to_string(r)).c_str(), 1);
                }
                if (one_device) {
                    // rehearsal on a box with fewer GPUs than ranks: RCCL refuses two ranks of one host on one device, so every
                    // rank claims a host of its own and RCCL talks through its socket transport over the loopback interface
                    setenv("NCCL_HOSTID", ("render-tiled-" + std::to_string((long)getppid()) + "-rank" + std::to_string(r)).c_str(), 1);
                    setenv("NCCL_SOCKET_IFNAME", "lo", 1);
                    setenv("NCCL_IB_DISABLE", "1", 1);
                    setenv("NCCL_NET_GDR_LEVEL", "0", 1);
                }
                const int rc = worker(spawn, r, one_device ? 0 : r, id_file, argc - a, argv + a);
                std::fflush(nullptr);   // _exit does not flush stdio
                _exit(rc);
            }
            if (pid < 0) { std::perror("fork"); return 1; }
            children.push_back(pid);
        }
        int failed = 0;
        for (pid_t pid : children) {
            int status = 0;
            waitpid(pid, &status, 0);
            failed += !(WIFEXITED(status) && WEXITSTATUS(status) == 0);
        }
        std::remove(id_file.c_str());
        if (g_self_test) {
            // "... Channel 03/1 : 2[2] -> 0[0] via P2P/IPC" and the like: count the channels by what follows "via"
            std::vector<std::pair<std::string, int>> transports;
            for (int r = 0; r < spawn; r++) {
                const std::string path = rccl_log + ".rank" + std::to_string(r);
                if (FILE* f = std::fopen(path.c_str(), "r")) {
                    char line[1024];
                    while (std::fgets(line, sizeof(line), f)) {
                        const char* via = std::strstr(line, " via ");
                        if (!via || !std::strstr(line, "Channel")) continue;
                        std::string kind(via + 5);
                        kind = kind.substr(0, kind.find_first_of(" \n"));   // "P2P/IPC", "NET/Socket/0" - what follows is the communicator
                        bool found = false;
                        for (auto& t : transports) if (t.first == kind) { t.second++; found = true; }
                        if (!found) transports.emplace_back(kind, 1);
                    }
                    std::fclose(f);
                    std::remove(path.c_str());
                }
            }
            std::printf("self-test: RCCL channels by transport:");
            if (transports.empty()) std::printf(" none reported (one rank, or RCCL's log is not where NCCL_DEBUG_FILE says)");
            for (auto& t : transports) std::printf("  %s x %d", t.first.c_str(), t.second);
            std::printf("\n");
        }
        return failed ? 1 : 0;
    }
    if (world > 1 && id_file.empty()) { std::fprintf(stderr, "--world > 1 needs --id-file\n"); return 2; }
    return worker(world, rank, device >= 0 ? device : rank, id_file, argc - a, argv + a);
}
