// render_kerr.cpp — the C ABI (include/geodesic_hip.h) used from C++ with nothing else: no Python, no torch, no HIP headers.
// What a host program such as the reference's main.cpp does per frame (metric script -> program -> frame -> screenshot):
//
//   make -C examples        (g++ -Iinclude examples/render_kerr.cpp -Lgeodesic_raytracing_amd -lgeodesic_hip, rpath to the library)
//   examples/render_kerr geodesic_raytracing_amd/scripts kerr_boyer 1920 1080 kerr.png a=0.45
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "geodesic_hip.h"

#define CHECK(call)                                                                 \
    do {                                                                            \
        if ((call) != GR_OK) {                                                      \
            std::fprintf(stderr, "%s failed: %s\n", #call, gr_last_error());        \
            return 1;                                                               \
        }                                                                           \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 6) {
        std::fprintf(stderr, "usage: %s <scripts dir> <metric> <width> <height> <out.png> [name=value ...]\n", argv[0]);
        return 2;
    }
    const char* scripts = argv[1];
    const char* name = argv[2];
    const int width = std::atoi(argv[3]), height = std::atoi(argv[4]);
    const char* out_path = argv[5];

    // 1. metric script -> symbolic metric -> macro string (content_manager + js_interop + metric.hpp in the reference)
    gr_metric* metric = nullptr;
    CHECK(gr_metric_load_script(scripts, name, &metric));
    gr_metric_info info;
    CHECK(gr_metric_get_info(metric, &info));
    std::vector<float> cfg(info.num_dynamic_vars > 0 ? info.num_dynamic_vars : 1, 0.f);
    for (int i = 0; i < info.num_dynamic_vars; i++) cfg[i] = gr_metric_dynamic_var_default(metric, i);
    for (int a = 6; a < argc; a++) {   // name=value overrides of the script's $cfg parameters
        const char* eq = std::strchr(argv[a], '=');
        if (!eq) continue;
        std::string key(argv[a], eq - argv[a]);
        for (int i = 0; i < info.num_dynamic_vars; i++)
            if (key == gr_metric_dynamic_var_name(metric, i)) cfg[i] = (float)std::atof(eq + 1);
    }
    gr_features features;
    gr_features_default(&features);
    features.adaptive_sampling = 0;
    features.max_acceleration_change = info.max_acceleration_change;   // metric_manager.hpp:50

    // 2. programs, the reference's way (metric_manager.hpp:19-219): the dynamic program is there at once, the substituted one
    //    (parameters and features baked in) builds in the background; an interactive host would call
    //    gr_program_manager_current(manager, 0, ...) once per frame and render with whatever it hands back - a one-frame program
    //    waits for the substituted build
    gr_program_manager* manager = nullptr;
    CHECK(gr_program_manager_create(metric, 0, &features, cfg.data(), info.num_dynamic_vars, &manager));
    gr_program* program = nullptr;
    int substituted = 0;
    CHECK(gr_program_manager_current(manager, 1, &program, &substituted));

    // 3. background: a procedural equirectangular sky (10-degree grid on a gradient), packed with its mip chain
    const int bw = 2048, bh = 1024;
    std::vector<unsigned char> sky((size_t)bw * bh * 4);
    for (int y = 0; y < bh; y++)
        for (int x = 0; x < bw; x++) {
            bool line = (x % (bw / 36) == 0) || (y % (bh / 18) == 0);
            unsigned char* p = &sky[((size_t)y * bw + x) * 4];
            p[0] = line ? 255 : (unsigned char)(40 + 100 * x / bw);
            p[1] = line ? 255 : (unsigned char)(40 + 120 * y / bh);
            p[2] = line ? 255 : 160;
            p[3] = 255;
        }
    const int levels = gr_pack_mipped_background(sky.data(), bw, bh, nullptr);
    std::vector<unsigned char> packed((size_t)levels * bw * bh * 4);
    if (gr_pack_mipped_background(sky.data(), bw, bh, packed.data()) != levels) return 1;
    void *d_background = nullptr, *d_frame = nullptr;
    CHECK(gr_device_alloc(0, packed.size(), &d_background));
    CHECK(gr_device_upload(0, d_background, packed.data(), packed.size()));
    CHECK(gr_device_alloc(0, (size_t)width * height * 16, &d_frame));

    // 4. one frame (main.cpp:2244-2526 in one call) on a stream of the library's runtime
    gr_render_state* state = nullptr;
    CHECK(gr_render_state_create(0, width, height, &state));
    void* stream = nullptr;
    CHECK(gr_stream_create(0, 0, &stream));
    gr_camera camera;
    gr_camera_default(&camera);
    gr_frame_options options;
    gr_frame_options_default(&options);
    options.time_kernels = 1;
    CHECK(gr_render_frame(state, program, metric, stream, &camera, &features, cfg.data(), info.num_dynamic_vars, d_background, d_background,
                          bw, bh, levels, d_frame, &options));
    CHECK(gr_stream_synchronize(stream));
    float trace_ms = 0;
    CHECK(gr_render_state_stage_ms(state, GR_STAGE_TRACE, &trace_ms));

    // 5. screenshot (main.cpp:2762-2808)
    std::vector<float> frame((size_t)width * height * 4);
    CHECK(gr_device_download(0, frame.data(), d_frame, frame.size() * sizeof(float)));
    CHECK(gr_write_frame_png(out_path, frame.data(), width, height));
    std::printf("%s %dx%d (%s program): trace %.3f ms, wrote %s\n", name, width, height, substituted ? "substituted" : "dynamic", trace_ms, out_path);

    gr_stream_destroy(stream);
    gr_render_state_destroy(state);
    gr_device_free(0, d_frame);
    gr_device_free(0, d_background);
    gr_program_manager_destroy(manager);   // owns the programs
    gr_metric_destroy(metric);
    return 0;
}
