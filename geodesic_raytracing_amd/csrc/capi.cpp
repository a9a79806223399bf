// capi.cpp — the C ABI of libgeodesic_hip.so (include/geodesic_hip.h): metric -> macro string on the
// host, macro string -> gfx950 code object through hiprtc, and one launcher per reference kernel.
//
// Reference counterparts: metric_manager.hpp:19-219 (program build + cache), main.cpp:139-205 and
// 2244-2526 (the launches).  HIP is used directly (module API); there is no fallback of any kind:
// without libamdhip64/hiprtc or without a device the calls fail with GR_ERROR_DEVICE/COMPILE.
#include "../../include/geodesic_hip_internal.h"

#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <hip/hiprtc.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <condition_variable>
#include <mutex>
#include <sstream>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "builtin_metrics.hpp"
#include "codeobject.hpp"
#include "jsfront.hpp"
#include "metric_codegen.hpp"

namespace {

thread_local std::string g_error;

int fail(gr_status code, const std::string& msg) {
    g_error = msg;
    return (int)code;
}

#ifndef GR_DEFAULT_SPECULATIVE_CLASSES
#define GR_DEFAULT_SPECULATIVE_CLASSES 5   // gr_trace_fused: the list's first classes do not wait for the prepass cells of their own launch
#endif

#define GR_TRY_BEGIN try {
#define GR_TRY_END                                                                 \
    }                                                                              \
    catch (const std::exception& e) { return fail(GR_ERROR_SCRIPT, e.what()); }    \
    catch (...) { return fail(GR_ERROR_SCRIPT, "unknown exception"); }

#define HIP_CHECK(expr)                                                                              \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess)                                                                        \
            return fail(GR_ERROR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));         \
    } while (0)

std::string library_dir() {
    Dl_info info;
    if (dladdr((void*)&gr_last_error, &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        size_t s = p.rfind('/');
        if (s != std::string::npos) return p.substr(0, s);
    }
    return ".";
}

bool read_file(const std::string& path, std::string& out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::stringstream ss;
    ss << f.rdbuf();
    out = ss.str();
    return true;
}

uint64_t fnv1a(const std::string& s, uint64_t h = 1469598103934665603ull) {
    for (unsigned char c : s) {
        h ^= c;
        h *= 1099511628211ull;
    }
    return h;
}

#ifndef GR_DEFAULT_VECTOR_RUN_LIMIT
#define GR_DEFAULT_VECTOR_RUN_LIMIT 8
#endif

const char* const KERNEL_NAMES[] = {
    "gr_cart_to_generic", "gr_init_basis_vectors", "gr_clear_termination_buffer", "gr_init_rays_generic",
    "gr_do_generic_rays", "gr_calculate_singularities", "gr_calculate_render_data",
    "gr_handle_adaptive_sampling", "gr_render", "gr_trace_fused", "gr_trace_fused_lattice", "gr_trace_pair", "gr_trace_compact", "gr_prepass_fused", "gr_camera_setup", "gr_order_tiles", "gr_adaptive_refine", "gr_trace_pending", "gr_apply_guessed", "gr_do_generic_rays_scheduled", "gr_sort_tiles_count", "gr_sort_tiles_place", "gr_trace_fused_parking", "gr_boost_tetrad", "gr_init_inertial_ray",
    "gr_get_geodesic_path", "gr_parallel_transport_quantity", "gr_handle_interpolating_geodesic"};
enum KernelId {
    K_CART_TO_GENERIC, K_INIT_BASIS, K_CLEAR_TERM, K_INIT_RAYS, K_DO_RAYS, K_CALC_SING, K_CALC_RDATA,
    K_ADAPTIVE, K_RENDER, K_TRACE_FUSED, K_TRACE_FUSED_LATTICE, K_TRACE_PAIR, K_TRACE_COMPACT, K_PREPASS_FUSED, K_CAMERA_SETUP, K_ORDER_TILES, K_ADAPTIVE_REFINE, K_TRACE_PENDING, K_APPLY_GUESSED, K_DO_RAYS_SCHEDULED, K_SORT_TILES_COUNT, K_SORT_TILES_PLACE, K_TRACE_FUSED_PARKING, K_BOOST_TETRAD, K_INIT_INERTIAL, K_GEODESIC_PATH, K_PARALLEL_TRANSPORT,
    K_INTERPOLATE_GEODESIC, K_COUNT
};

// the kernels of the set-up module (kernels/camera.hip, geodesic_camera.hip): once per frame, one lane, IEEE arithmetic
bool is_setup_kernel(int k) {
    return k == K_CART_TO_GENERIC || k == K_INIT_BASIS || k == K_CAMERA_SETUP || k == K_BOOST_TETRAD || k == K_INIT_INERTIAL ||
           k == K_GEODESIC_PATH || k == K_PARALLEL_TRANSPORT || k == K_INTERPOLATE_GEODESIC;
}

std::vector<std::string> split_arguments(const std::string& s) {
    std::vector<std::string> out;
    std::istringstream iss(s);
    std::string tok;
    while (iss >> tok) out.push_back(tok);
    return out;
}

// gr_trace_pair (two rays per lane, packed fp32) is built when
//  * the expressions evaluated inside the Verlet loop can be instantiated on pairs of floats - no `?:` (the comparison /
//    select forms of CMath.lt, CMath.select, csqrt) - and are small enough that two rays' temporaries fit the register
//    file (the complex-valued double-Kerr family already needs 170-260 VGPRs for one ray), and
//  * the program steps with the fixed heuristic step (no ADAPTIVE_PRECISION).  Measured on MI355X, 4K frames, substituted
//    programs: Schwarzschild 1.96 -> 1.38 ms, Minkowski 1.80 -> 1.16, wormhole 1.83 -> 1.42; with the adaptive controller
//    Kerr 7.9 -> 9.3 ms, Alcubierre 3.0 -> 3.3: the controller (sqrt, rsq, clamps, compares, the per-ray commit) has no
//    packed form, costs twice per lane what it costs the one-ray kernel per lane, and at 133 instead of 92 VGPRs only three
//    waves per SIMD are left to hide its serial tail - that outweighs what the packed multiplies save (EXPERIMENTS.md C.2).
// GR_TRACE_PAIR_BUILD=0 never builds it, =1 builds it for adaptive programs too (it is correct there, only slower).
bool pair_kernel_applies(const std::vector<std::string>& opts) {
    int mode = -1;
    if (const char* e = getenv("GR_TRACE_PAIR_BUILD")) mode = e[0] == '0' ? 0 : e[0] == '1' ? 1 : -1;
    if (mode == 0) return false;
    static const char* const LOOP_MACROS[] = {"-DGEO_ACCEL", "-DTEMPORARIES0=", "-DTO_COORD", "-DDISTANCE_FUNC="};
    size_t total = 0;
    for (auto& o : opts) {
        if (mode != 1 && o == "-DADAPTIVE_PRECISION") return false;
        for (const char* m : LOOP_MACROS)
            if (o.rfind(m, 0) == 0) {
                if (o.find('?') != std::string::npos) return false;
                total += o.size();
            }
    }
    return total > 0 && total < 16384;
}

// Do the expressions the Verlet loop evaluates call the range-limited sin / cos (kernels/metric.hip GR_ACCEL_TRIG: the bare polynomials,
// which answer an argument of 8 192 or more with a NaN so that the ray leaves the fast loop for the one that calls libm)?  A program whose
// accelerations hold none - every Cartesian chart: Kerr-Schild, Alcubierre, Krasnikov ... - can never see such a NaN: its loop then treats
// a non-finite rejected attempt the reference's way (retried with the smaller step in the same loop: -DGR_ACCEL_WITHOUT_TRIG,
// integrator.hip) instead of leaving for the slow loop at the first overshoot into a singularity.
bool accelerations_without_trig(const std::vector<std::string>& opts) {
    static const char* const CALLS[] = {"sin(", "cos(", "gr_sin2(", "gr_cos2(", "gr_sincos("};
    bool any = false;
    for (auto& o : opts) {
        const size_t eq = o.find('=');
        if (o.rfind("-D", 0) != 0 || eq == std::string::npos) continue;
        const std::string name = o.substr(2, eq - 2);
        if (name.find("ACCEL") == std::string::npos && name.find("TEMPORARIES") == std::string::npos) continue;
        any = true;
        for (const char* call : CALLS)
            for (size_t at = o.find(call, eq); at != std::string::npos; at = o.find(call, at + 1)) {
                const char before = o[at - 1];
                if (!(isalnum((unsigned char)before) || before == '_')) return false;   // ("asin(", "gm_cos(" ... are other functions)
            }
    }
    return any;
}

// VGPRs and scratch bytes per lane of one kernel, read from the code object's metadata note (msgpack: the kernel's map holds
// ".name", later ".private_segment_fixed_size" and ".vgpr_count" - keys are sorted).  false when the note is not understood.
bool kernel_resources(const std::string& code, const char* kernel, int& vgprs, int& scratch_bytes, int* sgprs = nullptr) {
    auto msgpack_uint = [&](size_t at, long& value) -> bool {
        if (at >= code.size()) return false;
        const unsigned char c = (unsigned char)code[at];
        if (c < 0x80) { value = c; return true; }
        if (c == 0xcc && at + 1 < code.size()) { value = (unsigned char)code[at + 1]; return true; }
        if (c == 0xcd && at + 2 < code.size()) { value = ((unsigned char)code[at + 1] << 8) | (unsigned char)code[at + 2]; return true; }
        if (c == 0xce && at + 4 < code.size()) {
            value = ((long)(unsigned char)code[at + 1] << 24) | ((unsigned char)code[at + 2] << 16) | ((unsigned char)code[at + 3] << 8) | (unsigned char)code[at + 4];
            return true;
        }
        return false;
    };
    const std::string name_key = std::string(".name") + (char)(0xa0 + strlen(kernel)) + kernel;   // fixstr key, fixstr value (< 32 chars)
    if (strlen(kernel) >= 32) return false;
    size_t at = code.find(name_key);
    if (at == std::string::npos) return false;
    const std::string scratch_key = ".private_segment_fixed_size", vgpr_key = ".vgpr_count";
    size_t s = code.find(scratch_key, at), v = code.find(vgpr_key, at);
    if (s == std::string::npos || v == std::string::npos) return false;
    long sv = 0, vv = 0;
    if (!msgpack_uint(s + scratch_key.size(), sv) || !msgpack_uint(v + vgpr_key.size(), vv)) return false;
    vgprs = (int)vv;
    scratch_bytes = (int)sv;
    if (sgprs) {
        const std::string sgpr_key = ".sgpr_count";
        const size_t g = code.find(sgpr_key, at);
        long gv = 0;
        *sgprs = (g != std::string::npos && g < v && msgpack_uint(g + sgpr_key.size(), gv)) ? (int)gv : 0;
    }
    return vgprs > 0 && vgprs <= 512;
}

// Waves per SIMD a kernel of 256-thread workgroups is resident with on gfx950, by registers.  Vector registers: 512 per lane in
// granules of 8.  Scalar registers: 800 per SIMD, a wave takes its count + 6 (VCC, flat scratch, XNACK) rounded up to 16, plus 16 -
// measured with a timeline of tile begin / end stamps (tools/timeline_probe.py): the Kerr kernel at 72 VGPRs and 94 SGPRs holds 6
// waves per SIMD, not the 7 its vector registers allow; capped to 90 or 78 SGPRs it holds 7; at 64 VGPRs and <= 74 SGPRs 8.
int resident_waves_per_simd(int vgprs, int sgprs) {
    int by_vgprs = 512 / (((vgprs + 7) / 8) * 8);
    int by_sgprs = sgprs > 0 ? 800 / ((((sgprs + 6) + 15) / 16) * 16 + 16) : 8;
    int w = by_vgprs < by_sgprs ? by_vgprs : by_sgprs;
    return w > 8 ? 8 : w;
}

// Compiles (or fetches from the on-disk cache) the code object for one macro string.
int compile_setup_module(const std::string& argument_string, std::string& code, bool cache_only = false);

// hiprtc's version (part of every cache key), asked once per process: the first call into hiprtc initialises the HIP runtime behind it,
// and two threads doing that at the same moment (the two builds of build_frame_path) left one of them without a device on the GPU box
// ("hipSetDevice: no ROCm-capable device is detected" in the first program a process created)
void rtc_version(int& major, int& minor) {
    static int v[2] = {0, 0};
    static std::once_flag once;
    std::call_once(once, [] { hiprtcVersion(&v[0], &v[1]); });
    major = v[0];
    minor = v[1];
}

// the fallback of both modules' builds: hiprtc itself (no pass over the code; a source error shows up here with its diagnostics)
int build_through_hiprtc(const std::string& source, const char* name, const std::vector<std::string>& options, std::string& out) {
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, source.c_str(), name, 0, nullptr, nullptr) != HIPRTC_SUCCESS)
        return fail(GR_ERROR_COMPILE, "hiprtcCreateProgram failed");
    std::vector<const char*> copts;
    for (auto& o : options) copts.push_back(o.c_str());
    hiprtcResult r = hiprtcCompileProgram(prog, (int)copts.size(), copts.data());
    if (r != HIPRTC_SUCCESS) {
        size_t n = 0;
        hiprtcGetProgramLogSize(prog, &n);
        std::string log(n, '\0');
        if (n) hiprtcGetProgramLog(prog, &log[0]);
        hiprtcDestroyProgram(&prog);
        return fail(GR_ERROR_COMPILE, std::string("hiprtc: ") + hiprtcGetErrorString(r) + "\n" + log);
    }
    size_t n = 0;
    hiprtcGetCodeSize(prog, &n);
    out.assign(n, '\0');
    hiprtcGetCode(prog, &out[0]);
    hiprtcDestroyProgram(&prog);
    return GR_OK;
}

// A program's ray kernels are two code objects (kernels/program.hip): PART_FRAME - what a fused frame launches - and PART_REST, the
// reference-shaped sequence and ray compaction.  cache_only: an empty `code` and GR_OK when the part is not in the cache (the caller
// builds it later, or on another thread).
enum BuildPart { PART_FRAME = 0, PART_REST = 1 };
int compile_code_object(const std::string& argument_string, std::string& code, std::string* key_out = nullptr, BuildPart part = PART_FRAME,
                        bool cache_only = false) {
    // the kernel source: the parts under csrc/kernels/ in this order, as one translation unit (GR_KERNEL_SOURCE: one file instead)
    static const char* const KERNEL_PARTS[] = {"program.hip",       // structs of the boundary, build switches
                                               "probes.inc",        // measurement hooks (all off by default)
                                               "metric.hip",        // hosts of the generated expressions
                                               "setup.hip",         // tetrads, ray set-up
                                               "integrator.hip",    // the Verlet loop, one and two rays per lane
                                               "trace.hip",         // render-data, the reference-shaped and the fused kernels, prepass, tile order, adaptive sampling
                                               "shading.hip"};      // texture sampling, gr_render
    // (camera.hip and geodesic_camera.hip - what runs once per frame on one lane - are the set-up module: compile_setup_module)
    std::string source;
    if (const char* env = getenv("GR_KERNEL_SOURCE")) {
        if (!read_file(env, source)) return fail(GR_ERROR_COMPILE, std::string("cannot read kernel source ") + env);
    } else {
        for (const char* part : KERNEL_PARTS) {
            std::string text;
            const std::string path = library_dir() + "/csrc/kernels/" + part;
            if (!read_file(path, text)) return fail(GR_ERROR_COMPILE, "cannot read kernel source " + path);
            source += text;
            if (!text.empty() && text.back() != '\n') source += '\n';
        }
    }

    std::vector<std::string> opts = {
        "--offload-arch=gfx950", "-O3", "-std=c++17",
        // the reference builds with -cl-unsafe-math-optimizations (metric_manager.hpp:70): reassociation,
        // reciprocal division, contraction - but NaN/Inf stay meaningful (IS_DEGENERATE, cl.cl:68)
        "-ffp-contract=fast", "-fno-math-errno", "-freciprocal-math", "-fassociative-math",
        "-fno-signed-zeros", "-fno-trapping-math",
        // OpenCL's default 2.5-ulp fp32 divide/sqrt (the reference does not pass -cl-fp32-correctly-rounded-divide-sqrt):
        // v_rcp_f32 / v_sqrt_f32 instead of the ~10-instruction correctly rounded sequences
        "-fno-hip-fp32-correctly-rounded-divide-sqrt",
        // ... and its "unsafe math": approximate-function semantics for divide/sqrt/libm (a/b = a * v_rcp_f32(b) with no
        // denormal rescaling) and flushed fp32 denormals.  Measured -18 % on the Kerr Verlet kernel, parity unchanged.
        "-fapprox-func", "-fgpu-flush-denormals-to-zero",
        // no SLP vectorisation: packed fp32 (v_pk_mul/fma_f32) is at best ~1.2x the plain rate on gfx950 and needs operand
        // pairs in adjacent registers - the straight-line metric code paid ~55 v_mov per Verlet step for it.
        // Measured on the Kerr kernel: 102 -> 80 VGPRs, 12.7 -> 10.1 ms.
        "-fno-slp-vectorize"};
    for (auto& tok : split_arguments(argument_string)) {
        if (tok.rfind("-D", 0) == 0) opts.push_back(tok);
        else if (tok == "-cl-fp32-correctly-rounded-divide-sqrt") {
            // OpenCL's own switch for IEEE divide and square root (the reference does not pass it, metric_manager.hpp:70; a caller who
            // appends it to the argument string gets what it means): the ray kernels without v_rcp_f32 / v_sqrt_f32 arithmetic.
            // Measured on the frame that shows the difference most (near-extreme double Kerr, tests/golden/soak/): masked pixel RMSE
            // 1.30e-4 -> 6.4e-5, pixels off 93 -> 9 of 9 216 - the reference's own distance from itself under a one-ulp change of the
            // camera position; the Verlet loop pays ~10 instructions per division.
            for (const char* drop : {"-freciprocal-math", "-fapprox-func", "-fno-hip-fp32-correctly-rounded-divide-sqrt"})
                opts.erase(std::remove(opts.begin(), opts.end(), std::string(drop)), opts.end());
            opts.push_back("-fhip-fp32-correctly-rounded-divide-sqrt");
        }
        else if (tok.rfind("-cl-", 0) == 0 || tok == "-I" || tok == "./") continue;   // OpenCL-only prefix flags
        else return fail(GR_ERROR_INVALID_ARGUMENT, "unsupported token in argument string: " + tok);
    }
    if (pair_kernel_applies(opts)) opts.push_back("-DGR_TWO_RAYS_PER_LANE");
    if (accelerations_without_trig(opts)) opts.push_back("-DGR_ACCEL_WITHOUT_TRIG");
    if (const char* extra = getenv("GR_EXTRA_FLAGS"))
        for (auto& tok : split_arguments(extra)) opts.push_back(tok);
    opts.push_back(part == PART_FRAME ? "-DGR_BUILD_FRAME_PATH" : "-DGR_BUILD_REST");

    int rtc_major = 0, rtc_minor = 0;
    rtc_version(rtc_major, rtc_minor);
    uint64_t h = fnv1a(source);
    for (auto& o : opts) h = fnv1a(o + "\n", h);
    h = fnv1a("hiprtc " + std::to_string(rtc_major) + "." + std::to_string(rtc_minor), h);
    {
        const char* tuning = getenv("GR_OCCUPANCY_TUNING");   // changes what is built for the same options (see below)
        if (tuning && tuning[0] == '0') h = fnv1a("no occupancy tuning", h);
    }
    // Pass over the compiled code (codeobject.hpp): no more than `run_limit` vector instructions in a row without a scalar one.
    // GR_VECTOR_RUN_LIMIT=0: no pass (the build still goes through the code-object manager, see below).
    int run_limit = GR_DEFAULT_VECTOR_RUN_LIMIT;
    if (const char* e = getenv("GR_VECTOR_RUN_LIMIT")) run_limit = atoi(e);
    if (run_limit > 0) h = fnv1a("vector runs <= " + std::to_string(run_limit) + " in the integrator kernels, list of round 5", h);   // (the list below is part of what is built)
    char name[64];
    snprintf(name, sizeof(name), "%016llx.hsaco", (unsigned long long)h);
    if (key_out) key_out->assign(name, 16);

    std::string cache_dir;
    if (const char* env = getenv("GR_CACHE_DIR")) cache_dir = env;
    else cache_dir = library_dir() + "/_cache";
    std::string cache_path = cache_dir + "/" + name;
    if (read_file(cache_path, code) && !code.empty()) return GR_OK;
    code.clear();
    if (cache_only) return GR_OK;

    // one build of the kernel source with `options`: through the assembly pass when it is on and the code-object manager is
    // there, else through hiprtc (a source error shows up there with its diagnostics)
    bool pass_not_applied = false;   // some build of this call went out as compiled although the pass is on
    auto build = [&](const std::vector<std::string>& options, std::string& out) -> int {
        {   // (GR_VECTOR_RUN_LIMIT=0 goes the same way without the pass: which code-object manager hiprtc would find depends on
            // what else the process has loaded, and the copy bundled with PyTorch aborts on these kernels)
            std::string assembly, log;
            if (gr::compile_to_assembly(source, options, assembly, log)) {
                if (run_limit <= 0 && gr::assemble_code_object(assembly, out, log)) return GR_OK;
                // the kernels that hold a Verlet loop; the others (set-up, shading, tile order ...) are left as compiled
                static const std::vector<std::string> integrators = {"gr_trace_fused", "gr_trace_fused_lattice", "gr_trace_pair", "gr_trace_compact", "gr_prepass_fused",
                                                                     "gr_do_generic_rays", "gr_do_generic_rays_scheduled", "gr_trace_fused_parking"};
                std::string patched = assembly;
                const gr::vector_run_stats st = gr::break_vector_runs(patched, run_limit, integrators);
                if (gr::assemble_code_object(patched, out, log)) {
                    if (getenv("GR_VERBOSE_BUILD"))
                        fprintf(stderr, "[gr] vector runs: %d longer than %d (longest %d) cut by %d s_nop, longest now %d\n", st.runs_broken,
                                run_limit, st.longest_before, st.inserted, st.longest_after);
                    return GR_OK;
                }
                // The compiler sized its branches for the code it emitted; in a very large function the added instructions can push
                // one past the 16-bit branch offset ("branch size exceeds simm16").  Then the code as compiled.
                if (getenv("GR_VERBOSE_BUILD")) fprintf(stderr, "[gr] assembly pass not applied (%s)\n", log.c_str());
                pass_not_applied = true;
                if (gr::assemble_code_object(assembly, out, log)) return GR_OK;
            }
            if (run_limit > 0) pass_not_applied = true;
            if (getenv("GR_VERBOSE_BUILD")) fprintf(stderr, "[gr] building through hiprtc (%s)\n", log.c_str());
        }
        return build_through_hiprtc(source, "geodesic_kernels_all_parts.hip", options, out);
    };
    // The occupancy rule below costs up to three more compiler runs.  Its outcome depends on the program's SHAPE - kernel source,
    // options, the metric's expressions - far more than on the literals a substituted program carries, and a slider move changes only
    // those: the decision is remembered per shape (the options with every float literal blanked) next to the code objects, and a program
    // of a known shape is built held to the remembered wave count straight away - one compiler run, the swap of the substituted program
    // after a parameter change ~20 s -> ~8 s - as long as that build still meets the rule's own conditions.
    bool tuned_by_caller = false;
    for (auto& o : opts) tuned_by_caller |= o.rfind("-DGR_FUSED_WAVES", 0) == 0 || o.rfind("-DGR_TRACE_WAVES", 0) == 0;
    const char* tuning = getenv("GR_OCCUPANCY_TUNING");
    // (the rule is about gr_trace_fused: the other part is built as the compiler allocates it)
    const bool rule_applies = part == PART_FRAME && !tuned_by_caller && !(tuning && tuning[0] == '0');
    std::string shape_path;
    {
        uint64_t sh = fnv1a(source);
        for (auto& o : opts) {
            // (round 6: the device's own rendering of the accelerations - GR_DEVICE_ACCEL*, GR_DEVICE_TEMPORARIES - shares other
            // sub-expressions from one parameter set to the next, so its text has another length and another set of temporaries; with it
            // in the key no two parameter sets of round 5 ever had the same shape and every slider move paid the rule's three builds)
            if (o.rfind("-DGR_DEVICE_", 0) == 0) continue;
            std::string blank;
            for (size_t i = 0; i < o.size();) {
                const bool starts_number = isdigit((unsigned char)o[i]) && (i == 0 || !(isalnum((unsigned char)o[i - 1]) || o[i - 1] == '_'));
                if (!starts_number) { blank += o[i++]; continue; }
                size_t j = i;
                while (j < o.size() && (isdigit((unsigned char)o[j]) || o[j] == '.' || ((o[j] == 'e' || o[j] == 'E') && j + 1 < o.size() && (isdigit((unsigned char)o[j + 1]) || o[j + 1] == '-' || o[j + 1] == '+')) ||
                                        ((o[j] == '-' || o[j] == '+') && j > i && (o[j - 1] == 'e' || o[j - 1] == 'E')))) j++;
                const bool is_float = j < o.size() && o[j] == 'f' && o.substr(i, j - i).find_first_of(".e") != std::string::npos;
                if (is_float) { blank += '#'; i = j + 1; } else { blank.append(o, i, j - i); i = j; }
            }
            // (the generator orders the operands of sums and products by a hash that takes the literals in, and numbers its temporaries as
            // it meets them: two parameter sets give the same expressions in another order.  What is left after blanking the literals
            // is therefore taken as a bag of characters, every digit the same - a hint's key may collide, the conditions above decide.)
            for (char& ch : blank) if (isdigit((unsigned char)ch)) ch = '9';
            const size_t eq = blank.find('=');
            if (eq != std::string::npos) std::sort(blank.begin() + (long)eq + 1, blank.end());
            sh = fnv1a(blank + "\n", sh);
        }
        sh = fnv1a("shape, hiprtc " + std::to_string(rtc_major) + "." + std::to_string(rtc_minor) + ", runs " + std::to_string(run_limit), sh);
        char shape_name[64];
        snprintf(shape_name, sizeof(shape_name), "%016llx.occupancy", (unsigned long long)sh);
        shape_path = cache_dir + "/" + shape_name;
    }
    bool settled_from_memory = false;
    if (rule_applies) {
        std::string note;
        int waves = 0, free_vgprs = 0, free_scratch = 0;
        if (read_file(shape_path, note) && sscanf(note.c_str(), "waves=%d free_vgprs=%d free_scratch=%d", &waves, &free_vgprs, &free_scratch) == 3 && waves >= 1 && waves <= 8) {
            std::vector<std::string> capped = opts;
            capped.push_back("-DGR_FUSED_WAVES=" + std::to_string(waves));
            std::string code2;
            int v2 = 0, s2 = 0, g2 = 0;
            if (build(capped, code2) == GR_OK && kernel_resources(code2, "gr_trace_fused", v2, s2, &g2) && s2 <= free_scratch + 96 &&
                resident_waves_per_simd(v2, g2) >= waves) {
                code.swap(code2);
                settled_from_memory = true;
                if (getenv("GR_VERBOSE_BUILD"))
                    fprintf(stderr, "[gr] gr_trace_fused: held to %d waves as remembered for programs of this shape: %d VGPRs / %d SGPRs / %d B scratch (kept, one compiler run)\n", waves, v2, g2, s2);
            }
        }
    }
    if (!settled_from_memory) {
        int rc = build(opts, code);
        if (rc != GR_OK) return rc;
    }

    // Occupancy of the fused trace kernel.  Left alone, the register allocator takes what the kernel could use at its widest
    // point (Kerr, substituted: 97 VGPRs, 5 waves per SIMD), part of which is cold inside the Verlet loop (set-up and epilogue
    // values).  Measured on MI355X, 4K Kerr, three frames in flight / one launch on its own, with the loop that still kept a
    // finished ray's state in twelve registers of its own (108 VGPRs free): free build 1 400 Mrays/s / 6.7 ms; held to 96 VGPRs (5
    // waves, nothing spilled) 1 492 / 6.5; to 80 (6 waves, 60 bytes per lane spilled, none of it inside the loop's attempts)
    // 1 535 / 6.2; to 72 (7 waves, 84 bytes) 1 530 / 6.2.  With today's loop: 6 waves spill 24 bytes; 7 and 8 waves measure the same.
    // Rule: rebuild with the register budget of five sixths of what the free build took, rounded down to an occupancy step,
    // and keep that build unless it spills more than 96 bytes per lane (the same source compiles to a spill that differs by 20 B
    // from one hiprtc run to the next, and a limit next to the expected number flipped the decision with it) - or unless its waves
    // would not be resident anyway (round 4): the kernel's ~94 scalar registers admit 6 waves per SIMD, so the "7 waves" build of
    // rounds 2 and 3 (72 VGPRs, 48 B spilled) ran 6 like the 80-register build that spills 16 B; that one is 3 % faster one frame at
    // a time (5.46 against 5.65 ms, 4K Kerr) and the same with frames in flight.
    if (rule_applies && !settled_from_memory) {
        int vgprs = 0, scratch = 0;
        if (kernel_resources(code, "gr_trace_fused", vgprs, scratch) && vgprs > 64) {
            auto waves_of = [](int regs) { int w = 512 / (((regs + 7) / 8) * 8); return w > 8 ? 8 : w; };
            int target_waves = waves_of(vgprs * 5 / 6);
            while (target_waves > 1 && (512 / target_waves) / 8 * 8 > vgprs * 5 / 6) target_waves++;   // budget of w waves <= 5/6 of the free build
            if (target_waves > 8) target_waves = 8;
            if (target_waves <= waves_of(vgprs) && getenv("GR_VERBOSE_BUILD"))
                fprintf(stderr, "[gr] gr_trace_fused: free build %d VGPRs / %d B scratch, left alone\n", vgprs, scratch);
            // from the rule's target down to one wave more than the free build holds: the first budget that does not spill too much
            // (double Kerr: 172 VGPRs = 2 waves; held to 4 waves it spills 148 B, held to 3 - 168 VGPRs - nothing)
            for (int waves = target_waves; waves > waves_of(vgprs); waves--) {
                std::vector<std::string> capped = opts;
                capped.push_back("-DGR_FUSED_WAVES=" + std::to_string(waves));
                std::string code2;
                int v2 = 0, s2 = 0, g2 = 0;
                const bool built = build(capped, code2) == GR_OK && kernel_resources(code2, "gr_trace_fused", v2, s2, &g2);
                const bool resident = built && resident_waves_per_simd(v2, g2) >= waves;
                const bool keep = built && s2 <= scratch + 96 && resident;
                if (getenv("GR_VERBOSE_BUILD"))
                    fprintf(stderr, "[gr] gr_trace_fused: free build %d VGPRs / %d B scratch; held to %d waves: %d VGPRs / %d SGPRs / %d B scratch%s\n", vgprs,
                            scratch, waves, v2, g2, s2, keep ? " (kept)" : resident ? " (dropped)" : " (dropped: its scalar registers admit fewer waves)");
                if (keep) {
                    code.swap(code2);
                    if (!pass_not_applied) {   // remembered for the next program of this shape
                        mkdir(cache_dir.c_str(), 0755);
                        std::ofstream f(shape_path + ".tmp" + std::to_string((long)getpid()));
                        f << "waves=" << waves << " free_vgprs=" << vgprs << " free_scratch=" << scratch << "\n";
                        f.close();
                        if (rename((shape_path + ".tmp" + std::to_string((long)getpid())).c_str(), shape_path.c_str()) != 0) remove((shape_path + ".tmp" + std::to_string((long)getpid())).c_str());
                    }
                    break;
                }
            }
        }
    }

    if (pass_not_applied) {
        // The cache key says "vector runs <= N"; this code object does not have them cut (a branch pushed past its 16-bit offset, or
        // the code-object manager could not be loaded and hiprtc built it).  It is used but not cached under that key: a later
        // process in which the pass can run builds it properly instead of being served this one for good.
        static std::atomic<bool> warned{false};
        if (!warned.exchange(true))
            fprintf(stderr, "[gr] warning: the pass over the compiled code (vector runs <= %d) could not be applied to a program; it runs as "
                            "compiled (~20 %% slower Verlet loop) and is not cached.  GR_VERBOSE_BUILD=1 says why.\n", run_limit);
        return GR_OK;
    }
    mkdir(cache_dir.c_str(), 0755);
    // unique per writer: a background build (gr_program_create_async) and a foreground build of the same key may run in one
    // process, and several processes share the cache directory; rename() publishes a complete file atomically
    static std::atomic<unsigned long> writer{0};
    std::string tmp = cache_path + ".tmp" + std::to_string((long)getpid()) + "." + std::to_string(writer.fetch_add(1));
    {
        std::ofstream f(tmp, std::ios::binary);
        f.write(code.data(), (std::streamsize)code.size());
        if (!f) { f.close(); remove(tmp.c_str()); return GR_OK; }   // cache write failed (disk full, read-only): the build still succeeded
    }
    if (rename(tmp.c_str(), cache_path.c_str()) != 0) remove(tmp.c_str());
    return GR_OK;
}

// The set-up module of a program: the kernels that run once per frame on one lane - camera coordinates, tetrad, the camera's own
// geodesic - from program + probes + metric + setup + camera + geodesic_camera, built with IEEE arithmetic (kernels/camera.hip says
// why).  Same macro string, a cache file of its own; no pass over the code, no occupancy rule: nothing here is issue-bound.
int compile_setup_module(const std::string& argument_string, std::string& code, bool cache_only) {
    static const char* const PARTS[] = {"program.hip", "probes.inc", "metric.hip", "setup.hip", "camera.hip", "geodesic_camera.hip"};
    std::string source;
    // GR_SETUP_KERNEL_SOURCE: one file instead of the parts, as GR_KERNEL_SOURCE is for the ray kernels' module.  (GR_KERNEL_SOURCE
    // alone replaces the ray kernels only - the tools that use it patch the trace kernel - and this module is then built from the
    // library's own parts: said once on stderr, so that nobody takes a patched metric.hip to reach the camera kernels.)
    if (const char* env = getenv("GR_SETUP_KERNEL_SOURCE")) {
        if (!read_file(env, source)) return fail(GR_ERROR_COMPILE, std::string("cannot read set-up kernel source ") + env);
    } else {
        if (getenv("GR_KERNEL_SOURCE")) {
            static std::atomic<bool> said{false};
            if (!said.exchange(true))
                fprintf(stderr, "[gr] note: GR_KERNEL_SOURCE replaces the ray kernels' source only; the set-up module (camera, tetrad, geodesic camera) is built "
                                "from the library's csrc/kernels - GR_SETUP_KERNEL_SOURCE replaces that\n");
        }
        for (const char* part : PARTS) {
            std::string text;
            const std::string path = library_dir() + "/csrc/kernels/" + part;
            if (!read_file(path, text)) return fail(GR_ERROR_COMPILE, "cannot read kernel source " + path);
            source += text;
            if (!text.empty() && text.back() != '\n') source += '\n';
        }
    }
    std::vector<std::string> opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-math-errno", "-fno-slp-vectorize",
                                     "-fhip-fp32-correctly-rounded-divide-sqrt", "-DGR_SETUP_MODULE", "-DGR_LIBM_TRIG", "-DGR_LIBM_TANH"};
    for (auto& tok : split_arguments(argument_string)) {
        if (tok.rfind("-D", 0) == 0) opts.push_back(tok);
        else if (tok.rfind("-cl-", 0) == 0 || tok == "-I" || tok == "./") continue;
        else return fail(GR_ERROR_INVALID_ARGUMENT, "unsupported token in argument string: " + tok);
    }
    if (const char* extra = getenv("GR_SETUP_EXTRA_FLAGS"))
        for (auto& tok : split_arguments(extra)) opts.push_back(tok);
    int rtc_major = 0, rtc_minor = 0;
    rtc_version(rtc_major, rtc_minor);
    uint64_t h = fnv1a(source);
    for (auto& o : opts) h = fnv1a(o + "\n", h);
    h = fnv1a("set-up module, hiprtc " + std::to_string(rtc_major) + "." + std::to_string(rtc_minor), h);
    char name[64];
    snprintf(name, sizeof(name), "%016llx.setup.hsaco", (unsigned long long)h);
    std::string cache_dir;
    if (const char* env = getenv("GR_CACHE_DIR")) cache_dir = env;
    else cache_dir = library_dir() + "/_cache";
    const std::string cache_path = cache_dir + "/" + name;
    if (read_file(cache_path, code) && !code.empty()) return GR_OK;
    code.clear();
    if (cache_only) return GR_OK;
    std::string assembly, log;
    if (!gr::compile_to_assembly(source, opts, assembly, log) || !gr::assemble_code_object(assembly, code, log)) {
        // the code-object manager could not be loaded (or is the copy bundled with another library): hiprtc, as for the ray kernels'
        // module - this module needs no pass over its code, so the result is the same program and is cached like any other
        if (getenv("GR_VERBOSE_BUILD")) fprintf(stderr, "[gr] set-up module: building through hiprtc (%s)\n", log.c_str());
        const int rc = build_through_hiprtc(source, "geodesic_setup_kernels.hip", opts, code);
        if (rc != GR_OK) return rc;
    }
    mkdir(cache_dir.c_str(), 0755);
    static std::atomic<unsigned long> writer{0};
    const std::string tmp = cache_path + ".tmp" + std::to_string((long)getpid()) + "." + std::to_string(writer.fetch_add(1));
    {
        std::ofstream f(tmp, std::ios::binary);
        f.write(code.data(), (std::streamsize)code.size());
        if (!f) { f.close(); remove(tmp.c_str()); return GR_OK; }
    }
    if (rename(tmp.c_str(), cache_path.c_str()) != 0) remove(tmp.c_str());
    return GR_OK;
}

}  // namespace

struct gr_metric {
    gr::MetricConfig cfg;
    gr::MetricFunctions functions;
    gr::DynamicVars vars;
    gr::MetricDescriptor desc;
    std::shared_ptr<void> keepalive;   // script interpreter state owning the closures
    bool settings_only = false;        // gr_metric_from_info: no symbolic content, only what the frame driver reads
    gr_metric_info stored_info = {};
};

struct gr_program {
    int device = 0;
    hipModule_t module = nullptr;
    hipModule_t setup_module = nullptr;   // camera / tetrad / geodesic-camera kernels, IEEE arithmetic (compile_setup_module)
    hipFunction_t fn[K_COUNT] = {};
    void* huge_count = nullptr;   // device int = INT_MAX: "no device-side count" for range launches
    // tile tickets of the persistent fused trace: one counter per launch, taken round robin from a small ring so that
    // launches in flight on different streams never share one
    static const int TICKET_RING = 64;
    unsigned int* tickets = nullptr;
    std::atomic<unsigned> next_ticket{0};
    int compute_units = 256;
    bool tile_shading = false;   // built with -DGR_TILE_SHADING: gr_trace_fused can shade the inner pixels of its tiles
    bool cell_rows = false;      // built with -DGR_CELL_BLOCK=0: a cell wave of an in-launch prepass is 64 cells of a row, not 8 x 8 cells
    int resident_groups_per_cu[K_COUNT] = {};   // of the trace kernels at the launch's workgroup size: 0 = not asked yet
    // The kernels a fused frame does not launch (PART_REST) are a code object of their own, loaded when first asked for: from the
    // cache when gr_program_create found it there, else from a build that started on a worker thread when the program was created.
    // Whoever asks first (launch of such a kernel, gr_program_kernel_info, gr_program_complete) waits for that build.
    struct rest_build {
        std::mutex mu;
        std::condition_variable cv;
        bool done = false;
        int rc = GR_OK;
        std::string code, error;
    };
    std::shared_ptr<rest_build> rest;     // shared with the worker, which may outlive the program
    std::thread rest_worker;
    std::mutex rest_mu;                   // serialises the load
    bool rest_loaded = false;
    int rest_rc = GR_OK;
    std::string rest_error;
    hipModule_t module_rest = nullptr;
    bool in_rest[K_COUNT] = {};           // kernels expected in the other code object
    std::string arguments;
    std::string key;   // what the code object was built from: kernel source, every compile option, hiprtc version (16 hex digits)
    // identity for caches keyed by program (frame.cpp prefetch slots): an address can be reused by a later program, this cannot
    unsigned long long serial = 0;
    ~gr_program() {
        if (module || tickets || huge_count) (void)hipSetDevice(device);
        if (tickets) (void)hipFree(tickets);
        if (huge_count) (void)hipFree(huge_count);
        if (module) (void)hipModuleUnload(module);
        if (module_rest) (void)hipModuleUnload(module_rest);
        if (setup_module) (void)hipModuleUnload(setup_module);
        // a build still inside the compiler cannot be interrupted; it holds its own state (rest_build) and is left to finish
        if (rest_worker.joinable()) {
            bool done;
            { std::lock_guard<std::mutex> lock(rest->mu); done = rest->done; }
            if (done) rest_worker.join(); else rest_worker.detach();
        }
    }
};

extern "C" {

const char* gr_last_error(void) { return g_error.c_str(); }

void gr_features_default(gr_features* f) {
    if (!f) return;
    // main.cpp:1123-1158
    f->adaptive_sampling_threshold = 64.f;
    f->field_of_view = 90.f;
    f->max_acceleration_change = 0.01f;
    f->max_precision_radius = 10.f;
    f->min_step = 0.000001f;
    f->ray_skip = 4.f;
    f->universe_size = 20.f;
    f->adaptive_sampling = 1;
    f->redshift = 0;
    f->reparameterisation = 0;
    f->use_old_redshift = 0;
    f->use_triangle_rendering = 0;
}

static gr::FeatureConfig to_feature_config(const gr_features* f) {
    gr_features d;
    gr_features_default(&d);
    if (f) d = *f;
    gr::FeatureConfig c;
    c.set("adaptive_sampling_threshold", d.adaptive_sampling_threshold);
    c.set("field_of_view", d.field_of_view);
    c.set("max_acceleration_change", d.max_acceleration_change);
    c.set("max_precision_radius", d.max_precision_radius);
    c.set("min_step", d.min_step);
    c.set("ray_skip", d.ray_skip);
    c.set("universe_size", d.universe_size);
    c.set("adaptive_sampling", d.adaptive_sampling != 0);
    c.set("redshift", d.redshift != 0);
    c.set("reparameterisation", d.reparameterisation != 0);
    c.set("use_old_redshift", d.use_old_redshift != 0);
    c.set("use_triangle_rendering", d.use_triangle_rendering != 0);
    return c;
}

int gr_metric_builtin(const char* name, gr_metric** out) {
    if (!name || !out) return fail(GR_ERROR_INVALID_ARGUMENT, "null argument");
    GR_TRY_BEGIN
    auto m = std::make_unique<gr_metric>();
    if (!gr::builtin_metric(name, m->cfg, m->functions, m->vars))
        return fail(GR_ERROR_INVALID_ARGUMENT, std::string("unknown built-in metric ") + name);
    m->desc.load(m->functions, m->cfg);
    *out = m.release();
    return GR_OK;
    GR_TRY_END
}

int gr_metric_load_script(const char* scripts_dir, const char* name, gr_metric** out) {
    if (!scripts_dir || !name || !out) return fail(GR_ERROR_INVALID_ARGUMENT, "null argument");
    GR_TRY_BEGIN
    auto m = std::make_unique<gr_metric>();
    m->keepalive = gr::load_metric_from_scripts(scripts_dir, name, m->cfg, m->functions, m->vars);
    m->desc.load(m->functions, m->cfg);
    *out = m.release();
    return GR_OK;
    GR_TRY_END
}

int gr_metric_from_info(const gr_metric_info* info, const char* const* var_names, const float* var_defaults, gr_metric** out) {
    if (!info || !out || info->num_dynamic_vars < 0 || (info->num_dynamic_vars > 0 && !var_defaults))
        return fail(GR_ERROR_INVALID_ARGUMENT, "gr_metric_from_info: null argument");
    GR_TRY_BEGIN
    auto m = std::make_unique<gr_metric>();
    m->settings_only = true;
    m->stored_info = *info;
    for (int i = 0; i < info->num_dynamic_vars; i++) {
        m->vars.names.push_back(var_names && var_names[i] ? var_names[i] : "v" + std::to_string(i));
        m->vars.defaults.push_back(var_defaults[i]);
    }
    *out = m.release();
    return GR_OK;
    GR_TRY_END
}

void gr_metric_destroy(gr_metric* m) { delete m; }

int gr_metric_get_info(const gr_metric* m, gr_metric_info* out) {
    if (!m || !out) return fail(GR_ERROR_INVALID_ARGUMENT, "null argument");
    if (m->settings_only) { *out = m->stored_info; return GR_OK; }
    out->is_big = m->desc.is_big;
    out->is_constant_theta = m->desc.is_spherical && m->cfg.system == gr::CoordinateSystem::X_Y_THETA_PHI;
    out->use_prepass = m->cfg.use_prepass;
    out->adaptive_precision = m->cfg.adaptive_precision;
    out->max_acceleration_change = m->cfg.max_acceleration_change;
    out->num_dynamic_vars = (int)m->vars.names.size();
    out->accel_ops = m->desc.accel_ops.ops;
    out->accel_transcendentals = m->desc.accel_ops.transcendental;
    out->coord_ops = m->desc.coord_ops.ops;
    return GR_OK;
}

const char* gr_metric_dynamic_var_name(const gr_metric* m, int i) {
    if (!m || i < 0 || i >= (int)m->vars.names.size()) return nullptr;
    return m->vars.names[i].c_str();
}

float gr_metric_dynamic_var_default(const gr_metric* m, int i) {
    if (!m || i < 0 || i >= (int)m->vars.defaults.size()) return 0.f;
    return m->vars.defaults[i];
}

int gr_metric_argument_string(const gr_metric* m, const gr_features* features, int is_static, const float* cfg_values,
                              int num_cfg_values, char* buffer, size_t capacity, size_t* needed) {
    if (!m) return fail(GR_ERROR_INVALID_ARGUMENT, "null metric");
    if (m->settings_only) return fail(GR_ERROR_INVALID_ARGUMENT, "a metric made by gr_metric_from_info has no expressions to generate from");
    GR_TRY_BEGIN
    gr::FeatureConfig fc = to_feature_config(features);
    std::string s;
    if (is_static) {
        std::vector<float> vals(cfg_values, cfg_values + (cfg_values ? num_cfg_values : 0));
        gr::MetricImpl concrete = m->desc.concrete(m->vars.substitution(vals));
        s = gr::build_argument_string(m->desc, concrete, m->cfg, m->vars, true, fc);
    } else {
        s = gr::build_argument_string(m->desc, m->desc.raw, m->cfg, m->vars, false, fc);
    }
    if (needed) *needed = s.size() + 1;
    if (!buffer && capacity == 0 && needed) return GR_OK;   // size query
    if (!buffer || capacity < s.size() + 1) return fail(GR_ERROR_BUFFER_TOO_SMALL, "buffer too small");
    memcpy(buffer, s.c_str(), s.size() + 1);
    return GR_OK;
    GR_TRY_END
}

int gr_metric_evaluate_count(int what) {
    switch (what) {
        case GR_EVAL_METRIC_TENSOR: return 16;
        case GR_EVAL_METRIC_DERIVATIVES: return 64;
        case GR_EVAL_ACCELERATION: case GR_EVAL_TO_POLAR: case GR_EVAL_FROM_POLAR: return 4;
        case GR_EVAL_ORIGIN_DISTANCE: return 1;
        default: return 0;
    }
}

// The host-side evaluator of the generated expressions: sym::eval over the metric's DAGs (the same graphs build_argument_string prints),
// position / velocity / $cfg bound by name.  One point per call; a graph node is visited once per call (memoised).
int gr_metric_evaluate(const gr_metric* m, int what, const double position[4], const double velocity[4], const float* cfg_values,
                       int num_cfg_values, double* out, int out_count) {
    if (!m || !position || !out) return fail(GR_ERROR_INVALID_ARGUMENT, "gr_metric_evaluate: null argument");
    if (m->settings_only) return fail(GR_ERROR_INVALID_ARGUMENT, "a metric made by gr_metric_from_info has no expressions to evaluate");
    const int count = gr_metric_evaluate_count(what);
    if (count == 0) return fail(GR_ERROR_INVALID_ARGUMENT, "gr_metric_evaluate: unknown kind");
    if (out_count < count) return fail(GR_ERROR_BUFFER_TOO_SMALL, "gr_metric_evaluate: out_count too small");
    if (what == GR_EVAL_ACCELERATION && !velocity) return fail(GR_ERROR_INVALID_ARGUMENT, "gr_metric_evaluate: the acceleration needs a velocity");
    if (cfg_values && num_cfg_values != (int)m->vars.names.size())
        return fail(GR_ERROR_INVALID_ARGUMENT, "gr_metric_evaluate: num_cfg_values is not the metric's number of $cfg parameters");
    GR_TRY_BEGIN
    std::map<std::string, double> env;
    static const char* const pos_names[4] = {"v1", "v2", "v3", "v4"};
    static const char* const vel_names[4] = {"iv1", "iv2", "iv3", "iv4"};
    for (int i = 0; i < 4; i++) {
        env[pos_names[i]] = position[i];
        env[vel_names[i]] = velocity ? velocity[i] : 0.0;
    }
    for (size_t i = 0; i < m->vars.names.size(); i++)
        env["cfg->" + m->vars.names[i]] = cfg_values ? (double)cfg_values[i] : (double)m->vars.defaults[i];
    env["always_lightlike"] = 0.0;
    const gr::MetricImpl& impl = m->desc.raw;
    auto eval_all = [&](const std::vector<sym::E>& v, double* dst) { for (size_t i = 0; i < v.size(); i++) dst[i] = sym::eval(v[i], env); };
    switch (what) {
        case GR_EVAL_METRIC_TENSOR:
            for (int i = 0; i < 16; i++) out[i] = 0.0;
            if (impl.real_eq.size() == 4) { for (int i = 0; i < 4; i++) out[i * 4 + i] = sym::eval(impl.real_eq[(size_t)i], env); }
            else eval_all(impl.real_eq, out);
            break;
        case GR_EVAL_METRIC_DERIVATIVES:
            for (int i = 0; i < 64; i++) out[i] = 0.0;
            if (impl.derivatives.size() == 16) { for (int k = 0; k < 4; k++) for (int i = 0; i < 4; i++) out[k * 16 + i * 4 + i] = sym::eval(impl.derivatives[(size_t)(k * 4 + i)], env); }
            else eval_all(impl.derivatives, out);
            break;
        case GR_EVAL_ACCELERATION: eval_all(impl.accel, out); break;
        case GR_EVAL_TO_POLAR: eval_all(impl.to_polar, out); break;
        case GR_EVAL_FROM_POLAR: eval_all(impl.from_polar, out); break;
        case GR_EVAL_ORIGIN_DISTANCE: {   // DISTANCE_FUNC is a function of the POLAR point (kernels/metric.hip distance_to_object): composed with TO_COORDn here
            double polar[4];
            eval_all(impl.to_polar, polar);
            for (int i = 0; i < 4; i++) env[pos_names[i]] = polar[i];
            out[0] = sym::eval(impl.distance_function, env);
            break;
        }
    }
    return GR_OK;
    GR_TRY_END
}

int gr_metric_substituted_op_counts(const gr_metric* m, const float* cfg_values, int num_cfg_values, int* accel_ops,
                                    int* accel_transcendentals, int* coord_ops) {
    if (!m) return fail(GR_ERROR_INVALID_ARGUMENT, "null metric");
    if (m->settings_only) return fail(GR_ERROR_INVALID_ARGUMENT, "a metric made by gr_metric_from_info has no expressions to count");
    GR_TRY_BEGIN
    std::vector<float> vals(cfg_values, cfg_values + (cfg_values ? num_cfg_values : 0));
    const gr::MetricImpl concrete = m->desc.concrete(m->vars.substitution(vals));
    const sym::OpCount accel = sym::count_ops(concrete.accel);
    std::vector<sym::E> coord = concrete.to_polar;
    coord.push_back(concrete.distance_function);
    if (accel_ops) *accel_ops = accel.ops;
    if (accel_transcendentals) *accel_transcendentals = accel.transcendental;
    if (coord_ops) *coord_ops = sym::count_ops(coord).ops;
    return GR_OK;
    GR_TRY_END
}

int gr_argument_string_accelerations_call_trig(const char* argument_string) {
    if (!argument_string) return -1;
    std::vector<std::string> opts;
    for (auto& tok : split_arguments(argument_string))
        if (tok.rfind("-D", 0) == 0) opts.push_back(tok);
    return accelerations_without_trig(opts) ? 0 : 1;
}

int gr_program_precompile(const char* argument_string) {
    if (!argument_string) return fail(GR_ERROR_INVALID_ARGUMENT, "null argument string");
    std::string code, rest, setup;
    int rc = compile_code_object(argument_string, code);
    if (rc != GR_OK) return rc;
    rc = compile_code_object(argument_string, rest, nullptr, PART_REST);
    if (rc != GR_OK) return rc;
    return compile_setup_module(argument_string, setup);
}

// Builds that outlive their program (a program destroyed while its second code object is still inside the compiler detaches the
// worker): the process must not run its exit handlers - which tear down the compiler's own static state - under such a thread.  They
// are counted, and an atexit handler waits for the count to reach zero (a build is seconds; bounded at two minutes).
static std::mutex g_background_mu;
static std::condition_variable g_background_cv;
static int g_background_builds = 0;
static void background_builds_wait() {
    std::unique_lock<std::mutex> lock(g_background_mu);
    g_background_cv.wait_for(lock, std::chrono::seconds(120), [] { return g_background_builds == 0; });
}
static void background_builds_begin() {
    static std::once_flag once;
    std::call_once(once, [] { atexit(background_builds_wait); });
    std::lock_guard<std::mutex> lock(g_background_mu);
    g_background_builds++;
}
static void background_builds_end() {
    std::lock_guard<std::mutex> lock(g_background_mu);
    g_background_builds--;
    g_background_cv.notify_all();
}

// the frame path of a program: its PART_FRAME code object and the set-up module, each from the cache or built - the two builds side by
// side on two threads when both are missing (the compiler runs are independent; ~1.3 s and ~2.5 s of one core each for Kerr)
static int build_frame_path(const std::string& arguments, std::string& code, std::string& setup_code, std::string* key) {
    std::string setup_error;
    int setup_rc = GR_OK;
    { int a, b; rtc_version(a, b); }   // (before the second thread exists)
    // (both in the cache - every program after its first use: no thread at all)
    int rc = compile_setup_module(arguments, setup_code, /*cache_only=*/true);
    if (rc != GR_OK) return rc;
    if (!setup_code.empty()) return compile_code_object(arguments, code, key);
    std::thread side([&]() {
        setup_rc = compile_setup_module(arguments, setup_code);
        if (setup_rc != GR_OK) setup_error = g_error;
    });
    rc = compile_code_object(arguments, code, key);
    side.join();
    if (rc != GR_OK) return rc;
    if (setup_rc != GR_OK) return fail((gr_status)setup_rc, setup_error);
    return GR_OK;
}

// the function of kernel k, loading the program's other code object when the kernel lives there (blocks while that is still being
// built); nullptr with the error set when it cannot be had
static hipFunction_t function_of(gr_program* p, int k) {
    if (p->fn[k] || !p->in_rest[k]) return p->fn[k];
    std::lock_guard<std::mutex> lock(p->rest_mu);
    if (!p->rest_loaded) {
        auto& b = *p->rest;
        {
            std::unique_lock<std::mutex> wait(b.mu);
            b.cv.wait(wait, [&] { return b.done; });
        }
        if (p->rest_worker.joinable()) p->rest_worker.join();
        p->rest_rc = b.rc;
        p->rest_error = b.error;
        if (p->rest_rc == GR_OK) {
            hipError_t e = hipSetDevice(p->device);
            if (e == hipSuccess) e = hipModuleLoadData(&p->module_rest, b.code.data());
            for (int i = 0; i < K_COUNT && e == hipSuccess; i++)
                if (p->in_rest[i] && !p->fn[i]) e = hipModuleGetFunction(&p->fn[i], p->module_rest, KERNEL_NAMES[i]);
            if (e != hipSuccess) { p->rest_rc = GR_ERROR_DEVICE; p->rest_error = std::string("loading the program's second code object: ") + hipGetErrorString(e); }
        }
        b.code.clear();
        p->rest_loaded = true;
    }
    if (p->rest_rc != GR_OK) { (void)fail((gr_status)p->rest_rc, p->rest_error); return nullptr; }
    return p->fn[k];
}

int gr_program_precompile_frame_path(const char* argument_string) {
    if (!argument_string) return fail(GR_ERROR_INVALID_ARGUMENT, "null argument string");
    std::string code, setup;
    return build_frame_path(argument_string, code, setup, nullptr);
}

int gr_program_complete(gr_program* p) {
    if (!p) return fail(GR_ERROR_INVALID_ARGUMENT, "null program");
    for (int k = 0; k < K_COUNT; k++)
        if (p->in_rest[k] && !function_of(p, k)) return p->rest_rc != GR_OK ? p->rest_rc : fail(GR_ERROR_DEVICE, std::string("kernel missing: ") + KERNEL_NAMES[k]);
    return GR_OK;
}

int gr_program_create(const char* argument_string, int device, gr_program** out) {
    if (!argument_string || !out) return fail(GR_ERROR_INVALID_ARGUMENT, "null argument");
    std::string code, key, setup_code;
    int rc = build_frame_path(argument_string, code, setup_code, &key);
    if (rc != GR_OK) return rc;
    HIP_CHECK(hipSetDevice(device));
    auto p = std::make_unique<gr_program>();
    p->key = key;
    {
        // What was built for this key is not a function of the key alone: the occupancy rule of compile_code_object depends on a
        // spill size that varies from one compiler run to the next, and a cache directory may hold either outcome.  The key a
        // caller sees (and the committed hardware counters carry) therefore names the outcome too: registers and scratch of the
        // fused trace kernel as loaded.
        int vgprs = 0, scratch = 0;
        if (kernel_resources(code, "gr_trace_fused", vgprs, scratch)) p->key += "-v" + std::to_string(vgprs) + "s" + std::to_string(scratch);
    }
    p->device = device;
    p->arguments = argument_string;
    {
        const char* extra = getenv("GR_EXTRA_FLAGS");
        p->tile_shading = p->arguments.find("-DGR_TILE_SHADING") != std::string::npos || (extra && strstr(extra, "-DGR_TILE_SHADING"));
        p->cell_rows = p->arguments.find("-DGR_CELL_BLOCK=0") != std::string::npos || (extra && strstr(extra, "-DGR_CELL_BLOCK=0"));
    }
    static std::atomic<unsigned long long> next_serial{1};
    p->serial = next_serial.fetch_add(1);
    HIP_CHECK(hipModuleLoadData(&p->module, code.data()));
    HIP_CHECK(hipModuleLoadData(&p->setup_module, setup_code.data()));
    for (int k = 0; k < K_COUNT; k++) {
        if (is_setup_kernel(k)) {
            HIP_CHECK(hipModuleGetFunction(&p->fn[k], p->setup_module, KERNEL_NAMES[k]));
            continue;
        }
        // what the frame-path code object does not hold is in the other one - but for the two kernels that are built for some programs
        // only (gr_trace_pair: pair_kernel_applies; gr_trace_fused_parking: -DGR_PARKING) and belong to the frame path when they exist
        if (hipModuleGetFunction(&p->fn[k], p->module, KERNEL_NAMES[k]) != hipSuccess) {
            p->fn[k] = nullptr;
            (void)hipGetLastError();
            p->in_rest[k] = !(k == K_TRACE_PAIR || k == K_TRACE_FUSED_PARKING);
        }
    }
    if (!p->fn[K_TRACE_FUSED] || !p->fn[K_RENDER] || !p->fn[K_PREPASS_FUSED] || !p->fn[K_ORDER_TILES])
        return fail(GR_ERROR_COMPILE, "the frame-path code object lacks one of gr_trace_fused / gr_render / gr_prepass_fused / gr_order_tiles");
    // the other kernels: from the cache now, or from a build that starts here and is waited for by whoever first needs one of them
    p->rest = std::make_shared<gr_program::rest_build>();
    rc = compile_code_object(argument_string, p->rest->code, nullptr, PART_REST, /*cache_only=*/true);
    if (rc != GR_OK) return rc;
    if (!p->rest->code.empty()) p->rest->done = true;
    else {
        auto state = p->rest;
        const std::string arguments = argument_string;
        background_builds_begin();
        try {
        p->rest_worker = std::thread([state, arguments]() {
            std::string built;
            const int build_rc = compile_code_object(arguments, built, nullptr, PART_REST);   // compiler only: no device work on this thread
            {
                std::lock_guard<std::mutex> lock(state->mu);
                state->rc = build_rc;
                if (build_rc != GR_OK) state->error = g_error;
                state->code.swap(built);
                state->done = true;
                state->cv.notify_all();
            }
            background_builds_end();
        });
        } catch (const std::system_error& e) {   // no thread to be had: built here and now, on the caller's thread
            background_builds_end();
            const int build_rc = compile_code_object(arguments, state->code, nullptr, PART_REST);
            state->rc = build_rc;
            if (build_rc != GR_OK) state->error = g_error;
            state->done = true;
        }
    }
    const int huge = 0x7fffffff;
    HIP_CHECK(hipMalloc((void**)&p->tickets, gr_program::TICKET_RING * sizeof(unsigned int)));
    HIP_CHECK(hipDeviceGetAttribute(&p->compute_units, hipDeviceAttributeMultiprocessorCount, p->device));
    HIP_CHECK(hipMalloc(&p->huge_count, sizeof(int)));
    HIP_CHECK(hipMemcpy(p->huge_count, &huge, sizeof(int), hipMemcpyHostToDevice));
    *out = p.release();
    return GR_OK;
}

// ---- background build of the substituted program (metric_manager.hpp:153-219) ---------------------------------

struct gr_program_future {
    std::string arguments;
    int device = 0;
    std::thread worker;
    std::mutex mu;
    bool done = false;
    int rc = GR_OK;
    std::string error;
    std::string code;
    std::atomic<bool> cancelled{false};   // nobody wants the result any more: the worker stops at its next stage boundary
};

int gr_program_create_async(const char* argument_string, int device, gr_program_future** out) {
    if (!argument_string || !out) return fail(GR_ERROR_INVALID_ARGUMENT, "null argument");
    auto* f = new gr_program_future();
    f->arguments = argument_string;
    f->device = device;
    f->worker = std::thread([f]() {
        // the program's frame path (compiler only: no device work on this thread); its other kernels are built behind the swap, by the
        // program itself (gr_program_create)
        std::string code, setup;
        int rc = build_frame_path(f->arguments, code, setup, nullptr);
        std::lock_guard<std::mutex> lock(f->mu);
        f->rc = rc;
        if (rc != GR_OK) f->error = g_error;
        f->code.swap(code);
        f->done = true;
    });
    *out = f;
    return GR_OK;
}

int gr_program_future_poll(gr_program_future* f, gr_program** out) {
    if (!f || !out) return fail(GR_ERROR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    {
        std::lock_guard<std::mutex> lock(f->mu);
        if (!f->done) return 0;
        if (f->rc != GR_OK) return fail((gr_status)f->rc, f->error);
    }
    int rc = gr_program_create(f->arguments.c_str(), f->device, out);   // code object now comes from the cache
    return rc == GR_OK ? 1 : rc;
}

void gr_program_future_destroy(gr_program_future* f) {
    if (!f) return;
    if (f->worker.joinable()) f->worker.join();
    delete f;
}

// ---- metric_manager (metric_manager.hpp:19-219): which program to launch this frame -----------------------------------------
// The dynamic program (parameters read from memory) is built first and is usable whatever the parameters; the substituted one
// (parameters and features folded into the code) builds on a worker thread and is swapped in by the first gr_program_manager_current
// call that finds it finished (check_substitution, :172-219).  Changing a parameter puts the dynamic program back at once
// (check_recompile's soft recompile, :129-166) and starts a new substituted build; the finished one is retired, not destroyed,
// because frames launched with it may still be in flight on the caller's streams.
struct gr_program_manager {
    const gr_metric* metric = nullptr;
    int device = 0;
    gr_program* dynamic = nullptr;
    gr_program* substituted = nullptr;       // swapped in (using_swapped)
    gr_program_future* pending = nullptr;    // substituted_program_opt
    bool pending_is_stale = false;           // the parameters changed while it was building: its result is discarded, the next build follows it
    std::vector<gr_program*> retired;
    gr_features features{};
    std::vector<float> cfg;
    unsigned long long swaps = 0, updates = 0, builds_started = 0;
};

static int manager_start_build(gr_program_manager* pm) {
    size_t need = 0;
    int rc = gr_metric_argument_string(pm->metric, &pm->features, 1, pm->cfg.data(), (int)pm->cfg.size(), nullptr, 0, &need);
    if (rc != GR_OK) return rc;
    std::string arguments(need, '\0');
    rc = gr_metric_argument_string(pm->metric, &pm->features, 1, pm->cfg.data(), (int)pm->cfg.size(), &arguments[0], need, &need);
    if (rc != GR_OK) return rc;
    pm->pending_is_stale = false;
    pm->builds_started++;
    return gr_program_create_async(arguments.c_str(), pm->device, &pm->pending);
}

// AT MOST ONE BUILD IN FLIGHT (round 5).  A build that a parameter change has overtaken is not waited for - its worker cannot be
// interrupted inside the compiler, and joining it would stall the caller's frame loop for the rest of the compile every time a slider
// moves - but neither is a second one started next to it: a slider dragged for ten seconds changes the parameters every frame, and a
// build per change (round 4) meant hundreds of compiler threads and code-object-manager contexts alive at once.  The overtaken build is
// marked stale and told to stop at its next stage; the manager only remembers the latest parameters; whoever next finds the worker
// finished (update or current) discards its result and starts the build of the latest parameters.  The reference cancels the pending
// build the same way (metric_manager.hpp:129-166).  Returns < 0 on an error of the build that was started.
static int manager_follow_stale_build(gr_program_manager* pm) {
    if (!pm->pending || !pm->pending_is_stale) return GR_OK;
    bool done;
    {
        std::lock_guard<std::mutex> lock(pm->pending->mu);
        done = pm->pending->done;
    }
    if (!done) return GR_OK;
    gr_program_future_destroy(pm->pending);   // joins a worker that has left its last statement
    pm->pending = nullptr;
    return manager_start_build(pm);
}

static void manager_retire(gr_program_manager* pm, gr_program* p) {
    if (!p) return;
    pm->retired.push_back(p);
    while (pm->retired.size() > 2) {   // the oldest is two parameter changes old: nothing launched with it can still be queued
        (void)hipSetDevice(pm->device);
        (void)hipDeviceSynchronize();
        gr_program_destroy(pm->retired.front());
        pm->retired.erase(pm->retired.begin());
    }
}

int gr_program_manager_create(const gr_metric* m, int device, const gr_features* features, const float* cfg_values, int num_cfg_values,
                              gr_program_manager** out) {
    if (!m || !out) return fail(GR_ERROR_INVALID_ARGUMENT, "null argument");
    auto pm = std::make_unique<gr_program_manager>();
    pm->metric = m;
    pm->device = device;
    gr_features_default(&pm->features);
    pm->features.max_acceleration_change = m->cfg.max_acceleration_change;   // metric_manager.hpp:50
    if (features) pm->features = *features;
    const int n = (int)m->vars.names.size();
    for (int i = 0; i < n; i++) pm->cfg.push_back(cfg_values && i < num_cfg_values ? cfg_values[i] : m->vars.defaults[i]);
    size_t need = 0;
    int rc = gr_metric_argument_string(m, nullptr, 0, nullptr, 0, nullptr, 0, &need);
    if (rc != GR_OK) return rc;
    std::string arguments(need, '\0');
    rc = gr_metric_argument_string(m, nullptr, 0, nullptr, 0, &arguments[0], need, &need);
    if (rc != GR_OK) return rc;
    rc = gr_program_create(arguments.c_str(), device, &pm->dynamic);   // the first program of a metric is waited for (should_block)
    if (rc != GR_OK) return rc;
    rc = manager_start_build(pm.get());
    if (rc != GR_OK) { gr_program_destroy(pm->dynamic); return rc; }
    *out = pm.release();
    return GR_OK;
}

int gr_program_manager_update(gr_program_manager* pm, const gr_features* features, const float* cfg_values, int num_cfg_values) {
    if (!pm) return fail(GR_ERROR_INVALID_ARGUMENT, "null argument");
    gr_features f = pm->features;
    if (features) f = *features;
    std::vector<float> cfg = pm->cfg;
    for (size_t i = 0; i < cfg.size(); i++)
        if (cfg_values && (int)i < num_cfg_values) cfg[i] = cfg_values[i];
    if (memcmp(&f, &pm->features, sizeof(f)) == 0 && cfg == pm->cfg) return GR_OK;   // nothing changed: keep what is running or building
    pm->features = f;
    pm->cfg = cfg;
    pm->updates++;
    manager_retire(pm, pm->substituted);   // the substituted program is invalid for the new values: the dynamic one again
    pm->substituted = nullptr;
    if (pm->pending) {   // one build in flight: it is overtaken, the next one starts when its worker has finished
        pm->pending_is_stale = true;
        pm->pending->cancelled.store(true);
        return manager_follow_stale_build(pm);
    }
    return manager_start_build(pm);
}

int gr_program_manager_current(gr_program_manager* pm, int wait, gr_program** program, int* is_substituted) {
    if (!pm || !program) return fail(GR_ERROR_INVALID_ARGUMENT, "null argument");
    while (pm->pending) {
        if (pm->pending_is_stale) {   // an overtaken build: nothing of it is wanted; the build of the latest parameters follows it
            const int rc = manager_follow_stale_build(pm);
            if (rc != GR_OK) return rc;
            if (pm->pending && pm->pending_is_stale) {   // its worker is still inside the compiler
                if (!wait) break;
                std::this_thread::sleep_for(std::chrono::milliseconds(5));
            }
            continue;
        }
        gr_program* ready = nullptr;
        const int rc = gr_program_future_poll(pm->pending, &ready);
        if (rc < 0) {   // the substituted build failed: the dynamic program goes on serving, the error is reported once
            gr_program_future_destroy(pm->pending);
            pm->pending = nullptr;
            return rc;
        }
        if (rc == 1) {
            gr_program_future_destroy(pm->pending);
            pm->pending = nullptr;
            pm->substituted = ready;
            pm->swaps++;
            break;
        }
        if (!wait) break;
        std::this_thread::sleep_for(std::chrono::milliseconds(5));
    }
    *program = pm->substituted ? pm->substituted : pm->dynamic;
    if (is_substituted) *is_substituted = pm->substituted ? 1 : 0;
    return GR_OK;
}

gr_program* gr_program_manager_dynamic(gr_program_manager* pm) { return pm ? pm->dynamic : nullptr; }

void gr_program_manager_counters(const gr_program_manager* pm, unsigned long long out[4]) {
    if (!pm || !out) return;
    out[0] = pm->updates; out[1] = pm->swaps; out[2] = pm->builds_started; out[3] = pm->pending && pm->pending_is_stale ? 1 : 0;
}

void gr_program_manager_destroy(gr_program_manager* pm) {
    if (!pm) return;
    if (pm->pending) { pm->pending->cancelled.store(true); gr_program_future_destroy(pm->pending); }
    for (gr_program* p : pm->retired) gr_program_destroy(p);
    gr_program_destroy(pm->substituted);
    gr_program_destroy(pm->dynamic);
    delete pm;
}

void gr_program_destroy(gr_program* p) {
    delete p;   // ~gr_program releases the module and its device buffers (also on gr_program_create's error paths)
}

unsigned long long gr_program_serial(const gr_program* p) { return p ? p->serial : 0; }

const char* gr_program_build_key(const gr_program* p) { return p ? p->key.c_str() : ""; }

int gr_program_kernel_info(const gr_program* p, const char* kernel_name, int* vgprs, int* sgprs, int* scratch_bytes) {
    if (!p || !kernel_name) return fail(GR_ERROR_INVALID_ARGUMENT, "null argument");
    for (int k = 0; k < K_COUNT; k++) {
        if (strcmp(kernel_name, KERNEL_NAMES[k]) != 0) continue;
        int v = 0, l = 0;
        hipFunction_t f = function_of(const_cast<gr_program*>(p), k);
        if (!f) return p->in_rest[k] ? p->rest_rc : fail(GR_ERROR_INVALID_ARGUMENT, std::string("this program has no ") + kernel_name);
        HIP_CHECK(hipFuncGetAttribute(&v, HIP_FUNC_ATTRIBUTE_NUM_REGS, f));
        HIP_CHECK(hipFuncGetAttribute(&l, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, f));
        if (vgprs) *vgprs = v;
        if (sgprs) *sgprs = 0;
        if (scratch_bytes) *scratch_bytes = l;
        return GR_OK;
    }
    return fail(GR_ERROR_INVALID_ARGUMENT, "unknown kernel");
}

// ---- launch helpers ---------------------------------------------------------------------------

static int launch(gr_program* p, int k, void* stream, unsigned gx, unsigned gy, unsigned bx, unsigned by, void** args) {
    if (!p) return fail(GR_ERROR_INVALID_ARGUMENT, "null program");
    if (gx == 0 || gy == 0) return GR_OK;
    hipFunction_t f = function_of(p, k);
    if (!f) return p->in_rest[k] && p->rest_rc != GR_OK ? p->rest_rc : fail(GR_ERROR_INVALID_ARGUMENT, std::string("this program has no ") + KERNEL_NAMES[k]);
    HIP_CHECK(hipModuleLaunchKernel(f, gx, gy, 1, bx, by, 1, 0, (hipStream_t)stream, args, nullptr));
    return GR_OK;
}

static unsigned blocks(long long n, int b) { return n <= 0 ? 0u : (unsigned)((n + b - 1) / b); }

// A launcher refuses a NULL where its kernel dereferences unconditionally: the reference's clSetKernelArg returns CL_INVALID_MEM_OBJECT
// for a null buffer, a HIP launch takes it and the device faults (and a device fault ends the process).  cfg / dfg are not checked:
// a substituted program reads neither.
static int need(const char* who, std::initializer_list<const void*> buffers) {
    for (const void* b : buffers)
        if (!b) return fail(GR_ERROR_INVALID_ARGUMENT, std::string(who) + ": a required buffer is NULL");
    return GR_OK;
}
#define GR_NEED(who, ...) do { int rc_ = need(who, {__VA_ARGS__}); if (rc_ != GR_OK) return rc_; } while (0)

int gr_cart_to_generic(gr_program* p, void* stream, const void* in, void* out, int count, float flip, const void* cfg) {
    GR_NEED("gr_cart_to_generic", in, out);
    void* args[] = {&in, &out, &count, &flip, &cfg};
    return launch(p, K_CART_TO_GENERIC, stream, blocks(count, 64), 1, 64, 1, args);
}

int gr_init_basis_vectors(gr_program* p, void* stream, const void* generic_in, int count, const float speed[3],
                          void* e0, void* e1, void* e2, void* e3, const void* cfg) {
    GR_NEED("gr_init_basis_vectors", generic_in, e0, e1, e2, e3);
    float sx = speed ? speed[0] : 0.f, sy = speed ? speed[1] : 0.f, sz = speed ? speed[2] : 0.f;
    void* args[] = {&generic_in, &count, &sx, &sy, &sz, &e0, &e1, &e2, &e3, &cfg};
    return launch(p, K_INIT_BASIS, stream, blocks(count, 64), 1, 64, 1, args);
}

int gr_clear_termination_buffer(gr_program* p, void* stream, void* buf, int width, int height) {
    GR_NEED("gr_clear_termination_buffer", buf);
    void* args[] = {&buf, &width, &height};
    return launch(p, K_CLEAR_TERM, stream, blocks((long long)width * height, 256), 1, 256, 1, args);
}

int gr_tiled_slot_count(int width, int height) {
    const int T = 8;
    return ((width + T - 1) / T) * ((height + T - 1) / T) * T * T;
}

int gr_init_rays_generic(gr_program* p, void* stream, const void* camera_generic, const void* camera_quat, void* rays,
                         void* ray_count, int width, int height, const void* termination_buffer, int prepass_width,
                         int prepass_height, int flip, const void* e0, const void* e1, const void* e2, const void* e3,
                         const void* cfg, const void* dfg, int i_am_prepass, int tiled) {
    GR_NEED("gr_init_rays_generic", camera_generic, camera_quat, rays, ray_count, e0, e1, e2, e3);
    if (!termination_buffer && prepass_width != width && prepass_height != height) return fail(GR_ERROR_INVALID_ARGUMENT, "gr_init_rays_generic: a prepass grid without its termination buffer");
    long long slots = tiled ? gr_tiled_slot_count(width, height) : (long long)width * height;
    void* args[] = {&camera_generic, &camera_quat, &rays, &ray_count, &width, &height, &termination_buffer,
                    &prepass_width, &prepass_height, &flip, &e0, &e1, &e2, &e3, &cfg, &dfg, &i_am_prepass, &tiled};
    return launch(p, K_INIT_RAYS, stream, blocks(slots, 256), 1, 256, 1, args);
}

int gr_do_generic_rays(gr_program* p, void* stream, void* rays, const void* ray_count, int num_rays, void* tmin, void* tmax,
                       const void* cfg, const void* dfg, int width, int height, int mouse_x, int mouse_y, void* ray_write,
                       void* ray_write_counts, int max_write, void* attempt_counter) {
    GR_NEED("gr_do_generic_rays", rays, ray_count);
    void* args[] = {&rays, &ray_count, &tmin, &tmax, &cfg, &dfg, &width, &height, &mouse_x, &mouse_y,
                    &ray_write, &ray_write_counts, &max_write, &attempt_counter};
    return launch(p, K_DO_RAYS, stream, blocks(num_rays, 64), 1, 64, 1, args);
}

int gr_calculate_singularities(gr_program* p, void* stream, const void* rays, const void* count, int num_rays, void* term,
                               int width, int height) {
    GR_NEED("gr_calculate_singularities", rays, count, term);
    void* args[] = {&rays, &count, &term, &width, &height};
    return launch(p, K_CALC_SING, stream, blocks(num_rays, 256), 1, 256, 1, args);
}

int gr_calculate_render_data(gr_program* p, void* stream, const void* rays, const void* ray_count, int num_rays, void* rdata,
                             void* rdata_count, int width, int height, const void* cfg, const void* dfg) {
    GR_NEED("gr_calculate_render_data", rays, ray_count, rdata, rdata_count);
    void* args[] = {&rays, &ray_count, &rdata, &rdata_count, &width, &height, &cfg, &dfg};
    return launch(p, K_CALC_RDATA, stream, blocks(num_rays, 256), 1, 256, 1, args);
}

int gr_handle_adaptive_sampling(gr_program* p, void* stream, const void* rays, const void* ray_count, void* rdata,
                                void* rdata_count, void* new_rays, void* new_ray_count, const void* camera_generic,
                                const void* camera_quat, const void* e0, const void* e1, const void* e2, const void* e3,
                                int width, int height, const void* cfg, const void* dfg) {
    GR_NEED("gr_handle_adaptive_sampling", rays, ray_count, rdata, rdata_count, new_rays, new_ray_count, camera_generic, camera_quat, e0, e1, e2, e3);
    void* args[] = {&rays, &ray_count, &rdata, &rdata_count, &new_rays, &new_ray_count, &camera_generic, &camera_quat,
                    &e0, &e1, &e2, &e3, &width, &height, &cfg, &dfg};
    return launch(p, K_ADAPTIVE, stream, blocks(width / 2, 8), blocks(height / 2, 8), 8, 8, args);
}

int gr_render(gr_program* p, void* stream, const void* rdata, const void* rdata_count, int num_pixels, void* out,
              const void* bg1, const void* bg2, int bg_width, int bg_height, int bg_levels, int width, int height,
              int max_probes, const void* cfg, const void* dfg) {
    GR_NEED("gr_render", rdata, rdata_count, out, bg1, bg2);
    if (bg_width <= 0 || bg_height <= 0 || bg_levels <= 0) return fail(GR_ERROR_INVALID_ARGUMENT, "gr_render: the background's width, height and levels");
    int block_pixels = num_pixels > 0 ? num_pixels : 1, rank = 0, count = 1, compact = 0, seams_only = 0;
    void* args[] = {&rdata, &rdata_count, &out, &bg1, &bg2, &bg_width, &bg_height, &bg_levels, &width, &height,
                    &max_probes, &cfg, &dfg, &num_pixels, &block_pixels, &rank, &count, &compact, &seams_only};
    return launch(p, K_RENDER, stream, blocks(num_pixels, 256), 1, 256, 1, args);
}

int gr_strip_local_blocks(int height, int block_rows, int strip_rank, int strip_count) {
    if (block_rows <= 0 || strip_count <= 0) return 0;
    int total = (height + block_rows - 1) / block_rows;
    return strip_rank < total ? (total - strip_rank + strip_count - 1) / strip_count : 0;
}

// strip mode: render_data is indexed by pixel; shade this device's row blocks (block-cyclic)
int gr_render_strips(gr_program* p, void* stream, const void* rdata, void* out, const void* bg1, const void* bg2, int bg_width,
                     int bg_height, int bg_levels, int width, int height, int block_rows, int strip_rank, int strip_count,
                     int compact_out, int max_probes, const void* cfg, const void* dfg) {
    if (block_rows <= 0 || strip_count <= 0 || strip_rank < 0 || strip_rank >= strip_count)
        return fail(GR_ERROR_INVALID_ARGUMENT, "bad strip parameters");
    int local_blocks = gr_strip_local_blocks(height, block_rows, strip_rank, strip_count);
    int block_pixels = block_rows * width;
    int num = local_blocks * block_pixels;
    const void* rdata_count = p ? p->huge_count : nullptr;
    int seams_only = 0;
    void* args[] = {&rdata, &rdata_count, &out, &bg1, &bg2, &bg_width, &bg_height, &bg_levels, &width, &height,
                    &max_probes, &cfg, &dfg, &num, &block_pixels, &strip_rank, &strip_count, &compact_out, &seams_only};
    return launch(p, K_RENDER, stream, blocks(num, 256), 1, 256, 1, args);
}

// the pixels gr_trace_fused_launch's in-tile shading leaves: last column and last row of every 8x8 tile of this device's blocks
int gr_render_seams(gr_program* p, void* stream, const void* rdata, void* out, const void* bg1, const void* bg2, int bg_width,
                    int bg_height, int bg_levels, int width, int height, int block_rows, int strip_rank, int strip_count,
                    int compact_out, int max_probes, const void* cfg, const void* dfg) {
    if (strip_count <= 1) { strip_count = 1; strip_rank = 0; block_rows = ((height + 7) / 8) * 8; }
    if (block_rows <= 0 || block_rows % 8 != 0 || strip_rank < 0 || strip_rank >= strip_count || width <= 0 || width % 8 != 0)
        return fail(GR_ERROR_INVALID_ARGUMENT, "gr_render_seams: width and block_rows must be multiples of 8");
    int local_blocks = gr_strip_local_blocks(height, block_rows, strip_rank, strip_count);
    int block_pixels = block_rows * width;
    long long items = (long long)local_blocks * (width / 8) * (block_rows / 8) * 15;
    if (items > 0x7fffffff) return fail(GR_ERROR_INVALID_ARGUMENT, "gr_render_seams: image too large");
    int num = (int)items, seams_only = 1;
    const void* rdata_count = p ? p->huge_count : nullptr;
    void* args[] = {&rdata, &rdata_count, &out, &bg1, &bg2, &bg_width, &bg_height, &bg_levels, &width, &height,
                    &max_probes, &cfg, &dfg, &num, &block_pixels, &strip_rank, &strip_count, &compact_out, &seams_only};
    return launch(p, K_RENDER, stream, blocks(num, 256), 1, 256, 1, args);
}

int gr_internal_fail(int code, const char* msg) { return fail((gr_status)code, msg ? msg : ""); }

int gr_prepass_fused_strips(gr_program* p, void* stream, const void* camera_generic, const void* camera_quat, void* term,
                            int prepass_width, int prepass_height, const void* e0, const void* e1, const void* e2, const void* e3,
                            const void* cfg, const void* dfg, int image_height, int block_rows, int strip_rank, int strip_count,
                            void* cell_attempts, int row_margin) {
    if (strip_count <= 1) { strip_count = 1; strip_rank = 0; block_rows = 8; }
    if (block_rows <= 0 || strip_rank < 0 || strip_rank >= strip_count || image_height <= 0)
        return fail(GR_ERROR_INVALID_ARGUMENT, "bad strip description");
    void* args[] = {&camera_generic, &camera_quat, &term, &prepass_width, &prepass_height, &e0, &e1, &e2, &e3, &cfg, &dfg,
                    &image_height, &block_rows, &strip_rank, &strip_count, &cell_attempts, &row_margin};
    if (row_margin < 0) return fail(GR_ERROR_INVALID_ARGUMENT, "negative row margin");
    return launch(p, K_PREPASS_FUSED, stream, blocks((long long)prepass_width * prepass_height, 64), 1, 64, 1, args);
}

int gr_camera_prepass(gr_program* p, void* stream, const void* position_cart, float flip, const float basis_speed[3], void* position_generic_out,
                      void* e0_out, void* e1_out, void* e2_out, void* e3_out, const void* camera_quat, void* term, int prepass_width,
                      int prepass_height, const void* cfg, const void* dfg, int image_height, int block_rows, int strip_rank, int strip_count,
                      void* cell_attempts, int row_margin) {
    if (!basis_speed) return fail(GR_ERROR_INVALID_ARGUMENT, "null basis speed");
    if (prepass_width < 0 || prepass_height < 0) return fail(GR_ERROR_INVALID_ARGUMENT, "negative prepass size");
    if (strip_count <= 1) { strip_count = 1; strip_rank = 0; block_rows = 8; }
    if (block_rows <= 0 || strip_rank < 0 || strip_rank >= strip_count) return fail(GR_ERROR_INVALID_ARGUMENT, "bad strip description");
    if (image_height <= 0) image_height = prepass_height > 0 ? prepass_height * 16 : 16;
    float sx = basis_speed[0], sy = basis_speed[1], sz = basis_speed[2];
    if (row_margin < 0) return fail(GR_ERROR_INVALID_ARGUMENT, "negative row margin");
    // the camera's coordinates and tetrad: one lane of the set-up module (IEEE arithmetic, kernels/camera.hip) ...
    void* setup_args[] = {&position_cart, &flip, &sx, &sy, &sz, &position_generic_out, &e0_out, &e1_out, &e2_out, &e3_out, &cfg};
    int rc = launch(p, K_CAMERA_SETUP, stream, 1, 1, 64, 1, setup_args);
    if (rc != GR_OK) return rc;
    // ... then the prepass grid reads them back (same stream)
    const long long cells = (long long)prepass_width * prepass_height;
    if (cells <= 0) return GR_OK;
    const void* camera_generic = position_generic_out;
    const void *e0 = e0_out, *e1 = e1_out, *e2 = e2_out, *e3 = e3_out;
    void* args[] = {&camera_generic, &camera_quat, &term, &prepass_width, &prepass_height, &e0, &e1, &e2, &e3, &cfg, &dfg,
                    &image_height, &block_rows, &strip_rank, &strip_count, &cell_attempts, &row_margin};
    return launch(p, K_PREPASS_FUSED, stream, blocks(cells, 64), 1, 64, 1, args);
}

int gr_prepass_fused(gr_program* p, void* stream, const void* camera_generic, const void* camera_quat, void* term,
                     int prepass_width, int prepass_height, const void* e0, const void* e1, const void* e2, const void* e3,
                     const void* cfg, const void* dfg) {
    return gr_prepass_fused_strips(p, stream, camera_generic, camera_quat, term, prepass_width, prepass_height, e0, e1, e2, e3, cfg, dfg,
                                   prepass_height * 16, 8, 0, 1, nullptr, 0);
}

// tiles (one wave each) a device traces: its row blocks cut into 8x8 tiles + per block the halo row in 64-pixel pieces; 0 when
// the description is invalid (trace_launch says why)
static long long device_tile_count(int width, int height, int block_rows, int strip_rank, int strip_count) {
    const int T = 8;
    if (width <= 0 || height <= 0) return 0;
    if (strip_count <= 1) { strip_count = 1; strip_rank = 0; block_rows = ((height + T - 1) / T) * T; }
    if (block_rows <= 0 || block_rows % T != 0 || strip_rank < 0 || strip_rank >= strip_count) return 0;
    long long per_block = (long long)((width + T - 1) / T) * (block_rows / T) + (strip_count > 1 ? (width + 63) / 64 : 0);
    return per_block * gr_strip_local_blocks(height, block_rows, strip_rank, strip_count);
}

long long gr_tile_order_bytes(int width, int height, int block_rows, int strip_rank, int strip_count) {
    return (32 + 2 * device_tile_count(width, height, block_rows, strip_rank, strip_count)) * 4;   // header (2 x 16 classes), list, classes
}

static int order_tiles_launch(gr_program* p, void* stream, const void* term, const void* cell_attempts, int prepass_width, int prepass_height,
                              int width, int height, int block_rows, int strip_rank, int strip_count, void* tile_order, const void* tile_history,
                              int shift_x = 0, int shift_y = 0) {
    if (strip_count <= 1) { strip_count = 1; strip_rank = 0; block_rows = ((height + 7) / 8) * 8; }
    long long tiles = device_tile_count(width, height, block_rows, strip_rank, strip_count);
    if (tiles <= 0 || tiles > 0x7fffffff) return fail(GR_ERROR_INVALID_ARGUMENT, "gr_order_tiles: bad image or strip description");
    int total = (int)tiles;
    HIP_CHECK(hipSetDevice(p->device));
    HIP_CHECK(hipMemsetAsync(tile_order, 0, 128, (hipStream_t)stream));
    // how far (in tiles) a tile looks around itself in the history: the image may have moved by that much since
    static const int reach_default = [] { const char* e = getenv("GR_TILE_HISTORY_REACH"); int v = e ? atoi(e) : 2; return (v >= 0 && v <= 8) ? v : 2; }();
    int history_reach = reach_default;
    for (int phase = 0; phase < 2; phase++) {
        void* args[] = {&term, &cell_attempts, &prepass_width, &prepass_height, &width, &height, &block_rows, &strip_rank, &strip_count,
                        &total, &tile_order, &phase, &tile_history, &history_reach, &shift_x, &shift_y};
        int rc = launch(p, K_ORDER_TILES, stream, blocks(total, 1024), 1, 1024, 1, args);
        if (rc != GR_OK) return rc;
    }
    return GR_OK;
}

int gr_order_tiles(gr_program* p, void* stream, const void* term, const void* cell_attempts, int prepass_width, int prepass_height,
                   int width, int height, int block_rows, int strip_rank, int strip_count, void* tile_order) {
    if (!p || !term || !cell_attempts || !tile_order || prepass_width <= 0 || prepass_height <= 0)
        return fail(GR_ERROR_INVALID_ARGUMENT, "gr_order_tiles: null argument or no prepass");
    return order_tiles_launch(p, stream, term, cell_attempts, prepass_width, prepass_height, width, height, block_rows, strip_rank, strip_count,
                              tile_order, nullptr);
}

int gr_order_tiles_by_history(gr_program* p, void* stream, const void* tile_history, int width, int height, int block_rows, int strip_rank,
                              int strip_count, void* tile_order, int shift_x, int shift_y) {
    if (!p || !tile_history || !tile_order) return fail(GR_ERROR_INVALID_ARGUMENT, "gr_order_tiles_by_history: null argument");
    return order_tiles_launch(p, stream, nullptr, nullptr, 0, 0, width, height, block_rows, strip_rank, strip_count, tile_order, tile_history,
                              shift_x, shift_y);
}

// workgroups of `wg` lanes of a trace kernel the device holds at once (asked of the runtime once per kernel); < 0: -error code
static long long resident_trace_groups(gr_program* p, int kernel_index, int wg) {
    if (!p->resident_groups_per_cu[kernel_index]) {
        int n = 0;
        if (hipSetDevice(p->device) != hipSuccess) return -(long long)fail(GR_ERROR_DEVICE, "hipSetDevice");
        if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&n, function_of(p, kernel_index), wg, 0) != hipSuccess || n < 1) {
            (void)hipGetLastError();
            n = 4 * 8 * 64 / wg;
        }
        p->resident_groups_per_cu[kernel_index] = n;
    }
    return (long long)p->compute_units * p->resident_groups_per_cu[kernel_index];
}

size_t gr_parking_lot_bytes(int slots, int groups, size_t* words_bytes) {
    if (words_bytes) *words_bytes = (16 + (size_t)(groups > 0 ? groups : 0)) * sizeof(unsigned int);
    return (size_t)(slots > 0 ? slots : 0) * 6 * 16;
}

long long gr_trace_fused_wave_slots(gr_program* p) {
    if (!p) return 0;
    const long long groups = resident_trace_groups(p, K_TRACE_FUSED, 256);
    return groups < 0 ? 0 : groups * 4;
}

// gr_trace_fused (rays_per_lane 1) and gr_trace_pair (2): same tiles, same arguments; a pair wave takes two tile-waves
static int trace_launch(gr_program* p, int rays_per_lane, void* stream, const void* camera_generic, const void* camera_quat, void* rdata,
                        int width, int height, int block_rows, int strip_rank, int strip_count, const void* term, int prepass_width,
                        int prepass_height, const void* e0, const void* e1, const void* e2, const void* e3, const void* cfg,
                        const void* dfg, void* attempt_counter, int lattice = 1, int pending_only = 0, const void* tile_order = nullptr,
                        int waves_per_simd = 0, const gr_trace_shading* shading_in = nullptr, int inline_prepass = 0, void* tile_cost = nullptr,
                        int tile_order_by_history = 0, void* lattice_rays = nullptr, const gr_parking_lot* parking = nullptr,
                        int speculative_classes_in = 0, void* guessed = nullptr) {
    const int T = 8;
    if (!p) return fail(GR_ERROR_INVALID_ARGUMENT, "null program");
    // the prepass inside the launch: its cell waves are the first tickets (gr_trace_fused's prepass_tickets)
    int prepass_tickets = 0;
    if (inline_prepass) {
        if (rays_per_lane != 1 || pending_only || (tile_order && !tile_order_by_history) || !term || prepass_width <= 0 ||
            prepass_height <= 0 || prepass_width == width || prepass_height == height)
            return fail(GR_ERROR_INVALID_ARGUMENT, "inline_prepass: gr_trace_fused on every pixel of its rows (or the lattice launch of adaptive sampling), in image order or the order of "
                                                   "gr_order_tiles_by_history (gr_order_tiles' needs the prepass first), with a prepass grid");
        // a cell wave is 8 x 8 cells (trace.hip: GR_CELL_BLOCK; a program built with -DGR_CELL_BLOCK=0 takes 64 cells of a row)
        prepass_tickets = p->cell_rows ? (int)(((long long)prepass_width * prepass_height + 63) / 64) : ((prepass_width + 7) / 8) * ((prepass_height + 7) / 8);
    }
    if ((lattice != 1 && lattice != 2) || ((lattice == 2 || pending_only) && rays_per_lane != 1))
        return fail(GR_ERROR_INVALID_ARGUMENT, "lattice / pending_only: gr_trace_fused only");
    const bool parks = parking && parking->lanes > 0;
    if (parks) {
        if (!p->fn[K_TRACE_FUSED_PARKING])
            return fail(GR_ERROR_INVALID_ARGUMENT, "parking: the program was built without -DGR_PARKING in its argument string (gr_program_has_parking)");
        if (rays_per_lane != 1 || lattice != 1 || pending_only || (shading_in && shading_in->out))
            return fail(GR_ERROR_INVALID_ARGUMENT, "parking: gr_trace_fused on every pixel of its rows, one ray per lane, no in-tile shading");
        if (!parking->records || !parking->words || parking->lanes > 64 || parking->trips < 1 || parking->slots < 64 || parking->slots > (1 << 24) ||
            parking->groups < 1)
            return fail(GR_ERROR_INVALID_ARGUMENT, "parking: records and words (gr_parking_lot_bytes), 1 <= lanes <= 64, trips >= 1, 64 <= slots <= 2^24, groups >= 1");
    }
    if (rays_per_lane == 2 && !p->fn[K_TRACE_PAIR])
        return fail(GR_ERROR_INVALID_ARGUMENT, "this program has no gr_trace_pair kernel (its expressions do not instantiate on pairs)");
    if (strip_count <= 1) {   // one block covering the image
        strip_count = 1;
        strip_rank = 0;
        block_rows = ((height + T - 1) / T) * T;
    }
    if (block_rows <= 0 || block_rows % T != 0 || strip_rank < 0 || strip_rank >= strip_count)
        return fail(GR_ERROR_INVALID_ARGUMENT, "block_rows must be a positive multiple of 8 and 0 <= strip_rank < strip_count");
    if (strip_count > 1 && height > 1 && (height - 1) % block_rows == 0)
        return fail(GR_ERROR_INVALID_ARGUMENT, "the last image row must not start a block (its filter reads the row above)");
    int local_blocks = gr_strip_local_blocks(height, block_rows, strip_rank, strip_count);
    long long waves_per_block = (long long)((width + T - 1) / T) * (block_rows / T) + (strip_count > 1 ? (width + 63) / 64 : 0);
    if (lattice == 2) {   // tiles of the whole half-resolution grid, whoever owns the rows (the kernel leaves the rows of others alone)
        waves_per_block = (long long)((width / 2 + T - 1) / T) * ((height / 2 + T - 1) / T);
        local_blocks = 1;
    }
    // four tile-waves per workgroup (measured on MI355X, 4K Kerr: 64 -> 7.59 ms, 128 -> 7.43, 256 -> 7.24; one wave per SIMD
    // still leaves the full 512-VGPR budget to the heaviest metrics).  Experiment hooks: GR_TRACE_BLOCK=64|128|256 with the
    // kernel built with the same -DGR_TRACE_BLOCK through GR_EXTRA_FLAGS; GR_TRACE_PERSISTENT=0 launches one wave per tile.
    static const int wg = [] { const char* e = getenv("GR_TRACE_BLOCK"); int v = e ? atoi(e) : 256; return (v == 64 || v == 128) ? v : 256; }();
    static const bool persistent = [] { const char* e = getenv("GR_TRACE_PERSISTENT"); return !(e && e[0] == '0'); }();
    long long waves = waves_per_block * local_blocks;
    if (waves <= 0) return GR_OK;
    if (waves > 0x7fffffff) return fail(GR_ERROR_INVALID_ARGUMENT, "too many tiles");
    int total_waves = (int)waves;
    long long groups = (((waves + rays_per_lane - 1) / rays_per_lane + prepass_tickets) * 64 + wg - 1) / wg;
    unsigned int* tickets = nullptr;
    // persistent mode only pays when there are more tiles than wave slots.  The launch is exactly as many workgroups as the
    // kernel's register and scratch footprint lets the device hold (asked of the runtime once per kernel): a larger one
    // leaves workgroups queued behind the resident ones which start only when the tickets are gone, and which the
    // dispatcher then has to retire before the next launch's workgroups get the freed slots
    // experiment hook: GR_TRACE_WAVES_PER_SIMD=k launches k persistent waves per SIMD whatever fits (occupancy studies)
    static const int forced_waves_per_simd = [] { const char* e = getenv("GR_TRACE_WAVES_PER_SIMD"); int v = e ? atoi(e) : 0; return (v >= 1 && v <= 8) ? v : 0; }();
    // (the lattice launch of adaptive sampling that leaves its rays behind is a kernel of its own: gr_trace_fused is not touched by it)
    const int kernel_index = parks ? K_TRACE_FUSED_PARKING : rays_per_lane == 2 ? K_TRACE_PAIR : (lattice == 2 && lattice_rays) ? K_TRACE_FUSED_LATTICE : K_TRACE_FUSED;
    long long resident_groups = resident_trace_groups(p, kernel_index, wg);
    if (resident_groups < 0) return (int)-resident_groups;
    // a caller that keeps several frames in flight may take fewer slots per launch: two smaller launches then share the device
    // and the one drains while the other is in full swing (gr_frame_options.trace_waves_per_simd)
    if (forced_waves_per_simd)
        resident_groups = (long long)p->compute_units * 4 * forced_waves_per_simd * 64 / wg;
    else if (waves_per_simd >= 1 && waves_per_simd <= 8)
        resident_groups = std::min(resident_groups, (long long)p->compute_units * 4 * waves_per_simd * 64 / wg);
    if ((persistent && groups > resident_groups) || prepass_tickets || parks || guessed) {   // (... and so do the pixels traced ahead)   // prepass tickets need the ticket order whatever the size, and so does a lot
        tickets = p->tickets + (p->next_ticket.fetch_add(1) % gr_program::TICKET_RING);
        HIP_CHECK(hipSetDevice(p->device));
        HIP_CHECK(hipMemsetAsync(tickets, 0, sizeof(unsigned int), (hipStream_t)stream));
        groups = std::min(groups, resident_groups);
    }
    // tiles per ticket: one, unless a wave would draw more than 32 tickets (then as many as keep it at 32, at most 8)
    int ticket_tiles = 1;
    if (tickets) {
        const long long launched_waves = groups * (wg / 64);
        const long long per_wave = (waves + prepass_tickets + launched_waves - 1) / launched_waves;
        ticket_tiles = (int)std::min<long long>(8, std::max<long long>(1, (per_wave + 31) / 32));
        if (const char* e = getenv("GR_TICKET_TILES")) { const int v = atoi(e); if (v >= 1 && v <= 64) ticket_tiles = v; }
    }
    // the kernel's trace_shading, by value (same layout)
    struct { void* out; const void* bg1; const void* bg2; int bg_width, bg_height, bg_levels, most_probes, compact_out; } shading = {};
    if (shading_in && shading_in->out) {
        if (!p->tile_shading)
            return fail(GR_ERROR_INVALID_ARGUMENT, "in-tile shading: the program was built without -DGR_TILE_SHADING in its argument string");
        if (rays_per_lane != 1 || lattice != 1 || pending_only || width % T != 0 || height % T != 0)
            return fail(GR_ERROR_INVALID_ARGUMENT, "in-tile shading: gr_trace_fused on every pixel of an image whose sides are multiples of 8");
        shading.out = shading_in->out; shading.bg1 = shading_in->background1; shading.bg2 = shading_in->background2;
        shading.bg_width = shading_in->bg_width; shading.bg_height = shading_in->bg_height; shading.bg_levels = shading_in->bg_levels;
        shading.most_probes = shading_in->max_probes; shading.compact_out = strip_count > 1 ? shading_in->compact_out : 0;
    }
    if (tile_cost && (rays_per_lane != 1 || pending_only || (lattice == 2 && strip_count > 1)))
        return fail(GR_ERROR_INVALID_ARGUMENT, "tile_cost: gr_trace_fused on every pixel of its rows, or the lattice launch of a whole frame");
    // (every argument has been checked by now: nothing below fails for a reason of the caller's, and nothing above has touched a buffer)
    if (prepass_tickets)   // every cell unknown (GR_CELL_UNKNOWN = -1) until its ray has been traced
        HIP_CHECK(hipMemsetAsync(const_cast<void*>(term), 0xff, (size_t)prepass_width * prepass_height * sizeof(int), (hipStream_t)stream));
    if (tile_cost) {
        HIP_CHECK(hipSetDevice(p->device));
        HIP_CHECK(hipMemsetAsync(tile_cost, 0, (size_t)total_waves * sizeof(unsigned int), (hipStream_t)stream));
    }
    // Whether the list's last class is a promise (gr_order_tiles: nothing to trace, nothing to look up) or a guess
    // (gr_order_tiles_by_history) is written into the list by the launch that made it, and the kernel reads it there; the caller's
    // flag only says which he thinks it is, for the checks above.
    int last_class_is_skipped = 0;   // (kept in the kernel's parameter list: 1 would force the promise, nothing passes it)
    // ... its upper bits: how many of the list's classes, dearest first, do not wait for the prepass cells they look at when those are
    // traced by this launch (trace_fused_body: speculative tiles).  Classes are octaves of a tile's longest ray in the frame before,
    // 16 384 attempts = class 0: the default, 5, are the tiles that had a ray of 1 024 attempts or more.
    static const int speculative_classes = [] { const char* e = getenv("GR_SPECULATIVE_CLASSES"); int v = e ? atoi(e) : GR_DEFAULT_SPECULATIVE_CLASSES; return (v >= 0 && v <= 14) ? v : 0; }();
    const int speculative = speculative_classes_in < 0 ? 0 : speculative_classes_in == 0 ? speculative_classes : std::min(speculative_classes_in, 14);
    if (prepass_tickets && tile_order && tile_order_by_history && !parks) last_class_is_skipped |= speculative << 8;
    // the kernel's parking_lot, by value (same layout); an empty lot before every launch
    struct { void* records; void* words; int lanes, trips, slots, groups; } lot = {};
    if (parks) {
        lot.records = parking->records; lot.words = parking->words; lot.lanes = parking->lanes; lot.trips = parking->trips;
        lot.slots = parking->slots; lot.groups = parking->groups;
        HIP_CHECK(hipSetDevice(p->device));
        HIP_CHECK(hipMemsetAsync(parking->words, 0, (16 + (size_t)parking->groups) * sizeof(unsigned int), (hipStream_t)stream));
    }
    void* args[] = {&camera_generic, &camera_quat, &rdata, &width, &height, &block_rows, &strip_rank, &strip_count, &term,
                    &prepass_width, &prepass_height, &e0, &e1, &e2, &e3, &cfg, &dfg, &attempt_counter, &tickets, &total_waves,
                    &lattice, &pending_only, &tile_order, &shading, &prepass_tickets, &ticket_tiles, &tile_cost,
                    &last_class_is_skipped, &lattice_rays, &lot, &guessed};   // the last eleven: gr_trace_fused only (gr_trace_pair's parameter list ends before them), the lot gr_trace_fused_parking and (unused) gr_trace_fused_lattice, the very last gr_trace_fused_lattice only
    return launch(p, kernel_index, stream, (unsigned)groups, 1, wg, 1, args);
}

int gr_trace_fused_adaptive(gr_program* p, void* stream, const void* camera_generic, const void* camera_quat, void* rdata, int width,
                            int height, const void* term, int prepass_width, int prepass_height, const void* e0, const void* e1,
                            const void* e2, const void* e3, const void* cfg, const void* dfg, void* attempt_counter, int lattice,
                            int pending_only, void* lattice_rays) {
    return trace_launch(p, 1, stream, camera_generic, camera_quat, rdata, width, height, 0, 0, 1, term, prepass_width, prepass_height, e0, e1,
                        e2, e3, cfg, dfg, attempt_counter, lattice, pending_only, nullptr, 0, nullptr, 0, nullptr, 0,
                        lattice == 2 ? lattice_rays : nullptr);
}

int gr_adaptive_refine_strips(gr_program* p, void* stream, void* rdata, void* pending_count, int width, int height, const void* dfg,
                              int block_rows, int strip_rank, int strip_count, const void* lattice_rays, const void* cfg) {
    if (strip_count <= 1) { strip_count = 1; strip_rank = 0; block_rows = ((height + 7) / 8) * 8; }
    if (block_rows <= 0 || block_rows % 8 != 0 || strip_rank < 0 || strip_rank >= strip_count)
        return fail(GR_ERROR_INVALID_ARGUMENT, "bad strip description");
    if (lattice_rays && !cfg) return fail(GR_ERROR_INVALID_ARGUMENT, "gr_adaptive_refine: lattice_rays needs the metric's cfg");
    void* pending_list = nullptr;
    int phase = 0;
    const void* block_cost_before = nullptr;
    void* args[] = {&rdata, &pending_count, &width, &height, &dfg, &block_rows, &strip_rank, &strip_count, &lattice_rays, &cfg, &pending_list, &phase,
                    &block_cost_before};
    return launch(p, K_ADAPTIVE_REFINE, stream, (unsigned)((width / 2 + 15) / 16), (unsigned)((height / 2 + 15) / 16), 16, 16, args);
}

size_t gr_pending_list_bytes(int width, int height) {
    if (width < 2 || height < 2) return 0;
    return (128 + 3 * (size_t)(width / 2) * (height / 2)) * sizeof(unsigned int);   // 64 class counts, 64 cursors, the pixels
}

size_t gr_lattice_rays_bytes(int width, int height) {
    if (width < 2 || height < 2) return 0;
    return (size_t)(width / 2) * (height / 2) * (3 * 16 + 4);   // three float4 of end state + the ray's attempts per lattice pixel
}

int gr_adaptive_refine_list(gr_program* p, void* stream, void* rdata, void* pending_count, int width, int height, const void* dfg, int block_rows,
                            int strip_rank, int strip_count, const void* lattice_rays, const void* cfg, void* pending_list,
                            const void* block_cost_before) {
    if (!p || !rdata || !pending_list) return fail(GR_ERROR_INVALID_ARGUMENT, "gr_adaptive_refine_list: null argument");
    if (strip_count <= 1) { strip_count = 1; strip_rank = 0; block_rows = ((height + 7) / 8) * 8; }
    if (block_rows <= 0 || block_rows % 8 != 0 || strip_rank < 0 || strip_rank >= strip_count)
        return fail(GR_ERROR_INVALID_ARGUMENT, "bad strip description");
    if (lattice_rays && !cfg) return fail(GR_ERROR_INVALID_ARGUMENT, "gr_adaptive_refine_list: lattice_rays needs the metric's cfg");
    HIP_CHECK(hipSetDevice(p->device));
    HIP_CHECK(hipMemsetAsync(pending_list, 0, 128 * sizeof(unsigned int), (hipStream_t)stream));
    for (int phase = 0; phase < 2; phase++) {   // decide, mark and count by cost class; then deal every marked pixel its place
        void* args[] = {&rdata, &pending_count, &width, &height, &dfg, &block_rows, &strip_rank, &strip_count, &lattice_rays, &cfg, &pending_list, &phase,
                        &block_cost_before};
        int rc = launch(p, K_ADAPTIVE_REFINE, stream, (unsigned)((width / 2 + 15) / 16), (unsigned)((height / 2 + 15) / 16), 16, 16, args);
        if (rc != GR_OK) return rc;
    }
    return GR_OK;
}

int gr_trace_pending(gr_program* p, void* stream, const void* camera_generic, const void* camera_quat, void* rdata, int width, int height,
                     const void* e0, const void* e1, const void* e2, const void* e3, const void* cfg, const void* dfg, void* attempt_counter,
                     const void* pending_list, int waves_per_simd, void* block_cost, void* guessed_next) {
    if (!p || !rdata || !pending_list) return fail(GR_ERROR_INVALID_ARGUMENT, "gr_trace_pending: null argument");
    if (width < 2 || height < 2) return GR_OK;
    const int wg = 256;
    long long groups = resident_trace_groups(p, K_TRACE_PENDING, wg);
    if (groups < 0) return (int)-groups;
    if (waves_per_simd >= 1 && waves_per_simd <= 8) groups = std::min(groups, (long long)p->compute_units * 4 * waves_per_simd * 64 / wg);
    // (how many entries the list holds is only known on the device: the launch fills the machine and its waves draw tickets until
    // the list is used up - no more workgroups than the worst case needs, though)
    const long long worst = (3LL * (width / 2) * (height / 2) + wg - 1) / wg;
    groups = std::max(1LL, std::min(groups, worst));
    unsigned int* tickets = p->tickets + (p->next_ticket.fetch_add(1) % gr_program::TICKET_RING);
    HIP_CHECK(hipSetDevice(p->device));
    HIP_CHECK(hipMemsetAsync(tickets, 0, sizeof(unsigned int), (hipStream_t)stream));
    void* args[] = {&camera_generic, &camera_quat, &rdata, &width, &height, &e0, &e1, &e2, &e3, &cfg, &dfg, &attempt_counter, &tickets, &pending_list,
                    &block_cost, &guessed_next};
    return launch(p, K_TRACE_PENDING, stream, (unsigned)groups, 1, wg, 1, args);
}

size_t gr_guessed_bytes(void) { return (8 + (size_t)65536 * (1 + 1 + 8)) * sizeof(unsigned int); }   // program.hip: GR_GUESSED_HEADER, GR_GUESSED_CAPACITY pixels, attempts, records

int gr_apply_guessed(gr_program* p, void* stream, void* rdata, int width, const void* guessed, void* guessed_next, void* block_cost, void* attempt_counter) {
    if (!p || !rdata || !guessed || width < 2) return fail(GR_ERROR_INVALID_ARGUMENT, "gr_apply_guessed: null argument");
    void* args[] = {&rdata, &width, &guessed, &guessed_next, &block_cost, &attempt_counter};
    return launch(p, K_APPLY_GUESSED, stream, 65536 / 256, 1, 256, 1, args);
}

// gr_do_generic_rays over ray records in 8x8-tile slot order with the fused trace's scheduling (kernels/trace.hip): the device filled
// once, tiles drawn from a ticket counter in `tile_order`'s order (NULL: slot order), each tile's cost left in `tile_cost` (NULL: not).
int gr_do_generic_rays_scheduled(gr_program* p, void* stream, void* rays, const void* ray_count, int tile_count, const void* cfg, const void* dfg,
                                 void* attempt_counter, const void* tile_order, void* tile_cost) {
    if (!p || !rays || !ray_count) return fail(GR_ERROR_INVALID_ARGUMENT, "gr_do_generic_rays_scheduled: null argument");
    if (tile_count < 1) return GR_OK;
    long long groups = resident_trace_groups(p, K_DO_RAYS_SCHEDULED, 64);
    if (groups < 0) return (int)-groups;
    // experiment hook: GR_SCHEDULED_WAVES_PER_SIMD=k launches k persistent waves per SIMD whatever fits (occupancy studies)
    static const int forced = [] { const char* e = getenv("GR_SCHEDULED_WAVES_PER_SIMD"); int v = e ? atoi(e) : 0; return (v >= 1 && v <= 8) ? v : 0; }();
    if (forced) groups = (long long)p->compute_units * 4 * forced;
    groups = std::max(1LL, std::min(groups, (long long)tile_count));
    unsigned int* tickets = p->tickets + (p->next_ticket.fetch_add(1) % gr_program::TICKET_RING);
    HIP_CHECK(hipSetDevice(p->device));
    HIP_CHECK(hipMemsetAsync(tickets, 0, sizeof(unsigned int), (hipStream_t)stream));
    void* args[] = {&rays, &ray_count, &cfg, &dfg, &attempt_counter, &tickets, &tile_count, &tile_order, &tile_cost};
    return launch(p, K_DO_RAYS_SCHEDULED, stream, (unsigned)groups, 1, 64, 1, args);
}

int gr_sort_tiles_by_cost(gr_program* p, void* stream, const void* tile_cost, int tiles_x, int tiles_y, void* tile_order, void* work) {
    if (!p || !tile_cost || !tile_order || !work) return fail(GR_ERROR_INVALID_ARGUMENT, "gr_sort_tiles_by_cost: null argument");
    const int tile_count = tiles_x * tiles_y;
    if (tile_count < 1) return GR_OK;
    HIP_CHECK(hipSetDevice(p->device));
    HIP_CHECK(hipMemsetAsync(work, 0, 128 * sizeof(unsigned int), (hipStream_t)stream));
    void* count_args[] = {&tile_cost, &tiles_x, &tiles_y, &work};
    int rc = launch(p, K_SORT_TILES_COUNT, stream, blocks(tile_count, 256), 1, 256, 1, count_args);
    if (rc != GR_OK) return rc;
    int n = tile_count;
    void* place_args[] = {&n, &work, &tile_order};
    return launch(p, K_SORT_TILES_PLACE, stream, blocks(tile_count, 256), 1, 256, 1, place_args);
}

int gr_adaptive_refine(gr_program* p, void* stream, void* rdata, void* pending_count, int width, int height, const void* dfg,
                       const void* lattice_rays, const void* cfg) {
    return gr_adaptive_refine_strips(p, stream, rdata, pending_count, width, height, dfg, 0, 0, 1, lattice_rays, cfg);
}

int gr_trace_fused(gr_program* p, void* stream, const void* camera_generic, const void* camera_quat, void* rdata, int width,
                   int height, int block_rows, int strip_rank, int strip_count, const void* term, int prepass_width,
                   int prepass_height, const void* e0, const void* e1, const void* e2, const void* e3, const void* cfg,
                   const void* dfg, void* attempt_counter) {
    return trace_launch(p, 1, stream, camera_generic, camera_quat, rdata, width, height, block_rows, strip_rank, strip_count, term,
                        prepass_width, prepass_height, e0, e1, e2, e3, cfg, dfg, attempt_counter);
}

int gr_trace_fused_launch(gr_program* p, void* stream, const gr_trace_fused_args* a) {
    if (!a) return fail(GR_ERROR_INVALID_ARGUMENT, "null argument");
    return trace_launch(p, 1, stream, a->camera_generic, a->camera_quat, a->render_data, a->width, a->height, a->block_rows, a->strip_rank,
                        a->strip_count, a->termination_buffer, a->prepass_width, a->prepass_height, a->e0, a->e1, a->e2, a->e3, a->cfg, a->dfg,
                        a->attempt_counter, a->lattice == 2 ? 2 : 1, a->pending_only ? 1 : 0, a->tile_order, a->waves_per_simd, &a->shading,
                        a->inline_prepass ? 1 : 0, a->tile_cost, a->tile_order_by_history ? 1 : 0, a->lattice == 2 ? a->lattice_rays : nullptr, &a->parking, a->speculative_classes, a->lattice == 2 ? a->guessed : nullptr);
}

int gr_trace_pair(gr_program* p, void* stream, const void* camera_generic, const void* camera_quat, void* rdata, int width,
                  int height, int block_rows, int strip_rank, int strip_count, const void* term, int prepass_width,
                  int prepass_height, const void* e0, const void* e1, const void* e2, const void* e3, const void* cfg,
                  const void* dfg, void* attempt_counter) {
    return trace_launch(p, 2, stream, camera_generic, camera_quat, rdata, width, height, block_rows, strip_rank, strip_count, term,
                        prepass_width, prepass_height, e0, e1, e2, e3, cfg, dfg, attempt_counter);
}

int gr_program_has_trace_pair(const gr_program* p) { return p && p->fn[K_TRACE_PAIR] ? 1 : 0; }
int gr_program_has_tile_shading(const gr_program* p) { return p && p->tile_shading ? 1 : 0; }
int gr_program_has_parking(const gr_program* p) { return p && p->fn[K_TRACE_FUSED_PARKING] ? 1 : 0; }

int gr_trace_compact(gr_program* p, void* stream, const void* camera_generic, const void* camera_quat, void* rdata, int width,
                     int height, int block_rows, int strip_rank, int strip_count, const void* term, int prepass_width,
                     int prepass_height, const void* e0, const void* e1, const void* e2, const void* e3, const void* cfg,
                     const void* dfg, void* attempt_counter, int keep_lanes) {
    const int T = 8;
    if (keep_lanes < 1 || keep_lanes > 64) return fail(GR_ERROR_INVALID_ARGUMENT, "keep_lanes must be 1..64");
    if (strip_count <= 1) {
        strip_count = 1;
        strip_rank = 0;
        block_rows = ((height + T - 1) / T) * T;
    }
    if (block_rows <= 0 || block_rows % T != 0 || strip_rank < 0 || strip_rank >= strip_count)
        return fail(GR_ERROR_INVALID_ARGUMENT, "block_rows must be a positive multiple of 8 and 0 <= strip_rank < strip_count");
    if (strip_count > 1 && height > 1 && (height - 1) % block_rows == 0)
        return fail(GR_ERROR_INVALID_ARGUMENT, "the last image row must not start a block (its filter reads the row above)");
    int local_blocks = gr_strip_local_blocks(height, block_rows, strip_rank, strip_count);
    long long waves_per_block = (long long)((width + T - 1) / T) * (block_rows / T) + (strip_count > 1 ? (width + 63) / 64 : 0);
    long long waves = waves_per_block * local_blocks;
    if (waves <= 0) return GR_OK;
    if (waves * 64 > 0x7fffffffLL) return fail(GR_ERROR_INVALID_ARGUMENT, "too many ray slots");
    const int wg = 256;
    unsigned int total_slots = (unsigned int)(waves * 64);
    long long groups = std::min<long long>((waves * 64 + wg - 1) / wg, (long long)p->compute_units * 32 * 64 / wg);
    unsigned int* tickets = p->tickets + (p->next_ticket.fetch_add(1) % gr_program::TICKET_RING);
    HIP_CHECK(hipSetDevice(p->device));
    HIP_CHECK(hipMemsetAsync(tickets, 0, sizeof(unsigned int), (hipStream_t)stream));
    void* args[] = {&camera_generic, &camera_quat, &rdata, &width, &height, &block_rows, &strip_rank, &strip_count, &term,
                    &prepass_width, &prepass_height, &e0, &e1, &e2, &e3, &cfg, &dfg, &attempt_counter, &tickets, &total_slots, &keep_lanes};
    return launch(p, K_TRACE_COMPACT, stream, (unsigned)groups, 1, wg, 1, args);
}

// ---- camera on a timelike geodesic (cl.cl:2441-2481, 3117-3141, 4735-4940, 2569-2620, 2738-2872) --------------------

int gr_boost_tetrad(gr_program* p, void* stream, const void* generic_in, int count, const void* basis_speed, void* e0, void* e1,
                    void* e2, void* e3, const void* cfg) {
    void* args[] = {&generic_in, &count, &basis_speed, &e0, &e1, &e2, &e3, &cfg};
    return launch(p, K_BOOST_TETRAD, stream, blocks(count, 64), 1, 64, 1, args);
}

int gr_init_inertial_ray(gr_program* p, void* stream, const void* generic_position_in, int ray_count, void* rays, void* ray_count_out,
                         const void* e0, const void* e1, const void* e2, const void* e3, const void* basis_speed, const void* cfg) {
    void* args[] = {&generic_position_in, &ray_count, &rays, &ray_count_out, &e0, &e1, &e2, &e3, &basis_speed, &cfg};
    return launch(p, K_INIT_INERTIAL, stream, blocks(ray_count, 64), 1, 64, 1, args);
}

int gr_get_geodesic_path(gr_program* p, void* stream, const void* rays, int num_rays, void* positions_out, void* velocities_out,
                         void* ds_out, const void* ray_count, int max_path_length, const void* cfg, const void* dfg, void* count_out) {
    void* args[] = {&rays, &positions_out, &velocities_out, &ds_out, &ray_count, &max_path_length, &cfg, &dfg, &count_out};
    return launch(p, K_GEODESIC_PATH, stream, blocks(num_rays, 64), 1, 64, 1, args);
}

int gr_parallel_transport_quantity(gr_program* p, void* stream, const void* geodesic_path, const void* geodesic_velocity,
                                   const void* ds_in, const void* quantity, const void* count_in, int count, void* quantity_out,
                                   const void* cfg) {
    void* args[] = {&geodesic_path, &geodesic_velocity, &ds_in, &quantity, &count_in, &count, &quantity_out, &cfg};
    return launch(p, K_PARALLEL_TRANSPORT, stream, blocks(count, 64), 1, 64, 1, args);
}

int gr_handle_interpolating_geodesic(gr_program* p, void* stream, const void* geodesic_path, const void* geodesic_velocity,
                                     const void* ds_in, void* camera_generic_out, const void* t_e0, const void* t_e1,
                                     const void* t_e2, const void* t_e3, void* e0_out, void* e1_out, void* e2_out, void* e3_out,
                                     float target_time, const void* count_in, int parallel_transport_observer,
                                     const void* basis_speed, void* interpolated_velocity, const void* cfg) {
    void* args[] = {&geodesic_path, &geodesic_velocity, &ds_in, &camera_generic_out, &t_e0, &t_e1, &t_e2, &t_e3, &e0_out, &e1_out,
                    &e2_out, &e3_out, &target_time, &count_in, &parallel_transport_observer, &basis_speed, &interpolated_velocity,
                    &cfg};
    return launch(p, K_INTERPOLATE_GEODESIC, stream, 1, 1, 64, 1, args);
}

int gr_pack_mipped_background(const unsigned char* rgba, int width, int height, unsigned char* out) {
    if (width <= 0 || height <= 0) return fail(GR_ERROR_INVALID_ARGUMENT, "bad size");
    // graphics_settings.cpp:164-168
    int levels = (int)std::floor(std::log2((double)std::min(width, height))) + 1;
    if (levels > 10) levels = 10;
    if (!out) return levels;
    if (!rgba) return fail(GR_ERROR_INVALID_ARGUMENT, "null image");
    std::vector<float> cur((size_t)width * height * 4);
    for (size_t i = 0; i < cur.size(); i++) cur[i] = rgba[i] / 255.f;
    int cw = width, ch = height;
    for (int l = 0; l < levels; l++) {
        if (l > 0) {
            // 2x2 box filter (the reference takes the GL driver's mip chain)
            int nw = std::max(cw / 2, 1), nh = std::max(ch / 2, 1);
            std::vector<float> next((size_t)nw * nh * 4);
            for (int y = 0; y < nh; y++)
                for (int x = 0; x < nw; x++)
                    for (int c = 0; c < 4; c++) {
                        int x0 = std::min(2 * x, cw - 1), x1 = std::min(2 * x + 1, cw - 1);
                        int y0 = std::min(2 * y, ch - 1), y1 = std::min(2 * y + 1, ch - 1);
                        next[((size_t)y * nw + x) * 4 + c] =
                            0.25f * (cur[((size_t)y0 * cw + x0) * 4 + c] + cur[((size_t)y0 * cw + x1) * 4 + c] +
                                     cur[((size_t)y1 * cw + x0) * 4 + c] + cur[((size_t)y1 * cw + x1) * 4 + c]);
                    }
            cur.swap(next);
            cw = nw;
            ch = nh;
        }
        unsigned char* slice = out + (size_t)l * width * height * 4;
        for (int y = 0; y < height; y++)
            for (int x = 0; x < width; x++) {
                int lx = std::min(x, cw - 1), ly = std::min(y, ch - 1);   // edge replicate
                for (int c = 0; c < 4; c++) {
                    float v = cur[((size_t)ly * cw + lx) * 4 + c];
                    v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
                    slice[((size_t)y * width + x) * 4 + c] = (unsigned char)(v * 255);
                }
            }
    }
    return levels;
}

}  // extern "C"
