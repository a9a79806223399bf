// codeobject.cpp — see codeobject.hpp.  The code-object manager and hiprtc's built-in header are loaded at run time (hiprtc
// itself does the same: libhiprtc.so has no link-time dependency on either), so the library still loads where they are absent.
#include "codeobject.hpp"

#include <amd_comgr/amd_comgr.h>
#include <dlfcn.h>
#include <link.h>
#include <hip/hip_version.h>

#include <cstring>
#include <mutex>

namespace gr {
namespace {

struct comgr_api {
    void* lib = nullptr;
    const char* header = nullptr;   // hiprtc_runtime.h: what hiprtc puts in front of every program
    unsigned header_size = 0;
#define GR_COMGR_FN(name) decltype(&amd_comgr_##name) name = nullptr
    GR_COMGR_FN(status_string);
    GR_COMGR_FN(create_data);
    GR_COMGR_FN(release_data);
    GR_COMGR_FN(set_data);
    GR_COMGR_FN(set_data_name);
    GR_COMGR_FN(get_data);
    GR_COMGR_FN(create_data_set);
    GR_COMGR_FN(destroy_data_set);
    GR_COMGR_FN(data_set_add);
    GR_COMGR_FN(create_action_info);
    GR_COMGR_FN(destroy_action_info);
    GR_COMGR_FN(action_info_set_isa_name);
    GR_COMGR_FN(action_info_set_language);
    GR_COMGR_FN(action_info_set_option_list);
    GR_COMGR_FN(do_action);
    GR_COMGR_FN(action_data_count);
    GR_COMGR_FN(action_data_get_data);
#undef GR_COMGR_FN
    std::string error;
};

const comgr_api& comgr() {
    static comgr_api api;
    static std::once_flag once;
    std::call_once(once, [] {
        // hiprtc's built-in header first, then the code-object manager FROM THE SAME DIRECTORY, by path and bound to itself: a
        // process may hold a second ROCm stack (PyTorch wheels bundle their own libamd_comgr.so.3, an older one, which a load by
        // soname would hand back - and which neither understands this header nor encodes every gfx950 instruction the kernels
        // compile to).  The two copies do not share symbols (RTLD_LOCAL | RTLD_DEEPBIND).
        void* builtins = nullptr;
        for (const char* name : {"libhiprtc-builtins.so.7", "libhiprtc-builtins.so"})
            if ((builtins = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
        if (builtins) {
            api.header = (const char*)dlsym(builtins, "__hipRTC_header");
            const unsigned* size = (const unsigned*)dlsym(builtins, "__hipRTC_header_size");
            if (size) api.header_size = *size;
        }
        if (!api.header || !api.header_size) { api.error = "libhiprtc-builtins.so (hiprtc's built-in header) not loadable"; return; }
        char origin[4096] = "";
        if (dlinfo(builtins, RTLD_DI_ORIGIN, origin) == 0 && origin[0])
            for (const char* name : {"/libamd_comgr.so.3", "/libamd_comgr.so"})
                if ((api.lib = dlopen((std::string(origin) + name).c_str(), RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND))) break;
        if (!api.lib) { api.error = std::string("libamd_comgr.so not loadable from ") + origin; return; }
        bool ok = true;
#define GR_COMGR_FN(name) ok &= (api.name = (decltype(api.name))dlsym(api.lib, "amd_comgr_" #name)) != nullptr
        GR_COMGR_FN(status_string); GR_COMGR_FN(create_data); GR_COMGR_FN(release_data); GR_COMGR_FN(set_data);
        GR_COMGR_FN(set_data_name); GR_COMGR_FN(get_data); GR_COMGR_FN(create_data_set); GR_COMGR_FN(destroy_data_set);
        GR_COMGR_FN(data_set_add); GR_COMGR_FN(create_action_info); GR_COMGR_FN(destroy_action_info);
        GR_COMGR_FN(action_info_set_isa_name); GR_COMGR_FN(action_info_set_language); GR_COMGR_FN(action_info_set_option_list);
        GR_COMGR_FN(do_action); GR_COMGR_FN(action_data_count); GR_COMGR_FN(action_data_get_data);
#undef GR_COMGR_FN
        if (!ok) { api.error = "libamd_comgr.so lacks an expected entry point"; api.lib = nullptr; return; }
    });
    return api;
}

// owns the handles of one comgr action chain
struct scope {
    const comgr_api& c;
    std::vector<amd_comgr_data_t> data;
    std::vector<amd_comgr_data_set_t> sets;
    amd_comgr_action_info_t info{};
    bool has_info = false;
    explicit scope(const comgr_api& api) : c(api) {}
    ~scope() {
        for (auto d : data) c.release_data(d);
        for (auto s : sets) c.destroy_data_set(s);
        if (has_info) c.destroy_action_info(info);
    }
    bool add(amd_comgr_data_set_t set, amd_comgr_data_kind_t kind, const char* name, const char* bytes, size_t size) {
        amd_comgr_data_t d;
        if (c.create_data(kind, &d) != AMD_COMGR_STATUS_SUCCESS) return false;
        data.push_back(d);
        return c.set_data(d, size, bytes) == AMD_COMGR_STATUS_SUCCESS && c.set_data_name(d, name) == AMD_COMGR_STATUS_SUCCESS &&
               c.data_set_add(set, d) == AMD_COMGR_STATUS_SUCCESS;
    }
    bool new_set(amd_comgr_data_set_t& set) {
        if (c.create_data_set(&set) != AMD_COMGR_STATUS_SUCCESS) return false;
        sets.push_back(set);
        return true;
    }
    bool first_of(amd_comgr_data_set_t set, amd_comgr_data_kind_t kind, std::string& out) {
        size_t n = 0;
        if (c.action_data_count(set, kind, &n) != AMD_COMGR_STATUS_SUCCESS || n == 0) return false;
        amd_comgr_data_t d;
        if (c.action_data_get_data(set, kind, 0, &d) != AMD_COMGR_STATUS_SUCCESS) return false;
        data.push_back(d);
        size_t size = 0;
        if (c.get_data(d, &size, nullptr) != AMD_COMGR_STATUS_SUCCESS) return false;
        out.assign(size, '\0');
        return size == 0 || c.get_data(d, &size, &out[0]) == AMD_COMGR_STATUS_SUCCESS;
    }
    // runs one action; on failure the log of the action (compiler diagnostics) goes to `log`
    bool run(amd_comgr_action_kind_t kind, amd_comgr_data_set_t in, amd_comgr_data_set_t out, const char* what, std::string& log) {
        amd_comgr_status_t s = c.do_action(kind, info, in, out);
        if (s == AMD_COMGR_STATUS_SUCCESS) return true;
        std::string text;
        first_of(out, AMD_COMGR_DATA_KIND_LOG, text);
        const char* msg = "";
        c.status_string(s, &msg);
        log = std::string(what) + ": " + msg + "\n" + text;
        return false;
    }
};

#define GR_STR2(x) #x
#define GR_STR(x) GR_STR2(x)

}  // namespace

bool compile_to_assembly(const std::string& source, const std::vector<std::string>& options, std::string& assembly, std::string& log) {
    // GR_NO_CODE_OBJECT_MANAGER=1: behave as where libamd_comgr cannot be loaded (tests of the hiprtc fallback of both modules)
    if (const char* e = getenv("GR_NO_CODE_OBJECT_MANAGER"); e && e[0] == '1') { log = "code-object manager switched off (GR_NO_CODE_OBJECT_MANAGER)"; return false; }
    const comgr_api& c = comgr();
    if (!c.lib) { log = c.error; return false; }
    scope sc(c);
    amd_comgr_data_set_t in, out;
    if (!sc.new_set(in) || !sc.new_set(out)) { log = "comgr: data set"; return false; }
    if (!sc.add(in, AMD_COMGR_DATA_KIND_SOURCE, "geodesic_kernels_all_parts.hip", source.data(), source.size()) ||
        !sc.add(in, AMD_COMGR_DATA_KIND_INCLUDE, "hiprtc_runtime.h", c.header, c.header_size)) { log = "comgr: inputs"; return false; }
    if (c.create_action_info(&sc.info) != AMD_COMGR_STATUS_SUCCESS) { log = "comgr: action info"; return false; }
    sc.has_info = true;
    // what hiprtc 7.x passes for a HIP program (AMD_COMGR_EMIT_VERBOSE_LOGS=1 shows it), then the caller's options, then -S: the
    // "relocatable" this action returns is then the assembly text of the same compilation
    std::vector<std::string> all = {"-O3", "--hip-version=" GR_STR(HIP_VERSION_MAJOR) "." GR_STR(HIP_VERSION_MINOR) "." GR_STR(HIP_VERSION_PATCH),
                                    "-DHIP_VERSION_MAJOR=" GR_STR(HIP_VERSION_MAJOR), "-DHIP_VERSION_MINOR=" GR_STR(HIP_VERSION_MINOR),
                                    "-DHIP_VERSION_PATCH=" GR_STR(HIP_VERSION_PATCH), "-D__HIPCC_RTC__", "-Wno-gnu-line-marker",
                                    "-Wno-missing-prototypes", "-nogpuinc", "-include", "hiprtc_runtime.h"};
    for (auto& o : options) all.push_back(o);
    all.push_back("-w");
    all.push_back("-S");
    std::vector<const char*> raw;
    for (auto& o : all) raw.push_back(o.c_str());
    if (c.action_info_set_isa_name(sc.info, "amdgcn-amd-amdhsa--gfx950") != AMD_COMGR_STATUS_SUCCESS ||
        c.action_info_set_language(sc.info, AMD_COMGR_LANGUAGE_HIP) != AMD_COMGR_STATUS_SUCCESS ||
        c.action_info_set_option_list(sc.info, raw.data(), raw.size()) != AMD_COMGR_STATUS_SUCCESS) { log = "comgr: options"; return false; }
    if (!sc.run(AMD_COMGR_ACTION_COMPILE_SOURCE_TO_RELOCATABLE, in, out, "compile", log)) return false;
    if (!sc.first_of(out, AMD_COMGR_DATA_KIND_RELOCATABLE, assembly) || assembly.find(".amdgcn_target") == std::string::npos) {
        log = "comgr: the compile action returned no assembly text";
        return false;
    }
    return true;
}

bool assemble_code_object(const std::string& assembly, std::string& code, std::string& log) {
    const comgr_api& c = comgr();
    if (!c.lib) { log = c.error; return false; }
    scope sc(c);
    amd_comgr_data_set_t in, rel, exe;
    if (!sc.new_set(in) || !sc.new_set(rel) || !sc.new_set(exe)) { log = "comgr: data set"; return false; }
    if (!sc.add(in, AMD_COMGR_DATA_KIND_SOURCE, "geodesic_kernels.s", assembly.data(), assembly.size())) { log = "comgr: inputs"; return false; }
    if (c.create_action_info(&sc.info) != AMD_COMGR_STATUS_SUCCESS) { log = "comgr: action info"; return false; }
    sc.has_info = true;
    if (c.action_info_set_isa_name(sc.info, "amdgcn-amd-amdhsa--gfx950") != AMD_COMGR_STATUS_SUCCESS) { log = "comgr: isa"; return false; }
    if (!sc.run(AMD_COMGR_ACTION_ASSEMBLE_SOURCE_TO_RELOCATABLE, in, rel, "assemble", log)) return false;
    if (!sc.run(AMD_COMGR_ACTION_LINK_RELOCATABLE_TO_EXECUTABLE, rel, exe, "link", log)) return false;
    if (!sc.first_of(exe, AMD_COMGR_DATA_KIND_EXECUTABLE, code) || code.empty()) { log = "comgr: no executable"; return false; }
    return true;
}

vector_run_stats break_vector_runs(std::string& assembly, int limit, const std::vector<std::string>& only_functions) {
    vector_run_stats stats;
    if (limit <= 0) return stats;
    // line starts
    std::vector<size_t> starts;
    for (size_t at = 0; at < assembly.size();) {
        starts.push_back(at);
        size_t nl = assembly.find('\n', at);
        at = nl == std::string::npos ? assembly.size() : nl + 1;
    }
    std::vector<size_t> insert_before;   // line indices
    std::vector<size_t> run;             // line indices of the vector instructions of the run being read
    bool in_scope = only_functions.empty();
    auto close_run = [&]() {
        const int L = (int)run.size();
        if (L > stats.longest_before) stats.longest_before = L;
        int longest_piece = L;
        if (L > limit && in_scope) {
            const int pieces = (L + limit - 1) / limit;
            for (int i = 1; i < pieces; i++) insert_before.push_back(run[(size_t)((long long)i * L / pieces)]);
            stats.runs_broken++;
            longest_piece = (L + pieces - 1) / pieces;
        }
        if (longest_piece > stats.longest_after) stats.longest_after = longest_piece;
        run.clear();
    };
    for (size_t li = 0; li < starts.size(); li++) {
        const char* p = assembly.c_str() + starts[li];
        const char* end = assembly.c_str() + (li + 1 < starts.size() ? starts[li + 1] : assembly.size());
        if (p < end && *p != '\t' && *p != ' ' && *p != '\n' && *p != ';' && *p != '.') {
            // a symbol label at column 0 ("gr_trace_fused:"): a new function; local labels (.LBB7_3:) start with '.'
            close_run();
            const char* colon = (const char*)memchr(p, ':', (size_t)(end - p));
            if (colon && !only_functions.empty()) {
                const std::string name(p, colon);
                in_scope = false;
                for (auto& f : only_functions) in_scope |= f == name;
            }
            continue;
        }
        if (p < end && *p == '.') {   // .LBB7_3: - a basic block starts; or a directive at column 0
            close_run();
            continue;
        }
        while (p < end && (*p == '\t' || *p == ' ')) p++;
        if (p >= end || *p == '\n' || *p == ';' || *p == '.') continue;   // blank, comment, directive: not an instruction
        if (end - p > 2 && p[0] == 'v' && p[1] == '_') run.push_back(li);
        else if (end - p > 2 && p[0] == 's' && p[1] == '_') close_run();
        // memory instructions (global_, scratch_, ds_, buffer_) neither extend nor end a run: none in the loops this is for
    }
    close_run();
    if (insert_before.empty()) return stats;
    std::string out;
    out.reserve(assembly.size() + insert_before.size() * 10);
    size_t next = 0;
    for (size_t li = 0; li < starts.size(); li++) {
        if (next < insert_before.size() && insert_before[next] == li) {
            out += "\ts_nop 0\n";
            next++;
        }
        const size_t end = li + 1 < starts.size() ? starts[li + 1] : assembly.size();
        out.append(assembly, starts[li], end - starts[li]);
    }
    stats.inserted = (int)insert_before.size();
    assembly.swap(out);
    return stats;
}

}  // namespace gr
