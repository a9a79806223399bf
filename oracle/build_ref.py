"""TEST INFRASTRUCTURE ONLY.  Builds oracle/_ref/libref_<metric>.so: the reference's own device source
(/root/reference/cl.cl) compiled as OpenCL C for x86-64 and linked with oracle/ref_shim.cpp.

Only works where /root/reference exists (the build container); the products stay under oracle/_ref/
(git-ignored) and no reference source is copied.  The `-D` macro set comes from this repository's own
code generator (libgeodesic_hip.so, "dynamic" flavour: cfg->NAME + feature struct) because the
reference's generator depends on un-vendored submodules (deps/vec, QuickJS) - see DESIGN.md.

Usage: python oracle/build_ref.py [metric ...]     (default: the five BASELINE config metrics)
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
OUT = os.path.join(HERE, "_ref")
CLANG = "/opt/rocm/lib/llvm/bin/clang"
DEFAULT_METRICS = ["minkowski", "schwarzschild", "kerr_boyer", "alcubierre"]


def reference_available():
    return os.path.exists(os.path.join(REF, "cl.cl")) and os.path.exists(CLANG)


def lib_path(tag):
    return os.path.join(OUT, f"libref_{tag}.so")


def _response_text(argument_string):
    return "".join('"' + tok.replace("\\", "\\\\").replace('"', '\\"') + '"\n' for tok in argument_string.split())


def prebuilt(tag, argument_string):
    """path of an oracle/_ref library built earlier (in the build container) for exactly this macro string, else None;
    needs neither the reference sources nor clang, so it works on the GPU box"""
    so, rsp = lib_path(tag), os.path.join(OUT, f"{tag}.rsp")
    if os.path.exists(so) and os.path.exists(rsp) and open(rsp).read() == _response_text(argument_string):
        return so
    return None


def build(tag, argument_string, force=False):
    """Compile cl.cl with `argument_string` and link the shim; returns the .so path."""
    if not reference_available():
        raise RuntimeError("reference sources or clang not available")
    os.makedirs(OUT, exist_ok=True)
    rsp = os.path.join(OUT, f"{tag}.rsp")
    text = _response_text(argument_string)
    so = lib_path(tag)
    shim = os.path.join(HERE, "ref_shim.cpp")
    if (not force and os.path.exists(so) and os.path.exists(rsp) and open(rsp).read() == text
            and os.path.getmtime(so) > os.path.getmtime(shim)):
        return so
    with open(rsp, "w") as f:
        f.write(text)
    obj = os.path.join(OUT, f"cl_{tag}.o")
    subprocess.check_call([CLANG, "-x", "cl", "-cl-std=CL1.2", "-Xclang", "-finclude-default-header", "-target",
                           "x86_64-unknown-linux-gnu", "-O2", "-fPIC", "-w", "-I", REF, "-cl-unsafe-math-optimizations",
                           "@" + rsp, "-c", os.path.join(REF, "cl.cl"), "-o", obj])
    shim_obj = os.path.join(OUT, "ref_shim.o")
    if not os.path.exists(shim_obj) or os.path.getmtime(shim_obj) < os.path.getmtime(shim):
        subprocess.check_call([CLANG + "++", "-std=c++17", "-O2", "-fPIC", "-c", shim, "-o", shim_obj])
    subprocess.check_call([CLANG + "++", "-shared", "-o", so, obj, shim_obj, "-lm", "-lpthread"])
    return so


def build_metric(name, scripts_dir=None, **kw):
    sys.path.insert(0, ROOT)
    import geodesic_raytracing_amd as gra
    m = gra.Metric(name, scripts_dir)
    return build(name, m.argument_string(), **kw)


if __name__ == "__main__":
    names = sys.argv[1:] or DEFAULT_METRICS
    for n in names:
        print(n, "->", build_metric(n))
