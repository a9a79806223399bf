"""Compiler-flag sweep on the bench workload: each variant is the substituted Kerr program built with extra flags (GR_EXTRA_FLAGS, part
of the cache key).   build container: python tools/flag_sweep.py build     GPU box: python tools/flag_sweep.py run"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CACHE = os.path.join(ROOT, "tools", "_variants", "flags")
VARIANTS = {
    "default": "",
    "relaxed_occupancy": "-mllvm -amdgpu-schedule-relaxed-occupancy=1",
    "no_misched": "-mllvm -enable-misched=0",
    "no_post_misched": "-mllvm -enable-post-misched=0",
    "O2": "-O2",
    "no_unroll": "-fno-unroll-loops",
    "no_licm_hoist": "-mllvm -disable-licm-promotion",
    "no_machine_sink": "-mllvm -disable-machine-sink",
    "no_machine_licm": "-mllvm -disable-machine-licm",
    "no_machine_licm_w7": "-mllvm -disable-machine-licm -DGR_FUSED_WAVES=7",
    "no_machine_licm_w6": "-mllvm -disable-machine-licm -DGR_FUSED_WAVES=6",
    "postra_misched": "-mllvm -misched-postra",
    "no_sink_no_licm": "-mllvm -disable-machine-sink -mllvm -disable-machine-licm",
}


def env_of(name):
    return dict(os.environ, GR_CACHE_DIR=os.path.join(CACHE, name), GR_EXTRA_FLAGS=VARIANTS[name])


BUILD = r'''
import sys; sys.path.insert(0, %r)
import geodesic_raytracing_amd as gra
m = gra.Metric("kerr_boyer", %r)
gra.Program.precompile(m.argument_string())
gra.Program.precompile(m.argument_string(features=m.features(adaptive_sampling=0), static=True, cfg_values=m.cfg_values(a=0.45)))
''' % (ROOT, os.path.join(ROOT, "geodesic_raytracing_amd", "scripts"))

if sys.argv[1] == "build":
    procs = []
    for name in VARIANTS:
        os.makedirs(env_of(name)["GR_CACHE_DIR"], exist_ok=True)
        procs.append((name, subprocess.Popen([sys.executable, "-c", BUILD], env=dict(env_of(name), GR_VERBOSE_BUILD="1"), stderr=subprocess.PIPE, text=True)))
    for name, proc in procs:
        err = proc.communicate()[1]
        kept = [l for l in err.splitlines() if "gr_trace_fused" in l]
        print(name, "rc", proc.returncode, kept[-1][5:] if kept else err[-300:], flush=True)
else:
    for name in VARIANTS:
        row = {"variant": name}
        for tag, mode in [("pipelined", []), ("alone", ["--frames-in-flight", "1", "--no-lookahead"])]:
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-secondary", "--steps", "40", "--warmup", "5"] + mode,
                                 env=env_of(name), capture_output=True, text=True)
            try:
                j = json.loads(out.stdout.strip().splitlines()[-1])
                row[tag] = round(j["value"], 1)
                row["kernel"] = j["config"]["build_key"].split("-")[-1]
            except Exception:
                row[tag] = (out.stderr or out.stdout)[-200:]
        print(json.dumps(row), flush=True)
