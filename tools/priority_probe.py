"""Issue priority of long waves (-DGR_PRIORITY_TRIPS=n, kernels/integrator.hip): the bench workloads one frame at a time and with
frames in flight, adaptive sampling, per variant.   build container: python tools/priority_probe.py build     GPU box: ... run"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CACHE = os.path.join(ROOT, "geodesic_raytracing_amd", "_cache_variants")
VARIANTS = [int(x) for x in os.environ.get("PRIORITY_TRIPS", "0,64,128,256,512").split(",")]


def env_of(n):
    return dict(os.environ, GR_CACHE_DIR=os.path.join(CACHE, "prio%d" % n), GR_EXTRA_FLAGS=("-DGR_PRIORITY_TRIPS=%d" % n) if n else "")


BUILD = r'''
import sys; sys.path.insert(0, %r)
import geodesic_raytracing_amd as gra
m = gra.Metric("kerr_boyer", %r)
gra.Program.precompile(m.argument_string())
for a in (0.45, 0.9):
    gra.Program.precompile(m.argument_string(features=m.features(adaptive_sampling=0), static=True, cfg_values=m.cfg_values(a=a)))
gra.Program.precompile(m.argument_string(features=m.features(adaptive_sampling=1, adaptive_sampling_threshold=32.0), static=True, cfg_values=m.cfg_values(a=0.45)))
''' % (ROOT, os.path.join(ROOT, "geodesic_raytracing_amd", "scripts"))

if sys.argv[1] == "build":
    procs = []
    for n in VARIANTS:
        os.makedirs(env_of(n)["GR_CACHE_DIR"], exist_ok=True)
        procs.append((n, subprocess.Popen([sys.executable, "-c", BUILD], env=dict(env_of(n), GR_VERBOSE_BUILD="1"), stderr=subprocess.PIPE, text=True)))
    for n, proc in procs:
        err = proc.communicate()[1]
        kept = [l for l in err.splitlines() if "gr_trace_fused" in l]
        print(n, "rc", proc.returncode, kept[-1][5:] if kept else err[-300:], flush=True)
else:
    for n in VARIANTS:
        row = {"priority_trips": n}
        for spin in ("0.45", "0.9"):
            for tag, mode in [("in_flight", []), ("alone", ["--frames-in-flight", "1", "--no-lookahead"])]:
                out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--spin", spin, "--no-cpu-baseline", "--no-secondary", "--steps", "30", "--warmup", "5"] + mode,
                                     env=env_of(n), capture_output=True, text=True)
                try:
                    j = json.loads(out.stdout.strip().splitlines()[-1])
                    row["a%s_%s_ms" % (spin, tag)] = round(j["ms_per_step"], 3)
                except Exception:
                    row["a%s_%s_ms" % (spin, tag)] = (out.stderr or out.stdout)[-200:]
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "adaptive_fused_probe.py")], env=env_of(n), capture_output=True, text=True)
        row["adaptive"] = [l[:60] for l in out.stdout.splitlines() if l.startswith("adaptive=1") or "adaptive=1, trace_waves_per_simd=4" in l]
        print(json.dumps(row), flush=True)
