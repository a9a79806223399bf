"""Sweep of the assembly pass (GR_VECTOR_RUN_LIMIT; 0 = plain hiprtc build) on the bench workload (profiles/r03_run_limit_sweep.txt).
  build container:  python tools/run_limit_sweep.py build      -> code objects of every variant under tools/_variants/cache
  GPU box:          python tools/run_limit_sweep.py run [args] -> one line per variant: pipelined Mrays/s, one-frame-at-a-time ms
"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CACHE = os.path.join(ROOT, "tools", "_variants", "cache")
VARIANTS = [4, 5, 6, 7, 8, 9, 10, 12, 16]


def env_of(limit):
    return dict(os.environ, GR_CACHE_DIR=CACHE, GR_VECTOR_RUN_LIMIT=str(limit))


BUILD = r'''
import sys; sys.path.insert(0, %r)
import geodesic_raytracing_amd as gra
scripts = %r
spins = [float(x) for x in sys.argv[1:]] or [0.45]
m = gra.Metric("kerr_boyer", scripts)
gra.Program.precompile(m.argument_string())
for a in spins:
    gra.Program.precompile(m.argument_string(features=m.features(adaptive_sampling=0), static=True, cfg_values=m.cfg_values(a=a)))
''' % (ROOT, os.path.join(ROOT, "geodesic_raytracing_amd", "scripts"))

if sys.argv[1] == "build":
    os.makedirs(CACHE, exist_ok=True)
    for l in VARIANTS:
        subprocess.check_call([sys.executable, "-c", BUILD] + sys.argv[2:], env=env_of(l))
        print("built", l, flush=True)
else:
    extra = sys.argv[2:]
    for l in VARIANTS:
        row = {"run_limit": l}
        for tag, mode in [("pipelined", []), ("alone", ["--frames-in-flight", "1", "--no-lookahead"])]:
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-secondary", "--steps", "40", "--warmup", "5"] + mode + extra,
                                 env=env_of(l), capture_output=True, text=True)
            try:
                j = json.loads(out.stdout.strip().splitlines()[-1])
                row[tag] = {"Mrays_s": round(j["value"], 1), "ms": round(j["ms_per_step"], 3), "trace_ms": j.get("roofline", {}).get("avg_launch_ms")}
            except Exception:
                row[tag] = (out.stderr or out.stdout)[-400:]
        print(json.dumps(row), flush=True)
