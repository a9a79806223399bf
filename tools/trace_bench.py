"""Times the ray-trace stage for kernel build variants (GR_EXTRA_FLAGS) and reports parity against the golden vectors.
usage: python tools/trace_bench.py "<flags variant 1>" "<flags variant 2>" ...   ("" = default build)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, os, json
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import geodesic_raytracing_amd as gra
from geodesic_raytracing_amd.pipeline import DeviceBuffer
from gpu_stages import Stages, load_golden, rel_err
res = {"flags": os.environ.get("GR_EXTRA_FLAGS", "")}
for name, w, h, cfg in [("kerr_boyer", 3840, 2160, {"a": 0.45}), ("schwarzschild", 1920, 1080, {}), ("alcubierre", 1920, 1080, {})]:
    m = gra.Metric(name); prog = gra.Program(m.argument_string(), 0); st = gra.RenderState(w, h, 0)
    feats = m.features(adaptive_sampling=0, redshift=int(name == "alcubierre"))
    opts = gra.frame_options(mode=gra.MODE_FUSED, time_kernels=1, count_attempts=1)
    ts = []
    for i in range(6):
        st.render(prog, m, gra.default_camera(), None, None, feats, m.cfg_values(**cfg), opts); st.synchronize()
        ts.append(st.stage_ms())
    tr = float(np.median([t["trace"] for t in ts[1:]])); pp = float(np.median([t["prepass"] for t in ts[1:]]))
    att = st.attempts()
    res[name] = {"trace_ms": round(tr, 3), "prepass_ms": round(pp, 3), "Gattempts_per_s": round(att / tr / 1e6, 2), "attempts": att,
                 "vgpr": prog.kernel_info("gr_trace_fused")}
# parity of the variant
par = {}
for case in ["kerr", "kerr_tilted", "schwarzschild", "alcubierre"]:
    meta, z = load_golden(case); s = Stages(meta)
    r = s.trace(z["rays_init"])
    both = (r["terminated"] == 1) & (z["rays"]["terminated"] == 1)
    e = rel_err(r["position"][both], z["rays"]["position"][both]).max(axis=1)
    par[case] = {"term_mismatch": float((r["terminated"] != z["rays"]["terminated"]).mean()), "pos_p90": float(np.percentile(e, 90)), "pos_p99": float(np.percentile(e, 99))}
res["parity"] = par
print(json.dumps(res))
'''

for flags in (sys.argv[1:] or [""]):
    env = dict(os.environ, GR_EXTRA_FLAGS=flags, GR_CACHE_DIR="/tmp/gr_cache_variants")
    out = subprocess.run([sys.executable, "-c", "ROOT=%r\n" % ROOT + CHILD], env=env, capture_output=True, text=True)
    print(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-2000:], flush=True)
