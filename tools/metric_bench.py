"""Times the fused trace of one script metric at 4K for kernel build variants (GR_EXTRA_FLAGS).
usage: python tools/metric_bench.py <metric> "<flags>" ..."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, json
sys.path.insert(0, ROOT)
import numpy as np
import geodesic_raytracing_amd as gra
name = os.environ["GR_BENCH_METRIC"]
m = gra.Metric(name, os.path.join(ROOT, "geodesic_raytracing_amd", "scripts"))
cfg = m.cfg_values(a=0.45) if name == "kerr_boyer" else m.cfg_values()
feats = m.features(adaptive_sampling=0)
static = bool(int(os.environ.get("GR_BENCH_STATIC", "0")))
prog = gra.Program(m.argument_string(features=feats, static=static, cfg_values=cfg), 0); st = gra.RenderState(3840, 2160, 0)
opts = gra.frame_options(mode=gra.MODE_FUSED, time_kernels=1, count_attempts=1)
cam = gra.default_camera([0, 0, -6, 0.5]) if name == "double_unequal_kerr" else gra.default_camera()
ts = []
for i in range(4):
    st.render(prog, m, cam, None, None, feats, cfg, opts); st.synchronize(); ts.append(st.stage_ms())
tr = float(np.median([t["trace"] for t in ts[1:]])); pp = float(np.median([t["prepass"] for t in ts[1:]]))
print(json.dumps({"metric": name, "flags": os.environ.get("GR_EXTRA_FLAGS", ""), "static": static, "trace_ms": round(tr, 3), "prepass_ms": round(pp, 3),
                  "attempts_per_ray": st.attempts() / (3840 * 2160), "Gatt_s": round(st.attempts() / tr / 1e6, 2), "regs": prog.kernel_info("gr_trace_fused")}))
'''
name = sys.argv[1]
for flags in (sys.argv[2:] or [""]):
    env = dict(os.environ, GR_EXTRA_FLAGS=flags, GR_CACHE_DIR="/tmp/gr_cache_variants", GR_BENCH_METRIC=name)
    out = subprocess.run([sys.executable, "-c", "ROOT=%r\n" % ROOT + CHILD], env=env, capture_output=True, text=True)
    print(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-1500:], flush=True)
