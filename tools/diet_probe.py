"""Instruction diet probe (GPU): the substituted 4K Kerr trace launch on its own - time, attempts - and, when run under
`rocprofv3 --pmc`, nothing else in the process, so that the counters divide cleanly by the attempts it prints.
usage: python tools/diet_probe.py [metric] [frames]      env GR_EXTRA_FLAGS / GR_CACHE_DIR as usual"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import geodesic_raytracing_amd as gra

name = sys.argv[1] if len(sys.argv) > 1 else "kerr_boyer"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cfgs = {"kerr_boyer": dict(a=0.45), "kerr_boyer_a09": dict(a=0.9)}
scripts = os.path.join(ROOT, "geodesic_raytracing_amd", "scripts")
m = gra.Metric(name.replace("_a09", ""), scripts)
cfg = m.cfg_values(**cfgs.get(name, {}))
w, h = (7680, 4320) if name == "alcubierre" else (3840, 2160)
feats = m.features(adaptive_sampling=0)
prog = gra.Program(m.argument_string(features=feats, static=True, cfg_values=cfg), 0)
st = gra.RenderState(w, h, 0)
opts = gra.frame_options(mode=gra.MODE_FUSED, time_kernels=1, count_attempts=1)
ts = []
for i in range(frames):
    st.render(prog, m, gra.default_camera(), None, None, feats, cfg, opts)
    st.synchronize()
    ts.append(st.stage_ms())
att = st.attempts()
clock = st.shader_clock_mhz()
tr = float(np.median([t["trace"] for t in ts[1:]]))
print(json.dumps({"metric": name, "trace_ms": round(tr, 4), "attempts": att, "Gattempts_per_s": round(att / tr / 1e6, 2),
                  "frames": frames, "shader_clock_mhz": round(clock, 1), "regs": prog.kernel_info("gr_trace_fused"), "flags": os.environ.get("GR_EXTRA_FLAGS", "")}))
