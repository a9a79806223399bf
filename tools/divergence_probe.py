"""Step-attempt throughput of the fused trace kernel for several workloads: a low Gattempts/s relative to the best case
means lanes of a wave idle while the longest ray of the tile finishes (divergence), not slower arithmetic.
usage: PYTHONPATH=. python tools/divergence_probe.py"""
import json
import os
import sys

import numpy as np

import geodesic_raytracing_amd as gra

SCRIPTS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "geodesic_raytracing_amd", "scripts")
CASES = [("kerr_boyer a=0.45", "kerr_boyer", dict(a=0.45), (0, 0, -4, 0)), ("kerr_boyer a=0.9", "kerr_boyer", dict(a=0.9), (0, 0, -4, 0)),
         ("kerr_boyer a=0.45 far", "kerr_boyer", dict(a=0.45), (0, 0, -15, 0)), ("schwarzschild", "schwarzschild", {}, (0, 0, -4, 0)),
         ("alcubierre", "alcubierre", {}, (0, 0, -6, 0.5)), ("double_unequal_kerr", "double_unequal_kerr", {}, (0, 0, -6, 0.5))]
w, h = 3840, 2160
for label, name, cfg, pos in CASES:
    m = gra.Metric(name, SCRIPTS)
    feats = m.features(adaptive_sampling=0)
    cv = m.cfg_values(**cfg)
    prog = gra.Program(m.argument_string(features=feats, static=True, cfg_values=cv), 0)
    st = gra.RenderState(w, h, 0)
    opts = gra.frame_options(mode=gra.MODE_FUSED, time_kernels=1, count_attempts=1)
    ts = []
    for i in range(4):
        st.render(prog, m, gra.default_camera(position=pos), None, None, feats, cv, opts)
        st.synchronize()
        ts.append(st.stage_ms())
    tr = float(np.median([t["trace"] for t in ts[1:]]))
    att = st.attempts()
    from geodesic_raytracing_amd.pipeline import download, RENDER_DATA_DTYPE
    rd = download(0, st.buffer(gra.BUF_RENDER_DATA), RENDER_DATA_DTYPE, w * h)
    print(json.dumps({"case": label, "trace_ms": round(tr, 3), "attempts": att, "Gattempts_per_s": round(att / tr / 1e6, 1),
                      "attempts_per_pixel": round(att / (w * h), 1), "prepass_ms": round(ts[-1]["prepass"], 3),
                      "vgpr": prog.kernel_info("gr_trace_fused")}), flush=True)
