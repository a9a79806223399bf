"""The slowest shipped scripts at 1920x1080 with the GUI's defaults (adaptive sampling on, threshold 32), one frame at a time: stage times with the
prepass as a launch of its own / inside the lattice launch / taken from the previous frame (still camera), and with every pixel traced.
usage: python tools/slow_scripts_probe.py [script[:parameter=value,...] ...]      PROBE_SIZE=WxH, PROBE_MOVE=<units sideways per frame>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import geodesic_raytracing_amd as gra
from geodesic_raytracing_amd.pipeline import DeviceBuffer
scripts = os.path.join(ROOT, "geodesic_raytracing_amd", "scripts")
W, H = (int(x) for x in os.environ.get("PROBE_SIZE", "1920x1080").split("x"))
bg_np, levels = gra.pack_background(gra.synthetic_background(4096, 2048))
bg = DeviceBuffer.from_numpy(0, bg_np)
out = DeviceBuffer(0, W * H * 16)
for name in sys.argv[1:] or ["kerr_schild", "kerr_boyer", "kerr_newman_boyer"]:
    # name or name:parameter=value,... (kerr_schild:a=-0.5 is the reference script's default: an extremal hole)
    name, _, overrides = name.partition(":")
    m = gra.Metric(name, scripts)
    cfg = m.cfg_values(**{k: float(v) for k, v in (kv.split("=") for kv in overrides.split(",") if kv)})
    for adaptive in (1, 0):
        f = m.features(adaptive_sampling=adaptive, adaptive_sampling_threshold=32.0)
        prog = gra.Program(m.argument_string(features=f, static=True, cfg_values=cfg), 0)
        for label, kw in (("prepass its own launch", dict(inline_prepass=0, reuse_still_camera=0)), ("prepass inside the launch", dict(inline_prepass=-1, reuse_still_camera=0)),
                          ("prepass of the frame before", dict(inline_prepass=-1, reuse_still_camera=1))):
            st = gra.RenderState(W, H, 0)
            o = gra.frame_options(mode=gra.MODE_FUSED, time_kernels=1, guess_still_camera=0, **kw)
            acc, wall = {}, []
            import time
            for i in range(7):
                st.synchronize()
                t = time.perf_counter()
                st.render(prog, m, gra.default_camera([0, float(os.environ.get("PROBE_MOVE", "0")) * i, -4, 0]), out.ptr, (bg.ptr, 4096, 2048, levels), f, cfg, o)
                st.synchronize()
                if i >= 2:
                    wall.append((time.perf_counter() - t) * 1e3)
                    for k, v in st.stage_ms().items():
                        acc.setdefault(k, []).append(v)
            ms = {k: round(float(np.mean(v)), 2) for k, v in acc.items() if np.mean(v) > 0.005}
            print(f"{name:20s} adaptive={adaptive} {label:28s} wall {np.mean(wall):6.2f} ms = {1000 / np.mean(wall):6.1f} fps  stages {ms}", flush=True)
