"""GPU box: what parking (gr_trace_fused_parking) does to a frame's trace launch, one frame at a time with the tiles handed out by the
frame before's costs: stage times of gr_render_frame for the plain kernel and a few (lanes, trips).
usage: python tools/park_probe.py [a ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import geodesic_raytracing_amd as gra
from geodesic_raytracing_amd.pipeline import DeviceBuffer

W, H = 3840, 2160
SCRIPTS = os.path.join(ROOT, "geodesic_raytracing_amd", "scripts")
bg_np, levels = gra.pack_background(gra.synthetic_background(1024, 512))
bg = DeviceBuffer.from_numpy(0, bg_np)
settings = [(0, 0), (8, 512), (16, 512), (32, 512), (16, 256), (32, 256), (16, 1024), (48, 256)]
if os.environ.get("PARK_SETTINGS"):
    settings = [(0, 0)] + [tuple(int(v) for v in t.split(",")) for t in os.environ["PARK_SETTINGS"].split(";")]
for a_spin in [float(x) for x in sys.argv[1:]] or [0.9, 0.45]:
    metric = gra.Metric("kerr_boyer", SCRIPTS)
    cfgv = metric.cfg_values(a=a_spin)
    feats = metric.features(adaptive_sampling=0)
    prog = gra.Program(metric.argument_string(feats, static=True, cfg_values=cfgv) + " -DGR_PARKING", 0)
    out = DeviceBuffer(0, W * H * 16)
    for name in ("gr_trace_fused", "gr_trace_fused_parking"):
        print(f"a={a_spin} {name}: {prog.kernel_info(name)}")
    reference = None
    for lanes, trips in settings:
        state = gra.RenderState(W, H, 0)
        times = []
        for i in range(8):
            state.render(prog, metric, gra.default_camera(), out.ptr, (bg.ptr, 1024, 512, levels), feats, cfgv,
                         gra.frame_options(mode=gra.MODE_FUSED, time_kernels=1, park_lanes=lanes, park_trips=max(trips, 1)))
            state.synchronize()
            if i >= 3:
                times.append(state.stage_ms()["trace"])
        frame = out.to_numpy(np.float32, (H, W, 4)).tobytes()
        if reference is None:
            reference = frame
        print(f"a={a_spin} park lanes {lanes:2d} trips {trips:4d}: trace {np.mean(times):7.3f} ms (min {np.min(times):7.3f})  same frame: {frame == reference}", flush=True)
