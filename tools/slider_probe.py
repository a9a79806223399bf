"""A slider being dragged: the dynamic program, a new parameter value every frame, one frame at a time (4K Kerr, still camera, every frame its
own prepass): a drifting by 0.002 a frame (the orders of the frame before are followed), a alternating between 0.45 and 0.30 (they are not: the
way every changed parameter was treated before), a constant.  usage: python tools/slider_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import geodesic_raytracing_amd as gra
from geodesic_raytracing_amd.pipeline import DeviceBuffer
W, H = (int(x) for x in os.environ.get("PROBE_SIZE", "3840x2160").split("x"))
m = gra.Metric("kerr_boyer", os.path.join(ROOT, "geodesic_raytracing_amd", "scripts"))
out = DeviceBuffer(0, W * H * 16)
bg_np, levels = gra.pack_background(gra.synthetic_background(4096, 2048))
bg = DeviceBuffer.from_numpy(0, bg_np)
for adaptive in (0, 1):
    f = m.features(adaptive_sampling=adaptive, adaptive_sampling_threshold=32.0)
    prog = gra.Program(m.argument_string(), 0)
    for label, value in (("constant", lambda k: 0.45), ("dragged by 0.002 a frame", lambda k: 0.40 + 0.002 * k), ("set to 0.45 / 0.30 in turn", lambda k: 0.45 if k % 2 else 0.30)):
        st = gra.RenderState(W, H, 0)
        o = gra.frame_options(mode=gra.MODE_FUSED, reuse_still_camera=0, guess_still_camera=0)
        ts = []
        for k in range(16):
            st.synchronize(); t = time.perf_counter()
            st.render(prog, m, gra.default_camera(), out.ptr, (bg.ptr, 4096, 2048, levels), f, m.cfg_values(a=value(k)), o)
            st.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
        print(f"adaptive={adaptive} a {label:28s}: {np.mean(ts[4:]):6.2f} ms/frame (min {min(ts[4:]):.2f}, max {max(ts[4:]):.2f}), histories followed {st.tile_history()[1]} of 16", flush=True)
