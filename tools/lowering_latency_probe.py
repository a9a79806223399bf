"""Does the device lowering of the accelerations (GR_DEVICE_ACCEL*: sin^2 / cos^2 / sin cos reduced by multiples of pi ...) cost a latency-bound
frame time?  Kerr (Boyer-Lindquist) at the reference script's default a = -0.5 (extremal), 1920x1080, every pixel, substituted program, one frame
at a time, several camera poses; run once plainly and once with GR_EXTRA_FLAGS=-DGR_NO_DEVICE_LOWERING.  usage: python tools/lowering_latency_probe.py [a]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import geodesic_raytracing_amd as gra
from geodesic_raytracing_amd.pipeline import DeviceBuffer
W, H = 1920, 1080
a = float(sys.argv[1]) if len(sys.argv) > 1 else -0.5
m = gra.Metric("kerr_boyer", os.path.join(ROOT, "geodesic_raytracing_amd", "scripts"))
cfg = m.cfg_values(a=a)
f = m.features(adaptive_sampling=0)
prog = gra.Program(m.argument_string(features=f, static=True, cfg_values=cfg), 0)
out = DeviceBuffer(0, W * H * 16)
bg_np, levels = gra.pack_background(gra.synthetic_background(4096, 2048))
bg = DeviceBuffer.from_numpy(0, bg_np)
for pos in ([0, 0, -4, 0], [0, 0.5, -4, 0.7], [0, 1.0, -6, 2.0], [0, -2.0, -3, -1.0], [0, 0, -10, 0.3]):
    st = gra.RenderState(W, H, 0)
    o = gra.frame_options(mode=gra.MODE_FUSED, count_attempts=1, reuse_still_camera=0, guess_still_camera=0)
    ts = []
    for i in range(8):
        st.synchronize(); t = time.perf_counter()
        st.render(prog, m, gra.default_camera(pos), out.ptr, (bg.ptr, 4096, 2048, levels), f, cfg, o)
        st.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    print(f"a={a} camera {pos}: {np.median(ts[3:]):7.2f} ms/frame, attempts {st.attempts()}", flush=True)
