"""Where the time of an adaptively sampled frame goes on the fused path (GPU): lattice launch, gr_adaptive_refine + second launch,
texture pass; how many pixels the second launch traces and what they cost.  4K Kerr a = 0.45, substituted program."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import geodesic_raytracing_amd as gra
from geodesic_raytracing_amd.pipeline import DeviceBuffer, RENDER_DATA_DTYPE, download
scripts = os.path.join(ROOT, "geodesic_raytracing_amd", "scripts")
W, H = 3840, 2160
m = gra.Metric("kerr_boyer", scripts)
cfg = m.cfg_values(a=0.45)
bg_np, levels = gra.pack_background(gra.synthetic_background(4096, 2048))
bg = DeviceBuffer.from_numpy(0, bg_np)
out = DeviceBuffer(0, W * H * 16)
st = gra.RenderState(W, H, 0)
for thr, adaptive in ((32.0, 1), (32.0, 0)):
    f = m.features(adaptive_sampling=adaptive, adaptive_sampling_threshold=thr)
    prog = gra.Program(m.argument_string(features=f, static=True, cfg_values=cfg), 0)
    o = gra.frame_options(mode=gra.MODE_FUSED, time_kernels=1, count_attempts=1)
    acc = {}
    for i in range(5):
        st.render(prog, m, gra.default_camera(), out.ptr, (bg.ptr, 4096, 2048, levels), f, cfg, o)
        st.synchronize()
        if i:
            for k, v in st.stage_ms().items():
                acc.setdefault(k, []).append(v)
    ms = {k: round(float(np.mean(v)), 3) for k, v in acc.items()}
    line = f"adaptive={adaptive} total {sum(ms.values()):.3f} ms {ms} attempts {st.attempts()}"
    if adaptive:
        rd = download(0, st.buffer(gra.BUF_RENDER_DATA), RENDER_DATA_DTYPE, W * H).reshape(H, W)
        n_new = int(download(0, st.buffer(gra.BUF_RAYS_ADAPTIVE_COUNT), np.int32, 1)[0])
        line += f" pixels of the second launch {n_new} ({n_new / (W * H) * 100:.1f} % of the frame)"
    print(line, flush=True)

# ---- three frames in flight (three render states on three streams, as bench.py's headline): throughput instead of latency
import ctypes, time
from geodesic_raytracing_amd import lib, check
streams = []
for i in range(3):
    sp = ctypes.c_void_p()
    check(lib.gr_stream_create(0, 0, ctypes.byref(sp)))
    streams.append(sp)
states = [gra.RenderState(W, H, 0) for _ in range(3)]
outs = [DeviceBuffer(0, W * H * 16) for _ in range(3)]
for thr, adaptive in ((32.0, 1), (32.0, 0)):
    f = m.features(adaptive_sampling=adaptive, adaptive_sampling_threshold=thr)
    prog = gra.Program(m.argument_string(features=f, static=True, cfg_values=cfg), 0)
    for wps in (0, 4):
        o = gra.frame_options(mode=gra.MODE_FUSED, trace_waves_per_simd=wps)
        def frame(i):
            k = i % 3
            check(lib.gr_stream_synchronize(streams[k]))
            states[k].render(prog, m, gra.default_camera(), outs[k].ptr, (bg.ptr, 4096, 2048, levels), f, cfg, o, streams[k])
        for i in range(6):
            frame(i)
        for sp in streams:
            check(lib.gr_stream_synchronize(sp))
        t = time.perf_counter()
        n = 30
        for i in range(n):
            frame(i)
        for sp in streams:
            check(lib.gr_stream_synchronize(sp))
        t = (time.perf_counter() - t) / n
        print(f"three frames in flight, adaptive={adaptive}, trace_waves_per_simd={wps}: {t * 1e3:.3f} ms per frame, {W * H / t / 1e6:.0f} Mpixels/s", flush=True)
