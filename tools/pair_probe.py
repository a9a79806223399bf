"""gr_trace_pair (two rays per lane) against gr_trace_fused: are the frames identical, and how long does a 4K frame take?
usage: python tools/pair_probe.py [metric ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import geodesic_raytracing_amd as gra  # noqa: E402
from geodesic_raytracing_amd.pipeline import DeviceBuffer  # noqa: E402

SCRIPTS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "geodesic_raytracing_amd", "scripts")


def main():
    names = sys.argv[1:] or ["kerr_boyer", "schwarzschild", "alcubierre", "kerr_newman_boyer", "minkowski", "wormhole"]
    packed, levels = gra.pack_background(gra.synthetic_background(1024, 512))
    bg = DeviceBuffer.from_numpy(0, packed)
    for name in names:
        m = gra.Metric(name, SCRIPTS)
        cfg = m.cfg_values(a=0.45) if name == "kerr_boyer" else m.cfg_values()
        feats = m.features(adaptive_sampling=0)
        for static in ((True,) if os.environ.get("PAIR_PROBE_STATIC_ONLY") else (False, True)):
            prog = gra.Program(m.argument_string(features=feats, static=static, cfg_values=cfg) if static else m.argument_string(), 0)
            if not prog.has_trace_pair:
                print(f"{name:22s} {'static' if static else 'dynamic':8s} no pair kernel")
                continue
            info1, info2 = prog.kernel_info("gr_trace_fused"), prog.kernel_info("gr_trace_pair")
            out = {}
            for W, H in ((640, 360), (3840, 2160)):
                state = gra.RenderState(W, H, 0)
                buf = DeviceBuffer(0, W * H * 16)
                cam = gra.default_camera()
                for rpl in (1, 2):
                    o = gra.frame_options(mode=gra.MODE_FUSED, rays_per_lane=rpl, count_attempts=1)
                    state.render(prog, m, cam, buf.ptr, (bg.ptr, 1024, 512, levels), feats, cfg, o)
                    state.synchronize()
                    img = buf.to_numpy(np.float32, (H, W, 4)).copy()
                    att = state.attempts() if hasattr(state, "attempts") else 0
                    o = gra.frame_options(mode=gra.MODE_FUSED, rays_per_lane=rpl)
                    t0 = time.perf_counter()
                    n = 5
                    for _ in range(n):
                        state.render(prog, m, cam, buf.ptr, (bg.ptr, 1024, 512, levels), feats, cfg, o)
                    state.synchronize()
                    ms = (time.perf_counter() - t0) / n * 1e3
                    o = gra.frame_options(mode=gra.MODE_FUSED, rays_per_lane=rpl, time_kernels=1)
                    state.render(prog, m, cam, buf.ptr, (bg.ptr, 1024, 512, levels), feats, cfg, o)
                    state.synchronize()
                    trace_ms = state.stage_ms()["trace"]
                    out[(W, rpl)] = (img, ms, att, trace_ms)
            a, b = out[(640, 1)][0], out[(640, 2)][0]
            a4, b4 = out[(3840, 1)][0], out[(3840, 2)][0]
            print(f"{name:22s} {'static' if static else 'dynamic':8s} vgprs {info1['vgprs']:3d}/{info2['vgprs']:3d} scratch {info2['scratch_bytes']}  "
                  f"640x360 identical {np.array_equal(a, b)} max|d| {np.abs(a - b).max():.2e}  4K identical {np.array_equal(a4, b4)} "
                  f"differing px {(np.abs(a4 - b4).max(axis=2) > 0).mean() * 100:.3f}%  attempts {out[(3840, 1)][2]} / {out[(3840, 2)][2]}  "
                  f"4K ms {out[(3840, 1)][1]:.2f} -> {out[(3840, 2)][1]:.2f}  trace ms {out[(3840, 1)][3]:.2f} -> {out[(3840, 2)][3]:.2f}  "
                  f"ns per count/1024 SIMDs: {out[(3840, 1)][3] * 1e6 / (out[(3840, 1)][2] / 64 / 1024):.0f} (x64) {out[(3840, 2)][3] * 1e6 / (out[(3840, 2)][2] / 128 / 1024):.0f} (x128)")


if __name__ == "__main__":
    main()
