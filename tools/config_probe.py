"""Dynamic vs substituted program on the other BASELINE configurations (one frame at a time, fused kernel)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import geodesic_raytracing_amd as gra
scripts = os.path.join(ROOT, "geodesic_raytracing_amd", "scripts")
bg_np, levels = gra.pack_background(gra.synthetic_background(4096, 2048))
bg = torch.from_numpy(bg_np).cuda()
for name, (w, h), cam, fk in (("schwarzschild", (1920, 1080), None, {}), ("double_unequal_kerr", (3840, 2160), [0, 0, -6, 0.5], {}),
                               ("alcubierre", (7680, 4320), [0, 0, -6, 0.5], {"redshift": 1}), ("kerr_schild", (3840, 2160), None, {})):
    m = gra.Metric(name, scripts)
    f = m.features(adaptive_sampling=0, **fk)
    cfg = m.cfg_values()
    st = gra.RenderState(w, h, 0)
    out = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    c = gra.default_camera(cam)
    for label, prog in (("dynamic", gra.Program(m.argument_string(), 0)),
                        ("substituted", gra.Program(m.argument_string(features=f, static=True, cfg_values=cfg), 0))):
        o = gra.frame_options(mode=gra.MODE_FUSED, time_kernels=1)
        ts = []
        for _ in range(4):
            st.render(prog, m, c, out.data_ptr(), (bg.data_ptr(), 4096, 2048, levels), f, cfg, o)
            st.synchronize()
            ts.append(st.stage_ms()["trace"])
        print(f"{name:22s} {w}x{h} {label:12s} trace {min(ts[1:]):8.3f} ms  vgpr {prog.kernel_info('gr_trace_fused')}", flush=True)
