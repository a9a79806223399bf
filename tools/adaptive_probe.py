"""Stage times of the adaptive-sampling frame (reference-shaped sequence), dynamic and substituted programs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import geodesic_raytracing_amd as gra
scripts = os.path.join(ROOT, "geodesic_raytracing_amd", "scripts")
W, H = 3840, 2160
m = gra.Metric("kerr_boyer", scripts)
cfg = m.cfg_values(a=0.45)
bg_np, levels = gra.pack_background(gra.synthetic_background(4096, 2048))
bg = torch.from_numpy(bg_np).cuda()
out = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
st = gra.RenderState(W, H, 0)
for thr in (32.0,):
    f = m.features(adaptive_sampling=1, adaptive_sampling_threshold=thr)
    for label, prog in (("dynamic", gra.Program(m.argument_string(), 0)),
                        ("substituted", gra.Program(m.argument_string(features=f, static=True, cfg_values=cfg), 0))):
        for tiled in (0, 1):
            o = gra.frame_options(mode=gra.MODE_REFERENCE, tiled=tiled, time_kernels=1, count_attempts=1)
            acc = {}
            for i in range(4):
                st.render(prog, m, gra.default_camera(), out.data_ptr(), (bg.data_ptr(), 4096, 2048, levels), f, cfg, o)
                st.synchronize()
                if i:
                    for k, v in st.stage_ms().items():
                        acc.setdefault(k, []).append(v)
            ms = {k: round(float(np.mean(v)), 3) for k, v in acc.items()}
            n_adaptive = gra.pipeline.download(0, st.buffer(gra.BUF_RAYS_ADAPTIVE_COUNT), np.int32, 1)[0]
            print(label, "tiled", tiled, "total", round(sum(ms.values()), 3), ms, "adaptive rays", int(n_adaptive), "attempts", st.attempts(), flush=True)
