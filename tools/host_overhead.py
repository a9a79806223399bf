"""Host-side cost of one gr_render_frame call (Python ctypes + HIP launches), measured with a frame so small that the GPU is never
the bottleneck.  usage: python tools/host_overhead.py"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import geodesic_raytracing_amd as gra

metric = gra.Metric("kerr_boyer", os.path.join(ROOT, "geodesic_raytracing_amd", "scripts"))
cfg = metric.cfg_values(a=0.45)
feats = metric.features(adaptive_sampling=0)
program = gra.Program(metric.argument_string(features=feats, static=True, cfg_values=cfg), 0)
W, H = 256, 144
state = gra.RenderState(W, H, 0)
bg_np, levels = gra.pack_background(gra.synthetic_background(256, 128))
bg = torch.from_numpy(bg_np).cuda()
out = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
camera = gra.default_camera()
look = ctypes.pointer(camera)
stream = torch.cuda.current_stream().cuda_stream
for label, kw in (("no look-ahead", {}), ("look-ahead 1", dict(next_camera=look)), ("look-ahead 2", dict(next_camera=look, next_camera2=look)),
                  ("look-ahead 2 + strips + trace log", dict(next_camera=look, next_camera2=look, strip_rank=0, strip_count=8, block_rows=16,
                                                             compact_out=1, time_kernels=2))):
    def frame():
        o = gra.frame_options(mode=gra.MODE_FUSED, **kw)
        state.render(program, metric, camera, out.data_ptr(), (bg.data_ptr(), 256, 128, levels), feats, cfg, o, stream)
    for _ in range(20):
        frame()
    torch.cuda.synchronize()
    n = 300
    t = time.perf_counter()
    for _ in range(n):
        frame()
    host = (time.perf_counter() - t) / n * 1e6
    torch.cuda.synchronize()
    total = (time.perf_counter() - t) / n * 1e6
    print(f"{label:36s} host {host:7.1f} us per frame, with GPU drain {total:7.1f} us")
    state.trace_log(reset=True)
