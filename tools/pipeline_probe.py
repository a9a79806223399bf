"""Frames per second of the fused 4K Kerr pipeline with three frames in flight, with and without the shading kernel and the
look-ahead prepass: what each costs in the pipelined steady state.   usage: python tools/pipeline_probe.py"""
import ctypes, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import torch
import geodesic_raytracing_amd as gra
W, H = 3840, 2160
metric = gra.Metric("kerr_boyer", os.path.join(ROOT, "geodesic_raytracing_amd", "scripts"))
cfg = metric.cfg_values(a=0.45); feats = metric.features(adaptive_sampling=0)
program = gra.pipeline.ProgramManager(metric, 0, feats, cfg).current(wait=True)
bg_np, levels = gra.pack_background(gra.synthetic_background(4096, 2048)); bg = torch.from_numpy(bg_np).cuda()
camera = gra.default_camera(); look = ctypes.pointer(camera)
for render in (1, 0):
  for inflight in (3,):
    for lookahead in (1, 0):
        states = [gra.RenderState(W, H, 0) for _ in range(inflight)]
        streams = [torch.cuda.Stream() for _ in range(inflight)]
        outs = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(inflight)]
        n = [0]
        def frame():
            k = n[0] % inflight; n[0] += 1
            o = gra.frame_options(mode=gra.MODE_FUSED)
            if lookahead: o.next_camera = look
            states[k].render(program, metric, camera, outs[k].data_ptr() if render else 0, (bg.data_ptr(), 4096, 2048, levels), feats, cfg, o, streams[k].cuda_stream)
        for _ in range(8): frame()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(60): frame()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t) / 60 * 1e3
        print(f"render={render} lookahead={lookahead} in flight {inflight}: {ms:.3f} ms/frame {W*H/ms/1e3:.0f} Mrays/s", flush=True)
        del states, streams, outs
