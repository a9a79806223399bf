"""Per-frame time of ONE rank's share of a frame split N ways (no communication), by look-ahead depth: what an N-GPU run
can reach per frame.   usage: [STRIP_PROBE_INFLIGHT=1,3,5] [STRIP_PROBE_BLOCK=48] [STRIP_PROBE_WAVES=2] python tools/strip_probe.py [N ...]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import geodesic_raytracing_amd as gra

W, H = 3840, 2160
metric = gra.Metric("kerr_boyer", os.path.join(ROOT, "geodesic_raytracing_amd", "scripts"))
cfg = metric.cfg_values(a=0.45)
feats = metric.features(adaptive_sampling=0)
program = gra.pipeline.ProgramManager(metric, 0, feats, cfg).current(wait=True)
states = [gra.RenderState(W, H, 0) for _ in range(8)]
streams = [torch.cuda.Stream() for _ in range(8)]
bg_np, levels = gra.pack_background(gra.synthetic_background(4096, 2048))
bg = torch.from_numpy(bg_np).cuda()
outs = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(8)]
camera = gra.default_camera()
BLOCK = int(os.environ.get("STRIP_PROBE_BLOCK", "16"))
look = ctypes.pointer(camera)
for world in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
  for rotate in ((os.environ.get("STRIP_PROBE_ROTATE", "1") != "0",) if os.environ.get("STRIP_PROBE_INFLIGHT") else (False, True)):
   for inflight in [int(x) for x in os.environ.get("STRIP_PROBE_INFLIGHT", "1,3").split(",")]:
    for depth in [int(x) for x in os.environ.get("STRIP_PROBE_DEPTH", "2").split(",")]:
        counter = [0]

        def frame():
            k = counter[0] % inflight
            counter[0] += 1
            state, out, stream = states[k], outs[k], streams[k].cuda_stream
            kf = counter[0] - 1
            o = gra.frame_options(mode=gra.MODE_FUSED, strip_rank=(kf % world) if rotate else 0, strip_count=world, block_rows=BLOCK, compact_out=1,
                                  trace_waves_per_simd=int(os.environ.get("STRIP_PROBE_WAVES", "0")),
                                  tile_history=int(os.environ.get("STRIP_PROBE_HISTORY", "-1")), inline_prepass=int(os.environ.get("STRIP_PROBE_INLINE", "-1")))
            if rotate:
                o.next_strip_rank = (kf + inflight) % world
                o.next_strip_rank2 = (kf + 2 * inflight) % world
            if depth >= 1:
                o.next_camera = look
            if depth >= 2:
                o.next_camera2 = look
            state.render(program, metric, camera, out.data_ptr(), (bg.data_ptr(), 4096, 2048, levels), feats, cfg, o, stream)
        for _ in range(5):
            frame()
        torch.cuda.synchronize()
        t = time.perf_counter()
        n = 40
        for _ in range(n):
            frame()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / n * 1e3
        print(f"1 of {world} ranks, rotate={rotate}, {inflight} frames in flight, look-ahead depth {depth}, {BLOCK}-row blocks: {ms:6.3f} ms/frame  -> {W * H / ms / 1e3:8.1f} Mrays/s if every rank keeps up", flush=True)
