"""Wave-slot occupancy of single fused trace launches of the 4K Kerr workload: launch duration, waves, summed wave lifetime, the
share of the launch's wave slots that was occupied, shader clock; with GR_EXTRA_FLAGS=-DGR_PROBE_LIFE_HISTOGRAM also the histogram
of wave lifetimes.   usage: [GR_TILE_ORDER=0] [GR_TRACE_WAVES_PER_SIMD=k] python tools/slot_probe.py"""
import os, sys, numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import geodesic_raytracing_amd as gra
W, H = 3840, 2160
metric = gra.Metric("kerr_boyer", os.path.join(ROOT, "geodesic_raytracing_amd", "scripts"))
cfgv = metric.cfg_values(a=0.45); features = metric.features(adaptive_sampling=0)
manager = gra.pipeline.ProgramManager(metric, 0, features, cfgv); prog = manager.current(wait=True)
print("kernel", prog.kernel_info("gr_trace_fused"))
state = gra.RenderState(W, H); out = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
bgbuf, levels = gra.pack_background(gra.synthetic_background(4096, 2048)); bg = torch.from_numpy(bgbuf).cuda()
cam = gra.default_camera()
opts = gra.frame_options(mode=gra.MODE_FUSED, tiled=1, time_kernels=1, count_attempts=1)
for i in range(5):
    state.render(prog, metric, cam, out.data_ptr(), (bg.data_ptr(), 4096, 2048, levels), features, cfgv, opts, None)
    torch.cuda.synchronize()
    ms, n = state.wave_time(); st = state.stage_ms()
    print("launch %.3f ms  waves %d  wave-ms %.1f  mean life %.3f ms  slot share %.3f  clock %.0f" % (st["trace"], n, ms, ms/n, ms/n/st["trace"], state.shader_clock_mhz()))
if "LIFE_HISTOGRAM" in os.environ.get("GR_EXTRA_FLAGS", ""):
    words = state.counters(256)
    h = words[8:128]
    print("lifetime histogram (0.125 ms bins):", " ".join("%d:%d" % (i, c) for i, c in enumerate(h) if c))
    for c in range(16):
        n, total, longest = words[128 + 3 * c: 131 + 3 * c]
        if n:
            print("  class %d: %6d tiles, mean %8.1f us, longest %8.1f us, %5.1f %% of the tile time" % (
                c, n, total / n / 100, longest / 100, 100.0 * total / max(1, sum(words[129 + 3 * k] for k in range(16)))))
