"""When do the tiles of a lone 4K trace launch run?  A build of the ray kernels whose ticket loop stamps every tile's begin and end
(100 MHz counter) into two fields of the tile's first render-data record that nothing reads without redshift (z_shift, side) - the kernel
source is a patched copy handed over through GR_KERNEL_SOURCE, the product's source is not touched.  Prints the launch's occupancy over
time (tiles in progress per wave slot), where tiles of which duration start, and the longest tiles.
    build container: python tools/timeline_probe.py build        GPU box: python tools/timeline_probe.py run [spin]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HOME = os.path.join(ROOT, "geodesic_raytracing_amd", "_cache_variants", os.environ.get("TIMELINE_HOME", "timeline"))
SOURCE = os.path.join(HOME, "kernels.hip")
PARTS = ["program.hip", "probes.inc", "metric.hip", "setup.hip", "integrator.hip", "trace.hip", "shading.hip"]
STAMPS = '''#define GR_PROBE_TILE_BEGAN const unsigned long long tile_began = __builtin_amdgcn_s_memrealtime();
#define GR_PROBE_TILE_ENDED if (lane == 0 && cell_wave < 0 && wave < total_waves && strip_count <= 1 && !pending_only) { \\
            const int tiles_x = (width / lattice + GR_TILE - 1) / GR_TILE; \\
            render_data* first = rdata + (size_t)(wave / tiles_x) * GR_TILE * lattice * width + (size_t)(wave % tiles_x) * GR_TILE * lattice; \\
            first->z_shift = __int_as_float((int)(unsigned int)tile_began); \\
            first->side = (int)(unsigned int)__builtin_amdgcn_s_memrealtime(); \\
        }
#define GR_PROBE_WAVE_ENDED
'''
os.environ["GR_KERNEL_SOURCE"] = SOURCE
os.environ["GR_CACHE_DIR"] = HOME
import geodesic_raytracing_amd as gra  # noqa: E402
scripts = os.path.join(ROOT, "geodesic_raytracing_amd", "scripts")
metric = gra.Metric("kerr_boyer", scripts)
spin = float(sys.argv[2]) if len(sys.argv) > 2 else 0.45
cfg = metric.cfg_values(a=spin)
feats = metric.features(adaptive_sampling=0)
substituted = metric.argument_string(features=feats, static=True, cfg_values=cfg)

if sys.argv[1] == "build":
    os.makedirs(HOME, exist_ok=True)
    text = ""
    for part in PARTS:
        text += open(os.path.join(ROOT, "geodesic_raytracing_amd", "csrc", "kernels", part)).read() + "\n"
    old = "#else\n#define GR_PROBE_TILE_BEGAN\n#define GR_PROBE_TILE_ENDED\n#define GR_PROBE_WAVE_ENDED\n#endif"
    assert old in text
    text = text.replace(old, "#else\n" + STAMPS + "#endif", 1)
    # the list launch of adaptive sampling: begin and end of a ticket in the record of the ticket's first pixel
    old = "        unsigned int tries = 0;\n        if (first + lane < total) {\n            const unsigned int pixel = pending_list[GR_PENDING_HEADER + first + lane];"
    assert old in text
    text = text.replace(old, "        const unsigned long long ticket_began = __builtin_amdgcn_s_memrealtime();\n" + old, 1)
    old = "        if (attempt_counter) atomicAdd(attempt_counter + GR_ATTEMPT_COUNTERS_AT + (blockIdx.x % GR_ATTEMPT_COUNTERS), (unsigned long long)tries);\n    }\n}"
    assert old in text
    text = text.replace(old, "        if (lane == 0) { render_data* r = rdata + pending_list[GR_PENDING_HEADER + first]; r->z_shift = __int_as_float((int)(unsigned int)ticket_began); "
                             "r->side = (int)(unsigned int)__builtin_amdgcn_s_memrealtime(); }\n" + old, 1)
    open(SOURCE, "w").write(text)
    for a in (0.45, 0.9):
        gra.Program.precompile(metric.argument_string(features=feats, static=True, cfg_values=metric.cfg_values(a=a)))
    gra.Program.precompile(metric.argument_string(features=metric.features(adaptive_sampling=1, adaptive_sampling_threshold=32.0), static=True, cfg_values=metric.cfg_values(a=0.45)))
    print("built", SOURCE)
    sys.exit(0)

from geodesic_raytracing_amd.pipeline import DeviceBuffer, RENDER_DATA_DTYPE, download  # noqa: E402
W, H = 3840, 2160
ADAPTIVE = os.environ.get("TIMELINE_ADAPTIVE") == "1"
if ADAPTIVE:
    feats = metric.features(adaptive_sampling=1, adaptive_sampling_threshold=32.0)
    substituted = metric.argument_string(features=feats, static=True, cfg_values=cfg)
prog = gra.Program(substituted, 0)
state = gra.RenderState(W, H, 0)
out = DeviceBuffer(0, W * H * 16)
packed, levels = gra.pack_background(gra.synthetic_background(4096, 2048))
bg = DeviceBuffer.from_numpy(0, packed)
cam = gra.default_camera()
inline = int(os.environ.get("TIMELINE_INLINE_PREPASS", "-1"))
for frame in range(4):      # the fourth frame follows the history of the third
    state.render(prog, metric, cam, out.ptr, (bg.ptr, 4096, 2048, levels), feats, cfg, gra.frame_options(mode=gra.MODE_FUSED, time_kernels=1, inline_prepass=inline))
    state.synchronize()
print("stage ms", {k: round(v, 3) for k, v in state.stage_ms().items()}, "tile history", state.tile_history())
full = download(0, state.buffer(gra.BUF_RENDER_DATA), RENDER_DATA_DTYPE, W * H).reshape(H, W)
slots = 256 * 4 * 6
if ADAPTIVE:
    # lattice tiles: the record of pixel (16 i, 16 j); tickets of the list launch: records whose `side` is a time stamp (not 0 / 1)
    lat = full[::16, ::16]
    lb = lat["z_shift"].view(np.uint32).astype(np.int64).ravel()
    le = lat["side"].view(np.uint32).astype(np.int64).ravel()
    mask = np.ones((H, W), dtype=bool)
    mask[::2, ::2] = False
    cand = full[mask]
    t0 = lb.min()
    cb = cand["z_shift"].view(np.uint32).astype(np.int64)
    ce = cand["side"].view(np.uint32).astype(np.int64)
    stamped = (cb > t0) & (cb < t0 + 2_000_000) & (ce > cb) & (ce < t0 + 2_000_000)     # both within 20 ms of the lattice launch's first tile
    tb, te = cb[stamped], ce[stamped]
    for tag, bb, ee in (("lattice launch: tiles", lb, le), ("list launch: tickets of 64 pixels", tb, te)):
        b, e = (bb - t0) / 100.0, (ee - t0) / 100.0
        dur = e - b
        print(f"{tag}: {len(b)}, from {b.min() / 1e3:.3f} to {e.max() / 1e3:.3f} ms, sum of their times {dur.sum() / 1e3:.1f} wave-ms = {dur.sum() / (e.max() - b.min()) / slots:.3f} of {slots} wave slots; "
              f"longest {dur.max():.0f} us (began at {b[dur.argmax()]:.0f}), median {np.median(dur):.0f}")
        edges = np.arange(b.min(), e.max() + 250, 250.0)
        print("   t (ms): in progress / slots:", " ".join(f"{lo / 1e3:.2f}:{((b <= lo + 125) & (e > lo + 125)).sum() / slots:.2f}" for lo in edges[:-1]))
    sys.exit(0)
rd = full[::8, ::8]
began = rd["z_shift"].view(np.uint32).astype(np.int64).ravel()
ended = rd["side"].view(np.uint32).astype(np.int64).ravel()
t0 = began.min()
b, e = (began - t0) / 100.0, (ended - t0) / 100.0          # microseconds
dur = e - b
span = e.max()
print(f"tiles {len(b)}, launch span {span / 1e3:.3f} ms, sum of tile times {dur.sum() / 1e3:.1f} wave-ms = {dur.sum() / span / slots:.3f} of {slots} wave slots")
edges = np.arange(0, span + 250, 250.0)
print(" t (ms)   tiles in progress / slots   tiles started   their mean duration (us)   of them > 500 us")
for lo, hi in zip(edges[:-1], edges[1:]):
    mid = (lo + hi) / 2
    active = ((b <= mid) & (e > mid)).sum()
    started = (b >= lo) & (b < hi)
    print(f" {lo / 1e3:5.2f}    {active / slots:6.3f}                     {started.sum():7d}         {dur[started].mean() if started.any() else 0:9.1f}              {(dur[started] > 500).sum():6d}")
order = np.argsort(-dur)[:12]
print("longest tiles: duration us, began at us, tile (x, y):", [(round(float(dur[i])), round(float(b[i])), (int(i % (W // 8)), int(i // (W // 8)))) for i in order])
late = np.argsort(-e)[:12]
print("last to end:   ended at us, duration us, began at us:", [(round(float(e[i])), round(float(dur[i])), round(float(b[i]))) for i in late])
