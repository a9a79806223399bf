"""How much of a tile-wave's work is useful: the attempts of a frame's rays against 64 x the attempts of each tile's longest ray
(what the wave executes), from gr_trace_fused's own counters (attempt_counter, tile_cost).  Also what parking a tile's last K rays
once fewer than K are left would save: an upper estimate from the per-pixel attempts is not available on the GPU, so only the
first number is printed per workload.   usage: python tools/lane_utilisation.py [a ...]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import geodesic_raytracing_amd as gra
from geodesic_raytracing_amd import check, lib
from geodesic_raytracing_amd.pipeline import DeviceBuffer, RENDER_DATA_DTYPE, download

W, H = 3840, 2160
SCRIPTS = os.path.join(ROOT, "geodesic_raytracing_amd", "scripts")
bg_np, levels = gra.pack_background(gra.synthetic_background(1024, 512))
bg = DeviceBuffer.from_numpy(0, bg_np)
for a_spin in [float(x) for x in sys.argv[1:]] or [0.45, 0.9]:
    metric = gra.Metric("kerr_boyer", SCRIPTS)
    cfgv = metric.cfg_values(a=a_spin)
    feats = metric.features(adaptive_sampling=0)
    prog = gra.Program(metric.argument_string(feats, static=True, cfg_values=cfgv), 0)
    state = gra.RenderState(W, H, 0)
    out = DeviceBuffer(0, W * H * 16)
    state.render(prog, metric, gra.default_camera(), out.ptr, (bg.ptr, 1024, 512, levels), feats, cfgv, gra.frame_options(mode=gra.MODE_FUSED, inline_prepass=0, tile_history=0))
    state.synchronize()
    b = state.buffer
    rows = ((H + 7) // 8) * 8
    tiles = (W // 8) * (rows // 8)
    rd = DeviceBuffer(0, W * H * RENDER_DATA_DTYPE.itemsize)
    cost = DeviceBuffer.from_numpy(0, np.zeros(tiles, dtype=np.uint32))
    attempts = DeviceBuffer.from_numpy(0, np.zeros(512, dtype=np.uint64))
    args = gra.TraceFusedArgs(camera_generic=b(gra.BUF_CAMERA_GENERIC), camera_quat=b(gra.BUF_CAMERA_QUAT), render_data=rd.ptr, width=W, height=H,
                              block_rows=0, strip_rank=0, strip_count=1, termination_buffer=b(gra.BUF_TERMINATION), prepass_width=W // 16,
                              prepass_height=H // 16, e0=b(gra.BUF_TETRAD0), e1=b(gra.BUF_TETRAD1), e2=b(gra.BUF_TETRAD2), e3=b(gra.BUF_TETRAD3),
                              cfg=b(gra.BUF_CFG), dfg=b(gra.BUF_DFG), tile_cost=cost.ptr, attempt_counter=attempts.ptr)
    check(lib.gr_trace_fused_launch(prog.handle, None, ctypes.byref(args)))
    check(lib.gr_device_synchronize(0))
    costs = cost.to_numpy(np.uint32, (tiles,)).astype(np.int64)
    counted = attempts.to_numpy(np.uint64, (512,))
    total = int(counted[0] + counted[256:].sum())
    executed = int(costs.sum()) * 64
    traced = download(0, rd.ptr, RENDER_DATA_DTYPE, W * H)["terminated"] != 2
    hist = np.histogram(costs[costs > 0], bins=[1, 128, 256, 512, 1024, 2048, 4096, 8192, 1 << 20])[0]
    print(f"a = {a_spin}: {traced.sum()} rays traced, {total} attempts, tile-waves execute {executed} lane-attempts: lane utilisation {total / executed:.3f}; "
          f"traced tiles {int((costs > 0).sum())}, by longest ray [1,128,256,512,1k,2k,4k,8k,..): {hist.tolist()}; "
          f"share of the executed lane-attempts in tiles whose longest ray is over 2048: {costs[costs > 2048].sum() * 64 / executed:.3f}", flush=True)
