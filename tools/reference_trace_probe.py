"""GPU box: the reference-shaped trace of a 4K Kerr frame's rays (tile slot order, prepass flags applied) three ways - gr_do_generic_rays (a
workgroup to a tile, slot order), gr_do_generic_rays_scheduled in slot order, and dearest first by the costs of the launch before - with
HIP events around each launch.    PYTHONPATH=. python tools/reference_trace_probe.py"""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import geodesic_raytracing_amd as gra
from geodesic_raytracing_amd import check, lib
from geodesic_raytracing_amd.pipeline import DeviceBuffer

w, h = 3840, 2160
metric = gra.Metric("kerr_boyer", os.path.join(ROOT, "geodesic_raytracing_amd", "scripts"))
cfgv, feats = metric.cfg_values(a=0.45), metric.features(adaptive_sampling=0)
prog = gra.Program(metric.argument_string(feats, static=True, cfg_values=cfgv), 0)
state = gra.RenderState(w, h, 0)
state.render(prog, metric, gra.default_camera(), None, None, feats, cfgv, gra.frame_options(mode=gra.MODE_REFERENCE, tiled=1))
state.synchronize()
b = state.buffer
slots = lib.gr_tiled_slot_count(w, h)
tiles_x, tiles_y = w // 8, h // 8
n = tiles_x * tiles_y
rays0 = DeviceBuffer(0, slots * 96)
count = DeviceBuffer.from_numpy(0, np.zeros(1, dtype=np.int32))
check(lib.gr_init_rays_generic(prog.handle, None, b(gra.BUF_CAMERA_GENERIC), b(gra.BUF_CAMERA_QUAT), rays0.ptr, count.ptr, w, h, b(gra.BUF_TERMINATION), w // 16, h // 16, 0,
                               b(gra.BUF_TETRAD0), b(gra.BUF_TETRAD1), b(gra.BUF_TETRAD2), b(gra.BUF_TETRAD3), b(gra.BUF_CFG), b(gra.BUF_DFG), 0, 1))
check(lib.gr_device_synchronize(0))
rays = DeviceBuffer(0, slots * 96)
cost = DeviceBuffer.from_numpy(0, np.zeros(n, dtype=np.uint32))
order = DeviceBuffer.from_numpy(0, np.zeros(n, dtype=np.uint32))
work = DeviceBuffer(0, (n + 128) * 4)


import ctypes as C
hip = C.CDLL("libamdhip64.so")
def as_int(p):
    return p.value if hasattr(p, "value") else int(p)
def reset():
    hip.hipMemcpy(C.c_void_p(as_int(rays.ptr)), C.c_void_p(as_int(rays0.ptr)), C.c_size_t(slots * 96), 3)

def run(label, f, repeat=4):
    ms = []
    for _ in range(repeat):
        reset(); torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s = torch.cuda.current_stream()
        a.record(s)
        f()
        e.record(s); torch.cuda.synchronize()
        ms.append(a.elapsed_time(e))
    print(f"{label:64s} {np.mean(ms[1:]):.3f} ms  ({', '.join('%.3f' % m for m in ms)})", flush=True)

null = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
run("gr_do_generic_rays, slot order (one workgroup per tile)", lambda: check(lib.gr_do_generic_rays(prog.handle, null, rays.ptr, count.ptr, slots, None, None, b(gra.BUF_CFG), b(gra.BUF_DFG), w, h, 0, 0, None, None, 0, None)))
run("gr_do_generic_rays_scheduled, slot order", lambda: check(lib.gr_do_generic_rays_scheduled(prog.handle, null, rays.ptr, count.ptr, n, b(gra.BUF_CFG), b(gra.BUF_DFG), None, None, cost.ptr)))
check(lib.gr_sort_tiles_by_cost(prog.handle, null, cost.ptr, tiles_x, tiles_y, order.ptr, work.ptr))
torch.cuda.synchronize()
c = cost.to_numpy(np.uint32, (n,)); o = order.to_numpy(np.uint32, (n,))
print("costs: max", c.max(), "median of traced", np.median(c[c > 0]), "tiles traced", (c > 0).sum(), "first of the list", c[o[:8]], "last", c[o[-4:]])
run("gr_do_generic_rays_scheduled, dearest first", lambda: check(lib.gr_do_generic_rays_scheduled(prog.handle, null, rays.ptr, count.ptr, n, b(gra.BUF_CFG), b(gra.BUF_DFG), None, order.ptr, None)))
run("gr_sort_tiles_by_cost", lambda: check(lib.gr_sort_tiles_by_cost(prog.handle, null, cost.ptr, tiles_x, tiles_y, order.ptr, work.ptr)))
