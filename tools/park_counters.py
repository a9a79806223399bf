"""GPU box: the lot's counters and the launch time of gr_trace_fused_parking launched directly (image order), 4K Kerr.
usage: python tools/park_counters.py <a> "lanes,trips;lanes,trips;..." """
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import geodesic_raytracing_amd as gra
from geodesic_raytracing_amd import check, lib
from geodesic_raytracing_amd.pipeline import DeviceBuffer, RENDER_DATA_DTYPE
import test_gpu_parking as tp

W, H = 3840, 2160
a = float(sys.argv[1])
settings = [tuple(int(v) for v in t.split(",")) for t in sys.argv[2].split(";")]
prog, state = tp.traced_state(a, width=W, height=H)
b = state.buffer
rd = DeviceBuffer.from_numpy(0, np.zeros(W * H, dtype=RENDER_DATA_DTYPE))


def launch(parking=None):
    counters = DeviceBuffer.from_numpy(0, np.zeros(512, dtype=np.uint64))
    extra = dict(parking=parking) if parking is not None else {}
    args = gra.TraceFusedArgs(camera_generic=b(gra.BUF_CAMERA_GENERIC), camera_quat=b(gra.BUF_CAMERA_QUAT), render_data=rd.ptr, width=W, height=H,
                              block_rows=0, strip_rank=0, strip_count=1, termination_buffer=b(gra.BUF_TERMINATION), prepass_width=W // 16,
                              prepass_height=H // 16, e0=b(gra.BUF_TETRAD0), e1=b(gra.BUF_TETRAD1), e2=b(gra.BUF_TETRAD2), e3=b(gra.BUF_TETRAD3),
                              cfg=b(gra.BUF_CFG), dfg=b(gra.BUF_DFG), attempt_counter=counters.ptr, **extra)
    best = 1e9
    every = []
    for rep in range(6):
        check(lib.gr_device_synchronize(0))
        t = time.perf_counter()
        check(lib.gr_trace_fused_launch(prog.handle, None, ctypes.byref(args)))
        check(lib.gr_device_synchronize(0))
        best = min(best, time.perf_counter() - t)
        every.append(round((time.perf_counter() - t) * 1e3, 1))
    print("     launches:", every)
    c = counters.to_numpy(np.uint64, (512,))
    return best * 1e3, c


ms, c = launch()
print(f"a={a} plain: {ms:.2f} ms, waves {int(c[3]) // 6}, mean wave lifetime {c[2] / max(c[3], 1) / 100:.1f} us at {c[1] / max(c[2], 1) * 100:.0f} MHz", flush=True)
for lanes, trips in settings:
    lot = tp.Lot(W * H // 2, W * H // 2, lanes, trips)
    ms, c = launch(lot.arg)
    k = lot.counters()
    print(f"a={a} lanes {lanes} trips {trips}: {ms:.2f} ms, mean wave lifetime {c[2] / max(c[3], 1) / 100:.1f} us, {k}, rays per claim {k['rays_handed'] / max(k['waves'], 1):.1f}", flush=True)
    if c[6]:   # -DGR_PROBE_LOT (three launches accumulated)
        print(f"     visits of parked rays: {int(c[6]) // 6}, mean {c[4] / c[6] / 100:.1f} us and {c[5] / c[6]:.0f} attempts: {c[4] / max(c[5], 1) * 10:.1f} ns per attempt of a visit "
              f"(all waves of the launch: {c[2] * 10 / max(float(c[0] + c[256:].sum()) / 64, 1):.1f} ns per 64 lane-attempts)", flush=True)
