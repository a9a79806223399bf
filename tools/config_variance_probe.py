"""Why did BENCH_r05 record configs[3] (double_unequal_kerr 4K) at 74.6 fps when its own trace launch is 10.6 ms?
Replays what bench.py's secondary block did for that configuration - program + render state + output made on the spot, ONE warm-up frame,
three frames timed back to back - and then times frames one by one, so that the per-frame numbers show which of them are slow and why:
frame 0 (buffers touched, kernels loaded, no cost history: tiles in image order), frame 1 (the first that follows a history), the
shader clock after a host-side pause (program load, allocation: the device idles and clocks down).  PYTHONPATH=. python tools/config_variance_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import geodesic_raytracing_amd as gra

scripts = os.path.join(ROOT, "geodesic_raytracing_amd", "scripts")
bg_np, levels = gra.pack_background(gra.synthetic_background(4096, 2048))
bg = torch.from_numpy(bg_np).cuda()
stream = torch.cuda.current_stream().cuda_stream


def make(name, w, h, cam, fk):
    m = gra.Metric(name, scripts)
    f = m.features(adaptive_sampling=0, **fk)
    p = gra.Program(m.argument_string(features=f, static=True, cfg_values=m.cfg_values()), 0)
    st = gra.RenderState(w, h, 0)
    out = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    c = gra.default_camera(cam)

    def once(**kw):
        st.render(p, m, c, out.data_ptr(), (bg.data_ptr(), 4096, 2048, levels), f, m.cfg_values(), gra.frame_options(mode=gra.MODE_FUSED, **kw), stream)
    return once, st


def r05_way(once):
    once()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        once()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / 3 * 1e3


def one_by_one(once, st, n):
    rows = []
    for _ in range(n):
        t = time.perf_counter()
        once(time_kernels=1, count_attempts=1)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t) * 1e3
        s = st.stage_ms()
        rows.append((wall, s["prepass"], s["trace"], s["render"], st.shader_clock_mhz()))
    return rows


for name, (w, h), cam, fk in (("double_unequal_kerr", (3840, 2160), [0, 0, -6, 0.5], {}), ("alcubierre", (7680, 4320), [0, 0, -6, 0.5], {"redshift": 1}),
                               ("schwarzschild", (1920, 1080), None, {})):
    print(f"== {name} {w}x{h}")
    for idle in (0.0, 1.0):
        once, st = make(name, w, h, cam, fk)
        time.sleep(idle)
        print(f"  fresh state, {idle:.0f} s idle, then bench.py r05's way (1 warm-up + 3 timed, back to back): {r05_way(once):8.3f} ms per frame")
        del once, st
    once, st = make(name, w, h, cam, fk)
    rows = one_by_one(once, st, 14)
    print("  fresh state, frames one by one:  wall ms | prepass | trace | render | shader MHz")
    for i, r in enumerate(rows):
        print(f"    frame {i:2d}: {r[0]:8.3f} | {r[1]:6.3f} | {r[2]:7.3f} | {r[3]:6.3f} | {r[4]:7.1f}")
    time.sleep(1.0)
    rows = one_by_one(once, st, 6)
    print("  the same state after 1 s of idling:")
    for i, r in enumerate(rows):
        print(f"    frame {i:2d}: {r[0]:8.3f} | {r[1]:6.3f} | {r[2]:7.3f} | {r[3]:6.3f} | {r[4]:7.1f}")
    # the new rule: 3 warm-up frames, 12 timed, three repeats
    reps = []
    for _ in range(3):
        for _ in range(3):
            once()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(12):
            once()
        torch.cuda.synchronize()
        reps.append((time.perf_counter() - t) / 12 * 1e3)
    print(f"  3 warm-up + 12 timed, three repeats: {['%.3f' % x for x in reps]} ms per frame, median {np.median(reps):.3f}")
    del once, st
