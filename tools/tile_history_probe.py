"""One frame at a time (the interactive use: nothing known about the next camera), tiles in image order against tiles handed out
dearest first by what they cost in the frame before (gr_frame_options.tile_history): ms per frame with a still camera and with one that
moves a little every frame, and that the pixels do not depend on it.
usage: python tools/tile_history_probe.py [a ...]        (Kerr spin parameters, default 0.45 0.9)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import geodesic_raytracing_amd as gra


def quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return [aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw,
            aw * bw - ax * bx - ay * by - az * bz]

W, H = (int(x) for x in os.environ.get("TILE_HISTORY_SIZE", "3840x2160").split("x"))
bg_np, levels = gra.pack_background(gra.synthetic_background(4096, 2048))
bg = torch.from_numpy(bg_np).cuda()
out = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
WORKLOADS = {"kerr": ("kerr_boyer", None), "double_kerr": ("double_unequal_kerr", {}), "alcubierre": ("alcubierre", {})}


def run(name, params, label, inflight=1):
    metric = gra.Metric(name, os.path.join(ROOT, "geodesic_raytracing_amd", "scripts"))
    cfg = metric.cfg_values(**params)
    feats = metric.features(adaptive_sampling=0)
    program = gra.pipeline.ProgramManager(metric, 0, feats, cfg).current(wait=True)
    pictures = {}
    for moving in (False, True):
        for history in [int(x) for x in os.environ.get("TILE_HISTORY_MODES", "0,-1").split(",")]:   # -1: the library's default (the order when the frame finds the device idle)
            states = [gra.RenderState(W, H, 0) for _ in range(inflight)]
            streams = [torch.cuda.Stream() for _ in range(inflight)]
            times = []
            n = int(os.environ.get('TILE_HISTORY_FRAMES', '12')) if inflight == 1 else 60

            def frame(k):
                # the camera of tools' default pose, pushed sideways by 0.02 M per frame when it moves (a brisk walk at 60 fps)
                camera = gra.default_camera()
                camera.position[1] += float(os.environ.get("TILE_HISTORY_OFFSET", "0"))   # where the still camera stands (a moving one starts)
                if moving:   # TILE_HISTORY_SPEED: sideways, distance units per frame; TILE_HISTORY_TURN: degrees per frame about the view's up axis
                    camera.position[1] += float(os.environ.get("TILE_HISTORY_SPEED", "0.02")) * k
                    turn = np.radians(float(os.environ.get("TILE_HISTORY_TURN", "0"))) * k
                    if turn:
                        q = [camera.quat[i] for i in range(4)]
                        camera.quat = (gra.c_float * 4)(*quat_mul([0.0, np.sin(turn / 2), 0.0, np.cos(turn / 2)], q))
                o = gra.frame_options(mode=gra.MODE_FUSED, tile_history=history)
                states[k % inflight].render(program, metric, camera, out.data_ptr(), (bg.data_ptr(), 4096, 2048, levels), feats, cfg, o,
                                            streams[k % inflight].cuda_stream)
            if inflight == 1:
                for k in range(n):
                    torch.cuda.synchronize()
                    t = time.perf_counter()
                    frame(k)
                    torch.cuda.synchronize()
                    times.append((time.perf_counter() - t) * 1e3)
                pictures[(moving, history)] = out.clone()
                if os.environ.get("TILE_HISTORY_CHECKSUM"):   # (to compare the last frame across processes: builds, environment switches)
                    print(f"{label:18s} camera {'moving' if moving else 'still '} checksum of the last frame {int(out.view(torch.int32).to(torch.int64).sum().item())}", flush=True)
                print(f"{label:18s} camera {'moving' if moving else 'still '} history {history:2d}: first {times[0]:6.2f} ms, then "
                      f"{sum(times[2:]) / len(times[2:]):6.2f} ms/frame  (min {min(times[2:]):6.2f}, max {max(times[2:]):6.2f})"
                      + ("  " + " ".join(f"{t:.1f}" for t in times[2:]) if os.environ.get("TILE_HISTORY_VERBOSE") else ""), flush=True)
            else:
                for k in range(2 * inflight):
                    frame(k)
                torch.cuda.synchronize()
                t = time.perf_counter()
                for k in range(n):
                    frame(2 * inflight + k)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t) / n * 1e3
                print(f"{label:18s} {inflight} in flight, camera {'moving' if moving else 'still '} history {history:2d}: {ms:6.2f} ms/frame", flush=True)
        if inflight == 1 and len([k for k in pictures if k[0] == moving]) >= 2:
            a, b = list(pictures.values())[-2:]
            same = torch.equal(a, b)
            print(f"{label:18s} camera {'moving' if moving else 'still '}: pixels identical with and without the history order: {same}", flush=True)
            assert same


args = sys.argv[1:] or ["0.45", "0.9"]
for a in args:
    if a in WORKLOADS:
        run(WORKLOADS[a][0], {}, a)
    else:
        run("kerr_boyer", {"a": float(a)}, f"kerr a={a}")
        if os.environ.get("TILE_HISTORY_INFLIGHT"):
            run("kerr_boyer", {"a": float(a)}, f"kerr a={a}", inflight=int(os.environ["TILE_HISTORY_INFLIGHT"]))
