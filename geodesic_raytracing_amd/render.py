"""Headless renderer: metric script -> PNG (the role of the reference's `-start <metric>` + screenshot key).

    python -m geodesic_raytracing_amd.render --metric kerr_boyer --cfg a=0.45 --size 1920x1080 --out kerr.png
    python -m geodesic_raytracing_amd.render --metric alcubierre --redshift --camera 0,0,-6,0.5 --background sky.png --out warp.png
    # camera riding its own timelike geodesic: 24 frames, 0.5 units of proper time apart -> fall_000.png .. fall_023.png
    python -m geodesic_raytracing_amd.render --metric schwarzschild --camera 0,0,-8,0 --geodesic-speed 0,0.3,0 \
        --geodesic-time 0 --geodesic-dt 0.5 --frames 24 --out fall.png
"""
import argparse
import ctypes
import os
import sys

import numpy as np

import geodesic_raytracing_amd as gra
from geodesic_raytracing_amd.pipeline import DeviceBuffer, ProgramManager

HERE = os.path.dirname(os.path.abspath(__file__))


def read_png(path):
    w, h = ctypes.c_int(), ctypes.c_int()
    gra.check(gra.lib.gr_read_png_rgba8(path.encode(), ctypes.byref(w), ctypes.byref(h), None, 0))
    out = np.empty((h.value, w.value, 4), dtype=np.uint8)
    gra.check(gra.lib.gr_read_png_rgba8(path.encode(), ctypes.byref(w), ctypes.byref(h), out.ctypes.data_as(ctypes.c_void_p), out.nbytes))
    return out


def write_frame_png(path, frame):
    frame = np.ascontiguousarray(frame, dtype=np.float32)
    h, w = frame.shape[:2]
    gra.check(gra.lib.gr_write_frame_png(path.encode(), frame.ctypes.data_as(ctypes.c_void_p), w, h))


def render(metric_name, width, height, scripts=None, cfg=None, camera_pos=None, camera_quat=None, redshift=False, adaptive=False,
           background=None, device=0, fov=90.0, universe=20.0, wait_for_static=True, geodesic_speed=None, geodesic_times=None,
           parallel_transport=True):
    """Returns the linear-light float32 frame [H, W, 4]; with geodesic_speed (camera on its own timelike geodesic,
    main.cpp:2675-2760) a list of frames, one per entry of geodesic_times (proper time along the path)."""
    metric = gra.Metric(metric_name, scripts or os.path.join(HERE, "scripts"))
    feats = metric.features(adaptive_sampling=int(adaptive), redshift=int(redshift), field_of_view=fov, universe_size=universe)
    cfg_values = metric.cfg_values(**(cfg or {}))
    manager = ProgramManager(metric, device, feats, cfg_values)
    program = manager.current(wait=wait_for_static)
    state = gra.RenderState(width, height, device)
    rgba = background if background is not None else gra.synthetic_background(2048, 1024)
    packed, levels = gra.pack_background(rgba)
    dbg = DeviceBuffer.from_numpy(device, packed)
    out = DeviceBuffer(device, width * height * 16)
    cam = gra.default_camera(camera_pos, camera_quat)
    mode = gra.MODE_REFERENCE if adaptive else gra.MODE_FUSED
    bg = (dbg.ptr, packed.shape[2], packed.shape[1], levels)
    if geodesic_speed is None:
        state.render(program, metric, cam, out.ptr, bg, feats, cfg_values, gra.frame_options(mode=mode))
        state.synchronize()
        return out.to_numpy(np.float32, (height, width, 4))
    gc = gra.GeodesicCamera(device=device)
    steps, tau = gc.snapshot(program, metric, cam, geodesic_speed, feats, cfg_values)
    print(f"geodesic snapshot: {steps} samples covering {tau:.3f} of proper time", file=sys.stderr)
    times = list(geodesic_times or [0.0])
    frames = []
    for i, t in enumerate(times):
        ahead = i + 1 < len(times) and mode == gra.MODE_FUSED
        opts = gra.frame_options(mode=mode, geodesic=gc.handle.value, geodesic_time=t, parallel_transport_observer=int(parallel_transport),
                                 next_camera=ctypes.pointer(cam) if ahead else None, next_geodesic_time=times[i + 1] if ahead else 0.0)
        state.render(program, metric, cam, out.ptr, bg, feats, cfg_values, opts)
        state.synchronize()
        frames.append(out.to_numpy(np.float32, (height, width, 4)))
    return frames


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--metric", required=True)
    ap.add_argument("--scripts", default=None, help="scripts folder (default: the one shipped with the package)")
    ap.add_argument("--size", default="1920x1080")
    ap.add_argument("--cfg", action="append", default=[], help="NAME=VALUE, a $cfg parameter of the metric (repeatable)")
    ap.add_argument("--camera", default=None, help="t,x,y,z")
    ap.add_argument("--quat", default=None, help="x,y,z,w")
    ap.add_argument("--fov", type=float, default=90.0)
    ap.add_argument("--universe", type=float, default=20.0)
    ap.add_argument("--redshift", action="store_true")
    ap.add_argument("--adaptive", action="store_true", help="quarter-resolution primary rays + refinement (reference mode)")
    ap.add_argument("--background", default=None, help="equirectangular PNG (default: synthetic grid + stars)")
    ap.add_argument("--geodesic-speed", default=None, help="vx,vy,vz (|v| < 1, camera tetrad frame): ride the timelike geodesic "
                    "launched from --camera with this speed")
    ap.add_argument("--geodesic-time", type=float, default=0.0, help="proper time of the first frame")
    ap.add_argument("--geodesic-dt", type=float, default=0.5, help="proper time between frames")
    ap.add_argument("--frames", type=int, default=1)
    ap.add_argument("--recompute-tetrads", action="store_true", help="rebuild the tetrad at every point instead of parallel transport")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--out", required=True)
    a = ap.parse_args(argv)
    w, h = (int(v) for v in a.size.lower().split("x"))
    cfg = {k: float(v) for k, v in (kv.split("=") for kv in a.cfg)}
    speed = [float(v) for v in a.geodesic_speed.split(",")] if a.geodesic_speed else None
    times = [a.geodesic_time + i * a.geodesic_dt for i in range(max(a.frames, 1))]
    result = render(a.metric, w, h, a.scripts, cfg, [float(v) for v in a.camera.split(",")] if a.camera else None,
                    [float(v) for v in a.quat.split(",")] if a.quat else None, a.redshift, a.adaptive,
                    read_png(a.background) if a.background else None, a.device, a.fov, a.universe, geodesic_speed=speed,
                    geodesic_times=times, parallel_transport=not a.recompute_tetrads)
    if speed is None:
        write_frame_png(a.out, result)
        print(f"wrote {a.out} ({w}x{h})")
        return 0
    stem, ext = os.path.splitext(a.out)
    for i, frame in enumerate(result):
        path = a.out if len(result) == 1 else f"{stem}_{i:03d}{ext}"
        write_frame_png(path, frame)
        print(f"wrote {path} ({w}x{h}, proper time {times[i]:.3f})")
    return 0


if __name__ == "__main__":
    sys.exit(main())
