"""Headless renderer: metric script -> PNG (the role of the reference's `-start <metric>` + screenshot key).

    python -m geodesic_raytracing_amd.render --metric kerr_boyer --cfg a=0.45 --size 1920x1080 --out kerr.png
    python -m geodesic_raytracing_amd.render --metric alcubierre --redshift --camera 0,0,-6,0.5 --background sky.png --out warp.png
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
           background=None, device=0, fov=90.0, universe=20.0, wait_for_static=True):
    """Returns the linear-light float32 frame [H, W, 4]."""
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
    opts = gra.frame_options(mode=gra.MODE_REFERENCE if adaptive else gra.MODE_FUSED)
    state.render(program, metric, cam, out.ptr, (dbg.ptr, packed.shape[2], packed.shape[1], levels), feats, cfg_values, opts)
    state.synchronize()
    return out.to_numpy(np.float32, (height, width, 4))


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
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--out", required=True)
    a = ap.parse_args(argv)
    w, h = (int(v) for v in a.size.lower().split("x"))
    cfg = {k: float(v) for k, v in (kv.split("=") for kv in a.cfg)}
    frame = render(a.metric, w, h, a.scripts, cfg, [float(v) for v in a.camera.split(",")] if a.camera else None,
                   [float(v) for v in a.quat.split(",")] if a.quat else None, a.redshift, a.adaptive,
                   read_png(a.background) if a.background else None, a.device, a.fov, a.universe)
    write_frame_png(a.out, frame)
    print(f"wrote {a.out} ({w}x{h})")


if __name__ == "__main__":
    sys.exit(main())
