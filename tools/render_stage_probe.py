"""The render stage alone on a fixture's golden render-data (GPU box): the pixels that differ most from the reference's, and for each
the footprint the reference's arithmetic gives it in float64 (probe count wanted, cap, level of detail) - is the pixel on a rounding
tie of the probe count (cl.cl:5597-5613)?    PYTHONPATH=. python tools/render_stage_probe.py <fixture> [<fixture> ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_stages import Stages, backgrounds, load_golden  # noqa: E402


def wrapped(a, b):
    d = (a - b) * 2 * np.pi
    d = np.where(np.abs(d) <= np.pi, d, d - 2 * np.pi * np.rint(d / (2 * np.pi)))
    return d / (2 * np.pi)


def footprint(rd, w, h, bw, bh):
    """wanted probes (2 long / short - 1) per pixel in float64, from the records' texture coordinates"""
    t = rd["tex_coord"].reshape(h, w, 2).astype(np.float64)
    beside = np.concatenate([t[:, 1:], t[:, -2:-1]], axis=1)
    below = np.concatenate([t[1:], t[-2:-1]], axis=0)
    sx = np.ones((h, w)); sx[:, -1] = -1
    sy = np.ones((h, w)); sy[-1, :] = -1
    ax = np.stack([wrapped(t[..., 0], beside[..., 0]) * sx * bw, wrapped(t[..., 1], beside[..., 1]) * sx * bh], -1) / 1.3
    ay = np.stack([wrapped(t[..., 0], below[..., 0]) * sy * bw, wrapped(t[..., 1], below[..., 1]) * sy * bh], -1) / 1.3
    A = ax[..., 1] ** 2 + ay[..., 1] ** 2 + 1
    B = -2 * (ax[..., 0] * ax[..., 1] + ay[..., 0] * ay[..., 1])
    C = ax[..., 0] ** 2 + ay[..., 0] ** 2 + 1
    F = A * C - B * B / 4
    A, B, C = A / F, B / F, C / F
    root = np.sqrt((A - C) ** 2 + B * B)
    major = np.maximum(1 / np.sqrt((A + C - root) / 2), 1.0)
    minor = np.maximum(1 / np.sqrt((A + C + root) / 2), 1.0)
    major = np.maximum(major, minor)
    return 2 * major / minor - 1, major, minor


for name in sys.argv[1:]:
    meta, z = load_golden(name)
    bg, bg2, levels = backgrounds(meta)
    got = Stages(meta).render(z["render_data"], bg, bg2, levels, meta["max_probes"])
    d = np.abs(got[..., :3] - z["pixels"][..., :3]).max(axis=2)
    wanted, major, minor = footprint(z["render_data"], meta["width"], meta["height"], bg.shape[2], bg.shape[1])
    print(f"{name}: rmse {np.sqrt(((got[..., :3] - z['pixels'][..., :3]) ** 2).mean()):.2e} max {d.max():.2e} pixels > 2e-4: {(d > 2e-4).sum()}  max_probes {meta['max_probes']}")
    for idx in np.argsort(d.ravel())[::-1][:12]:
        y, x = divmod(int(idx), meta["width"])
        print(f"   ({x:3d},{y:3d}) diff {d[y, x]:.2e}  wanted+0.5 = {wanted[y, x] + 0.5:.5f}  major {major[y, x]:.3f} minor {minor[y, x]:.3f}  "
              f"lod {np.log2(minor[y, x]):.3f}  gpu {got[y, x, :3]}  ref {z['pixels'][y, x, :3]}")
