"""End-to-end pixel statistics per golden case (GPU, fused kernel, substituted program): fraction of pixels off by > 1e-3 and the
RMSE of the rest, against the reference's pixels.  Honours GR_EXTRA_FLAGS for A/B runs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import geodesic_raytracing_amd as gra
from gpu_stages import load_golden, golden_names
from test_gpu_parity import _frame

print("flags:", os.environ.get("GR_EXTRA_FLAGS", ""))
tot_bad = 0
for name in golden_names():
    meta, z = load_golden(name)
    if meta["features"].get("adaptive_sampling"):
        continue
    for sub in (False, True):
        px, _ = _frame(meta, gra.MODE_FUSED, substituted=sub)
        d = px[..., :3] - z["pixels"][..., :3]
        bad = np.abs(d).max(axis=2) > 1e-3
        tot_bad += int(bad.sum())
        print(f"{name:28s} {'substituted' if sub else 'dynamic    '} bad {bad.sum():4d} ({bad.mean() * 100:.2f} %) rmse {np.sqrt((d[~bad] ** 2).mean()):.2e}", flush=True)
print("total bad pixels", tot_bad)
