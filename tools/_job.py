import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import geodesic_raytracing_amd as gra
from gpu_stages import load_golden
from test_gpu_parity import _frame
NAME = "soak/double_kerr_near_extreme_61_167"
meta, z = load_golden(NAME)
for extra in ("", " -DGR_REFINED_RECIPROCALS", " -cl-fp32-correctly-rounded-divide-sqrt"):
    for mode, label in ((gra.MODE_FUSED, "fused"), (gra.MODE_REFERENCE, "reference")):
        px, _ = _frame(meta, mode, substituted=True, extra_arguments=extra)
        d = px[..., :3] - z["pixels"][..., :3]
        bad = ~(np.abs(d).max(axis=2) <= 1e-3)
        print(f"{extra or ' (default)':45s} {label:10s} pixels off {int(bad.sum()):4d} of {bad.size} ({bad.mean()*100:.2f} %, limit 0.50 %)  masked RMSE {float(np.sqrt((d[~bad]**2).mean())):.2e}", flush=True)
