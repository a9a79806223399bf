"""The three polar-axis cases (tests/golden/polar): pixels off by > 1e-3 against the reference's pixels - GPU (fused kernel, dynamic and
substituted program) and CPU restatement - and, for the GPU, which sky coordinate differs.  Honours GR_EXTRA_FLAGS."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import geodesic_raytracing_amd as gra
from gpu_stages import load_golden, metric_for
from test_gpu_parity import _frame
from test_oracle import run_oracle
from oracle import build_restate
from geodesic_raytracing_amd.pipeline import RENDER_DATA_DTYPE, download

print("flags:", os.environ.get("GR_EXTRA_FLAGS", ""))
for name in ["kerr_newman_axis_13_3", "kerr_axis_14_212", "kerr_newman_axis_14_593", "kerr_axis_21_122", "kerr_axis_22_142", "kerr_newman_axis_23_63",
             "kerr_newman_axis_41_4", "kerr_newman_axis_41_48", "kerr_newman_axis_41_114"]:
    meta, z = load_golden(os.path.join("polar", name))
    r = run_oracle(build_restate.build(metric_for(meta).argument_string()), meta)
    cpu_bad = (np.abs(r["pixels"][..., :3] - z["pixels"][..., :3]).max(axis=2) > 1e-3)
    line = f"{name:26s} cpu {cpu_bad.sum():3d}"
    for sub in (False, True):
        px, state = _frame(meta, gra.MODE_FUSED, substituted=sub)
        rd = download(0, state.buffer(gra.BUF_RENDER_DATA), RENDER_DATA_DTYPE, meta["width"] * meta["height"])
        bad = np.abs(px[..., :3] - z["pixels"][..., :3]).max(axis=2) > 1e-3
        gd = z["render_data"]
        ok = (rd["terminated"] == 1) & (gd["terminated"] == 1)
        te = np.abs(rd["tex_coord"] - gd["tex_coord"])
        te = np.minimum(te, 1 - te)
        line += f" | {'sub' if sub else 'dyn'} {bad.sum():3d} (in cpu set {(bad & cpu_bad).sum():3d}) phi>1e-3 {(te[ok][:, 0] > 1e-3).sum():3d} theta>1e-3 {(te[ok][:, 1] > 1e-3).sum():3d} flags {(rd['terminated'] != gd['terminated']).sum()}"
    print(line, flush=True)
