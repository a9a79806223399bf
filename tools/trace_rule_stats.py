"""Trace-stage parity statistics per golden case (GPU): how many ordinary rays (fewer than twice the median attempts) miss the 1e-3
position rule, worst error, attempts against the oracle.  Honour GR_EXTRA_FLAGS (e.g. -DGR_LIBM_TRIG) for A/B runs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from gpu_stages import Stages, load_golden, golden_names, ordinary_rays, position_err

print("flags:", os.environ.get("GR_EXTRA_FLAGS", ""))
for name in golden_names():
    meta, z = load_golden(name)
    if meta["prepass"] or meta["features"].get("adaptive_sampling"):
        continue
    st = Stages(meta)
    got, att = st.trace(z["rays_init"], True)
    want = z["rays"]
    ordinary = ordinary_rays(meta, z)
    both = (got["terminated"] == 1) & (want["terminated"] == 1)
    err = position_err(got["position"], want["position"]).max(axis=1)
    sel = both & ordinary
    print(f"{name:28s} flags_differ {(got['terminated'] != want['terminated']).sum():3d} ordinary {sel.sum():4d}/{both.sum():4d} "
          f"over1e-3 {(err[sel] > 1e-3).sum():3d} max {err[sel].max() if sel.any() else 0:.2e} p90all {np.percentile(err[both], 90) if both.any() else 0:.2e} "
          f"attempts/ray {att / len(want):.1f}", flush=True)
