"""scratch probe (GPU): the near-extreme double Kerr soak frame stage by stage (where does the redshift error come from)"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import geodesic_raytracing_amd as gra
from gpu_stages import Stages, load_golden, metric_for, circ_diff, position_err
from test_gpu_parity import _frame, background
np.set_printoptions(linewidth=200, precision=7, suppress=False)
def cmp(a, b):
    d = a[..., :3] - b[..., :3]; bad = ~(np.abs(d).max(axis=2) <= 1e-3)
    return int(bad.sum()), float(np.sqrt((d[~bad] ** 2).mean()))
name = "soak/double_kerr_near_extreme_61_167"
meta, z = load_golden(name)
st = Stages(meta)
bg, levels = background(meta)
pw, ph = meta["width"] // 16, meta["height"] // 16
ri = st.init_rays(z["camera_generic"], z["tetrad"], termination=z["termination"].reshape(-1), prepass_size=(pw, ph)); want = z["rays_init"]
for f in ("position", "velocity", "acceleration", "initial_quat", "ku_uobsu"):
    print("  init", f, np.abs(ri[f] - want[f]).max(), "scale", np.abs(want[f]).max())
tr = st.trace(z["rays_init"]); b = (tr["terminated"] == 1) & (z["rays"]["terminated"] == 1)
e = position_err(tr["position"], z["rays"]["position"]).max(axis=1)[b]
ev = position_err(tr["velocity"], z["rays"]["velocity"]).max(axis=1)[b]
print("  trace from golden init: flags differ", (tr["terminated"] != z["rays"]["terminated"]).sum(), "pos err pct 50/90/99/100", np.percentile(e, [50, 90, 99, 100]), "vel err", np.percentile(ev, [50, 90, 99, 100]))
print("  running_dlambda err", np.percentile(np.abs(tr["running_dlambda_dnew"][b] / z["rays"]["running_dlambda_dnew"][b] - 1), [50, 90, 99, 100]))
wrd = z["render_data"]; ok = wrd["terminated"] == 1
rd = st.render_data(z["rays"])
dt = circ_diff(rd["tex_coord"][ok], wrd["tex_coord"][ok]); dz = np.abs(rd["z_shift"][ok] - wrd["z_shift"][ok])
print("  render_data from golden rays: tex err 50/99/100", np.percentile(dt, [50, 99, 100]), "z err 50/99/100", np.percentile(dz, [50, 99, 100]), "z range", wrd["z_shift"][ok].min(), wrd["z_shift"][ok].max())
px = st.render(z["render_data"], bg, levels, meta["max_probes"]); print("  render from golden rd:", cmp(px, z["pixels"]))
rd2 = st.render_data(tr); ok2 = ok & (rd2["terminated"] == 1)
dt = circ_diff(rd2["tex_coord"][ok2], wrd["tex_coord"][ok2]); dz = np.abs(rd2["z_shift"][ok2] - wrd["z_shift"][ok2])
print("  after gpu trace: tex err 50/99/100", np.percentile(dt, [50, 99, 100]), "z err 50/99/100", np.percentile(dz, [50, 90, 99, 100]))
px = st.render(rd2, bg, levels, meta["max_probes"]); print("  golden init -> trace -> rd -> render:", cmp(px, z["pixels"]))
# which of tex / z owns the pixel error: mix golden and gpu fields
mix = z["render_data"].copy(); mix["z_shift"] = rd2["z_shift"]
print("  render(golden tex, gpu z):", cmp(st.render(mix, bg, levels, meta["max_probes"]), z["pixels"]))
mix = z["render_data"].copy(); mix["tex_coord"] = rd2["tex_coord"]
print("  render(gpu tex, golden z):", cmp(st.render(mix, bg, levels, meta["max_probes"]), z["pixels"]))
# CPU restatement for comparison
from oracle import build_restate
from oracle.refpipe import OraclePipeline, pack_features
pipe = OraclePipeline(build_restate.build(metric_for(meta).argument_string()))
cpu = pipe.frame(meta["width"], meta["height"], meta["cfg"], pack_features(**meta["features"]), camera_pos=meta["camera_pos"], camera_quat=meta["camera_quat"], background=(bg, levels), basis_speed=meta["basis_speed"], use_prepass=True)
crd = cpu["render_data"]; okc = ok & (crd["terminated"] == 1)
print("  CPU restatement: tex err", np.percentile(circ_diff(crd["tex_coord"][okc], wrd["tex_coord"][okc]), [50, 99, 100]), "z err", np.percentile(np.abs(crd["z_shift"][okc] - wrd["z_shift"][okc]), [50, 90, 99, 100]), "pixels", cmp(cpu["pixels"], z["pixels"]))
