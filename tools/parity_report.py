"""Per-stage error statistics of the HIP kernels against the golden vectors (run on a GPU box)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import geodesic_raytracing_amd as gra
from gpu_stages import Stages, backgrounds, load_golden, golden_names, circ_diff, rel_err

def pct(x, q): return float(np.percentile(x, q)) if len(x) else 0.0

for name in golden_names():
    meta, z = load_golden(name)
    st = Stages(meta)
    rep = {"case": name}
    cam, tet = st.camera()
    rep["camera_abs"] = float(np.abs(cam - z["camera_generic"]).max()); rep["tetrad_abs"] = float(np.abs(tet - z["tetrad"]).max())
    if meta["prepass"] or meta["features"].get("adaptive_sampling"):
        print(json.dumps(rep)); continue
    gi = z["rays_init"]
    ri = st.init_rays(z["camera_generic"], z["tetrad"])
    for f in ["position", "velocity", "acceleration", "initial_quat"]:
        rep["init_" + f] = float(np.abs(ri[f] - gi[f]).max())
    rep["init_ku"] = float(np.abs(ri["ku_uobsu"] - gi["ku_uobsu"]).max())
    g = z["rays"]
    r, att = st.trace(gi, True)
    rep["attempts_per_ray"] = att / len(gi)
    rep["term_mismatch"] = float((r["terminated"] != g["terminated"]).mean())
    both = (r["terminated"] == 1) & (g["terminated"] == 1)
    pe = rel_err(r["position"][both], g["position"][both]).max(axis=1); ve = rel_err(r["velocity"][both], g["velocity"][both]).max(axis=1)
    rep["trace_pos_p50"], rep["trace_pos_p90"], rep["trace_pos_p99"], rep["trace_pos_max"] = pct(pe,50), pct(pe,90), pct(pe,99), pct(pe,100)
    rep["trace_vel_p90"], rep["trace_vel_p99"] = pct(ve,90), pct(ve,99)
    grd = z["render_data"]
    rd = st.render_data(g)
    ok = (rd["terminated"] == 1) & (grd["terminated"] == 1)
    rep["rd_term_mismatch"] = float((rd["terminated"] != grd["terminated"]).mean())
    te = circ_diff(rd["tex_coord"][ok], grd["tex_coord"][ok]).max(axis=1)
    rep["rd_tex_p99"], rep["rd_tex_max"] = pct(te,99), pct(te,100)
    rep["rd_z_max"] = float(np.abs(rd["z_shift"][ok] - grd["z_shift"][ok]).max()) if ok.any() else 0
    rep["rd_side_mismatch"] = int((rd["side"][ok] != grd["side"][ok]).sum())
    bg, bg2, levels = backgrounds(meta)
    px = st.render(grd, bg, bg2, levels)
    d = px[..., :3] - z["pixels"][..., :3]
    rep["render_rmse"] = float(np.sqrt((d**2).mean())); rep["render_max"] = float(np.abs(d).max())
    # end to end
    r2 = st.trace(ri); rd2 = st.render_data(r2); px2 = st.render(rd2, bg, bg2, levels)
    d2 = px2[..., :3] - z["pixels"][..., :3]
    rep["e2e_rmse"] = float(np.sqrt((d2**2).mean())); rep["e2e_max"] = float(np.abs(d2).max()); rep["e2e_bad_1e-3"] = float((np.abs(d2).max(axis=2) > 1e-3).mean())
    print(json.dumps({k: (round(v, 8) if isinstance(v, float) else v) for k, v in rep.items()}), flush=True)
