"""Which of the build's approximations owns a frame's distance from the reference: renders one golden frame (substituted program, fused path)
with the ray kernels rebuilt under each of a list of compiler-flag variants (GR_EXTRA_FLAGS; the set-up module is IEEE either way).
    python tools/flag_variants_probe.py precompile <fixture>     # build container: fills the code-object cache
    python tools/flag_variants_probe.py run <fixture>            # GPU box
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
VARIANTS = {
    "baseline": "",
    "no-approx-func": "-fno-approx-func",
    "ieee-div-sqrt": "-fhip-fp32-correctly-rounded-divide-sqrt",
    "no-reciprocal-math": "-fno-reciprocal-math",
    "no-approx+ieee+no-recip": "-fno-approx-func -fhip-fp32-correctly-rounded-divide-sqrt -fno-reciprocal-math",
    "no-contract": "-ffp-contract=off",
    "no-reassoc": "-fno-associative-math",
    "libm-trig": "-DGR_LIBM_TRIG",
    "all-exact": "-fno-approx-func -fhip-fp32-correctly-rounded-divide-sqrt -fno-reciprocal-math -ffp-contract=off -fno-associative-math -DGR_LIBM_TRIG",
}


def one(mode, fixture):
    import numpy as np
    import geodesic_raytracing_amd as gra
    from gpu_stages import load_golden, metric_for
    meta, z = load_golden(fixture)
    metric = metric_for(meta)
    feats = gra.default_features(**meta["features"])
    args = metric.argument_string(features=feats, static=True, cfg_values=meta["cfg"])
    if mode == "precompile":
        gra.Program.precompile(args)
        return
    from test_gpu_parity import _frame
    from gpu_stages import circ_diff
    px, state = _frame(meta, gra.MODE_FUSED, substituted=True, options=dict(count_attempts=1))
    d = px[..., :3] - z["pixels"][..., :3]
    bad = ~(np.abs(d).max(axis=2) <= 1e-3)
    from geodesic_raytracing_amd.pipeline import RENDER_DATA_DTYPE, download
    rd = download(0, state.buffer(gra.BUF_RENDER_DATA), RENDER_DATA_DTYPE, meta["width"] * meta["height"])
    ok = (rd["terminated"] == 1) & (z["render_data"]["terminated"] == 1)
    dt = circ_diff(rd["tex_coord"][ok], z["render_data"]["tex_coord"][ok])
    print(f"{os.environ.get('GR_VARIANT'):28s} pixels off {int(bad.sum()):4d}  masked rmse {float(np.sqrt((d[~bad] ** 2).mean())):.3e}  tex err 50/99 "
          f"{np.percentile(dt, 50):.2e} {np.percentile(dt, 99):.2e}  attempts {state.attempts()}", flush=True)


if __name__ == "__main__":
    mode, fixture = sys.argv[1], sys.argv[2]
    if os.environ.get("GR_VARIANT"):
        one(mode, fixture)
        sys.exit(0)
    procs = []
    for name, flags in VARIANTS.items():
        env = dict(os.environ, GR_VARIANT=name, GR_EXTRA_FLAGS=flags)
        if mode == "precompile":
            procs.append(subprocess.Popen([sys.executable, __file__, mode, fixture], env=env))
            if len(procs) >= 4:
                procs.pop(0).wait()
        else:
            subprocess.run([sys.executable, __file__, mode, fixture], env=env)
    for p in procs:
        p.wait()
