"""Randomised parity soak of the camera riding a timelike geodesic (GPU box): random metric / parameters / start / observer speed,
gr_boost_tetrad .. gr_get_geodesic_path .. gr_parallel_transport_quantity .. gr_handle_interpolating_geodesic through the C ABI against
the reference's own kernels of the same names compiled for x86-64 (oracle/_ref; built by `tests/fuzz_parity.py <n> <seed> precompile`
in the build container: the libraries are per metric, not per case).  Tolerances: those of tests/test_gpu_geodesic_camera.py.
Test infrastructure (it runs the oracle).  usage: PYTHONPATH=. python tests/fuzz_paths.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import geodesic_raytracing_amd as gra  # noqa: E402
from gpu_stages import GeodesicCamera, rel_err, vec_err  # noqa: E402
from oracle import build_ref  # noqa: E402
from oracle.refpipe import OraclePipeline, pack_features  # noqa: E402

SCRIPTS = os.path.join(ROOT, "geodesic_raytracing_amd", "scripts")
METRICS = {  # as tests/fuzz_parity.py (the oracle libraries are its `fuzz_<metric>` builds)
    "minkowski": {}, "schwarzschild": {}, "kerr_boyer": {"a": (-0.49, 0.49)}, "alcubierre": {}, "schwarzschild_ingoing_ef": {},
    "wormhole": {}, "cosmic_string": {"mu": (0.0, 0.1)}, "kerr_newman_boyer": {"a": (-0.3, 0.3), "rq": (0.0, 0.3)},
    "kerr_schild": {"a": (-0.45, 0.45)}, "schwarzschild_adaptive": {"rs": (0.5, 2.0)},
    "double_unequal_kerr": {"fa1": (-0.9, 0.9), "fa2": (-0.9, 0.9), "R": (3.0, 5.0)},
}
TIMES = (0.0, 0.37, 1.5, 7.3, 19.0, 1.0e6)


def draw_cases(cases, seed):
    """the soak's cases from one random stream: (index, metric name, metric, cfg, start, observer speed, transport, features, start radius)"""
    rng = np.random.default_rng(seed)
    names = sorted(METRICS)
    for case in range(cases):
        name = names[case % len(names)]
        metric = gra.Metric(name, SCRIPTS)
        cfg = metric.cfg_values(**{k: float(rng.uniform(*r)) for k, r in METRICS[name].items()})
        r = float(rng.uniform(4.0, 10.0))
        direction = rng.normal(size=3)
        direction /= np.linalg.norm(direction)
        pos = [float(rng.uniform(-1, 1))] + [float(x) for x in r * direction]
        speed = [float(x) for x in rng.uniform(-0.4, 0.4, 3)]
        transport = bool(rng.random() < 0.7)
        feats = dict(adaptive_sampling=0, max_acceleration_change=metric.info.max_acceleration_change, reparameterisation=int(rng.random() < 0.25))
        yield case, name, metric, cfg, pos, speed, transport, feats, r


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 71
    failed = skipped = 0
    worst = {"path": 0.0, "transported": 0.0, "interpolated": 0.0}
    for case, name, metric, cfg, pos, speed, transport, feats, r in draw_cases(cases, seed):
        only = int(sys.argv[3]) if len(sys.argv) > 3 else None   # replay one case (the stream is advanced through the earlier ones) with details
        if only is not None and case != only:
            continue
        so = build_ref.prebuilt("fuzz_" + name, metric.argument_string())
        if not so:
            skipped += 1
            continue
        want = OraclePipeline(so).geodesic_camera(cfg, pack_features(**feats), camera_pos=pos, basis_speed=speed, max_len=2048,
                                                  target_times=TIMES, parallel_transport=transport)
        meta = dict(metric=name, scripts=True, cfg=cfg, features=feats, camera_pos=pos, basis_speed=speed, max_len=2048,
                    parallel_transport=transport, target_times=list(TIMES), count=want["count"], width=1, height=1,
                    camera_quat=[0.0, 0.0, 0.0, 1.0])
        cam = GeodesicCamera(meta)
        got = cam.snapshot()
        problems = []
        if only is not None:
            print("start", pos, "speed", speed, "cfg", list(cfg))
            print("boosted tetrad difference\n", np.abs(got["tetrad_boosted"] - want["tetrad_boosted"]), "\nreference's\n", want["tetrad_boosted"])
            n = min(got["count"], want["count"])
            e = np.maximum(vec_err(got["path"][:n], want["path"][:n]).max(axis=1), vec_err(got["velocity"][:n], want["velocity"][:n]).max(axis=1))
            first = int(np.argmax(e > 1e-3)) if (e > 1e-3).any() else -1
            print("steps", got["count"], want["count"], "first step over 1e-3:", first, "of", n, "worst", float(e.max()), "at", int(e.argmax()))
            for i in sorted(set([0, max(first - 1, 0), max(first, 0), min(first + 1, n - 1), min(first + 2, n - 1), n // 2, n - 2, n - 1])):
                print("  step", i, "position", want["path"][i], "gpu", got["path"][i], "ds", want["ds"][i], got["ds"][i])
                print("       velocity", want["velocity"][i], "gpu", got["velocity"][i])
            if got["count"] == want["count"]:
                d = np.abs(got["transported"] - want["transported"])
                print("transported tetrads: shape", d.shape, "largest difference", float(d.max()), "at", np.unravel_index(int(d.argmax()), d.shape),
                      "scale", float(np.abs(want["transported"]).max()))
                worst_vector = np.unravel_index(int(d.argmax()), d.shape)[0]
                for i in sorted(set([0, 1, n // 2, n - 2, n - 1, int(np.unravel_index(int(d.argmax()), d.shape)[1]) if d.ndim == 3 else 0])):
                    print("  sample", i, "reference", want["transported"][worst_vector][i] if d.ndim == 3 else want["transported"][i],
                          "gpu", got["transported"][worst_vector][i] if d.ndim == 3 else got["transported"][i])
        if np.abs(got["tetrad_boosted"] - want["tetrad_boosted"]).max() > 2e-6 * max(1.0, np.abs(want["tetrad_boosted"]).max()):
            problems.append("boosted tetrad")
        if got["count"] != want["count"]:
            problems.append(f"steps {got['count']} vs {want['count']}")
        else:
            e_path = max(vec_err(got["path"], want["path"]).max(), vec_err(got["velocity"], want["velocity"]).max(),
                         rel_err(got["ds"], want["ds"], floor=1e-6).max())
            e_tr = np.abs(got["transported"] - want["transported"]).max() / max(1.0, np.abs(want["transported"]).max())
            worst["path"], worst["transported"] = max(worst["path"], e_path), max(worst["transported"], e_tr)
            if e_path > 1e-3:
                problems.append(f"path {e_path:.1e}")
            if e_tr > 1e-3:
                problems.append(f"transported tetrads {e_tr:.1e}")
            for k, t in enumerate(TIMES):
                camera, tetrad, velocity = cam.interpolate(t)
                w = want["interpolated"][k]
                e_i = max(vec_err(camera, w["camera"]).max(), vec_err(velocity, w["velocity"]).max(),
                          np.abs(tetrad - w["tetrad"]).max() / max(1.0, np.abs(w["tetrad"]).max()))
                worst["interpolated"] = max(worst["interpolated"], e_i)
                if e_i > 2e-3:
                    problems.append(f"interpolated at {t}: {e_i:.1e}")
                    break
        failed += bool(problems)
        print(f"{case:3d} {name:26s} r={r:5.2f} transport={int(transport)} reparam={feats['reparameterisation']} steps {want['count']:4d}"
              f"{'  <-- FAIL: ' + ', '.join(problems) if problems else ''}", flush=True)
    print(f"{cases} paths: {failed} outside tolerance, {skipped} skipped (no oracle/_ref build), worst path {worst['path']:.1e}, "
          f"transported {worst['transported']:.1e}, interpolated {worst['interpolated']:.1e}")
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
