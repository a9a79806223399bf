"""Randomised parity soak (GPU box): random metric / parameters / camera pose / observer speed / features, small frames, the HIP
fused kernel (dynamic and substituted program) against the CPU oracle (oracle/restate.cpp, pinned to the reference's kernels).
Prints one line per case and a summary; exit status 1 if a case is outside the end-to-end tolerance of the parity tests - unless the
reference's own x86-64 build (oracle/_ref, built by the `precompile` mode where /root/reference exists) differs from the restatement, or
from a float64 evaluation of the same algorithm, in as many places (an ill-conditioned frame - the rule of
tests/test_gpu_parity.py::test_polar_axis_cases_of_the_soak; reported, not failed).
Test infrastructure (it runs the oracle).  usage: PYTHONPATH=. python tests/fuzz_parity.py [cases] [seed] [only_case]
FUZZ_ADAPTIVE=1: adaptive sampling on in every case; FUZZ_PREPASS=1: the prepass on, 128 x 72 frames; FUZZ_MODE=reference: the
reference-shaped kernel sequence instead of the fused kernels; FUZZ_SIZE=WxH: another frame size.
With only_case the one case is replayed (the random stream is advanced through the earlier ones) and its inputs, the oracle's
and the GPU's pixels and render-data go to gpurun_out/fuzz_case_<seed>_<case>.npz for a closer look."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import geodesic_raytracing_amd as gra  # noqa: E402
from geodesic_raytracing_amd.pipeline import DeviceBuffer  # noqa: E402
from oracle import build_ref, build_restate  # noqa: E402
from oracle.refpipe import OraclePipeline, pack_features  # noqa: E402

SCRIPTS = os.path.join(ROOT, "geodesic_raytracing_amd", "scripts")
METRICS = {  # name -> parameter ranges
    "minkowski": {}, "schwarzschild": {}, "kerr_boyer": {"a": (-0.49, 0.49)}, "alcubierre": {}, "schwarzschild_ingoing_ef": {},
    "wormhole": {}, "cosmic_string": {"mu": (0.0, 0.1)}, "kerr_newman_boyer": {"a": (-0.3, 0.3), "rq": (0.0, 0.3)},
    "kerr_schild": {"a": (-0.45, 0.45)}, "schwarzschild_adaptive": {"rs": (0.5, 2.0)},
    # sub- and hyper-extreme constituents: the rod half-lengths are real for some draws and complex for others (csrc/sym.cpp folds the
    # complex roots of the substituted program when they turn out real)
    "double_unequal_kerr": {"fa1": (-1.2, 1.2), "fa2": (-1.2, 1.2), "R": (3.0, 5.0)},
}


def quat_from_axis_angle(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    axis /= np.linalg.norm(axis)
    s = np.sin(angle / 2)
    return [float(axis[0] * s), float(axis[1] * s), float(axis[2] * s), float(np.cos(angle / 2))]


def quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return [aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw,
            aw * bw - ax * bx - ay * by - az * bz]


ADAPTIVE = os.environ.get("FUZZ_ADAPTIVE", "0") not in ("", "0")   # every case with adaptive sampling on (the reference GUI's default)
PREPASS = os.environ.get("FUZZ_PREPASS", "0") not in ("", "0")     # every case with the low-resolution prepass on, frames of 128 x 72


def draw_cases(cases, seed):
    """the soak's cases, drawn from one random stream: (index, metric name, metric, cfg, camera position, orientation, observer speed,
    feature keywords, camera distance, max_probes).  FUZZ_ADAPTIVE=1: the same cases with adaptive sampling on, the threshold from a
    stream of its own (so that a seed names the same cameras either way).  Round 5: use_old_redshift (with redshift on), min_step
    (values at which calculate_ds_error's bail-out fires, cl.cl:3439-3450) and render's probe cap (cl.cl:5597-5613) from a third stream."""
    rng = np.random.default_rng(seed)
    thresholds = np.random.default_rng(seed + 1000003)
    knobs = np.random.default_rng(seed + 2000003)
    names = sorted(METRICS)
    for case in range(cases):
        name = names[case % len(names)]
        metric = gra.Metric(name, SCRIPTS)
        params = {k: float(rng.uniform(*r)) for k, r in METRICS[name].items()}
        cfg = metric.cfg_values(**params)
        r = float(rng.uniform(3.0, 12.0))
        direction = rng.normal(size=3)
        direction /= np.linalg.norm(direction)
        pos = [float(rng.uniform(-1, 1))] + [float(x) for x in r * direction]
        base = quat_from_axis_angle([1, 0, 0], -np.pi / 2)
        quat = quat_mul(quat_from_axis_angle(rng.normal(size=3), float(rng.uniform(0, 1.0))), base)
        speed = [float(x) for x in rng.uniform(-0.3, 0.3, 3)] if rng.random() < 0.5 else [0.0, 0.0, 0.0]
        fkw = dict(adaptive_sampling=0, max_acceleration_change=metric.info.max_acceleration_change, redshift=int(rng.random() < 0.4),
                   reparameterisation=int(rng.random() < 0.25), field_of_view=float(rng.choice([60.0, 90.0, 110.0])),
                   universe_size=float(rng.choice([20.0, 30.0])), max_precision_radius=float(rng.choice([10.0, 14.0])))
        if ADAPTIVE:
            fkw.update(adaptive_sampling=1, adaptive_sampling_threshold=float(thresholds.choice([16.0, 32.0, 64.0])))
        old_redshift, min_step, max_probes = knobs.random() < 0.5, float(knobs.choice([1e-6, 1e-6, 1e-3, 1e-2])), int(knobs.choice([1, 4, 8, 8, 16]))
        if fkw["redshift"]:
            fkw["use_old_redshift"] = int(old_redshift)
        fkw["min_step"] = min_step
        yield case, name, metric, cfg, pos, quat, speed, fkw, r, max_probes


def _precompile(argument_string):
    gra.Program.precompile(argument_string)
    return True


def precompile(cases, seed):
    """build container: the code objects of every case (dynamic and substituted program) and the CPU oracles, in parallel, so that
    the soak on the GPU box spends its time rendering"""
    import multiprocessing
    strings, oracle_keys = [], []
    for case, name, metric, cfg, pos, quat, speed, fkw, r, max_probes in draw_cases(cases, seed):
        key = metric.argument_string()
        if key not in oracle_keys:
            oracle_keys.append(key)
        for s in (key, metric.argument_string(features=gra.default_features(**fkw), static=True, cfg_values=cfg)):
            if s not in strings:
                strings.append(s)
    with multiprocessing.get_context("spawn").Pool(max(1, (os.cpu_count() or 2) - 1)) as pool:
        pool.map(_precompile, strings, chunksize=1)
        pool.map(build_restate.build, oracle_keys, chunksize=1)
    # the reference's own source compiled for x86-64 (oracle/_ref; where /root/reference exists): what a case outside the tolerance is
    # held against on the GPU box
    built = 0
    if build_ref.reference_available():
        for name in sorted(METRICS):
            build_ref.build("fuzz_" + name, gra.Metric(name, SCRIPTS).argument_string())
            built += 1
    print(f"precompiled {len(strings)} programs, {len(oracle_keys)} oracles, {built} reference builds")


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2026
    if len(sys.argv) > 3 and sys.argv[3] == "precompile":
        precompile(cases, seed)
        return 0
    w, h = (128, 72) if PREPASS else (64, 36)   # (a prepass grid of 8 x 4 cells)
    if os.environ.get("FUZZ_SIZE"):              # e.g. 100x60: partial 8 x 8 tiles on the right and at the bottom, a 6 x 3 prepass grid
        w, h = (int(v) for v in os.environ["FUZZ_SIZE"].split("x"))
    # (four sky texels to a pixel either way: with two, a frame that looks down the chart's axis - grid lines converging on the pole all
    # over it - turns sky coordinates that agree to 2e-6 into pixels 5e-4 apart on every line, masked RMSE 1.4e-4)
    sky_size = (512, 256) if PREPASS else (256, 128)
    bg_np, levels = gra.pack_background(gra.synthetic_background(*sky_size))
    bg2_np, _ = gra.pack_background(gra.synthetic_background(*sky_size, seed=0x2B5EED))   # what a ray that ends on the far side samples (cl.cl:5445-5448)
    bg, bg2 = DeviceBuffer.from_numpy(0, bg_np), DeviceBuffer.from_numpy(0, bg2_np)
    out = DeviceBuffer(0, w * h * 16)
    state = gra.RenderState(w, h, 0)
    oracles, worst, failed, explained = {}, 0.0, 0, 0
    for case, name, metric, cfg, pos, quat, speed, fkw, r, max_probes in draw_cases(cases, seed):
        only = int(sys.argv[3]) if len(sys.argv) > 3 else None
        if only is not None and case != only:
            continue
        feats = gra.default_features(**fkw)
        key = metric.argument_string()
        if key not in oracles:
            oracles[key] = OraclePipeline(build_restate.build(key))
        ref = oracles[key].frame(w, h, cfg, pack_features(**fkw), camera_pos=pos, camera_quat=quat, basis_speed=speed,
                                 background=(bg_np, bg2_np, levels), nthreads=os.cpu_count() or 4, use_prepass=PREPASS, max_probes=max_probes)
        oracles[key].lib.ref_last_attempts.restype = ctypes.c_uint64
        cam = gra.default_camera(pos, quat)
        cam.basis_speed = (gra.c_float * 3)(*speed)
        line = f"{case:3d} {name:26s} r={r:5.2f} speed={int(any(speed))} redshift={fkw['redshift']}{'o' if fkw.get('use_old_redshift') else ''} reparam={fkw['reparameterisation']} min_step={fkw['min_step']:.0e} probes={max_probes:2d}"
        for label, prog in (("dyn", gra.Program(key, 0)),
                            ("sub", gra.Program(metric.argument_string(features=feats, static=True, cfg_values=cfg), 0))):
            # FUZZ_MODE=reference: the reference-shaped kernel sequence instead of the fused kernels
            o = gra.frame_options(mode=gra.MODE_REFERENCE if os.environ.get("FUZZ_MODE") == "reference" else gra.MODE_FUSED,
                                  use_prepass=1 if PREPASS else 0, count_attempts=1, max_probes=max_probes)
            state.render(prog, metric, cam, out.ptr, ((bg.ptr, bg2.ptr), bg_np.shape[2], bg_np.shape[1], levels), feats, cfg, o)
            state.synchronize()
            px = out.to_numpy(np.float32, (h, w, 4))
            d = px[..., :3] - ref["pixels"][..., :3]
            bad = np.abs(d).max(axis=2) > 1e-3
            rmse = float(np.sqrt((d[~bad] ** 2).mean())) if (~bad).any() else 0.0
            # a hyper-extreme constituent of the double-Kerr solution (|a_i / m_i| > 1) is a naked singularity: chaotic orbits around
            # it, as for the super-extremal Kerr fixture - any two builds differ in a few per cent of the pixels (case 31/46: the
            # reference's own x86 build and the CPU restatement, same operation order, in 86 of 2304), so the mask is 10 % there
            chaotic = name == "double_unequal_kerr" and max(abs(cfg[2]), abs(cfg[3])) > 1.0
            ok = bad.mean() <= (0.10 if chaotic else 0.01) and rmse <= (3e-4 if chaotic else 1e-4) and np.isfinite(px).all()
            verdict = "" if ok else "  <-- FAIL"
            if not ok and np.isfinite(px).all() and rmse <= 3e-4 and bad.mean() > (0.10 if chaotic else 0.01):   # (too many pixels off; a bad RMSE alone stays a failure)
                # Too many pixels off: is it the case or the kernel?  The reference's own source compiled for x86-64 against the CPU
                # restatement (same algorithm, same operation order, another compiler): where those two differ in as many pixels, the
                # frame is ill-conditioned (rays grazing a chart axis: tests/test_gpu_parity.py::test_polar_axis_cases_of_the_soak,
                # same rule: twice the larger of that and the reference build's own distance from a float64 evaluation, + 4) and the
                # case is reported as such, not as a failure.
                so = build_ref.prebuilt("fuzz_" + name, key)
                if so:
                    theirs = OraclePipeline(so).frame(w, h, cfg, pack_features(**fkw), camera_pos=pos, camera_quat=quat, basis_speed=speed,
                                                      background=(bg_np, bg2_np, levels), nthreads=os.cpu_count() or 4, use_prepass=PREPASS, max_probes=max_probes)
                    scatter = int((np.abs(theirs["pixels"][..., :3] - ref["pixels"][..., :3]).max(axis=2) > 1e-3).sum())
                    against = int((np.abs(px[..., :3] - theirs["pixels"][..., :3]).max(axis=2) > 1e-3).sum())
                    # ... and the reference build's rays against a float64 evaluation of the same algorithm: sky angles off by > 1e-3
                    # (with adaptive sampling the traced rays are the lattice's and the refined pixels': a quarter to all of the frame's)
                    p64, t64 = oracles[key].trace_f64(theirs["rays_init"], cfg, pack_features(**fkw), nthreads=os.cpu_count() or 4)
                    both = (t64 == 1) & (theirs["rays"]["terminated"] == 1)
                    d64 = np.abs(np.asarray(theirs["rays"]["position"], dtype=np.float64)[both][:, 2:] - p64[both][:, 2:]).max(axis=1)
                    reference_off = int((d64 > 1e-3).sum())
                    detail = f"reference build vs restatement {scatter} px, vs float64 {reference_off} rays, GPU vs reference build {against} px"
                    if against <= 2 * max(scatter, reference_off) + 4:
                        verdict = f"  <-- ill-conditioned ({detail})"
                        explained += 1
                        ok = True
                    else:
                        verdict += f" ({detail})"
            failed += not ok
            worst = max(worst, rmse)
            line += f" | {label}: rmse {rmse:.1e} off {bad.mean() * 100:4.1f}%{verdict}"
            if only is not None:
                from geodesic_raytracing_amd.pipeline import RENDER_DATA_DTYPE, download
                rd = download(0, state.buffer(gra.BUF_RENDER_DATA), RENDER_DATA_DTYPE, w * h)
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                np.savez(os.path.join(ROOT, "gpurun_out", f"fuzz_case_{sys.argv[2]}_{case}_{label}.npz"), pixels=px, ref_pixels=ref["pixels"],
                         rd=rd, ref_rd=ref["render_data"], pos=np.array(pos), quat=np.array(quat), speed=np.array(speed), cfg=np.array(cfg),
                         features=np.frombuffer(bytes(pack_features(**fkw)), dtype=np.uint8))
        print(line, flush=True)
    print(f"{cases} cases x 2 programs: {failed} outside tolerance, {explained} ill-conditioned (the reference's own two builds differ as much), "
          f"worst masked RMSE {worst:.2e}")
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
