"""Randomised parity soak over EVERY script of the reference's scripts/ folder (31), against the reference's own device source: random
camera pose / observer speed / features / parameters (defaults scaled by 0.9 .. 1.1), small frames, the HIP fused kernels (dynamic and
substituted program) against /root/reference/cl.cl compiled for x86-64 (oracle/_ref, which travels to the GPU box) - not against the
restatement.  tests/fuzz_parity.py soaks the eleven metrics this repository ships scripts for; the fixtures of
tests/golden/refscripts/ hold the other twenty-two at one pose each.  Test infrastructure (it runs the oracle).

  python tests/fuzz_refscripts.py <cases per script> <seed> precompile [names...]
        build container (/root/reference present): the reference's unmodified scripts through this repository's front-end and
        generator, the cases drawn, their argument strings and the frame driver's settings written to
        tools/_manifests/fuzz_refscripts_<seed>.json (git-ignored; travels), every program into the code-object cache, cl.cl built
        per script (oracle/_ref/libref_fuzzref_<name>.so) and the restatement (oracle/_build) for the cases that need a second opinion
  PYTHONPATH=. python tests/fuzz_refscripts.py <cases per script> <seed> [names...]
        GPU box: one line per case, a summary; exit status 1 if a case is outside the end-to-end tolerance of the parity tests and
        is not ill-conditioned by the rule of tests/fuzz_parity.py (the reference's x86 build against the restatement and against a
        float64 evaluation of the same rays differ in as many places)
FUZZ_MODE=reference: the reference-shaped kernel sequence instead of the fused kernels.  FUZZ_ADAPTIVE=1 (both modes): adaptive sampling on
in every case (the reference GUI's default), threshold 16 / 32 / 64; FUZZ_PREPASS=1: the low-resolution prepass on, frames of 128 x 72."""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import geodesic_raytracing_amd as gra  # noqa: E402
from geodesic_raytracing_amd.pipeline import DeviceBuffer  # noqa: E402
from oracle import build_ref, build_restate  # noqa: E402
from oracle.refpipe import OraclePipeline, pack_features  # noqa: E402
from fuzz_parity import quat_from_axis_angle, quat_mul  # noqa: E402

REFERENCE_SCRIPTS = "/root/reference/scripts"
ADAPTIVE = os.environ.get("FUZZ_ADAPTIVE", "0") not in ("", "0")
PREPASS = os.environ.get("FUZZ_PREPASS", "0") not in ("", "0")
W, H = (128, 72) if PREPASS else (64, 36)


def manifest_path(seed):
    return os.path.join(ROOT, "tools", "_manifests", f"fuzz_refscripts_{seed}{'_adaptive' if ADAPTIVE else ''}.json")


def draw(per_script, seed, names):
    import glob
    rng = np.random.default_rng(seed)
    thresholds = np.random.default_rng(seed + 1000003)   # (a stream of its own: a seed names the same cameras with and without)
    out = {"scripts": {}, "cases": []}
    scripts = sorted(os.path.basename(p)[:-3] for p in glob.glob(os.path.join(REFERENCE_SCRIPTS, "*.js")))
    for name in scripts:
        if names and name not in names:
            continue
        m = gra.Metric(name, REFERENCE_SCRIPTS)
        out["scripts"][name] = {"info": {f: getattr(m.info, f) for f, _ in m.info._fields_}, "dynamic_vars": m.dynamic_vars,
                                "dynamic_defaults": m.dynamic_defaults, "dynamic": m.argument_string()}
        for k in range(per_script):
            cfg = [float(v * rng.uniform(0.9, 1.1)) for v in m.dynamic_defaults]
            r = float(rng.uniform(3.0, 12.0))
            direction = rng.normal(size=3)
            direction /= np.linalg.norm(direction)
            pos = [float(rng.uniform(-1, 1))] + [float(x) for x in r * direction]
            quat = quat_mul(quat_from_axis_angle(rng.normal(size=3), float(rng.uniform(0, 1.0))), quat_from_axis_angle([1, 0, 0], -np.pi / 2))
            speed = [float(x) for x in rng.uniform(-0.3, 0.3, 3)] if rng.random() < 0.5 else [0.0, 0.0, 0.0]
            fkw = dict(adaptive_sampling=0, max_acceleration_change=float(m.info.max_acceleration_change), redshift=int(rng.random() < 0.4),
                       reparameterisation=int(rng.random() < 0.25), field_of_view=float(rng.choice([60.0, 90.0, 110.0])),
                       universe_size=float(rng.choice([20.0, 30.0])), max_precision_radius=float(rng.choice([10.0, 14.0])),
                       min_step=float(rng.choice([1e-6, 1e-6, 1e-3])))
            if fkw["redshift"]:
                fkw["use_old_redshift"] = int(rng.random() < 0.5)
            if ADAPTIVE:
                fkw.update(adaptive_sampling=1, adaptive_sampling_threshold=float(thresholds.choice([16.0, 32.0, 64.0])))
            sub = m.argument_string(features=gra.default_features(**fkw), static=True, cfg_values=cfg)
            out["cases"].append({"script": name, "k": k, "cfg": cfg, "pos": pos, "quat": [float(q) for q in quat], "speed": speed, "features": fkw,
                                 "r": r, "max_probes": int(rng.choice([1, 4, 8, 8, 16])), "substituted": sub})
        print(f"{name:36s} accel ops {m.info.accel_ops:5d}", flush=True)
    return out


def _compile(text):
    gra.Program.precompile(text)
    return True


def _reference_build(job):
    name, text = job
    build_ref.build("fuzzref_" + name, text)
    build_restate.build(text)
    return True


def precompile(per_script, seed, names):
    import multiprocessing
    man = draw(per_script, seed, names)
    os.makedirs(os.path.dirname(manifest_path(seed)), exist_ok=True)
    with open(manifest_path(seed), "w") as f:
        json.dump(man, f)
    programs = [e["dynamic"] for e in man["scripts"].values()] + [c["substituted"] for c in man["cases"]]
    with multiprocessing.get_context("spawn").Pool(max(1, (os.cpu_count() or 2) - 1)) as pool:
        pool.map(_reference_build, [(n, e["dynamic"]) for n, e in man["scripts"].items()], chunksize=1)
        pool.map(_compile, programs, chunksize=1)
    print(f"{len(man['scripts'])} scripts, {len(man['cases'])} cases, {len(programs)} programs -> {manifest_path(seed)}")


def main():
    per_script, seed = int(sys.argv[1]), int(sys.argv[2])
    rest = sys.argv[3:]
    if rest and rest[0] == "precompile":
        precompile(per_script, seed, rest[1:])
        return 0
    man = json.load(open(manifest_path(seed)))
    names = set(rest)
    sky_size = (512, 256) if PREPASS else (256, 128)   # four sky texels to a pixel either way (tests/fuzz_parity.py)
    bg_np, levels = gra.pack_background(gra.synthetic_background(*sky_size))
    bg2_np, _ = gra.pack_background(gra.synthetic_background(*sky_size, seed=0x2B5EED))
    bg, bg2 = DeviceBuffer.from_numpy(0, bg_np), DeviceBuffer.from_numpy(0, bg2_np)
    out = DeviceBuffer(0, W * H * 16)
    state = gra.RenderState(W, H, 0)
    threads = os.cpu_count() or 4
    metrics, references, dynamic_programs = {}, {}, {}
    failed = explained = done = 0
    worst = 0.0
    for c in man["cases"]:
        name = c["script"]
        if names and name not in names:
            continue
        e = man["scripts"][name]
        key = e["dynamic"]
        if name not in metrics:
            metrics[name] = gra.Metric.from_info(name, e["info"], e["dynamic_vars"], e["dynamic_defaults"])
            so = build_ref.prebuilt("fuzzref_" + name, key)
            assert so, f"no reference build for {name} (run the precompile mode in the build container)"
            references[name] = OraclePipeline(so)
            dynamic_programs[name] = gra.Program(key, 0)
        metric, fkw, cfg = metrics[name], c["features"], c["cfg"]
        feats = gra.default_features(**fkw)
        frame_args = dict(camera_pos=c["pos"], camera_quat=c["quat"], basis_speed=c["speed"], background=(bg_np, bg2_np, levels), nthreads=threads,
                          use_prepass=PREPASS, max_probes=c["max_probes"])
        ref = references[name].frame(W, H, cfg, pack_features(**fkw), **frame_args)
        cam = gra.default_camera(c["pos"], c["quat"])
        cam.basis_speed = (gra.c_float * 3)(*c["speed"])
        lit = float((ref["render_data"]["terminated"] == 1).mean())
        line = (f"{name:34s} {c['k']} r={c['r']:5.2f} speed={int(any(c['speed']))} redshift={fkw['redshift']}{'o' if fkw.get('use_old_redshift') else ' '} "
                f"reparam={fkw['reparameterisation']} min_step={fkw['min_step']:.0e} probes={c['max_probes']:2d} lit {lit:4.2f}")
        for label, prog in (("dyn", dynamic_programs[name]), ("sub", gra.Program(c["substituted"], 0))):
            o = gra.frame_options(mode=gra.MODE_REFERENCE if os.environ.get("FUZZ_MODE") == "reference" else gra.MODE_FUSED, use_prepass=1 if PREPASS else 0,
                                  count_attempts=1, max_probes=c["max_probes"])
            state.render(prog, metric, cam, out.ptr, ((bg.ptr, bg2.ptr), bg_np.shape[2], bg_np.shape[1], levels), feats, cfg, o)
            state.synchronize()
            px = out.to_numpy(np.float32, (H, W, 4))
            # pixels the reference leaves undefined: a ray that ended with terminated = 1 on a non-finite position (a coordinate
            # singularity reached in one step) has NaN sky coordinates, and what an image read returns for those is the device's
            # business (OpenCL 1.2, 8.2); its neighbours to the left and above read them for their footprints.  Not compared.
            undefined = ~np.isfinite(ref["pixels"][..., :3]).all(axis=2)
            defined = ~undefined
            d = np.where(defined[..., None], px[..., :3] - ref["pixels"][..., :3], 0.0)
            bad = (np.abs(d).max(axis=2) > 1e-3) | (defined & ~np.isfinite(px[..., :3]).all(axis=2))
            good = defined & ~bad
            rmse = float(np.sqrt((d[good] ** 2).mean())) if good.any() else 0.0
            finite = np.isfinite(px[..., :3][defined]).all()
            ok = bad.sum() <= 0.01 * max(1, defined.sum()) and rmse <= 1e-4 and finite
            verdict = "" if ok else "  <-- FAIL"
            if undefined.any():
                verdict = f" [{int(undefined.sum())} px undefined in the reference]" + verdict
            if not ok and rmse <= 3e-4:
                # the same frame through the restatement (same algorithm and operation order, another compiler) and the reference build's
                # rays through a float64 evaluation: where those differ from the reference build in as many places as the GPU does, the
                # frame is ill-conditioned (the rule of tests/fuzz_parity.py and test_polar_axis_cases_of_the_soak)
                other = OraclePipeline(build_restate.build(key))
                theirs = other.frame(W, H, cfg, pack_features(**fkw), **frame_args)
                with np.errstate(invalid="ignore"):
                    scatter = int((defined & ((np.abs(theirs["pixels"][..., :3] - ref["pixels"][..., :3]).max(axis=2) > 1e-3)
                                              | ~np.isfinite(theirs["pixels"][..., :3]).all(axis=2))).sum())
                p64, t64 = other.trace_f64(ref["rays_init"], cfg, pack_features(**fkw), nthreads=threads)
                both = (t64 == 1) & (ref["rays"]["terminated"] == 1)
                d64 = np.abs(np.asarray(ref["rays"]["position"], dtype=np.float64)[both][:, 2:] - p64[both][:, 2:]).max(axis=1)
                reference_off = int((d64 > 1e-3).sum())
                against = int(bad.sum())
                detail = f"reference build vs restatement {scatter} px, vs float64 {reference_off} rays, GPU vs reference build {against} px"
                if against <= 2 * max(scatter, reference_off) + 4:
                    verdict, ok = verdict.replace("  <-- FAIL", "") + f"  <-- ill-conditioned ({detail})", True
                    explained += 1
                else:
                    verdict += f" ({detail})"
            failed += not ok
            worst = max(worst, rmse if ok and "ill-conditioned" not in verdict else 0.0)
            line += f" | {label}: rmse {rmse:.1e} off {bad.sum() * 100.0 / max(1, defined.sum()):4.1f}%{verdict}"
        done += 1
        print(line, flush=True)
    print(f"{done} cases x 2 programs over {len(metrics)} reference scripts against the reference's cl.cl (x86-64): {failed} outside tolerance, "
          f"{explained} ill-conditioned, worst masked RMSE of the others {worst:.2e}")
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
