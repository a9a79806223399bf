"""Randomised soak of everything that only SCHEDULES a fused frame (GPU box): camera paths of a few frames - repeated cameras, small moves, jumps, sliders dragged and set -
over five shipped scripts, the prepass on, adaptive sampling on and off, several frame sizes, dynamic programs with random parameters;
every frame rendered twice: by a render state with the library's defaults (the previous frame's prepass reused for a repeated camera, tiles by
the frame before's costs, speculative tiles, also in the lattice launch) and by one with all of it switched off (every frame its own prepass
as a launch of its own, image order).  The records, the prepass verdicts and the frame's attempt count must be the same, bit for bit.
Test infrastructure.  usage: PYTHONPATH=. python tests/fuzz_schedule.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import geodesic_raytracing_amd as gra  # noqa: E402
from geodesic_raytracing_amd.pipeline import RENDER_DATA_DTYPE, download  # noqa: E402

SCRIPTS = os.path.join(ROOT, "geodesic_raytracing_amd", "scripts")
METRICS = {"kerr_boyer": {"a": (-0.49, 0.49)}, "kerr_schild": {"a": (-0.45, 0.45)}, "kerr_newman_boyer": {"a": (-0.3, 0.3), "rq": (0.0, 0.3)},
           "schwarzschild_ingoing_ef": {}, "schwarzschild": {}}   # (the last two do not ask for a prepass in their JSON: they get one here)
SIZES = [(640, 360), (1280, 720), (1024, 1024), (1920, 1080), (322, 242)]


def draw_cases(cases, seed):
    rng = np.random.default_rng(seed)
    names = sorted(METRICS)
    for case in range(cases):
        name = names[int(rng.integers(len(names)))]
        metric = gra.Metric(name, SCRIPTS)
        params = {k: float(rng.uniform(*r)) for k, r in METRICS[name].items()}
        adaptive = int(rng.integers(2))
        size = SIZES[int(rng.integers(len(SIZES)))]
        pos = np.array([0.0, rng.uniform(-1.5, 1.5), -rng.uniform(3.5, 9.0), rng.uniform(-1.5, 1.5)])
        path = []
        for _ in range(int(rng.integers(4, 8))):
            kind = str(rng.choice(["same", "step", "jump"], p=[0.3, 0.5, 0.2]))
            if kind == "step":
                pos = pos + np.array([0.0, rng.normal(0, 0.02), rng.normal(0, 0.02), rng.normal(0, 0.02)])
            elif kind == "jump":
                pos = np.array([0.0, rng.uniform(-1.5, 1.5), -rng.uniform(3.5, 9.0), rng.uniform(-1.5, 1.5)])
            # ... and now and then a slider moves with the camera or instead of it: every parameter by a per cent or so (a "drag": the
            # orders of the frame before are still followed), or to a new value altogether (they are not)
            slider = str(rng.choice(["", "drag", "set"], p=[0.7, 0.2, 0.1])) if params else ""
            if slider == "drag":
                params = {k: float(np.clip(v * (1 + rng.normal(0, 0.01)) + rng.normal(0, 0.001), *METRICS[name][k])) for k, v in params.items()}
            elif slider == "set":
                params = {k: float(rng.uniform(*r)) for k, r in METRICS[name].items()}
            path.append((kind + ("+" + slider if slider else ""), pos.copy(), dict(params)))
        yield case, name, metric, params, adaptive, size, path


def run_case(name, metric, params, adaptive, size, path, device=0, peek_every=3):
    """-> (frames compared, frames that differ, frames of the default state that reused a prepass, frames that followed a history)"""
    w, h = size
    feats = metric.features(adaptive_sampling=adaptive, adaptive_sampling_threshold=32.0)
    prog = gra.Program(metric.argument_string(), device)
    plain = dict(mode=gra.MODE_FUSED, use_prepass=1, count_attempts=1, guess_still_camera=0)
    variants = {"defaults": (gra.RenderState(w, h, device), dict(plain)),
                "nothing": (gra.RenderState(w, h, device), dict(plain, reuse_still_camera=0, tile_history=0, inline_prepass=0, speculative_classes=0))}
    differ = 0
    for k, (kind, pos, frame_params) in enumerate(path):
        got = {}
        cfg = metric.cfg_values(**frame_params)
        for label, (state, options) in variants.items():
            state.render(prog, metric, gra.default_camera([float(x) for x in pos]), None, None, feats, cfg, gra.frame_options(**options))
            state.synchronize()
            rd = download(device, state.buffer(gra.BUF_RENDER_DATA), RENDER_DATA_DTYPE, w * h).tobytes()
            # (a pointer to the verdicts ends the reuse for the next frame: looked at on some frames only, so that both ways are tested)
            term = download(device, state.buffer(gra.BUF_TERMINATION), np.int32, (w // 16) * (h // 16)).tobytes() if k % peek_every == peek_every - 1 else b""
            got[label] = (rd, term, state.attempts())
        differ += got["defaults"] != got["nothing"]
    state = variants["defaults"][0]
    return len(path), differ, state.prepass_reused(), state.tile_history()[1]


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 141
    total = bad = reused = followed = 0
    for case, name, metric, params, adaptive, size, path in draw_cases(cases, seed):
        n, differ, r, f = run_case(name, metric, params, adaptive, size, path)
        total += n
        bad += differ
        reused += r
        followed += f
        print(f"{case:3d} {name:26s} adaptive={adaptive} {size[0]:4d}x{size[1]:<4d} {' '.join(k for k, _, _ in path):44s} frames {n} differ {differ}  "
              f"(prepass reused {r}, history followed {f})", flush=True)
    print(f"{cases} cases, {total} frames rendered both ways: {bad} differ; the default states reused {reused} prepasses and followed {followed} histories")
    sys.exit(1 if bad else 0)
