#!/usr/bin/env python3
"""bench.py — headline benchmark: Mrays/s (and frames/s) of the geodesic ray pipeline at 3840x2160 Kerr.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

--gpus N means N GPUs however the script is started (plan_launch): under torch.distributed.run it is one rank of N; as a plain command with
N > 1 it starts its own N ranks and prints rank 0's line (one process driving the N devices over peer copies if the ranks cannot be
brought up: --launch single-process); on a box with fewer GPUs it exits with code 2 and no line, unless GR_BENCH_ONE_DEVICE=1|rccl|peer
asks for a rehearsal of the N-GPU code path on one GPU (marked "rehearsal": not a measurement).

One "step" = one complete frame of BASELINE.json configs[2]: Kerr (Boyer-Lindquist, rs=1, a=0.45 i.e.
a/M=0.9 - SURVEY.md section 8d), 3840x2160, adaptive sampling off (one primary ray per pixel), prepass
as in the metric's config, all hot-path stages (camera tetrad, prepass, fused init+Verlet+render-data,
anisotropic texture render) into a float4 HBM buffer.  Inputs (background, camera, cfg) are resident
in HBM before the timed region.  With N > 1 the frame's rows are dealt to the ranks in 48-row blocks
(block-cyclic), each rank renders its rows, and the final float4 rows are gathered on rank 0 over RCCL.
Frames are rendered the way the reference's main loop does it (a ring of render_state objects, each frame on the next one: main.cpp:1463-1469, 1505-1510):
--frames-in-flight F states, each with its own HIP stream and output buffer, so that the low-occupancy tail of one
frame's trace, its texture pass and (N > 1) its gather overlap the next frame's trace.  Every one of the K timed frames
is complete (and gathered) when the clock stops.  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
VALU_PEAK_TFLOPS = 157.3       # fp32 vector peak
TRACE_BYTES_PER_RAY = 32       # gr_trace_fused: one 32-byte render_data record per pixel
STEP_OVERHEAD_FLOPS = 90       # integrator + step controller per Verlet attempt


# BASELINE.json configs[3] and configs[4] on one GPU: metric script, frame size, camera, features, tag of their committed counters
OTHER_CONFIGS = {3: {"label": "config3_double_unequal_kerr_3840x2160", "metric": "double_unequal_kerr", "size": (3840, 2160), "camera": [0, 0, -6, 0.5],
                     "features": {}, "tag": "double_unequal_kerr_4k"},
                 4: {"label": "config4_alcubierre_7680x4320_redshift", "metric": "alcubierre", "size": (7680, 4320), "camera": [0, 0, -6, 0.5],
                     "features": {"redshift": 1}, "tag": "alcubierre_8k_redshift"}}


def committed_counters(workload_tag, build_key):
    """Hardware counters of the fused trace kernel collected by tools/final_profiles.sh for one workload (profiles/pmc_<tag>.json).
    They describe ONE build of the kernel: the file carries that build's key (gr_program_build_key) and is only used when the
    program that runs now has the same one; otherwise the line says so instead of quoting counters of another kernel."""
    path = os.path.join(ROOT, "profiles", f"pmc_{workload_tag}.json")
    if not os.path.exists(path):
        return None, f"no profiles/pmc_{workload_tag}.json"
    with open(path) as f:
        pmc = json.load(f)
    if pmc.get("build_key") != build_key:
        return None, (f"profiles/pmc_{workload_tag}.json was collected for build {pmc.get('build_key')}, this run is build {build_key}: "
                      f"stale, not used (re-run tools/final_profiles.sh)")
    return pmc, f"profiles/pmc_{workload_tag}.json (build {build_key})"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--metric", default="kerr_boyer")
    ap.add_argument("--spin", type=float, default=0.45)
    ap.add_argument("--config", type=int, default=0, help="3 / 4: BASELINE.json configs[3] (double_unequal_kerr 3840x2160) / configs[4] "
                    "(alcubierre 7680x4320, redshift on) as the timed workload on this one GPU (profiling runs; the default line reports "
                    "them under `secondary`)")
    ap.add_argument("--camera", default="", help="camera position t,x,y,z (default 0,0,-4,0)")
    ap.add_argument("--redshift", type=int, default=0)
    ap.add_argument("--inline-prepass", type=int, default=-1, help="gr_frame_options.inline_prepass of the timed frames (-1 library default: a frame "
                    "whose prepass was not computed ahead traces it inside the trace launch; 0: as a launch of its own)")
    ap.add_argument("--use-prepass", type=int, default=-1, help="gr_frame_options.use_prepass of the timed frames: -1 per metric + policy, 0 / 1 forced")
    ap.add_argument("--mode", default="fused", choices=["fused", "reference"])
    ap.add_argument("--program", default="static", choices=["static", "dynamic"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "port", "reference"],
                    help="auto (default) / reference: the reference's own cl.cl compiled for x86-64 (oracle/_ref/*.so, built in the build container, travels "
                         "with the tree) where that object is present, else - and with `port` always - this repository's restatement oracle/restate.cpp")
    ap.add_argument("--no-build-timing", action="store_true", help="skip program_build_s (two cold compiles of the substituted program, ~30 s of a host core)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary figures (other poses / modes); used for profiling runs")
    ap.add_argument("--no-lookahead", action="store_true", help="do not overlap the next frame's prepass with this frame's trace")
    ap.add_argument("--lookahead-depth", type=int, default=0, help="frames of prepass look-ahead (1 or 2); default 1 on one GPU, "
                    "2 when the frame is split over several (a strip traces faster than one prepass runs)")
    ap.add_argument("--block-rows", type=int, default=48, help="rows per block of the block-cyclic row split (N > 1); multiple of 8")
    ap.add_argument("--fused-shading", type=int, default=-1, help="gr_frame_options.fused_shading of the timed frames (-1 library default, "
                    "0 separate gr_render pass, 1 shading inside the trace launch + seams)")
    ap.add_argument("--trace-waves-per-simd", type=int, default=-1, help="persistent waves per SIMD of a trace launch in the timed frames "
                    "(0 = all that fit, -1 = 4 with three or more frames in flight on one GPU, else all)")
    ap.add_argument("--frames-in-flight", type=int, default=3, help="render states / streams cycled through (1 = strictly one frame at a time)")
    ap.add_argument("--launch", default="auto", choices=["auto", "single-process"],
                    help="--gpus N > 1 started as a plain command (no WORLD_SIZE): auto = start the N ranks through torch.distributed.run and print rank "
                         "0's line (one process driving all devices through peer copies if they cannot be started); single-process = that path directly")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--measure-clock", action="store_true", help="count attempts in the timed frames too, so that the shader clock of the "
                    "overlapped launches can be read afterwards (two timestamp reads per wave; the counters cost an atomic per tile)")
    args = ap.parse_args()
    if args.config in OTHER_CONFIGS:
        c = OTHER_CONFIGS[args.config]
        args.metric, (args.width, args.height), args.camera, args.redshift = c["metric"], c["size"], ",".join(str(x) for x in c["camera"]), c["features"].get("redshift", 0)
    if args.fused_shading == 1:   # a build option of the programs (part of their cache key)
        os.environ["GR_EXTRA_FLAGS"] = (os.environ.get("GR_EXTRA_FLAGS", "") + " -DGR_TILE_SHADING").strip()
    return args


def cpu_baseline(metric_name, cfg_values, features_kw, target_seconds, prefer="port"):
    """Times the CPU side on a bounded sample of the same workload: a low-resolution frame with the same camera and field of
    view (same ray distribution), all host cores.  The reference's own cl.cl compiled for x86-64 (oracle/_ref, built in the
    build container for exactly this macro string and shipped as a .so) when it is there - kind "reference" - otherwise this
    repository's C++ restatement of it (oracle/restate.cpp) - kind "port", what a fresh clone without the build container's objects has; how
    the two compare is measured once in the build container (profiles/r05_cpu_calibration.txt) and reported with a "port" figure."""
    from oracle import build_restate, build_ref
    from oracle.refpipe import OraclePipeline, pack_features
    import geodesic_raytracing_amd as gra
    m = gra.Metric(metric_name, os.path.join(ROOT, "geodesic_raytracing_amd", "scripts"))
    so = None
    if prefer in ("reference", "auto"):
        so = build_ref.prebuilt(metric_name + "_script", m.argument_string()) or build_ref.prebuilt(metric_name, m.argument_string())
    kind = "reference" if so else "port"
    if not so:
        so = build_restate.build(m.argument_string())
    pipe = OraclePipeline(so)
    cores = os.cpu_count() or 1
    feats = pack_features(**features_kw)

    def run(w, h):
        t0 = time.perf_counter()
        pipe.frame(w, h, cfg_values, feats, use_prepass=False, nthreads=cores, stages="trace")
        return time.perf_counter() - t0

    w, h = 96, 54
    t = run(w, h)
    rate = w * h / max(t, 1e-6)
    # scale the sample (16:9) towards the target time, bounded
    scale = (min(target_seconds, 30.0) * rate / (w * h)) ** 0.5
    w2 = int(max(96, min(1920, round(96 * scale / 16) * 16)))
    h2 = w2 * 9 // 16
    t2 = run(w2, h2)
    what = "the reference's cl.cl compiled for x86-64 (oracle/_ref)" if kind == "reference" else "oracle/restate.cpp"
    # (the restatement is slower than the reference's own code on the same cores - measured once in the build container,
    # profiles/r05_cpu_calibration.txt: 1.535x on 8 threads - so a "port" figure understates the CPU side by that factor)
    calibration = None if kind == "reference" else {"restatement_slower_than_cl_x86_by": 1.535, "value_scaled_to_the_reference_build": round(w2 * h2 / t2 / 1e6 * 1.535, 6),
                                                    "source": "profiles/r05_cpu_calibration.txt (build container, 8 threads)"}
    return {"value": round(w2 * h2 / t2 / 1e6, 6), "unit": "Mrays/s", "cores": cores, "kind": kind, "calibration": calibration,
            "sample": f"{w2}x{h2} frame of the same camera/metric (init + Verlet trace of every pixel, no prepass skip) through {what}, "
                      f"{t2:.1f} s on {cores} threads"}


def program_build_seconds(metric_name, spin, redshift):
    """Cold build times of the substituted program on this host, in a process and a cache directory of their own with the compiler's own
    result cache off (AMD_COMGR_CACHE=0): (a) the first program of a shape - the free build, the occupancy rule's capped builds, the
    set-up module; (b) the next parameter set of the same shape (a slider moved: the remembered occupancy decision, one compiler run + the
    set-up module).  What a maintainer waits for between a parameter change and the swap of the substituted program: since round 6 the
    part of the program a fused frame launches (gr_program_precompile_frame_path), the rest is built behind the swap."""
    import subprocess
    import tempfile
    code = (
        "import sys, time\n"
        "import geodesic_raytracing_amd as gra\n"
        "m = gra.Metric(sys.argv[1], sys.argv[2])\n"
        "spin, redshift = float(sys.argv[3]), int(sys.argv[4])\n"
        "out = []\n"
        "for k in range(2):\n"
        "    cfg = m.cfg_values(a=spin + 0.01 * k) if 'a' in m.dynamic_vars else [v * (1 + 0.01 * k) for v in m.cfg_values()]\n"
        "    text = m.argument_string(features=m.features(adaptive_sampling=0, redshift=redshift), static=True, cfg_values=cfg)\n"
        "    t = time.perf_counter(); gra.check(gra.lib.gr_program_precompile_frame_path(text.encode())); out.append(time.perf_counter() - t)\n"
        "    t = time.perf_counter(); gra.Program.precompile(text); out.append(time.perf_counter() - t)\n"
        "print(*out)\n")
    with tempfile.TemporaryDirectory(prefix="gr_build_timing") as d:
        env = dict(os.environ, GR_CACHE_DIR=d, AMD_COMGR_CACHE="0", PYTHONPATH=ROOT)
        r = subprocess.run([sys.executable, "-c", code, metric_name, os.path.join(ROOT, "geodesic_raytracing_amd", "scripts"), str(spin), str(redshift)],
                           env=env, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        return {"error": r.stderr[-400:]}
    first, first_rest, again, again_rest = (float(x) for x in r.stdout.split()[-4:])
    return {"first_of_its_shape": round(first, 2), "next_parameters_same_shape": round(again, 2),
            "the_rest_of_the_program_behind_the_swap": [round(first_rest, 2), round(again_rest, 2)],
            "note": "substituted program, compiler cache off: seconds until it can be swapped in = the code object of the kernels a fused frame launches + the set-up "
                    "module, built side by side on two host threads (gr_program_create_async's worker); the kernels of the reference-shaped sequence are a second "
                    "code object built behind the swap on one thread (the third figure; whoever first launches one of them waits for it)"}


def plan_launch(gpus, world_env, devices_seen, rehearsal, launch="auto"):
    """What `bench.py --gpus N` does, however it was started - (action, reason); pure, tests/test_distributed_cpu.py holds it:
    "rank"    this process is one rank of N (torch.distributed.run set WORLD_SIZE = N), or N = 1: run the frames
    "spawn"   a plain `python bench.py --gpus N`, N > 1: start the N ranks (torch.distributed.run) and print rank 0's line
    "single"  one process driving the N devices through peer copies (gr_tiled_create_local): asked for, or the fallback of "spawn"
    "refuse"  exit non-zero: never a line whose n_gpus is smaller than --gpus"""
    if gpus < 1:
        return "refuse", f"--gpus {gpus}"
    if world_env is not None:
        if world_env != gpus:
            return "refuse", f"--gpus {gpus} does not match WORLD_SIZE {world_env}"
        return "rank", "one rank of %d" % gpus
    if gpus == 1:
        return "rank", "one GPU"
    if devices_seen < 1:
        return "refuse", f"--gpus {gpus}: no GPU visible (the HIP path has no CPU fallback)"
    if devices_seen < gpus and not rehearsal:
        return "refuse", (f"--gpus {gpus}: {devices_seen} GPU(s) visible.  On a box with fewer GPUs than ranks only a rehearsal of the N-GPU code path "
                          f"is possible: GR_BENCH_ONE_DEVICE=1 (inter-process transport), =rccl (RCCL over loopback) or =peer (one process, peer copies)")
    if launch == "single-process" or rehearsal == "peer":
        return "single", "one process, peer copies"
    return "spawn", f"{gpus} ranks through torch.distributed.run"


def last_json_line(text):
    for ln in reversed(text.splitlines()):
        ln = ln.strip()
        if ln.startswith("{") and '"metric"' in ln:
            try:
                return json.loads(ln)
            except ValueError:
                continue
    return None


def spawn_ranks(args, reason):
    """`python bench.py --gpus N` as a plain command: N ranks under torch.distributed.run (what the driver's N > 1 command does), rank 0's line
    printed as this process's one JSON line; when the ranks cannot be brought up (no process group, RCCL refuses) the same frames from one
    process over peer copies; and a non-zero exit when neither produced a line for N GPUs."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    argv = [a for a in sys.argv[1:]]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    print(f"[bench] {reason}: {' '.join(cmd)}", file=sys.stderr, flush=True)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    line = last_json_line(r.stdout or "")
    how = "bench.py started its own ranks (torch.distributed.run --nproc-per-node %d)" % args.gpus
    if r.returncode != 0 or line is None or line.get("n_gpus") != args.gpus:
        print(f"[bench] the {args.gpus} ranks did not produce a line (exit code {r.returncode}); the same frames from one process over peer copies",
              file=sys.stderr, flush=True)
        sys.stderr.write((r.stdout or "")[-2000:])
        cmd = [sys.executable, os.path.abspath(__file__)] + argv + ["--launch", "single-process"]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
        line = last_json_line(r.stdout or "")
        how = "one process driving every device (gr_tiled_create_local, peer copies): the ranks could not be started"
        if r.returncode != 0 or line is None or line.get("n_gpus") != args.gpus:
            sys.stderr.write((r.stdout or "")[-2000:])
            print(f"[bench] no line for {args.gpus} GPUs (exit code {r.returncode})", file=sys.stderr, flush=True)
            return r.returncode or 1
    line.setdefault("config", {})["launch"] = how
    print(json.dumps(line), flush=True)
    return 0


def single_process(args):
    """--gpus N from ONE process: a participant per device (gr_tiled_create_local, csrc/tiled.cpp: GR_TRANSPORT_PEER), every participant's
    share of a frame rendered on its own device and stream, its finished blocks copied (hipMemcpyPeerAsync over xGMI) straight to their
    rows of device 0's frame.  The split, the rotation of the shares, frames in flight and the look-ahead are those of the N-rank run; what
    differs is who issues the launches (one host thread for all devices) and the transport.  The fallback of a plain `bench.py --gpus N`
    whose ranks cannot be started, and `--launch single-process`.  GR_BENCH_ONE_DEVICE=peer: every participant on device 0 (rehearsal)."""
    import torch
    import geodesic_raytracing_amd as gra
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    n, seen = args.gpus, torch.cuda.device_count()
    rehearsal = seen < n
    if rehearsal and not os.environ.get("GR_BENCH_ONE_DEVICE"):
        raise SystemExit(f"--gpus {n}: {seen} GPU(s) visible")
    devices = [0] * n if rehearsal else list(range(n))
    W, H = args.width, args.height
    metric = gra.Metric(args.metric, os.path.join(ROOT, "geodesic_raytracing_amd", "scripts"))
    cfg_values = metric.cfg_values(a=args.spin) if "a" in metric.dynamic_vars else metric.cfg_values()
    features = metric.features(adaptive_sampling=0, redshift=args.redshift)
    text = metric.argument_string(features=features, static=(args.program == "static"), cfg_values=cfg_values)
    bg_np, levels = gra.pack_background(gra.synthetic_background(4096, 2048))
    programs, skies = {}, {}
    for d in sorted(set(devices)):
        programs[d] = gra.Program(text, d)
        skies[d] = torch.from_numpy(bg_np).to(torch.device("cuda", d))
    camera = gra.default_camera([float(x) for x in args.camera.split(",")] if args.camera else None)
    parts = gra.TiledFrame.local(devices, W, H, args.block_rows)
    in_flight = max(1, min(args.frames_in_flight, 4))   # a participant stages 4 frames (GR_TILED_STAGING)
    waves = args.trace_waves_per_simd if args.trace_waves_per_simd >= 0 else (2 if n >= 4 else 0)
    states = [[gra.RenderState(W, H, d) for _ in range(in_flight)] for d in devices]
    streams = [[torch.cuda.Stream(device=torch.device("cuda", d)) for _ in range(in_flight)] for d in devices]
    outs = [torch.zeros((H, W, 4), dtype=torch.float32, device=torch.device("cuda", devices[0])) for _ in range(in_flight)]
    lookahead = None if args.no_lookahead else ctypes.pointer(camera)
    depth = 0 if lookahead is None else (args.lookahead_depth if args.lookahead_depth in (1, 2) else 2)
    count = [0]

    def sync():
        for d in sorted(set(devices)):
            torch.cuda.synchronize(d)

    def frame(only=None, **more):
        k = count[0]
        count[0] += 1
        j = k % in_flight
        for r in (range(n) if only is None else [only]):
            o = gra.frame_options(mode=gra.MODE_FUSED, time_kernels=2, **more)
            o.trace_waves_per_simd = waves
            o.fused_shading, o.use_prepass, o.inline_prepass = args.fused_shading, args.use_prepass, args.inline_prepass
            if lookahead is not None:
                o.next_camera = lookahead
                o.next_strip_rank = parts[r].share(k + in_flight)
                if depth == 2:
                    o.next_camera2 = lookahead
                    o.next_strip_rank2 = parts[r].share(k + 2 * in_flight)
            parts[r].render(states[r][j], programs[devices[r]], metric, camera, outs[j].data_ptr(), (skies[devices[r]].data_ptr(), 4096, 2048, levels),
                            features, cfg_values, o, streams[r][j].cuda_stream, rotation=k)
        parts[0].join(streams[0][j].cuda_stream)
        return j

    # the first frame against device 0's own whole frame: a device's share of a split frame equals those rows bit for bit
    j = frame()
    sync()
    alone = torch.zeros_like(outs[0])
    check_state = gra.RenderState(W, H, devices[0])
    check_state.render(programs[devices[0]], metric, camera, alone.data_ptr(), (skies[devices[0]].data_ptr(), 4096, 2048, levels), features, cfg_values,
                       gra.frame_options(mode=gra.MODE_FUSED), streams[0][0].cuda_stream)
    sync()
    if not torch.equal(alone, outs[j]):
        raise SystemExit("[bench] the frame assembled from the devices' shares differs from device 0's own frame")
    del alone, check_state
    for _ in range(max(args.warmup - 1, in_flight)):
        frame()
    sync()
    for row in states:
        for st in row:
            st.trace_log(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frame()
    sync()
    elapsed = time.perf_counter() - t0
    logged = [st.trace_log(reset=True) for row in states for st in row]
    launches = sum(c for _, c in logged)
    ms_per_step = elapsed / args.steps * 1e3
    # participant 0's share on its own, one launch at a time: the kernel's duration the roofline divides by
    stage = {}
    for i in range(4):
        count[0] = 0
        o = gra.frame_options(mode=gra.MODE_FUSED, time_kernels=1, count_attempts=1, inline_prepass=0)
        parts[0].render(states[0][0], programs[devices[0]], metric, camera, outs[0].data_ptr(), (skies[devices[0]].data_ptr(), 4096, 2048, levels),
                        features, cfg_values, o, streams[0][0].cuda_stream, rotation=0)
        sync()
        if i:
            for key, v in states[0][0].stage_ms().items():
                stage.setdefault(key, []).append(v)
    stage = {key: float(np.mean(v)) for key, v in stage.items()}
    plan_rows = sum(1 for b in range((H + args.block_rows - 1) // args.block_rows) if b % n == 0)
    local_pixels = min(plan_rows * args.block_rows, H) * W
    launch_s = max(stage.get("trace", 0.0), 1e-6) * 1e-3
    achieved = TRACE_BYTES_PER_RAY * local_pixels / launch_s / 1e9
    line = {
        "metric": f"Mrays/sec at {W}x{H} {'Kerr' if args.metric == 'kerr_boyer' else args.metric} (whole frame: prepass + init + adaptive Verlet + render-data + anisotropic render)",
        "value": round(W * H / (elapsed / args.steps) / 1e6, 2), "unit": "Mrays/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.metric}{f' (Boyer-Lindquist rs=1 a={args.spin})' if args.metric == 'kerr_boyer' else ''} {W}x{H}, camera (0,0,-4,0) fov 90, "
                               f"adaptive_sampling off, prepass {'on' if metric.info.use_prepass else 'off'}, tol {metric.info.max_acceleration_change:g}, "
                               f"background 4096x2048 RGBA8 10 mips, anisotropy 8",
                   "mode": "fused", "frames_in_flight": in_flight, "trace_waves_per_simd": waves, "prepass_lookahead_depth": depth,
                   "build_key": programs[devices[0]].build_key,
                   "parallelism": f"{args.block_rows}-row blocks, block-cyclic over {n} devices driven by ONE process (assignment rotating per frame); "
                                  f"gr_render_frame_tiled with peer copies (hipMemcpyPeerAsync per block into device 0's frame)"},
        "roofline": {"bound": "hbm", "kernel": "gr_trace_fused", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None, "traffic_source": "counters are collected on one GPU",
                     "algorithmic_bytes_per_launch": TRACE_BYTES_PER_RAY * local_pixels, "avg_launch_ms": round(launch_s * 1e3, 4),
                     "avg_launch_basis": "participant 0's share, launches one at a time after the timed region (HIP events on the launch's stream)",
                     "avg_launch_ms_overlapped": round(sum(ms for ms, _ in logged) / max(launches, 1), 4), "launches_timed": launches,
                     "note": "register-resident ODE integrator: fp32 VALU bound (SURVEY.md 8d); the one-GPU line carries valu_roofline"},
        "cpu_baseline": None, "fps": round(1e3 / ms_per_step, 2),
        "stage_ms_sequential_frame": {key: round(v, 4) for key, v in stage.items()},
        "first_frame_check": "the assembled frame equals device 0's own whole frame bit for bit",
    }
    if rehearsal:
        line["rehearsal"] = f"GR_BENCH_ONE_DEVICE: {n} participants on ONE GPU, one process, peer-copy transport - a run of the single-process N-device path, not a measurement"
    for p in parts:
        p.close()
    print(json.dumps(line), flush=True)


def main():
    args = parse()
    # Most frames of this benchmark repeat one camera.  The library's default for such a frame - take the previous frame's camera set-up and
    # prepass verdicts as they stand (gr_frame_tuning.reuse_still_camera) - would be work skipped inside the timed region: off for every
    # frame here, except under the one label that measures it (frame_one_at_a_time_ms.still_camera_prepass_reused).  Frames that announce no
    # next camera are rendered as for a camera that moves every frame: the prepass inside the trace launch, the tiles on the critical path
    # speculative.  (The work-preserving still-camera form - guess_still_camera: the next frame's prepass on a side stream during this
    # frame's trace - was on in the earlier round-6 lines; it has its own label too, frame_one_at_a_time_ms.still_camera_next_prepass_guessed.)
    os.environ.setdefault("GR_REUSE_STILL_CAMERA", "0")
    world_env = int(os.environ["WORLD_SIZE"]) if os.environ.get("WORLD_SIZE") else None
    seen = 0
    if world_env is None and args.gpus > 1:
        import torch
        seen = torch.cuda.device_count() if torch.cuda.is_available() else 0
    action, reason = plan_launch(args.gpus, world_env, seen, os.environ.get("GR_BENCH_ONE_DEVICE", ""), args.launch)
    if action == "refuse":
        print(f"[bench] {reason}", file=sys.stderr, flush=True)
        raise SystemExit(2)
    if action == "spawn":
        raise SystemExit(spawn_ranks(args, reason))
    if action == "single":
        return single_process(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    rccl_log = None
    if world > 1:
        # RCCL says at start-up which transport every channel takes (P2P/IPC = xGMI between the GPUs of a node, SHM, NET/Socket ...): kept
        # in a file per rank and summarised in the line (`rccl`), so that the first run on a node says whether xGMI carried the frame.
        # Before torch is imported: the library settles its log level the first time anything asks it something.
        import tempfile
        rccl_log = os.path.join(tempfile.gettempdir(), f"gr_bench_rccl.{os.environ.get('MASTER_PORT', '0')}.rank{rank}")
        try:
            os.remove(rccl_log)
        except OSError:
            pass
        if os.environ.get("NCCL_DEBUG", "").upper() not in ("INFO", "TRACE"):
            os.environ["NCCL_DEBUG"] = "INFO"
        os.environ["NCCL_DEBUG_SUBSYS"] = "INIT,P2P,NET,SHM"
        os.environ["NCCL_DEBUG_FILE"] = rccl_log

    import torch
    import torch.distributed as dist
    import geodesic_raytracing_amd as gra
    from geodesic_raytracing_amd import distributed as grd

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # GR_BENCH_ONE_DEVICE=1: a dress rehearsal of the N > 1 run on a box with fewer GPUs than ranks - every rank on device 0, the
    # process group over gloo, and the frame's exchange through the library's inter-process transport (gr_tiled_create_ipc: RCCL's
    # call pattern - a group per frame, a send per block, the matching receives on rank 0 - across real process boundaries; RCCL
    # itself refuses two ranks on one device).  Everything else - rotation of the shares, frames in flight, look-ahead, the complete
    # N > 1 JSON line - is the code the 8-GPU run executes.  The numbers say nothing (N ranks share one GPU): "rehearsal" marks the line.
    # GR_BENCH_ONE_DEVICE=rccl: the same rehearsal through RCCL itself - the process group over "nccl" and the frame's exchange through
    # gr_tiled_create, i.e. EXACTLY the code of the 8-GPU run.  RCCL's "Duplicate GPU detected" check compares (host, bus id): every
    # rank claims a host of its own (NCCL_HOSTID), RCCL takes the ranks for nodes of a cluster and moves the data through its socket
    # transport over the loopback interface.
    rehearsal = os.environ.get("GR_BENCH_ONE_DEVICE", "") if world > 1 else ""
    rccl_rehearsal = rehearsal == "rccl"
    one_device = rehearsal == "1"
    if rccl_rehearsal:
        os.environ.update(NCCL_HOSTID="bench-rank%d" % rank, NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1", NCCL_NET_GDR_LEVEL="0")
    if one_device or rccl_rehearsal:
        local_rank = 0
    if one_device:
        rccl_log = None
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    coll_device = torch.device("cpu") if one_device else device   # where the tensors of the process group's collectives live
    # GR_BENCH_FORCE_DISTRIBUTED=1 drives the multi-GPU code path (process group, strip mode, gather) with one rank
    multi = world > 1 or os.environ.get("GR_BENCH_FORCE_DISTRIBUTED") == "1"
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if one_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    W, H = args.width, args.height
    # the metric comes from the script front-end (scripts/kerr_boyer.js + .json), as in the reference
    metric = gra.Metric(args.metric, os.path.join(ROOT, "geodesic_raytracing_amd", "scripts"))
    cfg_values = metric.cfg_values(a=args.spin) if "a" in metric.dynamic_vars else metric.cfg_values()
    features = metric.features(adaptive_sampling=0, redshift=args.redshift)
    # metric_manager.hpp:19-219: dynamic program first, substituted program (parameters baked in) built in the
    # background and swapped in; the steady state the reference runs in - and the one timed here - is the substituted one
    t_start = time.perf_counter()
    manager = gra.pipeline.ProgramManager(metric, local_rank, features, cfg_values)   # returns with the dynamic program loaded
    t_dynamic_ready = time.perf_counter() - t_start
    program = manager.current(wait=(args.program == "static"))
    t_substituted_ready = time.perf_counter() - t_start
    if args.program == "dynamic":
        program = manager.dynamic
    bg_np, levels = gra.pack_background(gra.synthetic_background(4096, 2048))
    bg = torch.from_numpy(bg_np).to(device)
    camera = gra.default_camera([float(x) for x in args.camera.split(",")] if args.camera else None)

    fused = args.mode == "fused"
    # 48-row blocks: measured per-rank frame time at 4K (tools/strip_probe.py, rotating strips, 3 in flight) 16 -> 48 rows:
    # 0.93 -> 0.87 ms at 8 ranks, 1.56 -> 1.47 at 4, 2.91 -> 2.79 at 2 (half the halo rows, a third of the redundant prepass
    # cells); the coarser deal is evened out by the rotation of the strips over the frames
    plan = grd.StripPlan(H, world, block_rows=args.block_rows)
    # (the reference-shaped sequence is timed one frame at a time unless asked: GR_BENCH_REFERENCE_IN_FLIGHT=1 keeps --frames-in-flight render states busy with it too)
    in_flight = max(1, min(args.frames_in_flight, 8)) if (fused or os.environ.get("GR_BENCH_REFERENCE_IN_FLIGHT", "0") not in ("", "0")) else 1
    # with frames in flight a trace launch takes 4 of the SIMDs' wave slots instead of all (6 for the Kerr kernel): the launches
    # then share the device and one drains while the next is in full swing (measured +2-3 %; one frame at a time: all slots).
    # A rank's share of a frame split 4 or 8 ways is so few tiles that 2 slots are best (tools/strip_probe.py, one of 8 ranks:
    # 0.760 ms per frame with all slots, 0.732 with 4, 0.711 with 3, 0.703 with 2; one of 4: 1.327 / 1.339 / 1.338 / 1.309)
    waves_per_launch = args.trace_waves_per_simd if args.trace_waves_per_simd >= 0 else (
        0 if in_flight < 3 else 4 if world == 1 else 2 if world >= 4 else 0)

    # N > 1: the C ABI's gr_render_frame_tiled (csrc/tiled.cpp) - this rank's share of the rows, then per block an ncclSend /
    # ncclRecv straight to the block's rows of rank 0's frame (no staging, no un-permute).  GR_BENCH_GATHER=torch selects the
    # round-1 path instead (compact strips -> torch.distributed.gather -> un-permute on rank 0), which is also the fallback
    # when the RCCL communicator cannot be built.
    tiled = None
    gather_path = "none"
    if multi:
        gather_path = "torch.distributed.gather + un-permute"
        if os.environ.get("GR_BENCH_GATHER", "rccl") != "torch":
            try:
                if one_device:
                    tiled = gra.TiledFrame.ipc(world, rank, 0, "bench%s" % os.environ["MASTER_PORT"], W, H, args.block_rows)
                    gather_path = ("gr_render_frame_tiled: send / receive per block into the frame through the inter-process transport "
                                   "(GR_BENCH_ONE_DEVICE rehearsal: every rank on device 0)")
                else:
                    uid = [gra.TiledFrame.unique_id() if rank == 0 else None] if world > 1 else [None]
                    if world > 1:
                        dist.broadcast_object_list(uid, src=0)
                    tiled = gra.TiledFrame(world, rank, local_rank, uid[0], W, H, args.block_rows)
                    gather_path = "gr_render_frame_tiled: ncclSend/ncclRecv per block into the frame (RCCL called directly)"
            except Exception as e:   # noqa: BLE001
                print(f"[bench] rank {rank}: gr_tiled_create failed ({e}); using the torch.distributed gather", file=sys.stderr)
                tiled = None
        if world > 1:   # every rank must take the same path
            flag = torch.tensor([1 if tiled is not None else 0], device=coll_device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                tiled, gather_path = None, "torch.distributed.gather + un-permute"

    class Slot:   # one frame in flight: render state (per-frame device buffers), stream, output, gather buffers
        def __init__(self, first):
            self.state = gra.RenderState(W, H, local_rank)
            self.stream = torch.cuda.current_stream() if first and in_flight == 1 else torch.cuda.Stream(device=device)
            self.gather = grd.FrameGather(plan, W, device, rank, world) if multi else None
            # rank 0's frame; in the torch gather path padded to whole blocks so that the un-permute writes it directly
            self.out = None
            if rank == 0:
                self.out = (self.gather.frame_buffer(device) if (multi and tiled is None)
                            else torch.zeros((H, W, 4), dtype=torch.float32, device=device))

    ring = [Slot(i == 0) for i in range(in_flight)]
    state, out, stream = ring[0].state, ring[0].out, ring[0].stream.cuda_stream

    # The RCCL exchange of gr_render_frame_tiled has run with several ranks only over RCCL's socket transport (GR_BENCH_ONE_DEVICE=rccl:
    # ranks that each claim a host of their own; RCCL refuses
    # two ranks on one device), so the first thing an N > 1 run does is check it: one frame through it, compared on rank 0 with the
    # frame rank 0 renders on its own (a device's share of a split frame equals those rows of the whole frame bit for bit, tests/).
    # A frame that differs sends every rank to the round-1 gather; a frame that does not come back within two minutes ends the run
    # with a message instead of hanging it.
    if tiled is not None and world > 1:
        import threading
        done = threading.Event()

        def watchdog():
            if not done.wait(120.0):
                print(f"[bench] rank {rank}: the first gr_render_frame_tiled frame did not complete in 120 s (RCCL send/recv); "
                      "re-run with GR_BENCH_GATHER=torch", file=sys.stderr, flush=True)
                os._exit(17)
        threading.Thread(target=watchdog, daemon=True).start()
        ok = 1
        try:
            o = gra.frame_options(mode=gra.MODE_FUSED)
            tiled.render(state, program, metric, camera, out.data_ptr() if rank == 0 else None, (bg.data_ptr(), 4096, 2048, levels),
                         features, cfg_values, o, stream, rotation=0)
            torch.cuda.synchronize()
            dist.barrier()
            if rank == 0:
                alone = torch.zeros((H, W, 4), dtype=torch.float32, device=device)
                check_state = gra.RenderState(W, H, local_rank)
                check_state.render(program, metric, camera, alone.data_ptr(), (bg.data_ptr(), 4096, 2048, levels), features, cfg_values,
                                   gra.frame_options(mode=gra.MODE_FUSED), stream)
                torch.cuda.synchronize()
                ok = int(torch.equal(alone, out))
                del alone, check_state
        except Exception as e:   # noqa: BLE001
            print(f"[bench] rank {rank}: gr_render_frame_tiled failed on its first frame ({e})", file=sys.stderr)
            ok = 0
        done.set()
        flag = torch.tensor([ok], device=coll_device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if rank == 0:
                print("[bench] the frame assembled by gr_render_frame_tiled differs from rank 0's own frame; using the "
                      "torch.distributed gather", file=sys.stderr)
            tiled, gather_path = None, "torch.distributed.gather + un-permute (gr_render_frame_tiled failed its self-check)"
            for slot in ring:
                if rank == 0:
                    slot.out = slot.gather.frame_buffer(device)
            out = ring[0].out
        else:
            gather_path += "; first frame checked against rank 0's own frame"
    frame_index = [0]
    globals_features = [features]

    # a batch renderer knows the next frames' cameras: their tetrad + prepass (1.2 ms of pure latency, 507 waves) run on
    # high-priority side streams while the current frame traces (gr_frame_options.next_camera / next_camera2).  Here every
    # frame uses one camera.
    lookahead = ctypes.pointer(camera) if (fused and not args.no_lookahead) else None
    depth = 0 if lookahead is None else (args.lookahead_depth if args.lookahead_depth in (1, 2) else (2 if world > 1 else 1))

    def frame(prog=None, cfgv=None, transfer=True, feats=None, use_prepass=None, reference_shaped=False):
        prog = program if prog is None else prog
        cfgv = cfg_values if cfgv is None else cfgv
        features = feats if feats is not None else globals_features[0]
        slot = ring[frame_index[0] % in_flight]
        frame_index[0] += 1
        with torch.cuda.stream(slot.stream):
            if not multi:
                opts = gra.frame_options(mode=gra.MODE_FUSED if (fused and not reference_shaped) else gra.MODE_REFERENCE, tiled=1, time_kernels=2)
                target = slot.out.data_ptr()
            else:
                # this rank's row blocks (block-cyclic) -> compact strip buffer -> ONE gather to rank 0 + local un-permute.  The
                # strip a rank renders rotates with the frame number: strips differ in cost (1.09 max/mean at 8 ranks) and with
                # frames in flight and an asynchronous gather the ranks then run at the mean, not at the slowest strip.
                k = frame_index[0] - 1
                opts = gra.frame_options(mode=gra.MODE_FUSED, strip_rank=slot.gather.strip_of(k), strip_count=world,
                                         block_rows=plan.block_rows, compact_out=1, time_kernels=2)
                opts.next_strip_rank = slot.gather.strip_of(k + in_flight)        # the frames this render state sees next
                opts.next_strip_rank2 = slot.gather.strip_of(k + 2 * in_flight)
                target = slot.gather.local_buffer().data_ptr()
            if args.measure_clock:
                opts.count_attempts = 1
            opts.trace_waves_per_simd = waves_per_launch
            opts.fused_shading = args.fused_shading
            opts.use_prepass = args.use_prepass if use_prepass is None else use_prepass
            opts.inline_prepass = args.inline_prepass
            if args.no_lookahead:
                opts.guess_still_camera = 0
            if lookahead is not None and not reference_shaped:
                opts.next_camera = lookahead
                if depth == 2:
                    opts.next_camera2 = lookahead
            if tiled is not None and transfer:
                k = frame_index[0] - 1
                opts.next_strip_rank = tiled.share(k + in_flight)
                opts.next_strip_rank2 = tiled.share(k + 2 * in_flight)
                tiled.render(slot.state, prog, metric, camera, slot.out.data_ptr() if rank == 0 else None, (bg.data_ptr(), 4096, 2048, levels),
                             features, cfgv, opts, slot.stream.cuda_stream, rotation=k)
                return
            slot.state.render(prog, metric, camera, target, (bg.data_ptr(), 4096, 2048, levels), features, cfgv, opts,
                              slot.stream.cuda_stream)
            if multi and transfer:
                slot.gather.submit(slot.out, rotation=frame_index[0] - 1)   # asynchronous: overlaps the following frames

    def barrier():
        if multi and tiled is not None:
            torch.cuda.synchronize()   # sends / receives are ordered on the frames' streams
            dist.barrier()
        elif multi:
            for slot in ring:
                with torch.cuda.stream(slot.stream):
                    slot.gather.drain(slot.out)   # every frame submitted so far is gathered and assembled on rank 0
            torch.cuda.synchronize()
            dist.barrier()
        torch.cuda.synchronize()

    t_first = time.perf_counter()
    frame()
    torch.cuda.synchronize()
    first_frame_s = time.perf_counter() - t_first
    for _ in range(max(0, args.warmup - 1)):
        frame()
    # every render state of the ring must have rendered once before the clock starts (buffers touched, its look-ahead slots
    # filled); with fewer warm-up steps than states the missing ones are rendered here, untimed, and reported as priming_frames
    priming = max(0, in_flight - args.warmup)
    for _ in range(priming):
        frame()
    barrier()
    for slot in ring:
        slot.state.trace_log(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frame()
    barrier()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    overlapped_clock = [round(slot.state.shader_clock_mhz(), 1) for slot in ring] if args.measure_clock else None
    ms_per_step = elapsed / args.steps * 1e3
    mrays = W * H / (elapsed / args.steps) / 1e6
    # every trace launch of the timed region, HIP events on the stream it was launched on
    logged = [slot.state.trace_log(reset=True) for slot in ring]
    launches = sum(n for _, n in logged)
    avg_launch_s = sum(ms for ms, _ in logged) / max(launches, 1) * 1e-3

    without_transfer_s = None
    if multi:
        # the same frames with every rank rendering its (rotating) share and nothing sent - what the exchange costs on
        # the critical path is the difference
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            frame(transfer=False)
        torch.cuda.synchronize()
        mine = time.perf_counter() - t1
        t = torch.tensor([mine], dtype=torch.float64, device=coll_device)
        everyone = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(everyone, t)
        per_rank_without = [float(x.item()) / args.steps * 1e3 for x in everyone]
        without_transfer_s = max(per_rank_without) * 1e-3
        for slot in ring:
            slot.state.trace_log(reset=True)

    # roofline of the dominant kernel.  Launches of frames in flight overlap on the GPU, so their durations say nothing about one
    # launch's cost; the per-launch duration the roofline divides by comes from launches run one at a time right after the timed
    # region (same program, same frame; HIP events on the launch's own stream) and is reported next to the overlapped average.
    extra = {}
    local_pixels = W * H if world == 1 else sum(b - a for a, b in plan.blocks_of(rank)) * W + plan.local_blocks(rank) * W
    headline_alg_bytes = (TRACE_BYTES_PER_RAY if fused else 140) * local_pixels
    extra["fps"] = round(1e3 / ms_per_step, 2)
    if without_transfer_s:
        extra["fps_without_transfer"] = round(1.0 / without_transfer_s, 2)
        extra["per_rank_ms_per_frame_without_transfer"] = [round(x, 4) for x in per_rank_without]
        # how long a frame waits for rows to arrive, beyond the slowest rank's rendering: the exchange on the critical path
        extra["gather_wait_ms_per_frame"] = round(ms_per_step - without_transfer_s * 1e3, 4)
    if overlapped_clock:
        extra["shader_clock_mhz_last_overlapped_launches"] = overlapped_clock

    class Workload:   # what a roofline block is measured on: the headline's by default, another configuration's under `secondary`
        def __init__(self, metric, state, camera, features, target, pixels):
            self.metric, self.state, self.camera, self.features, self.target, self.pixels = metric, state, camera, features, target, pixels

    def exclusive_frames(prog, cfgv, n=5, wl=None, inline_prepass=0, guess=0, reuse=0):
        """n frames one at a time with per-stage events and the attempt / shader-clock counters: stage ms (means), attempts,
        MHz.  inline_prepass = 0: the prepass as a launch of its own, so that the trace stage is the trace kernel's own work (what
        the roofline blocks describe); -1: as the library renders a frame whose camera was not announced (prepass inside the trace
        launch)"""
        stage_sum, attempts, clocks, shares = {}, 0, [], []
        for _ in range(n):
            if wl is not None:
                opts = gra.frame_options(mode=gra.MODE_FUSED, time_kernels=1, count_attempts=1, inline_prepass=inline_prepass, guess_still_camera=guess, reuse_still_camera=reuse)
                wl.state.render(prog, wl.metric, wl.camera, wl.target, (bg.data_ptr(), 4096, 2048, levels), wl.features, cfgv, opts, stream)
            elif multi:
                opts = gra.frame_options(mode=gra.MODE_FUSED, strip_rank=rank, strip_count=world, block_rows=plan.block_rows, compact_out=1,
                                         time_kernels=1, count_attempts=1)
                target = ring[0].gather.local_buffer().data_ptr()
            else:
                opts = gra.frame_options(mode=gra.MODE_FUSED if fused else gra.MODE_REFERENCE, tiled=1, time_kernels=1, count_attempts=1,
                                         use_prepass=args.use_prepass, inline_prepass=inline_prepass, guess_still_camera=guess, reuse_still_camera=reuse)
                target = out.data_ptr()
            if wl is None:
                state.render(prog, metric, camera, target, (bg.data_ptr(), 4096, 2048, levels), features, cfgv, opts, stream)
            torch.cuda.synchronize()
            st = wl.state if wl is not None else state
            attempts = st.attempts()
            clocks.append(st.shader_clock_mhz())
            stages_now = st.stage_ms()
            for k, v in stages_now.items():
                stage_sum.setdefault(k, []).append(v)
            wave_ms, waves = st.wave_time()   # fused trace only: summed wave lifetimes -> share of the launch's slots occupied
            if waves and stages_now.get("trace"):
                shares.append(wave_ms / waves / stages_now["trace"])
        exclusive_frames.slot_share = float(np.mean(shares[1:])) if len(shares) > 1 else None
        return {k: float(np.mean(v[1:])) for k, v in stage_sum.items()}, attempts, float(np.mean(clocks[1:]))

    def roofline_blocks(prog, cfgv, tag, wall_s_per_frame, overlapped_launch_s=None, launches=0, wl=None):
        """the contract's HBM roofline object and the binding fp32-VALU one for one workload"""
        stages, attempts, mhz = exclusive_frames(prog, cfgv, wl=wl)
        launch_s = stages["trace"] * 1e-3
        alg_bytes = TRACE_BYTES_PER_RAY * wl.pixels if wl is not None else headline_alg_bytes
        info = (wl.metric if wl is not None else metric).info
        accel_ops, coord_ops = info.accel_ops, info.coord_ops
        if args.program == "static" or wl is not None:   # the substituted program's DAG (parameters folded in) is what runs
            accel_ops, _, coord_ops = (wl.metric if wl is not None else metric).substituted_op_counts(cfgv)
        achieved = alg_bytes / launch_s / 1e9
        pmc, pmc_note = committed_counters(tag, prog.build_key) if (fused and world == 1) else (None, "counters are collected on one GPU")
        roof = {"bound": "hbm", "kernel": "gr_trace_fused" if fused else "gr_do_generic_rays", "achieved": round(achieved, 3),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                "traffic": pmc.get("hbm_bytes_per_launch") if pmc else None, "traffic_source": pmc_note,
                "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(launch_s * 1e3, 4),
                "avg_launch_basis": "launches run one at a time after the timed region (4 frames, HIP events on the launch's stream); such launches hand "
                                    "their tiles out dearest first by the previous frame's costs (gr_frame_options.tile_history), launches of frames in flight in image order",
                "shader_clock_mhz_during_launch": round(mhz, 1),
                "wave_slot_occupancy_of_launch": round(exclusive_frames.slot_share, 3) if exclusive_frames.slot_share else None,
                "note": "register-resident ODE integrator: fp32 VALU bound, see valu_roofline (SURVEY.md 8d)"}
        if overlapped_launch_s is not None:
            roof["avg_launch_ms_overlapped"] = round(overlapped_launch_s * 1e3, 4)
            roof["launches_timed"] = launches
            roof["concurrent_launches"] = in_flight
        # fp32 FLOP of one trace launch: counted by the hardware (SQ_INSTS_VALU_{ADD,MUL,FMA x2,TRANS}_F32 x 64 lanes) when the
        # committed counters belong to this build; otherwise the code generator's operation count (every DAG node + a fixed 90
        # for integrator and controller: more than the compiled loop executes), and the line says which
        model_flops_per_attempt = accel_ops + coord_ops + STEP_OVERHEAD_FLOPS
        counted = pmc.get("fp32_flop_per_launch") if pmc else None
        flop_per_frame = counted if counted else model_flops_per_attempt * attempts
        tflops_wall = flop_per_frame / wall_s_per_frame / 1e12        # all stages, the way frames are produced
        tflops_launch = flop_per_frame / launch_s / 1e12              # the kernel on its own
        peak_at_clock = VALU_PEAK_TFLOPS * mhz / 2400.0 if mhz > 0 else None
        valu = {"achieved": round(tflops_wall, 3), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tflops_wall / VALU_PEAK_TFLOPS, 4),
                "basis": "fp32 FLOP of the trace launch / wall clock per frame of the timed region (all stages)",
                "kernel_alone": {"achieved": round(tflops_launch, 3), "frac": round(tflops_launch / VALU_PEAK_TFLOPS, 4),
                                 "frac_of_peak_at_measured_clock": round(tflops_launch / peak_at_clock, 4) if peak_at_clock else None,
                                 "shader_clock_mhz": round(mhz, 1), "peak_at_measured_clock": round(peak_at_clock, 1) if peak_at_clock else None},
                "flop_per_frame": int(flop_per_frame), "flop_source": pmc_note if counted else "code generator model (" + pmc_note + ")",
                "flops_per_attempt": round(flop_per_frame / max(attempts, 1), 1), "flops_per_attempt_codegen_model": model_flops_per_attempt,
                "valu_instructions_per_attempt": round(pmc["valu_wave_instructions_per_launch"] * 64 / max(attempts, 1), 1) if pmc and pmc.get("valu_wave_instructions_per_launch") else None,
                "step_attempts_per_frame": int(attempts),
                # SURVEY 8d "wave efficiency": active lanes per issued VALU instruction (hardware counters)
                "lane_utilisation": pmc.get("valu_lane_utilisation") if pmc else None}
        # the counted FLOP are SQ_INSTS_VALU_* x 64 lanes, masked-off lanes included: what the live lanes did is frac x lane_utilisation
        lanes = valu["lane_utilisation"] if counted else None
        valu["useful_frac"] = round(valu["frac"] * lanes, 4) if lanes else None
        valu["kernel_alone"]["useful_frac"] = round(valu["kernel_alone"]["frac"] * lanes, 4) if lanes else None
        roof["binding"] = {"bound": "fp32 VALU (no MFMA, no HBM traffic to speak of)", "achieved": round(tflops_wall, 3), "peak": VALU_PEAK_TFLOPS,
                           "unit": "TFLOP/s", "frac": round(tflops_wall / VALU_PEAK_TFLOPS, 4)}
        # ... and the frame as the library renders it one at a time when the next camera is not known (prepass inside the trace launch)
        as_rendered, _, _ = exclusive_frames(prog, cfgv, n=4, wl=wl, inline_prepass=-1)
        # ... when the camera has stopped moving, with the next frame's set-up + prepass on the side stream while this frame traces, used
        # because the next frame's camera turns out to be the same (gr_frame_tuning.guess_still_camera: every frame still computes a prepass)
        still, _, _ = exclusive_frames(prog, cfgv, n=6, wl=wl, inline_prepass=-1, guess=1)
        # ... and as the library renders a repeated frame by default (reuse_still_camera): the previous frame's set-up and prepass verdicts
        # taken as they stand - NOT computed in these frames, which is why no other figure of this line is measured that way
        reused, _, _ = exclusive_frames(prog, cfgv, n=6, wl=wl, inline_prepass=-1, guess=0, reuse=1)
        roof["frame_one_at_a_time_ms"] = {"prepass_as_its_own_launch": round(sum(stages.values()), 4),
                                          "prepass_inside_the_trace_launch": round(sum(as_rendered.values()), 4),
                                          "still_camera_next_prepass_guessed": round(sum(still.values()), 4),
                                          "still_camera_prepass_reused": round(sum(reused.values()), 4)}
        return roof, valu, stages

    kerr_4k = args.metric == "kerr_boyer" and (W, H) == (3840, 2160)
    tag = ("kerr_a045_4k" if abs(args.spin - 0.45) < 1e-9 else "kerr_a09_4k" if abs(args.spin - 0.9) < 1e-9 else None) if kerr_4k else None
    if tag and not fused:
        tag = "reference_sequence_4k" if abs(args.spin - 0.45) < 1e-9 else None
    if tag and args.program == "dynamic":
        tag += "_dynamic"
    if args.config in OTHER_CONFIGS:
        tag = OTHER_CONFIGS[args.config]["tag"]
    tag = tag or f"{args.metric}_{W}x{H}"
    roofline, valu, stages = roofline_blocks(program, cfg_values, tag, elapsed / args.steps, avg_launch_s, launches)
    if world > 1:   # every rank's trace launch: as timed in the overlapped region, and on its own
        mine = torch.tensor([avg_launch_s * 1e3, stages.get("trace", 0.0), stages.get("prepass", 0.0)], dtype=torch.float64, device=coll_device)
        everyone = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(everyone, mine)
        extra["per_rank_trace_launch_ms"] = {"overlapped": [round(float(t[0]), 4) for t in everyone],
                                             "one_at_a_time": [round(float(t[1]), 4) for t in everyone],
                                             "prepass_one_at_a_time": [round(float(t[2]), 4) for t in everyone]}
    extra["valu_roofline"] = valu
    extra["stage_ms_sequential_frame"] = {k: round(v, 4) for k, v in stages.items()}
    if multi:
        # per-rank view of the split (rank 0's): what the SCALE run needs to tell imbalance from communication
        extra["rank0_rows"] = sum(b - a for a, b in plan.blocks_of(rank))
    if not multi:
        rd = np.empty(W * H, dtype=gra.pipeline.RENDER_DATA_DTYPE)
        gra.check(gra.lib.gr_device_download(local_rank, rd.ctypes.data_as(ctypes.c_void_p), state.buffer(gra.BUF_RENDER_DATA), rd.nbytes))
        skipped = int((rd["terminated"] == 2).sum())
        extra["traced_Mrays_per_s"] = round((W * H - skipped) / (elapsed / args.steps) / 1e6, 2)
        extra["prepass_skipped_fraction"] = round(skipped / (W * H), 4)

        # secondary figures SURVEY.md 8d asks for (same resolution, outside the headline timing)
        def timed(cam, feats, cfg, prog, mode, n=8, **tuning):
            opts = gra.frame_options(mode=mode, tiled=1, **tuning)
            for _ in range(3):
                state.render(prog, metric, cam, out.data_ptr(), (bg.data_ptr(), 4096, 2048, levels), feats, cfg, opts, stream)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                state.render(prog, metric, cam, out.data_ptr(), (bg.data_ptr(), 4096, 2048, levels), feats, cfg, opts, stream)
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / n

        def steady(submit, finish, warm=3, n=12, repeats=3):
            """seconds per frame of a steady run: `warm` untimed frames (buffers touched, tile history settled, clocks up), `n` timed, `repeats`
            times - the median, and all repeats so that a line shows its own spread (r05 timed 3 frames after 1 and put a 20 % swing on record)"""
            reps = []
            for _ in range(repeats):
                for _ in range(warm):
                    submit()
                finish()
                t = time.perf_counter()
                for _ in range(n):
                    submit()
                finish()
                reps.append((time.perf_counter() - t) / n)
            return float(np.median(reps)), [round(x * 1e3, 4) for x in reps]

        secondary = {}
        if args.no_secondary:
            timed = None
        if timed and "a" in metric.dynamic_vars:
            t = timed(gra.default_camera([0, 0, -15, 0]), features, cfg_values, manager.dynamic, gra.MODE_FUSED)
            secondary["far_pose_camera_r15_Mrays_per_s"] = round(W * H / t / 1e6, 1)
            t = timed(camera, features, metric.cfg_values(a=0.9), manager.dynamic, gra.MODE_FUSED)
            secondary["superextremal_a0.9_Mrays_per_s"] = round(W * H / t / 1e6, 1)
            # BASELINE.json configs[2] read literally ($cfg.a = 0.9, a naked singularity in the script's rs = 2M units), measured the
            # way the headline is: substituted program, frames in flight
            cfg09 = metric.cfg_values(a=0.9)
            prog09 = gra.Program(metric.argument_string(features=features, static=True, cfg_values=cfg09), local_rank)
            t, reps09 = steady(lambda: frame(prog09, cfg09), barrier, warm=in_flight + 1)
            secondary["superextremal_a0.9_substituted_pipelined_Mrays_per_s"] = round(W * H / t / 1e6, 1)
            # ... with its own roofline objects: every pixel is traced here (no shadow for the prepass to skip)
            roof09, valu09, stages09 = roofline_blocks(prog09, cfg09, "kerr_a09_4k", t)
            secondary["superextremal_a0.9_substituted"] = {"Mrays_per_s": round(W * H / t / 1e6, 1), "fps": round(1 / t, 2), "ms_per_frame_repeats": reps09, "roofline": roof09,
                                                          "valu_roofline": valu09, "stage_ms_sequential_frame": {k: round(v, 4) for k, v in stages09.items()}}
            # ... and the same frames without the prepass (gr_frame_options.use_prepass = 0): at a = 0.9 it skips 3 % of the pixels - above the
            # 2 % under which the opt-in policy (use_prepass = -2) would drop it by itself - for 6.7 ms of single-ray latency, which frames in
            # flight hide and a frame on its own pays unless it rides in the trace launch.  Not output-neutral: the skipped pixels are traced.
            t, reps = steady(lambda: frame(prog09, cfg09, use_prepass=0), barrier, warm=in_flight + 1)
            secondary["superextremal_a0.9_substituted"]["without_prepass"] = {"Mrays_per_s": round(W * H / t / 1e6, 1), "fps": round(1 / t, 2), "ms_per_frame_repeats": reps}
            for slot in ring:
                slot.state.trace_log(reset=True)
        if timed and args.program == "static":
            # the headline frame through the DYNAMIC program - what a user sees for the seconds between a parameter change and the swap of the
            # substituted program (metric_manager.hpp:19-170) - measured the way the headline is (frames in flight, look-ahead)
            dyn = manager.dynamic
            t, reps = steady(lambda: frame(dyn, cfg_values), barrier, warm=in_flight + 1)
            pmc_dyn, pmc_dyn_note = committed_counters("kerr_a045_4k_dynamic", dyn.build_key) if tag == "kerr_a045_4k" else (None, "counters are collected for the headline workload")
            secondary["dynamic_program_fused"] = {"Mrays_per_s": round(W * H / t / 1e6, 1), "fps": round(1 / t, 2), "ms_per_frame_repeats": reps, "build_key": dyn.build_key,
                                                  "trace_kernel": dyn.kernel_info("gr_trace_fused"),
                                                  "hbm_bytes_per_launch": pmc_dyn.get("hbm_bytes_per_launch") if pmc_dyn else None,
                                                  "valu_instructions_per_launch": pmc_dyn.get("valu_wave_instructions_per_launch") if pmc_dyn else None, "counters": pmc_dyn_note}
            for slot in ring:
                slot.state.trace_log(reset=True)
        if timed:
            t = timed(camera, metric.features(adaptive_sampling=1, adaptive_sampling_threshold=32.0), cfg_values, manager.dynamic, gra.MODE_REFERENCE)
            secondary["adaptive_sampling_on_threshold32_fps"] = round(1 / t, 1)
            t = timed(camera, features, cfg_values, manager.dynamic, gra.MODE_REFERENCE)
            secondary["reference_kernel_sequence_dynamic_program_fps"] = round(1 / t, 1)
            # The drop-in path as a maintainer binds it (one launch per reference kernel, 96-byte ray records in HBM, rays
            # in 8x8-tile slot order) with the roofline SURVEY.md 8d defines for THAT Verlet kernel: 140 B per ray (96 read + 12 of the
            # header re-read + 32 written back on termination) over the time of gr_do_generic_rays; counter traffic of the same launches:
            # profiles/r05_pmc_refseq.txt, profiles/pmc_reference_sequence_4k.json (tools/final_profiles.sh r05 refseq)
            def reference_sequence(prog, wall_s):
                acc = {}
                for _ in range(5):
                    state.render(prog, metric, camera, out.data_ptr(), (bg.data_ptr(), 4096, 2048, levels), features, cfg_values,
                                 gra.frame_options(mode=gra.MODE_REFERENCE, tiled=1, time_kernels=1, count_attempts=1), stream)
                    torch.cuda.synchronize()
                    for k, v in state.stage_ms().items():
                        acc.setdefault(k, []).append(v)
                stage = {k: round(float(np.mean(v[1:])), 4) for k, v in acc.items()}
                trace_s = stage["trace"] * 1e-3
                attempts_ref = int(state.attempts())
                flops_per_attempt = valu["flops_per_attempt"]
                hbm = 140 * W * H / trace_s / 1e9
                pmc_ref, pmc_ref_note = committed_counters("reference_sequence_4k" + ("" if prog is program else "_dynamic"), prog.build_key)
                return {"fps": round(1 / wall_s, 1), "ms_per_frame": round(wall_s * 1e3, 3), "stage_ms": stage,
                        "roofline": {"bound": "hbm", "kernel": "gr_do_generic_rays", "achieved": round(hbm, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": round(hbm / HBM_PEAK_GBS, 5), "algorithmic_bytes_per_launch": 140 * W * H, "launch_ms": stage["trace"],
                                     "traffic": pmc_ref.get("hbm_bytes_per_launch") if pmc_ref else None, "traffic_source": pmc_ref_note},
                        "valu_frac": round(flops_per_attempt * attempts_ref / trace_s / 1e12 / VALU_PEAK_TFLOPS, 4),
                        "step_attempts_per_frame": attempts_ref, "trace_over_fused_trace": round(stage["trace"] / max(stages.get("trace", 0.0), 1e-9), 3)}
            secondary["reference_kernel_sequence"] = {"dynamic_program": reference_sequence(manager.dynamic, t)}
            t = timed(camera, features, cfg_values, program, gra.MODE_REFERENCE)
            secondary["reference_kernel_sequence"]["substituted_program"] = reference_sequence(program, t)
            if not multi and in_flight > 1:
                # ... and the way the reference's main loop would run it: a ring of render states (main.cpp:1463-1469), here the headline's
                # ring - each frame's fourteen launches on its state's stream, the next frame's filling their gaps (one frame at a time
                # the device clocks down between them: 2.10 against 2.23 GHz, EXPERIMENTS.md round 6)
                t, reps = steady(lambda: frame(program, cfg_values, reference_shaped=True), barrier, warm=in_flight + 1)
                secondary["reference_kernel_sequence"]["substituted_program"]["frames_in_flight"] = {
                    "render_states": in_flight, "ms_per_frame": round(t * 1e3, 3), "fps": round(1 / t, 1), "Mrays_per_s": round(W * H / t / 1e6, 1),
                    "ms_per_frame_repeats": reps}
                for slot in ring:
                    slot.state.trace_log(reset=True)
            # the reference's own speed-up on the fused path with its substituted program: a quarter of the primary rays, the blocks
            # that need it refined (cl.cl:5223-5345), one frame at a time
            fa = metric.features(adaptive_sampling=1, adaptive_sampling_threshold=32.0)
            pa = gra.Program(metric.argument_string(features=fa, static=True, cfg_values=cfg_values), local_rank)
            t = timed(camera, fa, cfg_values, pa, gra.MODE_FUSED)
            secondary["adaptive_sampling_on_threshold32_fused_substituted_fps"] = round(1 / t, 1)
            # (nothing known or guessed about the next camera - what a camera that moves every frame gets: the prepass cells inside the lattice
            # launch, its tiles by the frame before's lattice costs, the first classes speculative)
            t_unannounced = t
            t = timed(camera, fa, cfg_values, pa, gra.MODE_FUSED, guess_still_camera=1)   # (the still-camera guess of the earlier round-6 lines)
            # ... its stages (one frame at a time: lattice launch with the prepass cells in front of its tiles, decisions + the list of
            # marked pixels + the launch over that list, texture pass), what it traces, and the same frames the way the headline is
            # measured (frames in flight, prepass look-ahead): the throughput adaptive sampling exists for
            stage_acc, traced = {}, 0
            for _ in range(4):
                state.render(pa, metric, camera, out.data_ptr(), (bg.data_ptr(), 4096, 2048, levels), fa, cfg_values,
                             gra.frame_options(mode=gra.MODE_FUSED, time_kernels=1, count_attempts=1), stream)
                torch.cuda.synchronize()
                for k, v in state.stage_ms().items():
                    stage_acc.setdefault(k, []).append(v)
            marked = np.empty(1, dtype=np.int32)
            gra.check(gra.lib.gr_device_download(local_rank, marked.ctypes.data_as(ctypes.c_void_p), state.buffer(gra.BUF_RAYS_ADAPTIVE_COUNT), 4))
            for _ in range(in_flight + 1):
                frame(pa, cfg_values, feats=fa)
            barrier()
            tp = time.perf_counter()
            for _ in range(12):
                frame(pa, cfg_values, feats=fa)
            barrier()
            tp = (time.perf_counter() - tp) / 12
            secondary["adaptive_sampling_on_threshold32_fused_substituted"] = {
                "one_frame_at_a_time_ms": round(t_unannounced * 1e3, 3), "one_frame_at_a_time_next_prepass_guessed_ms": round(t * 1e3, 3), "pipelined_ms_per_frame": round(tp * 1e3, 3), "pipelined_Mpixels_per_s": round(W * H / tp / 1e6, 1),
                "speed_up_over_every_pixel_pipelined": round(ms_per_step / (tp * 1e3), 3),
                "stage_ms": {k: round(float(np.mean(v[1:])), 4) for k, v in stage_acc.items()},
                "stage_note": "trace = the lattice launch (a quarter of the pixels; the prepass cells are its first tickets), adaptive = gr_adaptive_refine_list + gr_trace_pending",
                "step_attempts_per_frame": int(state.attempts()), "marked_pixels": int(marked[0]), "marked_fraction": round(float(marked[0]) / (W * H), 4)}
            for slot in ring:
                slot.state.trace_log(reset=True)
            del pa
            # the other BASELINE.json configurations, one frame at a time on this one GPU (substituted programs, fused kernel)
            scripts_dir = os.path.join(ROOT, "geodesic_raytracing_amd", "scripts")
            for label, name, (cw, ch), cam_pos, feats_kw, counters_tag in (
                    ("config1_schwarzschild_1920x1080", "schwarzschild", (1920, 1080), None, {}, None),                  # as shipped: fixed step
                    ("config1_schwarzschild_adaptive_1920x1080", "schwarzschild_adaptive", (1920, 1080), None, {}, None),  # as worded: adaptive
                    *[(c["label"], c["metric"], c["size"], c["camera"], c["features"], c["tag"]) for c in OTHER_CONFIGS.values()]):
                m2 = gra.Metric(name, scripts_dir)
                f2 = m2.features(adaptive_sampling=0, **feats_kw)
                p2 = gra.Program(m2.argument_string(features=f2, static=True, cfg_values=m2.cfg_values()), local_rank)
                st2 = gra.RenderState(cw, ch, local_rank)
                out2 = torch.zeros((ch, cw, 4), dtype=torch.float32, device=device)
                c2 = gra.default_camera(cam_pos)
                o2 = gra.frame_options(mode=gra.MODE_FUSED)

                def once():
                    st2.render(p2, m2, c2, out2.data_ptr(), (bg.data_ptr(), 4096, 2048, levels), f2, m2.cfg_values(), o2, stream)
                t, reps = steady(once, torch.cuda.synchronize)
                secondary[label] = {"Mrays_per_s": round(cw * ch / t / 1e6, 1), "fps": round(1 / t, 1), "ms_per_frame": round(t * 1e3, 4), "ms_per_frame_repeats": reps,
                                    "timing": "one frame at a time on one render state; 3 warm-up + 12 timed frames, median of three repeats"}
                if counters_tag:   # BASELINE configs[3] / [4]: their own roofline objects (one frame at a time on this GPU)
                    wl = Workload(m2, st2, c2, f2, out2.data_ptr(), cw * ch)
                    roof2, valu2, stages2 = roofline_blocks(p2, m2.cfg_values(), counters_tag, t, wl=wl)
                    secondary[label].update({"roofline": roof2, "valu_roofline": valu2, "stage_ms_sequential_frame": {k: round(v, 4) for k, v in stages2.items()},
                                             "build_key": p2.build_key, "trace_kernel": p2.kernel_info("gr_trace_fused")})
                del st2, out2
            # The reference's one published claim is "1080 at 30fps+" for "the vast majority" of its metrics (README.md:5, GUI defaults:
            # adaptive sampling on, threshold 32).  Measured now: every metric script this repository ships, 1920x1080, one frame at a time, the
            # dynamic and the substituted program.  The full table over the reference's own 31 scripts (needs the build container's
            # manifest of their generated strings) is tools/all_metrics_bench.py -> profiles/r06_all_metrics_1080p.txt, quoted beside it.
            import glob
            rows = {}
            for js in sorted(glob.glob(os.path.join(scripts_dir, "*.js"))):
                name = os.path.basename(js)[:-3]
                m3 = gra.Metric(name, scripts_dir)
                f3 = m3.features(adaptive_sampling=1, adaptive_sampling_threshold=32.0)
                st3 = gra.RenderState(1920, 1080, local_rank)
                out3 = torch.zeros((1080, 1920, 4), dtype=torch.float32, device=device)
                c3 = gra.default_camera()
                for kind, text in (("dynamic", m3.argument_string()), ("substituted", m3.argument_string(features=f3, static=True, cfg_values=m3.cfg_values()))):
                    p3 = gra.Program(text, local_rank)
                    t, _ = steady(lambda: st3.render(p3, m3, c3, out3.data_ptr(), (bg.data_ptr(), 4096, 2048, levels), f3, m3.cfg_values(), gra.frame_options(mode=gra.MODE_FUSED), stream),
                                  torch.cuda.synchronize, warm=3, n=8, repeats=1)
                    rows.setdefault(name, {})[kind] = round(1 / t, 1)
                    del p3
                del st3, out3
            slowest = min(rows, key=lambda k: min(rows[k].values()))
            full = None
            try:
                for text in open(os.path.join(ROOT, "profiles", "r06_all_metrics_1080p.txt")):
                    if text.startswith("# summary "):
                        full = json.loads(text[len("# summary "):])
            except (OSError, ValueError):
                pass
            secondary["all_reference_scripts"] = {
                "measured_now": {"what": "the metric scripts this repository ships, 1920x1080, GUI defaults (adaptive sampling on, threshold 32), one frame at a time",
                                 "scripts": len(rows), "min_fps": min(min(v.values()) for v in rows.values()),
                                 "median_fps": float(np.median([min(v.values()) for v in rows.values()])), "slowest": slowest, "fps": rows},
                "the_reference's_31_scripts": {"source": "profiles/r06_all_metrics_1080p.txt (tools/all_metrics_bench.py, this round's GPU box; not re-measured in this run)",
                                               "summary": full}}
            extra["secondary"] = secondary

    startup = None
    if rank == 0 and world == 1:
        # what a user waits for, warm cache (the code objects are in geodesic_raytracing_amd/_cache): library + dynamic program loaded,
        # substituted program swapped in, first frame on the screen (its buffers allocated, its kernels loaded)
        startup = {"dynamic_program_ready_s": round(t_dynamic_ready, 3), "substituted_program_ready_s": round(t_substituted_ready, 3),
                   "first_frame_s": round(first_frame_s, 3), "time_to_first_frame_s": round(t_substituted_ready + first_frame_s, 3),
                   "note": "from gr_program_manager_create to the first complete 4K frame; code objects from the on-disk cache (a cold build: program_build_s)"}
        if not args.no_build_timing and not args.no_secondary:
            startup["program_build_s"] = program_build_seconds(args.metric, args.spin, args.redshift)
        extra["startup"] = startup
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # (two seconds of the headline frame, untimed, right before the host takes over for the CPU baseline: the driver samples the
        # device's activity from outside, and bursts of 0.1 s between long host phases are easy to miss)
        t_busy = time.perf_counter()
        while time.perf_counter() - t_busy < 2.0:
            for _ in range(8):
                frame()
            torch.cuda.synchronize()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        feats_kw = dict(adaptive_sampling=0, max_acceleration_change=metric.info.max_acceleration_change)
        cpu = cpu_baseline(args.metric, cfg_values, feats_kw, args.cpu_seconds, prefer=args.cpu_baseline)

    if rank == 0:
        line = {
            "metric": f"Mrays/sec at {W}x{H} {'Kerr' if args.metric == 'kerr_boyer' else args.metric} (whole frame: prepass + init + adaptive Verlet + render-data + anisotropic render)",
            "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.metric}{f' (Boyer-Lindquist rs=1 a={args.spin})' if args.metric == 'kerr_boyer' else ''} {W}x{H}, camera (0,0,-4,0) fov 90, adaptive_sampling off, "
                                   f"prepass {'on' if metric.info.use_prepass else 'off'}, tol {metric.info.max_acceleration_change:g}, "
                                   f"background 4096x2048 RGBA8 10 mips, anisotropy 8",
                       "mode": args.mode, "prepass_lookahead": bool(fused and not args.no_lookahead), "prepass_lookahead_depth": depth,
                       "frames_in_flight": in_flight, "trace_waves_per_simd": waves_per_launch, "prepass": "look-ahead on side streams" if (fused and not args.no_lookahead) else "inside the trace launch (gr_frame_options.inline_prepass)" if fused else "own launches", "priming_frames": priming, "build_key": program.build_key, "counters_tag": tag, "program": "substituted (parameters baked in, metric_manager.hpp:153-166)" if args.program == "static" else "dynamic",
                       "parallelism": f"{plan.block_rows}-row blocks, block-cyclic over {world} GPUs (assignment rotating per frame); {gather_path}" if multi else "single GPU"},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        line.update(extra)
        if one_device:
            line["rehearsal"] = f"GR_BENCH_ONE_DEVICE=1: {world} ranks on ONE GPU through the inter-process transport - a run of the N > 1 code path, not a measurement"
        if rccl_rehearsal:
            line["rehearsal"] = (f"GR_BENCH_ONE_DEVICE=rccl: {world} ranks on ONE GPU, process group and frame exchange through RCCL (every rank claims a host "
                                 "of its own, RCCL's socket transport over loopback) - the code of the N-GPU run, not a measurement")
    if multi and rccl_log:
        # channels by transport, over all ranks: " ... Channel 02/1 : 3[3] -> 0[0] via P2P/IPC" -> {"P2P/IPC": n, ...}
        mine = {}
        try:
            ctypes.CDLL(None).fflush(None)
            for text in open(rccl_log, errors="replace"):
                if "Channel" in text and " via " in text:
                    kind = text.split(" via ", 1)[1].split()[0]
                    mine[kind] = mine.get(kind, 0) + 1
        except OSError as e:
            import glob
            print(f"[bench] rank {rank}: no RCCL log to read ({e}); NCCL_DEBUG={os.environ.get('NCCL_DEBUG')} NCCL_DEBUG_FILE={os.environ.get('NCCL_DEBUG_FILE')} "
                  f"in {os.path.dirname(rccl_log)}: {glob.glob(os.path.join(os.path.dirname(rccl_log), 'gr_bench_rccl*'))}", file=sys.stderr)
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        if rank == 0:
            total = {}
            for d in everyone:
                for k, v in (d or {}).items():
                    total[k] = total.get(k, 0) + v
            line["rccl"] = {"ranks": world, "channels_by_transport": total or None, "exchange": gather_path,
                            "note": "P2P/IPC between two GPUs of one node is xGMI; NET/Socket means the ranks were taken for separate hosts (the one-GPU rehearsal)"}
            if without_transfer_s:
                line["rccl"]["per_rank_ms_per_frame_without_transfer"] = [round(x, 4) for x in per_rank_without]
    if multi:
        if tiled is not None:
            tiled.close()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner through C stdio when a communicator comes up: get it out before the one JSON line,
        # which is the last thing on stdout
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:   # noqa: BLE001
            pass
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
