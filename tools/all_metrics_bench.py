"""Every script of the reference's scripts/ folder, timed the way the reference publishes its one performance claim (README.md:5: "1080 at
30fps+" for "the vast majority" of metrics; `-bench <metric>`, main.cpp:970-986, 2864-2871): 1920x1080, the GUI's defaults (camera
(0,0,-4,0), fov 90, adaptive sampling ON at threshold 32: graphics_settings.hpp:39-40) - and with every pixel traced - through the
dynamic program (what runs for the seconds after a slider moved) and the substituted one (the steady state).  Per script: frames per
second one frame at a time, Verlet attempts per ray, registers and SCRATCH BYTES of the trace kernel, fp32 fraction of the trace launch.

  python tools/all_metrics_bench.py manifest     build container: the reference's 31 scripts, unmodified, through this repository's front-end and
                                                 generator -> tools/_manifests/reference_scripts.json (generated macro strings + the frame driver's
                                                 settings; the GPU box has no /root/reference).  Git-ignored, travels with gpurun.
  python tools/all_metrics_bench.py precompile   build container: every program of the manifest into the code-object cache (one compiler per core)
  PYTHONPATH=. python tools/all_metrics_bench.py run [out.txt] [names...]    GPU box: the table
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# as bench.py: every timed frame computes its own camera set-up and prepass although the camera never moves (the library's default for a
# repeated frame, reuse_still_camera, would skip both), and nothing guesses the next camera (GR_GUESS_STILL_CAMERA=1: as the round's first table)
os.environ.setdefault("GR_REUSE_STILL_CAMERA", "0")
MANIFEST = os.path.join(ROOT, "tools", "_manifests", "reference_scripts.json")
REFERENCE_SCRIPTS = "/root/reference/scripts"
W, H = 1920, 1080
VALU_PEAK = 157.3e12
STEP_OVERHEAD_FLOPS = 90


def make_manifest():
    import glob
    import geodesic_raytracing_amd as gra
    out = {}
    for path in sorted(glob.glob(os.path.join(REFERENCE_SCRIPTS, "*.js"))):
        name = os.path.basename(path)[:-3]
        m = gra.Metric(name, REFERENCE_SCRIPTS)
        cfg = m.cfg_values()
        every = m.features(adaptive_sampling=0)
        gui = m.features(adaptive_sampling=1, adaptive_sampling_threshold=32.0)
        out[name] = {"info": {f: getattr(m.info, f) for f, _ in m.info._fields_}, "dynamic_vars": m.dynamic_vars, "dynamic_defaults": m.dynamic_defaults,
                     "dynamic": m.argument_string(),
                     "substituted_every_pixel": m.argument_string(features=every, static=True, cfg_values=cfg),
                     "substituted_adaptive": m.argument_string(features=gui, static=True, cfg_values=cfg),
                     "substituted_ops": list(m.substituted_op_counts(cfg))}
        print(f"{name:36s} accel ops {m.info.accel_ops:5d}  strings {len(out[name]['dynamic']) // 1024:4d} + {len(out[name]['substituted_every_pixel']) // 1024:4d} KiB", flush=True)
    os.makedirs(os.path.dirname(MANIFEST), exist_ok=True)
    with open(MANIFEST, "w") as f:
        json.dump(out, f)
    print(f"{len(out)} scripts -> {MANIFEST} ({os.path.getsize(MANIFEST) // 1024} KiB)")


def _compile(text):
    import geodesic_raytracing_amd as gra
    t = time.perf_counter()
    gra.Program.precompile(text)
    return time.perf_counter() - t


def precompile(names):
    import multiprocessing
    man = json.load(open(MANIFEST))
    jobs = []
    for name, e in man.items():
        if names and name not in names:
            continue
        for k in ("dynamic", "substituted_every_pixel", "substituted_adaptive"):
            if e[k] not in jobs:
                jobs.append(e[k])
    with multiprocessing.get_context("spawn").Pool(max(1, (os.cpu_count() or 2) - 1)) as pool:
        secs = pool.map(_compile, jobs, chunksize=1)
    print(f"{len(jobs)} programs, {sum(secs):.0f} compiler-seconds, longest {max(secs):.1f} s")


def run(out_path, names):
    import numpy as np
    import torch
    import geodesic_raytracing_amd as gra
    man = json.load(open(MANIFEST))
    bg_np, levels = gra.pack_background(gra.synthetic_background(4096, 2048))
    bg = torch.from_numpy(bg_np).cuda()
    out = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    camera = gra.default_camera()
    rows = []
    head = (f"# tools/all_metrics_bench.py run: the reference's 31 scripts at {W}x{H}, default camera (0,0,-4,0) fov 90, one frame at a time (3 warm-up + 12 timed frames, median of three\n"
            f"# repeats), fused path, prepass per the script's JSON.  gui = adaptive sampling on, threshold 32 (the reference's default); all = every pixel traced.\n"
            f"# dyn = dynamic program, sub = substituted.  att/ray = Verlet attempts per pixel (all, sub).  kernel = the trace kernel that ran (f = gr_trace_fused, p = gr_trace_pair:\n"
            f"# fixed-step programs, two rays per lane), its VGPRs and scratch bytes per lane (dyn | sub).  valu = (generator's op count + {STEP_OVERHEAD_FLOPS}) x attempts / trace launch / 157.3 TF (all, sub).\n"
            f"# lit = fraction of pixels whose ray reached the sky (all, sub).  x3 = three frames in flight on three render states with the next camera announced\n"
            f"# (the reference's main loop cycles a ring of render states: main.cpp:1463-1469; how bench.py times its headline).  GR_OCCUPANCY_TUNING={os.environ.get('GR_OCCUPANCY_TUNING', '1')}\n"
            f"{'script':34s} {'fps gui dyn':>11s} {'fps gui sub':>11s} {'fps all dyn':>11s} {'fps all sub':>11s} {'gui sub x3':>10s} {'all sub x3':>10s} {'att/ray':>8s} {'ops':>5s} {'kernel dyn | sub':>24s} {'trace ms':>9s} {'valu':>6s} {'lit':>5s}")
    print(head, flush=True)

    def fps_of(prog, m, feats, cfg):
        st = gra.RenderState(W, H, 0)
        o = gra.frame_options(mode=gra.MODE_FUSED)

        def once():
            st.render(prog, m, camera, out.data_ptr(), (bg.data_ptr(), 4096, 2048, levels), feats, cfg, o, stream)
        reps = []
        for _ in range(3):
            for _ in range(3):
                once()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(12):
                once()
            torch.cuda.synchronize()
            reps.append((time.perf_counter() - t) / 12)
        return 1.0 / float(np.median(reps)), st

    def pipelined_fps(prog, m, feats, cfg, in_flight=3):
        """frames per second the way the reference's main loop produces them (a ring of render states, each frame on the next one:
        main.cpp:1463-1469, 1505-1510) and bench.py's headline is timed: three states / streams, the next frame's camera announced"""
        import ctypes
        states = [gra.RenderState(W, H, 0) for _ in range(in_flight)]
        streams = [torch.cuda.Stream() for _ in range(in_flight)]
        outs = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(in_flight)]
        o = gra.frame_options(mode=gra.MODE_FUSED)
        o.next_camera = ctypes.pointer(camera)
        o.trace_waves_per_simd = 4
        k = [0]

        def once():
            j = k[0] % in_flight
            k[0] += 1
            states[j].render(prog, m, camera, outs[j].data_ptr(), (bg.data_ptr(), 4096, 2048, levels), feats, cfg, o, streams[j].cuda_stream)
        reps = []
        for _ in range(3):
            for _ in range(in_flight + 1):
                once()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(12):
                once()
            torch.cuda.synchronize()
            reps.append((time.perf_counter() - t) / 12)
        return 1.0 / float(np.median(reps))

    def kernel_of(prog):
        pair = prog.has_trace_pair
        k = prog.kernel_info("gr_trace_pair" if pair else "gr_trace_fused")
        return f"{'p' if pair else 'f'} v{k['vgprs']} s{k['scratch_bytes']}"

    for name, e in man.items():
        if names and name not in names:
            continue
        try:
            info = dict(e["info"])
            cfg = e["dynamic_defaults"]
            progs = {}
            for k in ("dynamic", "substituted_every_pixel", "substituted_adaptive"):
                progs[k] = gra.Program(e[k], 0)
            fp = {}
            for label, key, static, adaptive in (("gui_dyn", "dynamic", False, 1), ("gui_sub", "substituted_adaptive", True, 1),
                                                  ("all_dyn", "dynamic", False, 0), ("all_sub", "substituted_every_pixel", True, 0)):
                m = gra.Metric.from_info(name, info, e["dynamic_vars"], cfg, {False: e["dynamic"], True: e[key]})
                feats = m.features(adaptive_sampling=adaptive, adaptive_sampling_threshold=32.0)
                fp[label], st = fps_of(progs[key], m, feats, cfg)
                if label in ("gui_sub", "all_sub"):
                    fp[label + "_pipe"] = pipelined_fps(progs[key], m, feats, cfg)
                if label == "all_sub":
                    o = gra.frame_options(mode=gra.MODE_FUSED, time_kernels=1, count_attempts=1, inline_prepass=0)
                    tr = []
                    for _ in range(4):
                        st.render(progs[key], m, camera, out.data_ptr(), (bg.data_ptr(), 4096, 2048, levels), feats, cfg, o, stream)
                        torch.cuda.synchronize()
                        tr.append(st.stage_ms()["trace"])
                    trace_ms = float(np.median(tr[1:]))
                    attempts = st.attempts()
                    rd = np.empty(W * H, dtype=gra.pipeline.RENDER_DATA_DTYPE)
                    gra.check(gra.lib.gr_device_download(0, rd.ctypes.data_as(__import__("ctypes").c_void_p), st.buffer(gra.BUF_RENDER_DATA), rd.nbytes))
                    lit = float((rd["terminated"] == 1).mean())
                del st
            ops = e["substituted_ops"][0] + e["substituted_ops"][2]
            valu = (ops + STEP_OVERHEAD_FLOPS) * attempts / (trace_ms * 1e-3) / VALU_PEAK
            row = dict(name=name, **fp, attempts_per_ray=attempts / (W * H), ops=ops, kernel_dyn=kernel_of(progs["dynamic"]),
                       kernel_sub=kernel_of(progs["substituted_every_pixel"]), trace_ms=trace_ms, valu=valu, lit=lit)
            rows.append(row)
            print(f"{name:34s} {fp['gui_dyn']:11.1f} {fp['gui_sub']:11.1f} {fp['all_dyn']:11.1f} {fp['all_sub']:11.1f} {fp['gui_sub_pipe']:10.1f} {fp['all_sub_pipe']:10.1f} {row['attempts_per_ray']:8.1f} {ops:5d} "
                  f"{row['kernel_dyn'] + ' | ' + row['kernel_sub']:>24s} {trace_ms:9.3f} {valu:6.3f} {lit:5.2f}", flush=True)
            del progs
        except Exception as ex:   # noqa: BLE001
            print(f"{name:34s} FAILED: {ex}", flush=True)
            rows.append(dict(name=name, error=str(ex)))
    good = [r for r in rows if "error" not in r]
    if good:
        worst = min(good, key=lambda r: r["gui_dyn"])
        summary = {"scripts": len(rows), "failed": [r["name"] for r in rows if "error" in r], "resolution": [W, H],
                   "gui_defaults_dynamic_program": {"min_fps": round(worst["gui_dyn"], 1), "median_fps": round(float(np.median([r["gui_dyn"] for r in good])), 1), "slowest": worst["name"]},
                   "gui_defaults_substituted_program": {"min_fps": round(min(r["gui_sub"] for r in good), 1), "median_fps": round(float(np.median([r["gui_sub"] for r in good])), 1),
                                                        "slowest": min(good, key=lambda r: r["gui_sub"])["name"]},
                   "every_pixel_substituted_program": {"min_fps": round(min(r["all_sub"] for r in good), 1), "median_fps": round(float(np.median([r["all_sub"] for r in good])), 1),
                                                       "slowest": min(good, key=lambda r: r["all_sub"])["name"]},
                   "pipelined_gui_defaults_substituted_program": {"min_fps": round(min(r["gui_sub_pipe"] for r in good), 1), "median_fps": round(float(np.median([r["gui_sub_pipe"] for r in good])), 1),
                                                                  "slowest": min(good, key=lambda r: r["gui_sub_pipe"])["name"]},
                   "below_30_fps_one_frame_at_a_time": sorted(r["name"] for r in good if min(r["gui_dyn"], r["gui_sub"], r["all_dyn"], r["all_sub"]) < 30.0),
                   "below_30_fps_pipelined": sorted(r["name"] for r in good if min(r["gui_sub_pipe"], r["all_sub_pipe"]) < 30.0)}
        print("# summary " + json.dumps(summary), flush=True)
    if out_path:
        with open(out_path, "w") as f:
            f.write(head + "\n")
            for r in rows:
                if "error" in r:
                    f.write(f"{r['name']:34s} FAILED: {r['error']}\n")
                else:
                    f.write(f"{r['name']:34s} {r['gui_dyn']:11.1f} {r['gui_sub']:11.1f} {r['all_dyn']:11.1f} {r['all_sub']:11.1f} {r['gui_sub_pipe']:10.1f} {r['all_sub_pipe']:10.1f} {r['attempts_per_ray']:8.1f} {r['ops']:5d} "
                            f"{r['kernel_dyn'] + ' | ' + r['kernel_sub']:>24s} {r['trace_ms']:9.3f} {r['valu']:6.3f} {r['lit']:5.2f}\n")
            if good:
                f.write("# summary " + json.dumps(summary) + "\n")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "run"
    if what == "manifest":
        make_manifest()
    elif what == "precompile":
        precompile(sys.argv[2:])
    else:
        out_path = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2].endswith(".txt") else None
        run(out_path, [a for a in sys.argv[2:] if not a.endswith(".txt")])
