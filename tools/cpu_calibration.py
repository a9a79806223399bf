"""Build container: the CPU restatement (oracle/restate.cpp - what the GPU box times as `cpu_baseline`, kind "port") against the reference's own
cl.cl compiled for x86-64 (oracle/_ref) on the same sample - the 480x270 Kerr frame of SURVEY.md 8d(ii), init + Verlet trace of every pixel -
on 1 and on all threads.    python tools/cpu_calibration.py > profiles/r05_cpu_calibration.txt"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import geodesic_raytracing_amd as gra
from oracle import build_ref, build_restate
from oracle.refpipe import OraclePipeline, pack_features

m = gra.Metric("kerr_boyer", os.path.join(ROOT, "geodesic_raytracing_amd", "scripts"))
cfg = m.cfg_values(a=0.45)
feats = pack_features(adaptive_sampling=0, max_acceleration_change=m.info.max_acceleration_change)
w, h = 480, 270
print(f"# tools/cpu_calibration.py: kerr_boyer a=0.45, {w}x{h}, camera (0,0,-4,0) fov 90, stages up to the trace; host: {os.cpu_count()} threads")
rows = {}
for label, so in (("restatement (oracle/restate.cpp, g++ -O2)", build_restate.build(m.argument_string())),
                  ("reference (cl.cl for x86-64, oracle/_ref)", build_ref.build("kerr_boyer_script", m.argument_string()))):
    pipe = OraclePipeline(so)
    for threads in (1, os.cpu_count() or 1):
        t0 = time.perf_counter()
        pipe.frame(w, h, cfg, feats, use_prepass=False, nthreads=threads, stages="trace")
        t = time.perf_counter() - t0
        rows[(label, threads)] = t
        print(f"{label:48s} {threads:3d} thread(s)  {t:8.2f} s  {w * h / t / 1e6:.5f} Mrays/s")
a = [k for k in rows if k[0].startswith("restatement")]
for k in a:
    ref = rows[(("reference (cl.cl for x86-64, oracle/_ref)"), k[1])]
    print(f"# restatement / reference time on {k[1]} thread(s): {rows[k] / ref:.3f}")
