"""Drop-in check (build container only): every script of the reference's scripts/ folder -> macro string -> gfx950 code object
(hiprtc, no GPU needed), with the size of the generated acceleration and the register footprint of the fused trace kernel.
usage: python tools/compile_reference_scripts.py [scripts_dir]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GR_CACHE_DIR", tempfile.mkdtemp(prefix="gr_refscripts_"))
import geodesic_raytracing_amd as gra  # noqa: E402

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/scripts"
names = sorted(f[:-5] for f in os.listdir(ref) if f.endswith(".json") and os.path.exists(os.path.join(ref, f[:-5] + ".js")))
print(f"{'script':34s} {'system':>6s} {'big':>3s} {'c.theta':>7s} {'adapt':>5s} {'accel ops':>9s} {'transc.':>7s} {'compile s':>9s}")
failed = []
for n in names:
    try:
        m = gra.Metric(n, ref)
        t = time.time()
        gra.Program.precompile(m.argument_string())
        dt = time.time() - t
        i = m.info
        print(f"{n:34s} {'':>6s} {i.is_big:3d} {i.is_constant_theta:7d} {i.adaptive_precision:5d} {i.accel_ops:9d} {i.accel_transcendentals:7d} {dt:9.1f}",
              flush=True)
    except Exception as e:   # noqa: BLE001
        failed.append(n)
        print(f"{n:34s} FAILED: {str(e)[:200]}", flush=True)
print(f"{len(names) - len(failed)} of {len(names)} scripts compile for gfx950" + (f"; failed: {failed}" if failed else ""))
