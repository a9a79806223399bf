"""Instruction count and mix of one kernel of a program as compiled (build container, no GPU):
    python tools/kernel_isa.py gr_render [metric] [--full]"""
import os, re, subprocess, sys, tempfile
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def kernel_body(kernel, name="kerr_boyer", features=None):
    import geodesic_raytracing_amd as gra
    m = gra.Metric(name, os.path.join(ROOT, "geodesic_raytracing_amd", "scripts"))
    cfgv = m.cfg_values(**(dict(a=0.45) if name == "kerr_boyer" else {}))
    s = m.argument_string(features=m.features(adaptive_sampling=0, **(features or {})), static=True, cfg_values=cfgv)
    d = tempfile.mkdtemp(prefix="kisa")
    os.environ["GR_CACHE_DIR"] = d
    gra.Program.precompile(s)
    out = []
    for f in os.listdir(d):
        text = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--no-show-raw-insn", os.path.join(d, f)], capture_output=True, text=True).stdout
        lines = text.splitlines()
        starts = [i for i, l in enumerate(lines) if l.endswith("<%s>:" % kernel)]
        if not starts:
            continue
        end = [i for i, l in enumerate(lines) if i > starts[0] and re.match(r"^[0-9a-f]+ <", l)]
        out = [l.split("//")[0].strip() for l in lines[starts[0]:(end[0] if end else len(lines))] if l.startswith("\t")]
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    feats = dict(redshift=1) if "--redshift" in sys.argv else None
    body = kernel_body(args[0], args[1] if len(args) > 1 else "kerr_boyer", feats)
    c = Counter(b.split()[0] for b in body)
    print(args[0], "instructions", len(body), "VALU", sum(v for k, v in c.items() if k.startswith("v_")), "calls", c.get("s_swappc_b64", 0))
    print(c.most_common(30))
    if "--full" in sys.argv:
        print("\n".join(body))
