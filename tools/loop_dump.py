"""Prints the Verlet loop of gr_trace_fused as compiled for a program (default: the headline program - substituted Kerr a = 0.45) and
its instruction mix.  Build container, no GPU: python tools/loop_dump.py [metric] [--full]"""
import os, re, subprocess, sys, tempfile
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def hot_loop(name="kerr_boyer", cfg=None, kernel="gr_trace_fused", features=None):
    import geodesic_raytracing_amd as gra
    m = gra.Metric(name, os.path.join(ROOT, "geodesic_raytracing_amd", "scripts"))
    cfgv = m.cfg_values(**(cfg if cfg is not None else (dict(a=0.45) if name == "kerr_boyer" else {})))
    s = m.argument_string(features=m.features(adaptive_sampling=0, **(features or {})), static=True, cfg_values=cfgv)
    d = os.environ.get("GR_CACHE_DIR") or tempfile.mkdtemp(prefix="loopdump")
    os.environ["GR_CACHE_DIR"] = d
    before = set(os.listdir(d))
    gra.check(gra.lib.gr_program_precompile_frame_path(s.encode()))   # the code object of the kernels a fused frame launches (+ the set-up module)
    new = [f for f in os.listdir(d) if f not in before and f.endswith(".hsaco") and not f.endswith(".setup.hsaco")]
    path = os.path.join(d, new[0]) if new else None
    if path is None:   # already cached: find by content
        for f in os.listdir(d):
            if f.endswith(".hsaco") and not f.endswith(".setup.hsaco"):
                path = os.path.join(d, f)
    text = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--no-show-raw-insn", path], capture_output=True, text=True).stdout
    lines = text.splitlines()
    start = [i for i, l in enumerate(lines) if l.endswith("<%s>:" % kernel)][0]
    end = [i for i, l in enumerate(lines) if i > start and re.match(r"^[0-9a-f]+ <", l)][0]
    body = lines[start:end]
    addr = {}
    for i, l in enumerate(body):
        mm = re.search(r"//\s*([0-9A-Fa-f]+):", l)
        if mm:
            addr[int(mm.group(1), 16)] = i
    best = None
    for i, l in enumerate(body):
        mm = re.match(r"\s*s_c?branch\S*\s+(\d+)\s*//\s*([0-9A-Fa-f]+):", l)
        if not mm:
            continue
        off, pc = int(mm.group(1)), int(mm.group(2), 16)
        if off >= 32768:
            off -= 65536
        tgt = pc + 4 + off * 4
        if tgt < pc and tgt in addr:
            seg = body[addr[tgt]:i + 1]
            nv = sum(1 for x in seg if x.strip().startswith("v_"))
            if any("v_rsq_f32" in x for x in seg) and 100 <= nv and (best is None or len(seg) < len(best)):
                best = seg
    return best


def classify(op):
    if re.match(r"v_(fma|fmac|fmamk|fmaak)_f32", op): return "fma"
    if op.startswith("v_mul_f32"): return "mul"
    if re.match(r"v_(add|sub|subrev)_f32", op): return "add"
    if re.match(r"v_pk_", op): return "packed"
    if re.match(r"v_(rcp|rsq|sqrt|sin|cos|exp|log)", op): return "trans"
    if op.startswith("v_cmp"): return "cmp"
    if op.startswith("v_cndmask"): return "cndmask"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"): return "mov"
    if re.match(r"v_(min|max|med3)", op): return "minmax"
    return "other:" + op


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    seg = hot_loop(args[0] if args else "kerr_boyer")
    ops = [x.split("//")[0].strip() for x in seg if x.startswith("\t")]
    valu = [o.split()[0] for o in ops if o.startswith("v_")]
    salu = [o.split()[0] for o in ops if o.startswith("s_")]
    c = Counter(classify(v) for v in valu)
    print("loop instructions", len(ops), "VALU", len(valu), "SALU", len(salu), "(s_nop %d)" % sum(1 for o in salu if o == "s_nop"))
    print(sorted(c.items(), key=lambda t: -t[1]))
    if "--full" in sys.argv:
        for o in ops:
            if not o.startswith("s_nop"):
                print("   ", o)
