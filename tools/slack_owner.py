"""Which approximation owns the rays of the GPU trace that miss the 1e-3 position rule (tests/gpu_stages.py assert_traced_positions)?
Runs tools/trace_rule_stats.py on builds that take the approximations away one at a time and sums the per-fixture counts.
  build container:  python tools/slack_owner.py build   (code objects of every variant -> tools/_variants/slack/<variant>)
  GPU box:          python tools/slack_owner.py run     -> table (profiles/r03_trace_slack_owner.txt)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = {
    "default": "",
    "ieee_div_sqrt": "-fhip-fp32-correctly-rounded-divide-sqrt -fno-approx-func -fno-reciprocal-math",
    "no_contraction": "-ffp-contract=off",
    "libm_trig": "-DGR_LIBM_TRIG",
    "no_reassociation": "-fno-associative-math",
    "all_strict": "-fhip-fp32-correctly-rounded-divide-sqrt -fno-approx-func -fno-reciprocal-math -ffp-contract=off -DGR_LIBM_TRIG -fno-associative-math",
}


def env_of(name):
    return dict(os.environ, GR_EXTRA_FLAGS=VARIANTS[name], GR_CACHE_DIR=os.path.join(ROOT, "tools", "_variants", "slack", name))


BUILD = r'''
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import geodesic_raytracing_amd as gra
from gpu_stages import load_golden, golden_names, metric_for
seen = set()
for name in golden_names():
    meta, z = load_golden(name)
    if meta["prepass"] or meta["features"].get("adaptive_sampling"):
        continue
    s = metric_for(meta).argument_string()
    if s not in seen:
        seen.add(s); gra.Program.precompile(s)
''' % (ROOT, ROOT)

if sys.argv[1] == "build":
    procs = []
    for name in VARIANTS:   # one process per variant, side by side
        os.makedirs(env_of(name)["GR_CACHE_DIR"], exist_ok=True)
        procs.append((name, subprocess.Popen([sys.executable, "-c", BUILD], env=env_of(name))))
    for name, proc in procs:
        assert proc.wait() == 0, name
        print("built", name, flush=True)
else:
    print("# ordinary rays (fewer than twice the median attempts, terminated on both sides) whose final position differs from the reference's by > 1e-3,")
    print("# summed over the golden fixtures, per build of the kernels (tools/slack_owner.py; per fixture: tools/trace_rule_stats.py)")
    for name in VARIANTS:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "trace_rule_stats.py")], env=env_of(name), capture_output=True, text=True).stdout
        rows = [re.search(r"^(\S+)\s+flags_differ\s+(\d+) ordinary\s+(\d+)/\s*(\d+) over1e-3\s+(\d+) max (\S+)", l) for l in out.splitlines()]
        rows = [r for r in rows if r]
        over, ordinary, flags = sum(int(r.group(5)) for r in rows), sum(int(r.group(3)) for r in rows), sum(int(r.group(2)) for r in rows)
        worst = max(rows, key=lambda r: int(r.group(5)) / max(int(r.group(3)), 1)) if rows else None
        print(f"{name:18s} flags {VARIANTS[name] or '(product build)'}")
        print(f"{'':18s} fixtures {len(rows)}, ordinary rays {ordinary}, over 1e-3: {over} ({100.0 * over / max(ordinary, 1):.3f} %), termination flags differing: {flags}; "
              f"worst fixture {worst.group(1)}: {worst.group(5)}/{worst.group(3)}" if worst else "no rows", flush=True)
        if "-v" in sys.argv:
            print(out)
