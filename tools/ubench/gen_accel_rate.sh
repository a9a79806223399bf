#!/bin/bash
# builds tools/ubench/accel_rate: the substituted Kerr acceleration + a Verlet-like update, one ray per lane (fp32) against two
# rays per lane (packed fp32), at a chosen occupancy (argument: dynamic LDS bytes per workgroup).  Run from the repository root.
set -e
cd "$(dirname "$0")"
python - <<'PY'
import os, shlex, sys
sys.path.insert(0, os.path.abspath("../.."))
import geodesic_raytracing_amd as gra
m = gra.Metric("kerr_boyer", os.path.abspath("../../geodesic_raytracing_amd/scripts"))
s = m.argument_string(m.features(adaptive_sampling=0), True, m.cfg_values(a=0.45))
with open("macros.h", "w") as f:
    for tok in shlex.split(s):
        if tok.startswith(("-DGEO_ACCEL", "-DTEMPORARIES0")):
            k, _, v = tok[2:].partition("=")
            f.write(f"#define {k} {v}\n")
PY
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -fno-math-errno -freciprocal-math -fassociative-math -fno-signed-zeros \
      -fno-trapping-math -fno-hip-fp32-correctly-rounded-divide-sqrt -fapprox-func -fgpu-flush-denormals-to-zero -fno-slp-vectorize -w \
      -DITERS_SCALE=2 accel_rate.hip -o accel_rate
