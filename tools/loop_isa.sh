#!/bin/bash
# disassembles the cached code object of a builtin metric and prints the instruction histogram of gr_do_generic_rays
M=${1:-kerr_boyer}
rm -rf /tmp/kcache; GR_CACHE_DIR=/tmp/kcache python -c "
import sys; sys.path.insert(0,'/root/repo')
import geodesic_raytracing_amd as gra
gra.Program.precompile(gra.Metric('$M').argument_string())
"
/opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn /tmp/kcache/*.hsaco > /tmp/k.s 2>/dev/null
a=$(grep -n "<gr_do_generic_rays>:" /tmp/k.s | cut -d: -f1); b=$(grep -n "<gr_calculate_singularities>:" /tmp/k.s | cut -d: -f1)
sed -n "${a},${b}p" /tmp/k.s > /tmp/dgr.s
echo "total lines $(wc -l < /tmp/dgr.s)"
grep -n "s_cbranch\|s_branch" /tmp/dgr.s | awk '{print $1, $2, $3, $NF}'
