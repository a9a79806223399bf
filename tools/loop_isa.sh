#!/bin/bash
# disassembles the cached code object of a script metric; prints branch structure of gr_do_generic_rays
# usage: tools/loop_isa.sh <metric> [static]
M=${1:-kerr_boyer}; ST=${2:-0}
rm -rf /tmp/kcache; GR_CACHE_DIR=/tmp/kcache python -c "
import sys; sys.path.insert(0,'/root/repo')
import geodesic_raytracing_amd as gra
m = gra.Metric('$M', '/root/repo/geodesic_raytracing_amd/scripts')
cfg = m.cfg_values(a=0.45) if '$M' == 'kerr_boyer' else m.cfg_values()
s = m.argument_string(features=m.features(adaptive_sampling=0), static=bool($ST), cfg_values=cfg)
gra.Program.precompile(s)
"
/opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn /tmp/kcache/*.hsaco > /tmp/k.s 2>/dev/null
a=$(grep -n "<gr_do_generic_rays>:" /tmp/k.s | cut -d: -f1); b=$(grep -n "<gr_calculate_singularities>:" /tmp/k.s | cut -d: -f1)
sed -n "${a},${b}p" /tmp/k.s | sed 's#//.*##' > /tmp/dgr.s
echo "total lines $(wc -l < /tmp/dgr.s)"
grep -n "s_cbranch\|s_branch" /tmp/dgr.s | awk '{print $1, $2, $3}'
