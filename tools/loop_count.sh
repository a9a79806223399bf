#!/bin/bash
# usage: isa_count.sh <tag>   (env GR_ACCEL_FORM, GR_EXTRA_FLAGS honoured) -> counts VALU in trace_fused main loop
TAG=$1
rm -rf /tmp/kc_$TAG; mkdir -p /tmp/kc_$TAG
GR_CACHE_DIR=/tmp/kc_$TAG python - <<'PY'
import sys; sys.path.insert(0,'/root/repo')
import geodesic_raytracing_amd as gra
m = gra.Metric('kerr_boyer', '/root/repo/geodesic_raytracing_amd/scripts')
s = m.argument_string(features=m.features(adaptive_sampling=0), static=True, cfg_values=m.cfg_values(a=0.45))
gra.Program.precompile(s)
print("accel_ops", m.info.accel_ops)
PY
/opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn /tmp/kc_$TAG/*.hsaco > /tmp/kc_$TAG/k.s 2>/dev/null
python3 - $TAG <<'PY'
import re,sys
tag=sys.argv[1]
lines=open(f'/tmp/kc_{tag}/k.s').read().splitlines()
# function ranges
start=[i for i,l in enumerate(lines) if l.endswith('<gr_trace_fused>:')][0]
end=[i for i,l in enumerate(lines) if i>start and re.match(r'^[0-9a-f]+ <',l)][0]
body=lines[start:end]
# find the hot loop: the backward branch spanning the v_rsq
ins=[(i,l.split('//')[0].strip()) for i,l in enumerate(body) if l.startswith('\t')]
addr={}
for i,l in enumerate(body):
    m=re.search(r'//\s*([0-9A-Fa-f]+):',l)
    if m: addr[int(m.group(1),16)]=i
# compute loops via branch targets: s_cbranch X <label+off>
best=None
for i,l in enumerate(body):
    m=re.match(r'\s*s_c?branch\S*\s+(\d+)\s*//\s*([0-9A-Fa-f]+):',l)
    if not m: continue
    off=int(m.group(1)); pc=int(m.group(2),16)
    if off>=32768: off-=65536
    tgt=pc+4+off*4
    if tgt<pc and tgt in addr:
        j=addr[tgt]
        seg=body[j:i+1]
        if any('v_rsq_f32' in x or 'v_rcp_f32' in x for x in seg):
            nv=sum(1 for x in seg if x.strip().startswith('v_'))
            if 100<=nv<400 and (best is None or len(seg)>len(best)): best=seg
valu=[x.split()[0] for x in best if x.strip().startswith('v_')]
salu=[x.split()[0] for x in best if x.strip().startswith('s_')]
from collections import Counter
c=Counter()
for v in valu:
    k='fma' if re.match(r'v_(fma|fmac|fmamk|fmaak)_f32',v) else 'mul' if v.startswith('v_mul_f32') else 'add' if re.match(r'v_(add|sub|subrev)_f32',v) else 'trans' if re.match(r'v_(rcp|rsq|sqrt|sin|cos|exp|log)',v) else 'cmp' if v.startswith('v_cmp') else 'cndmask' if v.startswith('v_cndmask') else 'mov' if v.startswith('v_mov') else 'minmax' if re.match(r'v_(min|max|med3)',v) else 'other'
    c[k]+=1
print(tag,'loop lines',len(best),'VALU',len(valu),'SALU',len(salu),dict(c))
PY
