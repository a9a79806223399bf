#!/bin/bash
# GPU box: the three benched workloads, headline figures only (no secondary figures, no CPU baseline).  tools/quick_bench.sh <tag>
tag=${1:-quick}
mkdir -p gpurun_out
for cfg in "" "--config 3" "--config 4"; do
  python bench.py --no-secondary --no-cpu-baseline $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']; v=d.get('valu_roofline',{})
print('$cfg'.strip() or 'kerr', d['value'], d['unit'], 'ms/frame', d['ms_per_step'], 'launch alone ms', r.get('avg_launch_ms'), 'one at a time', r.get('frame_one_at_a_time_ms'), 'trace kernel', d.get('trace_kernel', r.get('trace_kernel')))
" | tee -a gpurun_out/quick_$tag.txt
done
