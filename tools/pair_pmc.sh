#!/bin/bash
# counters of gr_trace_fused (rays per lane 1) against gr_trace_pair (2) on the bench frame; run on the GPU box from the repo root
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for R in 1 2; do
  i=0
  for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_INSTS_BRANCH" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INSTS_VALU_TRANS_F32"; do
    i=$((i+1))
    GR_TRACE_RAYS_PER_LANE=$R timeout 240 rocprofv3 --kernel-trace --pmc $C -d $OUT/pair_r${R}_pmc$i -o pmc --output-format csv -- \
        python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --frames-in-flight 1 --no-lookahead > $OUT/pair_r${R}_pmc$i.log 2>&1
    echo "rays/lane $R pass $i: rc=$?"
  done
done
