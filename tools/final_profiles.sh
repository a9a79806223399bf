#!/bin/bash
# Round-end evidence, run on the GPU box from the repository root:
#   1. rocprofv3 --kernel-trace --stats of the default bench command (secondary figures and CPU baseline switched off so that
#      every launch in the trace belongs to the headline workload)
#   2. PMC passes of the same workload, one counter set per pass (FETCH_SIZE and WRITE_SIZE each on their own)
# usage: tools/final_profiles.sh <tag>        -> gpurun_out/<tag>_*
TAG=${1:-r01_final}
OUT=$GRAFT_REPO_ROOT/gpurun_out
ARGS="--steps 20 --warmup 3 --no-cpu-baseline --no-secondary"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats -o bench -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/${TAG}_stats.log 2>&1
tail -1 $OUT/${TAG}_stats.log | cut -c1-200
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU" \
         "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_INSTS_BRANCH" \
         "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    # counters serialise kernels: frames in flight 1 keeps the launches comparable with the sequential trace time
    timeout 240 rocprofv3 --kernel-trace --pmc $C -d $OUT/${TAG}_pmc$i -o pmc --output-format csv -- \
        python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --frames-in-flight 1 --no-lookahead > $OUT/${TAG}_pmc$i.log 2>&1
    echo "pass $i ($C): rc=$? $(ls $OUT/${TAG}_pmc$i 2>/dev/null | tr '\n' ' ')"
done
