#!/bin/bash
# Round-end evidence, run on the GPU box from the repository root, for the headline workload (Kerr a = 0.45), BASELINE configs[2]
# read literally (a = 0.9), configs[3] (double_unequal_kerr 4K) and configs[4] (alcubierre 8K, redshift on):
#   1. rocprofv3 --kernel-trace --stats of the bench command (frames in flight, as the number is produced)
#   2. the same with --frames-in-flight 1 --no-lookahead --inline-prepass 0: launches one at a time and the prepass a launch of its
#      own, so a trace launch's duration is the cost of the trace kernel's own work
#   3. PMC passes of the one-at-a-time run, one counter set per pass (FETCH_SIZE and WRITE_SIZE each on their own)
# (secondary figures and CPU baseline switched off so that every launch in a trace belongs to the workload)
# usage: tools/final_profiles.sh <tag> [workloads, default "a045 a09 dk alc"]   -> gpurun_out/<tag>_<workload>_*
#        then, in the container: tools/collect_profiles.sh <tag>
TAG=${1:-r03}
WORKLOADS=${2:-"a045 a09 dk alc"}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for W in $WORKLOADS; do
  case $W in
    a045) SEL="--spin 0.45"; STEPS=20;;
    a09)  SEL="--spin 0.9"; STEPS=20;;
    dk)   SEL="--config 3"; STEPS=10;;
    alc)  SEL="--config 4"; STEPS=10;;
    refseq) SEL="--mode reference"; STEPS=10;;   # the reference-shaped kernel sequence (one launch per reference kernel), 4K Kerr a = 0.45
    a045dyn) SEL="--spin 0.45 --program dynamic"; STEPS=20;;             # the headline frame through the DYNAMIC program (what runs after a slider moved)
    refseqdyn) SEL="--mode reference --program dynamic"; STEPS=10;;      # ... and the reference-shaped sequence through it
    *) echo "unknown workload $W"; continue;;
  esac
  ARGS="$SEL --steps $STEPS --warmup 3 --no-cpu-baseline --no-secondary"
  timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_${W}_stats -o bench -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/${TAG}_${W}_stats.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_${W}_exclusive_stats -o bench -- python $GRAFT_REPO_ROOT/bench.py $ARGS --frames-in-flight 1 --no-lookahead --inline-prepass 0 > $OUT/${TAG}_${W}_exclusive_stats.log 2>&1
  tail -1 $OUT/${TAG}_${W}_stats.log | cut -c1-160
  i=0
  for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_INSTS_BRANCH" \
           "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    # counters serialise kernels: frames in flight 1 keeps the launches comparable with the sequential trace time
    timeout 300 rocprofv3 --kernel-trace --pmc $C -d $OUT/${TAG}_${W}_pmc$i -o pmc --output-format csv -- \
        python $GRAFT_REPO_ROOT/bench.py $SEL --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --frames-in-flight 1 --no-lookahead --inline-prepass 0 > $OUT/${TAG}_${W}_pmc$i.log 2>&1
    echo "$W pass $i ($C): rc=$? $(ls $OUT/${TAG}_${W}_pmc$i 2>/dev/null | tr '\n' ' ')"
  done
done
