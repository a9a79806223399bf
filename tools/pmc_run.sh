#!/bin/bash
# usage: tools/pmc_run.sh <outdir-name> "<counters>" [extra bench args]
# collects PMC counters (own pass, no tracing domains besides kernel-trace) for a short bench run
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
CNT="$1"; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $CNT -d $OUT -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT.log 2>&1
ls $OUT
