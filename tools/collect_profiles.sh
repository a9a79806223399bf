#!/bin/bash
# gpurun_out/<tag>_* (tools/final_profiles.sh) -> the summaries under profiles/ that DESIGN.md and bench.py cite
# usage (container, repository root): tools/collect_profiles.sh <tag>
TAG=${1:-r02}
for W in a045 a09; do
  for K in stats exclusive_stats; do
    DB=$(find gpurun_out/${TAG}_${W}_${K} -name "*.db" | head -1)
    [ -n "$DB" ] && { echo "# rocprofv3 --kernel-trace --stats -- python bench.py $( [ $W = a09 ] && echo --spin 0.9 || echo --spin 0.45 ) --steps 20 --warmup 3 --no-cpu-baseline --no-secondary$( [ $K = exclusive_stats ] && echo ' --frames-in-flight 1 --no-lookahead' )"; python tools/rocprof_summary.py $DB | tail -n +2; echo "# bench line of the profiled run:"; grep '^{"metric"' gpurun_out/${TAG}_${W}_${K}.log | tail -1 | cut -c1-2000; } > profiles/${TAG}_kernel_${K}_${W}.txt
  done
  python tools/pmc_summary.py profiles/${TAG}_pmc_${W}.txt profiles/pmc_kerr_${W}_4k.json gpurun_out/${TAG}_${W}_pmc1.log gpurun_out/${TAG}_${W}_pmc[1-5]
done
