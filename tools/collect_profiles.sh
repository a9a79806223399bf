#!/bin/bash
# gpurun_out/<tag>_* (tools/final_profiles.sh) -> the summaries under profiles/ that DESIGN.md and bench.py cite
# usage (container, repository root): tools/collect_profiles.sh <tag> [workloads]
TAG=${1:-r03}
WORKLOADS=${2:-"a045 a09 dk alc"}
for W in $WORKLOADS; do
  case $W in
    a045) SEL="--spin 0.45"; JSON=pmc_kerr_a045_4k.json;;
    a09)  SEL="--spin 0.9"; JSON=pmc_kerr_a09_4k.json;;
    dk)   SEL="--config 3"; JSON=pmc_double_unequal_kerr_4k.json;;
    alc)  SEL="--config 4"; JSON=pmc_alcubierre_8k_redshift.json;;
    refseq) SEL="--mode reference"; JSON=pmc_reference_sequence_4k.json;;
    a045dyn) SEL="--spin 0.45 --program dynamic"; JSON=pmc_kerr_a045_4k_dynamic.json;;
    refseqdyn) SEL="--mode reference --program dynamic"; JSON=pmc_reference_sequence_4k_dynamic.json;;
  esac
  for K in stats exclusive_stats; do
    DB=$(find gpurun_out/${TAG}_${W}_${K} -name "*.db" 2>/dev/null | head -1)
    [ -n "$DB" ] && { echo "# rocprofv3 --kernel-trace --stats -- python bench.py $SEL --steps N --warmup 3 --no-cpu-baseline --no-secondary$( [ $K = exclusive_stats ] && echo ' --frames-in-flight 1 --no-lookahead --inline-prepass 0' )"; python tools/rocprof_summary.py $DB | tail -n +2; echo "# bench line of the profiled run:"; grep '^{"metric"' gpurun_out/${TAG}_${W}_${K}.log | tail -1 | cut -c1-2000; } > profiles/${TAG}_kernel_${K}_${W}.txt
  done
  [ -d gpurun_out/${TAG}_${W}_pmc1 ] && python tools/pmc_summary.py profiles/${TAG}_pmc_${W}.txt profiles/$JSON gpurun_out/${TAG}_${W}_pmc1.log gpurun_out/${TAG}_${W}_pmc[1-5]
done
