#!/bin/bash
# usage (GPU box, repo root): tools/diet_probe.sh <tag> [metric]   -> gpurun_out/<tag>/{time.json,pmc_*,summary.txt}
TAG=$1; M=${2:-kerr_boyer}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export PYTHONPATH=$GRAFT_REPO_ROOT
python $GRAFT_REPO_ROOT/tools/diet_probe.py $M 7 > $OUT/time.json 2> $OUT/time.err
cat $OUT/time.json
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAVES" \
         "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc$i -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/tools/diet_probe.py $M 3 > $OUT/pmc$i.log 2>&1
done
python - $OUT <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
att = json.load(open(out + "/time.json"))["attempts"]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("gr_trace_fused"):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: sum(v) / len(v) for k, v in agg.items()}
waves_attempts = att / 64.0   # lane attempts -> the counters count wave instructions; with ~97 % lane use this over-counts slightly
lines = [f"attempts per launch {att}"]
for k in sorted(c):
    lines.append(f"{k:28s} {c[k]:14.5g}   per 64 attempts {c[k] / waves_attempts:8.2f}")
if "SQ_INSTS_VALU_FMA_F32" in c and "SQ_INSTS_VALU" in c:
    lines.append(f"FMA share of VALU {c['SQ_INSTS_VALU_FMA_F32'] / c['SQ_INSTS_VALU']:.3f}")
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
