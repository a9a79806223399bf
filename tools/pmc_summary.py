"""Summarises rocprofv3 --pmc CSV passes (gpurun_out/<dir>/pmc_counter_collection.csv) into a text table and the
profiles/pmc_trace_kernel.json that bench.py reads for roofline.traffic."""
import collections, csv, json, os, sys

def load(dirs):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        f = os.path.join(d, "pmc_counter_collection.csv")
        if not os.path.exists(f):
            continue
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if k.startswith("gr_"):
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {n: sum(v) / len(v) for n, v in c.items()} for k, c in agg.items()}

if __name__ == "__main__":
    # usage: pmc_summary.py <out.txt> <out.json> <log of one of the profiled bench runs> <pmc dir> ...
    out_txt, out_json, bench_log = sys.argv[1], sys.argv[2], sys.argv[3]
    bench_line = json.loads([ln for ln in open(bench_log).read().splitlines() if ln.startswith("{")][-1])
    build_key, workload = bench_line["config"]["build_key"], bench_line["config"]["workload"]
    import re
    size = re.search(r"(\d+)x(\d+)", workload)
    pixels = int(size.group(1)) * int(size.group(2)) if size else 0
    data = load(sys.argv[4:])
    lines = ["# rocprofv3 --kernel-trace --pmc <set> -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --frames-in-flight 1 --no-lookahead   (tools/final_profiles.sh; one pass per counter set)",
             f"# {workload}; build {build_key}; per-dispatch means; FETCH_SIZE / WRITE_SIZE in KiB"]
    for k in sorted(data):
        for n in sorted(data[k]):
            lines.append(f"{k:22s} {n:26s} {data[k][n]:16.6g}")
    # the dominant kernel: the fused trace, or - a profile of the reference-shaped sequence (bench.py --mode reference) - its Verlet kernel,
    # whose algorithmic traffic is SURVEY.md 8d's 140 B per ray
    main_kernel, bytes_per_ray = "gr_trace_fused", 32
    if "gr_trace_fused" not in data:
        candidates = [k for k in data if k.startswith("gr_do_generic_rays") and "SQ_INSTS_VALU" in data[k]]
        if candidates:
            main_kernel, bytes_per_ray = max(candidates, key=lambda k: data[k]["SQ_INSTS_VALU"]), 140
    t = data.get(main_kernel, {})
    if t:
        lane_util = t["SQ_THREAD_CYCLES_VALU"] / (t["SQ_ACTIVE_INST_VALU"] * 64)
        fetch, write = t.get("FETCH_SIZE", 0) * 1024, t.get("WRITE_SIZE", 0) * 1024
        hbm = 2 * fetch + write       # gfx950: FETCH_SIZE reports half the bytes of a wide coalesced read (MI355X_MICROARCH.md, HBM)
        lines += ["", f"# derived, {main_kernel}:",
                  f"#   VALU lane utilisation = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU * 64) = {lane_util:.3f}",
                  f"#   SQ_INSTS_VALU = {t['SQ_INSTS_VALU']:.4g} wave-instructions per launch",
                  f"#   wave time: WAIT_INST_ANY {t['SQ_WAIT_INST_ANY'] / t['SQ_WAVE_CYCLES']:.2f}, WAIT_ANY {t['SQ_WAIT_ANY'] / t['SQ_WAVE_CYCLES']:.2f} of SQ_WAVE_CYCLES",
                  f"#   HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE = 2 x {fetch:.4g} + {write:.4g} = {hbm:.4g} B  (algorithmic: {bytes_per_ray} B x {pixels} = {bytes_per_ray * pixels:.4g} B)"]
        if "SQ_INSTS_VALU_FMA_F32" in t:
            flops = 64 * (t["SQ_INSTS_VALU_ADD_F32"] + t["SQ_INSTS_VALU_MUL_F32"] + 2 * t["SQ_INSTS_VALU_FMA_F32"] + t["SQ_INSTS_VALU_TRANS_F32"])
            other = t["SQ_INSTS_VALU"] - (t["SQ_INSTS_VALU_ADD_F32"] + t["SQ_INSTS_VALU_MUL_F32"] + t["SQ_INSTS_VALU_FMA_F32"] +
                                          t["SQ_INSTS_VALU_TRANS_F32"] + t["SQ_INSTS_VALU_INT32"] + t["SQ_INSTS_VALU_CVT"])
            lines += [f"#   fp32 FLOP per launch from the SQ_INSTS_VALU_{{ADD,MUL,FMA x2,TRANS}}_F32 counters x 64 lanes = {flops:.4g}",
                      f"#   VALU mix per launch: FMA {t['SQ_INSTS_VALU_FMA_F32']:.4g}, MUL {t['SQ_INSTS_VALU_MUL_F32']:.4g}, ADD {t['SQ_INSTS_VALU_ADD_F32']:.4g}, "
                      f"TRANS {t['SQ_INSTS_VALU_TRANS_F32']:.4g}, INT32 {t['SQ_INSTS_VALU_INT32']:.4g}, CVT {t['SQ_INSTS_VALU_CVT']:.4g}, "
                      f"other (mov/cmp/cndmask/min/max/bit ops) {other:.4g}",
                      f"#   instruction issue: SALU {t['SQ_INSTS_SALU']:.4g}, branches {t.get('SQ_INSTS_BRANCH', 0):.4g}, waves {t['SQ_WAVES']:.6g}"]
        flop_counters = None
        if "SQ_INSTS_VALU_FMA_F32" in t:
            flop_counters = round(64 * (t["SQ_INSTS_VALU_ADD_F32"] + t["SQ_INSTS_VALU_MUL_F32"] + 2 * t["SQ_INSTS_VALU_FMA_F32"] + t["SQ_INSTS_VALU_TRANS_F32"]))
        json.dump({"kernel": main_kernel, "workload": workload, "build_key": build_key,
                   "fp32_flop_per_launch": flop_counters, "valu_wave_instructions_per_launch": round(t["SQ_INSTS_VALU"]), "hbm_bytes_per_launch": round(hbm), "fetch_size_bytes": round(fetch), "write_size_bytes": round(write),
                   "fetch_correction": "x2 (gfx950 FETCH_SIZE counts 128-B requests at 64 B)", "valu_lane_utilisation": round(lane_util, 4),
                   "source": "profiles/" + os.path.basename(out_txt)}, open(out_json, "w"), indent=1)
    open(out_txt, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[-6:]))
