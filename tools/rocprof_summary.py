"""rocprofv3 (rocpd .db output of `--kernel-trace --stats`) -> per-kernel summary table (text)."""
import glob
import sqlite3
import sys


def summarize(db_path):
    con = sqlite3.connect(db_path)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = (f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
         f"max(s.arch_vgpr_count), max(s.sgpr_count), max(d.private_segment_size), max(d.grid_size_x), max(d.workgroup_size_x) "
         f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc")
    rows = list(con.execute(q))
    total = sum(r[2] for r in rows) or 1
    lines = [f"{'kernel':58s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>11s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s} {'vgpr':>5s} {'sgpr':>5s} {'scratch':>7s} {'grid_x':>9s} {'wg':>4s}"]
    for r in rows:
        lines.append(f"{r[0][:58]:58s} {r[1]:6d} {r[2] / 1e6:10.3f} {r[3] / 1e3:11.2f} {r[4] / 1e3:10.2f} {r[5] / 1e3:10.2f} "
                     f"{100 * r[2] / total:6.2f} {r[6]:5d} {r[7]:5d} {r[8]:7d} {r[9]:9d} {r[10]:4d}")
    return "\n".join(lines)


if __name__ == "__main__":
    paths = sys.argv[1:] or glob.glob("gpurun_out/**/*.db", recursive=True)
    for p in paths:
        print("#", p)
        print(summarize(p))
