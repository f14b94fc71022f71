"""Per-kernel summary (calls, total / average / min / max duration) of a rocprofv3 rocpd SQLite database, written as the
CSV that `rocprofv3 --stats` prints (the summaries under profiles/ are made with this).
    python tools/rocpd_stats.py gpurun_out/prof/x_results.db profiles/rNN_name_kernel_stats.csv [calls_per_step]"""
import csv
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
per = float(sys.argv[3]) if len(sys.argv) > 3 else None
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
rows = c.execute(f"select {name}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
                 f"group by {name} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
with open(out, "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], round(r[3], 1), round(100.0 * r[2] / tot, 3), r[4], r[5]])
print(f"{len(rows)} kernels, {tot / 1e6:.3f} ms total")
for r in rows[:40]:
    extra = f"  ms/step={r[2] / per / 1e6:7.3f}" if per else ""
    print(f"{r[0][:78]:78s} n={r[1]:5d} avg_us={r[3] / 1e3:9.1f}{extra}")
