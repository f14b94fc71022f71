"""Idle gaps of the kernel queue inside the last full training step of a rocprofv3 --kernel-trace database:
    python tools/step_gaps.py gpurun_out/prof/x_results.db [min_gap_us]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
rows = c.execute("select name,start,end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if 'rollout_fwd' in r[0]]
i0, i1 = idx[-2], idx[-1]
t0 = rows[i0][1]
print("step span ms", (rows[i1][1] - t0) / 1e6, "kernels", i1 - i0, "busy ms", sum(r[2] - r[1] for r in rows[i0:i1]) / 1e6)
tot = 0.0
for a, b in zip(rows[i0:i1], rows[i0 + 1:i1 + 1]):
    gap = (b[1] - a[2]) / 1e3
    if gap > thr:
        print(f"{(a[2] - t0) / 1e6:7.3f} ms  gap {gap:6.1f} us  after {a[0][:48]:48s} before {b[0][:48]}")
    tot += max(gap, 0.0)
print("total gaps us", round(tot, 1))
