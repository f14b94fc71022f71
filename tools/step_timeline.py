"""Summarise one training step from a rocprofv3 --kernel-trace sqlite db: kernels > 80 us with start offsets."""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name,start,end,queue_id from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if 'rollout_fwd' in r[0]]
i0, i1 = idx[-2], idx[-1]
t0 = rows[i0][1]
print("step span ms", (rows[i1][1] - t0) / 1e6, "kernels", i1 - i0)
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 80
for r in rows[i0:i1]:
    dur = (r[2] - r[1]) / 1e3
    if dur > thr:
        print(f"{(r[1]-t0)/1e6:8.3f} ms  +{dur/1e3:6.3f} ms  q{r[3]}  {r[0][:90]}")
