"""Per-kernel counter averages of a rocprofv3 --pmc rocpd SQLite database -> JSON (the PMC summaries under profiles/).
    python tools/rocpd_pmc.py out.json db1 [db2 ...]      (one database per --pmc pass)"""
import json
import sqlite3
import sys

out, dbs = sys.argv[1], sys.argv[2:]
res = {}
for db in dbs:
    c = sqlite3.connect(db)
    q = ("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name")
    for name, ctr, val, n in c.execute(q):
        short = name.split("(")[0].replace("void ", "")
        res.setdefault(short, {})[ctr] = {"avg_per_launch": val, "launches": n}
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
for k, v in sorted(res.items(), key=lambda kv: -max(x["avg_per_launch"] for x in kv[1].values()))[:12]:
    print(k[:70], {c: f"{x['avg_per_launch']:.3g}" for c, x in v.items()})
