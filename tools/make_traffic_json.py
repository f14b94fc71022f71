"""profiles/roofline_traffic.json from the two PMC passes of tools/measure_traffic.sh.
    python tools/make_traffic_json.py out.json <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <bench line of a pass>
HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes): on gfx950 FETCH_SIZE tallies 64 B per 128-B request of
wide coalesced reads (guides/MI355X_MICROARCH.md, HBM section) -> doubled, an upper bound (part of the kernel's reads are
dword loads, for which the counter is exact); WRITE_SIZE is exact.  The record carries the decode steps of the run it was
measured on and the digest of the kernel sources (elg_amd.build._digest): bench.py reports it only for that same build."""
import glob
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from elg_amd import build  # noqa: E402

out, d_fetch, d_write, bench_log = sys.argv[1:5]
KERNEL = "rollout_fwd_coop_kernel"


def counter(d, name):
    vals = []
    for db in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        c = sqlite3.connect(db)
        for kn, val in c.execute("select kernel_name, value from counters_collection where counter_name = ?", (name,)):
            if KERNEL in kn:
                vals.append(float(val))
    if not vals:
        raise SystemExit(f"no {name} rows for {KERNEL} under {d}")
    return sum(vals) / len(vals), len(vals)


fetch_kb, n_f = counter(d_fetch, "FETCH_SIZE")
write_kb, n_w = counter(d_write, "WRITE_SIZE")
line = None
for ln in open(bench_log):
    if ln.startswith("{"):
        line = json.loads(ln)
rec = {
    "kernel": "rollout_fwd_coop_kernel<false, true>  (CVRP-100, B=64, pomo 100, training forward of bench.py: saves the backward rows)",
    "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
    "fetch_bytes_x2": int(2 * fetch_kb * 1024), "write_bytes": int(write_kb * 1024),
    "hbm_bytes_per_launch": int((2 * fetch_kb + write_kb) * 1024),
    "launches": {"FETCH_SIZE": n_f, "WRITE_SIZE": n_w},
    "decode_steps_mean": line["roofline"]["decode_steps_mean"] if line else None,
    "decode_steps_max": line["roofline"]["decode_steps_max"] if line else None,
    "source_digest": build._digest(),
    "correction": "gfx950: FETCH_SIZE tallies 64 B per 128-B request of wide (16 B/lane) coalesced reads -> doubled (upper bound); "
                  "WRITE_SIZE exact (guides/MI355X_MICROARCH.md, HBM section); separate --pmc passes (tools/measure_traffic.sh)",
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline "
               "--no-secondary --no-fast --sustain-s 0",
}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec)[:600])
