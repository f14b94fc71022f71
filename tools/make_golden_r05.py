"""Round-5 fixtures: the three 1000-instance CVRP-100 validation sets `validate()` opens during training (reference
CVRP/train.py:42-80 reads CVRP/data/vrp_{uniform,cluster,mixed}100_1000_seed1234.pkl; their solver means are the constants of
train.py:146).  They are DATA files of the reference; kept here as one compressed npz, rounded to float32 -- which is what the reference's loader
does with them (generate_data.py:108-117 make_instance: torch.tensor(loc, dtype=torch.float); demands and capacities are small
integers) -- so that the full training schedule can be run -- and validated against the reference's own
comparators -- on a box where /root/reference does not exist.
    python tools/make_golden_r05.py        (in the build container)"""
import os, pickle, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/CVRP/data"
out = {}
for kind in ("uniform", "cluster", "mixed"):
    d = pickle.load(open(os.path.join(SRC, f"vrp_{kind}100_1000_seed1234.pkl"), "rb"))
    depot = np.array([x[0] for x in d], dtype=np.float64)
    loc = np.array([x[1] for x in d], dtype=np.float64)
    dem = np.array([x[2] for x in d], dtype=np.float64)
    cap = np.array([x[3] for x in d], dtype=np.float64)
    for a in (dem, cap):
        assert np.array_equal(a.astype(np.float32).astype(np.float64), a), "not float32-exact"
    out[f"{kind}_depot"], out[f"{kind}_loc"] = depot.astype(np.float32), loc.astype(np.float32)
    out[f"{kind}_demand"], out[f"{kind}_capacity"] = dem.astype(np.float32), cap.astype(np.float32)
    print(kind, depot.shape, loc.shape, dem.shape, "capacity", np.unique(cap))
p = os.path.join(ROOT, "tests", "golden", "r05_cvrp_val100_sets.npz")
np.savez_compressed(p, **out)
print(p, os.path.getsize(p) / 1e6, "MB")
