"""Reader for the TSPLIB95-style CVRP files of VRPLIB (`.vrp`) and their solution files (`.sol`).

The reference calls the third-party `vrplib` package (CVRP/test_vrplib.py:1,57,112-113; un-vendored, no
version pinned); this module returns the same dictionary fields the reference consumes:
read_instance -> node_coord (N1,2), demand (N1,), capacity, depot [0-based ids];
read_solution -> routes (lists of 1-based customer ids as in the file), cost."""
from __future__ import annotations

import numpy as np


def read_instance(path: str) -> dict:
    fields, coords, demands, depots = {}, [], [], []
    section = None
    with open(path) as fh:
        for raw in fh:
            line = raw.strip()
            if not line:
                continue
            key = line.upper()
            if key == "EOF":
                break
            if key.endswith("SECTION"):
                section = key
                continue
            if section is None:
                if ":" in line:
                    name, value = line.split(":", 1)
                    fields[name.strip().upper()] = value.strip()
                continue
            parts = line.split()
            if section.startswith("NODE_COORD"):
                coords.append((float(parts[1]), float(parts[2])))
            elif section.startswith("DEMAND"):
                demands.append(float(parts[1]))
            elif section.startswith("DEPOT"):
                v = int(parts[0])
                if v > 0:
                    depots.append(v - 1)
    dim = int(fields["DIMENSION"])
    if len(coords) != dim or len(demands) != dim:
        raise ValueError(f"{path}: DIMENSION {dim} but {len(coords)} coordinates / {len(demands)} demands")
    return {"name": fields.get("NAME", ""), "dimension": dim, "capacity": float(fields["CAPACITY"]),
            "edge_weight_type": fields.get("EDGE_WEIGHT_TYPE", "EUC_2D"),
            "node_coord": np.asarray(coords, dtype=np.float64), "demand": np.asarray(demands, dtype=np.float64),
            "depot": np.asarray(depots, dtype=np.int64)}


def read_solution(path: str) -> dict:
    routes, cost = [], None
    with open(path) as fh:
        for raw in fh:
            line = raw.strip()
            low = line.lower()
            if low.startswith("route"):
                routes.append([int(tok) for tok in line.split(":", 1)[1].split()])
            elif low.startswith("cost"):
                cost = float(line.split()[1])
    return {"routes": routes, "cost": cost}
