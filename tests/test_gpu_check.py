"""utils.check_feasible (elg_check_feasible) -- positive and NEGATIVE cases, against the reference's assertions
(CVRP/utils.py:90-119: "Invalid tour" / "Used more than capacity"; TSP/utils.py:72-78) restated in the oracle."""
import numpy as np
import pytest
import torch

from oracle import elg_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _tours(n=20, multi=7, seed=0):
    """Feasible CVRP tours: customers in random order, a depot return whenever the next customer would not fit."""
    rng = np.random.RandomState(seed)
    demand = rng.randint(1, 10, n).astype(np.float32) / np.float32(30.0)
    rows = []
    for _ in range(multi):
        perm = rng.permutation(n) + 1
        tour, used = [0], np.float32(0)
        for c in perm:
            if used + demand[c - 1] > 1.0:
                tour.append(0)
                used = np.float32(0)
            tour.append(int(c))
            used = np.float32(used + demand[c - 1])
        rows.append(tour)
    T = max(len(r) for r in rows) + 2
    pi = np.zeros((multi, T), dtype=np.int64)
    for i, r in enumerate(rows):
        pi[i, :len(r)] = r
    return pi, demand


def _run(pi, demand):
    from elg_amd.CVRP.utils import check_feasible
    check_feasible(torch.from_numpy(pi)[None].to(DEV), torch.from_numpy(demand)[None].to(DEV))


def test_feasible_tours_pass():
    pi, demand = _tours()
    orc.check_feasible(pi, demand)
    _run(pi, demand)


def test_duplicate_customer_is_an_invalid_tour():
    pi, demand = _tours()
    bad = pi.copy()
    row = bad[3]
    cust = np.nonzero(row)[0]
    row[cust[4]] = row[cust[2]]                       # one customer twice, another never
    with pytest.raises(AssertionError, match="Invalid tour"):
        orc.check_feasible(bad, demand)
    with pytest.raises(AssertionError, match="Invalid tour"):
        _run(bad, demand)


def test_missing_customer_and_out_of_range_entries():
    pi, demand = _tours()
    bad = pi.copy()
    bad[0][np.nonzero(bad[0])[0][0]] = 0               # a customer replaced by a depot visit
    with pytest.raises(AssertionError, match="Invalid tour"):
        _run(bad, demand)
    bad = pi.copy()
    bad[1, 1] = 21                                     # node id out of range
    with pytest.raises(AssertionError, match="Invalid tour"):
        _run(bad, demand)


def test_capacity_exceeded_by_a_hair():
    """The reference tolerates 1 + 1e-4 (fp32): 1.00005 passes, 1.0002 raises."""
    demand = np.full(4, 0.25, dtype=np.float32)
    pi = np.array([[0, 1, 2, 3, 4, 0]], dtype=np.int64)
    _run(pi, demand)                                   # exactly 1.0
    d_ok = demand.copy(); d_ok[3] = np.float32(0.25005)
    orc.check_feasible(pi, d_ok)
    _run(pi, d_ok)
    d_bad = demand.copy(); d_bad[3] = np.float32(0.2502)
    with pytest.raises(AssertionError, match="Used more than capacity"):
        orc.check_feasible(pi, d_bad)
    with pytest.raises(AssertionError, match="Used more than capacity"):
        _run(pi, d_bad)
    # a depot return in between resets the load
    _run(np.array([[0, 1, 2, 3, 0, 4, 0]], dtype=np.int64), d_bad)


def test_tsp_permutation_check():
    from elg_amd.TSP.utils import check_feasible
    g = torch.Generator().manual_seed(0)
    pi = torch.stack([torch.randperm(50, generator=g) for _ in range(9)])[None].to(DEV)
    assert check_feasible(pi)
    bad = pi.clone()
    bad[0, 4, 7] = bad[0, 4, 8]
    assert not check_feasible(bad)
