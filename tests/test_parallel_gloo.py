"""Data-parallel plumbing on CPU with the gloo backend, world_size 2: flat gradient bucket all-reduce
(mean over ranks), parameter broadcast, and the DP == large-batch identity the engine relies on
(loss is a mean over independent instances)."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from elg_amd import parallel
    r, w, _ = parallel.init_distributed("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                  # different initial weights per rank ...
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 1))
    parallel.broadcast_parameters(model)           # ... until rank 0's are broadcast
    torch.manual_seed(7)
    full = torch.randn(8, 6)                        # the "global batch"; each rank takes its shard
    shard = full[rank * 4:(rank + 1) * 4]
    loss = model(shard).pow(2).mean()
    loss.backward()
    bucket = parallel.GradBucket(model.parameters())
    bucket.allreduce(world)
    parallel.barrier()
    out[rank] = dict(params=[p.detach().clone() for p in model.parameters()],
                     grads=[p.grad.detach().clone() for p in model.parameters()], numel=bucket.numel)
    torch.distributed.destroy_process_group()


def test_gradient_allreduce_world2():
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    out = mgr.dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    a, b = out[0], out[1]
    for x, y in zip(a["params"], b["params"]):
        assert torch.equal(x, y)                    # identical replicas after broadcast
    for x, y in zip(a["grads"], b["grads"]):
        assert torch.allclose(x, y, atol=0)        # identical averaged gradients
    # equals the single-process gradient on the whole batch (mean over independent samples)
    torch.manual_seed(100)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 1))
    torch.manual_seed(7)
    full = torch.randn(8, 6)
    model(full).pow(2).mean().backward()
    for p, g in zip(model.parameters(), a["grads"]):
        assert torch.allclose(p.grad, g, rtol=1e-5, atol=1e-7)
    assert a["numel"] == sum(p.numel() for p in model.parameters())


def _curriculum_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import numpy as np
    from elg_amd import parallel
    from elg_amd.CVRP.train import softmax
    parallel.init_distributed("gloo")
    np.random.seed(1000 + rank)                    # every rank has its own generator state (seed + rank in train.py) ...
    gaps = np.array([0.3, 0.1, 0.2])
    draws = []
    for _ in range(20):                            # ... yet must train on the same family (reference train.py:98-100)
        kind = str(np.random.choice(['uniform', 'cluster', 'mixed'], size=1, p=softmax(gaps))[0])
        draws.append(parallel.broadcast_object(kind))
    val = parallel.broadcast_object([15.9, 8.1, 14.5] if rank == 0 else None)      # rank 0 validates, all update the gaps
    out[rank] = dict(draws=draws, val=val)
    torch.distributed.destroy_process_group()


def test_curriculum_choice_is_rank0s_on_every_rank():
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_curriculum_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert out[0]["draws"] == out[1]["draws"] and len(set(out[0]["draws"])) > 1
    assert out[0]["val"] == out[1]["val"] == [15.9, 8.1, 14.5]
    # without a process group the helper is the identity
    from elg_amd import parallel
    assert parallel.broadcast_object("x") == "x"


def _guard_worker(rank, world, port, out, fail_rank):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from elg_amd import parallel
    parallel.init_distributed("gloo", timeout_s=60.0)
    # a sharded evaluation: every rank sums its own instances, the sums meet in sum_over_ranks
    costs = [3.0, 5.0, 7.0, 11.0, 13.0]
    mine = costs[rank::world]

    def phase():                                            # rank-local, no collective (as validate()'s local_sums)
        if rank == fail_rank:
            raise ValueError("boom on this rank")
        return [sum(mine), float(len(mine))]
    try:
        tot = parallel.sum_over_ranks(parallel.guarded(phase))
        out[rank] = ("ok", tot[0] / tot[1])
    except Exception as e:                                  # noqa: BLE001
        out[rank] = (type(e).__name__, str(e))
    torch.distributed.destroy_process_group()


def _run_guard(fail_rank):
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_guard_worker, args=(r, 2, port, out, fail_rank)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0, "a rank hung or died instead of raising"
    return dict(out)


def test_sharded_validation_sum_and_error_guard_world2():
    """validate() under data parallelism: each rank evaluates its share, the sums are added over the ranks and every rank gets
    the same mean; a rank that raises inside the phase makes EVERY rank raise right after it (no rank is left waiting in the next
    collective for the 600 s timeout)."""
    ok = _run_guard(fail_rank=-1)
    assert ok[0] == ok[1] == ("ok", (3.0 + 5.0 + 7.0 + 11.0 + 13.0) / 5.0)
    bad = _run_guard(fail_rank=1)
    assert bad[1] == ("ValueError", "boom on this rank")                      # the failing rank: its own exception
    assert bad[0][0] == "RuntimeError" and "rank 1 failed" in bad[0][1] and "boom" in bad[0][1]
    from elg_amd import parallel
    assert parallel.guarded(lambda: 42) == 42 and parallel.sum_over_ranks([1, 2.5]) == [1.0, 2.5]


def test_cpu_quota_caps_the_intra_op_pool(tmp_path, monkeypatch):
    """parallel.respect_cpu_quota: cgroup v2 / v1 quota -> torch's thread count, shared between the node's ranks; no quota, no change."""
    import torch
    from elg_amd import parallel
    before = torch.get_num_threads()
    try:
        torch.set_num_threads(6)
        (tmp_path / "cpu.max").write_text("max 100000\n")
        assert parallel.respect_cpu_quota(str(tmp_path)) == 6
        (tmp_path / "cpu.max").write_text("400000 100000\n")
        assert parallel.respect_cpu_quota(str(tmp_path)) == 4 and torch.get_num_threads() == 4
        monkeypatch.setenv("LOCAL_WORLD_SIZE", "2")
        assert parallel.respect_cpu_quota(str(tmp_path)) == 2
        monkeypatch.delenv("LOCAL_WORLD_SIZE")
        (tmp_path / "cpu.max").unlink()
        (tmp_path / "cpu").mkdir()
        (tmp_path / "cpu" / "cpu.cfs_quota_us").write_text("100000\n")
        (tmp_path / "cpu" / "cpu.cfs_period_us").write_text("100000\n")
        torch.set_num_threads(3)
        assert parallel.respect_cpu_quota(str(tmp_path)) == 1
        assert parallel.respect_cpu_quota(str(tmp_path / "nothing_here")) == 1       # no cgroup files: unchanged
    finally:
        torch.set_num_threads(before)
