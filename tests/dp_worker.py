"""Worker of tests/test_gpu_zz_dp.py: one data-parallel rank doing product train_steps through parallel.GradBucket on the
optimiser's packed gradient buffer.  Writes its findings to a JSON file; prints a progress marker after every stage and
dumps every thread's stack if it is still alive after ELG_DP_WATCHDOG seconds, so that a hang names its place.

    dp_worker.py OUT two_ranks     gloo, world 2, both ranks on cuda:0 (RANK / WORLD_SIZE / MASTER_* from the env)
    dp_worker.py OUT rccl_world1   nccl (= RCCL), world 1, ELG_FORCE_DIST=1: the all-reduce of the training step on RCCL
    dp_worker.py OUT whole_step    a whole CVRP-100 train_step: world 2 (gloo, both ranks on cuda:0, 32 instances each) or,
                                   without WORLD_SIZE > 1, one process over the same 64 instances"""
import faulthandler
import json
import os
import sys
import time

faulthandler.enable()
faulthandler.dump_traceback_later(float(os.environ.get("ELG_DP_WATCHDOG", "100")), exit=True)

T0 = time.time()


def mark(what):
    print(f"[dp_worker rank {os.environ.get('RANK', '?')} +{time.time() - T0:6.1f}s] {what}", flush=True)


mark("start")
import torch  # noqa: E402

torch.set_num_threads(4)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _flat(model):
    return torch.cat([p.detach().reshape(-1).clone() for p in model.parameters()])


def _new_model(mp, dev, values=None):
    from elg_amd.CVRP.CVRPModel import CVRPModel
    model = CVRPModel(**mp)
    model.decoder.add_local_policy(dev)
    model.to(dev)
    if values is not None:
        off = 0
        with torch.no_grad():
            for p in model.parameters():
                p.copy_(values[off:off + p.numel()].view_as(p))
                off += p.numel()
    return model


def two_ranks(out_path):
    import golden_util as gu
    from elg_amd import parallel
    from elg_amd.CVRP.CVRPEnv import CVRPEnv
    from elg_amd.CVRP.generate_data import generate_vrp_data
    from elg_amd.CVRP.train import train_step
    from elg_amd.CVRP.utils import seed_everything
    from elg_amd.optim import Adam
    rank, world, _ = parallel.init_distributed(backend="gloo", timeout_s=90)
    mark("process group up")
    dev = "cuda:0"
    seed_everything(1234 + 17 * rank)                  # different initial weights per rank: the broadcast must fix that
    mp = dict(gu.CVRP_MODEL_PARAMS)
    model = _new_model(mp, dev)
    mark("model on the GPU")
    parallel.broadcast_parameters(model)
    mark("parameters broadcast")
    before = _flat(model)
    env = CVRPEnv(multi_width=20, device=dev)
    opt = Adam(model.parameters(), lr=1e-3, weight_decay=1e-6)
    bucket = parallel.make_bucket(model.parameters(), opt)
    assert bucket is not None and bucket.optimizer is opt and bucket.flat.data_ptr() == opt.grad_flat.data_ptr()
    batch = generate_vrp_data(6, 20, {"data_type": "uniform"})
    # hook the all-reduce to capture this rank's local packed gradient (staged through pinned host memory on gloo)
    captured = {}
    orig = torch.distributed.all_reduce

    def spy(t, op=torch.distributed.ReduceOp.SUM, **kw):
        if t.numel() == opt.numel:
            captured["local"] = t.detach().cpu().clone()
            captured["on_host"] = not t.is_cuda
        return orig(t, op=op, **kw)
    torch.distributed.all_reduce = spy
    model.train()
    train_step(model, env, opt, batch, True, bucket, world, check=True)
    torch.distributed.all_reduce = orig
    mark("train_step returned")
    torch.cuda.synchronize()
    mark("device idle")
    summed = opt.grad_flat.detach().clone()
    # the sum of the ranks' local gradients, through an independent gather on the host
    locals_ = [torch.zeros_like(captured["local"]) for _ in range(world)]
    torch.distributed.all_gather(locals_, captured["local"])
    ref_sum = sum(locals_)
    after = _flat(model)
    # the same update from the averaged gradient, by a fresh single-process optimiser on a copy of the initial weights
    model2 = _new_model(mp, dev, before)
    opt2 = Adam(model2.parameters(), lr=1e-3, weight_decay=1e-6)
    mean = summed * (1.0 / world)
    off = 0
    for p in model2.parameters():
        p.grad = mean[off:off + p.numel()].view_as(p).clone()
        off += p.numel()
    opt2.step()
    torch.cuda.synchronize()
    after2 = _flat(model2)
    # ---- sharded validation (train.validate under data parallelism): each rank evaluates the instances rank, rank + world, ...
    # of every set; against the one-process evaluation of the same sets (test_rollout, no collective), and identical on both ranks
    import pickle, tempfile
    import numpy as np
    from torch.utils.data import DataLoader
    from elg_amd.CVRP import train as T_
    from elg_amd.CVRP.generate_data import VRPDataset
    vdir = tempfile.mkdtemp()
    rs = np.random.RandomState(99)                                      # the same files on every rank
    for kind in ("uniform", "cluster", "mixed"):
        rows = [(rs.rand(2).tolist(), rs.rand(20, 2).tolist(), rs.randint(1, 10, 20).astype(float).tolist(), 30.0) for _ in range(7)]
        pickle.dump(rows, open(os.path.join(vdir, f"vrp_{kind}100_1000_seed1234.pkl"), "wb"))
    sharded = T_.validate(model, 20, dev, True, data_dir=vdir)
    venv = CVRPEnv(multi_width=20, device=dev)
    whole = [T_.test_rollout(DataLoader(VRPDataset(os.path.join(vdir, f"vrp_{k}100_1000_seed1234.pkl"), num_samples=1000), batch_size=1000),
                             venv, model) for k in ("uniform", "cluster", "mixed")]
    mark("sharded validation done")
    model.train()
    res = {
        "validate_sharded": sharded, "validate_whole": whole,
        "rank": rank, "world": world, "grad_scale": opt.grad_scale, "staged_on_host": bool(captured["on_host"]),
        "bucket_calls": bucket.calls,
        "allreduce_err": float((summed.cpu() - ref_sum).abs().max()),
        "grad_abs_max": float(ref_sum.abs().max()),
        "local_differs": float((locals_[0] - locals_[-1]).abs().max()),
        "param_checksum": [float(after.double().sum()), float(after.double().abs().sum())],
        "step_moved": float((after - before).abs().max()),
        "vs_single_process_adam": float((after - after2).abs().max()),
    }
    with open(out_path, "w") as f:
        json.dump(res, f)
    mark("results written")
    parallel.barrier()
    torch.distributed.destroy_process_group()
    mark("done")


def rccl_world1(out_path):
    """The RCCL branch of a training step on one GPU: backend nccl at world size 1 (ELG_FORCE_DIST=1).  A step through the
    bucket (pack -> ncclAllReduce on the optimiser's buffer -> ctypes-launched Adam on torch's current stream): the reduced
    buffer equals the local gradient bit for bit (sum over one rank), the update equals a separate Adam on that gradient
    bit for bit, and the gradient agrees with the same step taken without a process group in the data path (to the
    run-to-run noise of the backward's float atomics).  Two more steps prove the branch keeps running."""
    import golden_util as gu
    from elg_amd import parallel
    from elg_amd.CVRP.CVRPEnv import CVRPEnv
    from elg_amd.CVRP.generate_data import generate_vrp_data
    from elg_amd.CVRP.train import train_step
    from elg_amd.CVRP.utils import seed_everything
    from elg_amd.optim import Adam
    assert parallel.force_group()
    rank, world, local = parallel.init_distributed(timeout_s=90)
    assert torch.distributed.get_backend() == "nccl" and world == 1
    mark("RCCL process group up")
    dev = f"cuda:{local}"
    mp = dict(gu.CVRP_MODEL_PARAMS)
    seed_everything(77)
    init = _flat(_new_model(mp, dev))
    grads = {}
    res = {}
    for use_bucket in (True, False):
        model = _new_model(mp, dev, init)
        if use_bucket:
            parallel.broadcast_parameters(model)
            assert torch.equal(_flat(model), init)
        env = CVRPEnv(multi_width=20, device=dev)
        opt = Adam(model.parameters(), lr=1e-3, weight_decay=1e-6)
        bucket = parallel.make_bucket(model.parameters(), opt) if use_bucket else None
        assert (bucket is not None) == use_bucket
        captured = {}
        orig = torch.distributed.all_reduce

        def spy(t, op=torch.distributed.ReduceOp.SUM, **kw):
            if t.numel() == opt.numel:
                assert t.is_cuda and t.data_ptr() == opt.grad_flat.data_ptr()      # RCCL works on the optimiser's buffer
                captured["local"] = t.detach().clone()
            return orig(t, op=op, **kw)
        torch.distributed.all_reduce = spy
        seed_everything(5)
        model.train()
        batch = generate_vrp_data(6, 20, {"data_type": "uniform"})
        train_step(model, env, opt, batch, True, bucket, world, check=True)
        torch.cuda.synchronize()
        grads[use_bucket] = opt.grad_flat.detach().clone()
        if use_bucket:
            after = _flat(model)
            model2 = _new_model(mp, dev, init)
            opt2 = Adam(model2.parameters(), lr=1e-3, weight_decay=1e-6)
            off = 0
            for p in model2.parameters():
                p.grad = captured["local"][off:off + p.numel()].view_as(p).clone()
                off += p.numel()
            opt2.step()
            torch.cuda.synchronize()
            res["reduced_vs_local"] = float((grads[True] - captured["local"]).abs().max())
            res["vs_single_process_adam"] = float((after - _flat(model2)).abs().max())
            res["moved"] = float((after - init).abs().max())
            for _ in range(2):
                train_step(model, env, opt, generate_vrp_data(6, 20, {"data_type": "uniform"}), True, bucket, world, check=True)
            torch.cuda.synchronize()
            res["bucket_calls"] = bucket.calls
            res["finite_after_3_steps"] = bool(torch.isfinite(_flat(model)).all())
        torch.distributed.all_reduce = orig
        mark(f"bucket={use_bucket} done")
    res.update({"backend": torch.distributed.get_backend(), "world": world, "ranks_seen": parallel.ranks_seen(),
                "grad_abs_max": float(grads[False].abs().max()),
                "bucket_vs_plain_grad": float((grads[True] - grads[False]).abs().max())})
    with open(out_path, "w") as f:
        json.dump(res, f)
    parallel.barrier()
    torch.distributed.destroy_process_group()
    mark("done")


def whole_step(out_path):
    """A whole product train_step at CVRP-100 (POMO 100): two ranks x 32 instances against one process x the same 64 instances,
    the same sampled trajectories (the engine's `uniforms` hook: the philox stream is keyed by the trajectory's index IN the
    batch, so rank 1's instances would otherwise draw other numbers than instances 32..63 of the one process).  The loss is a
    mean over instances and the advantage's baseline is per instance, so the ranks' averaged gradient IS the one-process
    gradient; what differs is the order of the f32 sums over the rows (32 vs 64 instances per reduction)."""
    import numpy as np
    import golden_util as gu
    from elg_amd import engine as eng
    from elg_amd import parallel
    from elg_amd.CVRP.CVRPEnv import CVRPEnv
    from elg_amd.CVRP.generate_data import generate_vrp_data
    from elg_amd.CVRP.train import train_step
    from elg_amd.CVRP.utils import seed_everything
    from elg_amd.optim import Adam
    rank, world, _ = parallel.world_info()
    if world > 1:
        parallel.init_distributed(backend="gloo", timeout_s=90)
        mark("process group up")
    dev = "cuda:0"
    N, M, Btot = 100, 100, 64
    mp = dict(gu.CVRP_MODEL_PARAMS)
    seed_everything(4242)                               # the same weights, instances and uniforms in every process
    model = _new_model(mp, dev)
    parallel.broadcast_parameters(model)
    data = generate_vrp_data(Btot, N, {"data_type": "uniform"})
    Tcap = eng.max_steps(eng.L.PROBLEM_CVRP, N + 1)
    uni = torch.rand(Btot, M, Tcap)
    per = Btot // world
    sl = slice(rank * per, (rank + 1) * per)
    batch = {k: v[sl] for k, v in data.items()}
    orig_fwd = eng.rollout_forward

    def fwd(*a, **kw):
        kw["uniforms"] = uni[sl]
        return orig_fwd(*a, **kw)
    eng.rollout_forward = fwd
    import elg_amd.CVRP.utils as U_
    assert U_.eng is eng
    env = CVRPEnv(multi_width=M, device=dev)
    opt = Adam(model.parameters(), lr=1e-4, weight_decay=1e-6)
    bucket = parallel.make_bucket(model.parameters(), opt)
    assert (bucket is not None) == (world > 1)
    before = _flat(model)
    model.train()
    J, rewards = train_step(model, env, opt, batch, True, bucket, world, check=True)
    torch.cuda.synchronize()
    eng.rollout_forward = orig_fwd
    mark("train_step returned")
    if world > 1:
        grad = (opt.grad_flat.detach() * opt.grad_scale).cpu().numpy()       # the averaged gradient Adam applied
    else:
        grad = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).cpu().numpy()
    np.save(out_path + ".grad.npy", grad)
    np.save(out_path + ".after.npy", _flat(model).cpu().numpy())
    res = {"rank": rank, "world": world, "instances": per, "loss": float(J), "reward_sum": float(rewards.double().sum()),
           "moved": float((_flat(model) - before).abs().max()), "bucket_calls": bucket.calls if bucket else 0}
    with open(out_path, "w") as f:
        json.dump(res, f)
    mark("results written")
    if world > 1:
        parallel.barrier()
        torch.distributed.destroy_process_group()
    mark("done")


if __name__ == "__main__":
    {"two_ranks": two_ranks, "rccl_world1": rccl_world1, "whole_step": whole_step}[sys.argv[2]](sys.argv[1])
    faulthandler.cancel_dump_traceback_later()
