"""Worker of tests/test_gpu_dp.py: one data-parallel rank (gloo backend, every rank on cuda:0) doing ONE product train_step
through parallel.GradBucket on the optimiser's packed gradient buffer.  Writes its findings to a JSON file."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(out_path):
    import golden_util as gu
    from elg_amd import parallel
    from elg_amd.CVRP.CVRPEnv import CVRPEnv
    from elg_amd.CVRP.CVRPModel import CVRPModel
    from elg_amd.CVRP.generate_data import generate_vrp_data
    from elg_amd.CVRP.train import train_step
    from elg_amd.CVRP.utils import seed_everything
    from elg_amd.optim import Adam
    rank, world, _ = parallel.init_distributed(backend="gloo")
    dev = "cuda:0"
    seed_everything(1234 + 17 * rank)                  # different initial weights per rank: the broadcast must fix that
    mp = dict(gu.CVRP_MODEL_PARAMS)
    model = CVRPModel(**mp)
    model.decoder.add_local_policy(dev)
    model.to(dev)
    parallel.broadcast_parameters(model)
    before = torch.cat([p.detach().reshape(-1).clone() for p in model.parameters()])
    env = CVRPEnv(multi_width=20, device=dev)
    opt = Adam(model.parameters(), lr=1e-3, weight_decay=1e-6)
    bucket = parallel.GradBucket(model.parameters(), opt)
    assert bucket.optimizer is opt and bucket.flat.data_ptr() == opt.grad_flat.data_ptr()   # the product branch
    batch = generate_vrp_data(6, 20, {"data_type": "uniform"})
    # hook the all-reduce to capture this rank's local packed gradient
    captured = {}
    orig = torch.distributed.all_reduce

    def spy(t, op=torch.distributed.ReduceOp.SUM, **kw):
        if t.data_ptr() == opt.grad_flat.data_ptr():
            captured["local"] = t.detach().clone()
        return orig(t, op=op, **kw)
    torch.distributed.all_reduce = spy
    model.train()
    train_step(model, env, opt, batch, True, bucket, world, check=True)
    torch.distributed.all_reduce = orig
    torch.cuda.synchronize()
    summed = opt.grad_flat.detach().clone()
    # the sum of the ranks' local gradients, through an independent gather on the host
    locals_ = [torch.zeros_like(captured["local"].cpu()) for _ in range(world)]
    torch.distributed.all_gather(locals_, captured["local"].cpu())
    ref_sum = sum(locals_)
    after = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    # the same update from the averaged gradient, by a fresh single-process optimiser on a copy of the initial weights
    model2 = CVRPModel(**mp)
    model2.decoder.add_local_policy(dev)
    model2.to(dev)
    off = 0
    with torch.no_grad():
        for p in model2.parameters():
            p.copy_(before[off:off + p.numel()].view_as(p))
            off += p.numel()
    opt2 = Adam(model2.parameters(), lr=1e-3, weight_decay=1e-6)
    mean = summed * (1.0 / world)
    off = 0
    for p in model2.parameters():
        p.grad = mean[off:off + p.numel()].view_as(p).clone()
        off += p.numel()
    opt2.step()
    torch.cuda.synchronize()
    after2 = torch.cat([p.detach().reshape(-1) for p in model2.parameters()])
    res = {
        "rank": rank, "world": world, "grad_scale": opt.grad_scale,
        "allreduce_err": float((summed.cpu() - ref_sum).abs().max()),
        "grad_abs_max": float(ref_sum.abs().max()),
        "local_differs": float((locals_[0] - locals_[-1]).abs().max()),
        "param_checksum": [float(after.double().sum()), float(after.double().abs().sum())],
        "step_moved": float((after - before).abs().max()),
        "vs_single_process_adam": float((after - after2).abs().max()),
    }
    with open(out_path, "w") as f:
        json.dump(res, f)
    parallel.barrier()


if __name__ == "__main__":
    main(sys.argv[1])
