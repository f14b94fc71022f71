"""The data-parallel path the product uses (parallel.GradBucket on elg_amd.optim.Adam's packed gradient buffer, 1/world
folded into the Adam kernel), with two real ranks: both on cuda:0, gloo backend, fresh child processes.  And bench.py under
the launcher the driver uses (`python -m torch.distributed.run`)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(extra):
    e = dict(os.environ)
    e.update({"MASTER_ADDR": "127.0.0.1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    e.update(extra)
    return e


def test_two_ranks_share_one_gradient(tmp_path):
    port = str(29500 + (os.getpid() % 400))
    procs, outs = [], []
    for r in range(2):
        out = tmp_path / f"rank{r}.json"
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), str(out)],
                                      env=_env({"RANK": str(r), "WORLD_SIZE": "2", "LOCAL_RANK": "0", "MASTER_PORT": port}),
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=600)[0] for p in procs]
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-3000:]
    res = [json.load(open(o)) for o in outs]
    for r in res:
        assert r["world"] == 2 and r["grad_scale"] == 0.5
        assert r["local_differs"] > 0                                     # the ranks really had different shards
        assert r["allreduce_err"] <= 1e-6 * r["grad_abs_max"]               # bucket = sum of the local packed gradients
        assert r["step_moved"] > 0
        assert r["vs_single_process_adam"] == 0.0                          # = Adam on the averaged gradient, bit for bit
    assert res[0]["param_checksum"] == res[1]["param_checksum"]             # replicas identical after the step


def test_bench_under_the_launcher():
    port = str(29900 + (os.getpid() % 90))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=_env({}), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["value"] > 0 and out["scaling"] == "weak"
