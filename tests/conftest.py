import faulthandler
import os
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

# The oracle is torch-CPU.  On the GPU box's 256-thread host torch's default intra-op pool oversubscribes (one encoder pass
# took 98 s there, bench.py cpu_baseline); every oracle call of the suite runs on a fixed small pool instead.
os.environ.setdefault("OMP_NUM_THREADS", "8")
os.environ.setdefault("MKL_NUM_THREADS", "8")

# Run order of the `-m gpu` suite (the driver runs it with -x): the tests that state north_star's tolerances first,
# everything that starts child processes or opens sockets last.
GPU_ORDER = [
    "test_gpu_logits", "test_gpu_protocol", "test_gpu_tsp_protocol", "test_gpu_encoder", "test_gpu_fullsize", "test_gpu_forward",
    "test_gpu_coop", "test_gpu_backward", "test_gpu_check", "test_gpu_train_glue", "test_gpu_glimpse_bwd", "test_gpu_gemm",
    "test_gpu_variants", "test_gpu_ensemble", "test_gpu_train_large", "test_gpu_testers", "test_gpu_large", "test_gpu_xxl",
    "test_gpu_zz_dp",
]
PER_TEST_LIMIT_S = 420          # no single test may hang the run: dump every stack and abort (see pytest_runtest_call)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` via gpurun)")
    import torch
    torch.set_num_threads(8)
    try:
        torch.set_num_interop_threads(2)
    except RuntimeError:
        pass
    faulthandler.enable()


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests must fail (not skip) on a box without the HIP path, but plain `pytest tests`
    on a CPU-only container should not try them."""
    def rank(it):
        mod = os.path.splitext(os.path.basename(str(it.fspath)))[0]
        if mod in GPU_ORDER:
            return GPU_ORDER.index(mod)
        return len(GPU_ORDER) - 1 if mod.startswith("test_gpu") else -1      # unknown GPU files just before the dp tests
    items.sort(key=rank)                    # stable: the order inside a file is kept
    import torch
    if torch.cuda.is_available():
        return
    markexpr = config.getoption("-m") or ""
    if "gpu" in markexpr and "not gpu" not in markexpr:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_call(item):
    """Watchdog: a test still running after PER_TEST_LIMIT_S gets every thread's stack written to stderr and the process is
    ended (a hung HIP call cannot be interrupted from Python; a 20-minute silent hang tells nobody anything)."""
    faulthandler.dump_traceback_later(PER_TEST_LIMIT_S, exit=True)
    try:
        yield
    finally:
        faulthandler.cancel_dump_traceback_later()


_T0 = time.time()


def pytest_sessionfinish(session, exitstatus):
    """Leave the GPU quiet before interpreter exit, and bound the time that may take."""
    faulthandler.dump_traceback_later(120, exit=True)          # cancelled below; covers a hung device synchronisation
    try:
        import torch
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            torch.cuda.synchronize()
    finally:
        faulthandler.cancel_dump_traceback_later()
    # interpreter shutdown itself (atexit hooks, HIP runtime teardown) gets the same bound, keeping the suite's exit status
    import threading

    def _last_resort(status=int(exitstatus)):
        time.sleep(120)
        faulthandler.dump_traceback(all_threads=True)
        os._exit(status)
    threading.Thread(target=_last_resort, daemon=True).start()


@pytest.fixture(autouse=True)
def _fixed_host_seeds():
    """Every test starts from the same host RNG state: the POMO starts are drawn with Python's `random` (reference
    CVRPModel.py:46-51), the instance generators use numpy / torch -- a run of the suite is then reproducible draw for draw."""
    import random

    import numpy as np
    import torch
    random.seed(1234)
    np.random.seed(1234)
    torch.manual_seed(1234)
    yield
