import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def usable_cpus():
    """CPUs this process may actually use: the affinity mask capped by the cgroup's CPU quota (the GPU boxes show 256 logical
    CPUs and grant 16; torch's default of one thread per visible core then spends the quota in a fraction of every scheduling
    period and the fp64 oracle runs 2-3x slower, erratically)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, -(-q // per)))
        except (OSError, ValueError):
            pass
    return n


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "frozen_mode: GPU tests of the frozen arithmetic modes (f32x3, operand-only bf16); part of "
                            "-m gpu, can be left out with -m 'gpu and not frozen_mode'")
    import torch
    # the oracle (CPU, fp64) is where the suite's time goes; beyond ~32 threads its small fp64 convolutions gain nothing
    torch.set_num_threads(max(1, min(usable_cpus(), torch.get_num_threads(), 32)))


# Collection order of the GPU suite: the fp32 hot path first (op parity, step parity, BASELINE-shape parity), then the
# "next" rows (data pipeline, data parallel), the opt-in arithmetic modes last -- `pytest -x` (the driver's form) then reaches
# every core parity test before anything optional can stop the run.
_ORDER = ["test_gpu_ops", "test_gpu_step", "test_gpu_shapes", "test_gpu_data", "test_gpu_dp", "test_gpu_bf16s",
          "test_gpu_bf16", "test_gpu_f32x3", "test_gpu_hd_step"]


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return _ORDER.index(name) if name in _ORDER else -1       # CPU test files keep their place in front
    items.sort(key=rank)                                          # stable: order inside a file is unchanged


@pytest.fixture(scope="session")
def golden():
    import json
    import numpy as np
    here = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(here, "golden.json")) as f:
        meta = json.load(f)
    arrays = dict(np.load(os.path.join(here, "golden_arrays.npz")))
    return meta, arrays
