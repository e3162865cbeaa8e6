import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _cgroup_cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max / v1 cfs quota), rounded up; None when unlimited or unreadable."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else max(1, -(-int(q) // int(p)))
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else max(1, -(-q // p))
    except (OSError, ValueError):
        return None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The fp32 oracle is host arithmetic.  The GPU boxes of this pool show 256 logical CPUs under a 16-CPU cgroup quota: with torch's default
    # (128 threads) the oracle's 25-frame ViT chunk takes 15.8 s, with 16 threads 4.1 s (gpurun_out/r6a/threads.log) -- the throttled threads
    # spin.  One ATen thread per CPU of the quota.
    import torch
    quota = _cgroup_cpu_quota()
    if quota is not None and quota < torch.get_num_threads():
        torch.set_num_threads(quota)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def ctx():
    """libpgv context on cuda:0 -- GPU tests only.  Fails (never skips) when the HIP path is unavailable."""
    import torch
    from video_llava_amd import _lib
    assert torch.cuda.is_available(), "GPU tests need a GPU; the product path has no CPU fallback"
    return _lib.Context.get(0)
