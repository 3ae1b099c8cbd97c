import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _cpu_budget() -> int:
    """CPUs this process may actually use: affinity mask, capped by the cgroup quota (OpenMP only sees the former)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


# The CPU oracle is OpenMP code with many short parallel regions.  On a box whose container is limited to a few CPUs while all
# of the host's are visible, 16 spinning threads on 2 cores turn a 40-second suite into many minutes: size the team to the real
# budget and let idle threads sleep when the budget is small.  (bench.py's cpu_baseline does not go through here.)
_n = _cpu_budget()
os.environ.setdefault("LMRS_REF_THREADS", str(min(16, _n)))
if _n < 16:
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available() -> bool:
    """A HIP device is visible: probed once through the HIP runtime, without touching the product.  (With a device present a
    missing liblmrs_hip.so is NOT a reason to skip: the tests then fail loudly - there is no fallback to hide behind.)"""
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value >= 1
    except OSError:
        return False


def pytest_collection_modifyitems(config, items):
    """A plain `pytest` on a box without a GPU skips the gpu-marked tests instead of failing them (on a GPU box nothing is
    skipped: there the HIP path must run, and a missing library is an error, not a skip)."""
    if any(it.get_closest_marker("gpu") for it in items) and not _gpu_available():
        skip = pytest.mark.skip(reason="no HIP device visible (gpu-marked tests run on the MI355X box: pytest -m gpu)")
        for it in items:
            if it.get_closest_marker("gpu"):
                it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
