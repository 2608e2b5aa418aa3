import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "taichi-2d-vof_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


ORACLE_SO = os.path.join(ROOT, "oracle", "_build", "libvof_oracle.so")


def usable_cores():
    """Host cores this process may use: the affinity mask capped by the cgroup CPU quota.  The GPU box
    shows 256 logical CPUs behind a 16-CPU quota; libgomp would start 256 threads and thrash (the
    1536^2 oracle run of test_front_regime_gpu.py: 650 s instead of 20 s)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


# read by libgomp when the oracle library loads.  Capped at 16: most oracle runs of the suite are small grids
# whose hundreds of parallel regions per step cost more in barriers than they gain beyond that (a box
# without a CPU quota ran the GPU suite in 590 s with 256 threads, 85 s with 16)
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, usable_cores())))


@pytest.fixture(scope="session")
def oracle_api():
    """The CPU oracle (test infrastructure) bound through the same ctypes prototypes."""
    from vof2d import _abi
    if not os.path.exists(ORACLE_SO):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    lib = ctypes.CDLL(ORACLE_SO)
    return _abi.bind(lib, "ovof_", optional=_abi.GPU_ONLY)


@pytest.fixture(scope="session")
def hip_api():
    from vof2d._lib import hip_api as load
    return load()
