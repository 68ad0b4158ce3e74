import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _available_cpus() -> int:
    """affinity mask capped by the cgroup CPU quota (the GPU boxes give a container 16 of 256 hardware threads)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


os.environ.setdefault("OMP_NUM_THREADS", str(_available_cpus()))   # the CPU oracle (OpenMP) must not oversubscribe the quota


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def small_scene():
    from tsar_mvs_amd import synth
    return synth.make_scene(96, 64, 3, seed=7)


@pytest.fixture(scope="session")
def mid_scene():
    from tsar_mvs_amd import synth
    return synth.make_scene(192, 128, 4, seed=11)
