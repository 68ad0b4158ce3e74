import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


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
