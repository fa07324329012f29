"""pytest config: `gpu` marker = needs a real MI355X (run with -m gpu on the GPU box)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (MI355X); run with -m gpu")


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
        return cache[name]

    return load


@pytest.fixture(scope="session")
def synth_state():
    from ddpm_image_restoration_amd import synth
    cache = {}

    def get(arch, seed=0):
        if (arch, seed) not in cache:
            cache[(arch, seed)] = synth.synth_state_dict(arch, seed)
        return cache[(arch, seed)]

    return get
