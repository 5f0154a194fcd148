import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "golden.npz"))


@pytest.fixture(scope="session")
def meta():
    with open(os.path.join(ROOT, "tests", "golden", "golden_meta.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def sup_state():
    """Deterministic synthetic supernet weights (the ones the golden vectors were generated with)."""
    from sugar_amd import layout, synth
    return synth.fill_state_dict(layout.supernet_shapes(), seed=0)


@pytest.fixture(scope="session")
def supernet(sup_state):
    import sugar_amd
    sup = sugar_amd.tdnn8m2g()
    sup.load_state_dict(sup_state)
    return sup


def cfg_of(meta, name):
    c = meta["configs"][name]
    return (c[0], list(c[1]), list(c[2]), c[3])
