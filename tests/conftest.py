import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_lib
    oracle_lib.lib()
    oracle_lib.set_threads(1)  # deterministic summation order for the checker
    return oracle_lib


@pytest.fixture(scope="session")
def mavba():
    """The product library, built on demand (hipcc cross-compiles without a GPU)."""
    from mavmap_amd import build
    build.build()
    import mavmap_amd
    mavmap_amd.load()
    return mavmap_amd


def global_opts(**kw):
    """The option values the reference forces for global BA (src/mapper.cc:170-174)."""
    d = dict(max_num_iterations=200, function_tolerance=1e-6, gradient_tolerance=1e-10)
    d.update(kw)
    return d


def rel_err(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    den = max(float(np.abs(b).max()) if b.size else 0.0, 1e-300)
    return float(np.abs(a - b).max() / den) if a.size else 0.0
