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


# ---- per-block-kind parameter comparison --------------------------------------------------------------------------
# rel_err above divides by the largest entry of the WHOLE array: for `intrinsics` that is fx ~ 600, so 1e-6 would admit
# a 60 % error on p1 ~ 1e-3; for `poses` the translations (tens to hundreds of units) set the scale for rvec ~ 0.05.
# The north star asks for "camera/point parameters within 1e-6 relative": every KIND of parameter block is compared
# against its own scale (the largest reference magnitude of that kind in the problem).
PARAM_KINDS = (("rvec", "poses", slice(0, 3)), ("t", "poses", slice(3, 6)),
               ("fxfycxcy", "intrinsics", slice(0, 4)), ("k1k2", "intrinsics", slice(4, 6)),
               ("p1p2", "intrinsics", slice(6, 8)), ("xi", "intrinsics", slice(8, 9)),
               ("points", "points", slice(0, 3)))


def param_errors(got, ref):
    """got / ref: dicts (or BAProblem-like objects) with poses [NI,6], intrinsics [NC,9], points [NP,3]; any of them
    may be missing / None. Returns {kind: max|got - ref| / max|ref of that kind|} (a kind whose reference is all zero -
    the padding of a smaller camera model - must match exactly: its entry is inf otherwise)."""
    def arr(o, name):
        a = o.get(name) if isinstance(o, dict) else getattr(o, name, None)
        return None if a is None else np.asarray(a, float)
    out = {}
    for kind, name, cols in PARAM_KINDS:
        a, b = arr(got, name), arr(ref, name)
        if a is None or b is None or b.size == 0:
            continue
        a, b = a.reshape(b.shape)[:, cols], b[:, cols]
        if a.size == 0:
            continue
        err, scale = float(np.abs(a - b).max()), float(np.abs(b).max())
        out[kind] = err / scale if scale > 0.0 else (0.0 if err == 0.0 else float("inf"))
    return out


def assert_params_close(got, ref, tol=1e-6, what=""):
    errs = param_errors(got, ref)
    bad = {k: v for k, v in errs.items() if not v < tol}
    assert not bad, (what, "per-kind relative errors", errs, "tolerance", tol)
    return errs
