"""GPU parity at the sizes the benchmark runs (BASELINE.json configs C3 and C5-shaped), against the CPU oracle.

The small-scene tests never reach the code paths the full-size configurations take: the separate panel solve
+ lean trailing update of the factorisation (steps with more than 24 row blocks / 256 tile updates), the
depth-2 elimination tree with four concurrent fronts and shadow merges, the pre-reduction of long partial
runs, ~1800 point clusters. These tests do, with the oracle on all host cores (its dense Schur complement
and Cholesky take a few seconds per linear step at n = 3018).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from mavmap_amd import _abi as A
from mavmap_amd import synth
from tests.conftest import ROOT, global_opts, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture()
def fast_oracle(oracle):
    """The oracle on every host core (summation order then depends on the core count: fine at 1e-8)."""
    oracle.set_threads(oracle.max_threads())
    yield oracle
    oracle.set_threads(1)


@pytest.fixture(scope="module")
def c3_full():
    p = synth.make_config("C3")
    assert (p.num_images, p.num_points, p.num_obs) == (500, 200000, 2000000)
    return p


def _check_step(st, S, v, ref, tag):
    assert rel_err(S, ref["S"]) < 1e-9, tag
    assert np.abs(S - S.T).max() == 0.0
    assert rel_err(v, ref["v"]) < 1e-9, tag
    for k in ("d_poses", "d_intr", "d_points"):
        assert rel_err(st[k], ref[k]) < 1e-8, (tag, k, rel_err(st[k], ref[k]))
    assert abs(st["model_cost_change"] - ref["model_cost_change"]) < 1e-9 * abs(ref["model_cost_change"]), tag


def test_c3_full_size_reduced_system_and_step_match_oracle(mavba, fast_oracle, c3_full):
    """The headline configuration at full size: S, v and the LM step of the first iteration."""
    p = c3_full
    with mavba.Session(p) as s:
        info = s.info()
        # the structure the benchmark reports: depth-2 elimination tree, clusters, pre-reduced partial runs
        assert info["reduced_dim"] == 6 * 500 + 18
        assert info["nd_parts"] >= 2 and info["chain_steps"] < info["matrix_dim"] // 64
        assert info["num_clusters"] > 500 and info["clustered_points"] == p.num_points
        for radius in (1e4, 30.0):
            ref = fast_oracle.linear_step(p, radius, jac_mode=1)
            S, v = s.reduced_system(radius)
            st = s.linear_step(radius)
            _check_step(st, S, v, ref, ("C3", radius))


def test_c3_full_size_solve_matches_oracle(mavba, fast_oracle, c3_full):
    """One complete C3 solve with the reference's global-BA options: same iterations, same termination,
    final cost / RMSE / cameras / points / intrinsics / point errors within 1e-6 relative."""
    p = c3_full
    po, pg = p.copy(), p.copy()
    ro, eo = fast_oracle.solve(po, fast_oracle.options(**global_opts()), jac_mode=1, want_point_errors=True)
    eg = np.full(p.num_points, np.nan)
    _, rg = mavba.bundle_adjustment(pg, global_opts(), point3D_errors=eg)
    assert rg["termination"] == ro["termination"], (rg["termination_name"], ro["termination_name"])
    assert rg["num_successful_steps"] == ro["num_successful_steps"]
    assert rg["num_unsuccessful_steps"] == ro["num_unsuccessful_steps"]
    assert abs(rg["initial_cost"] - ro["initial_cost"]) <= 1e-10 * ro["initial_cost"]
    assert abs(rg["final_cost"] - ro["final_cost"]) <= 1e-6 * ro["final_cost"]
    rmse_g = np.sqrt(rg["final_cost"] / rg["num_residuals"])
    rmse_o = np.sqrt(ro["final_cost"] / ro["num_residuals"])
    assert abs(rmse_g - rmse_o) <= 1e-6 * rmse_o
    assert rel_err(pg.poses, po.poses) < 1e-6
    assert rel_err(pg.points, po.points) < 1e-6
    assert rel_err(pg.intrinsics, po.intrinsics) < 1e-6
    assert rel_err(eg, eo) < 1e-6


def test_c5_shaped_step_matches_oracle(mavba, fast_oracle):
    """Config C5 in shape (rotation priors, 5 % long loop-closure tracks that the clusters cannot take, mixed
    models) at a size where the elimination tree has depth 2: four concurrent fronts, two levels of shadow
    merges, pre-reduced partial runs AND generic term lists in the same system."""
    p = synth.make_config("C5", scale=0.2)
    assert p.num_images == 400
    with mavba.Session(p) as s:
        info = s.info()
        assert info["nd_parts"] >= 4, info            # two levels of bisection
        assert info["num_clusters"] > 64              # -> the intrinsics blocks' partial runs are pre-reduced
        assert 0 < info["clustered_points"] < p.num_points and info["schur_terms"][0] > 0
        for radius in (1e4, 100.0):
            ref = fast_oracle.linear_step(p, radius, jac_mode=1)
            S, v = s.reduced_system(radius)
            st = s.linear_step(radius)
            _check_step(st, S, v, ref, ("C5x0.2", radius))


def _spd(n, seed):
    rng = np.random.default_rng(seed)
    B = rng.normal(size=(n, n))
    Amat = B @ B.T + n * np.eye(n)
    Amat[np.arange(n), np.arange(n)] += np.arange(n) * 0.5
    return Amat, rng.normal(size=n)


@pytest.mark.parametrize("n", [1600, 3018, 3200])
def test_dense_spd_solve_large(mavba, n):
    """n / 64 = 25..50 row blocks: every panel step of the dense schedule beyond the last 24 takes the separate
    panel-solve launch + the lean trailing update (the branch the C3 factorisation takes in most of its steps)."""
    Amat, b = _spd(n, n)
    x = mavba.dense_spd_solve(Amat, b)
    x0 = np.linalg.solve(Amat, b)
    assert rel_err(x, x0) < 1e-10


_FUSE_SNIPPET = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
import mavmap_amd
from tests.test_gpu_fullsize import _spd
worst = 0.0
for n in (130, 700, 1700):
    A, b = _spd(n, n)
    x = mavmap_amd.dense_spd_solve(A, b)
    x0 = np.linalg.solve(A, b)
    worst = max(worst, float(np.abs(x - x0).max() / np.abs(x0).max()))
print("WORST", worst)
"""


@pytest.mark.parametrize("fuse", ["0", "1000"])
def test_dense_spd_solve_with_the_other_panel_schedule(mavba, fuse):
    """MAVBA_CHOL_FUSE=0: every step is panel solve + lean update; =1000 (with the task limit lifted): every step is
    the fused launch. Both schedules must give numpy's solution (the knob is read once per process)."""
    env = dict(os.environ, MAVBA_CHOL_FUSE=fuse, MAVBA_CHOL_FUSE_TASKS="0" if fuse == "0" else "100000000")
    out = subprocess.run([sys.executable, "-c", _FUSE_SNIPPET.format(root=ROOT)], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    worst = float(out.stdout.split("WORST")[1])
    assert worst < 1e-10, worst


_ABORT_SNIPPET = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
import mavmap_amd
from mavmap_amd import synth
from tests.test_gpu_fullsize import _spd
A, b = _spd(700, 7)
x = mavmap_amd.dense_spd_solve(A, b)
x0 = np.linalg.solve(A, b)
print("DENSE", float(np.abs(x - x0).max() / np.abs(x0).max()))
p = synth.make_scene(num_images=130, num_points=5000, track_len=4, models=[1, 2], seed=5, long_track_frac=0.01, long_track_len=12, spacing=6.0)
opts = dict(max_num_iterations=200, function_tolerance=1e-6, gradient_tolerance=1e-10)
q = p.copy()
cost, res = mavmap_amd.bundle_adjustment(q, opts)
print("SOLVE", cost, res["num_successful_steps"], res["termination"])
"""


def test_persistent_factorisation_gives_up_cleanly_and_the_solve_falls_back(mavba):
    """A work-group of the persistent launch that never runs (here: told to return at once; in the field: CU masking or a
    second persistent launch competing for the CUs) must not hang the device: the waits time out, the launch reports it,
    and the solve is repeated with the launch-per-panel schedule - same answers as an undisturbed process."""
    outs = []
    for drop in (None, "5"):
        env = dict(os.environ)
        env.pop("MAVBA_CHOL_TEST_DROP_WG", None)
        if drop:
            env["MAVBA_CHOL_TEST_DROP_WG"] = drop
        out = subprocess.run([sys.executable, "-c", _ABORT_SNIPPET.format(root=ROOT)], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        outs.append(out)
    assert "falling back" in outs[1].stderr and "falling back" not in outs[0].stderr
    d0, d1 = (float(o.stdout.split("DENSE")[1].split()[0]) for o in outs)
    assert d0 < 1e-10 and d1 < 1e-10
    s0, s1 = (o.stdout.split("SOLVE")[1].split() for o in outs)
    assert s0[1:] == s1[1:] and abs(float(s0[0]) - float(s1[0])) < 1e-9 * float(s0[0])
