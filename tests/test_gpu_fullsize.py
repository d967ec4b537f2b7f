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
from tests.conftest import assert_params_close, ROOT, global_opts, rel_err

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
    assert abs(rg["final_trust_region_radius"] - ro["final_trust_region_radius"]) <= 1e-7 * ro["final_trust_region_radius"]
    assert_params_close(pg, po)  # rvec, t, (fx fy cx cy), (k1 k2), (p1 p2), xi, points: each within 1e-6 of its own scale
    assert rel_err(eg, eo) < 1e-6


def test_c2_full_size_step_and_solve_match_oracle(mavba, fast_oracle):
    """BASELINE config C2 at FULL size (100 images / 30 000 points / 300 000 observations, PINHOLE only) - the one
    configuration whose dominant kernel is the factorisation, and the 4-parameter instantiation of the cluster kernel
    at benchmark size: S, v and the LM step at two radii, then one complete solve (iterations, termination, cost,
    every kind of parameter block, point errors) against the oracle."""
    p = synth.make_config("C2")
    assert (p.num_images, p.num_points, p.num_obs) == (100, 30000, 300000)
    with mavba.Session(p) as s:
        info = s.info()
        assert info["reduced_dim"] == 6 * 100 + 9 and info["clustered_points"] == p.num_points
        for radius in (1e4, 30.0):
            ref = fast_oracle.linear_step(p, radius, jac_mode=1)
            S, v = s.reduced_system(radius)
            st = s.linear_step(radius)
            _check_step(st, S, v, ref, ("C2", radius))
    po, pg = p.copy(), p.copy()
    ro, eo = fast_oracle.solve(po, fast_oracle.options(**global_opts()), jac_mode=1, want_point_errors=True)
    eg = np.full(p.num_points, np.nan)
    _, rg = mavba.bundle_adjustment(pg, global_opts(), point3D_errors=eg)
    assert rg["termination"] == ro["termination"], (rg["termination_name"], ro["termination_name"])
    assert rg["num_successful_steps"] == ro["num_successful_steps"]
    assert rg["num_unsuccessful_steps"] == ro["num_unsuccessful_steps"]
    assert abs(rg["initial_cost"] - ro["initial_cost"]) <= 1e-10 * ro["initial_cost"]
    assert abs(rg["final_cost"] - ro["final_cost"]) <= 1e-6 * ro["final_cost"]
    assert abs(rg["final_trust_region_radius"] - ro["final_trust_region_radius"]) <= 1e-7 * ro["final_trust_region_radius"]
    assert_params_close(pg, po)
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


@pytest.fixture(scope="module")
def c5_full():
    p = synth.make_config("C5")
    assert (p.num_images, p.num_points) == (2000, 1000000) and p.num_obs > 10_000_000
    assert len(p.rot_prior_image) > 1900
    return p


def test_c5_full_size_step_matches_oracle(mavba, fast_oracle, c5_full):
    """BASELINE.json's largest configuration at FULL size on one GPU (2000 images / 1 M points / 10.3 M observations,
    rotation priors, 5 % long loop-closure tracks): n = 12 018, the launch-per-panel factorisation, ~11 000 clusters AND
    generic term lists. One linear step against the oracle (block-sparse Schur complement + envelope Cholesky on all host
    cores - the same arithmetic as its dense path, tests/test_oracle.py)."""
    p = c5_full
    with mavba.Session(p) as s:
        info = s.info()
        assert info["reduced_dim"] == 6 * 2000 + 18
        assert 0 < info["clustered_points"] < p.num_points and info["schur_terms"][0] > 0
        for radius in (1e4, 50.0):  # the first iteration's radius and one after a few rejected steps
            with fast_oracle.linear_solver(fast_oracle.SPARSE):
                ref = fast_oracle.linear_step(p, radius, jac_mode=1)
            S, v = s.reduced_system(radius)
            st = s.linear_step(radius)
            _check_step(st, S, v, ref, ("C5", radius))


@pytest.mark.skipif(os.environ.get("MAVBA_SKIP_HEAVY") == "1", reason="MAVBA_SKIP_HEAVY=1 (tuning visits: ~7 minutes of oracle time on the host cores)")
def test_c5_full_size_solve_matches_oracle(mavba, fast_oracle, c5_full):
    """One COMPLETE full-size C5 solve with the reference's global-BA options (src/mapper.cc:170-174 through
    bundle_adjustment.cc:553-612) against the oracle's sparse mode on all host cores: rotation priors, generic term lists of
    the long tracks and the launch-per-panel factorisation of a 12 018-column system all the way to termination - same
    iteration counts, same termination, final cost / RMSE / every kind of parameter block / point errors within 1e-6."""
    p = c5_full
    po, pg = p.copy(), p.copy()
    with fast_oracle.linear_solver(fast_oracle.SPARSE):
        ro, eo = fast_oracle.solve(po, fast_oracle.options(**global_opts()), jac_mode=1, want_point_errors=True)
    eg = np.full(p.num_points, np.nan)
    _, rg = mavba.bundle_adjustment(pg, global_opts(), point3D_errors=eg)
    assert rg["termination"] == ro["termination"], (rg["termination_name"], ro["termination_name"])
    assert rg["num_successful_steps"] == ro["num_successful_steps"]
    assert rg["num_unsuccessful_steps"] == ro["num_unsuccessful_steps"]
    assert rg["num_residuals"] == ro["num_residuals"]
    assert abs(rg["initial_cost"] - ro["initial_cost"]) <= 1e-10 * ro["initial_cost"]
    assert abs(rg["final_cost"] - ro["final_cost"]) <= 1e-6 * ro["final_cost"]
    rmse_g = np.sqrt(rg["final_cost"] / rg["num_residuals"])
    rmse_o = np.sqrt(ro["final_cost"] / ro["num_residuals"])
    assert abs(rmse_g - rmse_o) <= 1e-6 * rmse_o
    assert_params_close(pg, po)
    assert rel_err(eg, eo) < 1e-6


def _solve_stepwise(s):
    """One LM iteration per call; returns (final result, cost after every iteration)."""
    costs = []
    while True:
        done, term = s.iterate(1)
        costs.append(s.result()["final_cost"])
        if term != A.TERM_RUNNING:
            return s.result(), costs
        assert len(costs) < 250


def test_c5_full_size_solve_properties_and_two_rank_shards(mavba, c5_full):
    """A complete full-size C5 solve: terminates by tolerance, the cost never rises, a repeat run is bit-identical, and the
    same problem sharded by point over two in-process ranks (packed tile exchange of a 12 018-column system, union
    envelope, max-reduced image graph) ends within 1e-8 of it."""
    from tests.test_gpu_sharded import solve_sharded
    p = c5_full
    opts = global_opts()
    with mavba.Session(p, opts) as s:
        r1, costs = _solve_stepwise(s)
        x1 = s.get_params()
        s.reset()
        r2 = s.solve()
        x2 = s.get_params()
    assert r1["termination"] in (A.TERM_FUNCTION_TOLERANCE, A.TERM_GRADIENT_TOLERANCE, A.TERM_PARAMETER_TOLERANCE), r1["termination_name"]
    assert r1["final_cost"] < 0.2 * r1["initial_cost"]
    assert all(b <= a for a, b in zip(costs, costs[1:])), costs
    rmse = np.sqrt(r1["final_cost"] / r1["num_residuals"])
    assert 0.2 < rmse < 0.6, rmse  # 0.5 px noise + 1 % gross outliers under the Cauchy loss
    for k in ("termination", "num_successful_steps", "num_unsuccessful_steps", "final_cost", "initial_cost"):
        assert r1[k] == r2[k], k
    for a, b in zip(x1, x2):
        assert np.array_equal(a, b)
    out, _ = solve_sharded(mavba, p, 2, opts)
    pts = np.zeros_like(p.points)
    for res, poses, intr, q, owned in out:
        assert res["termination"] == r1["termination"]
        assert res["num_successful_steps"] == r1["num_successful_steps"] and res["num_unsuccessful_steps"] == r1["num_unsuccessful_steps"]
        assert res["num_residuals"] == r1["num_residuals"] and res["num_parameters_reduced"] == r1["num_parameters_reduced"]
        assert abs(res["final_cost"] - r1["final_cost"]) <= 1e-8 * r1["final_cost"]
        assert np.array_equal(poses, out[0][1]) and np.array_equal(intr, out[0][2])
        pts[owned] = q
    assert_params_close(dict(poses=out[0][1], intrinsics=out[0][2], points=pts), dict(poses=x1[0], intrinsics=x1[1], points=x1[2]), tol=1e-8)


@pytest.mark.parametrize("world", [2, 8])
def test_c3_full_size_sharded_matches_single_rank(mavba, c3_full, world):
    """Configuration C4's structure at real size: full-size C3 sharded by point over 2 and 8 in-process ranks (19 MB of
    packed tiles per exchange, union envelope, rank-consistent dissection order) == the single-rank solve within 1e-8."""
    from tests.test_gpu_sharded import solve_sharded
    p = c3_full
    opts = global_opts()
    with mavba.Session(p, opts) as s:
        r1 = s.solve()
        x1 = s.get_params()
    out, ar = solve_sharded(mavba, p, world, opts)
    pts = np.zeros_like(p.points)
    for res, poses, intr, q, owned in out:
        assert res["termination"] == r1["termination"]
        assert res["num_successful_steps"] == r1["num_successful_steps"] and res["num_unsuccessful_steps"] == r1["num_unsuccessful_steps"]
        assert res["num_residuals"] == r1["num_residuals"] and res["num_parameters_reduced"] == r1["num_parameters_reduced"]
        assert abs(res["final_cost"] - r1["final_cost"]) <= 1e-8 * r1["final_cost"]
        assert np.array_equal(poses, out[0][1]) and np.array_equal(intr, out[0][2])
        pts[owned] = q
    assert_params_close(dict(poses=out[0][1], intrinsics=out[0][2], points=pts), dict(poses=x1[0], intrinsics=x1[1], points=x1[2]), tol=1e-8)


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
q2 = p.copy()
cost2, res2 = mavmap_amd.bundle_adjustment(q2, opts)  # (a second call of the process: in the cool-down it does not try the persistent launch again)
print("AGAIN", cost2, res2["num_successful_steps"], res2["termination"])
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
    assert outs[1].stderr.count("falling back") == 1 and "falling back" not in outs[0].stderr  # once: the next session skips the attempt
    for o in outs:
        a, b = o.stdout.split("SOLVE")[1].split()[:3], o.stdout.split("AGAIN")[1].split()[:3]
        assert a[1:] == b[1:] and abs(float(a[0]) - float(b[0])) < 1e-9 * float(a[0])
    d0, d1 = (float(o.stdout.split("DENSE")[1].split()[0]) for o in outs)
    assert d0 < 1e-10 and d1 < 1e-10
    s0, s1 = (o.stdout.split("SOLVE")[1].split()[:3] for o in outs)
    assert s0[1:] == s1[1:] and abs(float(s0[0]) - float(s1[0])) < 1e-9 * float(s0[0])
