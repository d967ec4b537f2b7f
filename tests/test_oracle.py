"""CPU tests of the oracle (oracle/ba_oracle.cpp): golden vectors, the reference's own test
properties, and independent maths (finite differences, scipy) for the un-pinned solver half."""
import json
import os

import numpy as np
import pytest

from mavmap_amd import _abi as A
from mavmap_amd import synth
from tests.conftest import global_opts, rel_err

HERE = os.path.dirname(os.path.abspath(__file__))
P4 = [651.123, 655.123, 386.123, 511.123]
P8 = P4 + [-0.471, 0.223, -0.001, 0.001]
# (model, params) exactly as instantiated by the reference's camera_models_test.cc:60-82
REFERENCE_TEST_SETS = [(1, P4), (3, P8 + [0.0]), (3, P8 + [1.0]), (3, P8 + [0.5]), (2, P8)]


def test_world2image_known_answers(oracle):
    """The reference-compiled world2image<double> (tests/golden/make_world2image_kat.py): the 10 vectors at the
    parameter sets of camera_models_test.cc:60-82 (the values SURVEY.md 8(c) records) and 300 seeded ones."""
    kat = json.load(open(os.path.join(HERE, "golden", "world2image_kat.json")))
    assert len(kat["vectors"]) == 10
    assert kat["vectors"][0]["uv"] == [711.68450000000007, 661.80128999999999]      # SURVEY.md 8(c), PINHOLE (0.5, 0.23, 1)
    assert kat["vectors"][4]["uv"] == [533.65136146442774, 579.33976752377532]      # CATA xi = 1
    jet = json.load(open(os.path.join(HERE, "golden", "world2image_jet_kat.json")))
    assert len(jet["vectors"]) == 310
    for c in kat["vectors"] + jet["vectors"]:
        u, v = oracle.world2image(c["code"], c["params"], *c["Xc"])
        assert abs(u - c["uv"][0]) <= 1e-12 * abs(c["uv"][0]), c
        assert abs(v - c["uv"][1]) <= 1e-12 * abs(c["uv"][1]), c


@pytest.mark.parametrize("mode", [0, 1])
def test_projection_jacobian_matches_reference_template_derivatives(oracle, mode):
    """d(u, v) / d(Xc, intrinsics) of the reference's OWN world2image<T> templates differentiated by forward-mode
    dual numbers (what ceres::AutoDiffCostFunction does with them; fixture world2image_jet_kat.json) against the oracle's
    Jets (mode 0) and its hand-derived analytic Jacobian (mode 1): with an identity pose the point block of the
    residual Jacobian IS d(u,v)/dXc and the intrinsics block IS d(u,v)/d kappa."""
    jet = json.load(open(os.path.join(HERE, "golden", "world2image_jet_kat.json")))
    pose = np.zeros(6)
    worst = 0.0
    for c in jet["vectors"]:
        K = A.MODEL_NUM_PARAMS[c["code"]]
        r, Jc, Jp, Jk = oracle.obs_jacobian(mode, c["code"], pose, np.array(c["Xc"]), c["params"], np.zeros(2))
        ref_p = np.array([c["du"][:3], c["dv"][:3]])
        ref_k = np.array([c["du"][3:], c["dv"][3:]])
        assert np.abs(r - np.array(c["uv"])).max() <= 1e-12 * np.abs(c["uv"]).max()
        for got, ref in ((Jp, ref_p), (Jk[:, :K], ref_k)):
            err = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-300)
            worst = max(worst, err)
            assert err <= 1e-11, (c["model"], err)
        assert not Jk[:, K:].any()
        # the translation block is the same matrix (Xc = R X + t)
        assert np.abs(Jc[:, 3:] - ref_p).max() <= 1e-11 * np.abs(ref_p).max()


@pytest.mark.parametrize("model,params", REFERENCE_TEST_SETS)
def test_camera_model_properties_of_reference_test(oracle, model, params):
    """The assertions of the reference's src/base3d/camera_models_test.cc:16-55, same tolerances."""
    x0, y0, z0 = 0.5, 0.23, 1.0
    u, v = oracle.world2image(model, params, x0, y0, z0)
    x, y, z = oracle.image2world(model, params, u, v)
    assert abs(x / z - x0) < 1e-5 and abs(y / z - y0) < 1e-5
    x, y, z = oracle.image2world(model, params, 200.0, 100.0)
    u, v = oracle.world2image(model, params, x, y, z)
    assert abs(u - 200.0) < 1e-1 and abs(v - 100.0) < 1e-1
    u, v = oracle.world2image(model, params, 0.0, 0.0, 1.0)
    assert abs(u - params[2]) < 1e-6 and abs(v - params[3]) < 1e-6
    x, y, z = oracle.image2world(model, params, params[2], params[3])
    assert abs(x / z) < 1e-4 and abs(y / z) < 1e-4


def _random_obs(rng, model, scale):
    K = A.MODEL_NUM_PARAMS[model]
    from scipy.spatial.transform import Rotation
    pose = np.concatenate([rng.normal(0, 1, 3) * scale, rng.normal(0, 1, 3)])
    # a point in FRONT of the camera (camera frame z in [3, 9]), expressed in world coordinates
    Xc = np.array([rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(3, 9)])
    X = Rotation.from_rotvec(pose[:3]).as_matrix().T @ (Xc - pose[3:])
    cam = np.zeros(9)
    cam[:4] = [600 + rng.normal(), 610 + rng.normal(), 376, 240]
    if K >= 8:
        cam[4:8] = np.array([-0.1, 0.02, 1e-3, -1e-3]) * (1 + 0.1 * rng.normal(size=4))
    if K == 9:
        cam[8] = rng.uniform(0, 1)
    return pose, X, cam, rng.normal(300, 50, 2)


@pytest.mark.parametrize("model", [1, 2, 3])
def test_jets_analytic_and_finite_differences_agree(oracle, model):
    rng = np.random.default_rng(100 + model)
    K = A.MODEL_NUM_PARAMS[model]
    for it in range(300):
        scale = [1e-3, 0.3, 1.5, 3.1][it % 4]
        pose, X, cam, uv = _random_obs(rng, model, scale)
        if it % 37 == 0:
            pose[:3] = 0.0  # the reference's first image: Taylor branch of AngleAxisRotatePoint
        r0, Jc0, Jp0, Jk0 = oracle.obs_jacobian(0, model, pose, X, cam, uv)
        r1, Jc1, Jp1, Jk1 = oracle.obs_jacobian(1, model, pose, X, cam, uv)
        for a, b in ((r0, r1), (Jc0, Jc1), (Jp0, Jp1), (Jk0, Jk1)):
            assert np.abs(a - b).max() <= 1e-10 * max(1.0, np.abs(a).max())
        assert np.all(Jk0[:, K:] == 0.0)
        # central differences (relative step) on every parameter
        def res(pose_, X_, cam_):
            return oracle.obs_jacobian(1, model, pose_, X_, cam_, uv)[0]
        for blk, J, n in (("pose", Jc0, 6), ("X", Jp0, 3), ("cam", Jk0, K)):
            for e in range(n):
                base = {"pose": pose, "X": X, "cam": cam}[blk]
                h = 1e-6 * max(1.0, abs(base[e]))
                hi, lo = base.copy(), base.copy()
                hi[e] += h; lo[e] -= h
                args = {"pose": (hi, X, cam), "X": (pose, hi, cam), "cam": (pose, X, hi)}[blk]
                args2 = {"pose": (lo, X, cam), "X": (pose, lo, cam), "cam": (pose, X, lo)}[blk]
                fd = (res(*args) - res(*args2)) / (2 * h)
                assert np.abs(fd - J[:, e]).max() <= 2e-5 * max(1.0, np.abs(J[:, e]).max()), (blk, e, it)


def test_rotation_matches_scipy(oracle):
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(5)
    for it in range(200):
        w = rng.normal(0, 1, 3) * [1e-9, 1e-3, 1.0, 3.0][it % 4]
        X = rng.normal(0, 2, 3)
        R = Rotation.from_rotvec(w).as_matrix()
        assert np.abs(oracle.rotate_point(w, X) - R @ X).max() < 1e-13 * max(1, np.abs(X).max())
        assert np.abs(oracle.rotation_matrix(w) - R).max() < 1e-13
    assert np.array_equal(oracle.rotate_point(np.zeros(3), np.array([1.0, 2, 3])), [1.0, 2, 3])


def test_rotation_prior_restates_reference_formula(oracle):
    """BARotationConstraintCostFunction (bundle_adjustment.cc:72-111) including its (6,7) index pair."""
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(6)
    for it in range(100):
        w, w0 = rng.normal(0, 1, 3), rng.normal(0, 1, 3)
        Rcm = Rotation.from_rotvec(w).as_matrix().T.ravel()    # column-major R[0..8]
        R0cm = Rotation.from_rotvec(w0).as_matrix().T.ravel()
        pairs = [(0, 0), (3, 1), (6, 2), (1, 3), (4, 4), (7, 5), (2, 6), (6, 7), (8, 8)]
        expect = 1.7 * np.sqrt(sum((Rcm[a] - R0cm[b]) ** 2 for a, b in pairs))
        res, jac = oracle.rot_prior(w, w0, 1.7)
        assert abs(res - expect) < 1e-12 * max(1, expect)
        for e in range(3):
            h = 1e-6
            hi, lo = w.copy(), w.copy()
            hi[e] += h; lo[e] -= h
            fd = (oracle.rot_prior(hi, w0, 1.7)[0] - oracle.rot_prior(lo, w0, 1.7)[0]) / (2 * h)
            assert abs(fd - jac[e]) < 1e-6 * max(1, abs(fd))


def test_loss_correction_is_cauchy_with_sqrt_rho_prime(oracle):
    p = synth.make_scene(num_images=4, num_points=60, track_len=3, models=[A.MODEL_PINHOLE], seed=3)
    for a in (1.0, 2.5):
        cost, r, Jc, Jp, Jk = oracle.eval_jacobian(p, oracle.options(loss_scale_factor=a), jac_mode=1)
        raw = np.array([oracle.obs_jacobian(1, 1, p.poses[i], p.points[j], p.intrinsics[0], uv)[0]
                        for i, j, uv in zip(p.obs_image, p.obs_point, p.obs_uv)])
        s = (raw ** 2).sum(1)
        assert abs(cost - 0.5 * np.sum(a * a * np.log1p(s / (a * a)))) < 1e-10 * cost
        w = 1 / np.sqrt(1 + s / (a * a))
        assert rel_err(r, raw * w[:, None]) < 1e-13


def _scipy_reference_minimum(p, a=1.0):
    """Independent minimiser of the same robust cost: scipy least_squares(loss='cauchy')."""
    from scipy.optimize import least_squares
    from scipy.sparse import lil_matrix
    from tests import oracle_lib as O
    NI, NP = p.num_images, p.num_points
    K = [A.MODEL_NUM_PARAMS[int(m)] for m in p.camera_model]
    free = []  # (kind, index, element)
    for i in range(NI):
        for e in range(6):
            const = (p.pose_const[i] & (1 if e < 3 else (2 << (e - 3)))) != 0
            if not const:
                free.append(("pose", i, e))
    for c in range(p.num_cameras):
        if not p.intr_const[c]:
            free += [("intr", c, e) for e in range(K[c])]
    for j in range(NP):
        if not p.point_const[j]:
            free += [("pt", j, e) for e in range(3)]
    arrays = {"pose": p.poses.copy(), "intr": p.intrinsics.copy(), "pt": p.points.copy()}

    def unpack(x):
        arr = {k: v.copy() for k, v in arrays.items()}
        for val, (k, i, e) in zip(x, free):
            arr[k][i, e] = val
        return arr

    def fun(x):
        arr = unpack(x)
        R = synth.rodrigues(arr["pose"][:, :3])
        Xc = np.einsum("nij,nj->ni", R[p.obs_image], arr["pt"][p.obs_point]) + arr["pose"][p.obs_image, 3:]
        uv = np.zeros((p.num_obs, 2))
        for c in range(p.num_cameras):
            sel = p.image_camera[p.obs_image] == c
            uv[sel] = synth.project(int(p.camera_model[c]), arr["intr"][c], Xc[sel])
        d = uv - p.obs_uv
        # scipy's robust loss acts per scalar residual; the BA loss acts per 2-D block. Feeding the
        # block norm as ONE residual makes rho(|r_block|^2) identical to ceres'.
        return np.sqrt((d ** 2).sum(1))

    x0 = np.array([arrays[k][i, e] for k, i, e in free])
    sol = least_squares(fun, x0, loss="cauchy", f_scale=a, method="trf", x_scale="jac", xtol=1e-15, ftol=1e-15,
                        gtol=1e-12, max_nfev=400)
    return sol.cost, unpack(sol.x)


def test_solver_reaches_the_same_minimum_as_scipy(oracle):
    """The solver half is parity-unpinned by the reference; pin the MINIMUM it finds independently."""
    p = synth.make_scene(num_images=4, num_points=40, track_len=3, models=[A.MODEL_PINHOLE], seed=8,
                         refine_camera_params=False)
    q = p.copy()
    res, _ = oracle.solve(q, oracle.options(**global_opts(function_tolerance=1e-14, gradient_tolerance=1e-14,
                                                          parameter_tolerance=1e-14)))
    # the robust cost is not convex: start scipy AT the oracle's answer and require that it is a
    # minimum of scipy's (independently coded) objective too: same cost, no movement.
    cost_sp, arr = _scipy_reference_minimum(q)
    assert abs(res["final_cost"] - cost_sp) <= 1e-9 * cost_sp
    assert rel_err(q.poses, arr["pose"]) < 1e-6 and rel_err(q.points, arr["pt"]) < 1e-6
    # and from the common start scipy must not find anything better than the oracle did by more
    # than a different local basin would explain (sanity: same order of magnitude)
    cost_sp0, _ = _scipy_reference_minimum(p)
    assert res["final_cost"] <= cost_sp0 * (1 + 1e-6) or abs(res["final_cost"] - cost_sp0) < 0.1 * cost_sp0


def test_ground_truth_recovery_on_clean_data(oracle):
    p = synth.make_scene(num_images=6, num_points=150, track_len=4, models=[A.MODEL_PINHOLE, A.MODEL_OPENCV],
                         seed=9, noise_px=0.0, outlier_frac=0.0)
    q = p.copy()
    res, _ = oracle.solve(q, oracle.options(**global_opts(function_tolerance=1e-16, gradient_tolerance=1e-16)))
    assert res["final_cost"] < 1e-12 * res["initial_cost"]
    cost, r, *_ = oracle.eval_jacobian(q, jac_mode=1)
    assert np.abs(r).max() < 1e-6


def test_problem_reduction_and_counts(oracle):
    p = synth.make_scene(num_images=6, num_points=200, track_len=3, models=[A.MODEL_PINHOLE], seed=41,
                         refine_camera_params=False)
    seen0 = np.unique(p.obs_point[p.obs_image == 0])
    p.point_const[seen0[:20]] = 1
    res, _ = oracle.solve(p.copy(), oracle.options(max_num_iterations=0))
    n_fixed_blocks = int(np.sum((p.obs_image == 0) & np.isin(p.obs_point, seen0[:20])))
    assert res["termination"] == A.TERM_NO_CONVERGENCE
    assert res["num_residuals"] == 2 * p.num_obs
    assert res["num_residuals_reduced"] == 2 * (p.num_obs - n_fixed_blocks)
    # image 0 fixed (0 params), image 1 FIXED_X (5), others 6; 20 constant points
    assert res["num_parameters_reduced"] == 5 + 6 * 4 + 3 * (p.num_points - 20)
    assert res["fixed_cost"] > 0 and res["initial_cost"] == res["final_cost"]


def test_point_errors_follow_reference_definition(oracle):
    p = synth.make_scene(num_images=5, num_points=80, track_len=3, models=[A.MODEL_PINHOLE], seed=12)
    q = p.copy()
    _, perr = oracle.solve(q, oracle.options(**global_opts()), want_point_errors=True)
    # bundle_adjustment.cc:590-596: sum over the point's residual blocks of |r_raw| / (#observations)
    R = synth.rodrigues(q.poses[:, :3])
    Xc = np.einsum("nij,nj->ni", R[q.obs_image], q.points[q.obs_point]) + q.poses[q.obs_image, 3:]
    uv = synth.project(1, q.intrinsics[0], Xc)
    e = np.sqrt(((uv - q.obs_uv) ** 2).sum(1))
    expect = np.bincount(q.obs_point, e, q.num_points) / np.bincount(q.obs_point, minlength=q.num_points)
    assert rel_err(perr, expect) < 1e-12


@pytest.mark.parametrize("kind", ["mixed_priors_long_tracks", "cata_constant_blocks"])
def test_sparse_linear_solver_mode_equals_the_dense_one(oracle, kind):
    """Linear-solver mode 1 (block-sparse Schur complement with one owner thread per row block + envelope Cholesky) is
    the arithmetic of the dense checker path in another order: same S, v, step and the same complete solve. It is what
    bench.py's cpu_baseline times and what checks the full-size C5 step, where the dense path would take minutes."""
    from mavmap_amd import _abi as A, synth
    if kind == "mixed_priors_long_tracks":
        p = synth.make_scene(num_images=40, num_points=3000, track_len=5, models=[A.MODEL_PINHOLE, A.MODEL_OPENCV], seed=3,
                             rot_priors=True, long_track_frac=0.02, long_track_len=12)
    else:
        p = synth.make_scene(num_images=12, num_points=500, track_len=4, models=[A.MODEL_CATA], seed=5)
        p.pose_const[3] = 15
        p.point_const[::7] = 1
    results = {}
    for threads in (1, 3):
        oracle.set_threads(threads)
        dense = oracle.linear_step(p, 100.0, jac_mode=1)
        with oracle.linear_solver(oracle.SPARSE):
            sparse = oracle.linear_step(p, 100.0, jac_mode=1)
            q = p.copy()
            rs, _ = oracle.solve(q, oracle.options(**global_opts()), jac_mode=1)
        for k in ("S", "v"):
            assert rel_err(sparse[k], dense[k]) < 1e-13, k
        for k in ("d_poses", "d_intr", "d_points"):
            assert rel_err(sparse[k], dense[k]) < 1e-10, k
        assert abs(sparse["model_cost_change"] - dense["model_cost_change"]) < 1e-10 * abs(dense["model_cost_change"])
        results[threads] = (sparse, rs, q)
    oracle.set_threads(1)
    qd = p.copy()
    rd, _ = oracle.solve(qd, oracle.options(**global_opts()), jac_mode=1)
    for threads, (sparse, rs, q) in results.items():
        assert rs["termination"] == rd["termination"] and rs["num_successful_steps"] == rd["num_successful_steps"]
        assert abs(rs["final_cost"] - rd["final_cost"]) < 1e-9 * rd["final_cost"]
        assert rel_err(q.poses, qd.poses) < 1e-7
    # every row block has one owner and chunk partials are added in chunk order: no dependence on the thread count
    assert np.array_equal(results[1][0]["S"], results[3][0]["S"]) and np.array_equal(results[1][0]["d_points"], results[3][0]["d_points"])


def test_dense_cholesky_of_oracle(oracle):
    rng = np.random.default_rng(2)
    for n in (1, 7, 64, 130):
        B = rng.normal(size=(n, n))
        M = B @ B.T + n * np.eye(n)
        b = rng.normal(size=n)
        rc, x = oracle.dense_spd_solve(M, b)
        assert rc == 0 and rel_err(x, np.linalg.solve(M, b)) < 1e-11
    rc, _ = oracle.dense_spd_solve(-np.eye(3), np.ones(3))
    assert rc != 0


def test_regression_fixture(oracle):
    """Frozen oracle output (tests/golden/oracle_solve_regression.json, written by
    tests/golden/make_regression_fixture.py): guards the checker itself against silent edits."""
    fx = json.load(open(os.path.join(HERE, "golden", "oracle_solve_regression.json")))
    p = synth.make_scene(**fx["scene"])
    res, _ = oracle.solve(p, oracle.options(**fx["options"]))
    assert res["num_successful_steps"] == fx["num_successful_steps"]
    assert res["num_unsuccessful_steps"] == fx["num_unsuccessful_steps"]
    assert abs(res["final_cost"] - fx["final_cost"]) <= 1e-9 * fx["final_cost"]
    assert rel_err(p.poses, np.array(fx["poses"])) < 1e-8
