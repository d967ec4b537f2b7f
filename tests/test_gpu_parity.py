"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Tolerances (north star: final RMSE / cameras / points within 1e-6 relative of the CPU path):
  single evaluation (residuals, Jacobians, cost)      1e-11 relative
  reduced camera system, LM step                      1e-9  relative
  full solve: final cost, parameters                  1e-6  relative
"""
import numpy as np
import pytest

from mavmap_amd import _abi as A
from mavmap_amd import synth
from tests.conftest import assert_params_close, global_opts, rel_err

pytestmark = pytest.mark.gpu


def _scene(kind):
    if kind == "pinhole":
        return synth.make_config("C2", scale=0.03, seed=11)
    if kind == "mixed":
        return synth.make_config("C3", scale=0.01, seed=12)
    if kind == "cata":
        return synth.make_scene(num_images=6, num_points=400, track_len=4,
                                models=[A.MODEL_CATA, A.MODEL_OPENCV, A.MODEL_PINHOLE], seed=13)
    if kind == "priors":
        return synth.make_scene(num_images=8, num_points=500, track_len=4,
                                models=[A.MODEL_PINHOLE], seed=14, rot_priors=True)
    if kind == "fixed_intr":
        return synth.make_scene(num_images=8, num_points=500, track_len=4,
                                models=[A.MODEL_OPENCV], seed=15, refine_camera_params=False)
    if kind == "gcp":
        p = synth.make_scene(num_images=8, num_points=500, track_len=4, models=[A.MODEL_PINHOLE], seed=16)
        p.point_const[::7] = 1
        p.points[::7] = p.truth["points"][::7]
        return p
    if kind == "long_tracks":
        # BASELINE config C5 in miniature: mixed models, rotation priors, 5 % long loop-closure tracks
        # (track length 36 > 2 x 16: exercises the strided loops of the 16-lane point reductions)
        return synth.make_scene(num_images=40, num_points=1500, track_len=10, models=[A.MODEL_PINHOLE, A.MODEL_OPENCV],
                                seed=17, rot_priors=True, long_track_frac=0.05, long_track_len=36, spacing=5.0)
    raise KeyError(kind)


KINDS = ["pinhole", "mixed", "cata", "priors", "fixed_intr", "gcp", "long_tracks"]


@pytest.mark.parametrize("kind", KINDS)
def test_jacobian_sweep_matches_oracle(mavba, oracle, kind):
    p = _scene(kind)
    c0, r0, Jc0, Jp0, Jk0 = oracle.eval_jacobian(p, jac_mode=0)
    with mavba.Session(p) as s:
        c, r, Jc, Jp, Jk = s.eval_jacobian()
    assert abs(c - c0) <= 1e-11 * abs(c0)
    for name, a, b in (("r", r, r0), ("Jc", Jc, Jc0), ("Jp", Jp, Jp0), ("Jk", Jk, Jk0)):
        assert rel_err(a, b) < 1e-11, (kind, name, rel_err(a, b))


@pytest.mark.parametrize("kind", KINDS)
def test_reduced_system_and_step_match_oracle(mavba, oracle, kind):
    p = _scene(kind)
    for radius in (1e4, 3.0):
        ref = oracle.linear_step(p, radius)
        with mavba.Session(p) as s:
            S, v = s.reduced_system(radius)
            st = s.linear_step(radius)
        assert rel_err(S, ref["S"]) < 1e-9, (kind, radius)
        assert np.abs(S - S.T).max() == 0.0
        assert rel_err(v, ref["v"]) < 1e-9
        assert rel_err(st["d_poses"], ref["d_poses"]) < 1e-8
        assert rel_err(st["d_intr"], ref["d_intr"]) < 1e-8
        assert rel_err(st["d_points"], ref["d_points"]) < 1e-8
        assert abs(st["model_cost_change"] - ref["model_cost_change"]) < 1e-9 * abs(ref["model_cost_change"])


def _solve_both(mavba, oracle, p, **optkw):
    po, pg = p.copy(), p.copy()
    ro, eo = oracle.solve(po, oracle.options(**optkw), want_point_errors=True)
    eg = np.full(p.num_points, np.nan)
    _, rg = mavba.bundle_adjustment(pg, dict(optkw), point3D_errors=eg)
    return po, ro, eo, pg, rg, eg


@pytest.mark.parametrize("kind", KINDS)
def test_full_solve_matches_oracle(mavba, oracle, kind):
    p = _scene(kind)
    po, ro, eo, pg, rg, eg = _solve_both(mavba, oracle, p, **global_opts())
    assert rg["termination"] == ro["termination"]
    assert rg["num_successful_steps"] == ro["num_successful_steps"]
    assert rg["num_unsuccessful_steps"] == ro["num_unsuccessful_steps"]
    for k in ("num_residuals", "num_residuals_reduced", "num_parameters_reduced"):
        assert rg[k] == ro[k], k
    assert abs(rg["initial_cost"] - ro["initial_cost"]) <= 1e-10 * ro["initial_cost"]
    assert abs(rg["final_cost"] - ro["final_cost"]) <= 1e-6 * ro["final_cost"]
    rmse_g = np.sqrt(rg["final_cost"] / rg["num_residuals"])
    rmse_o = np.sqrt(ro["final_cost"] / ro["num_residuals"])
    assert abs(rmse_g - rmse_o) <= 1e-6 * rmse_o
    assert_params_close(pg, po)  # rvec, t, (fx fy cx cy), (k1 k2), (p1 p2), xi, points: each within 1e-6 of its own scale
    m = ~np.isnan(eo)
    assert np.array_equal(m, ~np.isnan(eg))
    assert rel_err(eg[m], eo[m]) < 1e-6
    # constant blocks really stay put
    assert np.array_equal(pg.poses[0], p.poses[0])
    assert pg.poses[1, 3] == p.poses[1, 3]
    if kind == "gcp":
        assert np.array_equal(pg.points[::7], p.points[::7])
    if kind == "fixed_intr":
        assert np.array_equal(pg.intrinsics, p.intrinsics)


def test_local_ba_windows_c1(mavba, oracle):
    """Config C1: the reference's local-BA window [FIXED, FIXED, FREE x 6] slid over 10 images."""
    g = synth.make_config("C1")
    for first in range(0, g.num_images - 7):
        p = synth.local_ba_window(g, first)
        p.intr_const[:] = 0  # mapper.cc:882-886: refine_camera_params defaults to true for local BA
        po, ro, _, pg, rg, _ = _solve_both(mavba, oracle, p)  # BundleAdjustmentOptions defaults
        assert rg["termination"] == ro["termination"]
        assert abs(rg["final_cost"] - ro["final_cost"]) <= 1e-6 * ro["final_cost"]
        assert_params_close(pg, po)
        # images outside the window are not part of the problem and must not move
        out = np.setdiff1d(np.arange(g.num_images), np.arange(first, first + 8))
        assert np.array_equal(pg.poses[out], p.poses[out])


@pytest.mark.parametrize("n_images,free", [(30, range(17, 27)), (30, range(2, 22)), (24, [0, 5, 23]), (40, range(0, 40))])
def test_small_system_paths_match_oracle(mavba, oracle, n_images, free):
    """The one-work-group solve of small systems (k_chol_small): the blocks with a free parameter are ordered first and
    only their one or two tile columns are factorised, whatever the matrix size - here 3 to 4 tile columns in all with 10
    (one active tile), 20 (two), 3 scattered free images, and all 40 images free (four active tiles: the regular path)."""
    p = synth.make_scene(num_images=n_images, num_points=900, track_len=4, models=[A.MODEL_OPENCV], seed=21 + n_images)
    p.pose_const[:] = A.CONST_POSE
    p.pose_const[list(free)] = 0
    if len(list(free)) == n_images:
        p.pose_const[0] = A.CONST_POSE
        p.pose_const[1] = A.CONST_TX
    p.intr_const[:] = 1
    with mavba.Session(p) as s:
        info = s.info()
        assert info["matrix_dim"] == 64 * ((6 * n_images + 9 + 63) // 64)
        for radius in (1e4, 10.0):
            ref = oracle.linear_step(p, radius)
            st = s.linear_step(radius)
            assert rel_err(st["d_poses"], ref["d_poses"]) < 1e-8 and rel_err(st["d_points"], ref["d_points"]) < 1e-8
            assert abs(st["model_cost_change"] - ref["model_cost_change"]) < 1e-9 * abs(ref["model_cost_change"])
            fixed = np.setdiff1d(np.arange(n_images), list(free))
            if len(fixed):
                assert np.all(st["d_poses"][fixed[fixed > 1]] == 0.0)
    po, ro, _, pg, rg, _ = _solve_both(mavba, oracle, p)
    assert rg["termination"] == ro["termination"] and rg["num_successful_steps"] == ro["num_successful_steps"]
    assert abs(rg["final_cost"] - ro["final_cost"]) <= 1e-6 * ro["final_cost"]
    assert_params_close(pg, po)


def test_pose_refinement_matches_oracle(mavba, oracle):
    g = synth.make_scene(num_images=4, num_points=300, track_len=3, models=[A.MODEL_OPENCV], seed=21)
    sel = g.obs_image == 2
    uv, xyz = g.obs_uv[sel], g.truth["points"][g.obs_point[sel]]
    mask = np.ones(len(uv), np.uint8)
    mask[::9] = 0
    rvec, tvec = g.poses[2, :3].copy(), g.poses[2, 3:].copy()
    cam = np.concatenate([g.truth["intrinsics"][g.image_camera[2]][:8], [float(A.MODEL_OPENCV)]])
    cost, res = mavba.pose_refinement(rvec, tvec, cam, uv, xyz, mask)
    # oracle: same thing as a flat problem
    from mavmap_amd.problem import BAProblem
    keep = mask.astype(bool)
    q = BAProblem(poses=g.poses[2:3].copy(), pose_const=[0], image_camera=[0],
                  intrinsics=g.truth["intrinsics"][g.image_camera[2]][None].copy(), camera_model=[A.MODEL_OPENCV],
                  intr_const=[1], points=xyz[keep].copy(), point_const=np.ones(keep.sum(), np.uint8),
                  obs_uv=uv[keep].copy(), obs_image=np.zeros(keep.sum(), np.int32),
                  obs_point=np.arange(keep.sum(), dtype=np.int32))
    ro, _ = oracle.solve(q)
    assert res["termination"] == ro["termination"]
    assert abs(res["final_cost"] - ro["final_cost"]) <= 1e-6 * max(ro["final_cost"], 1e-12)
    assert rel_err(np.concatenate([rvec, tvec]), q.poses[0]) < 1e-6
    assert abs(cost - np.sqrt(ro["final_cost"] / ro["num_residuals"])) < 1e-6 * cost


@pytest.mark.parametrize("n", [5, 64, 65, 200, 777])
def test_dense_spd_solve(mavba, n):
    rng = np.random.default_rng(n)
    B = rng.normal(size=(n, n))
    Amat = B @ B.T + n * np.eye(n)
    # asymmetric-by-construction right-hand side and matrix: catches transposed MFMA tiles
    Amat[np.arange(n), np.arange(n)] += np.arange(n) * 0.5
    b = rng.normal(size=n)
    x = mavba.dense_spd_solve(Amat, b)
    x0 = np.linalg.solve(Amat, b)
    assert rel_err(x, x0) < 1e-10
    with pytest.raises(mavba.MavbaError):
        mavba.dense_spd_solve(-Amat, b)
    # one bad pivot anywhere - first, in the middle of a 16-pivot block, the very last - must be reported (the factorisation
    # does not test pivots one by one: the NaN of a non-positive pivot has to reach the end of its block)
    for k in sorted({0, n // 2, n - 1}):
        Bad = Amat.copy()
        Bad[k, k] = -1.0
        with pytest.raises(mavba.MavbaError):
            mavba.dense_spd_solve(Bad, b)


def test_edge_cases(mavba, oracle):
    # empty problem: returns immediately, cost 0 (the reference prints a warning and returns NaN
    # = sqrt(0/0); the shim computes that from these numbers)
    from mavmap_amd.problem import BAProblem
    e = BAProblem(poses=np.zeros((2, 6)), pose_const=[15, 2], image_camera=[0, 0], intrinsics=np.zeros((1, 9)),
                  camera_model=[1], intr_const=[0], points=np.zeros((0, 3)), point_const=[], obs_uv=np.zeros((0, 2)),
                  obs_image=[], obs_point=[])
    e.intrinsics[0, :4] = [600, 600, 376, 240]
    cost, res = mavba.bundle_adjustment(e)
    assert res["num_residuals"] == 0 and res["final_cost"] == 0.0 and np.isnan(cost)
    # a ragged problem: one point seen once (L = 1), an image without observations, a camera without images
    p = synth.make_scene(num_images=6, num_points=120, track_len=3, models=[A.MODEL_PINHOLE, A.MODEL_OPENCV], seed=31)
    keep = np.ones(p.num_obs, bool)
    first_of_pt5 = np.nonzero(p.obs_point == 5)[0]
    keep[first_of_pt5[1:]] = False
    keep[p.obs_image == 4] = False
    p.obs_uv, p.obs_image, p.obs_point = p.obs_uv[keep], p.obs_image[keep], p.obs_point[keep]
    po, ro, eo, pg, rg, eg = _solve_both(mavba, oracle, p, **global_opts())
    assert rg["termination"] == ro["termination"]
    assert abs(rg["final_cost"] - ro["final_cost"]) <= 1e-6 * ro["final_cost"]
    assert_params_close(pg, po)
    assert np.array_equal(pg.poses[4], p.poses[4])
    # bad inputs are rejected with the documented codes
    bad = p.copy(); bad.obs_image[0] = 99
    with pytest.raises(mavba.MavbaError) as ei:
        mavba.bundle_adjustment(bad)
    assert ei.value.code == A.ERR_BAD_INDEX
    bad = p.copy(); bad.camera_model[0] = 7
    with pytest.raises(mavba.MavbaError) as ei:
        mavba.bundle_adjustment(bad)
    assert ei.value.code == A.ERR_BAD_MODEL


def test_fixed_cost_blocks(mavba, oracle):
    """Residual blocks whose parameter blocks are all constant leave the program (fixed cost)."""
    p = synth.make_scene(num_images=6, num_points=200, track_len=3, models=[A.MODEL_PINHOLE], seed=41,
                         refine_camera_params=False)
    seen0 = np.unique(p.obs_point[p.obs_image == 0])
    p.point_const[seen0[:20]] = 1
    po, ro, eo, pg, rg, eg = _solve_both(mavba, oracle, p, **global_opts())
    assert ro["fixed_cost"] > 0
    assert abs(rg["fixed_cost"] - ro["fixed_cost"]) <= 1e-12 * ro["fixed_cost"]
    assert rg["num_residuals_reduced"] == ro["num_residuals_reduced"] < ro["num_residuals"]
    assert abs(rg["final_cost"] - ro["final_cost"]) <= 1e-6 * ro["final_cost"]
    # the dropped (all-constant) residual blocks still count in point3D_errors (bundle_adjustment.cc:583-596):
    # a GCP seen by a FIXED image under refine_camera_params = false
    m = ~np.isnan(eo)
    assert np.array_equal(m, ~np.isnan(eg))
    assert rel_err(eg[m], eo[m]) < 1e-6
    assert rel_err(eg[seen0[:20]], eo[seen0[:20]]) < 1e-6


def test_session_iterate_reset_and_determinism(mavba):
    p = synth.make_config("C2", scale=0.05, seed=51)
    with mavba.Session(p, global_opts()) as s:
        r1 = s.solve()
        x1 = s.get_params()
        s.reset()
        n = 0
        while True:
            done, term = s.iterate(1)
            n += done
            if term != A.TERM_RUNNING:
                break
        r2 = s.result()
        x2 = s.get_params()
    # the iteration that trips a tolerance is started but never counted (ceres returns before the
    # step is judged), hence the possible +1
    assert n - (r1["num_successful_steps"] + r1["num_unsuccessful_steps"]) in (0, 1)
    # fixed-shape reduction trees: bit-identical run to run
    assert r1["final_cost"] == r2["final_cost"]
    for a, b in zip(x1, x2):
        assert np.array_equal(a, b)


def test_full_size_c2_properties(mavba):
    """BASELINE config C2 at full size (100 images / 30k points / 300k observations): properties that
    do not need the oracle — monotone cost, first-order optimality, ground-truth recovery on clean data."""
    p = synth.make_config("C2")
    assert (p.num_images, p.num_points, p.num_obs) == (100, 30000, 300000)
    q = p.copy()
    cost, res = mavba.bundle_adjustment(q, global_opts())
    assert res["termination"] in (A.TERM_FUNCTION_TOLERANCE, A.TERM_GRADIENT_TOLERANCE, A.TERM_PARAMETER_TOLERANCE)
    assert res["final_cost"] < 0.2 * res["initial_cost"]
    # noise-free observations from the perturbed start -> exact recovery up to the datum
    clean = synth.make_scene(num_images=100, num_points=30000, track_len=10, models=[A.MODEL_PINHOLE], seed=1002,
                             noise_px=0.0, outlier_frac=0.0)
    c = clean.copy()
    cost, res = mavba.bundle_adjustment(c, global_opts(function_tolerance=1e-14, gradient_tolerance=1e-14))
    assert cost < 1e-6
    uv = _reproject(c)
    assert np.abs(uv - clean.truth["uv_clean"]).max() < 1e-5


def _reproject(p):
    R = synth.rodrigues(p.poses[:, :3])
    Xc = np.einsum("nij,nj->ni", R[p.obs_image], p.points[p.obs_point]) + p.poses[p.obs_image, 3:]
    uv = np.zeros((p.num_obs, 2))
    for c in range(p.num_cameras):
        sel = p.image_camera[p.obs_image] == c
        uv[sel] = synth.project(int(p.camera_model[c]), p.intrinsics[c], Xc[sel])
    return uv


def test_envelope_factorisation_on_a_banded_system(mavba, oracle):
    """A long strip of images gives a banded reduced camera system: the factorisation must skip the
    structurally-zero tiles (envelope < dense) and still reproduce the oracle's dense solve."""
    p = synth.make_scene(num_images=140, num_points=4000, track_len=6, models=[A.MODEL_PINHOLE, A.MODEL_OPENCV], seed=91)
    with mavba.Session(p) as s:
        info = s.info()
        assert info["reduced_dim"] == 6 * 140 + 18 and info["padded_dim"] % 64 == 0
        assert info["envelope_tiles"] < info["dense_tiles"]
        assert info["factor_flops"] < 64.0 ** 3 * 2 * info["dense_tiles"] * 14  # sanity: finite, bounded
        # round 6: the device keeps the envelope's tiles only (+ one right-hand-side tile per tile column), not a dense array
        nb = info["matrix_dim"] // 64
        assert info["reduced_store_bytes"] == (info["envelope_tiles"] + nb) * 64 * 64 * 8
        assert info["reduced_store_bytes"] < (info["matrix_dim"] + 64) * info["matrix_dim"] * 8
        S, v = s.reduced_system(1e4)   # (spread back into the dense form at the boundary)
        assert np.array_equal(S, S.T) and np.all(np.diag(S) > 0)
        st = s.linear_step(1e4)
    ref = oracle.linear_step(p, 1e4)
    assert rel_err(st["d_poses"], ref["d_poses"]) < 1e-8
    assert rel_err(st["d_intr"], ref["d_intr"]) < 1e-8
    assert rel_err(st["d_points"], ref["d_points"]) < 1e-8


# ---- elimination order / concurrent fronts ------------------------------------------------------------

def _band_scene():
    return synth.make_scene(num_images=130, num_points=5000, track_len=4, models=[A.MODEL_PINHOLE, A.MODEL_OPENCV],
                            seed=5, long_track_frac=0.01, long_track_len=12, spacing=6.0)


@pytest.mark.parametrize("parts", [0, 2, 3, 5])
def test_dissection_orders_give_the_oracle_step(mavba, oracle, parts, monkeypatch):
    """The reduced system is assembled in a nested-dissection order and its leading parts are factorised
    concurrently (shadow blocks for the separator). Whatever the number of parts, the system handed out in the
    variables' order and the solved step must be the oracle's."""
    p = _band_scene()
    ref = oracle.linear_step(p, 1e4)
    monkeypatch.setenv("MAVBA_ND_PARTS", str(parts))
    with mavba.Session(p) as s:
        info = s.info()
        S, v = s.reduced_system(1e4)
        st = s.linear_step(1e4)
    if parts > 1:
        assert 2 <= info["nd_parts"] <= parts and info["matrix_dim"] >= info["padded_dim"]
        if parts < 5:  # (5 parts of this small scene are forced, not profitable)
            assert info["chain_steps"] < info["padded_dim"] // 64
    else:
        assert info["nd_parts"] == 0 and info["chain_steps"] == info["padded_dim"] // 64
    assert rel_err(S, ref["S"]) < 1e-9 and np.abs(S - S.T).max() == 0.0
    assert rel_err(v, ref["v"]) < 1e-9
    for k in ("d_poses", "d_intr", "d_points"):
        assert rel_err(st[k], ref[k]) < 1e-8, (parts, k)
    assert abs(st["model_cost_change"] - ref["model_cost_change"]) < 1e-9 * abs(ref["model_cost_change"])


def test_dissection_is_chosen_automatically_and_solves_like_the_oracle(mavba, oracle):
    p = _band_scene()
    with mavba.Session(p) as s:
        info = s.info()
    assert info["nd_parts"] >= 2, info
    # the persistent launch's queues come from the host-side timing model: it has walked the tree (a few chain columns at least)
    assert info["chol_model_forward_us"] > 20.0, info
    po, ro, eo, pg, rg, eg = _solve_both(mavba, oracle, p, **global_opts())
    assert rg["termination"] == ro["termination"]
    assert rg["num_successful_steps"] == ro["num_successful_steps"]
    assert abs(rg["final_cost"] - ro["final_cost"]) <= 1e-6 * ro["final_cost"]
    assert_params_close(pg, po)


def test_dissection_with_constant_blocks_and_unused_images(mavba, oracle):
    """Dissection order + point clusters when some pose blocks are constant, some images have no observation
    and one camera's intrinsics are fixed: the unit-diagonal columns sit inside parts and separator."""
    p = _band_scene()
    rng = np.random.default_rng(3)
    p.pose_const = p.pose_const.copy()
    for i in rng.choice(np.arange(2, p.num_images), 12, replace=False):
        p.pose_const[i] = int(rng.choice([A.CONST_POSE, A.CONST_RVEC, A.CONST_TX | A.CONST_TZ]))
    drop = rng.choice(np.arange(2, p.num_images), 3, replace=False)
    keep = ~np.isin(p.obs_image, drop)
    p.obs_uv, p.obs_image, p.obs_point = p.obs_uv[keep], p.obs_image[keep], p.obs_point[keep]
    p.intr_const = np.array([0, 1], np.uint8)
    ref = oracle.linear_step(p, 300.0)
    with mavba.Session(p) as s:
        assert s.info()["nd_parts"] >= 2 and s.info()["num_clusters"] > 0
        st = s.linear_step(300.0)
    for k in ("d_poses", "d_intr", "d_points"):
        assert rel_err(st[k], ref[k]) < 1e-8, k
    assert not st["d_poses"][drop].any() and not st["d_intr"][1].any()


def test_point_seen_twice_by_one_image_and_long_tracks_mix_with_clusters(mavba, oracle):
    """Points the clusters cannot take (an image observing the point twice; more images than a cluster's local
    list) go through the generic term lists; both paths add into the same blocks."""
    p = synth.make_scene(num_images=30, num_points=900, track_len=5, models=[A.MODEL_PINHOLE, A.MODEL_OPENCV], seed=23,
                         long_track_frac=0.05, long_track_len=24, spacing=5.0)
    rng = np.random.default_rng(5)
    dup = rng.choice(p.num_obs, 40, replace=False)  # a second, slightly different observation of the same (image, point)
    p.obs_uv = np.concatenate([p.obs_uv, p.obs_uv[dup] + rng.normal(0, 0.3, (len(dup), 2))])
    p.obs_image = np.concatenate([p.obs_image, p.obs_image[dup]]).astype(np.int32)
    p.obs_point = np.concatenate([p.obs_point, p.obs_point[dup]]).astype(np.int32)
    for radius in (1e4, 10.0):
        ref = oracle.linear_step(p, radius)
        with mavba.Session(p) as s:
            info = s.info()
            S, v = s.reduced_system(radius)
            st = s.linear_step(radius)
        assert info["num_clusters"] > 0 and 0 < info["clustered_points"] < p.num_points and sum(info["schur_terms"]) > 0
        assert rel_err(S, ref["S"]) < 1e-9 and rel_err(v, ref["v"]) < 1e-9
        for k in ("d_poses", "d_intr", "d_points"):
            assert rel_err(st[k], ref[k]) < 1e-8, (radius, k)


def test_long_tracks_and_constant_points_behind_the_clusters(mavba, oracle, monkeypatch):
    """The point order puts the points that can never be clustered - more than 16 observations, constant points - behind all
    the others: the clustered part takes the fused kernel, that tail the separate front end + the generic term lists. Same
    system as the oracle's and as the route that does not fuse; a full solve like the oracle's."""
    p = synth.make_scene(num_images=40, num_points=1500, track_len=5, models=[A.MODEL_PINHOLE, A.MODEL_OPENCV], seed=29,
                         long_track_frac=0.06, long_track_len=24, spacing=5.0)
    p.point_const[::37] = 1  # a few constant (control) points with observations
    ref = oracle.linear_step(p, 1e3)
    with mavba.Session(p, dict(profile_kernels=1)) as s:
        info = s.info()
        S, v = s.reduced_system(1e3)
        st = s.linear_step(1e3)
        timers = set(s.kernel_stats())
    # both kernels ran: clusters fused, the tail through k_point_front and the term lists
    assert "schur_fused" in timers and "point_front" in timers and "schur_clusters" not in timers
    assert info["num_clusters"] > 0 and 0 < info["clustered_points"] < p.num_points and sum(info["schur_terms"]) > 0
    assert rel_err(S, ref["S"]) < 1e-9 and rel_err(v, ref["v"]) < 1e-9
    for k in ("d_poses", "d_intr", "d_points"):
        assert rel_err(st[k], ref[k]) < 1e-8, k
    po, ro, eo, pg, rg, eg = _solve_both(mavba, oracle, p, **global_opts())
    assert rg["termination"] == ro["termination"] and rg["num_successful_steps"] == ro["num_successful_steps"]
    assert_params_close(pg, po)
    monkeypatch.setenv("MAVBA_NO_FUSE", "1")
    with mavba.Session(p) as s:
        S2, v2 = s.reduced_system(1e3)
    assert rel_err(S2, S) < 1e-11 and rel_err(v2, v) < 1e-11


def test_clusters_off_gives_the_same_system(mavba, monkeypatch):
    p = _scene("mixed")
    with mavba.Session(p) as s:
        S1, v1 = s.reduced_system(1e4)
        assert s.info()["num_clusters"] > 0
    monkeypatch.setenv("MAVBA_CLUSTERS", "0")
    with mavba.Session(p) as s:
        S0, v0 = s.reduced_system(1e4)
        assert s.info()["num_clusters"] == 0 and sum(s.info()["schur_terms"]) > 0
    assert rel_err(S1, S0) < 1e-12 and rel_err(v1, v0) < 1e-12


@pytest.mark.parametrize("kind", ["mixed", "fixed_intr"])
def test_the_three_front_ends_give_the_same_system_and_solve(mavba, oracle, kind, monkeypatch):
    """k_schur_fused (every point clustered), k_point_front + k_schur_clusters (MAVBA_NO_FUSE) and the plane kernels of
    rounds 1-2 (MAVBA_FRONT_PLANES: still the route of a point seen by more than 96 refined cameras) are three routes to the
    same reduced system: identical up to summation order, and each solves like the oracle."""
    p = _scene(kind)
    po = p.copy()
    ro, _ = oracle.solve(po, oracle.options(**global_opts()), want_point_errors=True)
    systems, timers = [], []
    for env in (None, "MAVBA_NO_FUSE", "MAVBA_FRONT_PLANES"):
        if env:
            monkeypatch.setenv(env, "1")
        with mavba.Session(p, dict(profile_kernels=1)) as s:
            systems.append(s.reduced_system(1e4))
            timers.append(set(s.kernel_stats()))
        q = p.copy()
        _, res = mavba.bundle_adjustment(q, global_opts())
        assert res["termination"] == ro["termination"] and res["num_successful_steps"] == ro["num_successful_steps"], env
        assert_params_close(q, po, what=str(env))
    # the routes really differ
    assert "schur_fused" in timers[0] and "schur_fused" not in timers[1] and "point_front" in timers[1]
    assert "point_front" not in timers[2] and "entries_pose" in timers[2]
    S0, v0 = systems[0]
    for S, v in systems[1:]:
        assert rel_err(S, S0) < 1e-11 and rel_err(v, v0) < 1e-11


# ---- LM control flow around the deferred read-back of the evaluation ------------------------------------

@pytest.mark.parametrize("optkw", [dict(gradient_tolerance=1e-3), dict(gradient_tolerance=1e-2, function_tolerance=1e-12),
                                   dict(max_num_iterations=3), dict(max_num_iterations=1), dict(parameter_tolerance=1e-3)])
def test_termination_kinds_match_oracle(mavba, oracle, optkw):
    """Gradient tolerance (tested when an evaluation's scalars arrive, one host sync later than Ceres does it), the
    iteration limit and the parameter tolerance must end the solve at the same iteration as the oracle."""
    p = synth.make_scene(num_images=8, num_points=400, track_len=4, models=[A.MODEL_PINHOLE], seed=31, noise_px=0.05,
                         outlier_frac=0.0)
    kw = dict(global_opts(), **optkw)
    po, ro, eo, pg, rg, eg = _solve_both(mavba, oracle, p, **kw)
    assert rg["termination"] == ro["termination"], (rg["termination_name"], ro["termination_name"])
    assert rg["num_successful_steps"] == ro["num_successful_steps"]
    assert rg["num_unsuccessful_steps"] == ro["num_unsuccessful_steps"]
    assert abs(rg["final_cost"] - ro["final_cost"]) <= 1e-6 * ro["final_cost"] + 1e-18
    assert_params_close(pg, po)


def test_stepwise_iteration_equals_one_call(mavba):
    """mavba_session_iterate(1) repeated must walk the same path as one call (the deferred evaluation is completed
    at the end of every call)."""
    p = _scene("mixed")
    with mavba.Session(p, global_opts()) as s:
        ra = s.solve()
        pa = s.get_params()
    with mavba.Session(p, global_opts()) as s:
        steps = 0
        while True:
            done, term = s.iterate(1)
            steps += done
            if term != A.TERM_RUNNING or steps > 500:
                break
        rb = s.result()
        pb = s.get_params()
    assert rb["termination"] == ra["termination"] and rb["num_successful_steps"] == ra["num_successful_steps"]
    assert rb["num_unsuccessful_steps"] == ra["num_unsuccessful_steps"] and rb["final_cost"] == ra["final_cost"]
    for x, y in zip(pa, pb):
        assert np.array_equal(x, y)


# ---- batched pose refinement (on-device trust-region loop) ----------------------------------------------

def _pose_items(seed, count, model=A.MODEL_OPENCV):
    """`count` pose-refinement problems as MAVMAP poses them: one image, RANSAC inlier masks, perturbed start."""
    g = synth.make_scene(num_images=6, num_points=500, track_len=3, models=[model], seed=seed)
    rng = np.random.default_rng(seed)
    items = []
    for q in range(count):
        img = 1 + q % (g.num_images - 1)
        sel = g.obs_image == img
        uv, xyz = g.obs_uv[sel], g.truth["points"][g.obs_point[sel]]
        mask = (rng.random(len(uv)) > 0.15).astype(np.uint8)   # a different inlier set per hypothesis
        K = A.MODEL_NUM_PARAMS[model]
        cam = np.concatenate([g.truth["intrinsics"][g.image_camera[img]][:K], [float(model)]])
        pose = g.truth["poses"][img] + rng.normal(0, [0.01] * 3 + [0.3] * 3)
        items.append(dict(rvec=pose[:3].copy(), tvec=pose[3:].copy(), camera_params=cam, points2D=uv, points3D=xyz, inlier_mask=mask))
    return items


def _oracle_pose(oracle, it, **optkw):
    from mavmap_amd.problem import BAProblem
    keep = np.asarray(it["inlier_mask"], bool)
    cam = np.asarray(it["camera_params"], float)
    model = int(cam[-1])
    q = BAProblem(poses=np.concatenate([it["rvec"], it["tvec"]])[None].copy(), pose_const=[0], image_camera=[0],
                  intrinsics=np.pad(cam[:-1], (0, 9 - (len(cam) - 1)))[None].copy(), camera_model=[model], intr_const=[1],
                  points=it["points3D"][keep].copy(), point_const=np.ones(keep.sum(), np.uint8), obs_uv=it["points2D"][keep].copy(),
                  obs_image=np.zeros(keep.sum(), np.int32), obs_point=np.arange(keep.sum(), dtype=np.int32))
    ro, _ = oracle.solve(q, oracle.options(**optkw))
    return q.poses[0], ro


@pytest.mark.parametrize("model", [A.MODEL_PINHOLE, A.MODEL_OPENCV, A.MODEL_CATA])
def test_pose_refinement_batch_matches_oracle(mavba, oracle, model):
    items = _pose_items(71 + model, 12, model)
    ref = [_oracle_pose(oracle, it) for it in items]      # BundleAdjustmentOptions defaults, like the reference call
    out = mavba.pose_refinement_batch(items)
    for it, (cost, res), (pose_o, ro) in zip(items, out, ref):
        assert res["termination"] == ro["termination"], (res["termination_name"], ro["termination_name"])
        assert res["num_successful_steps"] == ro["num_successful_steps"] and res["num_unsuccessful_steps"] == ro["num_unsuccessful_steps"]
        assert res["num_residuals"] == ro["num_residuals"] and res["num_parameters_reduced"] == ro["num_parameters_reduced"]
        assert abs(res["initial_cost"] - ro["initial_cost"]) <= 1e-10 * ro["initial_cost"]
        assert abs(res["final_cost"] - ro["final_cost"]) <= 1e-6 * ro["final_cost"]
        assert rel_err(np.concatenate([it["rvec"], it["tvec"]]), pose_o) < 1e-6
        assert abs(cost - np.sqrt(ro["final_cost"] / ro["num_residuals"])) < 1e-6 * cost


def test_pose_refinement_batch_equals_single_calls_and_the_session_path(mavba, monkeypatch):
    items = _pose_items(5, 7)
    singles = [dict(it, rvec=it["rvec"].copy(), tvec=it["tvec"].copy()) for it in items]
    sess = [dict(it, rvec=it["rvec"].copy(), tvec=it["tvec"].copy()) for it in items]
    out = mavba.pose_refinement_batch(items, max_num_iterations=50, function_tolerance=1e-10)
    for it, s, (c, r) in zip(items, singles, out):
        c1, r1 = mavba.pose_refinement(s["rvec"], s["tvec"], s["camera_params"], s["points2D"], s["points3D"], s["inlier_mask"],
                                       max_num_iterations=50, function_tolerance=1e-10)
        assert c1 == c and np.array_equal(s["rvec"], it["rvec"]) and np.array_equal(s["tvec"], it["tvec"])  # bit-identical
        assert r1["num_successful_steps"] == r["num_successful_steps"]
    # the general session (host trust-region loop) walks the same path
    monkeypatch.setenv("MAVBA_POSE_REFINE_SESSION", "1")
    for it, s, (c, r) in zip(items, sess, out):
        c2, r2 = mavba.pose_refinement(s["rvec"], s["tvec"], s["camera_params"], s["points2D"], s["points3D"], s["inlier_mask"],
                                       max_num_iterations=50, function_tolerance=1e-10)
        assert r2["termination"] == r["termination"] and r2["num_successful_steps"] == r["num_successful_steps"]
        assert abs(c2 - c) < 1e-9 * c and rel_err(np.concatenate([s["rvec"], s["tvec"]]), np.concatenate([it["rvec"], it["tvec"]])) < 1e-9
    # empty inlier set and an empty batch
    e = dict(items[0], inlier_mask=np.zeros(len(items[0]["points2D"]), np.uint8))
    (c, r), = mavba.pose_refinement_batch([e])
    assert r["num_residuals"] == 0 and np.isnan(c)
    assert mavba.pose_refinement_batch([]) == []


def test_the_lm_decision_is_the_same_function_on_the_host_and_on_the_device(mavba):
    """The speculative evaluation (csrc/lm_decide.h) is enqueued before the host has read the candidate's scalars: a one-lane
    kernel takes the accept / reject / terminate decision and the new radius on the device, the host takes them from the same
    function on the same scalars - they must agree bit for bit (IEEE operations only, contraction off in both builds).
    Random scalars around every threshold, failure flags, NaN / negative model changes, pending evaluations."""
    from mavmap_amd import api
    rng = np.random.default_rng(5)
    n = 20000
    c = np.zeros((n, 24))
    cost = 10.0 ** rng.uniform(-3, 6, n)
    c[:, A.SC_COST] = cost
    c[:, A.SC_XNORM2] = 10.0 ** rng.uniform(-2, 8, n)
    c[:, A.SC_GRAD_MAX] = 10.0 ** rng.uniform(-12, 3, n)
    rel = np.where(rng.random(n) < 0.3, 10.0 ** rng.uniform(-9, -2, n), rng.uniform(-0.5, 1.5, n))   # cost decrease relative to the cost
    c[:, A.SC_NEW_COST] = cost * (1.0 - rel * 10.0 ** rng.uniform(-6, 0, n))
    c[:, A.SC_STEP_NORM2] = 10.0 ** rng.uniform(-20, 4, n)
    c[:, A.SC_MODEL_CHANGE] = np.abs(cost - c[:, A.SC_NEW_COST]) * rng.uniform(0.2, 3.0, n) * np.where(rng.random(n) < 0.05, -1.0, 1.0)
    c[:, A.SC_CAND_XNORM2] = c[:, A.SC_XNORM2]
    bad = rng.random(n)
    c[bad < 0.02, A.SC_FAIL] = 1.0
    c[(bad > 0.02) & (bad < 0.04), A.SC_FAIL_FRONT] = 2.0
    c[(bad > 0.04) & (bad < 0.05), A.SC_MODEL_CHANGE] = np.nan
    c[(bad > 0.05) & (bad < 0.06), A.SC_STEP_NORM2] = np.inf
    c[:, 16] = 10.0 ** rng.uniform(-3, 12, n)          # radius
    c[:, 17] = 2.0 ** rng.integers(1, 8, n)            # decrease factor
    c[:, 18] = 1e-8; c[:, 19] = 10.0 ** rng.uniform(-8, -3, n); c[:, 20] = 1e-3; c[:, 21] = 1e16
    c[:, 22] = 10.0 ** rng.uniform(-12, 0, n)          # absolute gradient tolerance
    c[:, 23] = rng.random(n) < 0.5                     # an evaluation's scalars arrive with the candidate
    host, dev = api.debug_lm_decide(c)
    assert np.array_equal(host, dev, equal_nan=True)
    codes = set(host[:, 0].astype(int))
    assert codes == {0, 1, 2, 3, 4, 5}, codes          # every outcome occurs in the sample


@pytest.mark.parametrize("refine_intr", [False, True])
def test_merged_launches_of_a_local_window_equal_the_launch_per_step_loop(mavba, refine_intr, monkeypatch):
    """Round 4: for a problem whose reduced system one work-group solves, the evaluation's reductions (k_eval_small), the
    camera update (inside k_chol_small) and the candidate's reductions + decision (k_lm_tail) run as three launches instead
    of nine. They walk the same device functions (lm_bodies.h) in the same order: every result is bit-identical to the
    launch-per-step loop (MAVBA_MERGE=0)."""
    g = synth.make_config("C1")
    outs = []
    for merge in ("1", "0"):
        monkeypatch.setenv("MAVBA_MERGE", merge)
        res = []
        for first in (0, 2):
            p = synth.local_ba_window(g, first)
            p.intr_const[:] = 0 if refine_intr else 1
            e = np.full(p.num_points, np.nan)
            _, r = mavba.bundle_adjustment(p, {}, point3D_errors=e)
            res.append((p.poses.copy(), p.intrinsics.copy(), p.points.copy(), e, r["final_cost"], r["num_successful_steps"],
                        r["num_unsuccessful_steps"], r["termination"], r["final_gradient_max_norm"], r["final_trust_region_radius"]))
        # a window with rotation priors and a constant point, 12 images (two tile columns with intrinsics)
        q = synth.make_scene(num_images=12, num_points=1500, track_len=4, models=[A.MODEL_PINHOLE, A.MODEL_OPENCV], seed=5, rot_priors=True)
        q.pose_const[:2] = A.CONST_POSE
        q.intr_const[:] = 0 if refine_intr else 1
        _, r = mavba.bundle_adjustment(q, dict(max_num_iterations=30))
        res.append((q.poses.copy(), q.intrinsics.copy(), q.points.copy(), r["final_cost"], r["num_successful_steps"],
                    r["num_unsuccessful_steps"], r["termination"]))
        outs.append(res)
    for a, b in zip(*outs):
        for x, y in zip(a, b):
            assert np.array_equal(np.asarray(x), np.asarray(y), equal_nan=True)


def test_the_speculative_loop_equals_the_plain_loop(mavba, monkeypatch):
    """Round 4's loop enqueues the evaluation at the candidate point BEFORE the host has read the candidate's scalars
    (k_lm_tail decides on the device, the host repeats the decision and keeps or forgets the evaluation: books restored,
    pointers swapped back, failure slots, the front end's radius taken from the device). MAVBA_SPECULATE=0 is the plain
    loop. Both must walk the same path to the last bit - accepted AND rejected steps, every way a solve can end."""
    def rough(seed, radius):
        p = synth.make_scene(num_images=24, num_points=2500, track_len=6, models=[A.MODEL_PINHOLE, A.MODEL_OPENCV], seed=seed, outlier_frac=0.1)
        rng = np.random.default_rng(seed)
        p.poses[:, :3] += rng.normal(0, 0.05, p.poses[:, :3].shape)
        p.poses[:, 3:] += rng.normal(0, 1.5, p.poses[:, 3:].shape)
        p.points += rng.normal(0, 1.5, p.points.shape)
        return p, global_opts(initial_trust_region_radius=radius)
    small = synth.make_scene(num_images=8, num_points=400, track_len=4, models=[A.MODEL_PINHOLE], seed=31, noise_px=0.05, outlier_frac=0.0)
    cases = [rough(41, 1e14), rough(42, 1e10),                                  # 9 and 6 rejected steps (oracle run of the same scenes)
             (_scene("mixed"), global_opts()), (_scene("priors"), global_opts()),
             (small, global_opts(gradient_tolerance=1e-3)),                         # gradient tolerance
             (small, global_opts(gradient_tolerance=1e-2, function_tolerance=1e-12)),
             (small, global_opts(max_num_iterations=3)), (small, global_opts(max_num_iterations=1)),
             (small, global_opts(parameter_tolerance=1e-3)),                        # parameter tolerance
             (synth.local_ba_window(synth.make_config("C1"), 2), {})]              # a local window (merged launches)
    outs = []
    for spec in ("1", "0"):
        monkeypatch.setenv("MAVBA_SPECULATE", spec)
        res = []
        for p0, kw in cases:
            p = p0.copy()
            e = np.full(p.num_points, np.nan)
            _, r = mavba.bundle_adjustment(p, kw, point3D_errors=e)
            res.append((p.poses.copy(), p.intrinsics.copy(), p.points.copy(), e, r["final_cost"], r["num_successful_steps"],
                        r["num_unsuccessful_steps"], r["termination"], r["final_gradient_max_norm"], r["final_trust_region_radius"]))
        outs.append(res)
    assert sum(r[6] for r in outs[0][:2]) > 0                      # rejected steps did occur
    assert len({r[7] for r in outs[0]}) >= 3                       # several kinds of termination
    for a, b in zip(*outs):
        for x, y in zip(a, b):
            assert np.array_equal(np.asarray(x), np.asarray(y), equal_nan=True)


def test_merged_evaluation_tail_of_a_large_problem_equals_the_four_launches(mavba, monkeypatch):
    """Round 5: for problems of up to 160 images on one rank the tail of an evaluation - per-image sums, per-camera sums, norms, the
    three scalar reductions - is two launches (k_eval_head / k_eval_tail) instead of four. They run the device functions of the
    four kernels in the same order: bit-identical to the launch-per-step loop (MAVBA_MERGE=0) - with free intrinsics,
    rotation priors, constant points and more camera columns than one norm group."""
    p1 = synth.make_config("C3", scale=0.12, seed=21)                       # 60 images: 378 camera columns, two cameras
    p2 = synth.make_scene(num_images=90, num_points=6000, track_len=6, models=[A.MODEL_PINHOLE, A.MODEL_OPENCV], seed=22, rot_priors=True)
    p2.point_const[::11] = 1
    p3 = synth.make_scene(num_images=70, num_points=5000, track_len=5, models=[A.MODEL_CATA], seed=23, refine_camera_params=False)
    assert all(6 * q.num_images > 128 and q.num_images <= 160 for q in (p1, p2, p3))   # (beyond the one-work-group path of local windows)
    outs = []
    for merge in ("1", "0"):
        monkeypatch.setenv("MAVBA_MERGE", merge)
        res = []
        for p0 in (p1, p2, p3):
            p = p0.copy()
            e = np.full(p.num_points, np.nan)
            with mavba.Session(p, dict(global_opts(), profile_kernels=1)) as s:
                r = s.solve()
                x = s.get_params()
                timers = set(s.kernel_stats())
            res.append((x[0], x[1], x[2], r["final_cost"], r["num_successful_steps"], r["num_unsuccessful_steps"], r["termination"],
                        r["final_gradient_max_norm"], r["final_trust_region_radius"]))
            assert ("eval_tail" in timers) == (merge == "1"), timers
        outs.append(res)
    for a, b in zip(*outs):
        for x, y in zip(a, b):
            assert np.array_equal(np.asarray(x), np.asarray(y), equal_nan=True)



@pytest.mark.parametrize("scale", [0.12, 0.4])
def test_pre_reduction_riding_in_the_evaluation_changes_nothing(mavba, monkeypatch, scale):
    """Round 6: the pre-reduction of long runs of block partials (k_partial_reduce's tasks) runs as extra work-groups of the
    evaluation's next kernel - k_eval_head up to 160 images (scale 0.12: 60 images), k_camera_reduce_img above (scale 0.4: 200
    images) - instead of a launch of its own before the finalize pass; a rejected step, whose speculative evaluation never
    ran, and a repeated solve with another radius must find the books right. Same tasks, same order: bit-identical solves."""
    p0 = synth.make_config("C3", scale=scale, seed=27)
    assert p0.num_obs >= 200_000   # (from there runs of more than 64 partials are pre-reduced)
    out = []
    for ride in ("1", "0"):
        monkeypatch.setenv("MAVBA_PARTIAL_RIDE", ride)
        p = p0.copy()
        with mavba.Session(p, global_opts()) as s:
            a = s.linear_step(1e4)
            b = s.linear_step(3e2)          # (another radius on the same evaluation: the front end runs again, without the evaluation)
            r = s.solve()
            x = s.get_params()
        out.append((a["d_poses"], a["d_points"], b["d_poses"], b["d_points"], x[0], x[1], x[2], r["final_cost"], r["num_successful_steps"],
                    r["num_unsuccessful_steps"], r["termination"]))
    for u, v in zip(*out):
        assert np.array_equal(np.asarray(u), np.asarray(v))


def test_lane_placement_and_its_fallback_match_the_oracle(mavba, oracle, monkeypatch, capfd):
    """k_schur_rows, round 6: in a cluster with two camera slots the set-up places a point's observations in the 8-lane half of
    their camera's slot; a point with more than 8 observations of ONE slot cannot be placed, its cluster keeps the
    reduce-scatter's first halving with the selects. Three of four images on the first camera: both kinds of cluster occur
    (the set-up's statistics say so), and the reduced system, the step and a complete solve are the oracle's."""
    image_camera = np.array([0, 0, 0, 1] * 12, np.int32)
    p = synth.make_scene(num_images=48, num_points=4000, track_len=12, models=[A.MODEL_PINHOLE, A.MODEL_OPENCV], seed=61,
                         spacing=5.0, image_camera=image_camera)
    monkeypatch.setenv("MAVBA_CLUSTER_STATS", "1")
    ref = oracle.linear_step(p, 1e4)
    with mavba.Session(p) as s:
        S, v = s.reduced_system(1e4)
        st = s.linear_step(1e4)
    err = capfd.readouterr().err
    line = [l for l in err.splitlines() if "clusters that keep the selects" in l]
    assert line, err[-2000:]
    kept = int(line[-1].rsplit(":", 1)[1])
    total = sum(int(tok.split("/")[0]) for tok in line[-1].split(";")[0].split() if "/" in tok and tok[0].isdigit())
    assert 0 < kept < total, line[-1]  # both the placed and the unplaced form ran
    assert rel_err(S, ref["S"]) < 1e-9 and rel_err(v, ref["v"]) < 1e-9
    assert rel_err(st["d_poses"], ref["d_poses"]) < 1e-8 and rel_err(st["d_intr"], ref["d_intr"]) < 1e-8
    assert rel_err(st["d_points"], ref["d_points"]) < 1e-8
    monkeypatch.delenv("MAVBA_CLUSTER_STATS")
    po, ro, eo, pg, rg, eg = _solve_both(mavba, oracle, p, **global_opts())
    assert ro["num_successful_steps"] == rg["num_successful_steps"] and ro["termination"] == rg["termination"]
    assert abs(rg["final_cost"] - ro["final_cost"]) <= 1e-6 * abs(ro["final_cost"])
    assert_params_close(pg, po)
