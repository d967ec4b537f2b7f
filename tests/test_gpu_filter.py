"""N3: point filtering + re-BA on the resident session against the oracle.

Reference flow (src/mapper.cc:1206, 1218-1224): adjust_global_bundle; filter_point_cloud (points whose point3D error
exceeds filter_max_error are deleted from the FeatureManager, :382-402); adjust_global_bundle again. The oracle does
exactly that with two fresh problems; the device keeps ONE session: the filtered points' residual blocks get zero
weight, counts / used blocks / fixed cost are re-derived, the solve restarts from the current parameters."""
import numpy as np
import pytest

from mavmap_amd import _abi as A
from mavmap_amd import synth
from tests.conftest import assert_params_close, global_opts, rel_err

pytestmark = pytest.mark.gpu


def _drop_points(p, removed):
    """The reference's delete_point3D: every observation of the point goes; the slot stays (unused)."""
    q = p.copy()
    keep = ~removed.astype(bool)[p.obs_point]
    q.obs_uv, q.obs_image, q.obs_point = (np.ascontiguousarray(p.obs_uv[keep]), np.ascontiguousarray(p.obs_image[keep]),
                                          np.ascontiguousarray(p.obs_point[keep]))
    return q


def _oracle_flow(oracle, p, max_error, keep=None, **optkw):
    a = p.copy()
    r1, e1 = oracle.solve(a, oracle.options(**optkw), want_point_errors=True)
    removed = np.zeros(p.num_points, np.uint8)
    sel = e1 > max_error            # NaN (no observations) never compares true
    if keep is not None:
        sel &= ~np.asarray(keep, bool)
    removed[sel] = 1
    b = _drop_points(a, removed)
    r2, e2 = oracle.solve(b, oracle.options(**optkw), want_point_errors=True)
    return a, b, removed, r1, r2, e1, e2


SCENES = {
    "mixed": lambda: synth.make_config("C3", scale=0.01, seed=12),
    "priors_long": lambda: synth.make_scene(num_images=40, num_points=1500, track_len=10, models=[A.MODEL_PINHOLE, A.MODEL_OPENCV],
                                            seed=17, rot_priors=True, long_track_frac=0.05, long_track_len=36, spacing=5.0),
    "band": lambda: synth.make_scene(num_images=130, num_points=5000, track_len=4, models=[A.MODEL_PINHOLE, A.MODEL_OPENCV],
                                     seed=5, long_track_frac=0.01, long_track_len=12, spacing=6.0),
}


@pytest.mark.parametrize("kind", sorted(SCENES))
def test_filter_and_rebundle_matches_oracle(mavba, oracle, kind):
    p = SCENES[kind]()
    max_error = 1.2  # px: the 1 % gross outliers (and a few noisy tracks) go
    a, b, removed_o, r1o, r2o, e1o, e2o = _oracle_flow(oracle, p, max_error, **global_opts())
    assert 0 < removed_o.sum() < p.num_points // 4
    g = p.copy()
    eg = np.full(p.num_points, np.nan)
    removed_g, r1g, r2g = mavba.bundle_adjustment_filter_rebundle(g, max_error, global_opts(), point3D_errors=eg)
    assert np.array_equal(removed_g, removed_o)
    for rg, ro in ((r1g, r1o), (r2g, r2o)):
        assert rg["termination"] == ro["termination"]
        assert rg["num_successful_steps"] == ro["num_successful_steps"] and rg["num_unsuccessful_steps"] == ro["num_unsuccessful_steps"]
        for k in ("num_residuals", "num_residuals_reduced", "num_parameters_reduced"):
            assert rg[k] == ro[k], k
        assert abs(rg["initial_cost"] - ro["initial_cost"]) <= 1e-6 * ro["initial_cost"]
        assert abs(rg["final_cost"] - ro["final_cost"]) <= 1e-6 * ro["final_cost"]
    live = removed_o == 0
    assert_params_close(dict(poses=g.poses, intrinsics=g.intrinsics, points=g.points[live]),
                        dict(poses=b.poses, intrinsics=b.intrinsics, points=b.points[live]))
    # filtered points keep the coordinates of the first solve (nothing references them any more)
    assert rel_err(g.points[~live], a.points[~live]) < 1e-6
    m = ~np.isnan(e2o)
    assert np.array_equal(m, ~np.isnan(eg)) and not m[~live].any()
    assert rel_err(eg[m], e2o[m]) < 1e-6


def test_filter_keep_set_constant_blocks_and_second_filter(mavba, oracle):
    """keep_point3D_ids (control points survive any error), GCP-style constant points seen by a FIXED image with
    refine_camera_params = false (their dropped residual blocks and fixed cost leave with them), and a second,
    stricter filter on the same session."""
    p = synth.make_scene(num_images=8, num_points=600, track_len=4, models=[A.MODEL_PINHOLE], seed=44, refine_camera_params=False)
    seen0 = np.unique(p.obs_point[p.obs_image == 0])
    p.point_const[seen0[:30]] = 1
    rng = np.random.default_rng(1)
    p.points[seen0[:30]] += rng.normal(0, 0.3, (30, 3))  # bad control points: large errors, some only on dropped blocks
    keep = np.zeros(p.num_points, np.uint8)
    keep[seen0[:10]] = 1
    opts = global_opts()
    with mavba.Session(p, opts) as s:
        s.solve()
        for max_error in (2.0, 0.9):
            res_before = s.result()
            removed, errs = s.filter_points(max_error, keep)
            # the oracle: same parameters, same decision rule, fresh problem without the filtered points
            cur = p.copy()
            cur.poses, cur.intrinsics, cur.points = s.get_params()
            q = _drop_points(cur, removed)
            ro, eo = oracle.solve(q, oracle.options(**opts), want_point_errors=True)
            rg = s.solve()
            assert removed[keep.astype(bool)].sum() == 0 and removed.sum() > 0
            assert rg["termination"] == ro["termination"] and rg["num_successful_steps"] == ro["num_successful_steps"]
            for k in ("num_residuals", "num_residuals_reduced", "num_parameters_reduced"):
                assert rg[k] == ro[k], k
            assert abs(rg["fixed_cost"] - ro["fixed_cost"]) <= 1e-9 * max(ro["fixed_cost"], 1e-300)
            assert abs(rg["final_cost"] - ro["final_cost"]) <= 1e-6 * ro["final_cost"]
            poses, intr, pts = s.get_params()
            assert_params_close(dict(poses=poses, intrinsics=intr, points=pts[removed == 0]),
                                dict(poses=q.poses, intrinsics=q.intrinsics, points=q.points[removed == 0]))
            eg = s.point_errors()
            m = ~np.isnan(eo)
            assert np.array_equal(m, ~np.isnan(eg)) and rel_err(eg[m], eo[m]) < 1e-6
            assert res_before["num_residuals"] > rg["num_residuals"]


def test_restart_and_set_params_resolve_without_setup(mavba, oracle):
    """A resident session re-solves the same problem from new parameter values (what the next bundle_adjustment()
    call on unchanged topology needs) and gives what a fresh session gives."""
    p = synth.make_config("C2", scale=0.05, seed=51)
    with mavba.Session(p, global_opts()) as s:
        s.solve()
        rng = np.random.default_rng(2)
        q = p.copy()
        q.poses[2:, 3:] += rng.normal(0, 0.2, q.poses[2:, 3:].shape)
        q.points += rng.normal(0, 0.2, q.points.shape)
        s.set_params(q.poses, q.intrinsics, q.points)
        s.restart()
        r_res = s.solve()
        x_res = s.get_params()
    with mavba.Session(q, global_opts()) as f:
        r_new = f.solve()
        x_new = f.get_params()
    assert r_res["final_cost"] == r_new["final_cost"] and r_res["num_successful_steps"] == r_new["num_successful_steps"]
    for u, v in zip(x_res, x_new):
        assert np.array_equal(u, v)
