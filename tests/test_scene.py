"""The incremental scene (mavba_scene_*): a flat mirror of the FeatureManager that is fed
deltas and flattens a bundle_adjustment() call itself, instead of the shim's hash-map walk per call.

CPU tests: the flat problem of a call must be EXACTLY what the drop-in shim hands over for the same FeatureManager
content (the shim is itself checked against an independent restatement of reference
src/base3d/bundle_adjustment.cc:228-549 in test_shim.py), for the global / local-window / GCP / rotation-prior /
duplicate-id cases, and after incremental growth and point deletion. GPU tests: a scene grown image by image and
solved through mavba_scene_bundle_adjust equals the one-shot solve of the equivalent fresh problem and the oracle.
"""
import numpy as np
import pytest

from mavmap_amd import _abi as A
from mavmap_amd import api, synth
from tests.conftest import assert_params_close, global_opts, rel_err
from tests.test_shim import Scene as FmScene
from tests.test_shim import expected_flat, mock, recorded, run, small_scene  # noqa: F401  (mock is a fixture)


def mirror(fm, images=None):
    """A mavba Scene holding the same content as the test FeatureManager `fm` (ids = index + 1, like the shim driver);
    `images`: only these images and their 2-D points (for growing a scene step by step)."""
    sc = api.Scene()
    for c in range(len(fm.cam)):
        sc.set_camera(c + 1, int(fm.cam[c, 9]), fm.cam[c, :9])
    for p in range(len(fm.points)):
        sc.set_point3D(p + 1, fm.points[p])
    add_images(sc, fm, range(len(fm.poses)) if images is None else images)
    return sc


def add_images(sc, fm, images, bulk=False):
    for i in images:
        sc.set_image(i + 1, int(fm.img_cam[i]) + 1, fm.poses[i, :3], fm.poses[i, 3:])
        if bulk:  # one call per image (mavba_scene_add_points2d)
            oo = np.nonzero(fm.obs_img == i)[0]
            sc.add_points2D(i + 1, oo + 1, fm.obs_uv[oo], np.where(fm.obs_pt[oo] >= 0, fm.obs_pt[oo] + 1, -1))
            continue
        for o in np.nonzero(fm.obs_img == i)[0]:
            sc.add_point2D(i + 1, int(o) + 1, fm.obs_uv[o])
            if fm.obs_pt[o] >= 0:
                sc.link(int(o) + 1, int(fm.obs_pt[o]) + 1)


def ids(x):
    return [int(v) + 1 for v in x]


def same_as_shim(L, f):
    """The scene's flat problem `f` against what the shim handed to the recording mock."""
    r = recorded(L)
    assert (len(f["image_ids"]), len(f["camera_ids"]), len(f["point_ids"]), len(f["obs_image"])) == (r["ni"], r["nc"], r["np"], r["no"])
    assert np.array_equal(f["poses"], r["poses"]) and np.array_equal(f["intrinsics"], r["intr"])
    assert np.array_equal(f["points"], r["points"]) and np.array_equal(f["obs_uv"], r["uv"])
    for a, b in (("pose_const", "pose_const"), ("intr_const", "intr_const"), ("point_const", "point_const"),
                 ("image_camera", "image_camera"), ("camera_model", "camera_model"), ("obs_image", "obs_image"),
                 ("obs_point", "obs_point"), ("rot_prior_image", "prior_image")):
        assert list(f[a]) == list(r[b]), a
    assert np.array_equal(f["rot_prior_rvec"], r["prior_rvec"])


@pytest.mark.parametrize("refine", [0, 1])
def test_scene_flatten_equals_the_shim_for_a_global_call(mock, refine):  # noqa: F811
    fm = FmScene(small_scene(), extra_unmatched=9)
    free, fixed, fixed_x = [2, 3, 4, 5], [0], [1]
    rc, *_ = run(mock, fm, free, fixed, fixed_x, refine_camera_params=refine)
    assert rc == 0
    with mirror(fm) as sc:
        f = sc.flatten(ids(free), ids(fixed), ids(fixed_x), refine_camera_params=refine)
        same_as_shim(mock, f)
        exp = expected_flat(fm, free, fixed, fixed_x, (), 2, refine)
        assert list(f["image_ids"]) == ids(exp["images"]) and list(f["point_ids"]) == ids(exp["pts"])
        assert list(f["camera_ids"]) == ids(exp["cams"])


def test_scene_flatten_local_window_min_track_len_and_single_residual_rule(mock):  # noqa: F811
    fm = FmScene(small_scene(seed=2))
    for mtl in (2, 3):
        rc, *_ = run(mock, fm, [2, 3], [0, 1], [], min_track_len=mtl)
        assert rc == 0
        with mirror(fm) as sc:
            same_as_shim(mock, sc.flatten(ids([2, 3]), ids([0, 1]), [], min_track_len=mtl))
    # a FIXED image with exactly one usable observation keeps its blocks variable (bundle_adjustment.cc:361)
    fm = FmScene(small_scene(seed=3))
    keep = np.ones(len(fm.obs_img), bool)
    keep[np.nonzero(fm.obs_img == 0)[0][1:]] = False
    fm.obs_img, fm.obs_pt, fm.obs_uv = fm.obs_img[keep], fm.obs_pt[keep], fm.obs_uv[keep]
    fm.cam = np.vstack([fm.cam, fm.cam[0]])
    fm.img_cam = fm.img_cam.copy()
    fm.img_cam[0] = 2
    rc, *_ = run(mock, fm, [3, 4, 5], [0, 1], [2])
    assert rc == 0
    with mirror(fm) as sc:
        f = sc.flatten(ids([3, 4, 5]), ids([0, 1]), ids([2]))
        same_as_shim(mock, f)
        assert f["pose_const"][list(f["image_ids"]).index(1)] == 0 and f["intr_const"][list(f["camera_ids"]).index(3)] == 0


def test_scene_gcp_points_duplicate_ids_and_validation(mock):  # noqa: F811
    fm = FmScene(small_scene(seed=4))
    free, gcp = [2, 3, 4, 5], [0, 5, 7]
    rc, *_ = run(mock, fm, free, [], [], gcp=gcp, refine_camera_params=1)
    assert rc == 0
    with mirror(fm) as sc:
        f = sc.flatten(ids(free), [], [], gcp=ids(gcp), refine_camera_params=1)
        same_as_shim(mock, f)
        assert f["point_const"].sum() > 0
        # the two std::invalid_argument cases of the reference (bundle_adjustment.cc:459-471)
        with pytest.raises(api.MavbaError, match="At least 7 parameters") as e:
            sc.flatten(ids(free), ids([0]), [])
        assert e.value.code == A.ERR_INVALID_ARGUMENT
        with pytest.raises(api.MavbaError, match="Minimum track length"):
            sc.flatten(ids(free), ids([0]), ids([1]), min_track_len=1)
        with pytest.raises(api.MavbaError, match="unknown image id"):
            sc.flatten(ids(free), ids([0]), [99])
        # an image id in two lists is one set of parameter blocks whose constancy accumulates
        rc, *_ = run(mock, fm, [2, 3, 4], [0, 4], [1, 2])
        assert rc == 0
        same_as_shim(mock, sc.flatten(ids([2, 3, 4]), ids([0, 4]), ids([1, 2])))


def test_scene_rotation_constraints_prerotate_the_scene_like_the_shim(mock):  # noqa: F811
    from scipy.spatial.transform import Rotation
    fm = FmScene(small_scene(seed=5))
    rng = np.random.default_rng(0)
    rot = np.array([(Rotation.from_rotvec(rng.normal(0, 0.05, 3)) * Rotation.from_rotvec(w)).as_rotvec() for w in fm.poses[:, :3]])
    free, fixed, fixed_x = [2, 3, 4], [0], [1]
    rc, _, _, poses, points, _ = run(mock, fm, free, fixed, fixed_x, rot=rot, constrain_rotation=1, constrain_rotation_weight=3.5)
    assert rc == 0
    with mirror(fm) as sc:
        rotmap = {i + 1: rot[i] for i in range(len(rot))}
        f = sc.flatten(ids(free), ids(fixed), ids(fixed_x), rot=rotmap, constrain_rotation=1, constrain_rotation_weight=3.5)
        r = recorded(mock)
        # same rotation arithmetic up to the last bits (Eigen's quaternion route in the shim's FeatureManager stub vs
        # the library's own rotation-vector conversion)
        assert np.abs(f["poses"] - r["poses"]).max() < 1e-12 and np.abs(f["points"] - r["points"]).max() < 1e-12
        assert list(f["rot_prior_image"]) == list(r["prior_image"]) and np.array_equal(f["rot_prior_rvec"], r["prior_rvec"])
        assert f["rot_prior_weight"] == 3.5 and list(f["pose_const"]) == list(r["pose_const"])
        # image 6 is in no list and was rotated too; the points as well
        for i in range(len(poses)):
            rv, tv = sc.get_image(i + 1)
            assert np.abs(rv - poses[i, :3]).max() < 1e-12 and np.array_equal(tv, poses[i, 3:])
        assert np.abs(sc.get_point3D(8) - points[7]).max() < 1e-12
        with pytest.raises(api.MavbaError, match="no rotation constraint"):
            sc.flatten(ids(free), ids(fixed), ids(fixed_x), rot={k: v for k, v in rotmap.items() if k != 4}, constrain_rotation=1)


def test_scene_bulk_loading_equals_point_by_point(mock):  # noqa: F811
    fm = FmScene(small_scene(seed=7), extra_unmatched=8)
    rc, *_ = run(mock, fm, [2, 3, 4, 5], [0], [1])
    assert rc == 0
    with mirror(fm, images=[]) as sc:
        add_images(sc, fm, range(len(fm.poses)), bulk=True)
        same_as_shim(mock, sc.flatten(ids([2, 3, 4, 5]), ids([0]), ids([1])))
        with pytest.raises(api.MavbaError, match="added twice"):
            sc.add_points2D(1, [3], [[0.0, 0.0]])
        with pytest.raises(api.MavbaError, match="twice in one call"):
            sc.add_points2D(1, [9001, 9001], [[0.0, 0.0], [1.0, 1.0]])


def test_scene_grown_step_by_step_and_pruned_equals_a_fresh_mirror(mock):  # noqa: F811
    """The mapper's life cycle: images arrive one at a time, points get deleted by the filters in between; at any moment
    the flat problem equals the one of a FeatureManager holding the same content."""
    fm = FmScene(small_scene(seed=6), extra_unmatched=6)
    with mirror(fm, images=[0, 1, 2]) as sc:
        add_images(sc, fm, [3])
        add_images(sc, fm, [4, 5])
        rc, *_ = run(mock, fm, [2, 3, 4, 5], [0], [1])
        assert rc == 0
        same_as_shim(mock, sc.flatten(ids([2, 3, 4, 5]), ids([0]), ids([1])))
        # delete_point3D (reference FeatureManager::delete_point3D): the 2-D points stay, their links go
        dead = [3, 10, 11]
        for p in dead:
            sc.delete_point3D(p + 1)
        fm2 = FmScene.__new__(FmScene)
        fm2.__dict__.update(fm.__dict__)
        fm2.obs_pt = np.where(np.isin(fm.obs_pt, dead), -1, fm.obs_pt).astype(np.int32)
        rc, *_ = run(mock, fm2, [3, 4, 5], [0, 1], [2])
        assert rc == 0
        f = sc.flatten(ids([3, 4, 5]), ids([0, 1]), ids([2]))
        same_as_shim(mock, f)
        assert not set(ids(dead)) & set(f["point_ids"].tolist())
        with pytest.raises(api.MavbaError, match="unknown 3-D point id"):
            sc.get_point3D(4)
        # a track merge drops duplicate observations of the kept point (feature_management.cc:196-204): link(id, -1)
        o_drop = int(np.nonzero(fm2.obs_pt >= 0)[0][5])
        sc.link(o_drop + 1, -1)
        fm2.obs_pt = fm2.obs_pt.copy()
        fm2.obs_pt[o_drop] = -1
        rc, *_ = run(mock, fm2, [3, 4, 5], [0, 1], [2])
        assert rc == 0
        same_as_shim(mock, sc.flatten(ids([3, 4, 5]), ids([0, 1]), ids([2])))
        # a pose update from outside (e.g. after pose_refinement) is a set_image without camera change
        sc.set_image(6, -1, [0.1, 0.2, 0.3], [1.0, 2.0, 3.0])
        rv, tv = sc.get_image(6)
        assert list(rv) == [0.1, 0.2, 0.3] and list(tv) == [1.0, 2.0, 3.0]
        f = sc.flatten(ids([3, 4, 5]), ids([0, 1]), ids([2]))
        assert list(f["poses"][list(f["image_ids"]).index(6)]) == [0.1, 0.2, 0.3, 1.0, 2.0, 3.0]


# --------------------------------------------------------------------------------------------------------------------
# GPU: the scene drives the real solver


def _fm_from(p):
    return FmScene(p)


def _fresh_problem(f):
    from mavmap_amd.problem import BAProblem
    return BAProblem(poses=f["poses"], pose_const=f["pose_const"], image_camera=f["image_camera"], intrinsics=f["intrinsics"],
                     camera_model=f["camera_model"], intr_const=f["intr_const"], points=f["points"], point_const=f["point_const"],
                     obs_uv=f["obs_uv"], obs_image=f["obs_image"], obs_point=f["obs_point"],
                     rot_prior_image=f["rot_prior_image"], rot_prior_rvec=f["rot_prior_rvec"],
                     rot_prior_weight=f["rot_prior_weight"]).copy()       # (the constructor keeps views of f's arrays)


@pytest.mark.gpu
def test_scene_local_ba_after_each_new_image_matches_fresh_problems_and_oracle(mavba, oracle):
    """Sequential mapping in miniature: after every new image a local BA over the last 4 images (2 free, 2 fixed), results
    written back into the scene; each call equals mavba.bundle_adjustment on the equivalent fresh flat problem bit for
    bit, and the oracle to 1e-6; the scene's state carries over from call to call."""
    p = synth.make_scene(num_images=10, num_points=400, track_len=4, models=[A.MODEL_PINHOLE, A.MODEL_OPENCV], seed=11)
    fm = _fm_from(p)
    opts = dict(max_num_iterations=50, function_tolerance=1e-6, gradient_tolerance=1e-10)
    with mirror(fm, images=[0, 1, 2, 3]) as sc:
        for new in range(4, 10):
            add_images(sc, fm, [new])
            free, fixed = ids([new - 1, new]), ids([new - 3, new - 2])
            f = sc.flatten(free, fixed, [])
            q, qo = _fresh_problem(f), _fresh_problem(f)
            errs = {}
            cost, res = sc.bundle_adjustment(free, fixed, [], opts, point3D_errors=errs)
            e_fresh = np.full(q.num_points, np.nan)
            cost_fresh, res_fresh = mavba.bundle_adjustment(q, opts, point3D_errors=e_fresh)
            assert cost == cost_fresh and res["final_cost"] == res_fresh["final_cost"]
            assert res["num_successful_steps"] == res_fresh["num_successful_steps"] > 0
            for k, iid in enumerate(f["image_ids"]):
                rv, tv = sc.get_image(int(iid))
                assert np.array_equal(np.r_[rv, tv], q.poses[k])
            for k, pid in enumerate(f["point_ids"][:50]):
                assert np.array_equal(sc.get_point3D(int(pid)), q.points[k])
            assert sorted(errs) == sorted(f["point_ids"].tolist())
            assert np.array_equal(np.array([errs[int(i)] for i in f["point_ids"]]), e_fresh)
            ro, _ = oracle.solve(qo, oracle.options(**opts), jac_mode=1)
            assert res["termination"] == ro["termination"] and res["num_successful_steps"] == ro["num_successful_steps"]
            assert abs(res["final_cost"] - ro["final_cost"]) <= 1e-6 * ro["final_cost"]
            assert_params_close(q, qo)
        # the last call started from what the previous ones left behind
        assert not np.array_equal(sc.get_image(7)[0], fm.poses[6, :3])


@pytest.mark.gpu
def test_scene_global_ba_with_gcps_and_camera_refinement(mavba, oracle):
    p = synth.make_scene(num_images=12, num_points=600, track_len=4, models=[A.MODEL_OPENCV], seed=12)
    fm = _fm_from(p)
    with mirror(fm) as sc:
        free, gcp = ids(range(12)), ids([0, 17, 33, 250])
        f = sc.flatten(free, [], [], gcp=gcp, refine_camera_params=1)
        qo = _fresh_problem(f)
        assert f["point_const"].sum() == 4 and f["intr_const"].sum() == 0
        cost, res = sc.bundle_adjustment(free, [], [], global_opts(), gcp=gcp, refine_camera_params=1)
        ro, _ = oracle.solve(qo, oracle.options(**global_opts()), jac_mode=1)
        assert res["termination"] == ro["termination"]
        assert abs(res["final_cost"] - ro["final_cost"]) <= 1e-6 * ro["final_cost"]
        assert abs(cost - np.sqrt(ro["final_cost"] / ro["num_residuals"])) <= 1e-6 * cost
        model, params = sc.get_camera(1)
        assert model == A.MODEL_OPENCV and rel_err(params, qo.intrinsics[0, :len(params)]) < 1e-6
        for g in gcp:
            assert np.array_equal(sc.get_point3D(g), fm.points[g - 1])       # GCPs did not move


@pytest.mark.gpu
def test_device_resident_scene_equals_the_host_route_bit_for_bit(mavba, oracle, monkeypatch):
    """The device-resident route (2-D / 3-D points in HBM, selection + first-appearance numbering in kernels, session built
    from the arrays where they are, refined points scattered back) against the host route (flatten + mavba_solve) on
    twin scenes: a global call, then growth by new images, a deleted 3-D point, re-linked 2-D points, and two more calls
    (dirty-range uploads, the resident points of the first call feeding the second). Same costs, same step counts, the
    same poses / points / point errors to the last bit; and the oracle on the first call's flat problem."""
    p = synth.make_scene(num_images=26, num_points=2500, track_len=5, models=[A.MODEL_PINHOLE, A.MODEL_OPENCV], seed=41, spacing=5.0)
    fm = FmScene(p, extra_unmatched=40)
    first = list(range(20))
    opts = global_opts()
    out = {}
    for route in ("host", "device"):
        monkeypatch.setenv("MAVBA_SCENE", route)
        log = []
        with mirror(fm, images=first) as sc:
            free, fixed, fixed_x = ids(first[2:]), ids([first[0]]), ids([first[1]])
            f0 = sc.flatten(free, fixed, fixed_x, refine_camera_params=1)
            errs = {}
            cost, res = sc.bundle_adjustment(free, fixed, fixed_x, opts, point3D_errors=errs, refine_camera_params=1)
            log.append((cost, res, dict(errs), [np.r_[sc.get_image(i)] for i in ids(first)], [sc.get_point3D(int(q)) for q in f0["point_ids"]]))
            # growth: six more images, one 3-D point deleted, two 2-D points re-linked / unlinked
            add_images(sc, fm, range(20, 26), bulk=True)
            dead = int(f0["point_ids"][5])
            sc.delete_point3D(dead)
            oo = np.nonzero(fm.obs_pt >= 0)[0]
            sc.link(int(oo[10]) + 1, -1)
            sc.link(int(oo[11]) + 1, int(fm.obs_pt[oo[12]]) + 1)
            allimg = list(range(26))
            free, fixed, fixed_x = ids(allimg[2:]), ids([0]), ids([1])
            for rep in range(2):
                f1 = sc.flatten(free, fixed, fixed_x, min_track_len=3)
                errs = {}
                cost, res = sc.bundle_adjustment(free, fixed, fixed_x, opts, point3D_errors=errs, min_track_len=3)
                log.append((cost, res, dict(errs), [np.r_[sc.get_image(i)] for i in ids(allimg)], [sc.get_point3D(int(q)) for q in f1["point_ids"]]))
                assert dead not in f1["point_ids"]
        out[route] = (log, f0)
    (lh, f0h), (ld, f0d) = out["host"], out["device"]
    assert len(lh) == len(ld) == 3
    for call, ((ch, rh, eh, ph, xh), (cd, rd, ed, pd, xd)) in enumerate(zip(lh, ld)):
        assert ch == cd
        for k in ("termination", "num_successful_steps", "num_unsuccessful_steps", "final_cost", "initial_cost", "num_residuals",
                  "num_residuals_reduced", "num_parameters_reduced"):
            assert rh[k] == rd[k], k
        assert rh["num_successful_steps"] > 0 or call == 2   # (the repeat call starts at the previous call's minimum)
        assert sorted(eh) == sorted(ed) and all(eh[k] == ed[k] for k in eh)
        assert all(np.array_equal(a, b) for a, b in zip(ph, pd)) and all(np.array_equal(a, b) for a, b in zip(xh, xd))
    # the first call against the oracle on its flat problem
    qo = _fresh_problem(f0h)
    ro, _ = oracle.solve(qo, oracle.options(**opts), jac_mode=1)
    assert ld[0][1]["termination"] == ro["termination"] and ld[0][1]["num_successful_steps"] == ro["num_successful_steps"]
    assert abs(ld[0][1]["final_cost"] - ro["final_cost"]) <= 1e-6 * ro["final_cost"]
