"""The drop-in shim (shim/base3d/bundle_adjustment.{h,cc}): FeatureManager -> flat problem.

CPU tests link the shim against a RECORDING MOCK of the C ABI and compare what the shim hands
over with an independent Python restatement of the reference's problem construction
(reference src/base3d/bundle_adjustment.cc:228-387, 459-549). The GPU test links the real
library and runs the reference-signature bundle_adjustment() end to end.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from mavmap_amd import _abi as A
from mavmap_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dp, ip, bp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)


REFERENCE_SRC = "/root/reference/src"   # build container only (never present on the GPU box)


def _build(real, reference_fm=False):
    """reference_fm: compile against the REFERENCE's own src/fm/feature_management.{h,cc} (the real FeatureManager class,
    its sources used where they lie, only the Eigen vector types come from the test double) instead of the 26-line
    FeatureManager double in tests/stubs."""
    out = os.path.join(ROOT, "tests", "shim", "_shim_real.so" if real else "_shim_mock.so")
    if reference_fm:  # anything compiled from the reference's sources lives under oracle/_ref (git-ignored)
        os.makedirs(os.path.join(ROOT, "oracle", "_ref"), exist_ok=True)
        out = os.path.join(ROOT, "oracle", "_ref", "_shim_mock_reffm.so")
    srcs = [os.path.join(ROOT, "tests", "shim", "shim_driver.cpp"), os.path.join(ROOT, "shim", "base3d", "bundle_adjustment.cc")]
    inc = ["-I" + os.path.join(ROOT, "shim")]
    if reference_fm:  # shim/ first (its base3d/bundle_adjustment.h replaces the reference's), then the reference tree for fm/
        inc.append("-I" + REFERENCE_SRC)
        srcs.append(os.path.join(REFERENCE_SRC, "fm", "feature_management.cc"))
    cmd = ["g++", "-std=c++11", "-O1", "-pthread", "-fPIC", "-shared"] + inc + ["-I" + os.path.join(ROOT, "tests", "stubs"),
           "-I" + os.path.join(ROOT, "include")] + srcs
    if real:
        libdir = os.path.join(ROOT, "mavmap_amd", "lib")
        cmd += ["-L" + libdir, "-lmavba", "-Wl,-rpath," + libdir]
    else:
        cmd += [os.path.join(ROOT, "tests", "shim", "mock_mavba.cpp")]
    subprocess.check_call(cmd + ["-o", out])
    return C.CDLL(out)


def _mock_library(reference_fm):
    L = _build(real=False, reference_fm=reference_fm)
    for name, rt in (("mock_poses", dp), ("mock_intr", dp), ("mock_points", dp), ("mock_uv", dp), ("mock_prior_rvec", dp),
                     ("mock_pose_const", bp), ("mock_intr_const", bp), ("mock_point_const", bp),
                     ("mock_image_camera", ip), ("mock_camera_model", ip), ("mock_obs_image", ip),
                     ("mock_obs_point", ip), ("mock_prior_image", ip)):
        getattr(L, name).restype = rt
    L.mock_options.restype = C.POINTER(A.COptions)
    L.mock_prior_weight.restype = C.c_double
    L.shim_last_exception.restype = C.c_char_p
    return L


@pytest.fixture(scope="module")
def mock():
    return _mock_library(False)


@pytest.fixture(scope="module")
def mock_reference_fm():
    """The shim compiled against the reference's real FeatureManager (header AND feature_management.cc)."""
    if not os.path.isdir(os.path.join(REFERENCE_SRC, "fm")):
        pytest.skip("needs /root/reference (build container only)")
    return _mock_library(True)


class Scene:
    """A FeatureManager-shaped scene in flat arrays (ids are index + 1 on the C++ side)."""

    def __init__(self, prob, extra_unmatched=0, rng=None):
        self.cam = np.zeros((prob.num_cameras, 10))
        self.cam[:, :9] = prob.intrinsics
        self.cam[:, 9] = prob.camera_model
        self.img_cam = prob.image_camera.copy()
        self.poses = prob.poses.copy()
        self.points = prob.points.copy()
        # feature-manager order: observations grouped by image in ascending image id
        order = np.argsort(prob.obs_image, kind="stable")
        self.obs_img = prob.obs_image[order].astype(np.int32)
        self.obs_pt = prob.obs_point[order].astype(np.int32)
        self.obs_uv = prob.obs_uv[order].copy()
        if extra_unmatched:
            # 2-D points without a 3-D point (point2D_to_point3D has no entry)
            rng = rng or np.random.default_rng(0)
            at = np.sort(rng.choice(len(self.obs_img), extra_unmatched, replace=False))
            self.obs_img = np.insert(self.obs_img, at, self.obs_img[at])
            self.obs_pt = np.insert(self.obs_pt, at, -1)
            self.obs_uv = np.insert(self.obs_uv, at, 7.0, axis=0)


def run(L, sc, free, fixed, fixed_x, gcp=(), rot=None, **o):
    opt = dict(max_num_iterations=100, function_tolerance=1e-4, gradient_tolerance=1e-8, update_point3D_errors=0,
               min_track_len=2, loss_scale_factor=1.0, constrain_rotation=0, constrain_rotation_weight=0.0,
               refine_camera_params=0, print_summary=0)
    opt.update(o)
    cam, poses, points = sc.cam.copy(), sc.poses.copy(), sc.points.copy()
    perr = np.full(len(points), np.nan)
    ret = C.c_double()
    arr = lambda x: np.ascontiguousarray(x, dtype=np.int32)  # noqa: E731
    fr, fx, fxx, g = arr(free), arr(fixed), arr(fixed_x), arr(gcp)
    rotp = None if rot is None else np.ascontiguousarray(rot, dtype=np.float64)
    rc = L.shim_bundle_adjustment(
        C.c_int(len(cam)), cam.ctypes.data_as(dp), C.c_int(len(poses)), sc.img_cam.ctypes.data_as(ip),
        poses.ctypes.data_as(dp), C.c_int(len(points)), points.ctypes.data_as(dp), C.c_longlong(len(sc.obs_img)),
        sc.obs_img.ctypes.data_as(ip), sc.obs_pt.ctypes.data_as(ip), sc.obs_uv.ctypes.data_as(dp),
        C.c_int(len(fr)), fr.ctypes.data_as(ip), C.c_int(len(fx)), fx.ctypes.data_as(ip), C.c_int(len(fxx)),
        fxx.ctypes.data_as(ip), C.c_int(len(g)), g.ctypes.data_as(ip), None if rotp is None else rotp.ctypes.data_as(dp),
        C.c_int(opt["max_num_iterations"]), C.c_double(opt["function_tolerance"]), C.c_double(opt["gradient_tolerance"]),
        C.c_int(opt["update_point3D_errors"]), C.c_int(opt["min_track_len"]), C.c_double(opt["loss_scale_factor"]),
        C.c_int(opt["constrain_rotation"]), C.c_double(opt["constrain_rotation_weight"]), C.c_int(opt["refine_camera_params"]),
        C.c_int(opt["print_summary"]), perr.ctypes.data_as(dp), C.byref(ret))
    return rc, ret.value, cam, poses, points, perr


def expected_flat(sc, free, fixed, fixed_x, gcp, min_track_len, refine):
    """Independent restatement of the reference's selection rules (see module docstring)."""
    sel = list(free) + list(fixed_x) + list(fixed)
    count = {}
    for o in range(len(sc.obs_img)):
        if sc.obs_img[o] in sel and sc.obs_pt[o] >= 0:
            count[sc.obs_pt[o]] = count.get(sc.obs_pt[o], 0) + 1
    images, cams, pts, obs = [], [], [], []
    pose_const, intr_const = [], {}
    for state, lst in ((0, free), (1, fixed), (2, fixed_x)):
        for img in lst:
            mine = [o for o in range(len(sc.obs_img)) if sc.obs_img[o] == img and sc.obs_pt[o] >= 0
                    and count[sc.obs_pt[o]] >= min_track_len]
            if not mine:
                continue
            c = int(sc.img_cam[img])
            if c not in cams:
                cams.append(c)
                intr_const[c] = 0
            images.append(img)
            for o in mine:
                if sc.obs_pt[o] not in pts:
                    pts.append(int(sc.obs_pt[o]))
                obs.append((len(images) - 1, pts.index(sc.obs_pt[o]), tuple(sc.obs_uv[o])))
            many = len(mine) > 1
            pose_const.append({0: 0, 1: A.CONST_POSE, 2: A.CONST_TX}[state] if many else 0)
            if many and not refine:
                intr_const[c] = 1
    return dict(images=images, cams=cams, pts=pts, obs=obs, pose_const=pose_const,
                intr_const=[intr_const[c] for c in cams], point_const=[1 if p in gcp else 0 for p in pts])


def recorded(L):
    sz = (C.c_int64 * 6)()
    L.mock_sizes(sz)
    ni, nc, npt, no, npri, wantpe = list(sz)
    g = lambda f, n: np.ctypeslib.as_array(f(), shape=(n,)).copy() if n else np.zeros(0)  # noqa: E731
    return dict(ni=ni, nc=nc, np=npt, no=no, npri=npri, wantpe=wantpe,
                poses=g(L.mock_poses, ni * 6).reshape(-1, 6), pose_const=g(L.mock_pose_const, ni),
                image_camera=g(L.mock_image_camera, ni), intr=g(L.mock_intr, nc * 9).reshape(-1, 9),
                camera_model=g(L.mock_camera_model, nc), intr_const=g(L.mock_intr_const, nc),
                points=g(L.mock_points, npt * 3).reshape(-1, 3), point_const=g(L.mock_point_const, npt),
                uv=g(L.mock_uv, no * 2).reshape(-1, 2), obs_image=g(L.mock_obs_image, no), obs_point=g(L.mock_obs_point, no),
                prior_image=g(L.mock_prior_image, npri), prior_rvec=g(L.mock_prior_rvec, npri * 3).reshape(-1, 3))


def check_against_expected(L, sc, exp):
    r = recorded(L)
    assert (r["ni"], r["nc"], r["np"], r["no"]) == (len(exp["images"]), len(exp["cams"]), len(exp["pts"]), len(exp["obs"]))
    assert np.array_equal(r["poses"], sc.poses[exp["images"]])
    assert list(r["pose_const"]) == exp["pose_const"]
    assert list(r["image_camera"]) == [exp["cams"].index(int(sc.img_cam[i])) for i in exp["images"]]
    assert np.array_equal(r["intr"], sc.cam[exp["cams"], :9])
    assert list(r["camera_model"]) == [int(sc.cam[c, 9]) for c in exp["cams"]]
    assert list(r["intr_const"]) == exp["intr_const"]
    assert np.array_equal(r["points"], sc.points[exp["pts"]])
    assert list(r["point_const"]) == exp["point_const"]
    assert list(r["obs_image"]) == [o[0] for o in exp["obs"]]
    assert list(r["obs_point"]) == [o[1] for o in exp["obs"]]
    assert np.array_equal(r["uv"], np.array([o[2] for o in exp["obs"]]).reshape(-1, 2))
    return r


def small_scene(seed=1, **kw):
    p = synth.make_scene(num_images=6, num_points=60, track_len=3, models=[A.MODEL_PINHOLE, A.MODEL_OPENCV], seed=seed, **kw)
    return p


def test_global_ba_flattening_order_and_constancy(mock):
    p = small_scene()
    sc = Scene(p, extra_unmatched=9)
    free, fixed, fixed_x = [2, 3, 4, 5], [0], [1]
    for refine in (0, 1):
        rc, ret, cam, poses, points, perr = run(mock, sc, free, fixed, fixed_x, refine_camera_params=refine,
                                                update_point3D_errors=1, max_num_iterations=200,
                                                function_tolerance=1e-6, gradient_tolerance=1e-10, loss_scale_factor=2.0)
        assert rc == 0
        exp = expected_flat(sc, free, fixed, fixed_x, (), 2, refine)
        r = check_against_expected(mock, sc, exp)
        assert exp["images"] == [2, 3, 4, 5, 0, 1]          # FREE, FIXED, FIXED_X
        assert list(r["pose_const"]) == [0, 0, 0, 0, A.CONST_POSE, A.CONST_TX]
        assert list(r["intr_const"]) == [1 - refine] * 2
        o = mock.mock_options().contents
        assert (o.max_num_iterations, o.function_tolerance, o.gradient_tolerance, o.loss_scale_factor) == (200, 1e-6, 1e-10, 2.0)
        assert o.update_point_errors == 1 and r["wantpe"] == 1
        # return value = sqrt(final_cost / num_residuals) (bundle_adjustment.cc:610); the mock reports 2/residual
        assert abs(ret - np.sqrt(2.0)) < 1e-15
        # point3D_errors: exactly the points of the problem get an entry (mock writes 100 + flat index)
        inprob = np.zeros(len(sc.points), bool)
        inprob[exp["pts"]] = True
        assert np.array_equal(~np.isnan(perr), inprob)
        assert np.array_equal(perr[exp["pts"]], 100.0 + np.arange(len(exp["pts"])))
        # nothing moved (the mock solves nothing) and the model code slot is intact
        assert np.array_equal(poses, sc.poses) and np.array_equal(points, sc.points) and np.array_equal(cam, sc.cam)


def test_shim_dumps_a_replay_file_of_what_it_hands_over(mock, tmp_path, monkeypatch):
    """With MAVBA_DUMP_DIR set the shim writes the flattened problem of every call; BAProblem.load reads back exactly
    what the C ABI received (so `bench.py --problem` replays real MAVMAP problems)."""
    import glob
    from mavmap_amd.problem import BAProblem
    monkeypatch.setenv("MAVBA_DUMP_DIR", str(tmp_path))
    p = small_scene(seed=4)
    sc = Scene(p, extra_unmatched=5)
    rc, *_ = run(mock, sc, [2, 3, 4, 5], [0], [1], refine_camera_params=1, max_num_iterations=77,
                 function_tolerance=1e-5, gradient_tolerance=1e-9, loss_scale_factor=1.5)
    assert rc == 0
    r = recorded(mock)
    files = sorted(glob.glob(str(tmp_path / "mavba_problem_*.bin")))
    assert len(files) == 1 and files[0].endswith("_6img.bin")
    q, o = BAProblem.load(files[0])
    assert (q.num_images, q.num_cameras, q.num_points, q.num_obs) == (r["ni"], r["nc"], r["np"], r["no"])
    assert np.array_equal(q.poses, r["poses"]) and np.array_equal(q.intrinsics, r["intr"]) and np.array_equal(q.points, r["points"])
    assert np.array_equal(q.obs_uv, r["uv"]) and list(q.obs_image) == list(r["obs_image"]) and list(q.obs_point) == list(r["obs_point"])
    assert list(q.pose_const) == list(r["pose_const"]) and list(q.intr_const) == list(r["intr_const"])
    assert list(q.point_const) == list(r["point_const"]) and list(q.image_camera) == list(r["image_camera"])
    assert list(q.camera_model) == list(r["camera_model"])
    assert o == dict(max_num_iterations=77, function_tolerance=1e-5, gradient_tolerance=1e-9, loss_scale_factor=1.5)


def test_local_window_min_track_len_counts_inside_the_selected_images(mock):
    p = small_scene(seed=2)
    sc = Scene(p)
    free, fixed, fixed_x = [2, 3], [0, 1], []
    rc, *_ = run(mock, sc, free, fixed, fixed_x)
    assert rc == 0
    exp = expected_flat(sc, free, fixed, fixed_x, (), 2, 0)
    check_against_expected(mock, sc, exp)
    n_sel = int(np.isin(sc.obs_img, free + fixed).sum())
    assert 0 < len(exp["obs"]) < n_sel            # some single-view points were dropped
    rc, *_ = run(mock, sc, free, fixed, fixed_x, min_track_len=3)
    assert rc == 0
    check_against_expected(mock, sc, expected_flat(sc, free, fixed, fixed_x, (), 3, 0))


def test_single_residual_image_keeps_its_blocks_variable(mock):
    """bundle_adjustment.cc:361 — constancy is applied only when num_residuals > 1."""
    p = small_scene(seed=3)
    sc = Scene(p)
    # leave FIXED image 0 with exactly one usable observation; give it a camera of its own
    keep = np.ones(len(sc.obs_img), bool)
    idx0 = np.nonzero(sc.obs_img == 0)[0]
    keep[idx0[1:]] = False
    sc.obs_img, sc.obs_pt, sc.obs_uv = sc.obs_img[keep], sc.obs_pt[keep], sc.obs_uv[keep]
    sc.cam = np.vstack([sc.cam, sc.cam[0]])
    sc.img_cam = sc.img_cam.copy(); sc.img_cam[0] = 2
    free, fixed, fixed_x = [3, 4, 5], [0, 1], [2]
    rc, *_ = run(mock, sc, free, fixed, fixed_x, refine_camera_params=0)
    assert rc == 0
    exp = expected_flat(sc, free, fixed, fixed_x, (), 2, 0)
    r = check_against_expected(mock, sc, exp)
    i0 = exp["images"].index(0)
    assert r["pose_const"][i0] == 0                                   # FIXED image, one residual: stays free
    assert r["intr_const"][exp["cams"].index(2)] == 0                 # and so does its private camera
    assert r["pose_const"][exp["images"].index(1)] == A.CONST_POSE


def test_gcp_points_and_validation(mock):
    p = small_scene(seed=4)
    sc = Scene(p)
    free, gcp = [2, 3, 4, 5], [0, 5, 7]
    # adjust_global_bundle_gcp: images 0 and 1 are in no list; 3 GCPs = 9 fixed parameters
    rc, *_ = run(mock, sc, free, [], [], gcp=gcp, refine_camera_params=1)
    assert rc == 0
    exp = expected_flat(sc, free, [], [], set(gcp), 2, 1)
    r = check_against_expected(mock, sc, exp)
    assert sum(r["point_const"]) == len([g for g in gcp if g in exp["pts"]]) > 0
    assert 0 not in exp["images"] and 1 not in exp["images"]
    n = mock.mock_calls()
    rc, *_ = run(mock, sc, free, [0], [])          # 6 fixed parameters < 7
    assert rc == 1 and b"At least 7 parameters" in mock.shim_last_exception()
    rc, *_ = run(mock, sc, free, [0], [1], min_track_len=1)
    assert rc == 1 and b"Minimum track length" in mock.shim_last_exception()
    assert mock.mock_calls() == n                   # rejected before reaching the backend


def test_rotation_constraints_prerotate_the_whole_scene_and_add_priors(mock):
    from scipy.spatial.transform import Rotation
    p = small_scene(seed=5)
    sc = Scene(p)
    rng = np.random.default_rng(0)
    rot = np.array([(Rotation.from_rotvec(rng.normal(0, 0.05, 3)) * Rotation.from_rotvec(w)).as_rotvec() for w in sc.poses[:, :3]])
    free, fixed, fixed_x = [2, 3, 4], [0], [1]     # image 5 is in no list but must be rotated too
    rc, ret, cam, poses, points, _ = run(mock, sc, free, fixed, fixed_x, rot=rot, constrain_rotation=1,
                                         constrain_rotation_weight=3.5)
    assert rc == 0
    R_fm = Rotation.from_rotvec(sc.poses[0, :3]).as_matrix()
    R_c = Rotation.from_rotvec(rot[0]).as_matrix()
    S = R_fm.T @ R_c                               # bundle_adjustment.cc:404-412
    for i in range(len(poses)):
        R_old = Rotation.from_rotvec(sc.poses[i, :3]).as_matrix()
        R_new = Rotation.from_rotvec(poses[i, :3]).as_matrix()
        assert np.abs(R_new - R_old @ S.T).max() < 1e-12       # transform_pose: [R|t] S^-1
        assert np.array_equal(poses[i, 3:], sc.poses[i, 3:])
    assert np.abs(points - sc.points @ S.T).max() < 1e-12       # transform_point
    r = recorded(mock)
    assert r["npri"] == 3 and mock.mock_prior_weight() == 3.5
    exp = expected_flat(sc, free, fixed, fixed_x, (), 2, 0)
    assert list(r["prior_image"]) == [exp["images"].index(i) for i in free]
    assert np.array_equal(r["prior_rvec"], rot[free])
    # the flat problem holds the ROTATED values
    assert np.abs(r["points"] - points[exp["pts"]]).max() == 0.0
    rot_missing = rot.copy(); rot_missing[3] = np.nan
    rc, *_ = run(mock, sc, free, fixed, fixed_x, rot=rot_missing, constrain_rotation=1)
    assert rc == 2                                   # .at() on a missing constraint, like the reference


@pytest.mark.gpu
def test_shim_end_to_end_on_gpu(mavba, oracle):
    """Reference-signature bundle_adjustment() through the real library == flat API == oracle."""
    L = _build(real=True)
    L.shim_last_exception.restype = C.c_char_p
    p = synth.make_scene(num_images=8, num_points=600, track_len=4, models=[A.MODEL_PINHOLE, A.MODEL_OPENCV], seed=61)
    sc = Scene(p, extra_unmatched=25)
    free, fixed, fixed_x = list(range(2, 8)), [0], [1]
    opts = dict(max_num_iterations=200, function_tolerance=1e-6, gradient_tolerance=1e-10)
    rc, ret, cam, poses, points, perr = run(L, sc, free, fixed, fixed_x, refine_camera_params=1,
                                            update_point3D_errors=1, **opts)
    assert rc == 0, L.shim_last_exception()
    q = p.copy()
    ro, eo = oracle.solve(q, oracle.options(**opts), want_point_errors=True)
    assert abs(ret - np.sqrt(ro["final_cost"] / ro["num_residuals"])) < 1e-6 * ret
    assert np.abs(poses - q.poses).max() < 1e-6 * np.abs(q.poses).max()
    assert np.abs(points - q.points).max() < 1e-6 * np.abs(q.points).max()
    assert np.abs(cam[:, :9] - q.intrinsics).max() < 1e-6 * np.abs(q.intrinsics).max()
    assert np.array_equal(cam[:, 9], sc.cam[:, 9])
    assert np.abs(perr - eo).max() < 1e-6 * np.abs(eo).max()
    # pose_refinement through the shim
    sel = p.obs_image == 3
    uv, xyz = np.ascontiguousarray(p.obs_uv[sel]), np.ascontiguousarray(q.points[p.obs_point[sel]])
    rvec, tvec = p.poses[3, :3].copy(), p.poses[3, 3:].copy()
    c10 = np.zeros(10); c10[:9] = q.intrinsics[p.image_camera[3]]; c10[9] = p.camera_model[p.image_camera[3]]
    mask = np.ones(len(uv), np.uint8)
    out = C.c_double()
    rc = L.shim_pose_refinement(rvec.ctypes.data_as(dp), tvec.ctypes.data_as(dp), c10.ctypes.data_as(dp),
                                C.c_longlong(len(uv)), uv.ctypes.data_as(dp), xyz.ctypes.data_as(dp),
                                mask.ctypes.data_as(bp), C.c_double(1.0), C.byref(out))
    assert rc == 0, L.shim_last_exception()
    assert np.abs(np.concatenate([rvec, tvec]) - q.poses[3]).max() < 5e-2   # single-image robust refit stays at the BA pose


@pytest.mark.gpu
def test_shim_reaches_several_ranks_with_mavba_gpus(mavba, oracle, monkeypatch):
    """The reference-signature bundle_adjustment() with MAVBA_GPUS set: the unchanged call site (sequential_mapper.cc:1074)
    ends up in the in-process multi-rank solve. (One GPU here: the ranks share it.)"""
    L = _build(real=True)
    L.shim_last_exception.restype = C.c_char_p
    p = synth.make_scene(num_images=10, num_points=900, track_len=4, models=[A.MODEL_PINHOLE, A.MODEL_OPENCV], seed=67)
    free, fixed, fixed_x = list(range(2, 10)), [0], [1]
    opts = dict(max_num_iterations=200, function_tolerance=1e-6, gradient_tolerance=1e-10)
    rc, ret1, cam1, poses1, points1, perr1 = run(L, Scene(p, extra_unmatched=10), free, fixed, fixed_x, refine_camera_params=1,
                                                  update_point3D_errors=1, **opts)
    assert rc == 0, L.shim_last_exception()
    monkeypatch.setenv("MAVBA_GPUS", "2")
    monkeypatch.setenv("MAVBA_GPUS_SAME_DEVICE", "1")
    monkeypatch.setenv("MAVBA_GPUS_MIN_OBS", "0")
    rc, ret2, cam2, poses2, points2, perr2 = run(L, Scene(p, extra_unmatched=10), free, fixed, fixed_x, refine_camera_params=1,
                                                  update_point3D_errors=1, **opts)
    assert rc == 0, L.shim_last_exception()
    assert abs(ret2 - ret1) < 1e-9 * ret1
    assert np.abs(poses2 - poses1).max() < 1e-8 * np.abs(poses1).max()
    assert np.abs(points2 - points1).max() < 1e-8 * np.abs(points1).max()
    assert np.abs(cam2 - cam1).max() < 1e-8 * np.abs(cam1).max()
    assert np.abs(perr2 - perr1).max() < 1e-8 * np.abs(perr1).max()


def test_print_report_matches_the_reference_layout(mock, capfd):
    """_print_report (reference src/base3d/bundle_adjustment.cc:114-136) + the header of :604-607, character for
    character: labels right-aligned in 18 columns, counts left-aligned, costs sqrt(cost / num_residuals) at 6
    significant digits followed by ' [px]', one empty line at the end."""
    sc = Scene(small_scene(seed=9))
    capfd.readouterr()
    rc, ret, *_ = run(mock, sc, [2, 3, 4, 5], [0], [1], print_summary=1)
    assert rc == 0
    out = capfd.readouterr().out
    r = recorded(mock)
    # the mock reports zero reduced counts / steps and cost = 2 * num_residuals -> sqrt(2) = 1.41421
    assert out == ("Bundle Adjustment Report\n"
                   "------------------------\n"
                   "      Residuals : 0\n"
                   "     Parameters : 0\n"
                   "     Iterations : 0\n"
                   "   Initial cost : 1.41421 [px]\n"
                   "     Final cost : 1.41421 [px]\n"
                   "\n"), out
    assert abs(ret - np.sqrt(2.0)) < 1e-15 and r["no"] > 0
    rc, *_ = run(mock, sc, [2, 3, 4, 5], [0], [1], print_summary=0)
    assert capfd.readouterr().out == ""


def test_an_image_id_listed_twice_is_one_set_of_blocks(mock):
    """Image 3 in the free AND the fixed list: in the reference both loops add residual blocks on the SAME parameter
    blocks and the fixed list's SetParameterBlockConstant applies - one pose block, observations twice, FIXED."""
    sc = Scene(small_scene(seed=11))
    rc, *_ = run(mock, sc, [2, 3, 4], [0, 3], [1])
    assert rc == 0
    r = recorded(mock)
    exp_once = expected_flat(sc, [2, 3, 4], [0], [1], (), 2, 0)
    assert r["ni"] == len(exp_once["images"])                      # no second block for image 3
    i3 = exp_once["images"].index(3)
    assert r["pose_const"][i3] == A.CONST_POSE
    n3 = int(np.sum((sc.obs_img == 3) & (sc.obs_pt >= 0)))
    assert int(np.sum(r["obs_image"] == i3)) == 2 * n3              # its residual blocks were added by both loops


@pytest.mark.parametrize("seed", [3, 4, 5])
def test_block_wise_assembly_hands_over_the_serial_walks_problem(mock, monkeypatch, seed):
    """Round 6: big calls assemble the flat problem on the worker threads - blocks of entries contiguous in list order, a
    point's owner = the lowest block that sees it, local numbers + a prefix sum over the blocks = the numbers of the serial
    "first appearance" walk (shim/base3d/bundle_adjustment.cc). MAVBA_SHIM_ASSEMBLY forces either form: on scenes with
    unmatched 2-D points, a min_track_len that drops points, an image listed twice, GCPs and lists in scrambled order both
    must hand over the SAME arrays - and the independent restatement's."""
    rng = np.random.default_rng(seed)
    p = synth.make_scene(num_images=40, num_points=1500, track_len=4, models=[A.MODEL_PINHOLE, A.MODEL_OPENCV], seed=seed,
                         long_track_frac=0.05, long_track_len=9)
    sc = Scene(p, extra_unmatched=200, rng=rng)
    ids = rng.permutation(40)
    free, fixed, fixed_x = list(ids[:30]), list(ids[30:36]) + [int(ids[3])], list(ids[36:39])   # (one image in two lists; one unused)
    gcp = [5, 77, 300]
    out = {}
    for mode in ("serial", "blocks"):
        monkeypatch.setenv("MAVBA_SHIM_ASSEMBLY", mode)
        rc, *_ = run(mock, sc, free, fixed, fixed_x, gcp=gcp, min_track_len=3, refine_camera_params=1)
        assert rc == 0
        out[mode] = recorded(mock)
    a, b = out["serial"], out["blocks"]
    assert a.keys() == b.keys()
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    # and once against the restatement (no image listed twice there: its rule is tested by the test above)
    monkeypatch.setenv("MAVBA_SHIM_ASSEMBLY", "blocks")
    fixed1 = list(ids[30:36])
    rc, *_ = run(mock, sc, free, fixed1, fixed_x, gcp=gcp, min_track_len=3, refine_camera_params=1)
    assert rc == 0
    check_against_expected(mock, sc, expected_flat(sc, free, fixed1, fixed_x, gcp, 3, 1))


@pytest.mark.gpu
def test_shim_rotation_constraints_end_to_end_on_gpu(mavba, oracle):
    """constrain_rotation through the reference-signature call into the real library: global pre-rotation of the whole
    FeatureManager (src/base3d/bundle_adjustment.cc:399-425), one prior per FREE image (:428-444), solve; against
    the oracle on the equivalent flat problem (pre-rotated scene + rot_prior arrays)."""
    from scipy.spatial.transform import Rotation
    L = _build(real=True)
    L.shim_last_exception.restype = C.c_char_p
    p = synth.make_scene(num_images=8, num_points=600, track_len=4, models=[A.MODEL_PINHOLE], seed=63)
    sc = Scene(p)
    rng = np.random.default_rng(4)
    rot = np.array([(Rotation.from_rotvec(rng.normal(0, 0.01, 3)) * Rotation.from_rotvec(w)).as_rotvec() for w in sc.poses[:, :3]])
    free, fixed, fixed_x = list(range(2, 8)), [0], [1]
    opts = dict(max_num_iterations=200, function_tolerance=1e-6, gradient_tolerance=1e-10)
    rc, ret, cam, poses, points, perr = run(L, sc, free, fixed, fixed_x, rot=rot, constrain_rotation=1, constrain_rotation_weight=50.0,
                                            refine_camera_params=1, update_point3D_errors=1, **opts)
    assert rc == 0, L.shim_last_exception()
    # the oracle's flat problem: the scene after the pre-rotation S = R_fm^T R_c of the first fixed image
    S = Rotation.from_rotvec(sc.poses[0, :3]).as_matrix().T @ Rotation.from_rotvec(rot[0]).as_matrix()
    q = p.copy()
    q.poses[:, :3] = np.array([Rotation.from_matrix(Rotation.from_rotvec(w).as_matrix() @ S.T).as_rotvec() for w in sc.poses[:, :3]])
    q.points = np.ascontiguousarray(sc.points @ S.T)
    q.rot_prior_image = np.array(free, np.int32)
    q.rot_prior_rvec = np.ascontiguousarray(rot[free])
    q.rot_prior_weight = 50.0
    ro, eo = oracle.solve(q, oracle.options(**opts), want_point_errors=True)
    assert ro["num_residuals"] == 2 * p.num_obs + len(free)
    assert abs(ret - np.sqrt(ro["final_cost"] / ro["num_residuals"])) < 1e-6 * ret
    assert np.abs(poses - q.poses).max() < 1e-6 * np.abs(q.poses).max()
    assert np.abs(points - q.points).max() < 1e-6 * np.abs(q.points).max()
    assert np.abs(perr - eo).max() < 1e-6 * np.abs(eo).max()


def test_shim_is_source_compatible_with_the_reference_feature_manager(mock_reference_fm):
    """Build-container-only: shim/base3d/bundle_adjustment.{h,cc} compiled against the REFERENCE's
    src/fm/feature_management.h and linked with its feature_management.cc (sources used where they lie; only
    Eigen::Vector2d/3d come from the test double, there is no Eigen here) hands the same flat problem to the C ABI as the
    independent restatement of bundle_adjustment.cc:228-549 expects: global call with camera refinement, local window with
    min_track_len 3, GCPs, rotation constraints."""
    L = mock_reference_fm
    sc = Scene(small_scene(seed=4), extra_unmatched=7)
    for free, fixed, fixed_x, gcp, mtl, refine in (([2, 3, 4, 5], [0], [1], (), 2, 1), ([2, 3], [0, 1], [], (), 3, 0),
                                                   ([0, 1, 2, 3, 4, 5], [], [], (0, 5, 9), 2, 1)):
        rc, *_ = run(L, sc, free, fixed, fixed_x, gcp=gcp, min_track_len=mtl, refine_camera_params=refine)
        assert rc == 0, L.shim_last_exception()
        r = check_against_expected(L, sc, expected_flat(sc, free, fixed, fixed_x, gcp, mtl, refine))
        assert r["no"] > 0
    # the two std::invalid_argument conditions of the reference (:459-471) through the real class as well
    rc, *_ = run(L, sc, [1, 2, 3], [], [0])
    assert rc == 1 and b"7 parameters" in L.shim_last_exception()
