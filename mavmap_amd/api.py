"""Python binding of the C ABI in include/mavba.h (ctypes; no torch types at the boundary).

Host-side mirror of the reference's bundle-adjustment interface
(reference src/base3d/bundle_adjustment.h:38-114, 212-230):

    BundleAdjustmentOptions      -> options dict / mavmap_amd.BundleAdjustmentOptions
    bundle_adjustment(...)       -> bundle_adjustment(problem, options)   (flat problem instead
                                    of FeatureManager; the C++ shim does the flattening for MAVMAP)
    pose_refinement(...)         -> pose_refinement(rvec, tvec, camera_params, points2D, points3D,
                                    inlier_mask, options)

There is no CPU fallback: without the built HIP library, or without a GPU, every compute call
raises MavbaError.
"""
import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

from . import _abi as A
from .problem import BAProblem

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libmavba.so")
_lib = None

EXPORTED_SYMBOLS = [
    "mavba_options_init", "mavba_last_error", "mavba_device_count", "mavba_solve", "mavba_pose_refine",
    "mavba_session_create", "mavba_session_destroy", "mavba_session_reset", "mavba_session_iterate",
    "mavba_session_result", "mavba_session_get_params", "mavba_session_point_errors",
    "mavba_session_set_allreduce", "mavba_session_eval_jacobian", "mavba_session_reduced_dim",
    "mavba_session_reduced_system", "mavba_session_linear_step", "mavba_session_time_jacobian", "mavba_session_time_front", "mavba_session_set_profiling",
    "mavba_session_kernel_stats", "mavba_session_get_info", "mavba_dense_spd_solve",
    "mavba_scene_create", "mavba_scene_destroy", "mavba_scene_set_camera", "mavba_scene_set_image", "mavba_scene_add_point2d",
    "mavba_scene_add_points2d", "mavba_scene_set_point3d", "mavba_scene_link", "mavba_scene_delete_point3d", "mavba_scene_get_image", "mavba_scene_get_point3d",
    "mavba_scene_get_camera", "mavba_scene_flatten", "mavba_scene_bundle_adjust",
    "mavba_rccl_unique_id", "mavba_session_set_rccl", "mavba_pose_refine_batch", "mavba_session_set_params", "mavba_session_restart", "mavba_session_filter_points", "mavba_solve_filter_solve",
    "mavba_debug_elimination_tree", "mavba_debug_radix_sort", "mavba_debug_lm_decide", "mavba_debug_chol_schedule",
    "mavba_debug_inproc_comms", "mavba_debug_upload_batch",
]

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32)


class MavbaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"mavba error {code}: {msg}")
        self.code = code


def lib_path():
    return _LIB_PATH


def load():
    """Load libmavba.so (raises MavbaError if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise MavbaError(A.ERR_NO_DEVICE, f"{_LIB_PATH} is missing: run `python -m mavmap_amd.build` "
                                          "(the HIP extension is mandatory, there is no CPU fallback)")
    L = C.CDLL(_LIB_PATH)
    dp, ip, bp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
    pp, op, rp = C.POINTER(A.CProblem), C.POINTER(A.COptions), C.POINTER(A.CResult)
    sp = C.c_void_p
    L.mavba_options_init.argtypes = [op]
    L.mavba_options_init.restype = None
    L.mavba_last_error.restype = C.c_char_p
    L.mavba_device_count.restype = C.c_int
    L.mavba_solve.argtypes = [pp, op, rp, dp]
    L.mavba_pose_refine.argtypes = [dp, dp, dp, C.c_int32, dp, dp, bp, C.c_int64, op, rp]
    L.mavba_session_create.argtypes = [pp, op, C.POINTER(sp)]
    L.mavba_session_destroy.argtypes = [sp]
    L.mavba_session_destroy.restype = None
    L.mavba_session_reset.argtypes = [sp]
    L.mavba_session_iterate.argtypes = [sp, C.c_int32, ip, ip]
    L.mavba_session_result.argtypes = [sp, rp]
    L.mavba_session_get_params.argtypes = [sp, dp, dp, dp]
    L.mavba_session_point_errors.argtypes = [sp, dp]
    L.mavba_session_set_allreduce.argtypes = [sp, ALLREDUCE_FN, C.c_void_p, C.c_int32, C.c_int32]
    L.mavba_session_eval_jacobian.argtypes = [sp, dp, dp, dp, dp, dp]
    L.mavba_session_reduced_dim.argtypes = [sp]
    L.mavba_session_reduced_system.argtypes = [sp, C.c_double, dp, dp]
    L.mavba_session_linear_step.argtypes = [sp, C.c_double, dp, dp, dp, dp]
    L.mavba_session_time_jacobian.argtypes = [sp, C.c_int32, C.POINTER(C.c_float)]
    L.mavba_session_time_front.argtypes = [sp, C.c_double, C.c_int32, C.POINTER(C.c_float)]
    L.mavba_session_set_profiling.argtypes = [sp, C.c_int32]
    L.mavba_session_kernel_stats.argtypes = [sp, C.POINTER(A.CKernelStat), C.c_int32]
    L.mavba_session_get_info.argtypes = [sp, C.POINTER(A.CSessionInfo)]
    L.mavba_dense_spd_solve.argtypes = [C.c_int32, dp, dp, dp, C.c_int32]
    i64p, i64 = C.POINTER(C.c_int64), C.c_int64
    sop = C.POINTER(A.CSceneOptions)
    L.mavba_scene_create.argtypes = [C.POINTER(sp)]
    L.mavba_scene_destroy.argtypes = [sp]
    L.mavba_scene_destroy.restype = None
    L.mavba_scene_set_camera.argtypes = [sp, i64, C.c_int32, dp]
    L.mavba_scene_set_image.argtypes = [sp, i64, i64, dp, dp]
    L.mavba_scene_add_point2d.argtypes = [sp, i64, i64, dp]
    L.mavba_scene_add_points2d.argtypes = [sp, i64, i64, i64p, dp, i64p]
    L.mavba_scene_set_point3d.argtypes = [sp, i64, dp]
    L.mavba_scene_link.argtypes = [sp, i64, i64]
    L.mavba_scene_delete_point3d.argtypes = [sp, i64]
    L.mavba_scene_get_image.argtypes = [sp, i64, dp, dp]
    L.mavba_scene_get_point3d.argtypes = [sp, i64, dp]
    L.mavba_scene_get_camera.argtypes = [sp, i64, ip, dp]
    lists = [i64p, i64, i64p, i64, i64p, i64, i64p, i64, i64p, dp, i64]
    L.mavba_scene_flatten.argtypes = [sp] + lists + [sop, pp, C.POINTER(i64p), C.POINTER(i64p), C.POINTER(i64p)]
    L.mavba_scene_bundle_adjust.argtypes = [sp] + lists + [sop, op, rp, dp, C.POINTER(i64p), C.POINTER(dp), i64p]
    L.mavba_rccl_unique_id.argtypes = [C.c_void_p]
    L.mavba_session_set_rccl.argtypes = [sp, C.c_void_p, C.c_int32, C.c_int32]
    L.mavba_pose_refine_batch.argtypes = [C.c_int32, C.POINTER(A.CPoseRefineItem), op, rp]
    L.mavba_session_set_params.argtypes = [sp, dp, dp, dp]
    L.mavba_session_restart.argtypes = [sp]
    L.mavba_session_filter_points.argtypes = [sp, C.c_double, bp, bp, dp, C.POINTER(C.c_int64)]
    L.mavba_solve_filter_solve.argtypes = [pp, op, C.c_double, bp, rp, rp, dp, bp, C.POINTER(C.c_int64)]
    L.mavba_debug_elimination_tree.argtypes = [C.c_int32, C.c_int32, C.c_int64, ip, ip, C.c_int32, ip, ip, C.c_int32]
    L.mavba_debug_radix_sort.argtypes = [C.c_int32, C.POINTER(C.c_uint32), C.c_int32, ip, C.c_int32]
    L.mavba_debug_lm_decide.argtypes = [C.c_int32, dp, dp, dp, C.c_int32]
    i32p, i64p = C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    L.mavba_debug_chol_schedule.argtypes = [C.c_int32, C.c_int32, i32p, i32p, i32p, C.c_int64, i32p, i32p, C.c_int32, dp,
                                            i32p, C.c_int64, i64p, i32p, C.c_int64, i64p, i32p]
    for f in EXPORTED_SYMBOLS:
        if f not in ("mavba_options_init", "mavba_last_error", "mavba_session_destroy", "mavba_scene_destroy", "mavba_debug_upload_batch"):
            getattr(L, f).restype = C.c_int
    _lib = L
    return L


def _check(rc):
    if rc != A.OK:
        raise MavbaError(rc, load().mavba_last_error().decode(errors="replace"))


def device_count():
    return int(load().mavba_device_count())


def _d(a):
    return A.ptr(a, C.c_double)


@dataclass
class BundleAdjustmentOptions:
    """Field-for-field mirror of the reference struct (bundle_adjustment.h:38-114)."""
    max_num_iterations: int = 100
    function_tolerance: float = 1e-4
    gradient_tolerance: float = 1e-8
    update_point3D_errors: bool = False
    min_track_len: int = 2
    loss_scale_factor: float = 1.0
    constrain_rotation: bool = False
    constrain_rotation_weight: float = 0.0
    refine_camera_params: bool = False
    print_progress: bool = False
    print_summary: bool = True

    @staticmethod
    def global_ba():
        """The values mapper.cc forces for global BA (reference src/mapper.cc:170-174, 878-881)."""
        return BundleAdjustmentOptions(max_num_iterations=200, function_tolerance=1e-6,
                                       gradient_tolerance=1e-10, update_point3D_errors=True,
                                       refine_camera_params=True, print_summary=False)


def make_options(opts=None, **kw):
    """COptions from a BundleAdjustmentOptions / dict plus overrides of the C-level fields."""
    o = A.COptions()
    load().mavba_options_init(C.byref(o))
    if isinstance(opts, BundleAdjustmentOptions):
        o.max_num_iterations = int(opts.max_num_iterations)
        o.function_tolerance = float(opts.function_tolerance)
        o.gradient_tolerance = float(opts.gradient_tolerance)
        o.loss_scale_factor = float(opts.loss_scale_factor)
        o.update_point_errors = int(bool(opts.update_point3D_errors))
        o.print_progress = int(bool(opts.print_progress))
    elif isinstance(opts, dict):
        kw = {**opts, **kw}
    for k, v in kw.items():
        if not hasattr(o, k):
            raise KeyError(k)
        setattr(o, k, v)
    return o


def print_report(res, title="Bundle Adjustment Report"):
    """stdout report in the reference's format (_print_report, bundle_adjustment.cc:114-136)."""
    n = max(res["num_residuals"], 1)
    print(title)
    print("-" * len(title))
    print(f"{'Residuals : ':>18}{res['num_residuals_reduced']}")
    print(f"{'Parameters : ':>18}{res['num_parameters_reduced']}")
    print(f"{'Iterations : ':>18}{res['num_successful_steps'] + res['num_unsuccessful_steps']}")
    print(f"{'Initial cost : ':>18}{np.sqrt(res['initial_cost'] / n):.6g} [px]")
    print(f"{'Final cost : ':>18}{np.sqrt(res['final_cost'] / n):.6g} [px]")
    print()


def bundle_adjustment(problem: BAProblem, options=None, point3D_errors=None, **kw):
    """Solve IN PLACE on `problem` (the reference mutates FeatureManager in place).

    Returns (final_cost_px, result) where final_cost_px = sqrt(final_cost / num_residuals)
    (the reference's return value, bundle_adjustment.cc:610) and `result` the summary dict.
    `point3D_errors`: optional float64 array [num_points], updated like the reference's map when
    options.update_point3D_errors is set.
    """
    if isinstance(options, BundleAdjustmentOptions):
        if options.min_track_len < 2:
            raise ValueError("Minimum track length must be >= 2 in order build valid bundle adjustment problem.")
    copt = make_options(options, **kw)
    res = A.CResult()
    cp = problem.c_struct()
    perr = None
    if point3D_errors is not None:
        copt.update_point_errors = 1
        perr = _d(point3D_errors)
    _check(load().mavba_solve(C.byref(cp), C.byref(copt), C.byref(res), perr))
    out = res.as_dict()
    if isinstance(options, BundleAdjustmentOptions) and options.print_summary:
        print_report(out)
    return float(np.sqrt(out["final_cost"] / out["num_residuals"])) if out["num_residuals"] else float("nan"), out


def bundle_adjustment_filter_rebundle(problem: BAProblem, filter_max_error, options=None, keep=None, point3D_errors=None, **kw):
    """Global BA, filter_point_cloud, global BA again on one resident session (reference src/mapper.cc:1206,
    1218-1224). Solves IN PLACE; returns (removed mask [num_points] uint8, first result, second result)."""
    copt = make_options(options, **kw)
    first, second = A.CResult(), A.CResult()
    cp = problem.c_struct()
    removed = np.zeros(problem.num_points, np.uint8)
    keep_arr = None if keep is None else np.ascontiguousarray(keep, dtype=np.uint8)
    n = C.c_int64()
    _check(load().mavba_solve_filter_solve(C.byref(cp), C.byref(copt), float(filter_max_error), A.ptr(keep_arr, C.c_uint8),
                                           C.byref(first), C.byref(second),
                                           _d(point3D_errors) if point3D_errors is not None else None,
                                           A.ptr(removed, C.c_uint8), C.byref(n)))
    assert int(removed.sum()) == n.value
    return removed, first.as_dict(), second.as_dict()


def pose_refinement(rvec, tvec, camera_params, points2D, points3D, inlier_mask=None, options=None, **kw):
    """Mirror of pose_refinement() (bundle_adjustment.h:212-218). `camera_params` carries the model
    code as its last element, exactly like FeatureManager.camera_params
    (reference src/sfm/sequential_mapper.cc:958-967). rvec/tvec are updated in place."""
    camera_params = np.asarray(camera_params, float)
    model = int(camera_params[-1])
    intr = A.as_f64(np.pad(camera_params[:-1], (0, 9 - (len(camera_params) - 1))))
    uv, xyz = A.as_f64(points2D, (-1, 2)), A.as_f64(points3D, (-1, 3))
    mask = None if inlier_mask is None else np.ascontiguousarray(inlier_mask, dtype=np.uint8)
    rv, tv = A.as_f64(rvec), A.as_f64(tvec)
    copt = make_options(options, **kw)
    res = A.CResult()
    _check(load().mavba_pose_refine(_d(rv), _d(tv), _d(intr), model, _d(uv), _d(xyz),
                                    A.ptr(mask, C.c_uint8), len(uv), C.byref(copt), C.byref(res)))
    np.asarray(rvec)[...] = rv
    np.asarray(tvec)[...] = tv
    out = res.as_dict()
    return float(np.sqrt(out["final_cost"] / out["num_residuals"])) if out["num_residuals"] else float("nan"), out


def pose_refinement_batch(items, options=None, **kw):
    """Many pose_refinement() problems in one launch (mavba_pose_refine_batch). `items`: list of dicts with rvec, tvec
    (float64 arrays of 3, updated in place), camera_params (model code last), points2D [n,2], points3D [n,3] and an
    optional inlier_mask [n]. Returns the list of (final_cost_px, result) pairs."""
    n = len(items)
    arr = (A.CPoseRefineItem * max(n, 1))()
    hold = []
    for q, it in enumerate(items):
        cp = np.asarray(it["camera_params"], float)
        intr = A.as_f64(np.pad(cp[:-1], (0, 9 - (len(cp) - 1))))
        uv, xyz = A.as_f64(it["points2D"], (-1, 2)), A.as_f64(it["points3D"], (-1, 3))
        mask = None if it.get("inlier_mask") is None else np.ascontiguousarray(it["inlier_mask"], dtype=np.uint8)
        hold.append((intr, uv, xyz, mask))
        for k in range(3):
            arr[q].rvec[k] = float(it["rvec"][k]); arr[q].tvec[k] = float(it["tvec"][k])
        arr[q].intrinsics = _d(intr); arr[q].camera_model = int(cp[-1])
        arr[q].uv = _d(uv); arr[q].xyz = _d(xyz); arr[q].inlier_mask = A.ptr(mask, C.c_uint8); arr[q].n = len(uv)
    res = (A.CResult * max(n, 1))()
    copt = make_options(options, **kw)
    _check(load().mavba_pose_refine_batch(n, arr, C.byref(copt), res))
    out = []
    for q, it in enumerate(items):
        np.asarray(it["rvec"])[...] = list(arr[q].rvec)
        np.asarray(it["tvec"])[...] = list(arr[q].tvec)
        d = res[q].as_dict()
        out.append((float(np.sqrt(d["final_cost"] / d["num_residuals"])) if d["num_residuals"] else float("nan"), d))
    return out


class Scene:
    """Incremental flat mirror of a FeatureManager (mavba_scene_*): deltas in, bundle_adjustment() calls out.
    Ids are the caller's (the FeatureManager's 1-based ids are fine)."""

    def __init__(self):
        self._h = C.c_void_p()
        _check(load().mavba_scene_create(C.byref(self._h)))

    def close(self):
        if self._h:
            load().mavba_scene_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_camera(self, camera_id, model, params):
        _check(load().mavba_scene_set_camera(self._h, int(camera_id), int(model), _d(A.as_f64(np.pad(np.asarray(params, float), (0, 9))[:9]))))

    def set_image(self, image_id, camera_id=-1, rvec=None, tvec=None):
        r = None if rvec is None else A.as_f64(rvec)
        t = None if tvec is None else A.as_f64(tvec)
        _check(load().mavba_scene_set_image(self._h, int(image_id), int(camera_id), None if r is None else _d(r), None if t is None else _d(t)))

    def add_point2D(self, image_id, point2D_id, xy):
        _check(load().mavba_scene_add_point2d(self._h, int(image_id), int(point2D_id), _d(A.as_f64(xy))))

    def add_points2D(self, image_id, point2D_ids, xy, point3D_ids=None):
        """Many 2-D points of one image in one call (`point3D_ids`: their links, < 0 = none)."""
        ids = np.ascontiguousarray(point2D_ids, dtype=np.int64)
        uv = A.as_f64(xy, (-1, 2))
        assert len(uv) == len(ids)
        l3 = None if point3D_ids is None else np.ascontiguousarray(point3D_ids, dtype=np.int64)
        _check(load().mavba_scene_add_points2d(self._h, int(image_id), len(ids), A.ptr(ids, C.c_int64), _d(uv),
                                               None if l3 is None else A.ptr(l3, C.c_int64)))

    def set_point3D(self, point3D_id, xyz):
        _check(load().mavba_scene_set_point3d(self._h, int(point3D_id), _d(A.as_f64(xyz))))

    def link(self, point2D_id, point3D_id):
        _check(load().mavba_scene_link(self._h, int(point2D_id), int(point3D_id)))

    def delete_point3D(self, point3D_id):
        _check(load().mavba_scene_delete_point3d(self._h, int(point3D_id)))

    def get_image(self, image_id):
        r, t = np.zeros(3), np.zeros(3)
        _check(load().mavba_scene_get_image(self._h, int(image_id), _d(r), _d(t)))
        return r, t

    def get_point3D(self, point3D_id):
        x = np.zeros(3)
        _check(load().mavba_scene_get_point3d(self._h, int(point3D_id), _d(x)))
        return x

    def get_camera(self, camera_id):
        m, p = C.c_int32(), np.zeros(9)
        _check(load().mavba_scene_get_camera(self._h, int(camera_id), C.byref(m), _d(p)))
        return m.value, p[:A.MODEL_NUM_PARAMS[m.value]]

    @staticmethod
    def _lists(free, fixed, fixed_x, gcp, rot):
        arrs = [np.ascontiguousarray(x, dtype=np.int64) for x in (free, fixed, fixed_x, gcp)]
        rot = rot or {}
        ri = np.ascontiguousarray(list(rot.keys()), dtype=np.int64)
        rv = A.as_f64(np.array([rot[k] for k in rot], float).reshape(-1, 3))
        args = []
        for a in arrs:
            args += [A.ptr(a, C.c_int64), len(a)]
        args += [A.ptr(ri, C.c_int64), _d(rv), len(ri)]
        return args, (arrs, ri, rv)

    @staticmethod
    def _scene_options(min_track_len=2, refine_camera_params=False, constrain_rotation=False, constrain_rotation_weight=0.0):
        return A.CSceneOptions(int(min_track_len), int(bool(refine_camera_params)), int(bool(constrain_rotation)), float(constrain_rotation_weight))

    def flatten(self, free, fixed, fixed_x, gcp=(), rot=None, **scene_opts):
        """The flat problem of a call as a dict of numpy COPIES (+ the flat-index -> id tables)."""
        args, hold = self._lists(free, fixed, fixed_x, gcp, rot)
        so = self._scene_options(**scene_opts)
        P = A.CProblem()
        i64p = C.POINTER(C.c_int64)
        ii, ci, pi = i64p(), i64p(), i64p()
        _check(load().mavba_scene_flatten(self._h, *args, C.byref(so), C.byref(P), C.byref(ii), C.byref(ci), C.byref(pi)))
        g = lambda ptr, n, dt: np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt).copy() if n else np.zeros(0, dt)  # noqa: E731
        ni, nc, npt, no, nr = P.num_images, P.num_cameras, P.num_points, P.num_obs, P.num_rot_priors
        return dict(image_ids=g(ii, ni, np.int64), camera_ids=g(ci, nc, np.int64), point_ids=g(pi, npt, np.int64),
                    poses=g(P.poses, ni * 6, float).reshape(-1, 6), pose_const=g(P.pose_const, ni, np.uint8),
                    image_camera=g(P.image_camera, ni, np.int32), intrinsics=g(P.intrinsics, nc * 9, float).reshape(-1, 9),
                    camera_model=g(P.camera_model, nc, np.int32), intr_const=g(P.intr_const, nc, np.uint8),
                    points=g(P.points, npt * 3, float).reshape(-1, 3), point_const=g(P.point_const, npt, np.uint8),
                    obs_uv=g(P.obs_uv, no * 2, float).reshape(-1, 2), obs_image=g(P.obs_image, no, np.int32),
                    obs_point=g(P.obs_point, no, np.int32), rot_prior_image=g(P.rot_prior_image, nr, np.int32),
                    rot_prior_rvec=g(P.rot_prior_rvec, nr * 3, float).reshape(-1, 3), rot_prior_weight=P.rot_prior_weight)

    def bundle_adjustment(self, free, fixed, fixed_x, options=None, gcp=(), rot=None, point3D_errors=None, **scene_opts):
        """bundle_adjustment() on the scene (results are written back into it). Returns (final_cost_px, result);
        `point3D_errors`: optional dict updated like the reference's map when options.update_point_errors is set."""
        args, hold = self._lists(free, fixed, fixed_x, gcp, rot)
        so = self._scene_options(**scene_opts)
        copt = make_options(options)
        if point3D_errors is not None:
            copt.update_point_errors = 1
        res = A.CResult()
        cost = C.c_double()
        i64p = C.POINTER(C.c_int64)
        eids, evals, n = i64p(), C.POINTER(C.c_double)(), C.c_int64()
        _check(load().mavba_scene_bundle_adjust(self._h, *args, C.byref(so), C.byref(copt), C.byref(res), C.byref(cost),
                                                C.byref(eids), C.byref(evals), C.byref(n)))
        if point3D_errors is not None and n.value:
            ids = np.ctypeslib.as_array(eids, shape=(n.value,))
            vals = np.ctypeslib.as_array(evals, shape=(n.value,))
            point3D_errors.update(zip(ids.tolist(), vals.tolist()))
        return cost.value, res.as_dict()


class Session:
    """Device-resident solver state (mavba_session_*)."""

    def __init__(self, problem: BAProblem, options=None, **kw):
        self.problem = problem
        self.copt = make_options(options, **kw)
        self._cp = problem.c_struct()
        self._h = C.c_void_p()
        self._cb = None
        _check(load().mavba_session_create(C.byref(self._cp), C.byref(self.copt), C.byref(self._h)))

    def close(self):
        if self._h:
            load().mavba_session_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def reset(self):
        _check(load().mavba_session_reset(self._h))

    def iterate(self, max_iters=1):
        done, term = C.c_int32(), C.c_int32()
        _check(load().mavba_session_iterate(self._h, int(max_iters), C.byref(done), C.byref(term)))
        return done.value, term.value

    def solve(self):
        self.iterate(self.copt.max_num_iterations + 1)
        return self.result()

    def result(self):
        r = A.CResult()
        _check(load().mavba_session_result(self._h, C.byref(r)))
        return r.as_dict()

    def get_params(self):
        p = self.problem
        poses, intr, pts = np.zeros((p.num_images, 6)), np.zeros((p.num_cameras, 9)), np.zeros((p.num_points, 3))
        _check(load().mavba_session_get_params(self._h, _d(poses), _d(intr), _d(pts)))
        return poses, intr, pts

    def point_errors(self):
        e = np.full(self.problem.num_points, np.nan)
        _check(load().mavba_session_point_errors(self._h, _d(e)))
        return e

    def set_params(self, poses=None, intrinsics=None, points=None):
        p = self.problem
        arrs = [None if a is None else A.as_f64(a, shape) for a, shape in
                ((poses, (p.num_images, 6)), (intrinsics, (p.num_cameras, 9)), (points, (p.num_points, 3)))]
        _check(load().mavba_session_set_params(self._h, *[None if a is None else _d(a) for a in arrs]))

    def restart(self):
        _check(load().mavba_session_restart(self._h))

    def filter_points(self, max_error, keep=None):
        """filter_point_cloud on the resident session: returns (removed mask, errors the decision used)."""
        n = self.problem.num_points
        removed, errors = np.zeros(n, np.uint8), np.full(n, np.nan)
        keep_arr = None if keep is None else np.ascontiguousarray(keep, dtype=np.uint8)
        cnt = C.c_int64()
        _check(load().mavba_session_filter_points(self._h, float(max_error), A.ptr(keep_arr, C.c_uint8),
                                                  A.ptr(removed, C.c_uint8), _d(errors), C.byref(cnt)))
        return removed, errors

    def set_allreduce(self, fn, rank, world_size):
        """fn(device_ptr:int, count:int, op:int) -> None; must all-reduce in place and return when done."""
        def trampoline(_ctx, ptr, count, op):
            try:
                fn(ptr, count, op)
                return 0
            except Exception as e:  # noqa: BLE001 - must not unwind through C
                print("all-reduce hook raised:", repr(e), flush=True)
                return 1
        self._cb = ALLREDUCE_FN(trampoline)
        _check(load().mavba_session_set_allreduce(self._h, self._cb, None, rank, world_size))

    def set_rccl(self, unique_id, rank, world_size):
        """Native RCCL exchange (mavba_session_set_rccl); `unique_id` = the 128 bytes of rccl_unique_id() from rank 0."""
        buf = C.create_string_buffer(bytes(unique_id), 128)
        _check(load().mavba_session_set_rccl(self._h, buf, rank, world_size))

    def eval_jacobian(self):
        n = self.problem.num_obs
        r, Jc, Jp, Jk = np.zeros((n, 2)), np.zeros((n, 2, 6)), np.zeros((n, 2, 3)), np.zeros((n, 2, 9))
        cost = C.c_double()
        _check(load().mavba_session_eval_jacobian(self._h, C.byref(cost), _d(r), _d(Jc), _d(Jp), _d(Jk)))
        return cost.value, r, Jc, Jp, Jk

    def reduced_system(self, radius):
        n = load().mavba_session_reduced_dim(self._h)
        S, v = np.zeros((n, n)), np.zeros(n)
        _check(load().mavba_session_reduced_system(self._h, float(radius), _d(S), _d(v)))
        return S, v

    def linear_step(self, radius):
        p = self.problem
        dp, di, dx = np.zeros((p.num_images, 6)), np.zeros((p.num_cameras, 9)), np.zeros((p.num_points, 3))
        mcc = C.c_double()
        _check(load().mavba_session_linear_step(self._h, float(radius), _d(dp), _d(di), _d(dx), C.byref(mcc)))
        return dict(d_poses=dp, d_intr=di, d_points=dx, model_cost_change=mcc.value)

    def set_profiling(self, on):
        """Event brackets around the kernels on / off for the iterations that follow."""
        _check(load().mavba_session_set_profiling(self._h, 1 if on else 0))

    def time_front(self, radius=1e4, reps=20):
        """Average milliseconds of one pass of the linear solve's front end (probe)."""
        ms = C.c_float(0)
        _check(load().mavba_session_time_front(self._h, float(radius), int(reps), C.byref(ms)))
        return ms.value

    def time_jacobian(self, reps=20):
        ms = C.c_float()
        _check(load().mavba_session_time_jacobian(self._h, int(reps), C.byref(ms)))
        return float(ms.value)

    def info(self):
        i = A.CSessionInfo()
        _check(load().mavba_session_get_info(self._h, C.byref(i)))
        d = {k: getattr(i, k) for k, _ in i._fields_ if k != "schur_terms"}
        d["schur_terms"] = list(i.schur_terms)
        return d

    def kernel_stats(self):
        buf = (A.CKernelStat * 64)()
        n = load().mavba_session_kernel_stats(self._h, buf, 64)
        return {buf[i].name.decode(): dict(launches=int(buf[i].launches), total_ms=float(buf[i].total_ms))
                for i in range(min(n, 64))}


def elimination_tree(num_images, num_cameras, pairs, max_depth=3):
    """The elimination tree of the reduced camera system for an image graph (`pairs`: [n, 2] coupled images); no GPU
    needed. Returns (node_of_image [num_images], node_parent [num_nodes]); no nodes = no dissection."""
    pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
    a, b = np.ascontiguousarray(pairs[:, 0]), np.ascontiguousarray(pairs[:, 1])
    node = np.zeros(max(num_images, 1), np.int32)
    parent = np.zeros(4096, np.int32)
    n = load().mavba_debug_elimination_tree(int(num_images), int(num_cameras), len(a), A.ptr(a, C.c_int32), A.ptr(b, C.c_int32),
                                           int(max_depth), A.ptr(node, C.c_int32), A.ptr(parent, C.c_int32), len(parent))
    if n < 0:
        _check(n)
    return node[:num_images], parent[:n]


def radix_sort_order(keys, key_bytes=4, device=-1):
    """Stable ascending order of range(len(keys)) by uint32 keys, computed by the set-up's device radix sort."""
    keys = np.ascontiguousarray(keys, dtype=np.uint32)
    out = np.zeros(len(keys), np.int32)
    _check(load().mavba_debug_radix_sort(len(keys), A.ptr(keys, C.c_uint32), int(key_bytes), A.ptr(out, C.c_int32), device))
    return out


def debug_upload_batch(sizes, arena_bytes=0, device=-1):
    """Wrong bytes after a batch of small uploads / clears of the given sizes (negative: a clear) - 0 = pass."""
    sizes = np.ascontiguousarray(sizes, dtype=np.int64)
    f = load().mavba_debug_upload_batch
    f.restype = C.c_int64   # (wrong bytes, not a status)
    r = f(C.c_int32(len(sizes)), A.ptr(sizes, C.c_int64), C.c_int64(int(arena_bytes)), C.c_int32(device))
    if r < 0:
        raise MavbaError(A.ERR_HIP, load().mavba_last_error().decode(errors="replace"))
    return int(r)


def debug_lm_decide(cases, device=-1):
    """The LM decision function (csrc/lm_decide.h) on rows of 16 scalars + 8 parameters: (host results, device results), 6 columns each."""
    cases = np.ascontiguousarray(cases, dtype=np.float64)
    n = cases.shape[0]
    assert cases.shape[1] == 24
    oh, od = np.zeros((n, 6)), np.zeros((n, 6))
    _check(load().mavba_debug_lm_decide(n, _d(cases), _d(oh), _d(od), device))
    return oh, od


def debug_chol_schedule(nb, nodes, pairs, cus=256):
    """Tile structure + persistent schedule of a factorisation on the HOST (no device): `nodes` = [(begin, end, parent)] of the
    elimination tree in tile columns, `pairs` = non-zero lower tiles (row, col). Returns a dict with the modelled times and
    the helpers' / chains' queues: tasks = [(work-group, kind, i, j, [update columns])], kinds 0 TILE 1 PRE_DIAG 2 PRE_SUB 3 CHAIN."""
    nodes = np.ascontiguousarray(nodes, dtype=np.int32).reshape(-1, 3)
    pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
    nbg, ne, npar = (np.ascontiguousarray(nodes[:, k]) for k in range(3))
    pr, pc = np.ascontiguousarray(pairs[:, 0]), np.ascontiguousarray(pairs[:, 1])
    i32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    out = np.zeros(8)
    nt, nu = C.c_int64(), C.c_int64()
    ci = np.zeros(nb, np.int32)
    L = load()
    _check(L.mavba_debug_chol_schedule(nb, len(nodes), i32(nbg), i32(ne), i32(npar), len(pairs), i32(pr), i32(pc), cus, _d(out),
                                       None, 0, C.byref(nt), None, 0, C.byref(nu), i32(ci)))
    tasks = np.zeros((max(nt.value, 1), 6), np.int32)
    upd = np.zeros(max(nu.value, 1), np.int32)
    _check(L.mavba_debug_chol_schedule(nb, len(nodes), i32(nbg), i32(ne), i32(npar), len(pairs), i32(pr), i32(pc), cus, _d(out),
                                       i32(tasks), nt.value, C.byref(nt), i32(upd), nu.value, C.byref(nu), i32(ci)))
    tl = [(int(t[0]), int(t[1]), int(t[2]), int(t[3]), [int(x) for x in upd[t[4]:t[5]]]) for t in tasks[:nt.value]]
    return dict(ok=bool(out[0]), model_forward_us=float(out[1]), launch_per_panel_us=float(out[2]), grid=int(out[3]),
                chain_wgs=int(out[4]), tiles=int(out[5]), updates=int(out[6]), nodes=int(out[7]), tasks=tl, chain_info=ci.tolist())


def rccl_unique_id():
    """128-byte ncclUniqueId for Session.set_rccl (call on rank 0, distribute to every rank)."""
    buf = C.create_string_buffer(128)
    _check(load().mavba_rccl_unique_id(buf))
    return buf.raw


def dense_spd_solve(Amat, b, device=-1):
    Amat, b = A.as_f64(Amat), A.as_f64(b)
    x = np.zeros_like(b)
    _check(load().mavba_dense_spd_solve(len(b), _d(Amat), _d(b), _d(x), device))
    return x
