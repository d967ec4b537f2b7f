"""Flat bundle-adjustment problem: the numpy image of `mavba_problem` (include/mavba.h).

This is the array form of what the reference keeps in FeatureManager hash maps
(reference src/fm/feature_management.h:189-230) after
_bundle_adjustment_extract_data / _fill_problem
(reference src/base3d/bundle_adjustment.cc:228-387) have selected the images
and observations of one bundle_adjustment() call.
"""
import copy
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _abi as A


@dataclass
class BAProblem:
    poses: np.ndarray            # (NI, 6) rvec | tvec, world -> camera
    pose_const: np.ndarray       # (NI,) uint8 MAVBA_CONST_* mask
    image_camera: np.ndarray     # (NI,) int32
    intrinsics: np.ndarray       # (NC, 9)
    camera_model: np.ndarray     # (NC,) int32
    intr_const: np.ndarray       # (NC,) uint8
    points: np.ndarray           # (NP, 3)
    point_const: np.ndarray      # (NP,) uint8
    obs_uv: np.ndarray           # (NO, 2)
    obs_image: np.ndarray        # (NO,) int32
    obs_point: np.ndarray        # (NO,) int32
    rot_prior_image: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    rot_prior_rvec: np.ndarray = field(default_factory=lambda: np.zeros((0, 3)))
    rot_prior_weight: float = 0.0
    truth: dict = field(default_factory=dict)   # generator ground truth (not part of the ABI)

    def __post_init__(self):
        self.poses = A.as_f64(self.poses, (-1, 6))
        self.intrinsics = A.as_f64(self.intrinsics, (-1, A.MAX_INTR))
        self.points = A.as_f64(self.points, (-1, 3))
        self.obs_uv = A.as_f64(self.obs_uv, (-1, 2))
        self.rot_prior_rvec = A.as_f64(self.rot_prior_rvec, (-1, 3))
        for name in ("image_camera", "camera_model", "obs_image", "obs_point", "rot_prior_image"):
            setattr(self, name, np.ascontiguousarray(getattr(self, name), dtype=np.int32))
        for name in ("pose_const", "intr_const", "point_const"):
            setattr(self, name, np.ascontiguousarray(getattr(self, name), dtype=np.uint8))
        assert len(self.pose_const) == len(self.image_camera) == self.num_images
        assert len(self.camera_model) == len(self.intr_const) == self.num_cameras
        assert len(self.point_const) == self.num_points
        assert len(self.obs_image) == len(self.obs_point) == self.num_obs

    num_images = property(lambda s: s.poses.shape[0])
    num_cameras = property(lambda s: s.intrinsics.shape[0])
    num_points = property(lambda s: s.points.shape[0])
    num_obs = property(lambda s: s.obs_uv.shape[0])

    def copy(self):
        return copy.deepcopy(self)

    def c_struct(self):
        """A `mavba_problem` pointing INTO this object's arrays (results land in place)."""
        p = A.CProblem()
        p.num_images, p.num_cameras, p.num_points = self.num_images, self.num_cameras, self.num_points
        p.num_obs = self.num_obs
        p.poses = A.ptr(self.poses, C.c_double)
        p.pose_const = A.ptr(self.pose_const, C.c_uint8)
        p.image_camera = A.ptr(self.image_camera, C.c_int32)
        p.intrinsics = A.ptr(self.intrinsics, C.c_double)
        p.camera_model = A.ptr(self.camera_model, C.c_int32)
        p.intr_const = A.ptr(self.intr_const, C.c_uint8)
        p.points = A.ptr(self.points, C.c_double)
        p.point_const = A.ptr(self.point_const, C.c_uint8)
        p.obs_uv = A.ptr(self.obs_uv, C.c_double)
        p.obs_image = A.ptr(self.obs_image, C.c_int32)
        p.obs_point = A.ptr(self.obs_point, C.c_int32)
        p.num_rot_priors = len(self.rot_prior_image)
        p.rot_prior_image = A.ptr(self.rot_prior_image, C.c_int32)
        p.rot_prior_rvec = A.ptr(self.rot_prior_rvec, C.c_double)
        p.rot_prior_weight = float(self.rot_prior_weight)
        return p

    # -- sharding by 3-D point ------------------------------
    # ---- replay files (same layout as the shim's MAVBA_DUMP_DIR dumps, shim/base3d/bundle_adjustment.cc) ----
    MAGIC = b"MAVBA1\0\0"

    def save(self, path, options=None):
        """Write the problem as a replay file; `options` = dict with max_num_iterations, function_tolerance,
        gradient_tolerance, loss_scale_factor (stored for information, defaults = reference global BA)."""
        o = dict(max_num_iterations=200, function_tolerance=1e-6, gradient_tolerance=1e-10, loss_scale_factor=1.0)
        o.update(options or {})
        with open(path, "wb") as f:
            f.write(self.MAGIC)
            np.array([self.num_images, self.num_cameras, self.num_points, len(self.rot_prior_image)], "<i4").tofile(f)
            np.array([self.num_obs], "<i8").tofile(f)
            np.array([self.rot_prior_weight, o["max_num_iterations"], o["function_tolerance"], o["gradient_tolerance"],
                      o["loss_scale_factor"]], "<f8").tofile(f)
            for a, t in ((self.poses, "<f8"), (self.pose_const, "u1"), (self.image_camera, "<i4"), (self.intrinsics, "<f8"),
                         (self.camera_model, "<i4"), (self.intr_const, "u1"), (self.points, "<f8"), (self.point_const, "u1"),
                         (self.obs_uv, "<f8"), (self.obs_image, "<i4"), (self.obs_point, "<i4"), (self.rot_prior_image, "<i4"),
                         (self.rot_prior_rvec, "<f8")):
                np.ascontiguousarray(a, dtype=t).tofile(f)

    @classmethod
    def load(cls, path):
        """(problem, options dict) from a replay file."""
        with open(path, "rb") as f:
            if f.read(8) != cls.MAGIC:
                raise ValueError(f"{path}: not a mavba replay file")
            ni, nc, npt, npri = (int(x) for x in np.fromfile(f, "<i4", 4))
            no = int(np.fromfile(f, "<i8", 1)[0])
            w, it, ftol, gtol, loss = (float(x) for x in np.fromfile(f, "<f8", 5))

            def rd(t, n, shape=None):
                a = np.fromfile(f, t, n)
                if len(a) != n:
                    raise ValueError(f"{path}: truncated")
                return a.reshape(shape) if shape else a
            p = cls(poses=rd("<f8", ni * 6, (ni, 6)), pose_const=rd("u1", ni), image_camera=rd("<i4", ni),
                    intrinsics=rd("<f8", nc * 9, (nc, 9)), camera_model=rd("<i4", nc), intr_const=rd("u1", nc),
                    points=rd("<f8", npt * 3, (npt, 3)), point_const=rd("u1", npt), obs_uv=rd("<f8", no * 2, (no, 2)),
                    obs_image=rd("<i4", no), obs_point=rd("<i4", no), rot_prior_image=rd("<i4", npri),
                    rot_prior_rvec=rd("<f8", npri * 3, (npri, 3)), rot_prior_weight=w)
        return p, dict(max_num_iterations=int(it), function_tolerance=ftol, gradient_tolerance=gtol, loss_scale_factor=loss)

    def shard_by_point(self, rank, world_size):
        """Points (and all their observations) owned by `rank`; cameras replicated.

        Points are dealt in contiguous blocks balanced by observation count, so
        every rank streams a similar number of observations.
        """
        if world_size == 1:
            return self.copy(), np.arange(self.num_points)
        cnt = np.bincount(self.obs_point, minlength=self.num_points).astype(np.int64)
        csum = np.cumsum(cnt)
        total = int(csum[-1]) if len(csum) else 0
        bounds = [int(np.searchsorted(csum, total * r / world_size, side="left")) for r in range(world_size)]
        bounds.append(self.num_points)
        bounds[0] = 0
        lo, hi = bounds[rank], bounds[rank + 1]
        sel = (self.obs_point >= lo) & (self.obs_point < hi)
        q = self.copy()
        q.points = np.ascontiguousarray(self.points[lo:hi])
        q.point_const = np.ascontiguousarray(self.point_const[lo:hi])
        q.obs_uv = np.ascontiguousarray(self.obs_uv[sel])
        q.obs_image = np.ascontiguousarray(self.obs_image[sel])
        q.obs_point = np.ascontiguousarray(self.obs_point[sel] - lo)
        if rank != 0:
            # rotation-prior residuals are camera-only: counted once, on rank 0
            q.rot_prior_image = np.zeros(0, np.int32)
            q.rot_prior_rvec = np.zeros((0, 3))
        return q, np.arange(lo, hi)
