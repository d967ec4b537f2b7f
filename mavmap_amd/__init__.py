"""mavmap_amd — MI355X-native bundle-adjustment backend for MAVMAP (hot path only).

Only what the path needs lives here: csrc/ (HIP kernels + the C ABI of include/mavba.h),
the ctypes binding (api), the flat problem container and the synthetic scene generator.
"""
from . import _abi  # noqa: F401
from .problem import BAProblem  # noqa: F401
from .api import (BundleAdjustmentOptions, MavbaError, Scene, Session, bundle_adjustment,  # noqa: F401
                  bundle_adjustment_filter_rebundle, pose_refinement, pose_refinement_batch, device_count, dense_spd_solve, rccl_unique_id, load, lib_path, radix_sort_order, elimination_tree, debug_upload_batch)
