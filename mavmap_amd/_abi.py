"""ctypes mirror of include/mavba.h (structs and constants only — no library here).

Shared by the product wrapper (mavmap_amd.api) and, in tests/, by the oracle
loader, so both sides of a parity test see byte-identical problem memory.
"""
import ctypes as C

import numpy as np

MODEL_PINHOLE, MODEL_OPENCV, MODEL_CATA = 1, 2, 3
MAX_INTR = 9
MODEL_NUM_PARAMS = {MODEL_PINHOLE: 4, MODEL_OPENCV: 8, MODEL_CATA: 9}

CONST_RVEC, CONST_TX, CONST_TY, CONST_TZ = 1, 2, 4, 8
CONST_POSE = 15
# reference src/base3d/bundle_adjustment.h:33-35
BA_POSE_FREE, BA_POSE_FIXED, BA_POSE_FIXED_X = 0, 1, 2
POSE_STATE_TO_MASK = {BA_POSE_FREE: 0, BA_POSE_FIXED: CONST_POSE, BA_POSE_FIXED_X: CONST_TX}

OK = 0
ERR_INVALID_ARGUMENT, ERR_NO_DEVICE, ERR_HIP, ERR_OUT_OF_MEMORY, ERR_BAD_INDEX, ERR_BAD_MODEL = (
    -1, -2, -3, -4, -5, -6)

TERM_RUNNING = -1
TERM_NO_CONVERGENCE, TERM_FUNCTION_TOLERANCE, TERM_GRADIENT_TOLERANCE = 0, 1, 2
TERM_PARAMETER_TOLERANCE, TERM_NUMERICAL_FAILURE = 3, 4
TERM_NAMES = {-1: "RUNNING", 0: "NO_CONVERGENCE", 1: "FUNCTION_TOLERANCE",
              2: "GRADIENT_TOLERANCE", 3: "PARAMETER_TOLERANCE", 4: "NUMERICAL_FAILURE"}

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_bp = C.POINTER(C.c_uint8)


class CProblem(C.Structure):
    _fields_ = [
        ("num_images", C.c_int32), ("num_cameras", C.c_int32), ("num_points", C.c_int32),
        ("num_obs", C.c_int64),
        ("poses", _dp), ("pose_const", _bp), ("image_camera", _ip),
        ("intrinsics", _dp), ("camera_model", _ip), ("intr_const", _bp),
        ("points", _dp), ("point_const", _bp),
        ("obs_uv", _dp), ("obs_image", _ip), ("obs_point", _ip),
        ("num_rot_priors", C.c_int32), ("rot_prior_image", _ip), ("rot_prior_rvec", _dp),
        ("rot_prior_weight", C.c_double),
    ]


class COptions(C.Structure):
    _fields_ = [
        ("max_num_iterations", C.c_int32), ("function_tolerance", C.c_double),
        ("gradient_tolerance", C.c_double), ("loss_scale_factor", C.c_double),
        ("update_point_errors", C.c_int32), ("print_progress", C.c_int32),
        ("parameter_tolerance", C.c_double), ("initial_trust_region_radius", C.c_double),
        ("max_trust_region_radius", C.c_double), ("min_trust_region_radius", C.c_double),
        ("min_relative_decrease", C.c_double), ("min_lm_diagonal", C.c_double),
        ("max_lm_diagonal", C.c_double), ("max_num_consecutive_invalid_steps", C.c_int32),
        ("jacobi_scaling", C.c_int32), ("device", C.c_int32), ("profile_kernels", C.c_int32),
    ]


class CResult(C.Structure):
    _fields_ = [
        ("initial_cost", C.c_double), ("final_cost", C.c_double), ("fixed_cost", C.c_double),
        ("num_residuals", C.c_int64), ("num_residuals_reduced", C.c_int64),
        ("num_parameters_reduced", C.c_int64),
        ("num_successful_steps", C.c_int32), ("num_unsuccessful_steps", C.c_int32),
        ("termination", C.c_int32),
        ("final_gradient_max_norm", C.c_double), ("final_trust_region_radius", C.c_double),
        ("setup_seconds", C.c_double), ("solve_seconds", C.c_double),
    ]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["termination_name"] = TERM_NAMES.get(self.termination, "?")
        return d


class CPoseRefineItem(C.Structure):
    _fields_ = [("rvec", C.c_double * 3), ("tvec", C.c_double * 3), ("intrinsics", _dp), ("camera_model", C.c_int32),
                ("uv", _dp), ("xyz", _dp), ("inlier_mask", _bp), ("n", C.c_int64)]


class CSceneOptions(C.Structure):
    _fields_ = [("min_track_len", C.c_int32), ("refine_camera_params", C.c_int32), ("constrain_rotation", C.c_int32),
                ("constrain_rotation_weight", C.c_double)]


class CSessionInfo(C.Structure):
    _fields_ = [("num_obs_kept", C.c_int64), ("reduced_dim", C.c_int32), ("padded_dim", C.c_int32),
                ("schur_terms", C.c_int64 * 3), ("schur_blocks", C.c_int64), ("intr_entries", C.c_int64),
                ("envelope_tiles", C.c_int64), ("dense_tiles", C.c_int64), ("factor_flops", C.c_double),
                ("dense_factor_flops", C.c_double), ("matrix_dim", C.c_int32), ("nd_parts", C.c_int32),
                ("chain_steps", C.c_int32), ("num_clusters", C.c_int32),
                ("clustered_points", C.c_int64), ("cluster_partials", C.c_int64),
                ("cluster_flops", C.c_double), ("chol_model_forward_us", C.c_double), ("reduced_store_bytes", C.c_int64)]


class CKernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int64), ("total_ms", C.c_double)]


def ptr(a, ctype):
    """Pointer to a C-contiguous numpy array (or NULL for None)."""
    if a is None:
        return C.cast(None, C.POINTER(ctype))
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(ctype))


def as_f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


# scalars of an LM iteration on the device (csrc/internal.h, SC_*): the layout mavba_debug_lm_decide takes
SC_COST, SC_XNORM2, SC_GRAD_MAX, SC_NEW_COST, SC_STEP_NORM2, SC_MODEL_CHANGE, SC_CAND_XNORM2, SC_FAIL, SC_FAIL_FRONT = range(9)
