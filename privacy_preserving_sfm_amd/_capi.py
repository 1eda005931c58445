"""ctypes binding of include/ppsfm_hip.h (libppsfm_hip.so, gfx950).

There is deliberately NO fallback: if the HIP library cannot be built/loaded the import of any
compute entry point raises, and every compute call needs a GPU.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int32)
c_u8p = C.POINTER(C.c_uint8)
c_u16p = C.POINTER(C.c_uint16)
c_u32p = C.POINTER(C.c_uint32)

PP_OK, PP_ERR_INVALID, PP_ERR_HIP, PP_ERR_NUMERIC, PP_ERR_NOMEM, PP_ERR_INTERNAL = 0, -1, -2, -3, -4, -5
CAM_STRIDE = 12
BA_T_NAMES = ("eval", "reduce", "schur", "cholesky", "backsub", "update_cost")


class PPError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("ppsfm_hip error %d: %s" % (code, msg))
        self.code = code


LINEAR_SOLVER_AUTO, LINEAR_SOLVER_DIRECT, LINEAR_SOLVER_ITERATIVE_SCHUR = 0, 1, 2      # PP_LINEAR_SOLVER_*


class BAProblemDesc(C.Structure):
    _fields_ = [("num_poses", C.c_int32), ("num_points", C.c_int32), ("num_cameras", C.c_int32), ("loss_type", C.c_int32),
                ("num_obs", C.c_int64), ("loss_scale", C.c_double),
                ("lines", c_dp), ("obs_pose", c_ip), ("obs_point", c_ip), ("pose_camera", c_ip), ("camera_model", c_ip),
                ("pose_const", c_u8p), ("tvec_const_mask", c_u8p), ("point_const", c_u8p), ("camera_const_mask", c_u16p),
                ("linear_solver", C.c_int32), ("ordering", C.c_int32), ("covisibility", c_u8p)]


ORDERING_DEFAULT, ORDERING_NATURAL, ORDERING_AUTO = 0, 1, 2      # PP_ORDERING_* (0 and 1: the caller's order)


class BAOptions(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int32), ("max_num_consecutive_invalid_steps", C.c_int32),
                ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
                ("initial_trust_region_radius", C.c_double), ("max_trust_region_radius", C.c_double),
                ("min_trust_region_radius", C.c_double), ("min_relative_decrease", C.c_double),
                ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
                ("jacobi_scaling", C.c_int32), ("phase_timings", C.c_int32),
                ("max_linear_solver_iterations", C.c_int32), ("reserved_", C.c_int32), ("eta", C.c_double),
                ("iteration_callback", C.c_void_p), ("iteration_callback_ctx", C.c_void_p)]


class BAIterationSummary(C.Structure):
    _fields_ = [("iteration", C.c_int32), ("step_is_successful", C.c_int32), ("cost", C.c_double), ("cost_change", C.c_double),
                ("gradient_max_norm", C.c_double), ("step_norm", C.c_double), ("relative_decrease", C.c_double),
                ("trust_region_radius", C.c_double)]


ITERATION_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(BAIterationSummary))
SOLVER_CONTINUE, SOLVER_ABORT, SOLVER_TERMINATE_SUCCESSFULLY = 0, 1, 2
TERM_CONVERGENCE, TERM_NO_CONVERGENCE, TERM_FAILURE, TERM_USER_SUCCESS, TERM_USER_FAILURE = 0, 1, 2, 3, 4


class BASummary(C.Structure):
    _fields_ = [("initial_cost", C.c_double), ("final_cost", C.c_double), ("num_successful_steps", C.c_int32),
                ("num_unsuccessful_steps", C.c_int32), ("termination", C.c_int32), ("num_iterations", C.c_int32),
                ("num_residuals", C.c_int32), ("num_effective_parameters", C.c_int32),
                ("total_time_s", C.c_double), ("device_time_s", C.c_double),
                ("linear_solver", C.c_int32), ("cholesky_fallbacks", C.c_int32), ("linear_solver_iterations", C.c_int32),
                ("reserved_", C.c_int32)]


class RansacOptions(C.Structure):
    _fields_ = [("max_error", C.c_double), ("min_inlier_ratio", C.c_double), ("confidence", C.c_double),
                ("dyn_num_trials_multiplier", C.c_double), ("min_num_trials", C.c_uint64), ("max_num_trials", C.c_uint64),
                ("seed", C.c_uint32), ("chunk_trials", C.c_uint32)]


class RansacReport(C.Structure):
    _fields_ = [("success", C.c_int32), ("best_model_index", C.c_int32), ("num_trials", C.c_uint64), ("num_inliers", C.c_uint64),
                ("residual_sum", C.c_double), ("model", C.c_double * 12), ("best_trial", C.c_int64),
                ("hypotheses_evaluated", C.c_uint64), ("models_scored", C.c_uint64),
                ("device_time_s", C.c_double), ("total_time_s", C.c_double)]


class TriangulationOptions(C.Structure):
    _fields_ = [("min_tri_angle", C.c_double), ("residual_type", C.c_int32), ("reserved", C.c_int32), ("ransac", RansacOptions)]


class FilterOptions(C.Structure):
    _fields_ = [("max_reproj_error", C.c_double), ("min_tri_angle_deg", C.c_double)]


class FilterReport(C.Structure):
    _fields_ = [("num_filtered", C.c_int64), ("num_points_deleted", C.c_int64), ("num_observations_deleted", C.c_int64)]


class LoMsacOptions(C.Structure):
    _fields_ = [("min_num_iterations", C.c_uint32), ("max_num_iterations", C.c_uint32), ("success_probability", C.c_double),
                ("squared_inlier_threshold", C.c_double), ("random_seed", C.c_uint32), ("num_lo_steps", C.c_int32),
                ("threshold_multiplier", C.c_double), ("num_lsq_iterations", C.c_int32), ("min_sample_multiplicator", C.c_int32),
                ("non_min_sample_multiplier", C.c_int32), ("lo_starting_iterations", C.c_uint32), ("final_least_squares", C.c_int32),
                ("chunk_iterations", C.c_uint32)]


class LoMsacReport(C.Structure):
    _fields_ = [("num_iterations", C.c_uint32), ("best_num_inliers", C.c_int32), ("best_model_score", C.c_double),
                ("inlier_ratio", C.c_double), ("number_lo_iterations", C.c_int32), ("num_inlier_indices", C.c_int32),
                ("hypotheses_evaluated", C.c_uint64), ("device_time_s", C.c_double), ("total_time_s", C.c_double)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32)

_EXPORTS = [
    "pp_last_error", "pp_device_count", "pp_debug_raise", "pp_camera_num_params", "pp_camera_image_to_world_threshold",
    "pp_ba_options_default", "pp_ba_create", "pp_ba_destroy", "pp_ba_set_parameters", "pp_ba_get_parameters",
    "pp_ba_eval", "pp_ba_eval_host_view", "pp_ba_eval_device", "pp_ba_solve", "pp_ba_get_trace", "pp_ba_get_structure", "pp_ba_plan_ordering", "pp_ba_pair_lists_host", "pp_ba_covisibility", "pp_ba_get_create_profile", "pp_ba_reduced_system", "pp_ba_set_allreduce", "pp_ba_set_communicator", "pp_comm_unique_id", "pp_comm_create", "pp_comm_destroy",
    "pp_comm_allreduce",
    "pp_ba_get_timings", "pp_pool_trim", "pp_dense_cholesky_solve", "pp_cholesky_task_list", "pp_cholesky_task_list_sparse", "pp_cholesky_task_plan",
    "pp_pose_create", "pp_pose_destroy", "pp_pose_residuals", "pp_pose_score", "pp_pose_support_sequential",
    "pp_pose_p6l_batch", "pp_re3q3_batch", "pp_ransac_options_default", "pp_pose_ransac", "pp_pose_hypotheses", "pp_pose_last_scores",
    "pp_sampler_draw", "pp_ransac_compute_num_trials",
    "pp_lomsac_options_default", "pp_planar_create", "pp_planar_destroy", "pp_planar_solve_batch", "pp_planar_score",
    "pp_planar_evaluate", "pp_planar_lomsac", "pp_fourview2d_create", "pp_fourview2d_destroy", "pp_fourview2d_score",
    "pp_triangulate_tracks", "pp_ba_filter_points", "pp_ba_filter_negative_depth", "pp_pose2d_create", "pp_pose2d_destroy", "pp_pose2d_solve_batch", "pp_pose2d_score", "pp_pose2d_evaluate", "pp_pose2d_lomsac",
    "pp_fourview2d_evaluate", "pp_fourview2d_evaluate_points", "pp_fourview2d_default_frames", "pp_fourview2d_minimal_batch", "pp_fourview2d_nonminimal_batch", "pp_fourview2d_least_squares", "pp_fourview2d_lomsac",
]

_lib = None


def exported_symbols():
    """Every symbol include/ppsfm_hip.h declares."""
    return list(_EXPORTS)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if _build.is_stale():
        _build.build_library()
    if not os.path.exists(_build.LIB):
        raise ImportError("libppsfm_hip.so is missing and could not be built (hipcc required); "
                          "there is no CPU fallback for the product path")
    L = C.CDLL(_build.LIB)
    L.pp_last_error.restype = C.c_char_p
    L.pp_ransac_compute_num_trials.restype = C.c_uint64
    L.pp_ransac_compute_num_trials.argtypes = [C.c_uint64, C.c_uint64, C.c_double, C.c_double]
    L.pp_ba_create.argtypes = [C.POINTER(BAProblemDesc), C.c_int, C.POINTER(C.c_void_p)]
    L.pp_ba_destroy.argtypes = [C.c_void_p]
    L.pp_ba_set_parameters.argtypes = [C.c_void_p, c_dp, c_dp, c_dp]
    L.pp_ba_get_parameters.argtypes = [C.c_void_p, c_dp, c_dp, c_dp]
    L.pp_ba_eval.argtypes = [C.c_void_p, C.c_int, C.c_int, c_dp, c_dp, c_dp, c_dp, c_dp]
    L.pp_ba_eval_host_view.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(c_dp), C.POINTER(c_dp), C.POINTER(c_dp), C.POINTER(c_dp), c_dp]
    L.pp_ba_eval_device.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
    L.pp_ba_solve.argtypes = [C.c_void_p, C.POINTER(BAOptions), C.POINTER(BASummary)]
    L.pp_ba_get_trace.argtypes = [C.c_void_p, c_dp, C.c_int32, c_ip]
    L.pp_ba_get_structure.argtypes = [C.c_void_p, c_ip]
    L.pp_ba_plan_ordering.argtypes = [C.POINTER(BAProblemDesc), c_ip, c_ip]
    L.pp_ba_pair_lists_host.argtypes = [C.POINTER(BAProblemDesc), C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), c_ip, c_ip, c_ip, C.c_int64, C.c_int64]
    L.pp_ba_covisibility.argtypes = [C.POINTER(BAProblemDesc), c_u8p]
    L.pp_ba_get_create_profile.argtypes = [C.c_void_p, c_dp]
    L.pp_ba_reduced_system.argtypes = [C.c_void_p, C.POINTER(BAOptions), C.c_double, c_ip, c_dp, c_dp, C.c_int64]
    L.pp_ba_set_allreduce.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
    L.pp_ba_set_communicator.argtypes = [C.c_void_p, C.c_void_p]
    L.pp_comm_unique_id.argtypes = [c_u8p]
    L.pp_comm_create.argtypes = [c_u8p, C.c_int32, C.c_int32, C.c_int, C.POINTER(C.c_void_p)]
    L.pp_comm_destroy.argtypes = [C.c_void_p]
    L.pp_comm_allreduce.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]
    L.pp_ba_get_timings.argtypes = [C.c_void_p, c_dp, c_ip]
    L.pp_dense_cholesky_solve.argtypes = [C.c_int32, c_dp, c_dp, c_dp, C.c_int, C.c_int32, C.POINTER(C.c_float)]
    L.pp_cholesky_task_list.argtypes = [C.c_int32, C.POINTER(C.c_int32), C.c_int64, C.POINTER(C.c_int64)]
    L.pp_cholesky_task_list_sparse.argtypes = [C.c_int32, c_u8p, c_u8p, C.POINTER(C.c_int32), C.c_int64, C.POINTER(C.c_int64)]
    L.pp_cholesky_task_plan.argtypes = [C.c_int32, c_u8p, C.c_int32, c_u8p, C.POINTER(C.c_int32), C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                                        C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.pp_pose_create.argtypes = [C.c_int32, c_dp, c_dp, c_u8p, C.c_int, C.POINTER(C.c_void_p)]
    L.pp_pose_destroy.argtypes = [C.c_void_p]
    L.pp_pose_residuals.argtypes = [C.c_void_p, C.c_int32, c_dp, c_dp]
    L.pp_pose_score.argtypes = [C.c_void_p, C.c_int32, c_dp, C.c_double, c_u32p, c_dp]
    L.pp_pose_support_sequential.argtypes = [C.c_void_p, C.c_int32, c_dp, C.c_double, c_u32p, c_dp]
    L.pp_pose_p6l_batch.argtypes = [C.c_void_p, C.c_int64, c_u32p, c_dp, c_ip]
    L.pp_re3q3_batch.argtypes = [C.c_int64, c_dp, c_dp, c_ip, C.c_int]
    L.pp_pose_ransac.argtypes = [C.c_void_p, C.POINTER(RansacOptions), C.POINTER(RansacReport), c_u8p]
    L.pp_pose_hypotheses.argtypes = [C.c_void_p, C.c_int64, c_u32p, C.c_uint32, C.c_double, C.POINTER(RansacReport)]
    L.pp_pose_last_scores.argtypes = [C.c_void_p, C.c_int64, c_ip, c_u32p, c_dp]
    L.pp_debug_raise.argtypes = [C.c_int]
    L.pp_sampler_draw.argtypes = [C.c_uint32, C.c_uint32, C.c_int32, C.c_int64, c_u32p]
    L.pp_planar_create.argtypes = [C.c_int32, c_dp, c_dp, c_dp, C.c_int, C.POINTER(C.c_void_p)]
    L.pp_planar_destroy.argtypes = [C.c_void_p]
    L.pp_planar_solve_batch.argtypes = [C.c_void_p, C.c_int64, C.c_int32, c_ip, c_dp]
    L.pp_planar_score.argtypes = [C.c_void_p, C.c_int32, c_dp, C.c_double, c_dp, c_ip]
    L.pp_planar_evaluate.argtypes = [C.c_void_p, c_dp, c_dp, c_dp, c_dp]
    L.pp_planar_lomsac.argtypes = [C.c_void_p, C.POINTER(LoMsacOptions), C.POINTER(LoMsacReport), c_dp, c_dp, c_ip]
    L.pp_triangulate_tracks.argtypes = [C.c_int, C.c_int32, c_ip, c_dp, c_ip, C.c_int32, c_dp, c_dp, c_ip, C.c_int32, c_ip, c_dp, c_ip,
                                        C.POINTER(TriangulationOptions), c_u8p, c_dp, c_u8p, c_ip, C.POINTER(C.c_float)]
    L.pp_ba_filter_points.argtypes = [C.c_void_p, C.POINTER(FilterOptions), c_u8p, c_ip, c_u8p, c_u8p, c_u8p, c_dp, C.POINTER(FilterReport)]
    L.pp_ba_filter_negative_depth.argtypes = [C.c_void_p, c_u8p, C.POINTER(C.c_int64)]
    L.pp_pose2d_create.argtypes = [C.c_int32, c_dp, c_dp, C.c_int, C.POINTER(C.c_void_p)]
    L.pp_pose2d_destroy.argtypes = [C.c_void_p]
    L.pp_pose2d_solve_batch.argtypes = [C.c_void_p, C.c_int64, C.c_int32, c_ip, c_dp]
    L.pp_pose2d_score.argtypes = [C.c_void_p, C.c_int32, c_dp, C.c_double, c_dp, c_ip]
    L.pp_pose2d_evaluate.argtypes = [C.c_void_p, c_dp, c_dp]
    L.pp_pose2d_lomsac.argtypes = [C.c_void_p, C.POINTER(LoMsacOptions), C.POINTER(LoMsacReport), c_dp, c_ip]
    L.pp_fourview2d_create.argtypes = [C.c_int32, c_dp, C.c_int, C.POINTER(C.c_void_p)]
    L.pp_fourview2d_destroy.argtypes = [C.c_void_p]
    L.pp_fourview2d_score.argtypes = [C.c_void_p, C.c_int32, c_dp, C.c_double, c_dp, c_ip]
    L.pp_fourview2d_evaluate.argtypes = [C.c_void_p, c_dp, c_dp, c_dp]
    L.pp_fourview2d_evaluate_points.argtypes = [C.c_void_p, c_dp, c_dp, c_dp]
    L.pp_fourview2d_default_frames.argtypes = [c_dp]
    L.pp_fourview2d_minimal_batch.argtypes = [C.c_void_p, C.c_int64, C.c_int32, c_ip, c_dp, c_dp, c_ip]
    L.pp_fourview2d_nonminimal_batch.argtypes = [C.c_void_p, C.c_int64, C.c_int32, c_ip, c_dp, C.c_double, c_dp, c_dp, c_ip]
    L.pp_fourview2d_least_squares.argtypes = [C.c_void_p, C.c_int32, c_ip, c_dp, c_dp]
    L.pp_fourview2d_lomsac.argtypes = [C.c_void_p, C.POINTER(LoMsacOptions), c_dp, C.POINTER(LoMsacReport), c_dp, c_dp, c_ip]
    L.pp_camera_image_to_world_threshold.argtypes = [C.c_int, c_dp, C.c_double, c_dp]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise PPError(rc, lib().pp_last_error().decode(errors="replace"))


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def dp(a):
    return None if a is None else a.ctypes.data_as(c_dp)


def ptr(a, t):
    return None if a is None else a.ctypes.data_as(t)
