"""Host mirror of the reference's absolute-pose estimation interface on top of the C ABI.

  RANSACOptions / RANSAC<P6LEstimator>::Report     reference src/optim/ransac.h:47-99
  P6LEstimator (Estimator concept)                 src/estimators/absolute_pose.h:48-80
  EstimateAbsolutePoseFromLines                    src/estimators/pose.cc:52-94
"""
import numpy as np

from .device import PoseProblem, ransac_options


class RANSACOptions:
    def __init__(self):
        self.max_error = 0.0
        self.min_inlier_ratio = 0.1
        self.confidence = 0.99
        self.dyn_num_trials_multiplier = 3.0
        self.min_num_trials = 0
        self.max_num_trials = 2**64 - 1

    def Check(self):
        assert self.max_error > 0
        assert 0 <= self.min_inlier_ratio <= 1
        assert 0 <= self.confidence <= 1
        assert self.min_num_trials <= self.max_num_trials


def _split(lines2D):
    """Accepts FeatureLine-like objects (Line(), IsAligned()) or a plain (n,3) array."""
    if len(lines2D) and hasattr(lines2D[0], "Line"):
        L = np.array([l.Line() for l in lines2D], dtype=np.float64).reshape(-1, 3)
        al = np.array([1 if l.IsAligned() else 0 for l in lines2D], dtype=np.uint8)
        return L, al
    L = np.asarray(lines2D, dtype=np.float64).reshape(-1, 3)
    return L, np.zeros(L.shape[0], dtype=np.uint8)


class P6LEstimator:
    """X_t = FeatureLine, Y_t = Vector3d, M_t = Matrix3x4d, kMinNumSamples = 6."""
    kMinNumSamples = 6

    def __init__(self, device=0):
        self.device = device

    def Estimate(self, lines2D, points3D):
        L, al = _split(lines2D)
        assert L.shape[0] == 6
        pp = PoseProblem(L, np.asarray(points3D, dtype=np.float64).reshape(6, 3), al, device=self.device)
        try:
            models, nm = pp.p6l_batch(np.arange(6, dtype=np.uint32)[None])
        finally:
            pp.close()
        return [models[0, k].copy() for k in range(nm[0])]

    def Residuals(self, lines2D, points3D, proj_matrix):
        L, al = _split(lines2D)
        pp = PoseProblem(L, np.asarray(points3D, dtype=np.float64).reshape(-1, 3), al, device=self.device)
        try:
            return pp.residuals(np.asarray(proj_matrix, dtype=np.float64).reshape(1, 12))[0]
        finally:
            pp.close()


class Support:
    def __init__(self, num_inliers=0, residual_sum=np.finfo(np.float64).max):
        self.num_inliers, self.residual_sum = num_inliers, residual_sum


class Report:
    def __init__(self):
        self.success = False
        self.num_trials = 0
        self.support = Support()
        self.inlier_mask = np.zeros(0, dtype=np.uint8)
        self.model = np.zeros((3, 4))


class RANSAC:
    """RANSAC<P6LEstimator, InlierSupportMeasurer, RandomSampler> (optim/ransac.h:78-278)."""

    def __init__(self, options, seed=0, device=0):
        options.Check()
        self.options_, self.seed, self.device = options, seed, device
        self.estimator = P6LEstimator(device)

    def Estimate(self, X, Y):
        L, al = _split(X)
        Y = np.asarray(Y, dtype=np.float64).reshape(-1, 3)
        assert L.shape[0] == Y.shape[0]          # CHECK_EQ(X.size(), Y.size())
        o = self.options_
        pp = PoseProblem(L, Y, al, device=self.device)
        try:
            rep, mask = pp.ransac(ransac_options(max_error=o.max_error, min_inlier_ratio=o.min_inlier_ratio, confidence=o.confidence,
                                                 dyn_num_trials_multiplier=o.dyn_num_trials_multiplier,
                                                 min_num_trials=o.min_num_trials, max_num_trials=o.max_num_trials, seed=self.seed))
        finally:
            pp.close()
        r = Report()
        r.success = bool(rep.success)
        r.num_trials = int(rep.num_trials)
        r.support = Support(int(rep.num_inliers), float(rep.residual_sum))
        r.inlier_mask = mask.copy() if r.success else np.zeros(0, dtype=np.uint8)
        r.model = np.array(rep.model).reshape(3, 4)
        return r


def RotationMatrixToQuaternion(R):
    """base/pose.cc:41-51 (Eigen::Quaterniond(rot_mat) -> (w,x,y,z)); no sign normalisation."""
    R = np.asarray(R, dtype=np.float64)
    t = np.trace(R)
    q = np.zeros(4)
    if t > 0:
        t = np.sqrt(t + 1.0)
        q[0] = 0.5 * t
        t = 0.5 / t
        q[1] = (R[2, 1] - R[1, 2]) * t
        q[2] = (R[0, 2] - R[2, 0]) * t
        q[3] = (R[1, 0] - R[0, 1]) * t
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        t = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q[1 + i] = 0.5 * t
        t = 0.5 / t
        q[0] = (R[k, j] - R[j, k]) * t
        q[1 + j] = (R[j, i] + R[i, j]) * t
        q[1 + k] = (R[k, i] + R[i, k]) * t
    return q


def EstimateAbsolutePoseFromLines(options, lines2D, points3D, seed=0, device=0):
    """pose.cc:52-94.  Returns (ok, qvec, tvec, num_inliers, inlier_mask)."""
    options.Check()
    report = RANSAC(options, seed=seed, device=device).Estimate(lines2D, points3D)
    num_inliers = report.support.num_inliers
    inlier_mask = report.inlier_mask
    if num_inliers == 0:
        return False, None, None, 0, inlier_mask
    _, al = _split(lines2D)
    # more than 90 % gravity-aligned inliers: the pose is likely off (pose.cc:71-83)
    num_aligned_inliers = int(np.sum((inlier_mask != 0) & (al != 0))) if len(inlier_mask) else 0
    if num_aligned_inliers > num_inliers * 0.9:
        return False, None, None, num_inliers, inlier_mask
    qvec = RotationMatrixToQuaternion(report.model[:, :3])
    tvec = report.model[:, 3].copy()
    if np.isnan(qvec).any() or np.isnan(tvec).any():
        return False, None, None, num_inliers, inlier_mask
    return True, qvec, tvec, num_inliers, inlier_mask


class AbsolutePoseRefinementOptions:
    """src/estimators/pose.h:84-108"""

    def __init__(self):
        self.gradient_tolerance = 1.0
        self.max_num_iterations = 100
        self.loss_function_scale = 1.0
        self.refine_focal_length = False
        self.refine_extra_params = False
        self.print_summary = True

    def Check(self):
        assert self.gradient_tolerance >= 0.0 and self.max_num_iterations >= 0 and self.loss_function_scale >= 0.0


def refine_pose_scene(options, inlier_mask, lines2D, points3D, qvec, tvec, camera):
    """The flat BA problem RefineAbsolutePoseFromLines builds (src/estimators/pose.cc:96-176): one pose, one camera,
    the inlier observations of CONSTANT points, Cauchy loss; the camera block is constant unless a refine flag is set,
    then the principal point (always) and the non-refined groups are held by the SubsetParameterization."""
    mask = np.asarray(inlier_mask).astype(bool)
    lines = np.asarray(lines2D, dtype=np.float64).reshape(-1, 3)
    pts = np.asarray(points3D, dtype=np.float64).reshape(-1, 3)
    assert len(mask) == len(lines) == len(pts)
    lines, pts = lines[mask], pts[mask]
    n = len(lines)
    q = np.asarray(qvec, dtype=np.float64)
    q = q / np.linalg.norm(q)                                  # NormalizeQuaternion (:143)
    npar = camera.NumParams()
    if not options.refine_focal_length and not options.refine_extra_params:
        const_bits = 0xFFFF
    else:
        idxs = list(camera.PrincipalPointIdxs())
        if not options.refine_focal_length:
            idxs += list(camera.FocalLengthIdxs())
        if not options.refine_extra_params:
            idxs += list(camera.ExtraParamsIdxs())
        const_bits = 0xFFFF if len(set(idxs)) == npar else sum(1 << i for i in set(idxs))
    intr = np.zeros((1, 12)); intr[0, :npar] = camera.params
    return dict(lines=lines, obs_pose=np.zeros(n, dtype=np.int32), obs_point=np.arange(n, dtype=np.int32), pose_camera=np.zeros(1, dtype=np.int32),
                camera_model=np.array([camera.model_id], dtype=np.int32), poses=np.concatenate([q, np.asarray(tvec, dtype=np.float64)])[None, :],
                points=pts.copy(), intr=intr, pose_const=np.zeros(1, dtype=np.uint8), tvec_const_mask=np.zeros(1, dtype=np.uint8),
                point_const=np.ones(n, dtype=np.uint8), camera_const_mask=np.array([const_bits], dtype=np.uint16), loss_type=2,
                loss_scale=float(options.loss_function_scale))


def RefineAbsolutePoseFromLines(options, inlier_mask, lines2D, points3D, qvec, tvec, camera, device=0):
    """src/estimators/pose.cc:96-213 — SURVEY.md §8(f) rank 2: a one-camera instance of the device BA (K1 + K2 with no
    variable point, 6 (+ intrinsics) columns, Cauchy loss).  Solver options: gradient tolerance and iteration cap from
    `options`, everything else Ceres' defaults (function tolerance 1e-6, parameter tolerance 1e-8, 5 invalid steps).
    qvec, tvec and camera.params are updated in place; returns Summary::IsSolutionUsable()."""
    from . import _capi
    from .device import BAProblem, ba_options
    options.Check()
    scene = refine_pose_scene(options, inlier_mask, lines2D, points3D, qvec, tvec, camera)
    if scene["lines"].shape[0] == 0:
        return True, None         # no residuals: Ceres reports a usable (trivially converged) solution
    pb = BAProblem(scene, device=device)
    try:
        try:
            summary = pb.solve(ba_options(max_num_iterations=options.max_num_iterations, gradient_tolerance=options.gradient_tolerance,
                                          function_tolerance=1e-6, parameter_tolerance=1e-8, max_num_consecutive_invalid_steps=5))
            usable = True
        except _capi.PPError as e:
            if e.code != _capi.PP_ERR_NUMERIC:
                raise
            summary, usable = e.summary, False       # termination FAILURE: not IsSolutionUsable(), parameters stay as given
        poses, _, intr = pb.get_parameters()
    finally:
        pb.close()
    if not usable:
        return usable, summary
    qvec[:] = poses[0, :4]
    tvec[:] = poses[0, 4:]
    camera.params = intr[0, : camera.NumParams()].copy()
    return usable, summary
