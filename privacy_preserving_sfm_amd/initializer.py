"""Host mirror of the reference's four-view initialisation entry point

    bool init::initialize_reconstruction(lines, gravity, options, &poses, &inlier_ratio)     (src/init/initializer.cc:58-216)

Control flow, thresholds and return conditions follow the reference line by line (including its two quirks: the mean
triangulation angle is computed in DEGREES but compared with `min_tri_angle`, documented "in rad", :186-190; and the
planar-offset inlier count is compared with `min_tri_angle`, :209-210).  The two RansacLib LO-MSAC runs execute on the
device through the C-ABI (`pp_fourview2d_lomsac`, `pp_planar_lomsac`); this module is host bookkeeping only."""
import numpy as np

from .device import FourView2dProblem, PlanarOffsetProblem, lomsac_options


class InitOptions:
    """src/init/initializer.h:48-58"""

    def __init__(self):
        self.min_tri_angle = 0.1        # "Minimum mean triangulation angle (in rad)"
        self.min_num_inliers = 6
        self.max_error = 0.005          # in normalised coordinates


def from_two_vectors(a, b):
    """Eigen::Quaterniond::FromTwoVectors(a, b).toRotationMatrix(): the rotation taking a to b."""
    v0 = np.asarray(a, dtype=np.float64) / np.linalg.norm(a)
    v1 = np.asarray(b, dtype=np.float64) / np.linalg.norm(b)
    c = float(v0 @ v1)
    if c < -1.0 + 1e-12:                # antiparallel: half turn about any axis orthogonal to a
        axis = np.linalg.svd(np.stack([v0, v1]))[2][2]
        w2 = (1.0 + c) * 0.5
        q = np.concatenate([[np.sqrt(max(w2, 0.0))], axis * np.sqrt(1.0 - w2)])
    else:
        axis = np.cross(v0, v1)
        s = np.sqrt((1.0 + c) * 2.0)
        q = np.concatenate([[s * 0.5], axis / s])
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def lift_camera(pose2d):
    """initializer.cc:45-56: 2D pose [a -b tx; b a tz] acting on (x, z) -> 3x4 pose with t_y = 0."""
    P = np.zeros((3, 4))
    P[0, 0] = pose2d[0, 0]; P[0, 2] = pose2d[0, 1]
    P[2, 0] = pose2d[1, 0]; P[2, 2] = pose2d[1, 1]
    P[1, 1] = 1.0
    P[0, 3] = pose2d[0, 2]; P[2, 3] = pose2d[1, 2]
    return P


def initialize_reconstruction(lines, aligned, gravity, options=None, device=0, frames=None, random_seed=0):
    """lines: 4 arrays [n_i, 3] (FeatureLine::Line()), aligned: 4 bool arrays (FeatureLine::IsAligned()), gravity: 4 x 3.
    Returns (ok, poses [4,3,4] or None, inlier_ratio)."""
    options = options or InitOptions()
    x = [[], [], [], []]
    lines_r = [[], [], [], []]
    Rg = []
    for i in range(4):
        Rg.append(from_two_vectors(gravity[i], [0.0, 1.0, 0.0]))
        for l, al in zip(np.asarray(lines[i], dtype=np.float64), np.asarray(aligned[i], dtype=bool)):
            if al:
                l = Rg[i] @ l                                   # only the aligned lines are pre-rotated
                assert abs(l[1]) <= 1e-6, "CHECK_NEAR(l(1), 0.0, 1e-6)"
                xl = np.array([l[2], -l[0]])
                if xl[1] < 0:
                    xl = -xl
                x[i].append(xl / np.linalg.norm(xl))
            else:
                lines_r[i].append(l)
    assert len(x[0]) == len(x[1]) == len(x[2]) == len(x[3])
    assert len(lines_r[0]) == len(lines_r[1]) == len(lines_r[2]) == len(lines_r[3])
    x = np.array(x).reshape(4, -1, 2)
    lines_r = np.array(lines_r).reshape(4, -1, 3)
    Rg = np.array(Rg)

    # Estimate a four view reconstruction (initializer.cc:114-127)
    opts = lomsac_options(final_least_squares=1, min_num_iterations=1000, squared_inlier_threshold=options.max_error, random_seed=random_seed)
    if x.shape[1] < 5:
        return False, None, 0.0
    fv = FourView2dProblem(x, device=device)
    try:
        rep, cams2d, X2d, inl = fv.lomsac(opts, frames=frames)
    finally:
        fv.close()
    if rep.best_num_inliers < options.min_num_inliers:
        return False, None, 0.0

    # mean minimum triangulation angle over the first three cameras (initializer.cc:157-190)
    centers = [-cams2d[c][:, :2].T @ cams2d[c][:, 2] for c in range(3)]
    angle_sum = 0.0
    for i in inl:
        best = np.inf
        for c1 in range(3):
            for c2 in range(c1 + 1, 3):
                v1 = centers[c1] - X2d[i]; v2 = centers[c2] - X2d[i]
                ang = np.arccos(np.clip((v1 / np.linalg.norm(v1)) @ (v2 / np.linalg.norm(v2)), -1.0, 1.0))
                best = min(best, ang)
        angle_sum += best
    mean_tri_angle = (angle_sum / len(inl)) / np.pi * 180.0
    if mean_tri_angle < options.min_tri_angle:
        return False, None, 0.0

    # lift to 3D (only t_y is missing) and estimate the out-of-plane translations (initializer.cc:192-215)
    poses = np.array([lift_camera(cams2d[c]) for c in range(4)])
    if lines_r.shape[1] < 3:
        return False, None, 0.0
    pp = PlanarOffsetProblem(poses, lines_r, Rg, device=device)
    try:
        rep3, offsets, cams3d, inl3 = pp.lomsac(opts)
    finally:
        pp.close()
    if rep3.best_num_inliers < options.min_tri_angle:
        return False, None, 0.0
    return bool(rep3.best_num_inliers >= options.min_num_inliers), cams3d, float(rep3.inlier_ratio)
