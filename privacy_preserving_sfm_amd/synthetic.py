"""Deterministic synthetic line-feature scenes.

Recipes follow the reference's own test generators — reference src/init/initializer_test.cc:52-137
(`setup_plausible_scene`, `setup_random_lines`: a 2D line through the projection x~ of a point is
l = x~ x n for a random direction n) and src/feature/extraction.cc:499-503 (lines are scaled so that
||(a,b)|| = 1) — but with explicit numpy seeds instead of rand()/std::random_device.

All arrays are plain numpy, laid out as the C-ABI of include/ppsfm_hip.h expects.
"""
import numpy as np

NUM_PARAMS = (3, 4, 4, 5, 8, 8, 12, 5, 4, 5, 12)
CAM_STRIDE = 12


def quat_to_rot(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def rot_to_quat(R):
    """Rotation matrix -> (w,x,y,z), w >= 0."""
    tr = np.trace(R)
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = np.zeros(4)
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    if q[0] < 0:
        q = -q
    return q / np.linalg.norm(q)


def _small_rot(w):
    th = np.linalg.norm(w)
    if th < 1e-300:
        return np.eye(3)
    k = w / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def default_intrinsics(model):
    """One camera of the given model with README-style parameters (w x h = 1280 x 960)."""
    f, cx, cy = 1000.0, 640.0, 480.0
    p = np.zeros(CAM_STRIDE)
    table = {
        0: [f, cx, cy], 1: [f, 1010.0, cx, cy], 2: [f, cx, cy, 0.01], 3: [f, cx, cy, 0.01, -0.002],
        4: [f, 1010.0, cx, cy, 0.01, -0.002, 0.0005, -0.0003],
        5: [f, 1010.0, cx, cy, 0.01, -0.002, 0.0005, -0.0001],
        6: [f, 1010.0, cx, cy, 0.01, -0.002, 0.0005, -0.0003, 0.001, 0.005, -0.001, 0.0002],
        7: [f, 1010.0, cx, cy, 0.3], 8: [f, cx, cy, 0.01], 9: [f, cx, cy, 0.01, -0.002],
        10: [f, 1010.0, cx, cy, 0.01, -0.002, 0.0005, -0.0003, 0.0002, -0.0001, 0.0003, -0.0002],
    }
    v = table[model]
    p[: len(v)] = v
    return p


def make_ba_scene(num_cams, num_points, track, seed=0xC0FFEE, model=2, num_intrinsics=1,
                  noise_point=1e-2, noise_t=1e-3, noise_q=1e-3, sort="point", window=None, loop=False, clusters=None, bridge=4,
                  topology="star"):
    """Cameras on a circle of radius 4 looking at the origin (+ jitter), points uniform in [-1,1]^3,
    every point observed by `track` distinct cameras; start = ground truth + noise.  Gauge as the
    reference's global BA: pose[0] constant, tvec[1].x constant (sfm/incremental_mapper.cc:922-926).

    `window` (None = every camera can see every point): a point is observed by `track` cameras out of `window` CONSECUTIVE
    ones (a sequence: images only share points with their neighbours), which makes the reduced camera system block-banded.
    `loop`: the sequence closes (the window wraps around from the last image to the first one - a loop closure): a ring, no order
    of the images makes it a narrow band.
    `clusters` = k: a photo collection - k groups of images (contiguous ids; shuffle_image_ids hides that), every point is seen by `track` images of ONE
    group, and the groups are joined by a few images only: `bridge` images of a group also see points of the group it is attached to - the first group
    (`topology` "star": a hub with satellites, which no order of the images turns into a narrow band) or the previous group ("chain").

    Returns a dict with ground truth (`gt_*`) and perturbed start (`poses`, `points`, `intr`).
    """
    rng = np.random.default_rng(seed)
    C, P = int(num_cams), int(num_points)
    poses_gt = np.zeros((C, 7))
    Rs = np.zeros((C, 3, 3))
    for c in range(C):
        ang = 2 * np.pi * c / C
        centre = np.array([4 * np.cos(ang), 0.3 * np.sin(3 * ang), 4 * np.sin(ang)])
        zc = -centre / np.linalg.norm(centre)
        up = np.array([0.0, 1.0, 0.0])
        xc = np.cross(up, zc); xc /= np.linalg.norm(xc)
        yc = np.cross(zc, xc)
        R = np.stack([xc, yc, zc])            # world -> camera
        R = _small_rot(rng.uniform(-0.1, 0.1, 3)) @ R
        Rs[c] = R
        poses_gt[c, :4] = rot_to_quat(R)
        poses_gt[c, 4:] = -R @ centre
    points_gt = rng.uniform(-1, 1, (P, 3))
    intr = np.stack([default_intrinsics(model) for _ in range(num_intrinsics)])
    pose_camera = (np.arange(C) % num_intrinsics).astype(np.int32)
    camera_model = np.full(num_intrinsics, model, dtype=np.int32)

    # tracks: `track` distinct cameras per point
    obs_point = np.repeat(np.arange(P, dtype=np.int32), track)
    obs_pose = np.empty(P * track, dtype=np.int32)
    if clusters:
        k = int(clusters)
        bounds = [(C * j) // k for j in range(k + 1)]
        members = [np.arange(bounds[j], bounds[j + 1]) for j in range(k)]
        seen_by = [list(m) for m in members]            # the images that may see a point of group j: its own + the bridges attached to it
        for j in range(1, k):
            target = 0 if topology == "star" else j - 1
            seen_by[target] = seen_by[target] + list(members[j][: int(bridge)])
        seen_by = [np.array(v) for v in seen_by]
        group_of_point = rng.integers(0, k, P)
        # (the first images of the first group keep a share of its points: the gauge images 0 and 1 must see something)
        for p in range(P):
            cand = seen_by[group_of_point[p]]
            obs_pose[p * track:(p + 1) * track] = np.sort(rng.choice(cand, size=track, replace=False))
    for p in range(P if not clusters else 0):
        if window is None:
            obs_pose[p * track:(p + 1) * track] = np.sort(rng.choice(C, size=track, replace=False))
        else:
            first = int(rng.integers(0, C if loop else C - int(window) + 1))
            obs_pose[p * track:(p + 1) * track] = np.sort((first + rng.choice(int(window), size=track, replace=False)) % C) if loop else \
                first + np.sort(rng.choice(int(window), size=track, replace=False))
    M = P * track
    Xc = np.einsum("mij,mj->mi", Rs[obs_pose], points_gt[obs_point]) + poses_gt[obs_pose, 4:]
    assert np.all(Xc[:, 2] > 0.5)
    xh = np.stack([Xc[:, 0] / Xc[:, 2], Xc[:, 1] / Xc[:, 2], np.ones(M)], axis=1)
    n = rng.uniform(-1, 1, (M, 3))
    lines = np.cross(xh, n)
    lines /= np.linalg.norm(lines[:, :2], axis=1, keepdims=True)

    if sort == "pose":
        order = np.lexsort((obs_point, obs_pose))
        obs_pose, obs_point, lines = obs_pose[order], obs_point[order], lines[order]

    poses = poses_gt.copy()
    points = points_gt + rng.normal(0, noise_point, (P, 3))
    for c in range(1, C):
        R = _small_rot(rng.normal(0, noise_q, 3)) @ Rs[c]
        poses[c, :4] = rot_to_quat(R)
        poses[c, 4:] = poses_gt[c, 4:] + rng.normal(0, noise_t, 3)
    poses[1, 4] = poses_gt[1, 4]  # the gauge-fixed component starts at its true value

    pose_const = np.zeros(C, dtype=np.uint8); pose_const[0] = 1
    tvec_const_mask = np.zeros(C, dtype=np.uint8); tvec_const_mask[1] = 1
    point_const = np.zeros(P, dtype=np.uint8)
    camera_const_mask = np.full(num_intrinsics, 0xFFFF, dtype=np.uint16)
    return dict(lines=np.ascontiguousarray(lines), obs_pose=np.ascontiguousarray(obs_pose),
                obs_point=np.ascontiguousarray(obs_point), pose_camera=pose_camera, camera_model=camera_model,
                poses=poses, points=points, intr=intr, gt_poses=poses_gt, gt_points=points_gt,
                pose_const=pose_const, tvec_const_mask=tvec_const_mask, point_const=point_const,
                camera_const_mask=camera_const_mask, loss_type=0, loss_scale=1.0)


def shuffle_image_ids(scene, seed=0):
    """The same scene with its image ids in random order (image `old` becomes image new_of_old[old]): what a reconstruction looks
    like whose images were not registered in capture order.  Returns (scene, new_of_old)."""
    rng = np.random.default_rng(seed)
    C = scene["poses"].shape[0]
    new_of_old = rng.permutation(C).astype(np.int32)
    old_of_new = np.empty(C, dtype=np.int32)
    old_of_new[new_of_old] = np.arange(C, dtype=np.int32)
    out = dict(scene)
    out["obs_pose"] = np.ascontiguousarray(new_of_old[scene["obs_pose"]])
    for k in ("poses", "gt_poses", "pose_camera", "pose_const", "tvec_const_mask"):
        if k in scene:
            out[k] = np.ascontiguousarray(scene[k][old_of_new])
    return out, new_of_old


def make_ransac_scene(n, outlier_ratio=0.5, noise_px=0.5, focal=1000.0, seed=0xBADC0DE, aligned_ratio=0.0):
    """One ground-truth pose; n points in [-1,1]^2 x [2,6] (camera frame of the identity pose, then
    moved by the inverse pose so the world frame is generic); inlier lines pass through the noisy
    projection with random direction; outliers are random lines through random image points."""
    rng = np.random.default_rng(seed)
    R = _small_rot(rng.uniform(-0.5, 0.5, 3))
    t = rng.uniform(-0.5, 0.5, 3)
    Xc = np.stack([rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), rng.uniform(2, 6, n)], axis=1)
    Xw = (Xc - t) @ R          # R^T (Xc - t)
    x = Xc[:, :2] / Xc[:, 2:3] + rng.normal(0, noise_px / focal, (n, 2))
    is_out = rng.uniform(size=n) < outlier_ratio
    x[is_out] = rng.uniform(-0.5, 0.5, (int(is_out.sum()), 2))
    th = rng.uniform(0, 2 * np.pi, n)
    a, b = np.cos(th), np.sin(th)
    c = -(a * x[:, 0] + b * x[:, 1])
    lines = np.stack([a, b, c], axis=1)
    aligned = (rng.uniform(size=n) < aligned_ratio).astype(np.uint8)
    P = np.concatenate([R, t[:, None]], axis=1)
    return dict(lines=np.ascontiguousarray(lines), points=np.ascontiguousarray(Xw), aligned=aligned,
                gt_pose=P, is_outlier=is_out, max_error=12.0 / focal)


def _rot_y(th):
    c, s = np.cos(th), np.sin(th)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def make_planar_offset_scene(npts, n_outliers=0, seed=0, noise=0.0):
    """Recipe of reference src/init/initializer_test.cc:52-137, 234-286 (PlanarOffsetEstimator tests): four upright
    cameras (rotation about y), first camera identity, points in front of all cameras, a random line through
    every projection (l = x~ x n, unit 3-vector).  The poses handed to the estimator have t_y zeroed; the truth is
    `t_gt`.  Rg = identity.  Outliers: the line of view 3 is replaced by a random line."""
    rng = np.random.default_rng(seed)
    while True:
        cams = [np.hstack([np.eye(3), np.zeros((3, 1))])]
        for i in range(1, 4):
            t = rng.uniform(-1, 1, 3)
            if i == 1:
                t /= np.linalg.norm(t)
            cams.append(np.hstack([_rot_y(rng.uniform(-0.5, 0.5)), t[:, None]]))
        X = rng.uniform(-1, 1, (npts, 3))
        X[:, 2] = np.abs(X[:, 2]) + 2.5
        z = [X @ c[:, :3].T + c[:, 3] for c in cams]
        if all((zz[:, 2] > 0.2).all() for zz in z):
            break
    lines = np.zeros((4, npts, 3))
    for j in range(4):
        xh = z[j] / z[j][:, 2:3]
        xh[:, :2] += rng.normal(0, noise, (npts, 2)) if noise > 0 else 0
        l = np.cross(xh, rng.uniform(-1, 1, (npts, 3)))
        lines[j] = l / np.linalg.norm(l, axis=1, keepdims=True)
    is_out = np.zeros(npts, dtype=bool)
    if n_outliers:
        is_out[rng.choice(npts, n_outliers, replace=False)] = True
        l = rng.uniform(-1, 1, (n_outliers, 3))
        lines[3, is_out] = l / np.linalg.norm(l, axis=1, keepdims=True)
    t_gt = np.array([cams[1][1, 3], cams[2][1, 3], cams[3][1, 3]])
    poses = np.array(cams)
    poses[1:, 1, 3] = 0.0
    return dict(poses=poses, lines=lines, Rg=np.tile(np.eye(3), (4, 1, 1)), t_gt=t_gt, gt_cams=np.array(cams), X=X, is_outlier=is_out)


def make_scene_2d(ncams, npts, n_outliers=0, seed=0):
    """Recipe of reference src/init/sfm2d_test.cc:50-110: 2D cameras [R2 | t] (first = identity), 2D points in front
    of every camera, bearings x = normalize(P X~); outliers get a random bearing in every view but the first."""
    rng = np.random.default_rng(seed)
    while True:
        cams = [np.array([[1.0, 0, 0], [0, 1.0, 0]])]
        for i in range(1, ncams):
            th = rng.uniform(-0.6, 0.6)
            R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
            cams.append(np.hstack([R, rng.uniform(-1, 1, (2, 1))]))
        X = np.stack([rng.uniform(-1, 1, npts), rng.uniform(2, 5, npts)], axis=1)
        z = [X @ c[:, :2].T + c[:, 2] for c in cams]
        if all((zz[:, 1] > 0.2).all() for zz in z):
            break
    x = np.array([zz / np.linalg.norm(zz, axis=1, keepdims=True) for zz in z])
    is_out = np.zeros(npts, dtype=bool)
    if n_outliers:
        is_out[rng.choice(npts, n_outliers, replace=False)] = True
        for j in range(1, ncams):
            b = rng.uniform(-1, 1, (n_outliers, 2)); b[:, 1] = np.abs(b[:, 1]) + 0.2
            x[j, is_out] = b / np.linalg.norm(b, axis=1, keepdims=True)
    return dict(cams=np.array(cams), X=X, x=x, is_outlier=is_out)


def make_init_scene(npts, ngrav, n_outliers=0, seed=0, gravity_noise=0.0):
    """Recipe of reference src/init/initializer_test.cc:43-157, 346-470 (Initializer* tests): four UPRIGHT cameras
    (first = identity, |t_1| = 1), points in front of all of them, `ngrav` tracks observed as gravity-aligned lines
    l = normalize(x~ x g_i) and the rest as random lines through the projection; outliers: the projection of a track
    is replaced by a random point in one random view before the lines are drawn."""
    rng = np.random.default_rng(seed)
    while True:
        cams = [np.hstack([np.eye(3), np.zeros((3, 1))])]
        for i in range(1, 4):
            t = rng.uniform(-1, 1, 3)
            if i == 1:
                t /= np.linalg.norm(t)
            cams.append(np.hstack([_rot_y(rng.uniform(-0.6, 0.6)), t[:, None]]))
        X = rng.uniform(-1, 1, (npts, 3))
        X[:, 2] = np.abs(X[:, 2]) + 1.2      # the reference draws Z in [-1,1]^3 with |z|: close points, strong geometry
        z = [X @ c[:, :3].T + c[:, 3] for c in cams]
        if all((zz[:, 2] > 0.2).all() for zz in z):
            break
    xs = [zz[:, :2] / zz[:, 2:3] for zz in z]
    is_out = np.zeros(npts, dtype=bool)
    if n_outliers:
        idx = rng.choice(npts, n_outliers, replace=False)
        is_out[idx] = True
        for i in idx:
            xs[rng.integers(0, 4)][i] = rng.uniform(-1, 1, 2)
    order = rng.permutation(npts)
    aligned_track = order < ngrav
    lines, aligned, gravity = [], [], []
    for i in range(4):
        g = cams[i][:, 1].copy()
        if gravity_noise > 0:
            ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
            K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
            g = (np.eye(3) + np.sin(gravity_noise) * K + (1 - np.cos(gravity_noise)) * K @ K) @ g
        gravity.append(g)
        xh = np.hstack([xs[i], np.ones((npts, 1))])
        n = rng.uniform(-1, 1, (npts, 3))
        l = np.where(aligned_track[:, None], np.cross(xh, g), np.cross(xh, n))
        lines.append(l / np.linalg.norm(l, axis=1, keepdims=True))
        aligned.append(aligned_track.copy())
    return dict(cams=np.array(cams), X=X, lines=lines, aligned=aligned, gravity=np.array(gravity), is_outlier=is_out)


def make_track_scene(num_views, num_tracks, seed=0, min_len=3, max_len=12, noise=2e-4, outlier_frac=0.15, model=2):
    """Views on a circle looking at the origin, one 3D point per track observed as a line (random direction through its
    projection, normalised coordinates, a^2+b^2 = 1) in `len` distinct views; a fraction of the observations gets a random
    line instead.  For the batched track triangulation (reference estimators/triangulation.cc)."""
    rng = np.random.default_rng(seed)
    P, centers = [], []
    for v in range(num_views):
        ang = 2 * np.pi * v / num_views + rng.normal(0, 0.05)
        c = np.array([4.0 * np.cos(ang), rng.normal(0, 0.3), 4.0 * np.sin(ang)])
        z = -c / np.linalg.norm(c)
        x = np.cross([0, 1.0, 0], z); x /= np.linalg.norm(x)
        y = np.cross(z, x)
        R = np.stack([x, y, z])
        P.append(np.hstack([R, (-R @ c)[:, None]])); centers.append(c)
    P = np.array(P); centers = np.array(centers)
    pts = rng.uniform(-1, 1, (num_tracks, 3))
    track_start = [0]; lines, obs_view, is_out = [], [], []
    for t in range(num_tracks):
        n = int(rng.integers(min_len, max_len + 1))
        views = rng.choice(num_views, min(n, num_views), replace=False)
        for v in views:
            z = P[v][:, :3] @ pts[t] + P[v][:, 3]
            xh = np.array([z[0] / z[2], z[1] / z[2], 1.0])
            xh[:2] += rng.normal(0, noise, 2)
            out = rng.random() < outlier_frac
            if out:
                xh[:2] = rng.uniform(-0.4, 0.4, 2)
            l = np.cross(xh, rng.uniform(-1, 1, 3))
            lines.append(l / np.linalg.norm(l[:2])); obs_view.append(int(v)); is_out.append(out)
        track_start.append(len(lines))
    f = 800.0
    intr = np.zeros((1, 12)); intr[0, :4] = [f, 640.0, 480.0, 0.01] if model == 2 else [f, 640.0, 480.0, 0.0]
    return dict(track_start=np.array(track_start, dtype=np.int32), lines=np.array(lines), obs_view=np.array(obs_view, dtype=np.int32), P=P, centers=centers,
                view_camera=np.zeros(num_views, dtype=np.int32), camera_model=np.array([model], dtype=np.int32), intr=intr,
                cam_size=np.array([[1280, 960]], dtype=np.int32), points=pts, is_outlier=np.array(is_out))
