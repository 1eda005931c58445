"""ctypes bindings to the CPU oracle (oracle/libppsfm_oracle.so) — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")
_ASAN = os.environ.get("PPSFM_ORACLE_ASAN") == "1"      # the sanitizer build (make -C oracle asan; libasan must be preloaded into the interpreter)
_LIB_PATH = os.path.join(_ORACLE_DIR, "libppsfm_oracle_asan.so" if _ASAN else "libppsfm_oracle.so")

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int32)
c_u8p = C.POINTER(C.c_uint8)
c_u16p = C.POINTER(C.c_uint16)
c_u32p = C.POINTER(C.c_uint32)


def build():
    subprocess.check_call(["make", "-s", "-C", _ORACLE_DIR, os.path.basename(_LIB_PATH)])


def _dp(a):
    return None if a is None else a.ctypes.data_as(c_dp)


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


class BAProblemC(C.Structure):
    _fields_ = [("num_poses", C.c_int32), ("num_points", C.c_int32), ("num_cameras", C.c_int32), ("loss_type", C.c_int32),
                ("num_obs", C.c_int64), ("loss_scale", C.c_double),
                ("lines", c_dp), ("obs_pose", c_ip), ("obs_point", c_ip), ("pose_camera", c_ip), ("camera_model", c_ip),
                ("pose_const", c_u8p), ("tvec_const_mask", c_u8p), ("point_const", c_u8p), ("camera_const_mask", c_u16p)]


class BAOptionsC(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int32), ("max_num_consecutive_invalid_steps", C.c_int32),
                ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
                ("initial_trust_region_radius", C.c_double), ("max_trust_region_radius", C.c_double),
                ("min_trust_region_radius", C.c_double), ("min_relative_decrease", C.c_double),
                ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
                ("jacobi_scaling", C.c_int32), ("blocked_cholesky", C.c_int32),
                ("iterative_schur", C.c_int32), ("max_linear_solver_iterations", C.c_int32), ("eta", C.c_double)]

    @staticmethod
    def defaults(**kw):
        o = BAOptionsC(100, 10, 0.0, 0.0, 0.0, 1e4, 1e16, 1e-32, 1e-3, 1e-6, 1e32, 1, 0, 0, 200, 0.1)
        for k, v in kw.items():
            setattr(o, k, v)
        return o


class BASummaryC(C.Structure):
    _fields_ = [("initial_cost", C.c_double), ("final_cost", C.c_double), ("num_successful_steps", C.c_int32),
                ("num_unsuccessful_steps", C.c_int32), ("termination", C.c_int32), ("num_iterations", C.c_int32),
                ("time_s", C.c_double), ("linear_solver_iterations", C.c_int32), ("reserved", C.c_int32)]


class RansacOptionsC(C.Structure):
    _fields_ = [("max_error", C.c_double), ("min_inlier_ratio", C.c_double), ("confidence", C.c_double),
                ("dyn_num_trials_multiplier", C.c_double), ("min_num_trials", C.c_uint64), ("max_num_trials", C.c_uint64)]


class RansacReportC(C.Structure):
    _fields_ = [("success", C.c_int32), ("best_model_idx", C.c_int32), ("num_trials", C.c_uint64), ("num_inliers", C.c_uint64),
                ("residual_sum", C.c_double), ("model", C.c_double * 12), ("best_trial", C.c_int64), ("time_s", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_ba_cost.restype = C.c_double
        _lib.orc_p6l_hypotheses_timed.restype = C.c_double
        _lib.orc_compute_num_trials.restype = C.c_uint64
        _lib.orc_compute_num_trials.argtypes = [C.c_uint64, C.c_uint64, C.c_double, C.c_double]
        _lib.orc_world_to_image.argtypes = [C.c_int, c_dp, C.c_double, C.c_double, c_dp]
    return _lib


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def world_to_image(model, params, u, v):
    params = f64(params)
    out = np.zeros(2)
    lib().orc_world_to_image(int(model), _dp(params), float(u), float(v), _dp(out))
    return out


def line_cost(model, line, q, t, X, cam):
    n = lib().orc_num_params(int(model))
    line, q, t, X, cam = f64(line), f64(q), f64(t), f64(X), f64(cam)
    r = np.zeros(2); Jq = np.zeros((2, 4)); Jt = np.zeros((2, 3)); JX = np.zeros((2, 3)); Jc = np.zeros((2, n))
    rc = lib().orc_line_cost(int(model), _dp(line), _dp(q), _dp(t), _dp(X), _dp(cam), _dp(r), _dp(Jq), _dp(Jt), _dp(JX), _dp(Jc))
    assert rc == 0
    return r, Jq, Jt, JX, Jc


def ba_eval(scene, ambient=False, want_cam=False, cam_stride=12):
    """scene: dict with lines, obs_pose, obs_point, pose_camera, camera_model, poses, points, intr."""
    M = len(scene["obs_pose"])
    r = np.zeros(2 * M); Jp = np.zeros((M, 14 if ambient else 12)); Jx = np.zeros((M, 6))
    Jc = np.zeros((M, 2 * cam_stride)) if want_cam else None
    lib().orc_ba_eval(C.c_int64(M), _dp(scene["lines"]), _p(scene["obs_pose"], c_ip), _p(scene["obs_point"], c_ip),
                      _p(scene["pose_camera"], c_ip), _p(scene["camera_model"], c_ip), _dp(scene["poses"]),
                      _dp(scene["points"]), _dp(scene["intr"]), int(ambient), _dp(r), _dp(Jp), _dp(Jx), _dp(Jc), int(cam_stride))
    return r, Jp, Jx, Jc


def make_problem(scene):
    """Returns (BAProblemC, keepalive list)."""
    keep = []

    def k(a):
        keep.append(a)
        return a
    p = BAProblemC()
    p.num_poses = scene["poses"].shape[0]; p.num_points = scene["points"].shape[0]; p.num_cameras = scene["intr"].shape[0]
    p.loss_type = int(scene.get("loss_type", 0)); p.loss_scale = float(scene.get("loss_scale", 1.0))
    p.num_obs = len(scene["obs_pose"])
    p.lines = _dp(k(f64(scene["lines"]))); p.obs_pose = _p(k(i32(scene["obs_pose"])), c_ip)
    p.obs_point = _p(k(i32(scene["obs_point"])), c_ip); p.pose_camera = _p(k(i32(scene["pose_camera"])), c_ip)
    p.camera_model = _p(k(i32(scene["camera_model"])), c_ip)
    p.pose_const = _p(k(np.ascontiguousarray(scene["pose_const"], dtype=np.uint8)), c_u8p)
    p.tvec_const_mask = _p(k(np.ascontiguousarray(scene["tvec_const_mask"], dtype=np.uint8)), c_u8p)
    p.point_const = _p(k(np.ascontiguousarray(scene["point_const"], dtype=np.uint8)), c_u8p)
    p.camera_const_mask = _p(k(np.ascontiguousarray(scene["camera_const_mask"], dtype=np.uint16)), c_u16p)
    return p, keep


def ba_solve(scene, options=None, trace_cap=256):
    p, keep = make_problem(scene)
    o = options or BAOptionsC.defaults()
    poses, points, intr = f64(scene["poses"]).copy(), f64(scene["points"]).copy(), f64(scene["intr"]).copy()
    s = BASummaryC()
    trace = np.zeros((trace_cap, 7))
    lib().orc_ba_solve(C.byref(p), C.byref(o), _dp(poses), _dp(points), _dp(intr), C.byref(s), _dp(trace), trace_cap)
    return poses, points, intr, s, trace[: s.num_iterations + 1]


def ba_cost(scene, poses=None, points=None, intr=None):
    p, keep = make_problem(scene)
    poses = f64(scene["poses"] if poses is None else poses); points = f64(scene["points"] if points is None else points)
    intr = f64(scene["intr"] if intr is None else intr)
    r = np.zeros(2 * p.num_obs)
    c = lib().orc_ba_cost(C.byref(p), _dp(poses), _dp(points), _dp(intr), _dp(r))
    return c, r


def ba_reduced_system(scene, radius, options=None):
    p, keep = make_problem(scene)
    o = options or BAOptionsC.defaults()
    poses, points, intr = f64(scene["poses"]).copy(), f64(scene["points"]).copy(), f64(scene["intr"]).copy()
    C_, P_ = poses.shape[0], points.shape[0]
    ncmax = 6 * C_ + 12 * intr.shape[0]
    S = np.zeros(ncmax * ncmax); rhs = np.zeros(ncmax); step = np.zeros(ncmax + 3 * P_)
    scale = np.zeros(ncmax + 3 * P_); grad = np.zeros(ncmax + 3 * P_)
    npc = C.c_int32(0)
    nc = lib().orc_ba_reduced_system(C.byref(p), C.byref(o), C.c_double(radius), _dp(poses), _dp(points), _dp(intr),
                                     _dp(S), _dp(rhs), _dp(step), _dp(scale), _dp(grad), C.byref(npc))
    assert nc >= 0
    n = nc + npc.value
    return dict(nc=nc, np=npc.value, S=S[: nc * nc].reshape(nc, nc).copy(), rhs=rhs[:nc].copy(), step=step[:n].copy(),
                scale=scale[:n].copy(), grad=grad[:n].copy())


def line_residuals(lines, pts, P):
    lines, pts, P = f64(lines), f64(pts), f64(P)
    n = lines.shape[0]
    out = np.zeros(n)
    lib().orc_line_residuals(n, _dp(lines), _dp(pts), _dp(P), _dp(out))
    return out


def support(residuals, max_residual):
    residuals = f64(residuals)
    ni = C.c_uint64(0); rs = C.c_double(0)
    lib().orc_support(len(residuals), _dp(residuals), C.c_double(max_residual), C.byref(ni), C.byref(rs))
    return ni.value, rs.value


def support_better(n1, sum1, n2, sum2):
    f = lib().orc_support_better
    f.restype = C.c_int; f.argtypes = [C.c_uint64, C.c_double, C.c_uint64, C.c_double]
    return bool(f(int(n1), float(sum1), int(n2), float(sum2)))


def re3q3(coeffs, affine=None):
    coeffs = f64(coeffs).reshape(3, 10)
    sol = np.zeros((3, 8))
    aff = None if affine is None else f64(affine)
    n = lib().orc_re3q3(_dp(coeffs), _dp(sol), _dp(aff))
    return sol[:, :n].copy()


def p6l(lines6, points6, aligned6=None, mix=None, affine=None):
    lines6, points6 = f64(lines6), f64(points6)
    al = np.zeros(6, dtype=np.uint8) if aligned6 is None else np.ascontiguousarray(aligned6, dtype=np.uint8)
    models = np.zeros((8, 12))
    n = lib().orc_p6l(_dp(lines6), _dp(points6), _p(al, c_u8p), _dp(models), _dp(None if mix is None else f64(mix)),
                      _dp(None if affine is None else f64(affine)))
    return models[:n].reshape(n, 3, 4).copy()


def sampler(seed, n, k, count):
    out = np.zeros((count, k), dtype=np.uint32)
    lib().orc_sampler(C.c_uint32(seed), C.c_uint32(n), int(k), C.c_int64(count), _p(out, c_u32p))
    return out


def mt19937(seed, count):
    out = np.zeros(count, dtype=np.uint32)
    lib().orc_mt19937(C.c_uint32(seed), C.c_int64(count), _p(out, c_u32p))
    return out


def std_uniform(seed, lo, hi):
    lo = np.ascontiguousarray(lo, dtype=np.uint32); hi = np.ascontiguousarray(hi, dtype=np.uint32)
    a = np.zeros(len(lo), dtype=np.uint32); b = np.zeros(len(lo), dtype=np.uint32)
    lib().orc_std_uniform(C.c_uint32(seed), C.c_int64(len(lo)), _p(lo, c_u32p), _p(hi, c_u32p), _p(a, c_u32p), _p(b, c_u32p))
    return a, b


def compute_num_trials(num_inliers, num_samples, confidence, mult):
    return lib().orc_compute_num_trials(int(num_inliers), int(num_samples), float(confidence), float(mult))


def p6l_ransac(lines, pts, aligned, max_error, seed=0, min_inlier_ratio=0.1, confidence=0.99, mult=3.0,
               min_num_trials=0, max_num_trials=2**64 - 1):
    lines, pts = f64(lines), f64(pts)
    n = lines.shape[0]
    al = None if aligned is None else np.ascontiguousarray(aligned, dtype=np.uint8)
    o = RansacOptionsC(max_error, min_inlier_ratio, confidence, mult, min_num_trials, max_num_trials)
    rep = RansacReportC()
    mask = np.zeros(n, dtype=np.uint8)
    lib().orc_p6l_ransac(C.byref(o), n, _dp(lines), _dp(pts), _p(al, c_u8p), C.c_uint32(seed), C.byref(rep), _p(mask, c_u8p))
    return rep, mask


def p6l_hypotheses_timed(lines, pts, aligned, samples, max_residual):
    lines, pts = f64(lines), f64(pts)
    samples = np.ascontiguousarray(samples, dtype=np.uint32)
    al = None if aligned is None else np.ascontiguousarray(aligned, dtype=np.uint8)
    nm = C.c_int64(0); best = C.c_uint64(0)
    t = lib().orc_p6l_hypotheses_timed(lines.shape[0], _dp(lines), _dp(pts), _p(al, c_u8p), C.c_int64(samples.shape[0]),
                                       _p(samples, c_u32p), C.c_double(max_residual), C.byref(nm), C.byref(best))
    return t, nm.value, best.value


# ---- four-view initialisation path ---------------------------------------------------------------------
class LoMsacOptionsC(C.Structure):
    _fields_ = [("min_num_iterations", C.c_uint32), ("max_num_iterations", C.c_uint32), ("success_probability", C.c_double),
                ("squared_inlier_threshold", C.c_double), ("random_seed", C.c_uint32), ("final_least_squares", C.c_int32)]

    @staticmethod
    def defaults(**kw):
        o = LoMsacOptionsC(100, 10000, 0.9999, 1.0, 0, 0)
        for k, v in kw.items():
            setattr(o, k, v)
        return o


class LoMsacStatsC(C.Structure):
    _fields_ = [("num_iterations", C.c_uint32), ("best_num_inliers", C.c_int32), ("best_model_score", C.c_double),
                ("inlier_ratio", C.c_double), ("number_lo_iterations", C.c_int32), ("pad", C.c_int32)]


def lomsac_line_trace(n=200):
    buf = C.create_string_buffer(8192)
    r = lib().orc_lomsac_line_trace(int(n), buf, 8192)
    assert r > 0
    return buf.value.decode()


def abspose2d_nonminimal(x, X, sample):
    x, X = f64(x), f64(X)
    s = i32(sample)
    P = np.zeros((2, 3))
    lib().orc_abspose2d_nonminimal(_dp(x), _dp(X), x.shape[0], _p(s, c_ip), len(s), _dp(P))
    return P


def abspose2d_lomsac(x, X, options=None):
    x, X = f64(x), f64(X)
    o = options or LoMsacOptionsC.defaults()
    P = np.zeros((2, 3)); st = LoMsacStatsC(); idx = np.zeros(x.shape[0], dtype=np.int32)
    inl = lib().orc_abspose2d_lomsac(_dp(x), _dp(X), x.shape[0], C.byref(o), _dp(P), C.byref(st), _p(idx, c_ip))
    return inl, P, st, idx[:inl].copy()


def fourview2d_score(cams, x, thr):
    cams, x = f64(cams).reshape(24), f64(x)
    n = x.shape[1]
    X = np.zeros((n, 2)); err = np.zeros(n); inl = C.c_int32(0)
    lib().orc_fourview2d_score.restype = C.c_double
    sc = lib().orc_fourview2d_score(_dp(cams), _dp(x), n, C.c_double(thr), _dp(X), _dp(err), C.byref(inl))
    return sc, inl.value, err, X


def planar_minimal(scene, samples, want_cams=False):
    poses, lines, Rg = f64(scene["poses"]).reshape(48), f64(scene["lines"]), f64(scene["Rg"]).reshape(36)
    samples = i32(samples)
    num, k = samples.shape
    off = np.zeros((num, 3)); cams = np.zeros((num, 4, 12)) if want_cams else None
    lib().orc_planar_minimal(_dp(poses), _dp(lines), lines.shape[1], _dp(Rg), _p(samples, c_ip), num, k, _dp(off), _dp(cams))
    return (off, cams.reshape(num, 4, 3, 4)) if want_cams else off


def planar_score(scene, offsets, thr):
    poses, lines, Rg = f64(scene["poses"]).reshape(48), f64(scene["lines"]), f64(scene["Rg"]).reshape(36)
    n = lines.shape[1]
    X = np.zeros((n, 3)); err = np.zeros(n); inl = C.c_int32(0)
    lib().orc_planar_score.restype = C.c_double
    sc = lib().orc_planar_score(_dp(poses), _dp(lines), n, _dp(Rg), _dp(f64(offsets)), C.c_double(thr), _dp(X), _dp(err), C.byref(inl))
    return sc, inl.value, err, X


def planar_lomsac(scene, options):
    poses, lines, Rg = f64(scene["poses"]).reshape(48), f64(scene["lines"]), f64(scene["Rg"]).reshape(36)
    n = lines.shape[1]
    cams = np.zeros((4, 12)); st = LoMsacStatsC(); idx = np.zeros(n, dtype=np.int32)
    inl = lib().orc_planar_lomsac(_dp(poses), _dp(lines), n, _dp(Rg), C.byref(options), _dp(cams), C.byref(st), _p(idx, c_ip))
    return inl, cams.reshape(4, 3, 4), st, idx[:max(inl, 0)].copy()


def fourview2d_minimal(x, samples, A123):
    x = f64(x); samples = i32(samples); A123 = f64(A123).reshape(12)
    num, k = samples.shape
    cams = np.zeros((num, 16, 4, 6)); cnt = np.zeros(num, dtype=np.int32)
    lib().orc_fourview2d_minimal(_dp(x), x.shape[1], _p(samples, c_ip), num, k, _dp(A123), _dp(cams), _p(cnt, c_ip))
    return cams.reshape(num, 16, 4, 2, 3), cnt


def fourview2d_least_squares(x, sample, frames, cams, X):
    x = f64(x); s = i32(sample); fr = f64(frames).reshape(12)
    cams = f64(cams).reshape(24).copy(); X = f64(X).copy()
    n = x.shape[1]
    lib().orc_fourview2d_least_squares(_dp(x), n, _p(s, c_ip), len(s), _dp(fr), _dp(cams), _dp(X))
    return cams.reshape(4, 2, 3), X


def fourview2d_lomsac(x, frames, options=None):
    x = f64(x); fr = f64(frames).reshape(12)
    n = x.shape[1]
    o = options or LoMsacOptionsC.defaults()
    cams = np.zeros(24); X = np.zeros((n, 2)); st = LoMsacStatsC(); idx = np.zeros(n, dtype=np.int32)
    inl = lib().orc_fourview2d_lomsac(_dp(x), n, C.byref(o), _dp(fr), _dp(cams), _dp(X), C.byref(st), _p(idx, c_ip))
    return inl, cams.reshape(4, 2, 3), X, st, idx[:inl].copy()


def filter_points3d(scene, max_reproj_error, min_tri_angle_deg, cam_size, obs_aligned, point_subset=None):
    k = lambda a: a
    lines = k(f64(scene["lines"])); op = k(i32(scene["obs_pose"])); oj = k(i32(scene["obs_point"]))
    pc = k(i32(scene["pose_camera"])); cm = k(i32(scene["camera_model"])); cs = k(i32(cam_size))
    poses = k(f64(scene["poses"])); pts = k(f64(scene["points"])); intr = k(f64(scene["intr"]))
    al = k(np.ascontiguousarray(obs_aligned, dtype=np.uint8))
    sub = None if point_subset is None else k(np.ascontiguousarray(point_subset, dtype=np.uint8))
    M, P, Cn = len(op), pts.shape[0], poses.shape[0]
    od = np.zeros(M, dtype=np.uint8); pd = np.zeros(P, dtype=np.uint8); pe = np.zeros(P)
    c_u8p = C.POINTER(C.c_uint8)
    f = lib().orc_filter_points3d
    f.restype = C.c_int64
    nf = f(C.c_int64(M), P, Cn, _dp(lines), _p(op, c_ip), _p(oj, c_ip), _p(al, c_u8p), _p(pc, c_ip), _p(cm, c_ip), _p(cs, c_ip), _dp(poses), _dp(pts), _dp(intr),
           C.c_double(max_reproj_error), C.c_double(min_tri_angle_deg), None if sub is None else _p(sub, c_u8p), _p(od, c_u8p), _p(pd, c_u8p), _dp(pe))
    return int(nf), od.astype(bool), pd.astype(bool), pe


def filter_negative_depth(scene):
    op = i32(scene["obs_pose"]); oj = i32(scene["obs_point"]); poses = f64(scene["poses"]); pts = f64(scene["points"])
    neg = np.zeros(len(op), dtype=np.uint8)
    f = lib().orc_filter_negative_depth
    f.restype = C.c_int64
    n = f(C.c_int64(len(op)), _p(op, c_ip), _p(oj, c_ip), _dp(poses), _dp(pts), _p(neg, C.POINTER(C.c_uint8)))
    return int(n), neg.astype(bool)


class TriangulationOptionsC(C.Structure):
    _fields_ = [("min_tri_angle", C.c_double), ("residual_type", C.c_int32), ("pad", C.c_int32), ("ransac", RansacOptionsC)]


def triangulate_tracks(sc, min_tri_angle, residual_type, **ransac_kw):
    o = TriangulationOptionsC()
    o.min_tri_angle = min_tri_angle; o.residual_type = residual_type
    o.ransac = RansacOptionsC(0.0, 0.1, 0.99, 3.0, 0, 2 ** 64 - 1)
    for k, v in ransac_kw.items():
        setattr(o.ransac, k, v)
    ts = i32(sc["track_start"]); T = len(ts) - 1
    ln = f64(sc["lines"]); ov = i32(sc["obs_view"]); P = f64(sc["P"]).reshape(-1, 12); ctr = f64(sc["centers"]); vc = i32(sc["view_camera"])
    cm = i32(sc["camera_model"]); it = f64(sc["intr"]); cs = i32(sc["cam_size"])
    ok = np.zeros(T, dtype=np.uint8); xyz = np.zeros((T, 3)); mask = np.zeros(len(ov), dtype=np.uint8); nt = np.zeros(T, dtype=np.int32)
    u8 = C.POINTER(C.c_uint8)
    lib().orc_triangulate_tracks(T, _p(ts, c_ip), _dp(ln), _p(ov, c_ip), P.shape[0], _dp(P), _dp(ctr), _p(vc, c_ip), _p(cm, c_ip), _dp(it), _p(cs, c_ip), C.byref(o),
                                 _p(ok, u8), _dp(xyz), _p(mask, u8), _p(nt, c_ip))
    return ok.astype(bool), xyz, mask.astype(bool), nt


def powell_trace(max_num_iterations=100, function_tolerance=1e-6, gradient_tolerance=1e-10, parameter_tolerance=1e-8):
    """oracle/trust_region.h on Powell's function (Ceres' examples/powell.cc): (rows of {cost, cost_change, |gradient|, |step|, tr_ratio, tr_radius, ok}, final x)"""
    x = np.zeros(4); tr = np.zeros((512, 7))
    f = lib().orc_powell_trace
    f.argtypes = [c_dp, C.c_int, C.c_double, C.c_double, C.c_double, c_dp, C.c_int]
    n = f(_dp(x), int(max_num_iterations), function_tolerance, gradient_tolerance, parameter_tolerance, _dp(tr), 512)
    return tr[:n].copy(), x


def helloworld_trace(max_num_iterations=50, function_tolerance=1e-6, gradient_tolerance=1e-10, parameter_tolerance=1e-8):
    """the same for Ceres' examples/helloworld.cc (f = 10 - x from x = 0.5)"""
    x = np.zeros(1); tr = np.zeros((64, 7))
    f = lib().orc_helloworld_trace
    f.argtypes = [c_dp, C.c_int, C.c_double, C.c_double, C.c_double, c_dp, C.c_int]
    n = f(_dp(x), int(max_num_iterations), function_tolerance, gradient_tolerance, parameter_tolerance, _dp(tr), 64)
    return tr[:n].copy(), x
