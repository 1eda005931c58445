"""Host mirror of the reference's bundle-adjustment interface on top of the C ABI.

Same names, argument meaning and error behaviour as reference src/optim/bundle_adjustment.{h,cc}:
`BundleAdjustmentOptions` (:49-100), `BundleAdjustmentConfig` (:103-167), `BundleAdjuster` (:171-203), and the
minimal data model it reads raw doubles from (`Reconstruction`, `Image`, `Camera`, `Point3D`, `FeatureLine`:
base/image.h, base/camera.h, base/point3d.h, feature/types.h:98-138).  `BundleAdjuster.Solve` performs exactly
the SetUp of bundle_adjustment.cc:326-542 (which observations exist, which blocks are constant), flattens it to
the arrays of `pp_ba_problem_desc`, and hands the solve to the device (`pp_ba_solve`).
"""
import numpy as np

from . import _capi
from .device import BAProblem, ba_options

kInvalidPoint3DId = -1


class FeatureLine:
    """feature/types.h:98-138: a 2D line (a,b,c) in normalised coordinates, gravity-alignment flag, 3D point id."""

    def __init__(self, line, is_aligned=False, point3D_id=kInvalidPoint3DId):
        self._line = np.asarray(line, dtype=np.float64).copy()
        self._aligned = bool(is_aligned)
        self.point3D_id = point3D_id

    def Line(self):
        return self._line

    def IsAligned(self):
        return self._aligned

    def HasPoint3D(self):
        return self.point3D_id != kInvalidPoint3DId

    def Point3DId(self):
        return self.point3D_id


class Camera:
    def __init__(self, camera_id, model_id, params):
        self.camera_id, self.model_id = camera_id, int(model_id)
        n = _capi.lib().pp_camera_num_params(self.model_id)
        if n < 0:
            raise ValueError("camera model %d does not exist" % model_id)   # CAMERA_MODEL_DOES_NOT_EXIST_EXCEPTION
        self.params = np.asarray(params, dtype=np.float64).copy()
        assert self.params.shape == (n,)

    def ModelId(self):
        return self.model_id

    def NumParams(self):
        return len(self.params)

    def FocalLengthIdxs(self):
        return [0] if self.model_id in (0, 2, 3, 8, 9) else [0, 1]

    def PrincipalPointIdxs(self):
        return [1, 2] if self.model_id in (0, 2, 3, 8, 9) else [2, 3]

    def ExtraParamsIdxs(self):
        first = 3 if self.model_id in (0, 2, 3, 8, 9) else 4
        return list(range(first, len(self.params)))


class Image:
    def __init__(self, image_id, camera_id, qvec, tvec, lines=()):
        self.image_id, self.camera_id = image_id, camera_id
        self.qvec = np.asarray(qvec, dtype=np.float64).copy()
        self.tvec = np.asarray(tvec, dtype=np.float64).copy()
        self.lines = list(lines)

    def CameraId(self):
        return self.camera_id

    def NormalizeQvec(self):
        n = np.linalg.norm(self.qvec)
        self.qvec = np.array([1.0, 0, 0, 0]) if n == 0 else self.qvec / n      # base/pose.cc NormalizeQuaternion

    def Lines(self):
        return self.lines

    def Line(self, idx):
        return self.lines[idx]


class Point3D:
    def __init__(self, xyz, track=()):
        self.xyz = np.asarray(xyz, dtype=np.float64).copy()
        self.track = list(track)          # [(image_id, line_idx)]
        self.error = -1.0                 # Point3D::Error(), set by FilterPoints3DWithLargeReprojectionError


class Reconstruction:
    def __init__(self):
        self.cameras, self.images, self.points3D = {}, {}, {}

    def Camera(self, cid):
        return self.cameras[cid]

    def Image(self, iid):
        return self.images[iid]

    def Point3D(self, pid):
        return self.points3D[pid]

    # ---- filters run after every bundle adjustment (base/reconstruction.cc:425-460, 594-719) on the device ------------
    def _filter_scene(self):
        """flat problem over ALL registered images and points (every observation of every point)"""
        image_ids = sorted(self.images)
        point_ids = sorted(self.points3D)
        cam_ids = sorted(self.cameras)
        pose_index = {iid: k for k, iid in enumerate(image_ids)}
        point_index = {pid: k for k, pid in enumerate(point_ids)}
        cam_index = {cid: k for k, cid in enumerate(cam_ids)}
        lines, obs_pose, obs_point, aligned, obs_ref = [], [], [], [], []
        for pid in point_ids:                       # observations in track order: the track IS the per-point list
            for (iid, idx) in self.points3D[pid].track:
                fl = self.images[iid].lines[idx]
                lines.append(fl.Line()); obs_pose.append(pose_index[iid]); obs_point.append(point_index[pid]); aligned.append(bool(fl.IsAligned()))
                obs_ref.append((iid, idx))
        intr = np.zeros((len(cam_ids), 12))
        for cid, k in cam_index.items():
            intr[k, : self.cameras[cid].NumParams()] = self.cameras[cid].params
        scene = dict(lines=np.array(lines, dtype=np.float64).reshape(-1, 3), obs_pose=np.array(obs_pose, dtype=np.int32), obs_point=np.array(obs_point, dtype=np.int32),
                     pose_camera=np.array([cam_index[self.images[i].camera_id] for i in image_ids], dtype=np.int32),
                     camera_model=np.array([self.cameras[c].model_id for c in cam_ids], dtype=np.int32),
                     poses=np.array([np.concatenate([self.images[i].qvec, self.images[i].tvec]) for i in image_ids]),
                     points=np.array([self.points3D[p].xyz for p in point_ids]), intr=intr)
        cam_size = np.array([[getattr(self.cameras[c], "width", 1 << 30), getattr(self.cameras[c], "height", 1 << 30)] for c in cam_ids], dtype=np.int32)
        return scene, np.array(aligned, dtype=bool), cam_size, point_ids, obs_ref

    def DeleteObservation(self, image_id, line_idx):
        fl = self.images[image_id].lines[line_idx]
        pid = fl.point3D_id
        self.points3D[pid].track = [t for t in self.points3D[pid].track if t != (image_id, line_idx)]
        fl.point3D_id = kInvalidPoint3DId

    def DeletePoint3D(self, pid):
        for (iid, idx) in self.points3D[pid].track:
            self.images[iid].lines[idx].point3D_id = kInvalidPoint3DId
        del self.points3D[pid]

    def FilterPoints3D(self, max_reproj_error, min_tri_angle, point3D_ids=None, device=0):
        """Reconstruction::FilterPoints3D / FilterAllPoints3D (point3D_ids = None): returns the number of filtered
        observations as the reference counts them; points and observations are deleted, Point3D.error is set."""
        from .device import BAProblem
        if not self.points3D:
            return 0
        scene, aligned, cam_size, point_ids, obs_ref = self._filter_scene()
        if len(obs_ref) == 0:
            return 0
        subset = None if point3D_ids is None else np.array([p in set(point3D_ids) for p in point_ids], dtype=np.uint8)
        pb = BAProblem(scene, device=device)
        try:
            rep, od, pd, pe = pb.filter_points(max_reproj_error, min_tri_angle, cam_size, obs_aligned=aligned, point_subset=subset)
        finally:
            pb.close()
        for k, pid in enumerate(point_ids):
            if pd[k]:
                self.DeletePoint3D(pid)
            elif pe[k] >= 0:
                self.points3D[pid].error = float(pe[k])
        for o, (iid, idx) in enumerate(obs_ref):
            pid = point_ids[scene["obs_point"][o]]
            if od[o] and pid in self.points3D:
                self.DeleteObservation(iid, idx)
        return int(rep.num_filtered)

    def FilterAllPoints3D(self, max_reproj_error, min_tri_angle, device=0):
        return self.FilterPoints3D(max_reproj_error, min_tri_angle, None, device=device)

    def FilterObservationsWithNegativeDepth(self, device=0):
        from .device import BAProblem
        if not self.points3D:
            return 0
        scene, aligned, cam_size, point_ids, obs_ref = self._filter_scene()
        if len(obs_ref) == 0:
            return 0
        pb = BAProblem(scene, device=device)
        try:
            n, neg = pb.filter_negative_depth()
        finally:
            pb.close()
        for o, (iid, idx) in enumerate(obs_ref):
            if neg[o]:
                self.DeleteObservation(iid, idx)
        return n

    @staticmethod
    def from_scene(scene):
        """Builds the object model from the flat synthetic scene (images 0..C-1, one line per observation)."""
        rec = Reconstruction()
        for k in range(scene["intr"].shape[0]):
            m = int(scene["camera_model"][k])
            rec.cameras[k] = Camera(k, m, scene["intr"][k, : _capi.lib().pp_camera_num_params(m)])
        for c in range(scene["poses"].shape[0]):
            rec.images[c] = Image(c, int(scene["pose_camera"][c]), scene["poses"][c, :4], scene["poses"][c, 4:])
        for p in range(scene["points"].shape[0]):
            rec.points3D[p] = Point3D(scene["points"][p])
        for o in range(len(scene["obs_pose"])):
            c, p = int(scene["obs_pose"][o]), int(scene["obs_point"][o])
            rec.images[c].lines.append(FeatureLine(scene["lines"][o], False, p))
            rec.points3D[p].track.append((c, len(rec.images[c].lines) - 1))
        return rec


class SolverOptions:
    """The ceres::Solver::Options fields the reference sets (bundle_adjustment.h:80-93)."""

    def __init__(self):
        self.function_tolerance = 0.0
        self.gradient_tolerance = 0.0
        self.parameter_tolerance = 0.0
        self.minimizer_progress_to_stdout = False
        self.max_num_iterations = 100
        self.max_linear_solver_iterations = 200
        self.max_num_consecutive_invalid_steps = 10
        self.max_consecutive_nonmonotonic_steps = 10
        self.num_threads = -1
        # Solver::Options::callbacks: the reference pushes ONE ceres::IterationCallback (controllers/bundle_adjustment.cc:87-88)
        self.iteration_callback = None


class BundleAdjustmentOptions:
    TRIVIAL, SOFT_L1, CAUCHY = 0, 1, 2      # enum class LossFunctionType

    def __init__(self):
        self.loss_function_type = self.TRIVIAL
        self.loss_function_scale = 1.0
        self.refine_focal_length = False
        self.refine_principal_point = False
        self.refine_extra_params = False
        self.refine_extrinsics = True
        self.print_summary = True
        self.min_num_residuals_for_multi_threading = 50000
        self.solver_options = SolverOptions()

    def Check(self):
        if not self.loss_function_scale >= 0:       # CHECK_OPTION_GE(loss_function_scale, 0)
            return False
        return True


class BundleAdjustmentConfig:
    """bundle_adjustment.h:103-167: which images / points take part and what is held constant."""

    def __init__(self):
        self._constant_camera_ids, self._image_ids = set(), set()
        self._variable_point3D_ids, self._constant_point3D_ids = set(), set()
        self._constant_poses, self._constant_tvecs = set(), {}

    def NumImages(self):
        return len(self._image_ids)

    def NumPoints(self):
        return len(self._variable_point3D_ids) + len(self._constant_point3D_ids)

    def NumConstantCameras(self):
        return len(self._constant_camera_ids)

    def NumConstantPoses(self):
        return len(self._constant_poses)

    def NumConstantTvecs(self):
        return len(self._constant_tvecs)

    def NumVariablePoints(self):
        return len(self._variable_point3D_ids)

    def NumConstantPoints(self):
        return len(self._constant_point3D_ids)

    def NumResiduals(self, reconstruction):
        # bundle_adjustment.cc:109-140: two residuals per observation of the images and of the added points
        n = 0
        for iid in self._image_ids:
            n += sum(1 for l in reconstruction.Image(iid).Lines() if l.HasPoint3D())
        for pid in self._variable_point3D_ids | self._constant_point3D_ids:
            n += sum(1 for (iid, _) in reconstruction.Point3D(pid).track if iid not in self._image_ids)
        return 2 * n

    def AddImage(self, image_id):
        self._image_ids.add(image_id)

    def HasImage(self, image_id):
        return image_id in self._image_ids

    def RemoveImage(self, image_id):
        self._image_ids.discard(image_id)

    def SetConstantCamera(self, camera_id):
        self._constant_camera_ids.add(camera_id)

    def SetVariableCamera(self, camera_id):
        self._constant_camera_ids.discard(camera_id)

    def IsConstantCamera(self, camera_id):
        return camera_id in self._constant_camera_ids

    def SetConstantPose(self, image_id):
        assert self.HasImage(image_id) and not self.HasConstantTvec(image_id)
        self._constant_poses.add(image_id)

    def SetVariablePose(self, image_id):
        self._constant_poses.discard(image_id)

    def HasConstantPose(self, image_id):
        return image_id in self._constant_poses

    def SetConstantTvec(self, image_id, idxs):
        idxs = list(idxs)
        assert 0 < len(idxs) <= 3 and self.HasImage(image_id) and not self.HasConstantPose(image_id)
        assert len(set(idxs)) == len(idxs), "Tvec indices must not contain duplicates"
        self._constant_tvecs[image_id] = idxs

    def RemoveConstantTvec(self, image_id):
        self._constant_tvecs.pop(image_id, None)

    def HasConstantTvec(self, image_id):
        return image_id in self._constant_tvecs

    def ConstantTvec(self, image_id):
        return self._constant_tvecs[image_id]

    def AddVariablePoint(self, pid):
        assert not self.HasConstantPoint(pid)
        self._variable_point3D_ids.add(pid)

    def AddConstantPoint(self, pid):
        assert not self.HasVariablePoint(pid)
        self._constant_point3D_ids.add(pid)

    def HasPoint(self, pid):
        return self.HasVariablePoint(pid) or self.HasConstantPoint(pid)

    def HasVariablePoint(self, pid):
        return pid in self._variable_point3D_ids

    def HasConstantPoint(self, pid):
        return pid in self._constant_point3D_ids

    def RemoveVariablePoint(self, pid):
        self._variable_point3D_ids.discard(pid)

    def RemoveConstantPoint(self, pid):
        self._constant_point3D_ids.discard(pid)

    def Images(self):
        return self._image_ids

    def VariablePoints(self):
        return self._variable_point3D_ids

    def ConstantPoints(self):
        return self._constant_point3D_ids


class BundleAdjuster:
    """bundle_adjustment.h:171-203.  `Solve(reconstruction)` -> bool, `Summary()` afterwards."""

    def __init__(self, options, config, device=0):
        assert options.Check()
        self.options_, self.config_, self.device_ = options, config, device
        self.summary_ = None
        self._used = False
        # options.solver_options.callbacks of the reference (controllers/bundle_adjustment.cc:87-88): one
        # ceres::IterationCallback, fn(BAIterationSummary) -> SOLVER_CONTINUE / SOLVER_ABORT / SOLVER_TERMINATE_SUCCESSFULLY
        self.iteration_callback_ = getattr(options.solver_options, "iteration_callback", None)

    def Summary(self):
        return self.summary_

    def flatten(self, reconstruction):
        """SetUp (bundle_adjustment.cc:326-542) -> flat scene dict + id maps.  Host-only; no GPU needed."""
        opt, cfg = self.options_, self.config_
        pose_index, point_index, cam_index = {}, {}, {}
        lines, obs_pose, obs_point = [], [], []
        pose_const, point_num_obs, camera_ids = {}, {}, []

        def pose_of(iid, const):
            if iid not in pose_index:
                pose_index[iid] = len(pose_index)
            pose_const[iid] = const
            return pose_index[iid]

        def point_of(pid):
            if pid not in point_index:
                point_index[pid] = len(point_index)
            return point_index[pid]

        def cam_of(cid):
            if cid not in cam_index:
                cam_index[cid] = len(cam_index)
            return cam_index[cid]

        # AddImageToProblem (:348-435)
        for iid in sorted(cfg.Images()):
            image = reconstruction.Image(iid)
            image.NormalizeQvec()
            constant_pose = (not opt.refine_extrinsics) or cfg.HasConstantPose(iid)
            nobs = 0
            for line in image.Lines():
                if not line.HasPoint3D():
                    continue
                l = line.Line()
                if abs(np.hypot(l[0], l[1]) - 1.0) > 1e-6:
                    raise ValueError("CHECK_NEAR(line.head<2>().norm(), 1.0, 1e-6) failed")   # :373
                nobs += 1
                pid = line.Point3DId()
                point_num_obs[pid] = point_num_obs.get(pid, 0) + 1
                lines.append(l); obs_pose.append(pose_of(iid, constant_pose)); obs_point.append(point_of(pid))
            if nobs > 0 and image.CameraId() not in camera_ids:
                camera_ids.append(image.CameraId())
        # AddPointToProblem (:437-488) for variable then constant points
        constant_cameras = set(c for c in cfg._constant_camera_ids)
        for pid in sorted(cfg.VariablePoints()) + sorted(cfg.ConstantPoints()):
            point = reconstruction.Point3D(pid)
            if point_num_obs.get(pid, 0) == len(point.track):
                continue
            for (iid, line_idx) in point.track:
                if cfg.HasImage(iid):
                    continue
                point_num_obs[pid] = point_num_obs.get(pid, 0) + 1
                image = reconstruction.Image(iid)
                if image.CameraId() not in camera_ids:
                    camera_ids.append(image.CameraId())
                    constant_cameras.add(image.CameraId())
                lines.append(image.Line(line_idx).Line()); obs_pose.append(pose_of(iid, True)); obs_point.append(point_of(pid))
        if not lines:
            return None
        # ParameterizeCameras (:490-528)
        constant_camera = not (opt.refine_focal_length or opt.refine_principal_point or opt.refine_extra_params)
        for iid in pose_index:
            cam_of(reconstruction.Image(iid).CameraId())
        camera_const_mask = np.zeros(len(cam_index), dtype=np.uint16)
        for cid, k in cam_index.items():
            cam = reconstruction.Camera(cid)
            if constant_camera or cid in constant_cameras:
                camera_const_mask[k] = 0xFFFF
                continue
            idxs = []
            if not opt.refine_focal_length:
                idxs += cam.FocalLengthIdxs()
            if not opt.refine_principal_point:
                idxs += cam.PrincipalPointIdxs()
            if not opt.refine_extra_params:
                idxs += cam.ExtraParamsIdxs()
            camera_const_mask[k] = sum(1 << i for i in idxs)
        # ParameterizePoints (:530-542)
        point_const = np.zeros(len(point_index), dtype=np.uint8)
        for pid, k in point_index.items():
            if len(reconstruction.Point3D(pid).track) > point_num_obs.get(pid, 0) or cfg.HasConstantPoint(pid):
                point_const[k] = 1
        C, P, K = len(pose_index), len(point_index), len(cam_index)
        poses = np.zeros((C, 7)); pose_camera = np.zeros(C, dtype=np.int32)
        pconst = np.zeros(C, dtype=np.uint8); tmask = np.zeros(C, dtype=np.uint8)
        for iid, k in pose_index.items():
            image = reconstruction.Image(iid)
            poses[k, :4], poses[k, 4:] = image.qvec, image.tvec
            pose_camera[k] = cam_index[image.CameraId()]
            pconst[k] = 1 if pose_const[iid] else 0
            if not pose_const[iid] and cfg.HasConstantTvec(iid):
                tmask[k] = sum(1 << i for i in cfg.ConstantTvec(iid))
        points = np.zeros((P, 3))
        for pid, k in point_index.items():
            points[k] = reconstruction.Point3D(pid).xyz
        intr = np.zeros((K, _capi.CAM_STRIDE)); camera_model = np.zeros(K, dtype=np.int32)
        for cid, k in cam_index.items():
            cam = reconstruction.Camera(cid)
            intr[k, : cam.NumParams()] = cam.params
            camera_model[k] = cam.ModelId()
        scene = dict(lines=np.array(lines), obs_pose=np.array(obs_pose, dtype=np.int32), obs_point=np.array(obs_point, dtype=np.int32),
                     pose_camera=pose_camera, camera_model=camera_model, poses=poses, points=points, intr=intr,
                     pose_const=pconst, tvec_const_mask=tmask, point_const=point_const, camera_const_mask=camera_const_mask,
                     loss_type=int(opt.loss_function_type), loss_scale=float(opt.loss_function_scale))
        return scene, pose_index, point_index, cam_index

    def Solve(self, reconstruction):
        assert reconstruction is not None
        assert not self._used, "Cannot use the same BundleAdjuster multiple times"
        self._used = True
        flat = self.flatten(reconstruction)
        if flat is None:            # problem_->NumResiduals() == 0
            return False
        scene, pose_index, point_index, cam_index = flat
        so = self.options_.solver_options
        opts = ba_options(max_num_iterations=so.max_num_iterations, function_tolerance=so.function_tolerance,
                          gradient_tolerance=so.gradient_tolerance, parameter_tolerance=so.parameter_tolerance,
                          max_num_consecutive_invalid_steps=so.max_num_consecutive_invalid_steps,
                          max_linear_solver_iterations=so.max_linear_solver_iterations)
        # bundle_adjustment.cc:273-286: DENSE_SCHUR up to 50 images, SPARSE_SCHUR up to 1000 (both: the device's direct solve, which
        # finds the block sparsity itself), ITERATIVE_SCHUR + SCHUR_JACOBI above - by the number of images IN THE CONFIG
        kMaxNumImagesDirectSparseSolver = 1000
        linear_solver = _capi.LINEAR_SOLVER_DIRECT if self.config_.NumImages() <= kMaxNumImagesDirectSparseSolver else _capi.LINEAR_SOLVER_ITERATIVE_SCHUR
        pb = BAProblem(scene, device=self.device_, linear_solver=linear_solver)
        try:
            try:
                self.summary_ = pb.solve(opts, iteration_callback=self.iteration_callback_)
            except _capi.PPError as e:
                if e.code != _capi.PP_ERR_NUMERIC:
                    raise
                self.summary_ = e.summary       # termination FAILURE, costs and step counts filled (Ceres returns such a Summary too)
            poses, points, intr = pb.get_parameters()
        finally:
            pb.close()
        if self.options_.print_summary and self.summary_ is not None:
            PrintSolverSummary(self.summary_)
        # Solver::Summary::IsSolutionUsable(): after FAILURE / USER_FAILURE Ceres restores the parameter blocks it was given
        # (recollection of ceres/solver.cc, Ceres absent here: unpinned), so nothing is written back
        if self.summary_.termination in (_capi.TERM_FAILURE, _capi.TERM_USER_FAILURE):
            return True
        # parameter memory is updated in place, as Ceres does through the raw pointers
        for iid, k in pose_index.items():
            if not scene["pose_const"][k]:
                reconstruction.Image(iid).qvec = poses[k, :4].copy()
                reconstruction.Image(iid).tvec = poses[k, 4:].copy()
        for pid, k in point_index.items():
            if not scene["point_const"][k]:
                reconstruction.Point3D(pid).xyz = points[k].copy()
        for cid, k in cam_index.items():     # variable camera blocks (refine_* flags, bundle_adjustment.cc:490-528)
            cam = reconstruction.Camera(cid)
            n = cam.NumParams()
            if (int(scene["camera_const_mask"][k]) & ((1 << n) - 1)) != (1 << n) - 1:
                cam.params = intr[k, :n].copy()
        return True


def PrintSolverSummary(s):
    """bundle_adjustment.cc:544-598."""
    term = {0: "CONVERGENCE", 1: "NO_CONVERGENCE", 2: "FAILURE", 3: "USER_SUCCESS", 4: "USER_FAILURE"}[s.termination]
    rows = (("Residuals", s.num_residuals), ("Parameters", s.num_effective_parameters),
            ("Iterations", s.num_successful_steps + s.num_unsuccessful_steps), ("Time", "%g [s]" % s.total_time_s),
            ("Initial cost", "%g [px]" % np.sqrt(s.initial_cost / max(s.num_residuals, 1))),
            ("Final cost", "%g [px]" % np.sqrt(s.final_cost / max(s.num_residuals, 1))), ("Termination", term))
    for k, v in rows:
        print("%16s%s" % (k + " : ", v))
