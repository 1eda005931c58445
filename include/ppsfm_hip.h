/* ppsfm_hip.h — C ABI of the MI355X (gfx950) line-feature bundle adjustment + P6L RANSAC hot path.
 *
 * This is the drop-in boundary.  The reference (colmap/privacy_preserving_sfm) has no
 * C ABI: its extension surface is three compile-time C++ concepts.  Each entry point below replaces
 * the arithmetic behind one of them; the ppsfm/ C++ headers and the privacy_preserving_sfm_amd ctypes module
 * are thin host mirrors of the reference interfaces on top of these functions.
 *
 * Conventions
 *   - POD only; caller owns every host buffer; a handle owns its device memory and one HIP stream.
 *   - every function returns 0 (PP_OK) or a negative error code and never aborts/throws across the
 *     boundary: every entry point is a function-try-block (csrc/common.hpp PP_API_CATCH) that turns
 *     a failed host allocation (bad_alloc / length_error) into PP_ERR_NOMEM and anything else into PP_ERR_INTERNAL;
 *     worker threads of the host builders hand their exceptions to the calling thread.
 *     pp_last_error() gives the message of the calling thread's last failure
 *     (the reference uses glog CHECK aborts: util/logging.h:43-59).
 *   - a handle is not re-entrant (one caller thread at a time); distinct handles may run concurrently.
 *   - all floating point is IEEE binary64, as in the reference.
 *   - quaternions are (w, x, y, z) (base/image.h:219), projection matrices 3x4 ROW-major.
 *   - camera models are identified by the reference's model ids 0..10 (base/camera_models.h:189-349);
 *     every intrinsics block occupies PP_CAM_STRIDE doubles, only the first kNumParams are used.
 */
#ifndef PPSFM_HIP_H_
#define PPSFM_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PP_OK 0
#define PP_ERR_INVALID (-1)    /* bad argument / inconsistent problem description            */
#define PP_ERR_HIP (-2)        /* a HIP runtime call failed (no device, out of memory, ...)  */
#define PP_ERR_NUMERIC (-3)    /* linear system not positive definite / non-finite values     */
#define PP_ERR_NOMEM (-4)      /* a HOST allocation failed (C++ bad_alloc / length_error) inside the library           */
#define PP_ERR_INTERNAL (-5)   /* any other C++ exception stopped at the boundary (a system_error from a worker
                                  thread, an exception thrown by a caller's callback, ...): pp_last_error() has what()  */

#define PP_CAM_STRIDE 12
#define PP_NUM_CAMERA_MODELS 11

const char* pp_last_error(void);
int pp_device_count(int* count);
/* Test hook of the exception containment: raises a C++ exception INSIDE a guarded entry point and returns what the boundary made of
 * it - kind 0 bad_alloc, 1 runtime_error, 2 a non-standard exception, 3 bad_alloc in a worker thread of a host builder,
 * 4 length_error of an over-sized vector, 5 system_error: PP_ERR_NOMEM for 0 / 3 / 4, PP_ERR_INTERNAL otherwise.  Host only. */
int pp_debug_raise(int kind);
/* number of intrinsic parameters of a camera model id (base/camera_models.h:189-349); -1 if unknown */
int pp_camera_num_params(int model_id);
/* pixel threshold -> normalised-plane threshold: BaseCameraModel::ImageToWorldThreshold
 * (base/camera_models.h:533-543), used for RANSACOptions::max_error (sfm/incremental_mapper.cc:673-675) */
int pp_camera_image_to_world_threshold(int model_id, const double* params, double threshold_px, double* out);

/* ======================================================================================== *
 *  Bundle adjustment                                                                        *
 *  replaces: BundleAdjuster::SetUp/Solve  (optim/bundle_adjustment.cc:260-542) and the      *
 *  ceres::Problem / ceres::Solve it drives, with the residual of base/cost_functions.h:46-191 *
 * ======================================================================================== */

enum { PP_LOSS_TRIVIAL = 0, PP_LOSS_SOFT_L1 = 1, PP_LOSS_CAUCHY = 2 }; /* bundle_adjustment.h:51-52 */

/* What BundleAdjuster::SetUp builds from Reconstruction + BundleAdjustmentConfig, flattened:
 * one entry per residual block (= one line observation that has a 3D point).                  */
typedef struct pp_ba_problem_desc {
  int32_t num_poses;    /* C images in the problem (config Images() plus out-of-config observers) */
  int32_t num_points;   /* P 3D points                                                           */
  int32_t num_cameras;  /* K intrinsics blocks                                                    */
  int32_t loss_type;    /* PP_LOSS_*  (BundleAdjustmentOptions::CreateLossFunction, .cc:55-70)    */
  int64_t num_obs;      /* M residual blocks                                                      */
  double loss_scale;
  const double* lines;          /* M x 3  FeatureLine::Line(): (a,b,c), a^2+b^2 = 1 (.cc:373)     */
  const int32_t* obs_pose;      /* M      image (pose) index of each observation                  */
  const int32_t* obs_point;     /* M      3D point index                                          */
  const int32_t* pose_camera;   /* C      Image::CameraId                                          */
  const int32_t* camera_model;  /* K      Camera::ModelId                                          */
  const uint8_t* pose_const;        /* C  1: constant pose -> ConstantPose functor (.cc:361-398, :470-486) */
  const uint8_t* tvec_const_mask;   /* C  bit i: tvec[i] constant, SubsetParameterization (.cc:426-432)   */
  const uint8_t* point_const;       /* P  1: SetParameterBlockConstant (.cc:530-542)                       */
  const uint16_t* camera_const_mask;/* K  bit i: intrinsic i constant; all bits of the model set =>
                                          block constant (.cc:490-528).  NULL => all constant.             */
  /* ceres::Solver::Options::linear_solver_type as BundleAdjuster::Solve picks it from the image count before it builds the
   * problem (.cc:273-286): PP_LINEAR_SOLVER_*.  AUTO applies the reference's rule to num_poses (> 1000 images:
   * ITERATIVE_SCHUR + SCHUR_JACOBI); the host mirrors pass the choice made from BundleAdjustmentConfig::NumImages().
   * The structure built at create depends on it (an iterative handle builds no pair lists and no N x N system).
   * NOTE (zero-initialised descriptors): AUTO is 0, so a caller with more than 1000 poses gets the iterative solver without asking -
   * inexact steps (eta), and pp_ba_reduced_system refuses such a handle; PP_LINEAR_SOLVER_DIRECT (or the environment override
   * PPSFM_BA_LINEAR_SOLVER=direct) requests the direct solve.  VARIABLE intrinsics (camera_const_mask) ride along on the iterative path: their
   * columns follow the pose columns in the conjugate-gradient vectors and get one preconditioner block per intrinsics block, as Ceres lays the
   * parameter blocks out; in a point-sharded group their rows ride in the same exchanges (6 C + NI doubles per product, the compact diagonal blocks and
   * right-hand-side rows once per LM iteration). */
  int32_t linear_solver;
  /* Order of the images' columns in the reduced camera system: PP_ORDERING_*.
   * DEFAULT (0, what a zero-initialised descriptor gets) and NATURAL (1): the caller's order.
   * AUTO (2): pp_ba_create may renumber the images INTERNALLY - what Ceres' SPARSE_SCHUR ordering does for the reference between 50 and 1000 images
   * (bundle_adjustment.cc:279-282): reverse Cuthill-McKee on the co-visibility graph when that removes at least a tenth of the factor's non-zero 64x64
   * tiles (a sequence scene whose image ids are not in capture order gets its block-banded system back), and a nested dissection - of that band, or of the
   * graph itself (clusters joined by a few images) - when the independent parts, factorised side by side, shorten the critical path of the factorisation
   * (pp_ba_get_structure: info[6], info[7]).  The solve is the same to rounding, not bit for bit, as in the caller's order.  Every per-image array of
   * this interface stays in the CALLER's order.  The host mirrors of BundleAdjuster (ppsfm/ppsfm.hpp, bundle_adjustment.py) pass AUTO, as the reference's
   * solver orders without being asked.
   * Point-sharded groups (pp_ba_set_allreduce / pp_ba_set_communicator): every rank must lay out the exchanged system the same way, and a rank only sees its
   * own shard's co-visibility.  A handle of a group therefore keeps the caller's order - or, with AUTO, takes its order and its tile structure from
   * `covisibility` below, which every rank of the group passes alike (the UNION over the shards).  The attach calls refuse a handle that renumbered its
   * images from its own shard (an error on every rank that does; make the request the same on all ranks). */
  int32_t ordering;
  /* Optional (NULL: derived from this descriptor's own observations): C x C bytes, row-major, non-zero where two images share a variable point (symmetric;
   * the diagonal is ignored).  Read by PP_ORDERING_AUTO for the order and by pp_ba_create for the tile structure of the reduced system: the ranks of a
   * point-sharded group pass the element-wise MAX of their pp_ba_covisibility matrices (one all-reduce of C x C bytes at create), which lets the group keep
   * the block-sparse, several-chain factorisation a whole sequence scene gets on one GPU. */
  const uint8_t* covisibility;
} pp_ba_problem_desc;
enum { PP_LINEAR_SOLVER_AUTO = 0, PP_LINEAR_SOLVER_DIRECT = 1, PP_LINEAR_SOLVER_ITERATIVE_SCHUR = 2 };
enum { PP_ORDERING_DEFAULT = 0, PP_ORDERING_NATURAL = 1, PP_ORDERING_AUTO = 2 };
enum { PP_MAX_NUM_IMAGES_DIRECT_SOLVER = 1000 };   /* kMaxNumImagesDirectSparseSolver, bundle_adjustment.cc:276 */

/* the fields of ceres::IterationSummary the LM loop has */
typedef struct pp_ba_iteration_summary {
  int32_t iteration;           /* 0 = initial evaluation */
  int32_t step_is_successful;  /* 1 for iteration 0 */
  double cost, cost_change, gradient_max_norm, step_norm, relative_decrease, trust_region_radius;
} pp_ba_iteration_summary;
/* return values of the callback = ceres::CallbackReturnType */
enum { PP_SOLVER_CONTINUE = 0, PP_SOLVER_ABORT = 1, PP_SOLVER_TERMINATE_SUCCESSFULLY = 2 };

/* ceres::Solver::Options fields the reference sets (bundle_adjustment.h:80-93,
 * controllers/incremental_mapper.cc:196-243) + the Ceres trust-region defaults it inherits. */
typedef struct pp_ba_options {
  int32_t max_num_iterations;                /* 100 */
  int32_t max_num_consecutive_invalid_steps; /* 10  */
  double function_tolerance;                 /* 0   */
  double gradient_tolerance;                 /* 0 (1.0 global BA, 10.0 local BA) */
  double parameter_tolerance;                /* 0   */
  double initial_trust_region_radius;        /* 1e4   */
  double max_trust_region_radius;            /* 1e16  */
  double min_trust_region_radius;            /* 1e-32 */
  double min_relative_decrease;              /* 1e-3  */
  double min_lm_diagonal;                    /* 1e-6  */
  double max_lm_diagonal;                    /* 1e32  */
  int32_t jacobi_scaling;                    /* 1 */
  int32_t phase_timings;                     /* 0; 1 = record HIP events between the phases of every iteration for
                                                pp_ba_get_timings (each record costs ~5 us of stream time) */
  int32_t max_linear_solver_iterations;      /* 200 (bundle_adjustment.h:87): cap of the conjugate-gradient loop of an iterative handle */
  int32_t reserved_;
  double eta;                                /* 1e-1 (Ceres default): forcing term of the inexact step = q tolerance of the CG loop */
  /* ceres::IterationCallback (Solver::Options::callbacks).  The reference registers one callback,
   * BundleAdjustmentIterationCallback (controllers/bundle_adjustment.cc:43-61, 87-88), which blocks while the
   * controller thread is paused and returns SOLVER_TERMINATE_SUCCESSFULLY once it was stopped.  Called on the caller's thread after every
   * iteration (iteration 0 = the initial evaluation) with the iteration's summary; with a callback set
   * the solver waits for the evaluation at an accepted point before calling (one extra host round trip per
   * iteration), so `cost` / `gradient_max_norm` are the values at the point just accepted.  NULL = none. */
  int32_t (*iteration_callback)(void* ctx, const pp_ba_iteration_summary* it);
  void* iteration_callback_ctx;
} pp_ba_options;
void pp_ba_options_default(pp_ba_options* o);

/* ceres::TerminationType; USER_FAILURE (callback returned PP_SOLVER_ABORT) is NOT a usable solution in Ceres
 * (Solver::Summary::IsSolutionUsable): the host mirrors do not copy the parameters back in that case. */
enum { PP_TERM_CONVERGENCE = 0, PP_TERM_NO_CONVERGENCE = 1, PP_TERM_FAILURE = 2, PP_TERM_USER_SUCCESS = 3, PP_TERM_USER_FAILURE = 4 };

/* the part of ceres::Solver::Summary the reference prints (bundle_adjustment.cc:544-598) */
typedef struct pp_ba_summary {
  double initial_cost, final_cost;
  int32_t num_successful_steps, num_unsuccessful_steps;
  int32_t termination;        /* PP_TERM_* */
  int32_t num_iterations;     /* successful + unsuccessful */
  int32_t num_residuals;      /* 2 M */
  int32_t num_effective_parameters;
  double total_time_s;        /* wall clock of pp_ba_solve, host side */
  double device_time_s;       /* HIP-event time of the LM loop */
  /* what solved the reduced camera system in the LAST iteration of this solve (Ceres prints linear_solver_type_used,
   * bundle_adjustment.cc:585-590): PP_LINSOLVE_* */
  int32_t linear_solver;
  /* one-launch factorisations of this HANDLE (all its solves so far) that ran into a bounded wait and were repeated with
   * per-column launches; after the first one the handle stays with per-column launches */
  int32_t cholesky_fallbacks;
  int32_t linear_solver_iterations;   /* PP_LINSOLVE_PCG: conjugate-gradient iterations summed over the LM iterations; 0 otherwise */
  int32_t reserved_;
} pp_ba_summary;
enum { PP_LINSOLVE_CHOLESKY_COLUMNS = 0,   /* dense Cholesky, one launch per block column */
       PP_LINSOLVE_CHOLESKY_TASKS = 1,     /* dense Cholesky, the whole factorisation in one launch */
       PP_LINSOLVE_CHOLESKY_SPARSE = 2,    /* block-sparse Cholesky: the one-launch factorisation over the non-zero tiles, a chain workgroup per independent
                                              sub-tree of the elimination tree (per-column launches over the tile lists above 128 block columns / as the fallback) */
       PP_LINSOLVE_PCG = 3 };              /* matrix-free conjugate gradients on the implicit Schur complement (ITERATIVE_SCHUR + SCHUR_JACOBI) */

typedef struct pp_ba_impl* pp_ba_handle;

/* Upload the static structure (observations, index lists, masks).  `device` is the HIP device
 * ordinal.  One handle per sub-model / per GPU.                                                    */
int pp_ba_create(const pp_ba_problem_desc* desc, int device, pp_ba_handle* out);
int pp_ba_destroy(pp_ba_handle h);
/* pp_ba_create / pp_ba_destroy sit in the mapper's inner loop (a new BundleAdjuster per registered image, sfm/incremental_mapper.cc:813-858): the
 * device blocks, pinned blocks, stream and events of a destroyed handle are kept (by size class, at most PPSFM_POOL_MAX_MB = 1024 MB of device memory
 * per process; 0 disables) and handed to the next one.  pp_pool_trim frees everything that is cached. */
int pp_pool_trim(void);

/* parameter blocks: poses C x 7 (qw,qx,qy,qz,tx,ty,tz), points P x 3, intrinsics K x PP_CAM_STRIDE.
 * PRECONDITION: unit quaternions.  BundleAdjuster::AddImageToProblem normalises every image of the configuration before it hands the block to Ceres
 * ("CostFunction assumes unit quaternions", src/optim/bundle_adjustment.cc:354-355); a binding does the same on its side of the boundary (the host
 * mirrors do: BundleAdjuster.AddImageToProblem).  The residual is the rotate-point polynomial of q as given (no re-normalisation, as Ceres'
 * UnitQuaternionRotatePoint); the Jacobian on the rotation tangent is exact for unit q only - for a quaternion of length L it differs from Ceres'
 * jets by factors of L (tools/fuzz_line_eval.py measures this; the ambient 2x4 Jacobian of pp_ba_evaluate is exact for any q).  pp_ba_set_parameters
 * therefore returns PP_ERR_INVALID for a VARIABLE pose whose |q|^2 is off 1 by more than 1e-6; a constant pose is taken as given (it enters through the
 * polynomial only, and AddPointToProblem does not normalise the out-of-configuration observers either).  pp_ba_attach (device pointers) cannot look. */
int pp_ba_set_parameters(pp_ba_handle h, const double* poses, const double* points, const double* intr);
int pp_ba_get_parameters(pp_ba_handle h, double* poses, double* points, double* intr);

/* Batched ceres::CostFunction::Evaluate for all M residual blocks at the parameters currently on the
 * device (kernel K1).  Results stay on the device; any of the *_out host pointers may be NULL.
 *   jac_mode 0: J_pose is M x (2x6) = [d r / d rotation-tangent (3) | d r / d tvec (3)]   (local
 *               parameterisation already applied: J_q (2x4) * QuaternionParameterization plus-Jacobian)
 *   jac_mode 1: J_pose is M x (2x7) = [d r / d qvec (w,x,y,z) | d r / d tvec]  (what Ceres hands to
 *               Evaluate: jacobians[0] 2x4 and jacobians[1] 2x3 side by side)
 *   J_point M x (2x3), J_cam M x (2 x PP_CAM_STRIDE) (only if want_cam), residuals 2M (NOT loss-corrected),
 *   cost = 1/2 sum rho(|r|^2).
 */
int pp_ba_eval(pp_ba_handle h, int jac_mode, int want_cam, double* residuals_out, double* jpose_out,
               double* jpoint_out, double* jcam_out, double* cost_out);

/* The same evaluation for a ceres::EvaluationCallback adaptor (ppsfm/ceres_adaptor.hpp): the results are copied by DMA into
 * PINNED host mirrors owned by the handle and the caller gets pointers to them (valid until the next call on the handle);
 * each residual block's CostFunction::Evaluate then copies its own slice - no staging copy through pageable memory, no
 * per-call allocation.  want_jacobians = 0 (Ceres evaluates residuals alone at every trial point) runs the cost-only
 * variant of K1 and moves 16 B per observation instead of 236 B+.  Layouts as pp_ba_eval.                          */
int pp_ba_eval_host_view(pp_ba_handle h, int jac_mode, int want_cam, int want_jacobians, const double** residuals,
                         const double** jpose, const double** jpoint, const double** jcam, double* cost_out);

/* device-resident variant used by benchmarks:
 * runs K1 `repeat` times on the handle's stream without copying anything back; returns the
 * HIP-event time per launch in *ms_per_launch (may be NULL).                                  */
int pp_ba_eval_device(pp_ba_handle h, int jac_mode, int want_cam, int repeat, float* ms_per_launch);

/* ceres::Solve replacement: Levenberg-Marquardt with exact point-Schur elimination, everything on the
 * device; parameters are read from / left on the device (use set/get_parameters).                  */
int pp_ba_solve(pp_ba_handle h, const pp_ba_options* options, pp_ba_summary* summary);

/* per-iteration trace of the last pp_ba_solve: rows of 7 doubles
 * {cost, cost_change, gradient_max_norm, step_norm, relative_decrease, trust_region_radius, successful} */
int pp_ba_get_trace(pp_ba_handle h, double* trace, int32_t capacity_rows, int32_t* num_rows);
/* Structure of the reduced camera system as the handle factorises it: info[0] = 64x64 tiles in the lower triangle, info[1] = non-zero
 * tiles of the factor (fill-in included) in the caller's image order, info[2] = in the order the handle uses, info[3] = 1 if the images
 * were renumbered internally (pp_ba_problem_desc::ordering), info[4] = 1 if the block-sparse path is taken (outside a group),
 * info[5] = 1 for an iterative handle (no reduced system is formed), info[6] = chain workgroups of the one-launch factorisation (more than one: the
 * internal order is a nested dissection whose independent parts are factorised side by side), info[7] = block-column steps on its longest
 * dependency path (info[0]'s block-column count for one chain). */
int pp_ba_get_structure(pp_ba_handle h, int32_t* info /* 8 */);
/* The image order pp_ba_create would give this problem's reduced camera system, computed on the host alone (no device is touched: what the ordering tests
 * run without a GPU).  Only the structure fields of the descriptor are read (counts, obs_pose, obs_point, pose_camera, camera_model, the const masks,
 * linear_solver, ordering).  old_of_new (num_poses ints, may be NULL): the caller's index of the image at every internal position.  info[0] = 1 if the
 * images are renumbered, info[1] = non-zero tiles of the factor in the caller's order (-1: no candidate order was looked at, or the co-visibility turned out too dense for any
 * order to pay before it was complete), info[2] = in the order chosen,
 * info[3] / info[4] = chain workgroups / chain steps of its one-launch factorisation, info[5] = block columns, info[6] = 1 if the block-sparse path applies,
 * info[7] = variable intrinsics columns. */
int pp_ba_plan_ordering(const pp_ba_problem_desc* d, int32_t* old_of_new /* num_poses or NULL */, int32_t* info /* 8 */);
/* The Schur pair lists of the problem in the CALLER's image order, built by the library's HOST builder (the builder of small problems and the one the
 * device-built lists are tested against), host only: for every pair of variable images (ci >= cj) that share a variable point the (observation of ci,
 * observation of cj) pairs - pair_start (lists + 1 offsets), pair_ij (2 per list), pair_entries (2 per entry); NULL arrays are skipped, nothing is
 * written beyond the capacities.  threads: 0 = by size, otherwise that many host threads (what the sanitizer configuration runs: tests/test_host_sanitizers.py). */
int pp_ba_pair_lists_host(const pp_ba_problem_desc* d, int32_t threads, int64_t* num_lists, int64_t* num_entries, int32_t* pair_start, int32_t* pair_ij,
                          int32_t* pair_entries, int64_t capacity_lists, int64_t capacity_entries);
/* The co-visibility matrix of a descriptor's own observations (host only): out = C x C bytes, 1 where two variable images share a variable point.
 * A point-sharded group all-reduces (MAX) these at create and passes the result as pp_ba_problem_desc::covisibility. */
int pp_ba_covisibility(const pp_ba_problem_desc* d, uint8_t* out /* num_poses x num_poses */);
/* Host wall time of the pp_ba_create that made this handle, ms: [0] image ordering (co-visibility, candidate orders, their chain plans), [1] CSR and
 * Schur pair lists, [2] tile structure + the intrinsics lists, [3] device allocation + upload, [4] the factorisation's task plan (made with the solver
 * buffers, at the handle's first solve or attach: 0 before; a process-wide cache keyed on the tile map answers repeated structures), [5] total of [0]..[3].  The mapper builds a new BundleAdjuster per global
 * bundle adjustment (src/sfm/incremental_mapper.cc:893-936): this is what that costs here. */
int pp_ba_get_create_profile(pp_ba_handle h, double* ms /* 6 */);

/* The damped Jacobi-scaled reduced camera system at the current parameters for a given radius, as the
 * solver builds it (kernels K2/K3a): S is n x n row-major (n = 6 C + variable intrinsics, constant
 * columns are identity rows), rhs n.  For parity tests and for multi-GPU reduction. */
int pp_ba_reduced_system(pp_ba_handle h, const pp_ba_options* options, double radius, int32_t* n, double* S,
                         double* rhs, int64_t capacity);

/* Multi-GPU hook: one BA whose POINTS (with their observations) are sharded across the
 * ranks of a group; poses/intrinsics are replicated.  Each rank creates its handle from its own shard
 * of the observations (same pose/point/camera index spaces, points it does not own simply have no
 * observation) and registers a reduction callback: `fn(ctx, device_ptr, count, op)` reduces `count`
 * doubles in place across the group (op PP_REDUCE_SUM / PP_REDUCE_MAX) on the handle's stream
 * semantics (the solver synchronises its stream before calling and expects the result to be visible
 * to later work on any stream when fn returns).  Per LM iteration the solver reduces: the per-pose
 * normal-equation blocks (42 doubles per pose), the reduced camera system S (lower triangle is what
 * matters) + rhs, and a handful of scalars.  rank 0 of the group adds the damping/diagonal terms.
 * NULL fn => single GPU.
 * ATTACHING IS COLLECTIVE (fn != NULL, group_size > 1): the call itself runs ONE reduction through fn - three doubles,
 * PP_REDUCE_MAX: whether any rank must refuse (a handle that renumbered its images from its own shard) and a hash of the
 * layout of the reduced system (image count, variable intrinsics, internal order, block-sparse tile map), so that the ranks
 * of a group fail TOGETHER (PP_ERR_INVALID everywhere) instead of one erroring while the others enter a collective, and a
 * group whose handles would exchange differently sized systems is refused before its first exchange.  Hence: every rank of
 * the group must make the call, each from its own thread / process (attaching the group's handles one after another from ONE
 * thread deadlocks in the callback's rendezvous), and fn must be ready for traffic at attach time, not only inside
 * pp_ba_solve.  A group of one rank and a detach (fn == NULL) exchange nothing.                                          */
enum { PP_REDUCE_SUM = 0, PP_REDUCE_MAX = 1 };
typedef int (*pp_allreduce_fn)(void* ctx, void* device_ptr, int64_t count, int32_t op);
int pp_ba_set_allreduce(pp_ba_handle h, pp_allreduce_fn fn, void* ctx, int32_t group_rank, int32_t group_size);

/* The same exchange as RCCL collectives inside the library (one process per GPU).  The ranks of ONE sub-model's group create a communicator from a shared 128-byte id
 * (rank 0 of the group calls pp_comm_unique_id and sends it to the others by any means: MPI, a socket, a file), then
 * every rank hands its communicator to its handle.  Per LM iteration the solver then issues on the handle's own stream,
 * without synchronising the host: ONE grouped all-reduce of the per-pose blocks U, g_c (+ intrinsics sums), ONE all-reduce of
 * the reduced system packed as its lower triangle + rhs row ((n+1)(n+2)/2 doubles: 36 MB at 500 images instead of the
 * 72 MB square), and ONE grouped all-reduce of the handful of scalars (cost, model cost change, |step|^2, |x|^2 as sums with
 * the replicated pose part counted on rank 0 only; gradient max norm as a max).  Every rank then factorises the (identical)
 * reduced system itself.  BASELINE configs[4]: 4 sub-models over 8 GPUs = 4 communicators of 2 ranks (bench.py --submodels 4).
 * librccl is loaded with dlopen on first use: pp_comm_* return PP_ERR_HIP where it is absent.                        */
#define PP_COMM_ID_BYTES 128
typedef struct pp_comm_impl* pp_comm_handle;
int pp_comm_unique_id(uint8_t* id /* PP_COMM_ID_BYTES */);
int pp_comm_create(const uint8_t* id, int32_t num_ranks, int32_t rank, int device, pp_comm_handle* out);
int pp_comm_destroy(pp_comm_handle c);
/* in-place all-reduce of `count` device doubles on the null stream, synchronous (set-up and tests; the solver uses its own stream) */
int pp_comm_allreduce(pp_comm_handle c, double* device_ptr, int64_t count, int32_t op);
/* NULL detaches.  The communicator must outlive the handle's solves; it replaces a pp_ba_set_allreduce callback.
 * COLLECTIVE like pp_ba_set_allreduce when comm has more than one rank: every rank of the communicator calls it (one
 * ncclAllReduce of three doubles on the null stream inside the call: the common refusal + the structure hash).        */
int pp_ba_set_communicator(pp_ba_handle h, pp_comm_handle comm);

/* The dense SPD solver of the reduced camera system on its own (kernel K3b: blocked fp64 Cholesky on
 * v_mfma_f64_16x16x4_f64 + triangular solves): solves A x = b for a symmetric positive definite n x n
 * row-major A (only the lower triangle is read).  repeat > 1 re-runs the device part for timing;
 * *ms_per_solve (may be NULL) receives the HIP-event time of one factorisation + solve: the MEDIAN
 * over the `repeat` solves (an untimed one precedes them).                                        */
int pp_dense_cholesky_solve(int32_t n, const double* A, const double* b, double* x, int device, int32_t repeat,
                            float* ms_per_solve);
/* The work list of the one-launch factorisation for `block_columns` 64-wide block columns (4 .. 128), in launch order: four
 * int32 per task - type (1 PrepX, 2 PrepD, 3 solve, 4 update), step k, a, b (solve: a = block row; update: a = I,
 * b = J | part << 8 | parts << 12 | target << 16 of super-tile (I,J)).  Host-only (no device needed): lets a test check that
 * the order is topological, i.e. that no workgroup waits for one dispatched after it.  *count receives the number of tasks;
 * tasks beyond `capacity` are not written.                                                                                  */
int pp_cholesky_task_list(int32_t block_columns, int32_t* tasks, int64_t capacity, int64_t* count);
/* The same for a BLOCK-SPARSE system: tile_nz = T x T bytes, non-zero 64x64 tiles of the lower triangle of the matrix (closed under the
 * fill-in of the factorisation here).  map_out (T x T bytes, may be NULL): the tile map the one-launch mode works with (fill-in + the
 * two sub-diagonals the chain owns).  tasks: rows of SEVEN ints {type, k, a, b, w0, w1, w2} - w0 / w1: the values the task's
 * panel counters must have reached, w2: the value of the row counter of the row it solves a tile of (which depend on the tasks that
 * exist; tests/test_cholesky_task_order.py replays them). */
int pp_cholesky_task_list_sparse(int32_t block_columns, const uint8_t* tile_nz, uint8_t* map_out, int32_t* tasks, int64_t capacity, int64_t* count);
/* The plan of the one-launch factorisation of a block-sparse system with SEVERAL CHAINS (a nested-dissection order: the independent sub-trees of the
 * elimination tree are factorised side by side, cholesky.hip "ChainRanges").  max_chains <= 0: as many as the structure has (at most 16).
 * tasks: rows of SIXTEEN ints {type, k, a, b, w0, w1, w2, flags, cidx, sidx, zsel, mask, slot[4]}; chains_out (49 ints, may be NULL): n, then {begin, end, post} per chain;
 * time_out / rho1_out (T ints each, may be NULL): the step at which a block column is eliminated and 1 + its rank in the elimination order;
 * *verified (may be NULL): 1 when the host replay finds every wait of the list met by an earlier task (what pp_ba_solve requires of a list of
 * several chains before it uses it).  tests/test_cholesky_task_order.py replays the arithmetic side. */
int pp_cholesky_task_plan(int32_t block_columns, const uint8_t* tile_nz, int32_t max_chains, uint8_t* map_out, int32_t* tasks, int64_t capacity,
                          int64_t* count, int32_t* chains_out, int32_t* time_out, int32_t* rho1_out, int32_t* verified);

/* timing breakdown of the last solve (HIP events, ms, averaged per call): index by PP_BA_T_* */
enum { PP_BA_T_EVAL = 0, PP_BA_T_REDUCE = 1, PP_BA_T_SCHUR = 2, PP_BA_T_CHOLESKY = 3, PP_BA_T_BACKSUB = 4,
       PP_BA_T_UPDATE_COST = 5, PP_BA_T_COUNT = 6 };
int pp_ba_get_timings(pp_ba_handle h, double* ms /* PP_BA_T_COUNT */, int32_t* calls /* PP_BA_T_COUNT */);

/* ---- filters the mapper runs after every bundle adjustment (on the parameters the handle currently holds) ----
 * replaces Reconstruction::FilterPoints3D (= FilterPoints3DWithLargeReprojectionError +
 * FilterPoints3DWithSmallTriangulationAngle, base/reconstruction.cc:425-439, 594-719) and
 * FilterObservationsWithNegativeDepth (:441-460); the track of a point = its observations in problem order.
 * Deletions are returned as masks (the reference calls DeletePoint3D / DeleteObservation):
 *   obs_deleted[o] = 1: observation o is removed (its point was deleted or its pixel line error > max_reproj_error,
 *   behind the camera or outside the image, base/projection.cc:153-203); point_deleted[p] = 1; point_error[p] = mean
 *   error of the surviving track (Point3D::SetError), -1 where not set.
 * obs_aligned M (FeatureLine::IsAligned; NULL = none aligned), cam_size K x 2 (width, height), point_subset P or NULL. */
typedef struct { double max_reproj_error; double min_tri_angle_deg; } pp_filter_options;
typedef struct { int64_t num_filtered; /* the reference's return value */ int64_t num_points_deleted, num_observations_deleted; } pp_filter_report;
int pp_ba_filter_points(pp_ba_handle h, const pp_filter_options* options, const uint8_t* obs_aligned, const int32_t* cam_size,
                        const uint8_t* point_subset, uint8_t* obs_deleted, uint8_t* point_deleted, double* point_error, pp_filter_report* report);
int pp_ba_filter_negative_depth(pp_ba_handle h, uint8_t* obs_negative /* M */, int64_t* num_filtered);

/* ======================================================================================== *
 *  Absolute pose from six 2D-line / 3D-point pairs, RANSAC                                  *
 *  replaces: RANSAC<P6LEstimator>::Estimate (optim/ransac.h:178-278) with                   *
 *  P6LEstimator::Estimate/Residuals (estimators/absolute_pose.cc:79-174), re3q3              *
 *  (lib/re3q3/re3q3/re3q3.h:16-200), ComputeSquaredLineReprojectionError                     *
 *  (estimators/utils.cc:40-89), InlierSupportMeasurer (optim/support_measurement.cc:36-60). *
 * ======================================================================================== */

typedef struct pp_pose_impl* pp_pose_handle;

/* X = lines2D (n x 3, FeatureLine::Line()), Y = points3D (n x 3), aligned (n, FeatureLine::IsAligned(),
 * may be NULL) are uploaded once and stay resident (the Estimator's X/Y of optim/ransac.h:178-180).   */
int pp_pose_create(int32_t n, const double* lines2D, const double* points3D, const uint8_t* aligned,
                   int device, pp_pose_handle* out);
int pp_pose_destroy(pp_pose_handle h);

/* P6LEstimator::Residuals for `num_models` 3x4 row-major models: residuals_out num_models x n
 * (squared normalised point-to-line distance, DBL_MAX behind the camera).  Bit-identical to the
 * reference arithmetic (no FMA contraction, IEEE division).                                       */
int pp_pose_residuals(pp_pose_handle h, int32_t num_models, const double* models, double* residuals_out);

/* InlierSupportMeasurer::Evaluate for a batch of models (kernel K4, one wavefront per model):
 * num_inliers[m] = #{r <= max_residual} (exact); residual_sum[m] = sum of inlier residuals in a
 * fixed tree order (deterministic; NOT the sequential order — see pp_pose_support_sequential).       */
int pp_pose_score(pp_pose_handle h, int32_t num_models, const double* models, double max_residual,
                  uint32_t* num_inliers, double* residual_sum);
/* the exact sequential-order residual_sum of support_measurement.cc:40-47 (one lane per model) */
int pp_pose_support_sequential(pp_pose_handle h, int32_t num_models, const double* models, double max_residual,
                               uint32_t* num_inliers, double* residual_sum);

/* P6LEstimator::Estimate for H six-tuples (kernel K5, one lane per hypothesis):
 * samples H x 6 indices into X/Y; models_out H x 8 x 12 (3x4 row-major), num_models_out H.
 * Roots are returned in ascending order of the eliminated variable.                               */
int pp_pose_p6l_batch(pp_pose_handle h, int64_t num_hyp, const uint32_t* samples, double* models_out,
                      int32_t* num_models_out);
/* stand-alone three-quadrics solver, one system per lane: coeffs H x 30 (3x10 row-major),
 * solutions H x 24 (3x8 row-major), num H                                                          */
int pp_re3q3_batch(int64_t num, const double* coeffs, double* solutions, int32_t* num_solutions, int device);

/* RANSACOptions (optim/ransac.h:47-76) */
typedef struct pp_ransac_options {
  double max_error;                  /* threshold on the UNSQUARED error; residuals are squared */
  double min_inlier_ratio;           /* 0.1  */
  double confidence;                 /* 0.99 */
  double dyn_num_trials_multiplier;  /* 3.0  */
  uint64_t min_num_trials;           /* 0    */
  uint64_t max_num_trials;           /* SIZE_MAX */
  uint32_t seed;                     /* PRNG seed: util/random.h:46 kDefaultPRNGSeed = 0 */
  uint32_t chunk_trials;             /* speculation width (trials solved+scored per launch), 0 = auto */
} pp_ransac_options;
void pp_ransac_options_default(pp_ransac_options* o);

/* RANSAC::Report (optim/ransac.h:82-99) */
typedef struct pp_ransac_report {
  int32_t success;
  int32_t best_model_index;   /* index of the winner among its trial's models (trace, not in the reference) */
  uint64_t num_trials;
  uint64_t num_inliers;       /* support.num_inliers */
  double residual_sum;        /* support.residual_sum, sequential order */
  double model[12];           /* 3x4 row-major */
  int64_t best_trial;         /* trial that produced the winner (trace) */
  uint64_t hypotheses_evaluated;  /* trials actually solved+scored on the device (>= num_trials: speculation) */
  uint64_t models_scored;
  double device_time_s, total_time_s;
} pp_ransac_report;

/* RANSAC<P6LEstimator, InlierSupportMeasurer, RandomSampler>::Estimate: the host draws the samples
 * (mt19937 + persistent partial Fisher-Yates, bit-compatible with optim/random_sampler.cc:43-62 on
 * libstdc++), the device solves and scores them in speculative chunks, the host replays the
 * sequential accept / adaptive-termination logic in trial order (optim/ransac.h:213-249), so that
 * num_trials, the winner and the inlier mask equal the sequential loop's.  inlier_mask: n bytes. */
int pp_pose_ransac(pp_pose_handle h, const pp_ransac_options* options, pp_ransac_report* report,
                   uint8_t* inlier_mask);

/* throughput form (BASELINE cfg 4): solve + score `num_hyp` pre-drawn six-tuples, keep only the best
 * (num_inliers, tree residual_sum) — no adaptive termination.  samples may be NULL: drawn on the host
 * with the RANSAC sampler from `seed`.                                                             */
int pp_pose_hypotheses(pp_pose_handle h, int64_t num_hyp, const uint32_t* samples, uint32_t seed,
                       double max_residual, pp_ransac_report* report);

/* scores of EVERY model of the last pp_pose_hypotheses call (they stay on the device; this copies them out):
 * num_models num_hyp, num_inliers / residual_sum num_hyp x 8 (entries >= num_models[h] are unspecified).  For tests
 * that re-derive the winner on the host at BASELINE cfg 4's full size; any pointer may be NULL.                */
int pp_pose_last_scores(pp_pose_handle h, int64_t num_hyp, int32_t* num_models, uint32_t* num_inliers, double* residual_sum);

/* the RandomSampler stream alone (host): first `count` k-subsets for total size n */
int pp_sampler_draw(uint32_t seed, uint32_t n, int32_t k, int64_t count, uint32_t* out);
/* RANSAC::ComputeNumTrials for kMinNumSamples = 6 (optim/ransac.h:158-176) */
uint64_t pp_ransac_compute_num_trials(uint64_t num_inliers, uint64_t num_samples, double confidence,
                                      double num_trials_multiplier);


/* ---- batched robust track triangulation (one LORANSAC per track, one GPU lane per track) --------------------------
 * replaces the per-track EstimateTriangulation calls of the incremental triangulator (estimators/triangulation.cc:111-149,
 * sfm/incremental_triangulator.cc:214, 536): LORANSAC<TriangulationEstimator, ..., CombinationSampler> (optim/loransac.h,
 * deterministic: the sampler enumerates the 3-combinations), TriangulateMultiViewPoint (base/triangulation.cc:41-57), cheirality
 * and minimum-triangulation-angle checks, squared angular (residual_type 0) or squared pixel (1) line residuals.
 * Track t = observations track_start[t] .. track_start[t+1]-1: line (a,b,c) seen in view obs_view[i].  Views: proj_matrices
 * V x (3x4 row-major), proj_centers V x 3, view_camera V; cameras: model id, intr K x PP_CAM_STRIDE, cam_size K x 2.
 * Out per track: success, xyz (3), num_trials; per observation: inlier mask (0 where the track failed).               */
typedef struct pp_triangulation_options {
  double min_tri_angle;      /* radians */
  int32_t residual_type;     /* 0 ANGULAR_ERROR (EstimateTriangulationOptions default), 1 REPROJECTION_ERROR */
  int32_t reserved;
  pp_ransac_options ransac;  /* max_error: radians resp. pixels (squared internally) */
} pp_triangulation_options;
int pp_triangulate_tracks(int device, int32_t num_tracks, const int32_t* track_start, const double* lines, const int32_t* obs_view,
                          int32_t num_views, const double* proj_matrices, const double* proj_centers, const int32_t* view_camera,
                          int32_t num_cameras, const int32_t* camera_model, const double* intr, const int32_t* cam_size,
                          const pp_triangulation_options* options, uint8_t* success, double* xyz, uint8_t* inlier_mask,
                          int32_t* num_trials, float* device_ms /* may be NULL */);

/* ======================================================================================== *
 *  Four-view line initialisation (LO-MSAC)                                                   *
 *  replaces, for the out-of-plane-translation stage: ransac_lib::LocallyOptimizedMSAC<        *
 *  PlanarOffsetEstimator::Reconstruction, ..., PlanarOffsetEstimator>::EstimateModel            *
 *  (init/initializer.cc:196-206, lib/RansacLib/RansacLib/ransac.h:127-428) with                 *
 *  PlanarOffsetEstimator::{MinimalSolver, EvaluateModelOnPoint} + four_view_triangulate         *
 *  (init/initializer.cc:219-333); and the triangulate-all + score half of                      *
 *  FourView2dEstimator (init/sfm2d.cc:194-213, 302-319).                                        *
 *  FourView2dEstimator::MinimalSolver (trifocal tensor) and its Ceres LeastSquares: pp_fourview2d_minimal_batch,  *
 *  pp_fourview2d_least_squares, pp_fourview2d_lomsac below.                                      *
 *  Non-finite values keep the reference's meaning: an error is the reference's nested maximum   *
 *  max(e1, max(e2, max(e3, e4))) (sfm2d.cc:316, initializer.cc:332), a score its                *
 *  min(error, threshold) sums (ransac.h:302-305) - a NaN error is never an inlier and makes the *
 *  model's score NaN, which no `score < best` accepts: the model of a degenerate sample is      *
 *  never returned, and data with NaN bearings ends with 0 inliers as in the reference.          *
 * ======================================================================================== */

/* ransac_lib::LORansacOptions (lib/RansacLib/RansacLib/ransac.h:46-92) */
typedef struct pp_lomsac_options {
  uint32_t min_num_iterations;        /* 100   */
  uint32_t max_num_iterations;        /* 10000 */
  double success_probability;         /* 0.9999 */
  double squared_inlier_threshold;    /* 1.0 (the init solvers pass an UNSQUARED threshold: SURVEY.md App. B) */
  uint32_t random_seed;               /* 0 */
  int32_t num_lo_steps;               /* 10 */
  double threshold_multiplier;        /* sqrt(2) */
  int32_t num_lsq_iterations;         /* 4 */
  int32_t min_sample_multiplicator;   /* 7 */
  int32_t non_min_sample_multiplier;  /* 3 */
  uint32_t lo_starting_iterations;    /* 50 */
  int32_t final_least_squares;        /* 0 (initialize_reconstruction sets 1) */
  uint32_t chunk_iterations;          /* speculation width, 0 = auto */
} pp_lomsac_options;
void pp_lomsac_options_default(pp_lomsac_options* o);

/* ransac_lib::RansacStatistics (ransac.h:94-101) */
typedef struct pp_lomsac_report {
  uint32_t num_iterations;
  int32_t best_num_inliers;
  double best_model_score;
  double inlier_ratio;
  int32_t number_lo_iterations;
  int32_t num_inlier_indices;
  uint64_t hypotheses_evaluated;      /* minimal samples solved + scored on the device (>= num_iterations) */
  double device_time_s, total_time_s;
} pp_lomsac_report;

typedef struct pp_planar_impl* pp_planar_handle;
/* PlanarOffsetEstimator(poses, lines_r, Rg, threshold): poses 4 x (3x4 row-major, lifted 2D cameras, t_y = 0),
 * lines 4 x n x 3 (the unaligned line of every track in each view), Rg 4 x (3x3 row-major).          */
int pp_planar_create(int32_t n, const double* poses, const double* lines, const double* Rg, int device, pp_planar_handle* out);
int pp_planar_destroy(pp_planar_handle h);
/* MinimalSolver / NonMinimalSolver for a batch of samples (sample_size 3 = minimal; larger = least squares):
 * offsets num x 3 = t_y of cameras 1..3 (NaN where the 3x3 system is singular).  One lane per sample.   */
int pp_planar_solve_batch(pp_planar_handle h, int64_t num, int32_t sample_size, const int32_t* samples, double* offsets);
/* four_view_triangulate over ALL n tracks + EvaluateModelOnPoint + MSAC score sum_i min(err_i, thr) (summed
 * in index order, one lane per model) and strict-< inlier count, for a batch of models (= offset triples). */
int pp_planar_score(pp_planar_handle h, int32_t num_models, const double* offsets, double threshold, double* msac_score,
                    int32_t* num_inliers);
/* per-track errors (n) and triangulated points (n x 3, may be NULL) of ONE model; cams_out 4 x 12 (may be NULL) */
int pp_planar_evaluate(pp_planar_handle h, const double* offsets, double* errors, double* X, double* cams_out);
/* LocallyOptimizedMSAC<...PlanarOffsetEstimator>::EstimateModel.  cams_out 4 x 12, inlier_indices n ints. */
int pp_planar_lomsac(pp_planar_handle h, const pp_lomsac_options* options, pp_lomsac_report* report, double* offsets_out,
                     double* cams_out, int32_t* inlier_indices);

typedef struct pp_pose2d_impl* pp_pose2d_handle;
/* AbsolutePose2dEstimator(x, X) (src/init/sfm2d.h:99-143, sfm2d.cc:491-530; the estimator of the reference's own
 * RansacLib tests sfm2d_test.cc:164-236): x n x 2 bearings (normalised at create, as the ctor does), X n x 2 points. */
int pp_pose2d_create(int32_t n, const double* x, const double* X, int device, pp_pose2d_handle* out);
int pp_pose2d_destroy(pp_pose2d_handle h);
/* NonMinimalSolver (== MinimalSolver for sample_size 3) per sample, one lane each: poses num x (2x3 row-major) */
int pp_pose2d_solve_batch(pp_pose2d_handle h, int64_t num, int32_t sample_size, const int32_t* samples, double* poses);
/* EvaluateModelOnPoint (1 - cos) over all n points, MSAC sum in index order, strict-< inlier count, per model */
int pp_pose2d_score(pp_pose2d_handle h, int32_t num_models, const double* poses, double threshold, double* msac_score, int32_t* num_inliers);
/* EvaluateModelOnPoint of ONE model on all n points (the vector a RansacLib Solver adaptor answers the driver's per-point
 * calls from, ppsfm/ransaclib_solvers.hpp) */
int pp_pose2d_evaluate(pp_pose2d_handle h, const double* pose, double* errors);
/* LocallyOptimizedMSAC<Pose2d, ..., AbsolutePose2dEstimator>::EstimateModel; LeastSquares == NonMinimalSolver */
int pp_pose2d_lomsac(pp_pose2d_handle h, const pp_lomsac_options* options, pp_lomsac_report* report, double* pose_out, int32_t* inlier_indices);

typedef struct pp_fourview2d_impl* pp_fourview2d_handle;
/* FourView2dEstimator(x1..x4, thr): x = 4 x n x 2 bearings (normalised to unit length at create, as the ctor does) */
int pp_fourview2d_create(int32_t n, const double* x, int device, pp_fourview2d_handle* out);
int pp_fourview2d_destroy(pp_fourview2d_handle h);
/* for each model (4 cameras, 2x3 row-major each): three_view_triangulate2d over ALL n tracks with cameras 0..2,
 * EvaluateModelOnPoint, MSAC score (index order) and strict-< inlier count                                */
int pp_fourview2d_score(pp_fourview2d_handle h, int32_t num_models, const double* cams, double threshold, double* msac_score,
                        int32_t* num_inliers);
int pp_fourview2d_evaluate(pp_fourview2d_handle h, const double* cams, double* errors, double* X);
/* EvaluateModelOnPoint against the model's OWN points X (n x 2): what the reference evaluates (sfm2d.cc:302-319) for a model
 * whose points were refined by LeastSquares instead of triangulated from cameras 0..2 */
int pp_fourview2d_evaluate_points(pp_fourview2d_handle h, const double* cams, const double* X, double* errors);
/* The reference's factorize_trifocal_tensor draws three random 2x2 coordinate changes per call
 * (Matrix2d::setRandom(), sfm2d.cc:231-235).  Here they are an input: frames = A1,A2,A3 row-major (12 doubles);
 * NULL selects the fixed set pp_fourview2d_default_frames() returns.                                       */
int pp_fourview2d_default_frames(double* frames);
/* FourView2dEstimator::MinimalSolver (sfm2d.cc:363-444) for a batch of samples, one lane per sample: trifocal
 * tensor from the sample's bearings in views 1..3, factorisation, metric upgrade, the 8 sign choices, fourth
 * camera by AbsPoseSolver.  cams num x 16 x (4 x 2x3); counts[i] = 0 or 16 (NaN models where 0).           */
int pp_fourview2d_minimal_batch(pp_fourview2d_handle h, int64_t num, int32_t sample_size, const int32_t* samples, const double* frames,
                                double* cams, int32_t* counts);
/* FourView2dEstimator::NonMinimalSolver (sfm2d.cc:446-467): MinimalSolver, MSAC score of every candidate over ALL
 * n tracks, first strictly-smallest score wins.  cams num x (4 x 2x3) (NaN, score DBL_MAX, index -1 where the
 * minimal solver returned nothing); model_index may be NULL.  Everything stays on the device in between.   */
int pp_fourview2d_nonminimal_batch(pp_fourview2d_handle h, int64_t num, int32_t sample_size, const int32_t* samples, const double* frames,
                                   double threshold, double* cams, double* msac_score, int32_t* model_index);

/* FourView2dEstimator::LeastSquares (sfm2d.cc:469-489: bundle_adjust2d on the sample when it has >= 10 tracks, then
 * optimize_points2d on ALL tracks).  cams_inout 4 x (2x3), X_inout n x 2 = the model's points.  The reference calls
 * Ceres (absent, unpinned); the device restates its Levenberg-Marquardt: parity is "same minimiser", unpinned.      */
int pp_fourview2d_least_squares(pp_fourview2d_handle h, int32_t sample_size, const int32_t* sample, double* cams_inout, double* X_inout);
/* LocallyOptimizedMSAC<..., FourView2dEstimator>::EstimateModel, everything on the device (minimal solver + scoring in
 * batches, LeastSquares by the two LM kernels).  frames as for pp_fourview2d_minimal_batch.  cams_out 4 x (2x3),
 * X_out n x 2 (may be NULL), inlier_indices n ints.                                                               */
int pp_fourview2d_lomsac(pp_fourview2d_handle h, const pp_lomsac_options* options, const double* frames, pp_lomsac_report* report,
                         double* cams_out, double* X_out, int32_t* inlier_indices);

#ifdef __cplusplus
}
#endif
#endif /* PPSFM_HIP_H_ */
