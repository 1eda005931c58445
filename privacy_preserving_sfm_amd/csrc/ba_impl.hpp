// Internal state of a pp_ba_handle: device-resident problem structure, parameter blocks, the
// materialised residual/Jacobian buffers of K1 and the normal-equation / Schur workspaces.
//
// HBM layout (all fp64 unless noted; M observations, C poses, P points, K intrinsics blocks):
//   static   : line (a,b,c) as three SoA streams la/lb/lc [M] (coalesced 8 B/lane reads),
//              obs_pose/obs_point int32 [M], pose_camera [C], camera_model [K], masks
//   CSR      : observations grouped by point (pt_start/pt_obs) and by pose (pose_start/pose_obs)
//   pairs    : for every lower-triangular 6x6 block (i >= j) of the reduced camera matrix that has at
//              least one common point: the list of (obs of pose i, obs of pose j) entries
//   params   : poses [C][7], points [P][3], intr [K][12]  (+ candidate copies for the LM trial point)
//   K1 out   : r [M][2], Jpose [M][12] (2x6 rows) or [M][14], Jpoint [M][6]  — AoS rows so the
//              per-point / per-pose gathers of K2/K3 read whole 48/96-byte rows
//   normal eq: U [C][36], gc [C][6], V [P][6] (sym), gp [P][3], Vinv [P][6], vb [P][3]
//   reduced  : S [N][N] row-major, N = roundup(6C + 1, 64); row 6C carries the rhs (augmented
//              Cholesky: the forward substitution falls out of the factorisation)
#pragma once
#include <mutex>
#include <functional>
#include <vector>

#include "common.hpp"

namespace ppsfm {

// Gather record of one observation (K3a): [ T_o = J_pt,o s_p (V+D^2)^-1 s_p (2 x 3) | J_pose,o diag(s_c) (2 x 6) | J_pt,o (2 x 3) ] = 24 doubles =
// 192 bytes = exactly three 64-byte lines.  The row side of a block pair reads T and J_pose (bytes 0..143), the column side J_pose and J_pt
// (bytes 48..191): three lines either way (two 96-byte arrays cost 3.5 on average - the gather is bound by its requests below the L2).
constexpr int kRecStride = 24;
#ifdef __HIPCC__
__device__ __forceinline__ const double* RecT(const double* rec, size_t o) { return rec + kRecStride * o; }
__device__ __forceinline__ const double* RecJ(const double* rec, size_t o) { return rec + kRecStride * o + 6; }
__device__ __forceinline__ const double* RecX(const double* rec, size_t o) { return rec + kRecStride * o + 18; }
#endif
struct ChainTask;
// launch-structure state of the dense Cholesky (cholesky.hip)
struct CholeskyAux {
  int mode = -1;                    // -1: decide at the first solve (PPSFM_CHOL_MODE); 1 = task mode (one launch), 0 = one launch per block column, 2 = by size
  bool use_graph = true;            // capture the launch structure once, replay per solve
  hipGraphExec_t graph_exec = nullptr;
  double *g_S = nullptr, *g_Linv = nullptr, *g_x = nullptr, *g_Lfac = nullptr;
  int32_t* g_flag = nullptr;
  int g_N = 0, g_rhs = 0, g_mode = -1;
  hipStream_t g_stream = nullptr;
  struct ChainTask* tasks = nullptr;      // task mode: the sorted task list for tasks_T block columns (device memory)
  int num_tasks = 0, tasks_T = 0;
  bool tasks_rejected = false;            // the list for (tasks_T, tasks_src_nz) did not pass its host replay: per-column launches for this structure
  const uint8_t* tasks_src_nz = nullptr;  // the tile map (tile_nz) the list was built for (null: dense)
  uint8_t* tasks_nz = nullptr;            // device: that map + the two sub-diagonals, closed under fill-in (what the one-launch kernel and its back substitution skip by)
  int32_t chains[1 + 3 * 16] = {1};       // the chains of the task list (ChainRanges of cholesky.hip: n, begin[16], end[16], post[16])
  double* scratch = nullptr;              // several chains: pool of 64 x 64 tiles in which a chain accumulates for another chain's tiles
  int scratch_tiles = 0;
  int critical_path = 0;                  // block-column steps on the longest dependency path of the list (T for one chain)
  double plan_ms = 0;                     // host time of the last EnsureTaskList (plan, list, replay, upload; a cache hit: the upload)
  bool test_drop_tasks = false;     // PPSFM_CHOL_TEST_DROP_TASKS=1: launch only half of the list (exercises the timeout -> per-column fallback)
  // block-sparse factor: tile_nz = tile_T x tile_T bytes (lower triangle, closed under fill-in; owned by the caller, null = dense);
  // from it: the per-launch row / super-tile lists (host + device copies) and the byte map on the device
  const uint8_t* tile_nz = nullptr;
  int tile_T = 0, sparse_T = 0, sparse_base_rows = 0, sparse_base_sups = 0;
  std::vector<int32_t> sparse_host;
  int32_t* sparse_lists = nullptr;
  uint8_t* sparse_nz = nullptr;
  bool g_sparse = false;
  int last_used = -1;               // launch structure of the last enqueued solve: PP_LINSOLVE_CHOLESKY_* (pp_ba_summary::linear_solver)
  int fallbacks = 0;                // one-launch factorisations that ran into a bounded wait and were repeated per column (pp_ba_summary::cholesky_fallbacks)
};
// closes a T x T lower-triangular tile map under the fill-in of a Cholesky factorisation (in place); returns the number of non-zero tiles
int SymbolicTileFill(int T, uint8_t* nz);
// chain steps of the one-launch factorisation of a T x T tile map (closed under fill-in): the block columns on the longest dependency path when its
// elimination tree has independent sub-trees (several chains, cholesky.hip), T otherwise; *chains (may be null): the number of chains
int CholeskyChainSteps(int T, const uint8_t* nz, int* chains = nullptr);
// the same from the chain plan alone (no task list is built or replayed): what a candidate image order costs (image_ordering.hip)
int CholeskyPlanSteps(int T, const uint8_t* nz, int* chains = nullptr);
// The image order pp_ba_create gives the reduced camera system (image_ordering.hip; host only): old_of_new empty = the caller's order.
// nnz_*: non-zero tiles of the factor in the caller's order / in the order taken (-1: not computed); dense_exit: the co-visibility turned out too dense
// for any order to pay and was not completed; chains / chain_steps: of the order taken (0: not planned); plan_ms: host time spent.
struct ImageOrdering { std::vector<int32_t> old_of_new, new_of_old; int nnz_natural = -1, nnz_ordered = -1, chains = 0, chain_steps = 0; bool dense_exit = false; double plan_ms = 0; };
ImageOrdering ChooseImageOrdering(const pp_ba_problem_desc* d, int NI, const uint64_t* graph_bits = nullptr);
bool OrderingReadsObservations(const pp_ba_problem_desc* d, int NI);      // ChooseImageOrdering would walk the observations for the co-visibility graph
int CountVariableIntrinsics(const pp_ba_problem_desc* d);
// the Schur pair lists on the device (pair_lists.hip)
bool PairListsOnDeviceEligible(int C, int64_t M);
int BuildPairListsOnDevice(int C, int64_t M, const int32_t* d_pt_start, const int32_t* d_pt_obs, const int32_t* d_obs_pose, const int32_t* d_obs_point,
                           const uint8_t* d_pose_const, const uint8_t* d_point_const, hipStream_t s, int32_t** entries_out, int64_t* num_entries,
                           std::vector<int32_t>* pair_start, std::vector<int32_t>* pair_ij, bool* fallback);
void BuildPairListsOnHost(int C, int P, int64_t M, const int32_t* pt_start, const int32_t* pt_obs, const int32_t* obs_pose, const uint8_t* list_const,
                          const uint8_t* point_const, int threads, const std::function<void(const char*)>& lap, int64_t* total_entries,
                          std::vector<int32_t>* pair_start, std::vector<int32_t>* pair_ij, std::vector<int32_t>* pair_entries);
// the co-visibility graph of the variable images from the same arrays (any image numbering): bits[i * ceil(C / 64) + (j >> 6)] bit (j & 63), j < i
int CoVisibilityOnDevice(int C, int64_t M, const int32_t* d_pt_start, const int32_t* d_pt_obs, const int32_t* d_obs_pose, const int32_t* d_obs_point,
                         const uint8_t* d_pose_const, const uint8_t* d_point_const, hipStream_t s, std::vector<uint64_t>* bits);
int PrivateIntrinsicsColumns(const pp_ba_problem_desc* d);      // n_v > 0: every image carries its own n_v variable intrinsics beside its pose columns (image_ordering.hip)
int CholeskyAuxCreate(CholeskyAux* aux);
void CholeskyAuxDestroy(CholeskyAux* aux);
}  // namespace ppsfm

namespace ppsfm {
// state of the conjugate-gradient loop of an iterative handle (ba_pcg.hip): device copy + pinned host mirror
struct PcgState { double rho, Q0, norm_b, alpha; int32_t iter, done, status, pad_; };
}  // namespace ppsfm

struct pp_ba_impl {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;

  int32_t C = 0, P = 0, K = 0;
  int64_t M = 0;
  int32_t loss_type = 0;
  double loss_scale = 1.0;
  bool intrinsics_variable = false;

  // static structure
  double *la = nullptr, *lb = nullptr, *lc = nullptr;
  int32_t *obs_pose = nullptr, *obs_point = nullptr, *obs_cam = nullptr, *pose_camera = nullptr, *camera_model = nullptr;
  uint8_t *pose_const = nullptr, *tvec_mask = nullptr, *point_const = nullptr;
  std::vector<uint8_t> host_pose_const;      // (the handle's image order) pp_ba_set_parameters checks the quaternions of variable poses against it
  int32_t *pt_start = nullptr, *pt_obs = nullptr, *pose_start = nullptr, *pose_obs = nullptr;
  int64_t num_pairs = 0, num_entries = 0;
  // every off-diagonal block of two variable poses has a pair list and nothing else writes into the pose part of S:
  // k_schur_pairs stores its blocks (no read-modify-write) and S needs no per-iteration clearing
  bool pairs_complete = false;
  // block-sparse reduced system: the 64x64 tiles of its lower triangle the co-visibility (+ fill-in) touches; sparse_tiles = worth skipping the rest
  std::vector<uint8_t> tile_nz;
  int num_nz_tiles = 0;
  bool sparse_tiles = false;
  int32_t* nz_tile_list = nullptr;      // device: (tile row, tile column) of the non-zero tiles, cleared before every assembly
  // internal image order (pp_ba_create: reverse Cuthill-McKee on the co-visibility graph when it makes the factor's tile structure
  // sparser): image `old` of the caller sits at position pose_new_of_old[old]; both empty = the caller's order.  nnz_tiles_*: non-zero
  // tiles of the factor in the caller's order / in the candidate order (-1: no ordering was considered)
  std::vector<int32_t> pose_old_of_new, pose_new_of_old;
  int nnz_tiles_natural = -1, nnz_tiles_ordered = -1;
  // position in the reduced system S of every column of the parameter vectors (gc, scale_c, diag_c, step_c: pose c at 6c .. 6c+5, intrinsics block k at
  // 6C + intr_off[k]): the identity unless every image carries its own variable intrinsics beside its pose columns (PrivateIntrinsicsColumns).  step_s: the
  // solution in S order (gathered into step_c).
  int32_t* spos = nullptr;
  std::vector<int32_t> spos_host;
  bool spos_identity = true;
  int jcam_stride = 2;              // row width of the compact camera Jacobians (the widest camera's variable parameters, even)
  bool jcam_compact = false;        // layout the last evaluation with camera Jacobians left in Jcam (EvalArgs::cam_col)
  int intr_wide_nv = 0;             // > 0: every image carries n_v variable intrinsics beside its pose columns and its (6 + n_v)-wide blocks come from the pose gather with wider rows (k_schur_wide_*)
  double* step_s = nullptr;
  double* attach_slot = nullptr;    // four doubles of the collective attach check (pp_ba_set_allreduce / pp_ba_set_communicator)
  bool structure_from_covisibility = false;      // order and tile map come from pp_ba_problem_desc::covisibility (a group's union): the same on every rank that was given it
  int structure_chains = -1, structure_steps = -1;      // pp_ba_get_structure: chains / chain steps of the one-launch factorisation (planned once)
  double create_ms[6] = {0, 0, 0, 0, 0, 0};      // pp_ba_get_create_profile
  int32_t *pair_start = nullptr, *pair_ij = nullptr, *pair_entries = nullptr;

  // variable intrinsics (refine_focal_length / principal_point / extra_params): compact columns after the 6C pose
  // columns of the reduced system, intrinsics block k at [6C + intr_off[k], + intr_nv[k])
  int32_t NI = 0;        // number of variable intrinsic parameters over all blocks
  int32_t n_red = 0;     // order of the reduced system = 6C + NI (= index of the rhs row)
  int32_t *intr_off = nullptr, *intr_nv = nullptr;   // K: compact offset (-1: nothing variable), number of variable parameters
  int32_t* intr_col = nullptr;                        // K x 12: compact column of parameter j inside its block, -1 if constant
  int32_t *cam_start = nullptr, *cam_obs = nullptr;   // CSR of the observations by intrinsics block
  int32_t* cam_np = nullptr;                          // K: number of parameters of the block's camera model
  // block pairs (row block = an intrinsics block; column block = a pose or an intrinsics block) of the reduced matrix:
  // per pair {row offset, row width, column offset, column width | kind<<8}, chunks of <= kGenChunk list entries
  int64_t gen_num_pairs = 0, gen_num_chunks = 0, isum_num_chunks = 0;
  int32_t *gen_pair = nullptr, *gen_pair_chunk = nullptr, *gen_chunk = nullptr, *gen_entries = nullptr;
  int32_t *gen_grp_start = nullptr, *gen_grp_obs = nullptr; double* gen_L = nullptr;      // direct handles: the (point, camera) groups and their L = sum J_k^T T [36 per group]
  // direct handles: the diagonal blocks S_kk apart (k_intr_kk, the layout of an iterative handle's gen_* lists)
  int32_t *kk_entries = nullptr, *kk_pair = nullptr, *kk_pair_chunk = nullptr, *kk_chunk = nullptr, *kk_multi = nullptr; double* kk_partial = nullptr;
  int64_t kk_num_groups = 0, kk_num_pairs = 0, kk_num_chunks = 0, kk_num_multi = 0;
  int32_t* gen_multi = nullptr; int64_t gen_num_multi = 0;      // the pairs that are not finished by their only chunk (none or several chunks): k_schur_gen_reduce's list
  int64_t gen_num_groups = 0;      // iterative handles: gen_entries = [group starts (gen_num_groups + 1) | observations by group], gen_chunk = (pair, first group, last group + 1)
  int32_t *isum_chunk = nullptr, *isum_cam_chunk = nullptr;
  double *gen_partial = nullptr, *isum_partial = nullptr, *cnI = nullptr, *JkS_intr = nullptr;

  // parameters
  double *poses = nullptr, *points = nullptr, *intr = nullptr;
  double *poses_c = nullptr, *points_c = nullptr, *intr_c = nullptr;

  // K1 outputs
  double *r = nullptr, *Jpose = nullptr, *Jpoint = nullptr, *Jcam = nullptr;
  int jpose_width = 0;  // 12 or 14 as last allocated
  double* partials = nullptr;  // per-block partial sums
  int num_partials = 0, partials_stride = 0, trial_partials = 0;      // K1's workgroups; doubles per region of `partials`; partials of the last trial step (0: num_partials)
  // pinned host mirrors of the K1 outputs (pp_ba_eval_host_view: the buffers a Ceres cost-function adaptor reads its slices
  // from), allocated on first use; pin_width = J_pose row width they were sized for, pin_cam = J_cam mirror present
  double *pin_r = nullptr, *pin_jpose = nullptr, *pin_jpoint = nullptr, *pin_jcam = nullptr;
  int pin_width = 0;

  // normal equations / Schur
  double *U = nullptr, *gc = nullptr, *V = nullptr, *gp = nullptr, *Vinv = nullptr, *vb = nullptr;
  double *scale_c = nullptr, *scale_p = nullptr, *diag_c = nullptr, *diag_p = nullptr;
  double *S = nullptr, *Linv = nullptr, *Lfac = nullptr, *step_c = nullptr, *step_p = nullptr;
  double *JpS = nullptr, *Q = nullptr, *norm_part = nullptr;   // JpS: the per-attempt gather records (kRecStride doubles per observation; Q unused), norm partials
  int32_t N = 0;      // padded order of S (multiple of 64), rhs row index = 6*C
  double* scal = nullptr;   // device scalars
  double* h_scal = nullptr; // pinned host mirror
  int num_effective_pose_point = 0;   // tangent dimensions of the variable poses and points (pp_ba_create)
  unsigned long long ticket_seq = 0;   // last ticket handed to a norms kernel (host polls the pinned slot for it)
  double* h_scal_dev = nullptr;   // its device-side address (the norms kernel writes the scalars there itself)
  hipEvent_t ev_readback = nullptr;   // pp_ba_solve: marks the scalar read-back of a trial step inside the stream
  int32_t* d_flag = nullptr;

  // host copies needed by the LM driver
  std::vector<double> trace;
  double timings_ms[PP_BA_T_COUNT] = {0};
  int32_t timing_calls[PP_BA_T_COUNT] = {0};

  // ITERATIVE_SCHUR + SCHUR_JACOBI (the reference's choice above 1000 images): no pair lists, no N x N system; S v applied from the records
  bool iterative = false;
  double *pcg_Sd = nullptr, *pcg_binv = nullptr, *pcg_b = nullptr, *pcg_r = nullptr, *pcg_z = nullptr, *pcg_p = nullptr, *pcg_q = nullptr, *pcg_a = nullptr,
         *pcg_dot = nullptr, *pcg_part = nullptr;      // diagonal blocks of S [C][36], their 3x3 inverses [C][2][9], rhs, CG vectors [6C], per-point product [3P], per-image dot parts
  ppsfm::PcgState *pcg_state = nullptr, *pcg_state_host = nullptr;
  double *pcg_tk = nullptr, *pcg_w = nullptr, *pcg_Scomp = nullptr, *pcg_binvI = nullptr;      // variable intrinsics: J^_k v_k and the image kernel's m per observation [2M], the intrinsics' diagonal blocks and their inverses [NI][12]
  void *pcg_pt_entry = nullptr, *pcg_pose_entry = nullptr;      // int2 [M]: (observation, image) per point-list entry, (observation, point) per image-list entry
  int linear_solver_iterations = 0;      // CG iterations of the current pp_ba_solve
  int32_t pcg_ticket = 0;                // the host's looks at the CG state are numbered (k_pcg_decide writes the number last)
  int pcg_last_iterations = 0;           // CG iterations of the handle's previous linear solve (sizes the first batch of the next one)

  // the pair lists cut into chunks of 16 or 8 entries (first entry, last + 1 per chunk; first chunk per pair) and the chunks' partial blocks: built for
  // handles with long lists (k_schur_self_chunks)
  bool pairs_chunked = false;      // a pair list is longer than 64 entries: the per-kernel path assembles the off-diagonal blocks from the chunks too (k_schur_self_chunks)
  int32_t *small_chunk = nullptr, *small_pair_chunk = nullptr;
  int small_num_chunks = 0;
  double* small_partials = nullptr;

  pp_allreduce_fn allreduce = nullptr;
  void* allreduce_ctx = nullptr;
  int32_t group_rank = 0, group_size = 1;
  struct pp_comm_impl* comm = nullptr;      // RCCL communicator of the group (pp_ba_set_communicator); excludes `allreduce`
  int64_t Spack_cap = 0;
  double* Spack = nullptr;                  // lower triangle + rhs row of S, packed for the group all-reduce
  hipEvent_t tev[8] = {nullptr};
  hipEvent_t tev_eval[2] = {nullptr, nullptr};   // deferred timing of the evaluation at an accepted point
  ppsfm::CholeskyAux chol_aux;
};

namespace ppsfm {
enum Scalar { kCost = 0, kCostCand = 1, kModelChange = 2, kGradMax = 3, kStepNorm2 = 4, kXNorm2 = 5, kTicketSlot = 6 /* host slot only */, kNumScalars = 8 };

int BaEnsureJacobianBuffers(pp_ba_impl* h, int jac_mode, int want_cam);
// K1 launchers (ba_eval.hip)
int LaunchEval(pp_ba_impl* h, int jac_mode, int want_cam, bool loss_correct, const double* poses, const double* points,
               double* cost_slot, bool compact_cam = false);
int LaunchCostOnly(pp_ba_impl* h, const double* poses, const double* points, const double* intr, double* cost_slot);
constexpr int kGenChunk = 32;      // list entries per chunk of a generic block pair (256: twelve lanes walked a chunk for ~200 us with one wavefront per CU)
constexpr int kIsumChunk = 2048;   // observations per chunk of a per-camera sum
// variable-intrinsics part of the LM iteration (ba_intr.hip)
int IntrSumsAfterEval(pp_ba_impl* h);                                   // column norms^2 -> cnI, gradient -> gc[6C..]
int IntrScale(pp_ba_impl* h, int jacobi);                               // Jacobi scale of the intrinsics columns
int IntrDiagonal(pp_ba_impl* h, double dmin, double dmax);              // clamped LM diagonal of the intrinsics columns
int IntrScaledJacobians(pp_ba_impl* h);      // JkS_intr of the current linearisation and scales (k_intr_prepare)
int IntrAssemble(pp_ba_impl* h, double inv_radius, int add_diagonal);   // rows 6C.. of S and of the rhs (after k_prepare)
// dense Cholesky of the augmented reduced system (cholesky.hip)
// doubles in the Cholesky workspace `Linv_ws` for an N x N system (N a multiple of 64): the 64x64 inverses of the diagonal
// factors and two X staging tiles
// L_kk^-1 blocks (= the M_k mailboxes), three more mailbox arrays of T + 1 slots (X, D, solved X), the progress counters of task mode
// ... and the pair inverses / pair couplings of the paired back substitution (one + four tiles per pair of block columns)
inline size_t CholeskyWorkspaceDoubles(int N) { return (size_t)(4 * (N / 64) + 3) * 64 * 64 + 8192 + (size_t)(N / 128 + 1) * 5 * 64 * 64; }
// Lfac: N x N array for the solved tiles of task mode (the factor ends up there); null = per-column mode only
std::recursive_mutex& DeviceSetupMutex();      // held while a handle allocates / uploads / captures its graph: none of that may run beside another host thread's capture
bool CholeskyWantsFactorArray(const CholeskyAux* aux, int N);      // the one-launch mode would be used for this size (it needs Lfac); block-sparse systems never do
int CholeskyPrepare(CholeskyAux* aux, int N, bool has_factor_array, hipStream_t s);      // device lists for this size (done by the first solve otherwise)
int CholeskySolveAugmented(double* S, int N, int rhs_row, double* Linv_ws, double* Lfac, double* x_out, int32_t* d_flag, hipStream_t s, CholeskyAux* aux);
// the group exchange of a point-sharded handle (ba_solver.hip): true inside a group; in-place reduction of `count` doubles on the handle's stream
bool BaInGroup(const pp_ba_impl* h);
int BaGroupReduce(pp_ba_impl* h, double* ptr, int64_t count, int op);
// matrix-free PCG on the implicit Schur complement (ba_pcg.hip)
int PcgEnsureBuffers(pp_ba_impl* h);
void PcgFreeBuffers(pp_ba_impl* h);
int PcgSolve(pp_ba_impl* h, double radius, int max_iterations, double eta, int* iterations);
// rho(s) and rho'(s) of the loss functions the reference uses (TrivialLoss, SoftLOneLoss, CauchyLoss; ceres/loss_function.cc as configured
// in src/optim/bundle_adjustment.cc:260-271)
#ifdef __HIPCC__
__device__ __forceinline__ void LossRho(int type, double scale, double s, double* rho0, double* rho1) {
  if (type == PP_LOSS_TRIVIAL) { *rho0 = s; *rho1 = 1.0; return; }
  const double b = scale * scale, c = 1.0 / b;
  const double sum = 1.0 + s * c;
  if (type == PP_LOSS_SOFT_L1) {
    const double tmp = sqrt(sum);
    *rho0 = 2.0 * b * (tmp - 1.0);
    *rho1 = fmax(2.2250738585072014e-308, 1.0 / tmp);
  } else {
    *rho0 = b * log(sum);
    *rho1 = fmax(2.2250738585072014e-308, 1.0 / sum);
  }
}

#endif

}  // namespace ppsfm
