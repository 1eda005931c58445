// Drives the HOST code of libppsfm_hip through the C ABI under AddressSanitizer + UndefinedBehaviorSanitizer ("build should add
// -fsanitize=address,undefined CPU test config").  Linked against build/host_san/libppsfm_host_san.so (privacy_preserving_sfm_amd/build.py --host-asan:
// every translation unit's host half, kernels not compiled); needs no device.  tests/test_host_sanitizers.py builds and runs it; any sanitizer report
// ends the process with a non-zero status.  What runs: the image ordering (band / nested dissection / graph separators / early exit / variable
// intrinsics / constant images / iterative sizes), the Cholesky task planner and its host replay over dense, banded, arrow and dissected tile maps,
// the host pair-list builder on 1 / 3 / 8 threads, the co-visibility matrix, the sampler and trial-count rule, the exception containment, and 240 seeded corruptions of a valid descriptor
// through every host entry point that takes one.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <random>
#include <vector>

#include "../include/ppsfm_hip.h"

#define CHECK(cond)                                                                                         \
  do {                                                                                                      \
    if (!(cond)) { std::fprintf(stderr, "CHECK failed: %s (%s:%d) last error: %s\n", #cond, __FILE__, __LINE__, pp_last_error()); std::exit(2); } \
  } while (0)

struct Scene {
  int C = 0, P = 0, K = 1;
  std::vector<int32_t> obs_pose, obs_point, pose_camera, camera_model;
  std::vector<uint8_t> pose_const, point_const;
  std::vector<uint16_t> cam_mask;
  std::vector<double> lines;
  pp_ba_problem_desc desc() {
    pp_ba_problem_desc d;
    std::memset(&d, 0, sizeof(d));
    d.num_poses = C; d.num_points = P; d.num_cameras = K; d.num_obs = (int64_t)obs_pose.size();
    d.loss_scale = 1.0;
    lines.assign(3 * obs_pose.size(), 0.0);
    for (size_t o = 0; o < obs_pose.size(); ++o) lines[3 * o] = 1.0;
    d.lines = lines.data(); d.obs_pose = obs_pose.data(); d.obs_point = obs_point.data(); d.pose_camera = pose_camera.data(); d.camera_model = camera_model.data();
    d.pose_const = pose_const.empty() ? nullptr : pose_const.data(); d.point_const = point_const.empty() ? nullptr : point_const.data();
    d.camera_const_mask = cam_mask.empty() ? nullptr : cam_mask.data();
    d.ordering = PP_ORDERING_AUTO;
    return d;
  }
};

// every point seen by `track` images inside a window of `window` consecutive images (window <= 0: any images), observations grouped by point
static Scene MakeScene(int C, int P, int track, int window, uint64_t seed, bool per_image_cameras = false) {
  Scene s; s.C = C; s.P = P;
  std::mt19937_64 rng(seed);
  for (int p = 0; p < P; ++p) {
    const int w = window > 0 ? std::min(window, C) : C;
    const int start = (int)(rng() % (uint64_t)(C - w + 1));
    std::vector<int> im(w); std::iota(im.begin(), im.end(), start);
    std::shuffle(im.begin(), im.end(), rng);
    im.resize(std::min(track, w)); std::sort(im.begin(), im.end());
    for (int c : im) { s.obs_pose.push_back(c); s.obs_point.push_back(p); }
  }
  s.K = per_image_cameras ? C : 1;
  s.pose_camera.resize(C); for (int c = 0; c < C; ++c) s.pose_camera[c] = per_image_cameras ? c : 0;
  s.camera_model.assign(s.K, 2);      // SIMPLE_RADIAL: f, cx, cy, k
  return s;
}
static void ShuffleImages(Scene* s, uint64_t seed) {
  std::vector<int> perm(s->C); std::iota(perm.begin(), perm.end(), 0);
  std::mt19937_64 rng(seed); std::shuffle(perm.begin(), perm.end(), rng);
  for (auto& c : s->obs_pose) c = perm[c];
}
static void ShuffleObservations(Scene* s, uint64_t seed) {
  std::vector<size_t> idx(s->obs_pose.size()); std::iota(idx.begin(), idx.end(), (size_t)0);
  std::mt19937_64 rng(seed); std::shuffle(idx.begin(), idx.end(), rng);
  std::vector<int32_t> a(idx.size()), b(idx.size());
  for (size_t i = 0; i < idx.size(); ++i) { a[i] = s->obs_pose[idx[i]]; b[i] = s->obs_point[idx[i]]; }
  s->obs_pose.swap(a); s->obs_point.swap(b);
}

static void PlanOrdering(Scene s, const char* what, int expect_reordered /* -1: any */, int min_chains) {
  pp_ba_problem_desc d = s.desc();
  std::vector<int32_t> oon(s.C, -1);
  int32_t info[8];
  CHECK(pp_ba_plan_ordering(&d, oon.data(), info) == PP_OK);
  std::vector<char> seen(s.C, 0);
  for (int c = 0; c < s.C; ++c) { CHECK(oon[c] >= 0 && oon[c] < s.C && !seen[oon[c]]); seen[oon[c]] = 1; }      // a permutation
  if (expect_reordered >= 0) CHECK(info[0] == expect_reordered);
  CHECK(info[3] >= min_chains && info[4] >= 1 && info[5] >= 1);
  std::printf("ordering %-34s reordered %d tiles %d -> %d chains %d steps %d of %d columns sparse %d NI %d\n", what, info[0], info[1], info[2], info[3], info[4], info[5], info[6], info[7]);
}

static void TaskPlans() {
  auto run = [&](int T, const std::vector<uint8_t>& nz, const char* what, int expect_chains_at_least) {
    std::vector<uint8_t> map((size_t)T * T);
    int64_t count = 0;
    int32_t chains[1 + 3 * 16] = {0}, verified = 0;
    std::vector<int32_t> time(T), rho(T);
    CHECK(pp_cholesky_task_plan(T, nz.data(), 0, map.data(), nullptr, 0, &count, chains, time.data(), rho.data(), &verified) == PP_OK);
    std::vector<int32_t> tasks((size_t)16 * count);
    CHECK(pp_cholesky_task_plan(T, nz.data(), 0, map.data(), tasks.data(), count, &count, chains, time.data(), rho.data(), &verified) == PP_OK);
    CHECK(verified == 1 && chains[0] >= expect_chains_at_least && count > 0);
    int64_t c1 = 0;
    CHECK(pp_cholesky_task_list_sparse(T, nz.data(), map.data(), nullptr, 0, &c1) == PP_OK && c1 > 0);
    std::printf("task plan %-22s T %3d tasks %6lld chains %d\n", what, T, (long long)count, chains[0]);
  };
  for (int T : {4, 5, 9, 16, 33, 47, 64, 128}) {
    std::vector<uint8_t> nz((size_t)T * T, 0);
    for (int i = 0; i < T; ++i) for (int j = 0; j <= i; ++j) nz[(size_t)i * T + j] = 1;
    run(T, nz, "dense", 1);
    int64_t n = 0;
    CHECK(pp_cholesky_task_list(T, nullptr, 0, &n) == PP_OK && n > 0);
  }
  for (int band : {1, 2, 4, 9}) {
    const int T = 47;
    std::vector<uint8_t> nz((size_t)T * T, 0);
    for (int i = 0; i < T; ++i) for (int j = std::max(0, i - band); j <= i; ++j) nz[(size_t)i * T + j] = 1;
    for (int j = 0; j < T; ++j) nz[(size_t)(T - 1) * T + j] = 1;      // the right-hand side's row
    run(T, nz, "band + rhs row", 1);
  }
  {      // two independent parts + a separator: [0,12) [12,24) | [24,30)
    const int T = 30;
    std::vector<uint8_t> nz((size_t)T * T, 0);
    auto blk = [&](int a, int b) { for (int i = a; i < b; ++i) for (int j = a; j <= i; ++j) if (i - j <= 3) nz[(size_t)i * T + j] = 1; };
    blk(0, 12); blk(12, 24);
    for (int i = 24; i < T; ++i) for (int j = 0; j <= i; ++j) nz[(size_t)i * T + j] = 1;
    run(T, nz, "two parts + separator", 2);
  }
  {      // arrow
    const int T = 40;
    std::vector<uint8_t> nz((size_t)T * T, 0);
    for (int i = 0; i < T; ++i) { nz[(size_t)i * T + i] = 1; if (i) nz[(size_t)i * T + i - 1] = 1; }
    for (int i = 36; i < T; ++i) for (int j = 0; j <= i; ++j) nz[(size_t)i * T + j] = 1;
    run(T, nz, "arrow", 1);
  }
  int64_t n = 0;
  CHECK(pp_cholesky_task_list(3, nullptr, 0, &n) == PP_ERR_INVALID);      // below four block columns: refused, not walked
}

static void PairLists(Scene s, const char* what) {
  pp_ba_problem_desc d = s.desc();
  int64_t nl[3] = {0, 0, 0}, ne[3] = {0, 0, 0};
  std::vector<int32_t> ps[3], pij[3], pe[3];
  const int threads[3] = {1, 3, 8};
  for (int v = 0; v < 3; ++v) {
    CHECK(pp_ba_pair_lists_host(&d, threads[v], &nl[v], &ne[v], nullptr, nullptr, nullptr, 0, 0) == PP_OK);
    ps[v].resize(nl[v] + 1); pij[v].resize(2 * nl[v]); pe[v].resize(2 * ne[v]);
    CHECK(pp_ba_pair_lists_host(&d, threads[v], &nl[v], &ne[v], ps[v].data(), pij[v].data(), pe[v].data(), nl[v], ne[v]) == PP_OK);
    CHECK(ps[v] == ps[0] && pij[v] == pij[0] && pe[v] == pe[0]);      // the same lists whatever the thread count
  }
  // every list: ci >= cj, entries sorted by (oi, oj), observations of the right images, both observers of one variable point
  for (int64_t l = 0; l < nl[0]; ++l) {
    const int ci = pij[0][2 * l], cj = pij[0][2 * l + 1];
    CHECK(ci >= cj && cj >= 0 && ci < s.C);
    CHECK(l == 0 || pij[0][2 * l - 2] < ci || (pij[0][2 * l - 2] == ci && pij[0][2 * l - 1] < cj));
    for (int64_t e = ps[0][l]; e < ps[0][l + 1]; ++e) {
      const int oi = pe[0][2 * e], oj = pe[0][2 * e + 1];
      CHECK(s.obs_pose[oi] == ci && s.obs_pose[oj] == cj && s.obs_point[oi] == s.obs_point[oj] && oi != oj);
      CHECK(s.pose_const.empty() || (!s.pose_const[ci] && !s.pose_const[cj]));
      CHECK(s.point_const.empty() || !s.point_const[s.obs_point[oi]]);
      if (e > ps[0][l]) CHECK(pe[0][2 * e - 2] < oi || (pe[0][2 * e - 2] == oi && pe[0][2 * e - 1] < oj));
    }
  }
  // capacities are respected: nothing beyond one list / one entry is written
  std::vector<int32_t> a(2, -7), b(2, -7), c(2, -7);
  int64_t l1 = 0, e1 = 0;
  CHECK(pp_ba_pair_lists_host(&d, 1, &l1, &e1, a.data(), b.data(), c.data(), 1, 1) == PP_OK && l1 == nl[0] && e1 == ne[0]);
  // the co-visibility matrix marks exactly the off-diagonal list pairs
  std::vector<uint8_t> cov((size_t)s.C * s.C);
  CHECK(pp_ba_covisibility(&d, cov.data()) == PP_OK);
  int64_t marked = 0, offdiag = 0;
  for (int i = 0; i < s.C; ++i) for (int j = 0; j < i; ++j) { CHECK(cov[(size_t)i * s.C + j] == cov[(size_t)j * s.C + i]); marked += cov[(size_t)i * s.C + j]; }
  for (int64_t l = 0; l < nl[0]; ++l) if (pij[0][2 * l] != pij[0][2 * l + 1]) { ++offdiag; CHECK(cov[(size_t)pij[0][2 * l] * s.C + pij[0][2 * l + 1]]); }
  CHECK(marked == offdiag);
  std::printf("pair lists %-30s lists %lld entries %lld\n", what, (long long)nl[0], (long long)ne[0]);
}

// Seeded corruption of a valid descriptor: one field broken per round (an index out of range either way, a camera model that does not exist, a count
// below zero, a required array missing, a line off unit length, an unknown loss / solver / ordering).  Every host entry point that takes the descriptor
// must refuse it (or, where it does not read the broken field, answer as before) - and must not read outside the arrays it was given: that is what the
// sanitizers watch.  pp_ba_create validates before it touches the device (there is none here: a descriptor that passes ends in PP_ERR_HIP).
static void CorruptedDescriptors() {
  std::mt19937_64 rng(2024);
  int refused[4] = {0, 0, 0, 0}, rounds = 0;
  for (int it = 0; it < 240; ++it) {
    Scene s = MakeScene(24 + (int)(rng() % 40), 200 + (int)(rng() % 300), 4, 10, 100 + it, (it & 3) == 0);
    if (it % 5 == 0) { s.pose_const.assign(s.C, 0); s.pose_const[0] = 1; }
    if (it % 7 == 0) s.cam_mask.assign(s.K, (uint16_t)~0x9u);
    pp_ba_problem_desc d = s.desc();
    const size_t M = s.obs_pose.size();
    const int kind = (int)(rng() % 14);
    const size_t at = (size_t)(rng() % M);
    bool index_broken = false;
    switch (kind) {
      case 0: s.obs_pose[at] = s.C + (int)(rng() % 5); index_broken = true; break;
      case 1: s.obs_pose[at] = -1 - (int)(rng() % 5); index_broken = true; break;
      case 2: s.obs_point[at] = s.P + (int)(rng() % 5); index_broken = true; break;
      case 3: s.obs_point[at] = -1; index_broken = true; break;
      case 4: s.pose_camera[rng() % s.C] = s.K + (int)(rng() % 3); index_broken = true; break;
      case 5: s.pose_camera[rng() % s.C] = -2; index_broken = true; break;
      case 6: s.camera_model[rng() % s.K] = 11 + (int)(rng() % 50); break;
      case 7: d.num_poses = -(int)(rng() % 3); break;
      case 8: d.num_obs = -1; break;
      case 9: d.obs_pose = nullptr; break;
      case 10: d.obs_point = nullptr; break;
      case 11: s.lines[3 * at] = 2.0; break;
      case 12: d.loss_type = 7; break;
      default: d.ordering = 9; break;
    }
    int32_t info[8];
    std::vector<int32_t> oon(s.C);
    std::vector<uint8_t> cov((size_t)s.C * s.C);
    int64_t nl = 0, ne = 0;
    pp_ba_handle h = nullptr;
    const int rc[4] = {pp_ba_plan_ordering(&d, oon.data(), info), pp_ba_covisibility(&d, cov.data()),
                       pp_ba_pair_lists_host(&d, 2, &nl, &ne, nullptr, nullptr, nullptr, 0, 0), pp_ba_create(&d, 0, &h)};
    CHECK(h == nullptr && rc[3] != PP_OK);      // (no device here: never a handle)
    for (int f = 0; f < 4; ++f) {
      const bool reads_it = !(kind == 4 || kind == 5) || f == 0 || f == 3;      // (pose_camera: the co-visibility only looks with variable intrinsics, the pair lists never do)
      if ((index_broken || kind == 7 || kind == 8 || kind == 9 || kind == 10) && reads_it) CHECK(rc[f] == PP_ERR_INVALID);      // every entry point walks these
      refused[f] += rc[f] != PP_OK;
    }
    if (kind == 6 || kind == 11 || kind == 12 || kind == 13) CHECK(rc[3] == PP_ERR_INVALID);      // what only pp_ba_create looks at
    ++rounds;
  }
  std::printf("corrupted descriptors: %d rounds; refused by plan_ordering %d, covisibility %d, pair_lists_host %d, create %d\n", rounds, refused[0], refused[1], refused[2], refused[3]);
}

int main() {
  // ---- image ordering -------------------------------------------------------------------------------------------------------------------
  PlanOrdering(MakeScene(500, 6000, 6, 40, 1), "sequence 500 / window 40", -1, 2);
  { Scene s = MakeScene(500, 6000, 6, 40, 2); ShuffleImages(&s, 3); PlanOrdering(s, "the same, image ids shuffled", 1, 2); }
  { Scene s = MakeScene(300, 4000, 5, 20, 4); ShuffleObservations(&s, 5); PlanOrdering(s, "observations in random order", -1, 2); }
  PlanOrdering(MakeScene(120, 3000, 8, 0, 6), "dense (early exit)", 0, 1);
  { Scene s = MakeScene(400, 5000, 6, 30, 7, true); s.cam_mask.assign(s.K, (uint16_t)~0x9u); PlanOrdering(s, "a camera per image, f and k variable", -1, 1); }
  { Scene s = MakeScene(400, 5000, 6, 30, 8); s.cam_mask.assign(1, (uint16_t)~0x9u); PlanOrdering(s, "shared camera, f and k variable", -1, 1); }
  { Scene s = MakeScene(260, 3000, 6, 25, 9); s.pose_const.assign(s.C, 0); for (int c = 0; c < s.C; c += 7) s.pose_const[c] = 1; s.point_const.assign(s.P, 0); for (int p = 0; p < s.P; p += 5) s.point_const[p] = 1; PlanOrdering(s, "constant images and points", -1, 1); }
  { Scene s = MakeScene(1100, 4000, 4, 30, 10); PlanOrdering(s, "1100 images (iterative by size)", -1, 1); }
  {      // five clusters joined by bridge images (a photo collection: no narrow band)
    Scene s; s.C = 330; s.P = 0; s.K = 1;
    std::mt19937_64 rng(11);
    auto point = [&](std::vector<int> im) { std::sort(im.begin(), im.end()); im.erase(std::unique(im.begin(), im.end()), im.end()); for (int c : im) { s.obs_pose.push_back(c); s.obs_point.push_back(s.P); } ++s.P; };
    for (int g = 0; g < 5; ++g) for (int p = 0; p < 900; ++p) { std::vector<int> im; for (int t = 0; t < 6; ++t) im.push_back(g * 64 + (int)(rng() % 64)); point(im); }
    for (int g = 0; g < 5; ++g) for (int b = 0; b < 2; ++b) for (int p = 0; p < 40; ++p) { std::vector<int> im{320 + 2 * g + b}; for (int t = 0; t < 3; ++t) { im.push_back(g * 64 + (int)(rng() % 64)); im.push_back(((g + 1) % 5) * 64 + (int)(rng() % 64)); } point(im); }
    s.pose_camera.assign(s.C, 0); s.camera_model.assign(1, 2);
    ShuffleImages(&s, 12);
    PlanOrdering(s, "five clusters + bridge images", -1, 1);
  }
  { Scene s = MakeScene(40, 300, 4, 10, 13); pp_ba_problem_desc d = s.desc(); s.obs_pose[5] = 40; int32_t info[8]; CHECK(pp_ba_plan_ordering(&d, nullptr, info) == PP_ERR_INVALID); }
  // ---- the task planner -----------------------------------------------------------------------------------------------------------------
  TaskPlans();
  // ---- the host pair-list builder -------------------------------------------------------------------------------------------------------
  PairLists(MakeScene(60, 1500, 6, 12, 20), "sequence, grouped by point");
  { Scene s = MakeScene(60, 1500, 6, 12, 21); ShuffleObservations(&s, 22); PairLists(s, "observations in random order"); }
  { Scene s = MakeScene(50, 1200, 5, 0, 23); s.pose_const.assign(s.C, 0); s.pose_const[0] = s.pose_const[17] = 1; s.point_const.assign(s.P, 0); for (int p = 0; p < s.P; p += 3) s.point_const[p] = 1; PairLists(s, "constant images and points"); }
  { Scene s = MakeScene(30, 400, 4, 8, 24); for (int p = 0; p < 50; ++p) { s.obs_pose.push_back(s.obs_pose[4 * p]); s.obs_point.push_back(p); } PairLists(s, "a point seen twice by an image"); }
  { Scene s = MakeScene(8, 1, 8, 0, 25); PairLists(s, "one point"); }
  // ---- sampler, trial count, defaults, exception containment ----------------------------------------------------------------------------
  {
    std::vector<uint32_t> out(6 * 1000);
    CHECK(pp_sampler_draw(0, 300, 6, 1000, out.data()) == PP_OK);
    for (uint32_t v : out) CHECK(v < 300);
    CHECK(pp_sampler_draw(0, 5, 6, 1, out.data()) != PP_OK);
    CHECK(pp_ransac_compute_num_trials(50, 100, 0.99, 3.0) > 0);
    pp_ba_options o; pp_ba_options_default(&o); CHECK(o.max_num_iterations == 100);
    pp_ransac_options r; pp_ransac_options_default(&r);
    pp_lomsac_options l; pp_lomsac_options_default(&l);
    CHECK(pp_camera_num_params(2) == 4 && pp_camera_num_params(99) < 0);
  }
  CorruptedDescriptors();
  for (int kind = 0; kind <= 5; ++kind) { const int rc = pp_debug_raise(kind); CHECK(rc == ((kind == 0 || kind == 3 || kind == 4) ? PP_ERR_NOMEM : PP_ERR_INTERNAL)); }
  std::printf("host sanitizer driver: ok\n");
  return 0;
}
