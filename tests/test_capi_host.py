"""CPU-only checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, exports every
symbol include/ppsfm_hip.h declares, its host-side logic (sampler, trial-count rule, defaults) matches the
oracle, and compute entry points FAIL LOUDLY without a GPU (there is no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "ppsfm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pp_[a-z0-9_]+)\s*\(", text)) - {"pp_allreduce_fn"})


def test_library_builds_and_exports_every_declared_symbol():
    from privacy_preserving_sfm_amd import _capi, build
    build.build_library()
    assert os.path.exists(build.LIB)
    L = _capi.lib()
    declared = _declared_functions()
    assert len(declared) >= 28
    for name in declared:
        assert hasattr(L, name), "libppsfm_hip.so does not export %s" % name
    assert sorted(_capi.exported_symbols()) == declared


def test_no_torch_types_and_extern_c_in_header():
    text = open(os.path.join(ROOT, "include", "ppsfm_hip.h")).read()
    assert 'extern "C"' in text and "torch" not in text and "at::" not in text and "std::" not in text


def test_defaults_match_reference_presets():
    from privacy_preserving_sfm_amd.device import ba_options, ransac_options
    o = ba_options()
    # optim/bundle_adjustment.h:80-93
    assert (o.max_num_iterations, o.max_num_consecutive_invalid_steps) == (100, 10)
    assert (o.function_tolerance, o.gradient_tolerance, o.parameter_tolerance) == (0.0, 0.0, 0.0)
    assert (o.initial_trust_region_radius, o.min_relative_decrease, o.min_lm_diagonal, o.max_lm_diagonal) == (1e4, 1e-3, 1e-6, 1e32)
    r = ransac_options()
    # optim/ransac.h:47-66, util/random.h:46
    assert (r.max_error, r.min_inlier_ratio, r.confidence, r.dyn_num_trials_multiplier) == (0.0, 0.1, 0.99, 3.0)
    assert r.min_num_trials == 0 and r.max_num_trials == 2**64 - 1 and r.seed == 0


def test_host_sampler_is_the_toolchain_stream(oracle):
    from privacy_preserving_sfm_amd.device import sampler_draw
    for seed, n in ((0, 100), (5, 7), (123, 50000)):
        assert np.array_equal(sampler_draw(seed, n, 6, 300), oracle.sampler(seed, n, 6, 300))
    s = sampler_draw(0, 6, 6, 10)      # n == k: every sample is a permutation of 0..5
    assert all(sorted(row) == list(range(6)) for row in s)


def test_compute_num_trials_matches_oracle(oracle):
    from privacy_preserving_sfm_amd import _capi
    L = _capi.lib()
    for inl, n, conf, mult in ((50, 100, 0.99, 3.0), (25000, 100000, 0.99999, 3.0), (100, 100, 0.99, 3.0), (0, 100, 0.99, 3.0), (7, 9, 1.0, 1.0)):
        assert L.pp_ransac_compute_num_trials(inl, n, conf, mult) == oracle.compute_num_trials(inl, n, conf, mult)
    # the mapper's P6L preset: the a-priori cap evaluates above 10 000
    assert L.pp_ransac_compute_num_trials(25000, 100000, 0.99999, 3.0) > 10000


def test_camera_helpers():
    from privacy_preserving_sfm_amd import _capi
    L = _capi.lib()
    assert [L.pp_camera_num_params(m) for m in range(11)] == [3, 4, 4, 5, 8, 8, 12, 5, 4, 5, 12]
    assert L.pp_camera_num_params(11) == -1
    out = C.c_double(0)
    p = np.array([1000.0, 1200.0, 640, 480, 0, 0, 0, 0], dtype=np.float64)
    assert L.pp_camera_image_to_world_threshold(4, _capi.dp(p), 12.0, C.cast(C.byref(out), _capi.c_dp)) == 0
    assert out.value == 12.0 / 1100.0          # mean of (fx, fy): camera_models.h:533-543
    assert L.pp_camera_image_to_world_threshold(2, _capi.dp(p), 12.0, C.cast(C.byref(out), _capi.c_dp)) == 0
    assert out.value == 12.0 / 1000.0


def test_compute_entry_points_fail_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from privacy_preserving_sfm_amd import synthetic
    from privacy_preserving_sfm_amd._capi import PPError
    from privacy_preserving_sfm_amd.device import BAProblem, PoseProblem
    sc = synthetic.make_ba_scene(4, 10, 2, seed=1)
    with pytest.raises(PPError) as e:
        BAProblem(sc)
    assert e.value.code == -2            # PP_ERR_HIP: no silent CPU path
    rs = synthetic.make_ransac_scene(20, seed=1)
    with pytest.raises(PPError):
        PoseProblem(rs["lines"], rs["points"])


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "privacy_preserving_sfm_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle_lib" not in text and "libppsfm_oracle" not in text and '"oracle/' not in text, os.path.join(dirpath, f)


def test_cpp_host_mirror_compiles_and_links(tmp_path):
    """ppsfm/ppsfm.hpp (the C++ face of the Estimator concept / RANSAC / BA problem) builds with g++ -std=c++14
    against the C ABI and runs its host-only part."""
    import subprocess
    from privacy_preserving_sfm_amd import build
    exe = str(tmp_path / "mirror_test")
    src = os.path.join(ROOT, "tests", "cpp_mirror_compile_test.cpp")
    libdir = os.path.dirname(build.LIB)
    subprocess.check_call(["g++", "-std=c++14", "-Wall", "-Wextra", "-o", exe, src, "-L" + libdir, "-lppsfm_hip",
                           "-Wl,-rpath," + libdir])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ok kMin=6 iters=100" in out.stdout and "caught:" in out.stdout


def test_ransaclib_solver_adaptors_satisfy_the_concept(tmp_path):
    """ppsfm/ransaclib_solvers.hpp: every method the RansacLib driver calls on a Solver (reference
    lib/RansacLib/RansacLib/ransac.h:134-135, 181, 296, 314, 345, 383, 412-418) exists with the driver's signatures on the
    three adaptors; where /root/reference is present the reference's own driver template is instantiated over them too
    (oracle/Makefile, target _ref) - that build is what tests/test_gpu_ransaclib_adaptor.py runs on the device."""
    import subprocess
    from privacy_preserving_sfm_amd import build
    src = tmp_path / "concept.cpp"
    src.write_text(r"""
#include <type_traits>
#include "%s/ppsfm/ransaclib_solvers.hpp"
template <class Model, class ModelVector, class Solver> void UseAsTheDriverDoes() {
  using S = const Solver&;
  static_assert(std::is_same<decltype(std::declval<S>().min_sample_size()), int>::value, "min_sample_size");
  static_assert(std::is_same<decltype(std::declval<S>().non_minimal_sample_size()), int>::value, "non_minimal_sample_size");
  static_assert(std::is_same<decltype(std::declval<S>().num_data()), int>::value, "num_data");
  static_assert(std::is_same<decltype(std::declval<S>().MinimalSolver(std::declval<const std::vector<int>&>(), (ModelVector*)nullptr)), int>::value, "MinimalSolver");
  static_assert(std::is_same<decltype(std::declval<S>().NonMinimalSolver(std::declval<const std::vector<int>&>(), (Model*)nullptr)), int>::value, "NonMinimalSolver");
  static_assert(std::is_same<decltype(std::declval<S>().EvaluateModelOnPoint(std::declval<const Model&>(), 0)), double>::value, "EvaluateModelOnPoint");
  static_assert(std::is_same<decltype(std::declval<S>().LeastSquares(std::declval<const std::vector<int>&>(), (Model*)nullptr)), void>::value, "LeastSquares");
}
int main() {
  using namespace ppsfm::init;
  UseAsTheDriverDoes<PlanarOffsetSolver::Reconstruction, PlanarOffsetSolver::ReconstructionVector, PlanarOffsetSolver>();
  UseAsTheDriverDoes<FourView2dSolver::Reconstruction, FourView2dSolver::ReconstructionVector, FourView2dSolver>();
  UseAsTheDriverDoes<Pose2d, std::vector<Pose2d>, AbsolutePose2dSolver>();
  return 0;
}
""" % ROOT)
    libdir = os.path.dirname(build.LIB)
    exe = str(tmp_path / "concept")
    subprocess.check_call(["g++", "-std=c++14", "-Wall", "-Wextra", "-o", exe, str(src), "-L" + libdir, "-lppsfm_hip", "-Wl,-rpath," + libdir])
    assert subprocess.run([exe]).returncode == 0
    if os.path.isdir("/root/reference/lib/RansacLib"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref"])
        assert os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ransaclib_adaptor"))


def test_ceres_adaptor_type_checks_against_the_interface_shapes(tmp_path):
    """ppsfm/ceres_adaptor.hpp compiles against the three Ceres interface shapes it derives from (tests/stubs/ceres/ceres.h - a
    stub, Ceres itself is absent) and its cost functions declare the block sizes of the reference's factories
    (cost_functions.h:55-60: <2, 4, 3, 3, N>; :130-137: <2, 3, N>)."""
    import subprocess
    from privacy_preserving_sfm_amd import build
    src = tmp_path / "sizes.cpp"
    src.write_text(r"""
#include <cstdio>
#include "%s/ppsfm/ceres_adaptor.hpp"
int main() {
  ppsfm::ceres_adaptor::SlicedLineCostFunction<8> a(nullptr, 0);
  ppsfm::ceres_adaptor::SlicedConstantPoseLineCostFunction<3> b(nullptr, 0);
  std::printf("%%d", a.num_residuals());
  for (int s : a.parameter_block_sizes()) std::printf(" %%d", s);
  std::printf(" | %%d", b.num_residuals());
  for (int s : b.parameter_block_sizes()) std::printf(" %%d", s);
  return 0;
}
""" % ROOT)
    libdir = os.path.dirname(build.LIB)
    exe = str(tmp_path / "sizes")
    subprocess.check_call(["g++", "-std=c++14", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "tests", "stubs"), "-o", exe, str(src),
                           "-L" + libdir, "-lppsfm_hip", "-Wl,-rpath," + libdir])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.stdout == "2 4 3 3 8 | 2 3 3"


def test_cpp_exceptions_stop_at_the_c_boundary():
    """"never abort/throw across the boundary" (the reference aborts through CHECK, optim/bundle_adjustment.cc:261-262).  Every entry point
    is a function-try-block (csrc/common.hpp PP_API_CATCH); pp_debug_raise throws inside one: std::bad_alloc / std::length_error -> PP_ERR_NOMEM,
    anything else -> PP_ERR_INTERNAL, also from a worker thread of the host builders (ParallelFor hands the exception to the caller)."""
    from privacy_preserving_sfm_amd import _capi
    L = _capi.lib()
    expect = {0: (_capi.PP_ERR_NOMEM, b"bad_alloc"), 1: (_capi.PP_ERR_INTERNAL, b"pp_debug_raise(1)"), 2: (_capi.PP_ERR_INTERNAL, b"unknown C++ exception"),
              3: (_capi.PP_ERR_NOMEM, b"bad_alloc"), 4: (_capi.PP_ERR_NOMEM, b"length_error"), 5: (_capi.PP_ERR_INTERNAL, b"pp_debug_raise(5)")}
    for kind, (code, text) in expect.items():
        assert L.pp_debug_raise(kind) == code, kind
        assert text in L.pp_last_error(), (kind, L.pp_last_error())
    assert L.pp_debug_raise(99) == _capi.PP_OK
    # every int-returning entry point of the sources carries the guard (the four that cannot allocate are exempt)
    import glob
    exempt = {"pp_camera_num_params"}
    for path in glob.glob(os.path.join(ROOT, "privacy_preserving_sfm_amd", "csrc", "*.hip")):
        text = open(path).read()
        for m in re.finditer(r'^(?:extern "C" )?int (pp_[a-z0-9_]+)\(', text, flags=re.M):
            name = m.group(1)
            if name in exempt:
                continue
            tail = text[m.end():]
            head = tail[:tail.index("{")] if "{" in tail else ""
            if ";" in head:      # a declaration
                continue
            assert head.rstrip().endswith("try"), "%s in %s is not a function-try-block" % (name, os.path.basename(path))
            assert 'PP_API_CATCH("%s")' % name in text, name


def test_out_of_host_memory_is_an_error_code_not_an_abort():
    """The real path, not the hook: pp_ba_plan_ordering (host only) with an absurd num_points under a 3 GB address-space limit - its per-point
    lists are 2^31 x 24 bytes - returns PP_ERR_NOMEM with a message; the process lives and the next call works.  In a subprocess (the limit is
    process-wide)."""
    import subprocess
    import sys
    code = r'''
import resource, sys
import numpy as np
sys.path.insert(0, %r)
from privacy_preserving_sfm_amd import _capi, synthetic
from privacy_preserving_sfm_amd.device import _ba_desc, plan_ordering
import ctypes as C
L = _capi.lib()
sc = synthetic.make_ba_scene(12, 200, 4, seed=1, model=2)
_, ok = plan_ordering(sc)
resource.setrlimit(resource.RLIMIT_AS, (3 << 30, 3 << 30))
keep = []
d = _ba_desc(sc, keep)
d.num_points = 2**31 - 1
info = np.zeros(8, dtype=np.int32)
rc = L.pp_ba_plan_ordering(C.byref(d), None, info.ctypes.data_as(_capi.c_ip))
print("RC", rc, L.pp_last_error().decode())
_, again = plan_ordering(sc)
print("AGAIN", again == ok)
''' % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "RC -4 pp_ba_plan_ordering: out of host memory" in out.stdout, out.stdout
    assert "AGAIN True" in out.stdout


def test_covisibility_entry_point_checks_its_camera_arrays():
    """pp_ba_covisibility with a const mask but without pose_camera / camera_model (or with indices out of range) is PP_ERR_INVALID, not a
    walk through a null pointer (round-5 advice)."""
    from privacy_preserving_sfm_amd import _capi, synthetic
    from privacy_preserving_sfm_amd.device import _ba_desc
    L = _capi.lib()
    sc = synthetic.make_ba_scene(8, 100, 3, seed=2, model=2)
    keep = []
    d = _ba_desc(sc, keep)
    out = np.zeros((8, 8), dtype=np.uint8)
    assert L.pp_ba_covisibility(C.byref(d), out.ctypes.data_as(_capi.c_u8p)) == _capi.PP_OK
    d.pose_camera = None
    assert L.pp_ba_covisibility(C.byref(d), out.ctypes.data_as(_capi.c_u8p)) == _capi.PP_ERR_INVALID
    assert b"camera_const_mask without" in L.pp_last_error()
    keep2 = []
    bad = dict(sc, pose_camera=np.full(8, 5, dtype=np.int32))
    d2 = _ba_desc(bad, keep2)
    assert L.pp_ba_covisibility(C.byref(d2), out.ctypes.data_as(_capi.c_u8p)) == _capi.PP_ERR_INVALID
    d3 = _ba_desc(sc, keep2)
    d3.camera_const_mask = None      # no mask: the camera arrays are not needed
    d3.pose_camera = None
    assert L.pp_ba_covisibility(C.byref(d3), out.ctypes.data_as(_capi.c_u8p)) == _capi.PP_OK
