#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X line-feature BA + P6L RANSAC hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): "BA iterations/sec + RANSAC hypotheses/sec, 500 cams / 200k line obs".
  * `value` / `unit` = bundle-adjustment LM iterations per second, whole job, on BASELINE configs[2]
    (500 cams / 200k line observations, 25k points x track 8, SIMPLE_RADIAL, TRIVIAL loss, gauge =
    pose[0] + tvec[1].x constant), inputs resident in HBM.  One step = ONE Levenberg-Marquardt
    iteration: Jacobian evaluation (K1) + normal equations (K2) + point-Schur assembly (K3a) + dense
    MFMA Cholesky of the 3000x3000 reduced camera system (K3b) + back-substitution, step, cost at the
    trial point (K3c).  The K steps are run as K/CHUNK_ITERS solves of CHUNK_ITERS (= 10) iterations each from the same
    noise-perturbed start (every one of those iterations is a successful step, verified from the
    summary), because an LM run that is allowed to converge stops doing full iterations.
  * `ransac` = the second half of the metric on BASELINE configs[3]: P6L (re3q3) hypotheses per second,
    one hypothesis = minimal solve + scoring of every returned model over all 50 000 correspondences.
  * N > 1: one process per GPU, each rank owns an independent 500-camera sub-model (N replicas of
    configs[2]; SURVEY.md §8e "independent sub-models": no data-path collective), WEAK scaling;
    value = total iterations of all ranks / max-over-ranks time.  BASELINE configs[4] taken literally
    (N/2 sub-models over N GPUs - 4 over 8 at N = 8) is the extra row `widened.cfg5_literal`: the same
    sub-models as whole-GPU replicas on half of the ranks, and point-sharded over groups of two ranks
    with the library's RCCL exchange.
  * `roofline`: the HBM-bound Jacobian evaluation kernel K1 (north_star's ">= 60% HBM roofline on
    Jacobian eval"), algorithmic bytes 220 B/observation, timed with HIP events on the
    library's own stream.  `kernels` adds the fp64-MFMA Cholesky and the RANSAC scoring kernel.
  * `cpu_baseline`: the oracle (CPU restatement, "port") timed on this box's host cores on a bounded
    sample of the same workloads.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix (= vector) peak, AMD spec; not listed in the guide's table
FP64_VALU_PEAK_TFLOPS = 78.6
BYTES_PER_OBS = 220.0          # 60 B in + 160 B out
SCORE_FLOP_PER_PAIR = 30.0     # per (model, correspondence)
SCORE_VALU_SLOTS_PER_PAIR = 37.25

BA_CFG = dict(num_cams=500, num_points=25000, track=8)
RANSAC_N = 50000
CHUNK_ITERS = 10    # LM iterations per solve call: every step of the first ~13 from the perturbed start is a successful (full-work) step; see run_ba


def _k1_traffic_file():
    """the newest committed PMC digest (by name: profiles/r<round><letter>_pmc.json) that holds the K1 passes"""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]*_pmc.json"))):
        try:
            with open(path) as f:
                json.load(f)["k_line_eval"]["k1"]["traffic_bytes_per_launch"]
            best = path
        except Exception:
            continue
    return best


K1_TRAFFIC_FILE = _k1_traffic_file()


def k1_traffic():
    """HBM bytes per K1 launch from the COMMITTED PMC passes (the newest profiles/r*_pmc.json, produced by tools/pmc_passes.sh +
    tools/pmc_digest.py: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of tools/profile_workload.py k1, FETCH x2 per the gfx950
    correction) - a counter run cannot share a process with the timed region, so this figure is read from the file, not measured in
    this run (`roofline.traffic_source` says so); None if absent."""
    try:
        with open(K1_TRAFFIC_FILE) as f:
            return json.load(f)["k_line_eval"]["k1"]["traffic_bytes_per_launch"]
    except Exception:
        return None


def dist_env():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    return rank, world, local


def run_ba(pb, scene, steps, opts_fn):
    """Runs exactly `steps` LM iterations (chunks of CHUNK_ITERS from the perturbed start)."""
    done = 0
    succ = 0
    while done < steps:
        k = min(CHUNK_ITERS, steps - done)
        pb.set_parameters(scene["poses"], scene["points"], None)
        s = pb.solve(opts_fn(k))
        if s.num_iterations != k:
            raise RuntimeError("LM chunk did %d of %d iterations (termination %d)" % (s.num_iterations, k, s.termination))
        done += k
        succ += s.num_successful_steps
    return succ


def cpu_baseline(scene, ransac_scene, device_params=None):
    """Oracle timed on the host cores: a bounded sample of the same two workloads."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as orc
    orc.build()
    out = {}
    L = orc.lib()
    L.orc_set_num_threads(0)                    # all host cores
    all_threads = int(L.orc_num_threads())
    # timing path: the oracle's LM loop with its cache-blocked Cholesky (oracle/linalg.h CholeskyFactorBlocked; the parity tests use the
    # simple column form).  The port does not scale to every thread count (its Schur complement and blocked Cholesky synchronise per block
    # column): it is timed at {all, 64, 16, 1} threads and the BEST rate is the stated baseline - a run at more threads is never reported
    # when fewer threads are faster.  Per count: one warm-up run, then the median of the timed runs of three LM iterations each.
    def rate_at(threads, runs, iters):
        L.orc_set_num_threads(threads)
        rates, last = [], None
        for run in range(runs + 1):
            t0 = time.time()
            last = orc.ba_solve(scene, orc.BAOptionsC.defaults(max_num_iterations=iters, blocked_cholesky=1))
            dt = time.time() - t0
            if run > 0 or runs == 1:
                rates.append(last[3].num_iterations / dt)
            if runs == 1:
                break
        rates.sort()
        return rates, last
    counts = sorted({c for c in (all_threads, 64, 16, 1) if c <= all_threads}, reverse=True)
    by_threads, best = {}, None
    oposes = opoints = None
    for c in counts:
        rates, last = rate_at(c, 5 if c == all_threads else (1 if c == 1 else 2), 1 if c == 1 else 3)
        med = rates[len(rates) // 2]
        by_threads[str(c)] = {"value": med, "runs": len(rates), "min": rates[0], "max": rates[-1]}
        if c == all_threads:
            oposes, opoints = last[0], last[1]
        if best is None or med > best[1]:
            best = (c, med, rates, int(last[3].num_iterations))
    L.orc_set_num_threads(0)
    threads, _, rates, iters_run = best
    ba_s = iters_run / rates[len(rates) // 2]
    if device_params is not None:       # the same three iterations on the device: the bench line is only valid if they agree (1e-5 rel)
        dposes, dpoints = device_params
        out["parity_vs_oracle_3_iterations"] = {"points_rel": float(np.abs(dpoints - opoints).max() / np.abs(opoints).max()),
                                                "poses_rel": float(np.abs(dposes - oposes).max() / np.abs(oposes).max())}
        if max(out["parity_vs_oracle_3_iterations"].values()) > 1e-5:
            raise RuntimeError("device BA differs from the oracle after 3 iterations: %r" % out["parity_vs_oracle_3_iterations"])
    out["value"] = rates[len(rates) // 2]
    out["unit"] = "LM iterations/s"
    out["cores"] = threads
    out["kind"] = "port"
    out["runs"] = {"min": rates[0], "median": rates[len(rates) // 2], "max": rates[-1], "count": len(rates)}
    out["by_threads"] = by_threads
    out["sample"] = ("best of the thread counts %s (by_threads): median of %d runs of %d LM iterations of the same 500 cam / 200k obs problem (whole solve incl. its "
                     "set-up), oracle/bundle_adjustment.h with the blocked Cholesky, OpenMP on %d threads (%.2f s per run)" % (counts, len(rates), iters_run, threads, ba_s))
    out["one_thread"] = dict(by_threads.get("1", {}), unit="LM iterations/s", cores=1, sample="1 LM iteration, single run")
    # RANSAC: a few hundred hypotheses over all 50k correspondences, single thread like optim/ransac.h:213-249
    from privacy_preserving_sfm_amd.device import sampler_draw
    H = 64
    samples = sampler_draw(0, RANSAC_N, 6, H)
    t, nm, _ = orc.p6l_hypotheses_timed(ransac_scene["lines"], ransac_scene["points"], ransac_scene["aligned"], samples,
                                        ransac_scene["max_error"] ** 2)
    if t < 3.0:
        H = int(min(4096, max(64, H * 6.0 / max(t, 1e-3))))
        samples = sampler_draw(0, RANSAC_N, 6, H)
        t, nm, _ = orc.p6l_hypotheses_timed(ransac_scene["lines"], ransac_scene["points"], ransac_scene["aligned"], samples,
                                            ransac_scene["max_error"] ** 2)
    out["ransac"] = {"value": H / t, "unit": "hypotheses/s", "cores": 1, "kind": "port",
                     "sample": "%d P6L hypotheses (%d models) x 50000 correspondences, oracle/ransac.h, 1 thread (%.1f s)" % (H, nm, t)}
    # the four-view LO-MSAC of widened.fourview2d_lomsac (lib/RansacLib over FourView2dEstimator, src/init/sfm2d.cc): the oracle's sequential loop on the same
    # 2000 tracks, a bounded sample of 256 iterations (the device row runs 4096) - one thread, as the reference's driver
    try:
        from privacy_preserving_sfm_amd import synthetic as _syn
        fsc = _syn.make_scene_2d(4, 2000, n_outliers=400, seed=7)
        fo = orc.LoMsacOptionsC.defaults()
        fo.squared_inlier_threshold = 1e-6; fo.min_num_iterations = 256; fo.max_num_iterations = 256
        from privacy_preserving_sfm_amd.device import fourview2d_default_frames
        t0 = time.time()
        finl, _, _, fst, _ = orc.fourview2d_lomsac(fsc["x"], fourview2d_default_frames(), fo)
        ft = time.time() - t0
        out["fourview2d_lomsac"] = {"wall_s": ft, "iterations": int(fst.num_iterations), "lo_runs": int(fst.number_lo_iterations), "best_inliers": int(finl), "cores": 1, "kind": "port",
                                    "s_per_lo_run": ft / max(1, int(fst.number_lo_iterations)),
                                    "sample": "256 RANSAC iterations + their local optimisations on 2000 tracks / 400 outliers, oracle/init_solvers.h, 1 thread"}
    except Exception as e:      # (the row is a side measurement: it never costs the headline its baseline)
        out["fourview2d_lomsac"] = {"error": repr(e)}
    out["host_cores_available"] = os.cpu_count()
    return out


class GpuBackend:
    """What bench.py needs from the machine: the GPU library + RCCL.  tests/test_distributed_cpu.py runs main() with a CPU stand-in
    (gloo, a stub problem whose solve performs the group exchange through torch.distributed) so that the N > 1 control flow - group
    construction, communicator id broadcast, sharding, timing reduction, the JSON line - is executed before an 8-GPU node sees it."""
    dist_backend = "nccl"
    device_type = "cuda"
    has_gpu_rows = True

    def init(self, local):
        import torch
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
        torch.cuda.set_device(local)
        self.local = local

    def init_process_group(self):
        import torch
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", self.local))

    def bind_thread(self):      # the current HIP device is per host thread
        import torch
        torch.cuda.set_device(self.local)

    def sync(self):
        import torch
        torch.cuda.synchronize()

    def ba_problem(self, scene):
        from privacy_preserving_sfm_amd.device import BAProblem
        return BAProblem(scene, device=self.local)

    def communicator(self, group):
        from privacy_preserving_sfm_amd.distributed import make_communicator
        return make_communicator(group, device=self.local)


def BAProblemLS(scene, device, linear_solver):
    from privacy_preserving_sfm_amd.device import BAProblem
    return BAProblem(scene, device=device, linear_solver=linear_solver)


def make_scene(model_id):
    from privacy_preserving_sfm_amd import synthetic
    return synthetic.make_ba_scene(BA_CFG["num_cams"], BA_CFG["num_points"], BA_CFG["track"], seed=0xC0FFEE + 3 + 101 * model_id, model=2)


def opts_fn(k):
    from privacy_preserving_sfm_amd.device import ba_options
    return ba_options(max_num_iterations=k, gradient_tolerance=0.0)


def timed_ba(be, use_dist, pb, scene, warmup, steps, active=True):
    """W untimed + K timed LM iterations between barriers; returns (max-over-ranks seconds, successful steps).  Ranks with
    active = False only take part in the barriers and the reduction (they hold no sub-model in this row)."""
    import torch.distributed as dist

    def barrier():
        be.sync()
        if use_dist:
            dist.barrier()
        be.sync()

    if active:
        run_ba(pb, scene, max(warmup, 1), opts_fn)
    barrier()
    t0 = time.perf_counter()
    succ = run_ba(pb, scene, steps, opts_fn) if active else 0
    be.sync()
    elapsed = time.perf_counter() - t0
    if use_dist:
        from privacy_preserving_sfm_amd.distributed import max_over_ranks
        elapsed = max_over_ranks(elapsed if active else 0.0, be.device_type)
    barrier()
    return elapsed, succ


def concurrent_submodels(be, scene, steps, counts=(1, 2, 4)):
    """k independent sub-models sharing ONE GPU: one handle (own stream) and one host thread each; aggregate LM iterations/s and
    the one-launch factorisations that timed out under the contention (pp_ba_summary::cholesky_fallbacks)."""
    import threading
    rows = []
    for k in counts:
        pbs = [be.ba_problem(scene) for _ in range(k)]
        fallbacks = [0] * k
        errors = []

        def work(i, n):
            try:
                done = 0
                while done < n:
                    c = min(CHUNK_ITERS, n - done)
                    pbs[i].set_parameters(scene["poses"], scene["points"], None)
                    s = pbs[i].solve(opts_fn(c))
                    if s.num_iterations != c:
                        raise RuntimeError("LM chunk did %d of %d iterations" % (s.num_iterations, c))
                    fallbacks[i] = int(s.cholesky_fallbacks)
                    done += c
            except Exception as e:      # noqa
                errors.append(repr(e))

        for n, timed in ((CHUNK_ITERS, False), (steps, True)):
            th = [threading.Thread(target=work, args=(i, n)) for i in range(k)]
            be.sync()
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            be.sync()
            dt = time.perf_counter() - t0
        for pb in pbs:
            pb.close()
        rows.append({"handles": k, "value": k * steps / dt, "unit": "LM iterations/s (aggregate)", "ms_per_iteration_per_handle": 1e3 * dt / steps,
                     "cholesky_fallbacks": int(sum(fallbacks)), "errors": errors})
    return rows


def main(argv=None, backend=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--ransac-hyp", type=int, default=1048576, help="hypotheses in the RANSAC leg (cfg 4: 1M)")
    ap.add_argument("--submodels", type=int, default=0, help="N > 1 only: number of independent sub-models of the HEADLINE run (default = N: one per GPU, no "
                    "data-path collective).  M < N (M divides N): every sub-model is point-sharded over N/M ranks that exchange the "
                    "normal equations per LM iteration through the library's RCCL collectives (pp_ba_set_communicator)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ransac", action="store_true")
    ap.add_argument("--no-widened", action="store_true", help="skip the rows beside the headline (widened.*)")
    ap.add_argument("--no-beyond-l3", action="store_true", help="skip the 2M-observation K1 launches (roofline.beyond_l3); use it for the rocprofv3 --stats run whose\n                    k_line_eval average profiles/ compares with roofline.ms_per_launch (the stats file averages over all launches of a kernel name)")
    ap.add_argument("--cfg5-timeout", type=float, default=240.0, help="seconds after which the widened.cfg5_literal row (the first thing that exercises the "
                    "multi-rank RCCL exchange) is abandoned and the line printed without it")
    args = ap.parse_args(argv)
    be = backend or GpuBackend()

    rank, world, local = dist_env()
    if world != args.gpus and world > 1:
        print("warning: WORLD_SIZE %d != --gpus %d" % (world, args.gpus), file=sys.stderr)
    be.init(local)
    use_dist = world > 1
    if use_dist:
        import torch.distributed as dist
        be.init_process_group()

    from privacy_preserving_sfm_amd import synthetic
    from privacy_preserving_sfm_amd.distributed import make_submodel_groups, shard_scene_by_points, submodel_layout

    # ---- headline workload: every rank owns one 500-camera sub-model (different seed per rank), or, with --submodels M < N, the ranks
    # of a group of N/M share one (points and their observations sharded, poses replicated, RCCL exchange inside the library)
    submodels = args.submodels if (use_dist and args.submodels > 0) else world
    model_id, grank, gsize = submodel_layout(rank, world, submodels)
    full_scene = make_scene(model_id)
    scene, comm = full_scene, None
    if gsize > 1:
        groups = make_submodel_groups(world, submodels)
        comm = be.communicator(groups[model_id])
        scene = shard_scene_by_points(full_scene, grank, gsize)
    pb = be.ba_problem(scene)
    if comm is not None:
        pb.set_communicator(comm)
    M = pb.M
    elapsed, succ = timed_ba(be, use_dist, pb, scene, args.warmup, args.steps)
    timings, last_summary = None, None
    if be.has_gpu_rows:
        # phase breakdown: a separate, untimed pass with the per-phase HIP events switched on (each event record costs ~5 us of
        # stream time, so the timed run above leaves them off)
        from privacy_preserving_sfm_amd.device import ba_options
        pb.set_parameters(scene["poses"], scene["points"], None)
        last_summary = pb.solve(ba_options(max_num_iterations=CHUNK_ITERS, gradient_tolerance=0.0, phase_timings=1))
        timings = pb.timings()

    result = None
    n = 6 * BA_CFG["num_cams"]
    if rank == 0:
        total_steps = args.steps * submodels      # LM iterations of all sub-models (a group's ranks iterate together)
        value = total_steps / elapsed
        if world == 1:
            workload = "configs[2]: 500 cams / 200k line obs full BA (K1+K2+K3, MFMA Schur solve) on 1xMI355X"
        elif gsize == 1:
            workload = ("%d replicas of configs[2] (500 cams / 200k line obs full BA), one independent sub-model per GPU, no data-path collective: WEAK "
                        "scaling; configs[4] taken literally (%d sub-models over %d GPUs) is the row widened.cfg5_literal" % (world, max(world // 2, 1), world))
        else:
            workload = ("configs[4] shape: %d sub-models of configs[2]'s size, each point-sharded over %d ranks with the RCCL normal-equation "
                        "all-reduce inside the library" % (submodels, gsize))
        result = {
            "metric": "BA iterations/sec + RANSAC hypotheses/sec, 500 cams / 200k line obs",
            "value": value, "unit": "LM iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload,
                       "cams": BA_CFG["num_cams"], "points": BA_CFG["num_points"], "obs": int(M), "camera_model": "SIMPLE_RADIAL",
                       "reduced_system": n, "successful_steps": int(succ), "lm_chunk": CHUNK_ITERS,
                       "submodels": int(submodels), "ranks_per_submodel": int(gsize),
                       "exchange": "none (independent sub-models)" if gsize == 1 else "RCCL all-reduce of U/g_c, packed lower triangle of S, scalars per LM iteration"},
        }
        if last_summary is not None:
            result["config"]["linear_solver"] = LINSOLVE_NAMES.get(int(last_summary.linear_solver), str(int(last_summary.linear_solver)))
            result["config"]["cholesky_fallbacks"] = int(last_summary.cholesky_fallbacks)
    if rank == 0 and be.has_gpu_rows:
        from privacy_preserving_sfm_amd.device import dense_cholesky_solve
        result["phase_ms_per_call"] = {k: (v[0] / v[1] if v[1] else 0.0) for k, v in timings.items()}
        # ---- K1 roofline: HIP events on the library's stream around 200 launches ---------------
        pb.set_parameters(scene["poses"], scene["points"], None)
        pb.evaluate_device(repeat=20)
        k1_ms = pb.evaluate_device(repeat=200)
        k1_gbs = BYTES_PER_OBS * M / (k1_ms * 1e-3) / 1e9
        # ---- Cholesky (fp64 MFMA) on a matrix of the reduced system's size ------------------
        rng = np.random.default_rng(0)
        B = rng.normal(size=(n, 64))
        A = B @ B.T + n * np.eye(n)
        _, chol_ms = dense_cholesky_solve(A, rng.normal(size=n), device=local, repeat=5)
        chol_flops = n ** 3 / 3.0 + 2.0 * n * n
        result["roofline"] = {"kernel": "k_line_eval (K1 Jacobian+residual eval)", "bound": "hbm", "achieved": k1_gbs, "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": k1_gbs / HBM_PEAK_GBS, "traffic": k1_traffic(),
                              "traffic_source": "read from %s (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 correction applied); NOT "
                                                "measured in this run: the counter passes are their own command (tools/pmc_passes.sh, last run on the final code of round 5; k_line_eval itself is unchanged since round 3)"
                                                % (os.path.relpath(K1_TRAFFIC_FILE, ROOT) if K1_TRAFFIC_FILE else None),
                              "ms_per_launch": k1_ms, "bytes_per_launch": BYTES_PER_OBS * M}
        result["kernels"] = {"cholesky_3000": {"bound": "mfma", "ms": chol_ms, "achieved": chol_flops / (chol_ms * 1e-3) / 1e12,
                                               "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                               "frac": chol_flops / (chol_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS}}
        result["roofline"]["cholesky"] = dict(result["kernels"]["cholesky_3000"], kernel="k_potrf64 + k_cholesky_tasks (one launch: persistent chain workgroup + "
                                              "task list; PPSFM_CHOL_MODE=columns: 46 x k_column_step) + k_backsub_pairs (K3b, the largest share of an LM "
                                              "iteration)", flops_per_solve=chol_flops)
        # K1 beyond the Infinity Cache: cfg 3's 44 MB per launch (and anything below 256 MiB) can be served by the MALL, which the
        # FETCH/WRITE_SIZE counters do not separate from HBM.  The same kernel on a 2M-observation problem moves 440 MB per launch.
        try:
            if args.no_beyond_l3:
                raise RuntimeError("--no-beyond-l3")
            big = synthetic.make_ba_scene(BA_CFG["num_cams"], 10 * BA_CFG["num_points"], BA_CFG["track"], seed=1, model=2)
            pbig = be.ba_problem(big)
            pbig.evaluate_device(repeat=5)
            big_ms = pbig.evaluate_device(repeat=30)
            big_gbs = BYTES_PER_OBS * pbig.M / (big_ms * 1e-3) / 1e9
            result["roofline"]["beyond_l3"] = {"obs": int(pbig.M), "bytes_per_launch": BYTES_PER_OBS * pbig.M, "ms_per_launch": big_ms,
                                               "achieved": big_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": big_gbs / HBM_PEAK_GBS,
                                               "note": "same kernel, 2M observations: 440 MB per launch > 256 MiB Infinity Cache"}
            pbig.close()
        except Exception as e:
            result["roofline"]["beyond_l3"] = {"skipped": str(e)}
    # ---- rows widened after the hot path: post-BA filters on the same handle, four-view initialisation ----
    if rank == 0 and world == 1 and be.has_gpu_rows and not args.no_widened:
        try:
            f = float(scene["intr"][0, 0])
            cam_size = np.tile(np.array([[int(4 * f), int(4 * f)]], dtype=np.int32), (scene["intr"].shape[0], 1))
            pb.filter_points(4.0, 1.5, cam_size)                                   # warm-up
            t0 = time.perf_counter()
            frep, _, _, _ = pb.filter_points(4.0, 1.5, cam_size)
            filt_s = time.perf_counter() - t0
            from privacy_preserving_sfm_amd.device import FourView2dProblem, lomsac_options
            isc = synthetic.make_scene_2d(4, 2000, n_outliers=400, seed=7)
            fv = FourView2dProblem(isc["x"], device=local)
            fv.lomsac(lomsac_options(squared_inlier_threshold=1e-6, min_num_iterations=256, max_num_iterations=256))   # warm-up
            t0 = time.perf_counter()
            irep, _, _, _ = fv.lomsac(lomsac_options(squared_inlier_threshold=1e-6, min_num_iterations=4096, max_num_iterations=4096))
            init_s = time.perf_counter() - t0
            fv.close()
            from privacy_preserving_sfm_amd.device import triangulate_tracks, triangulation_options
            tsc = synthetic.make_track_scene(64, 25000, seed=1, min_len=8, max_len=8, outlier_frac=0.1)
            targs = (tsc["track_start"], tsc["lines"], tsc["obs_view"], tsc["P"], tsc["centers"], tsc["view_camera"], tsc["camera_model"], tsc["intr"], tsc["cam_size"],
                     triangulation_options(min_tri_angle=0.02, residual_type=0, max_error=2e-3))
            triangulate_tracks(*targs, device=local)
            tok, _, _, tnt, tri_ms = triangulate_tracks(*targs, device=local)
            # cfg 3's size with the co-visibility of a SEQUENCE (every point seen by 8 of 40 consecutive images): the reduced camera system is
            # block-banded and the device skips its empty 64x64 tiles (the reference runs Ceres' SPARSE_SCHUR there, bundle_adjustment.cc:275-286)
            bsc = synthetic.make_ba_scene(BA_CFG["num_cams"], BA_CFG["num_points"], BA_CFG["track"], seed=0xC0FFEE + 3, model=2, window=40)
            pbb = be.ba_problem(bsc)
            run_ba(pbb, bsc, CHUNK_ITERS, opts_fn)
            t0 = time.perf_counter()
            run_ba(pbb, bsc, 2 * CHUNK_ITERS, opts_fn)
            band_s = time.perf_counter() - t0
            band_struct = pbb.structure()
            pbb.close()
            # the same scene with its image ids SHUFFLED (a reconstruction whose images were not registered in capture order): dense in the
            # caller's order; pp_ba_create renumbers the images (reverse Cuthill-McKee on the co-visibility graph, Ceres' SPARSE_SCHUR ordering
            # for the reference) and the block-banded system is back
            ssc, _ = synthetic.shuffle_image_ids(bsc, seed=1)
            pbs = be.ba_problem(ssc)
            run_ba(pbs, ssc, CHUNK_ITERS, opts_fn)
            t0 = time.perf_counter()
            run_ba(pbs, ssc, 2 * CHUNK_ITERS, opts_fn)
            shuf_s = time.perf_counter() - t0
            shuf_struct = pbs.structure()
            pbs.set_parameters(ssc["poses"], ssc["points"], None)
            shuf_sm = pbs.solve(opts_fn(CHUNK_ITERS))
            pbs.close()
            # the largest sequence scene the reference still solves directly (1000 images: SPARSE_SCHUR up to there, bundle_adjustment.cc:279-286)
            lsc = synthetic.make_ba_scene(1000, 50000, BA_CFG["track"], seed=0xC0FFEE + 3, model=2, window=40)
            pbl = be.ba_problem(lsc)
            run_ba(pbl, lsc, CHUNK_ITERS, opts_fn)
            t0 = time.perf_counter()
            run_ba(pbl, lsc, 2 * CHUNK_ITERS, opts_fn)
            seq1000_s = time.perf_counter() - t0
            seq1000_struct = pbl.structure()
            pbl.close()
            result["widened"] = {
                "banded_1000_images": {"cams": 1000, "obs": int(len(lsc["obs_pose"])), "window": 40, "value": 2 * CHUNK_ITERS / seq1000_s, "unit": "LM iterations/s",
                                       "tiles": seq1000_struct["tiles"], "nonzero_tiles": seq1000_struct["nnz_used"], "chains": seq1000_struct.get("chains"),
                                       "chain_steps": seq1000_struct.get("chain_steps"), "block_columns": 94},
                "banded_cfg3": {"cams": BA_CFG["num_cams"], "obs": int(len(bsc["obs_pose"])), "window": 40, "value": 2 * CHUNK_ITERS / band_s, "unit": "LM iterations/s",
                                "note": "same size as the headline, block-banded reduced system (block-sparse assembly / Cholesky / back substitution; images ordered by nested dissection, one chain workgroup per part)",
                                "tiles": band_struct["tiles"], "nonzero_tiles": band_struct["nnz_used"], "reordered": band_struct["reordered"],
                                "chains": band_struct.get("chains"), "chain_steps": band_struct.get("chain_steps"),
                                "shuffled_image_ids": {"value": 2 * CHUNK_ITERS / shuf_s, "unit": "LM iterations/s", "tiles": shuf_struct["tiles"],
                                                       "nonzero_tiles_callers_order": shuf_struct["nnz_natural"], "nonzero_tiles_after_ordering": shuf_struct["nnz_used"],
                                                       "reordered": shuf_struct["reordered"], "block_sparse": shuf_struct["block_sparse"], "chains": shuf_struct.get("chains"), "chain_steps": shuf_struct.get("chain_steps"),
                                                       "linear_solver": LINSOLVE_NAMES.get(int(shuf_sm.linear_solver))}},
                "triangulate_tracks": {"tracks": 25000, "observations": int(tsc["track_start"][-1]), "device_ms": tri_ms, "value": 25000 / (tri_ms * 1e-3),
                                       "unit": "tracks/s (one LORANSAC each)", "mean_trials": float(np.mean(tnt)), "success": float(np.mean(tok))},
                "filter_points3d": {"observations": int(M), "wall_ms": 1e3 * filt_s, "value": M / filt_s, "unit": "observations/s (host wall, incl. mask read-back)",
                                    "num_filtered": int(frep.num_filtered)},
                "fourview2d_lomsac": {"tracks": 2000, "iterations": int(irep.num_iterations), "lo_runs": int(irep.number_lo_iterations),
                                      "hypotheses": int(irep.hypotheses_evaluated), "device_s_minimal_and_score": float(irep.device_time_s), "wall_s": init_s,
                                      "value": irep.hypotheses_evaluated / max(irep.device_time_s, 1e-12), "unit": "minimal samples/s (16 candidates each, scored on all tracks)",
                                      "best_inliers": int(irep.best_num_inliers)}}
            # the headline problem with refine_focal_length / refine_extra_params (bundle_adjustment.cc:490-528): f and k of SIMPLE_RADIAL variable - one camera
            # shared by all images (3000 + 2 columns), then a camera per image (3000 + 1000 columns); the direct solve (500 images: SPARSE_SCHUR in the reference)
            virows = {}
            for vname, nintr, vwin in (("shared_camera", 1, None), ("camera_per_image", 500, None), ("banded_shared_camera", 1, 40), ("banded_camera_per_image", 500, 40)):
                # (banded_*: the sequence scene of widened.banded_cfg3 - the intrinsics rows are the border of an arrow, the block-sparse path and the dissection apply)
                vsc = synthetic.make_ba_scene(BA_CFG["num_cams"], BA_CFG["num_points"], BA_CFG["track"], seed=0xC0FFEE + 3, model=2, num_intrinsics=nintr,
                                              **({"window": vwin} if vwin else {}))
                vsc["camera_const_mask"] = np.full(nintr, 0b0110, dtype=np.uint16)
                pbv = be.ba_problem(vsc)
                vo = opts_fn(CHUNK_ITERS)
                pbv.solve(vo)
                pbv.set_parameters(vsc["poses"], vsc["points"], vsc["intr"])
                t0 = time.perf_counter()
                smv = pbv.solve(vo)
                dtv = time.perf_counter() - t0
                virows[vname] = {"value": smv.num_iterations / dtv, "unit": "LM iterations/s", "intrinsics_blocks": nintr, "reduced_system": 6 * BA_CFG["num_cams"] + 2 * nintr,
                                 "linear_solver": LINSOLVE_NAMES.get(int(smv.linear_solver)), "cost_after_10_iterations": float(smv.final_cost)}
                pbv.close()
            result["widened"]["cfg3_variable_intrinsics"] = dict(virows, note="configs[2] with the focal length and the distortion of its SIMPLE_RADIAL cameras variable: the "
                                                                 "intrinsics rows of the reduced system are assembled in factored form (k_intr_L, k_schur_gen, k_intr_kk) for a shared camera; a camera per image sits beside its "
                                                                 "image's pose columns and is assembled with them as one (6 + n_v)-wide block (k_schur_wide_self / k_schur_wide_pairs)")
            # above 1000 images the reference switches to ITERATIVE_SCHUR + SCHUR_JACOBI (bundle_adjustment.cc:283-286): matrix-free PCG here
            isc2 = synthetic.make_ba_scene(1100, 22000, 8, seed=0xC0FFEE + 5, model=2)
            rows = {}
            for name, ls in (("iterative_schur", 0), ("direct", 1)):
                pbi = be.ba_problem(isc2) if ls == 0 else BAProblemLS(isc2, local, ls)
                run_ba(pbi, isc2, CHUNK_ITERS, opts_fn)
                t0 = time.perf_counter()
                run_ba(pbi, isc2, CHUNK_ITERS, opts_fn)
                dt = time.perf_counter() - t0
                pbi.set_parameters(isc2["poses"], isc2["points"], None)
                sm = pbi.solve(opts_fn(CHUNK_ITERS))
                rows[name] = {"value": CHUNK_ITERS / dt, "unit": "LM iterations/s", "cost_after_10_iterations": float(sm.final_cost),
                              "linear_solver": LINSOLVE_NAMES.get(int(sm.linear_solver)), "cg_iterations_per_lm_iteration": sm.linear_solver_iterations / CHUNK_ITERS}
                pbi.close()
            # the same with refine_focal_length / refine_extra_params (bundle_adjustment.cc:490-528): one SIMPLE_RADIAL camera shared by all images, f and k variable
            isc3 = synthetic.make_ba_scene(1100, 22000, 8, seed=0xC0FFEE + 5, model=2, num_intrinsics=1)
            isc3["camera_const_mask"] = np.full(1, 0b0110, dtype=np.uint16)
            vrows = {}
            for name, ls in (("iterative_schur", 0), ("direct", 1)):
                pbi = BAProblemLS(isc3, local, ls)
                vo = opts_fn(CHUNK_ITERS)
                pbi.solve(vo)
                pbi.set_parameters(isc3["poses"], isc3["points"], isc3["intr"])
                t0 = time.perf_counter()
                sm = pbi.solve(vo)
                dt = time.perf_counter() - t0
                vrows[name] = {"value": sm.num_iterations / dt, "unit": "LM iterations/s", "cost_after_10_iterations": float(sm.final_cost),
                               "linear_solver": LINSOLVE_NAMES.get(int(sm.linear_solver)), "cg_iterations_per_lm_iteration": sm.linear_solver_iterations / max(sm.num_iterations, 1)}
                pbi.close()
            rows["variable_intrinsics"] = dict(vrows, note="one shared SIMPLE_RADIAL camera, focal length and distortion variable: 6600 pose + 2 intrinsics columns")
            result["widened"]["iterative_schur_1100"] = dict(rows, cams=1100, obs=int(len(isc2["obs_pose"])), note="1100 images / 176k observations: the handle picks "
                                                             "the solver by the image count like BundleAdjuster::Solve; `direct` = the same problem forced onto the dense "
                                                             "Cholesky (6600 columns).  Inexact steps (eta = 0.1) cost less per iteration and gain less per iteration")
            # several sub-models sharing this GPU (SURVEY §7 "batching several sub-models per launch"; the mapper's many local BAs): the
            # factorisation's chain leaves most of the chip idle, concurrent handles fill it
            # (under rocprofv3 the rows with several launching host threads are left out: its kernel-trace tool has crashed - SIGSEGV inside the
            # launch hook of a worker thread, in 2 of 8 profiled runs of this command with the 16-handle row - and timings under a tracer are not
            # what these rows are for; the kernels of a handle are the same ones the single-handle rows launch)
            profiled = "rocprofiler" in os.environ.get("LD_PRELOAD", "") or "ROCP_TOOL_LIBRARIES" in os.environ
            if profiled:
                result["widened"]["concurrent_submodels"] = result["widened"]["concurrent_local_ba"] = {"skipped": "under rocprofv3 (rows with several launching host threads)"}
            else:
                result["widened"]["concurrent_submodels"] = {
                    "note": "k handles of the headline problem on ONE GPU, one host thread + stream each; value = aggregate over the handles",
                    "rows": concurrent_submodels(be, scene, 2 * CHUNK_ITERS)}
                # the mapper's LOCAL bundle adjustments (src/sfm/incremental_mapper.cc:857-858: a handful of images around the new one, after
                # every registration): configs[0]-sized problems, launch-bound one at a time - how many of them one GPU turns over together
                # ONE local bundle adjustment as the mapper issues it (src/sfm/incremental_mapper.cc:813-858, controllers/incremental_mapper.cc:196-219:
                # a new BundleAdjuster per registered image, 6 images, SOFT_L1 for the first refinement then TRIVIAL, at most 25 iterations each):
                # the wall of create + both solves + read-back + destroy
                from privacy_preserving_sfm_amd.device import ba_options as _bo
                l6 = synthetic.make_ba_scene(6, 334, 6, seed=0xC0FFEE + 7, model=2)
                walls, its = [], 0
                for rep in range(6):
                    t0 = time.perf_counter()
                    sc1 = dict(l6, loss_type=1, loss_scale=1.0)
                    p1 = be.ba_problem(sc1)
                    s1 = p1.solve(_bo(max_num_iterations=25, gradient_tolerance=0.0))      # (tolerance off: every call runs its full 25 iterations - the upper bound of a local BA's work)
                    q1, x1, _ = p1.get_parameters()
                    p1.close()
                    sc2 = dict(l6, loss_type=0, poses=q1, points=x1)
                    p2 = be.ba_problem(sc2)
                    s2 = p2.solve(_bo(max_num_iterations=25, gradient_tolerance=0.0))
                    p2.get_parameters()
                    p2.close()
                    if rep > 0:
                        walls.append(time.perf_counter() - t0)
                    its = int(s1.num_iterations + s2.num_iterations)
                pl = be.ba_problem(l6)
                pl.solve(_bo(max_num_iterations=25, gradient_tolerance=0.0))
                per_it = []
                for rep in range(5):
                    pl.set_parameters(l6["poses"], l6["points"], None)
                    t0 = time.perf_counter()
                    sl = pl.solve(_bo(max_num_iterations=25, gradient_tolerance=0.0))
                    per_it.append((time.perf_counter() - t0) / max(int(sl.num_iterations), 1))
                pl.close()
                per_it = float(np.median(per_it))
                result["widened"]["local_ba_call"] = {
                    "cams": 6, "obs": int(len(l6["obs_pose"])), "wall_ms": 1e3 * float(np.median(walls)), "lm_iterations": its,
                    "us_per_lm_iteration_resident": 1e6 * per_it, "value": 1.0 / per_it, "unit": "LM iterations/s (one resident handle, a 25-iteration solve; wall_ms = two creates + a "
                    "SOFT_L1 and a TRIVIAL solve of 25 iterations each + read-backs + destroys, median of 5)",
                    "note": "long pair lists (15 image pairs x 334 shared points) are assembled from 32-entry chunks; the reduced system (N = 64) is factorised and solved in one launch"}
                # ONE global bundle adjustment as the mapper issues it (src/sfm/incremental_mapper.cc:893-936, controllers/incremental_mapper.cc:221-243: a new
                # BundleAdjuster per call, TRIVIAL loss, at most 50 iterations): the wall of create + a 50-iteration solve + read-back + destroy, with the
                # create split into its host phases (pp_ba_get_create_profile).  Tolerances off: every call runs its 50 iterations - the upper bound.
                def global_ba_call(sc, reps=5):
                    walls, creates, profs, solves = [], [], [], []
                    for rep in range(reps + 1):
                        t0 = time.perf_counter()
                        pg = be.ba_problem(sc)
                        t1 = time.perf_counter()
                        sg = pg.solve(_bo(max_num_iterations=50, gradient_tolerance=0.0))
                        t2 = time.perf_counter()
                        pg.get_parameters()
                        prof = pg.create_profile()
                        st = pg.structure()
                        pg.close()
                        t3 = time.perf_counter()
                        if rep > 0:      # (the first call pays the plan of the structure; the later ones find it in the plan cache - reported separately)
                            walls.append(t3 - t0); creates.append(t1 - t0); solves.append(t2 - t1); profs.append(prof)
                        else:
                            first = dict(wall_ms=1e3 * (t3 - t0), create_ms=1e3 * (t1 - t0), create_profile_ms=prof)
                    med = int(np.argsort(walls)[len(walls) // 2])
                    return {"cams": int(sc["poses"].shape[0]), "obs": int(len(sc["obs_pose"])), "lm_iterations": int(sg.num_iterations), "wall_ms": 1e3 * walls[med],
                            "create_ms": 1e3 * creates[med], "solve_ms": 1e3 * solves[med], "create_share": creates[med] / walls[med], "create_profile_ms": profs[med],
                            "first_call": first, "chains": st.get("chains"), "chain_steps": st.get("chain_steps"), "reordered": st["reordered"]}
                gsc, _ = synthetic.shuffle_image_ids(synthetic.make_ba_scene(BA_CFG["num_cams"], BA_CFG["num_points"], BA_CFG["track"], seed=0xC0FFEE + 3, model=2, window=40), seed=1)
                g1000 = synthetic.make_ba_scene(1000, 50000, BA_CFG["track"], seed=0xC0FFEE + 3, model=2, window=40)
                result["widened"]["global_ba_call"] = {
                    "note": "create + 50-iteration TRIVIAL solve + read-back + destroy, median of 5 (mapper: a new BundleAdjuster per global BA); create_profile_ms: host "
                            "phases of pp_ba_create (ordering = co-visibility + candidate orders + their chain plans; task_plan = the factorisation's list, cached per tile map)",
                    "dense_headline": global_ba_call(scene), "banded_cfg3_shuffled": global_ba_call(gsc), "banded_1000_images": global_ba_call(g1000)}
                # a photo collection at cfg-3 size: five groups of images joined by four bridge images each (a hub with satellites), ids shuffled - no band in any
                # order; the bridge images are the separators, every group a chain of its own (Ceres' SPARSE_SCHUR ordering for the reference, bundle_adjustment.cc:279-282)
                csc, _ = synthetic.shuffle_image_ids(synthetic.make_ba_scene(BA_CFG["num_cams"], BA_CFG["num_points"], BA_CFG["track"], seed=0xC0FFEE + 55, model=2,
                                                                             clusters=5, bridge=4, topology="star"), seed=2)
                crow = {}
                for cname, env in (("dissected", None), ("one_chain", "band")):
                    if env:
                        os.environ["PPSFM_BA_ORDERING"] = env
                    try:
                        pc = be.ba_problem(csc)
                    finally:
                        if env:
                            del os.environ["PPSFM_BA_ORDERING"]
                    run_ba(pc, csc, CHUNK_ITERS, opts_fn)
                    t0 = time.perf_counter()
                    run_ba(pc, csc, 2 * CHUNK_ITERS, opts_fn)
                    c_s = time.perf_counter() - t0
                    cst = pc.structure()
                    pc.close()
                    crow[cname] = {"value": 2 * CHUNK_ITERS / c_s, "unit": "LM iterations/s", "chains": cst.get("chains"), "chain_steps": cst.get("chain_steps"),
                                   "nonzero_tiles": cst["nnz_used"], "tiles": cst["tiles"], "block_sparse": cst["block_sparse"]}
                result["widened"]["clustered_cfg3"] = dict(crow, cams=BA_CFG["num_cams"], obs=int(len(csc["obs_pose"])), clusters=5, bridge_images=4,
                                                           note="one_chain = the Cuthill-McKee band order alone (PPSFM_BA_ORDERING=band), dissected = as pp_ba_create orders the images")
                lsc = synthetic.make_ba_scene(20, 250, 8, seed=0xC0FFEE + 1, model=2)
                result["widened"]["concurrent_local_ba"] = {
                    "note": "k handles of a configs[0]-sized problem (20 cams / 2k line obs, the size of the mapper's local BA) on ONE GPU, one host thread + "
                            "stream each; value = aggregate over the handles",
                    "cams": 20, "obs": int(len(lsc["obs_pose"])),
                    "rows": concurrent_submodels(be, lsc, 8 * CHUNK_ITERS, counts=(1, 4, 8, 16))}
        except Exception as e:      # the widened rows never take the headline measurement down
            result["widened"] = dict(result.get("widened", {}), error=repr(e))
    # ---- RANSAC leg (every rank runs its share: hypotheses h = rank mod world) -----------------
    rs = None
    if not args.no_ransac and be.has_gpu_rows:
        from privacy_preserving_sfm_amd.device import PoseProblem
        rsc = synthetic.make_ransac_scene(RANSAC_N, outlier_ratio=0.5, noise_px=0.5, seed=0xBADC0DE)
        pp = PoseProblem(rsc["lines"], rsc["points"], rsc["aligned"], device=local)
        H = args.ransac_hyp
        pp.hypotheses(min(H, 8192), rsc["max_error"] ** 2, seed=rank)      # warm-up (also draws+caches nothing of the timed run)
        rep = pp.hypotheses(H, rsc["max_error"] ** 2, seed=1000 + rank)
        dev_s = rep.device_time_s
        if use_dist:
            from privacy_preserving_sfm_amd.distributed import max_over_ranks
            dev_s = max_over_ranks(dev_s, be.device_type)
        if rank == 0:
            pairs = rep.models_scored * RANSAC_N
            rs = {"value": H * world / dev_s, "unit": "hypotheses/s", "hypotheses": H * world, "correspondences": RANSAC_N,
                  "models_scored_rank0": int(rep.models_scored), "best_inliers": int(rep.num_inliers), "device_s": dev_s,
                  "scoring": {"bound": "fp64 valu", "achieved": pairs * SCORE_FLOP_PER_PAIR / rep.device_time_s / 1e12,
                              "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": pairs * SCORE_FLOP_PER_PAIR / rep.device_time_s / 1e12 / FP64_VALU_PEAK_TFLOPS,
                              # issue-slot view: k_score_flat<true> spends 37.25 fp64 VALU issue slots per (model,
                              # correspondence) pair (ISA count of the inner loop, v_rcp_f64 = 3 slots); the chip issues
                              # 256 CU x 4 SIMD x 16 lanes x 2.4 GHz = 39.3e12 lane-slots/s.  Solver time included.
                              "valu_slots_per_pair": SCORE_VALU_SLOTS_PER_PAIR,
                              "valu_issue_frac": pairs * SCORE_VALU_SLOTS_PER_PAIR / rep.device_time_s / 39.3216e12}}
            result["ransac"] = rs
            if world == 1 and not args.no_widened:
                # the shape RegisterNextImage actually runs (src/sfm/incremental_mapper.cc:673-723): ONE image's correspondences, adaptive
                # termination on, the mapper's P6L options - wall of a whole pp_pose_ransac call (host sampler + replay + device chunks)
                from privacy_preserving_sfm_amd.device import ransac_options
                tsc = synthetic.make_ransac_scene(2000, outlier_ratio=0.5, noise_px=0.5, seed=0xBADC0DE + 1)
                ppt = PoseProblem(tsc["lines"], tsc["points"], tsc["aligned"], device=local)
                ro = ransac_options(max_error=tsc["max_error"], seed=0, min_inlier_ratio=0.25, confidence=0.99999, min_num_trials=100, max_num_trials=10000)
                ppt.ransac(ro)      # warm-up
                walls = []
                for _ in range(5):
                    t0 = time.perf_counter()
                    trep, _ = ppt.ransac(ro)
                    walls.append(time.perf_counter() - t0)
                ppt.close()
                result.setdefault("widened", {})["pose_ransac_typical"] = {
                    "correspondences": 2000, "outlier_ratio": 0.5, "num_trials": int(trep.num_trials), "num_inliers": int(trep.num_inliers),
                    "wall_ms": 1e3 * float(np.median(walls)), "value": float(trep.num_trials / np.median(walls)), "unit": "trials/s (wall of a whole pp_pose_ransac call, "
                    "adaptive termination, median of 5)", "note": "EstimateAbsolutePoseFromLines' RANSAC as RegisterNextImage calls it (min 100 / max 10000 trials, "
                    "confidence 0.99999, min inlier ratio 0.25)"}
        pp.close()
    # ---- BASELINE configs[4] taken literally: world/2 sub-models over `world` GPUs (4 over 8 at N = 8), (a) as whole-GPU replicas on the
    # first world/2 ranks, (b) point-sharded over groups of two ranks with the RCCL exchange inside the library.  Runs after the headline
    # has been measured and under a deadline: it is the first code that needs the multi-rank collectives to work.
    if use_dist and gsize == 1 and world % 2 == 0 and not args.no_widened:
        row = cfg5_literal(be, args, rank, world, full_scene)
        if rank == 0:
            result.setdefault("widened", {})["cfg5_literal"] = row
        if row.get("abandoned"):      # a stuck collective: no later collective (not even the final barrier) can be trusted - print and leave
            if rank == 0:
                print(json.dumps(result), flush=True)
            os._exit(0)
    if comm is not None:
        pb.set_communicator(None)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and be.has_gpu_rows:
            from privacy_preserving_sfm_amd.device import ba_options
            rsc = synthetic.make_ransac_scene(RANSAC_N, outlier_ratio=0.5, noise_px=0.5, seed=0xBADC0DE)
            base_scene = make_scene(0)
            pb.set_parameters(base_scene["poses"], base_scene["points"], None)      # rank 0's scene IS base_scene (same seed)
            pb.solve(ba_options(max_num_iterations=3))
            dposes, dpoints, _ = pb.get_parameters()
            result["cpu_baseline"] = cpu_baseline(base_scene, rsc, (dposes, dpoints))
            result["speedup_vs_cpu_baseline"] = {"ba": result["value"] / result["cpu_baseline"]["value"],
                                                 "ransac": (rs["value"] / result["cpu_baseline"]["ransac"]["value"]) if rs else None}
        print(json.dumps(result), flush=True)
    pb.close()
    if comm is not None:
        comm.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return result


LINSOLVE_NAMES = {0: "cholesky (one launch per block column)", 1: "cholesky (one launch: chain workgroup + task list)", 2: "block-sparse cholesky", 4: "one-launch small-problem solver (LDS Cholesky)",
                  3: "pcg (implicit Schur complement, block-Jacobi)"}


def cfg5_literal(be, args, rank, world, full_scene):
    """BASELINE configs[4]: m = world/2 independent sub-models on `world` GPUs, two ways; every rank calls this (collective).
    A watchdog ends the row after --cfg5-timeout seconds on every rank (a collective that never completes cannot be caught as an
    exception): the row then says so and the bench line is printed without its numbers."""
    import threading
    from privacy_preserving_sfm_amd.distributed import make_submodel_groups, shard_scene_by_points, submodel_layout
    m = world // 2
    row = {"submodels": m, "gpus": world, "literal_configs4": bool(world == 8), "steps": args.steps}
    state = {"done": False}

    def body():
        be.bind_thread()
        # (a) whole-GPU replicas: sub-model i on rank i < m, the other ranks idle
        active = rank < m
        sc = make_scene(rank) if active else None
        pb = be.ba_problem(sc) if active else None
        el, _ = timed_ba(be, True, pb, sc, args.warmup, args.steps, active=active)
        if pb is not None:
            pb.close()
        row["replicas"] = {"value": m * args.steps / el, "unit": "LM iterations/s", "ms_per_step": 1e3 * el / args.steps, "busy_gpus": m,
                           "exchange": "none"}
        # (b) every sub-model point-sharded over a group of two neighbouring ranks, exchange = the library's RCCL collectives
        model_id, grank, gsize = submodel_layout(rank, world, m)
        groups = make_submodel_groups(world, m)
        comm = be.communicator(groups[model_id])
        sh = shard_scene_by_points(make_scene(model_id), grank, gsize)
        pb = be.ba_problem(sh)
        pb.set_communicator(comm)
        el, _ = timed_ba(be, True, pb, sh, args.warmup, args.steps)
        pb.set_communicator(None)
        pb.close()
        comm.close()
        row["sharded"] = {"value": m * args.steps / el, "unit": "LM iterations/s", "ms_per_step": 1e3 * el / args.steps, "busy_gpus": world,
                          "ranks_per_submodel": gsize,
                          "exchange": "RCCL all-reduce of U/g_c (42 doubles per pose), the packed lower triangle of S (36 MB) and the scalars, per LM iteration"}
        # (c) the iterative regime sharded: one 1100-image sub-model (ITERATIVE_SCHUR by its image count) per group of two ranks.  Per CG
        # iteration the group sums the Schur product - 6 C doubles = 53 KB - instead of the 36 MB triangle of the direct solver per LM iteration
        from privacy_preserving_sfm_amd import synthetic
        big = synthetic.make_ba_scene(1100, 22000, 8, seed=0xC0FFEE + 5 + 101 * model_id, model=2)
        comm = be.communicator(groups[model_id])
        shb = shard_scene_by_points(big, grank, gsize)
        pb = be.ba_problem(shb)
        pb.set_communicator(comm)
        el, _ = timed_ba(be, True, pb, shb, args.warmup, args.steps)
        pb.set_communicator(None)
        pb.close()
        comm.close()
        row["sharded_iterative_1100"] = {"value": m * args.steps / el, "unit": "LM iterations/s", "ms_per_step": 1e3 * el / args.steps, "cams": 1100,
                                         "obs": int(len(big["obs_pose"])), "ranks_per_submodel": gsize,
                                         "exchange": "RCCL all-reduce of the Schur product (6 C doubles = 53 KB) per CG iteration, of the diagonal blocks + rhs per LM iteration"}
        # (d) a SEQUENCE sub-model (500 images / 200k observations, every point inside a 40-image window, ids shuffled) point-sharded over the same groups with the
        # group's union co-visibility at create (one all-reduce MAX of C x C bytes; pp_ba_problem_desc::covisibility): every rank takes the unsharded problem's
        # nested-dissection order and tile map, the exchanged system is factorised block-sparse by several chains instead of the dense 47-step chain
        from privacy_preserving_sfm_amd.distributed import group_covisibility, with_group_structure
        seq, _ = synthetic.shuffle_image_ids(synthetic.make_ba_scene(BA_CFG["num_cams"], BA_CFG["num_points"], BA_CFG["track"], seed=0xC0FFEE + 3 + 101 * model_id, model=2,
                                                                     window=min(40, max(BA_CFG["track"], BA_CFG["num_cams"] // 2))), seed=1)
        shq = shard_scene_by_points(seq, grank, gsize)
        union = group_covisibility(shq, groups[model_id], be.device_type)
        comm = be.communicator(groups[model_id])
        pb = be.ba_problem(with_group_structure(shq, union))
        pb.set_communicator(comm)
        st = pb.structure() if hasattr(pb, "structure") else {}
        el, _ = timed_ba(be, True, pb, shq, args.warmup, args.steps)
        pb.set_communicator(None)
        pb.close()
        comm.close()
        row["sharded_banded"] = {"value": m * args.steps / el, "unit": "LM iterations/s", "ms_per_step": 1e3 * el / args.steps, "cams": BA_CFG["num_cams"], "obs": int(len(seq["obs_pose"])),
                                 "ranks_per_submodel": gsize, "chains": st.get("chains"), "chain_steps": st.get("chain_steps"), "block_sparse": st.get("block_sparse"),
                                 "exchange": "create: all-reduce MAX of the C x C co-visibility bytes; per LM iteration as `sharded`"}
        state["done"] = True

    err = []

    def guarded():
        try:
            body()
        except Exception as e:      # noqa
            err.append(repr(e))
            state["done"] = True

    t = threading.Thread(target=guarded, daemon=True)
    t.start()
    t.join(timeout=args.cfg5_timeout)
    if err:      # a rank that failed left its peers inside a collective: they run into their deadline, this one leaves the same way
        row["error"] = err[0]
        row["abandoned"] = True
    elif not state["done"]:
        # a collective is stuck: nothing after this row could run either.  Rank 0 prints what it has; every rank leaves.
        row["error"] = "abandoned after %.0f s (a collective of the sharded run did not complete)" % args.cfg5_timeout
        row["abandoned"] = True
    return row


if __name__ == "__main__":
    main()
