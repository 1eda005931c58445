"""Digest of the counter passes of tools/pmc_passes.sh (gpurun_out/<V>_pmc/*.json) into one tracked file:
    python tools/pmc_digest.py gpurun_out/r02a_pmc profiles/r02_pmc.json
Per kernel of interest: the raw per-launch counter means and the derived figures quoted in DESIGN.md / bench.py
(MFMA utilisation, flops from the MFMA op counter, L2 hit rate, fabric bytes, VALU issue share).
Units (MI355X_MICROARCH.md, and calibrated on the one-workgroup k_column_step launch: 424 MFMAs -> 27136 busy cycles):
  SQ_INSTS_VALU_MFMA_MOPS_F64   512 flop each (a v_mfma_f64_16x16x4 = 2048 flop counts 4)
  SQ_VALU_MFMA_BUSY_CYCLES      shader cycles a SIMD's matrix pipe is busy, summed over SIMDs (64 per v_mfma_f64_16x16x4)
  FETCH_SIZE / WRITE_SIZE       KB; FETCH_SIZE x2 for wide coalesced streaming reads on gfx950 (K1), x1 for the 8/16-byte gathers
  TCC_EA0_RDREQ/WRREQ           requests of the L2s to the fabric (64 B each for the request mix of these kernels)"""
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
CLK_GHZ, SIMDS = 2.4, 1024


def load(name):
    p = os.path.join(src, name + ".json")
    return json.load(open(p)) if os.path.exists(p) else {}


def total(per_grid, counter):
    """sum over the grid sizes of a kernel name of (mean per launch), i.e. per ONE factorisation / iteration when every
    grid size occurs once per iteration (k_column_step: 46 launches of 46 different grids)"""
    return sum(v.get(counter, 0.0) for v in per_grid.values())


out = {"source": src, "clock_ghz": CLK_GHZ, "simds": SIMDS}
insts, cyc, tcc, fetch, write = load("ba_sq_insts"), load("ba_sq_cycles"), load("ba_tcc"), load("ba_fetch"), load("ba_write")
# ---- Cholesky: k_potrf64 + k_cholesky_tasks (or 46 x k_column_step) + k_backsub_all --------------------------------------------------------------
chol = {}
for k in ("k_potrf64", "k_column_step", "k_cholesky_tasks", "k_backsub_all", "k_backsub_prepare", "k_backsub_pairs"):
    if k not in insts:
        continue
    ns = total(cyc.get(k, {}), "mean_ns_under_pmc")
    busy = total(cyc.get(k, {}), "SQ_VALU_MFMA_BUSY_CYCLES")
    chol[k] = {"launches_per_factorisation": len(insts[k]), "ns_under_pmc": ns,
               "mfma_insts": total(insts[k], "SQ_INSTS_MFMA"), "mfma_mops_f64": total(insts[k], "SQ_INSTS_VALU_MFMA_MOPS_F64"),
               "flop": 512.0 * total(insts[k], "SQ_INSTS_VALU_MFMA_MOPS_F64"), "valu_insts": total(insts[k], "SQ_INSTS_VALU"),
               "waves": total(insts[k], "SQ_WAVES"), "mfma_busy_cycles": busy,
               "mfma_util": busy / (ns * CLK_GHZ * SIMDS) if ns else None,
               "wave_cycles": total(cyc.get(k, {}), "SQ_WAVE_CYCLES"), "wait_any": total(cyc.get(k, {}), "SQ_WAIT_ANY"),
               "wait_inst_any": total(cyc.get(k, {}), "SQ_WAIT_INST_ANY")}
if chol:
    ns = sum(v["ns_under_pmc"] for v in chol.values())
    busy = sum(v["mfma_busy_cycles"] for v in chol.values())
    flop = sum(v["flop"] for v in chol.values())
    chol["whole_solve"] = {"ns_under_pmc": ns, "flop_by_counter": flop, "tflops_by_counter": flop / ns / 1e3 if ns else None,
                           "mfma_util": busy / (ns * CLK_GHZ * SIMDS) if ns else None,
                           "note": "n = 3001 padded to 3008: n^3/3 + 2 n^2 = 9.02 GFLOP algorithmic; the counter also sees the explicit 64x64 inverses, the "
                                   "redundant prep products and the padding"}
    # the per-launch profile of the column steps (grid size -> duration, MFMA utilisation): early launches are bulk-bound, late ones chain-bound
    steps = []
    for g in sorted(cyc.get("k_column_step", {}), key=int, reverse=True):
        c = cyc["k_column_step"][g]
        steps.append({"grid": int(g), "ns": c.get("mean_ns_under_pmc"), "mfma_util": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (c["mean_ns_under_pmc"] * CLK_GHZ * SIMDS)})
    chol["column_steps"] = steps
out["cholesky"] = chol
# ---- Schur gather ----------------------------------------------------------------------------------------------------------
for k in ("k_schur_blocks", "k_prepare<true>", "k_step_points", "k_obs_prepare", "k_reduce", "k_point_prepare<true>", "k_backsub_points", "k_model_cost_apply", "k_norms_partial"):
    if k not in tcc:
        continue
    g = next(iter(tcc[k]))
    t = tcc[k][g]
    d = {"grid": int(g), "ns_under_pmc": t.get("mean_ns_under_pmc"), "l2_hits": t.get("TCC_HIT_sum"), "l2_misses": t.get("TCC_MISS_sum"),
         "l2_hit_rate": t.get("TCC_HIT_sum", 0.0) / max(t.get("TCC_HIT_sum", 0.0) + t.get("TCC_MISS_sum", 0.0), 1.0),
         "fabric_read_requests": t.get("TCC_EA0_RDREQ_sum"), "fabric_write_requests": t.get("TCC_EA0_WRREQ_sum"),
         "fabric_read_bytes_64B": 64.0 * t.get("TCC_EA0_RDREQ_sum", 0.0), "fabric_write_bytes_64B": 64.0 * t.get("TCC_EA0_WRREQ_sum", 0.0)}
    if k in fetch:
        d["FETCH_SIZE_KB"] = next(iter(fetch[k].values())).get("FETCH_SIZE")
    if k in write:
        d["WRITE_SIZE_KB"] = next(iter(write[k].values())).get("WRITE_SIZE")
    if k in insts:
        i = next(iter(insts[k].values()))
        d.update(valu_insts=i.get("SQ_INSTS_VALU"), waves=i.get("SQ_WAVES"), lds_insts=i.get("SQ_INSTS_LDS"))
    out[k] = d
# ---- K1 at cfg 3 (200k obs) and beyond the Infinity Cache (2M obs) ---------------------------------------------------------
k1 = {}
for tag, obs in (("k1", 200000), ("k1big", 2000000)):
    f, w = load(tag + "_fetch"), load(tag + "_write")
    name = "k_line_eval<1, false, false>"
    if name in f and name in w:
        fk = next(iter(f[name].values())); wk = next(iter(w[name].values()))
        fetch_b = 2.0 * 1024.0 * fk["FETCH_SIZE"]; write_b = 1024.0 * wk["WRITE_SIZE"]
        k1[tag] = {"observations": obs, "FETCH_SIZE_KB": fk["FETCH_SIZE"], "WRITE_SIZE_KB": wk["WRITE_SIZE"], "fetch_bytes_x2": fetch_b, "write_bytes": write_b,
                   "traffic_bytes_per_launch": fetch_b + write_b, "algorithmic_bytes_per_launch": 220.0 * obs,
                   "traffic_over_algorithmic": (fetch_b + write_b) / (220.0 * obs), "launches": fk["launches"]}
out["k_line_eval"] = k1
# ---- matrix-free PCG (1100 images / 176k observations): S v from the 192-byte records ------------------------------------------------
pf, pw = load("pcg_fetch"), load("pcg_write")
pcg = {}
PCG_OBS, PCG_POINTS, PCG_IMAGES = 176000, 22000, 1100
alg = {"k_pcg_points": PCG_OBS * (144.0 + 48.0) + PCG_POINTS * 24.0,       # record bytes 48..191 + the image's 6 entries of v (gathered) per observation; a_p out
       "k_pcg_images": PCG_OBS * (144.0 + 24.0) + PCG_IMAGES * 2 * 48.0}    # record bytes 0..143 + a_p (gathered) per observation; v in, S v out per image
for k in ("k_pcg_points", "k_pcg_images", "k_pcg_vec", "k_pcg_block_inverse"):
    if k in pf and k in pw:
        fk = next(iter(pf[k].values())); wk = next(iter(pw[k].values()))
        d = {"launches": fk["launches"], "ns_under_pmc": fk["mean_ns_under_pmc"], "FETCH_SIZE_KB": fk["FETCH_SIZE"], "WRITE_SIZE_KB": wk["WRITE_SIZE"],
             "traffic_bytes_per_launch": 1024.0 * (fk["FETCH_SIZE"] + wk["WRITE_SIZE"])}
        if k in alg:
            d["algorithmic_bytes_per_launch"] = alg[k]
            d["algorithmic_GBps"] = alg[k] / fk["mean_ns_under_pmc"]
            d["traffic_over_algorithmic"] = d["traffic_bytes_per_launch"] / alg[k]
            d["note"] = "the records (33.8 MB) and the vectors stay in the L2s / Infinity Cache between the two kernels of a product: the counters see less than the algorithmic bytes"
        pcg[k] = d
out["pcg_1100_images"] = pcg
# ---- RANSAC scoring --------------------------------------------------------------------------------------------------------
r = load("ransac_sq_insts")
for k in ("k_score_flat<true>", "k_p6l"):
    if k in r:
        v = next(iter(r[k].values()))
        d = dict(v)
        ns = v.get("mean_ns_under_pmc")
        # a wave-level VALU instruction occupies its SIMD's fp64 pipe for 4 cycles (16 lanes x 4); v_rcp_f64 and friends longer
        d["valu_issue_share_lower_bound"] = 4.0 * v.get("SQ_INSTS_VALU", 0.0) / (ns * CLK_GHZ * SIMDS) if ns else None
        d["active_inst_valu_over_wave_cycles"] = v.get("SQ_ACTIVE_INST_VALU", 0.0) / max(v.get("SQ_WAVE_CYCLES", 1.0), 1.0)
        out[k] = d
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
print("wrote", dst)
for k in ("cholesky", "k_schur_blocks", "k_line_eval", "pcg_1100_images", "k_score_flat<true>"):
    v = out.get(k)
    if k == "cholesky" and v:
        print(k, {kk: vv for kk, vv in v.get("whole_solve", {}).items() if kk != "note"})
        for name in ("k_column_step", "k_cholesky_tasks"):
            if name in v:
                print(" ", name, {kk: v[name][kk] for kk in ("mfma_util", "ns_under_pmc", "flop", "valu_insts", "wave_cycles", "wait_any")})
    elif v:
        print(k, v)
