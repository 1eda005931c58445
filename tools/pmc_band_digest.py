"""Digest of tools/pmc_band.sh's passes (gpurun_out/<V>_pmc/band_*.json) into one tracked file:   python tools/pmc_band_digest.py gpurun_out/r06_band_pmc profiles/r06_band_pmc.json
Per kernel of the banded LM iteration (cfg-3 size, window 40, four chains): time under the counters, FETCH / WRITE in MB (x1: gathers), L2 hit rate, fabric
requests, MFMA utilisation, share of wave-cycles waiting."""
import json, os, sys
src, dst = sys.argv[1], sys.argv[2]
def load(n):
    p = os.path.join(src, n + ".json")
    return json.load(open(p)) if os.path.exists(p) else {}
fetch, write, tcc, cyc, insts = (load("band_" + n) for n in ("fetch", "write", "tcc", "sq_cycles", "sq_insts"))
def tot(d, k, c): return sum(v.get(c, 0.0) for v in d.get(k, {}).values())
out = {"source": src, "workload": "tools/profile_workload.py band: 500 images / 200k observations, every point inside a 40-image window, images dissected (4 chain workgroups)"}
for k in ("k_schur_self_chunks", "k_schur_chunk_reduce", "k_prepare<true>", "k_prepare<false>", "k_cholesky_tasks", "k_potrf64", "k_backsub_all", "k_line_eval<1, false, false>", "k_reduce", "k_step_points<false>", "k_norms_partial"):
    names = [n for n in fetch if n.startswith(k.split("<")[0])] if k not in fetch else [k]
    for n in names:
        if n in out or n not in fetch: continue
        hit, miss = tot(tcc, n, "TCC_HIT_sum"), tot(tcc, n, "TCC_MISS_sum")
        busy, waves_c = tot(cyc, n, "SQ_VALU_MFMA_BUSY_CYCLES"), tot(cyc, n, "SQ_WAVE_CYCLES")
        ns = tot(cyc, n, "mean_ns_under_pmc")
        out[n] = {"ns_under_pmc": ns, "fetch_MB": tot(fetch, n, "FETCH_SIZE") / 1024.0, "write_MB": tot(write, n, "WRITE_SIZE") / 1024.0,
                  "l2_hit_rate": hit / max(hit + miss, 1.0), "fabric_read_requests": tot(tcc, n, "TCC_EA0_RDREQ_sum"), "fabric_write_requests": tot(tcc, n, "TCC_EA0_WRREQ_sum"),
                  "mfma_util": busy / max(ns * 2.4 * 1024, 1.0), "wait_any_share": tot(cyc, n, "SQ_WAIT_ANY") / max(waves_c, 1.0), "valu_insts": tot(insts, n, "SQ_INSTS_VALU"), "waves": tot(insts, n, "SQ_WAVES")}
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
print(json.dumps({k: {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in out.items() if isinstance(v, dict)}, indent=1))
