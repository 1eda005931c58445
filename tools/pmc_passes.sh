# GPU-side: rocprofv3 counter passes behind profiles/rNN_*_pmc.json (one counter group per pass: --pmc is never combined
# with a trace domain other than the kernel dispatch records rocprofv3 writes by itself).
#   gpurun -- bash tools/pmc_passes.sh r02
# Passes:  ba workload (3 LM iterations at 500 cams / 200k obs): MFMA/VALU instruction counts + busy cycles, L2 hit/miss +
# fabric requests, FETCH_SIZE, WRITE_SIZE;  ransac workload (16384 hypotheses x 50k correspondences): VALU counts;
# k1 at 2M observations (beyond the 256 MiB Infinity Cache): FETCH_SIZE, WRITE_SIZE.
set -x
V=${1:-r02}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${V}_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_available.txt 2>&1
pass() {   # name, workload, counters...
  name=$1; wl=$2; shift 2
  rm -rf /tmp/pmc_$name
  timeout 600 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_$name -- python $R/tools/profile_workload.py $wl > /tmp/pmc_$name.log 2>&1
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then PYTHONPATH=$R python $R/tools/pmc_summary.py --json "$f" > $OUT/$name.json; else tail -5 /tmp/pmc_$name.log > $OUT/$name.err; fi
}
pass ba_sq_insts ba SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_SALU SQ_INSTS_LDS
pass ba_sq_cycles ba SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
pass ba_tcc ba TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
pass ba_fetch ba FETCH_SIZE
pass ba_write ba WRITE_SIZE
pass pcg_fetch pcg FETCH_SIZE
pass pcg_write pcg WRITE_SIZE
pass ransac_sq_insts ransac SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass k1big_fetch k1big FETCH_SIZE
pass k1big_write k1big WRITE_SIZE
pass k1_fetch k1 FETCH_SIZE
pass k1_write k1 WRITE_SIZE
ls -la $OUT
