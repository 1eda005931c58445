# GPU-side: the counter passes of tools/pmc_passes.sh for the BANDED workload (tools/profile_workload.py band: cfg-3 size, window 40, images dissected, four
# chain workgroups) - one counter group per pass, never combined with a trace domain.    gpurun -- bash tools/pmc_band.sh r04_band
set -x
V=${1:-r04_band}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${V}_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {   # name, counters...
  name=$1; shift
  rm -rf /tmp/pmc_$name
  timeout 600 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_$name -- python $R/tools/profile_workload.py band > /tmp/pmc_$name.log 2>&1
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then PYTHONPATH=$R python $R/tools/pmc_summary.py --json "$f" > $OUT/$name.json; else tail -5 /tmp/pmc_$name.log > $OUT/$name.err; fi
}
pass band_sq_insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_SALU SQ_INSTS_LDS
pass band_sq_cycles SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
pass band_tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
pass band_fetch FETCH_SIZE
pass band_write WRITE_SIZE
ls -la $OUT
