# GPU-side: PMC passes over the chain workgroup ALONE in both Cholesky modes (k_column_step grid 1 in chol_phase_bench,
# k_cholesky_tasks grid 1 in chol_task_trace ... iso): instruction cache, issue and wait counters per dispatch.
R=$GRAFT_REPO_ROOT
hipcc -O3 -std=c++17 --offload-arch=gfx950 -w tools/chol_task_trace.hip privacy_preserving_sfm_amd/csrc/capi_misc.hip privacy_preserving_sfm_amd/csrc/resource_pool.hip -o /tmp/chol_task_trace || exit 1
hipcc -O3 -std=c++17 --offload-arch=gfx950 -w tools/chol_phase_bench.hip privacy_preserving_sfm_amd/csrc/capi_misc.hip privacy_preserving_sfm_amd/csrc/resource_pool.hip -o /tmp/chol_phase || exit 1
cd /tmp && export TMPDIR=/tmp
pass() {  # name, counters
  name=$1; shift
  for b in task phase; do
    rm -rf /tmp/cp_${name}_$b
    if [ $b = task ]; then cmd="/tmp/chol_task_trace 47 n iso"; else cmd="/tmp/chol_phase"; fi
    timeout 300 rocprofv3 --pmc "$@" --output-format csv -d /tmp/cp_${name}_$b -- $cmd > /tmp/cp_${name}_$b.log 2>&1
    f=$(find /tmp/cp_${name}_$b -name "*counter_collection.csv" | head -1)
    echo "== $name $b"
    python3 - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.OrderedDict()
for r in rows:
    k=(r['Kernel_Name'].split('(')[0][-30:], r['Grid_Size'], r['Counter_Name'])
    agg.setdefault(k,[]).append(float(r['Counter_Value']))
for k,v in agg.items():
    if 'column_step' in k[0] or 'cholesky_tasks' in k[0]:
        if k[1] in ('1024',): print(k, 'n=%d mean=%.0f min=%.0f max=%.0f' % (len(v), sum(v)/len(v), min(v), max(v)))
PY
  done
}
pass icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
pass issue SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_IFETCH
pass other SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
