# GPU-side command list behind profiles/rNN_vMM_*: tests, smoke, bench line, rocprofv3 kernel stats of the profile workload and of
# bench.py itself (without the 2M-observation K1 launches: the stats file averages over all launches of a kernel name), Cholesky timings,
# per-config baseline table.     gpurun -- bash tools/refresh_profiles.sh r02_v21
set -x
V=$1
mkdir -p gpurun_out/$V
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/$V/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 > gpurun_out/$V/smoke.txt
timeout 600 python bench.py > gpurun_out/$V/bench.json 2> gpurun_out/$V/bench.err
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_w -- python $R/tools/profile_workload.py all > /tmp/prof_w.log 2>&1
cp $(find /tmp/prof_w -name "*kernel_stats.csv" | head -1) $R/gpurun_out/$V/kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_n -- python $R/tools/profile_workload.py band > /tmp/prof_n.log 2>&1
cp $(find /tmp/prof_n -name "*kernel_stats.csv" | head -1) $R/gpurun_out/$V/band_kernel_stats.csv; tail -1 /tmp/prof_n.log > $R/gpurun_out/$V/band_workload.txt
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -- python $R/bench.py --no-beyond-l3 --no-cpu-baseline > /tmp/prof_b.log 2>&1
cp $(find /tmp/prof_b -name "*kernel_stats.csv" | head -1) $R/gpurun_out/$V/bench_py_kernel_stats.csv
tail -1 /tmp/prof_b.log > $R/gpurun_out/$V/bench_under_rocprof.json
cd $R; cat gpurun_out/$V/pytest_gpu.txt gpurun_out/$V/smoke.txt; head -c 600 gpurun_out/$V/bench.json
PYTHONPATH=$R python $R/tools/chol_time.py > $R/gpurun_out/$V/chol_time.txt 2>&1; cat $R/gpurun_out/$V/chol_time.txt
PYTHONPATH=$R timeout 600 python $R/tools/baseline_table.py > $R/gpurun_out/$V/baseline_table.md 2> $R/gpurun_out/$V/baseline_table.err; cat $R/gpurun_out/$V/baseline_table.md
