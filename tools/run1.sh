set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02a
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 2>&1 | tail -40 > $R/gpurun_out/r02a/pytest_gpu.txt
cat $R/gpurun_out/r02a/pytest_gpu.txt
timeout 600 python bench.py > $R/gpurun_out/r02a/bench.json 2> $R/gpurun_out/r02a/bench.err
tail -3 $R/gpurun_out/r02a/bench.err; head -c 3000 $R/gpurun_out/r02a/bench.json
bash tools/pmc_passes.sh r02a
