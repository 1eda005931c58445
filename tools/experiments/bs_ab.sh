# A/B of the back substitution's placement (PPSFM_BACKSUB_LOCAL) on one box:  gpurun -- bash tools/experiments/bs_ab.sh
export PYTHONPATH=$PWD
for L in 1 0 1 0; do echo LOCAL=$L; PPSFM_BACKSUB_LOCAL=$L timeout 120 python tools/chol_time.py 3000 2>&1 | tail -2; done
