# A/B of the working tree's cholesky.hip against a baseline copy of it (tools/_ab/cholesky_base.inc: `git show <rev>:.../cholesky.hip`,
# git-ignored) ON ONE BOX: the task-mode factorisation + back substitution (T = 47, production speed) timed alternately.   gpurun -- bash tools/ab_rev.sh [T]
# make the baseline first:  mkdir -p tools/_ab && git show <rev>:privacy_preserving_sfm_amd/csrc/cholesky.hip > tools/_ab/cholesky_base.inc
T=${1:-47}
B='-DPP_CHOL_SRC="_ab/cholesky_base.inc"'
hipcc -O3 -std=c++17 --offload-arch=gfx950 -w -Iprivacy_preserving_sfm_amd/csrc -DPP_CHOL_NO_STAMPS "$B" tools/chol_task_trace.hip privacy_preserving_sfm_amd/csrc/capi_misc.hip privacy_preserving_sfm_amd/csrc/resource_pool.hip -o /tmp/tt_base || exit 1
hipcc -O3 -std=c++17 --offload-arch=gfx950 -w -DPP_CHOL_NO_STAMPS tools/chol_task_trace.hip privacy_preserving_sfm_amd/csrc/capi_misc.hip privacy_preserving_sfm_amd/csrc/resource_pool.hip -o /tmp/tt_new || exit 1
for round in 1 2 3; do
  for v in base new; do echo "$v: $(/tmp/tt_$v $T n | grep '^rep' | tr '\n' ' ')"; done
done
echo "---- new: wait of the spare wavefronts / step length"
/tmp/tt_new $T n | grep -A50 "chain, per step: wait" | head -52
