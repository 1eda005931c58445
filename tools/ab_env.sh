# A/B of ONE build of the task-mode factorisation under an environment switch, alternately on one box:   gpurun -- bash tools/ab_env.sh PPSFM_CHOL_QUIET_XCD [T]
# prints the three repetitions of factorisation + back substitution (production speed, no stamps) per setting
VAR=$1; T=${2:-47}
hipcc -O3 -std=c++17 --offload-arch=gfx950 -w -DPP_CHOL_NO_STAMPS tools/chol_task_trace.hip privacy_preserving_sfm_amd/csrc/capi_misc.hip privacy_preserving_sfm_amd/csrc/resource_pool.hip -o /tmp/tt_env || exit 1
for round in 1 2 3; do
  for v in 0 1; do echo "$VAR=$v: $(env $VAR=$v /tmp/tt_env $T n | grep '^rep' | tr '\n' ' ')"; done
done
