# the knobs of the task list on ONE box: PPSFM_CHOL_SLOPE (far updates deferred by this many steps per super-column),
# PPSFM_CHOL_WHOLE_FROM (whole super-tiles from this many super-columns right of the front) and PPSFM_CHOL_TWO_PANELS (0: one panel per update task); production speed (no stamps), the three
# repetitions of factorisation + back substitution per setting.   gpurun -- bash tools/sched_sweep.sh [T ...]
hipcc -O3 -std=c++17 --offload-arch=gfx950 -w -DPP_CHOL_NO_STAMPS tools/chol_task_trace.hip privacy_preserving_sfm_amd/csrc/capi_misc.hip privacy_preserving_sfm_amd/csrc/resource_pool.hip -o /tmp/tt_nostamp || exit 1
for round in 1 2; do
for T in ${@:-47}; do
for cfg in "0.3 3 1" "0.3 2 1" "0.3 4 1" "0.3 6 1" "0.2 3 1" "0.4 3 1" "0.3 7 0" "0.5 12 0"; do      # the last one: round 2's schedule
  set -- $cfg
  echo "T $T slope $1 whole_from $2 two_panels $3: $(PPSFM_CHOL_SLOPE=$1 PPSFM_CHOL_WHOLE_FROM=$2 PPSFM_CHOL_TWO_PANELS=$3 /tmp/tt_nostamp $T n | grep '^rep' | tr '\n' ' ')"
done
done
done
