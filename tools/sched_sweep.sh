# the two knobs of the task list's priority on ONE box: PPSFM_CHOL_SLOPE (far updates deferred by this many steps per super-column) and
# PPSFM_CHOL_WHOLE_FROM (whole super-tiles from this many super-columns right of the front); production speed (no stamps), the three
# repetitions of factorisation + back substitution per setting.   gpurun -- bash tools/sched_sweep.sh [T ...]
hipcc -O3 -std=c++17 --offload-arch=gfx950 -w -DPP_CHOL_NO_STAMPS tools/chol_task_trace.hip privacy_preserving_sfm_amd/csrc/capi_misc.hip -o /tmp/tt_nostamp || exit 1
for round in 1 2; do
for T in ${@:-47}; do
for cfg in "0.5 5" "0.3 0" "0.3 -2" "0.3 2" "0.4 0" "0.2 0"; do
  set -- $cfg
  W=$(python3 -c "T=$T; print(max(2, round(16.0 - 0.19 * T) + $2))")
  echo "T $T slope $1 whole_from $W: $(PPSFM_CHOL_SLOPE=$1 PPSFM_CHOL_WHOLE_FROM=$W /tmp/tt_nostamp $T n | grep '^rep' | tr '\n' ' ')"
done
done
done
