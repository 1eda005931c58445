# A/B of two builds of the task-mode factorisation on the SAME box (box-to-box spread is a few per cent): tools/ab_chain.sh "<flags A>" "<flags B>"
# prints the three repetitions of the full solve (factorisation + back substitution, T = 47) per build, production speed (no stamps)
for v in A B; do
  if [ $v = A ]; then F="$1"; else F="$2"; fi
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -w -DPP_CHOL_NO_STAMPS $F tools/chol_task_trace.hip privacy_preserving_sfm_amd/csrc/capi_misc.hip privacy_preserving_sfm_amd/csrc/resource_pool.hip -o /tmp/tt_$v || exit 1
done
for round in 1 2 3; do
  for v in A B; do echo "build $v: $(/tmp/tt_$v 47 | grep '^rep' | tr '\n' ' ')"; done
done
