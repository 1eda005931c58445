# A/B of two builds of the per-column chain's phase stamps (tools/chol_phase_bench.hip) on the SAME box: tools/ab_phase.sh "<flags A>" "<flags B>"
for v in A B; do
  if [ $v = A ]; then F="$1"; else F="$2"; fi
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -w $F tools/chol_phase_bench.hip privacy_preserving_sfm_amd/csrc/capi_misc.hip privacy_preserving_sfm_amd/csrc/resource_pool.hip -o /tmp/ph_$v || exit 1
done
for v in A B; do echo "== build $v"; /tmp/ph_$v | grep "chain workgroup" | head -6; done
