# build + run tools/chol_task_trace.hip (and the column-mode phase bench) on the GPU box; extra args go to hipcc (-D experiment switches)
hipcc -O3 -std=c++17 --offload-arch=gfx950 -w "$@" tools/chol_task_trace.hip privacy_preserving_sfm_amd/csrc/capi_misc.hip privacy_preserving_sfm_amd/csrc/resource_pool.hip -o /tmp/chol_task_trace || exit 1
hipcc -O3 -std=c++17 --offload-arch=gfx950 -w "$@" tools/chol_phase_bench.hip privacy_preserving_sfm_amd/csrc/capi_misc.hip privacy_preserving_sfm_amd/csrc/resource_pool.hip -o /tmp/chol_phase || exit 1
timeout 120 /tmp/chol_task_trace 47 c iso
echo ==== column mode; timeout 120 /tmp/chol_phase
