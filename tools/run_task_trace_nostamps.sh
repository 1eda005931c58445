# the chain of the one-launch Cholesky at production speed (PP_CHOL_NO_STAMPS: the trace variables stay, the per-phase stamps go): per step the wait
# of the spare wavefronts for the next inputs after the last panel and the step length.   gpurun -- bash tools/run_task_trace_nostamps.sh
hipcc -O3 -std=c++17 --offload-arch=gfx950 -w -DPP_CHOL_NO_STAMPS tools/chol_task_trace.hip privacy_preserving_sfm_amd/csrc/capi_misc.hip privacy_preserving_sfm_amd/csrc/resource_pool.hip -o /tmp/tt_nostamp || exit 1
/tmp/tt_nostamp 47 n | grep -A50 "chain, per step: wait" | head -52; /tmp/tt_nostamp 47 n | grep "^rep"
