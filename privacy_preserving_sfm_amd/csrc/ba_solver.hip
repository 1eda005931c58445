// placeholder until the LM solver lands (next commit)
#include "ba_impl.hpp"
using namespace ppsfm;
extern "C" {
int pp_ba_solve(pp_ba_handle, const pp_ba_options*, pp_ba_summary*) { SetLastError("pp_ba_solve: not built yet"); return PP_ERR_INVALID; }
int pp_ba_get_trace(pp_ba_handle, double*, int32_t, int32_t*) { SetLastError("not built yet"); return PP_ERR_INVALID; }
int pp_ba_reduced_system(pp_ba_handle, const pp_ba_options*, double, int32_t*, double*, double*, int64_t) { SetLastError("not built yet"); return PP_ERR_INVALID; }
int pp_ba_set_allreduce(pp_ba_handle h, pp_allreduce_fn fn, void* ctx) { if (!h) return PP_ERR_INVALID; h->allreduce = fn; h->allreduce_ctx = ctx; return PP_OK; }
int pp_ba_get_timings(pp_ba_handle, double*, int32_t*) { SetLastError("not built yet"); return PP_ERR_INVALID; }
}
