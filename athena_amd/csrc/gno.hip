// Graph neural operator kernels -- placeholder entry points until the re-associated kernels land.
// They fail loudly (never a CPU detour).
#include "common.h"
using namespace amp;
extern "C" {
int athena_mp_gno_aggregate_fwd(const athena_mp_graph *, int32_t, int32_t, int32_t, int32_t, const float *,
                                const float *, const float *, float *)
{ set_error("gno_aggregate_fwd: not implemented"); return 3; }
int athena_mp_gno_aggregate_bwd_x(const athena_mp_graph *, int32_t, int32_t, int32_t, int32_t, const float *,
                                  const float *, const float *, float *)
{ set_error("gno_aggregate_bwd_x: not implemented"); return 3; }
int athena_mp_gno_aggregate_bwd_theta(const athena_mp_graph *, int32_t, int32_t, int32_t, int32_t,
                                      const float *, const float *, const float *, const float *, float *)
{ set_error("gno_aggregate_bwd_theta: not implemented"); return 3; }
int athena_mp_gno_aggregate_bwd_coords(const athena_mp_graph *, int32_t, int32_t, int32_t, int32_t,
                                       const float *, const float *, const float *, const float *, float *)
{ set_error("gno_aggregate_bwd_coords: not implemented"); return 3; }
}
