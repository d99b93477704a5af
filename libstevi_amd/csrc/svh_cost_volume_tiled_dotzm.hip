// float cost volumes, the kernels of svh_cost_volume_tiled_impl.h: zero-mean products (ZCC, ZNCC)
#include "svh_cost_volume_tiled_impl.h"

namespace svh {

SVH_TILED_ENTRY(cost_tiled_dot_zm) { return dispatch_hr<T_DOT, true>(ctx, h_r, a, src, tgt, v_r, sign, ms, mt, ns, nt, zc, cv, shmem); }

} // namespace svh
