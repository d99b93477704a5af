// float cost volumes, the kernels of svh_cost_volume_tiled_impl.h: products (CC, NCC)
#include "svh_cost_volume_tiled_impl.h"

namespace svh {

SVH_TILED_ENTRY(cost_tiled_dot) { return dispatch_hr<T_DOT, false>(ctx, h_r, a, src, tgt, v_r, sign, ms, mt, ns, nt, zc, cv, shmem); }

} // namespace svh
