// float cost volumes, the kernels of svh_cost_volume_tiled_impl.h: absolute differences (SAD, ZSAD)
#include "svh_cost_volume_tiled_impl.h"

namespace svh {

SVH_TILED_ENTRY(cost_tiled_sad) { return dispatch_hr<T_SAD, false>(ctx, h_r, a, src, tgt, v_r, sign, ms, mt, ns, nt, zc, cv, shmem); }
SVH_TILED_ENTRY(cost_tiled_sad_zm) { return dispatch_hr<T_SAD, true>(ctx, h_r, a, src, tgt, v_r, sign, ms, mt, ns, nt, zc, cv, shmem); }

} // namespace svh
