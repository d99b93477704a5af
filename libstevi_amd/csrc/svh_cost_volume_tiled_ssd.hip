// float cost volumes, the kernels of svh_cost_volume_tiled_impl.h: squared differences (SSD, ZSSD)
#include "svh_cost_volume_tiled_impl.h"

namespace svh {

SVH_TILED_ENTRY(cost_tiled_ssd) { return dispatch_hr<T_SSD, false>(ctx, h_r, a, src, tgt, v_r, sign, ms, mt, ns, nt, zc, cv, shmem); }
SVH_TILED_ENTRY(cost_tiled_ssd_zm) { return dispatch_hr<T_SSD, true>(ctx, h_r, a, src, tgt, v_r, sign, ms, mt, ns, nt, zc, cv, shmem); }

} // namespace svh
