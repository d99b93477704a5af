// Census + SGM with the disparity range split over the GPUs of one node -- one process per GPU, one RCCL exchange per frame.
// Not a reference header (the reference is single-node CPU code); it is the C++ face of the protocol of include/stevi_hip.h
// ("disparity-sharded census"): what a C++ host calls where a single-GPU host calls
//     selectedIndexToDisp(extractSelectedIndex<Cost>(sgmCostVolume<8, Cost>(unfoldBasedCostVolume<CENSUS>(l, r, h, v, D), P1, P2, margins, Pout)), 0)
// (benchmarkCrossCorrelationAlgorithms.cpp:288-294; sgm.h:360-404), with the same result bit for bit on every rank:
//
//     rank r:  keys  <- svh_census_shard_keys     its slice [begin, begin + count) of the disparity range
//              keys  <- int32 MIN all-reduce      svh_census_exchange_keys over the caller's ncclComm_t (xGMI)
//              disp  <- svh_census_shard_finish   line recurrences on the reduced keys + winner, replicated
//
// The communicator is the caller's: this header does not include or link RCCL.
#pragma once

#include "./correlation_base.h"
#include "./matching_costs.h"
#include "./stevi_hip_bridge.h"
#include "../utils/margins.h"

namespace StereoVision {
namespace Correlation {

struct ShardCommunicator {
    void *ncclComm = nullptr; // ncclComm_t of this process (one process per GPU)
    int rank = 0, nRanks = 1;
    bool exchangeAlways = false; // run the all-reduce with a single rank too (a one-GPU rehearsal of the code path)
};

// contiguous split of `total` disparities in units of `align` (32 keeps every shard a whole number of the matrix-core sweep's row
// tiles) when that divides and leaves every rank a unit, else single disparities; the first ranks take the remainder
inline std::array<int, 2> disparityShard(int total, int rank, int nRanks, int align = 32) {
    if (align > 1 && total % align == 0 && total / align >= nRanks) {
        const std::array<int, 2> u = disparityShard(total / align, rank, nRanks, 1);
        return {u[0] * align, u[1] * align};
    }
    const int base = total / nRanks, rem = total % nRanks;
    return {rank * base + (rank < rem ? rank : rem), base + (rank < rem ? 1 : 0)};
}

// The disparity map of census + SGM (Cost branch) over the ranks of `comm`; every rank passes the same images and gets the same map.
// Images: float or uint8, host arrays (uploaded per call) or HipBridge::DeviceArray (resident).  Empty result where the reference's
// chain returns empty arrays (row mismatch).  Throws where the protocol does not apply (non-integer Pout, more than 4096 disparities).
template <matchingFunctions matchFunc = matchingFunctions::CENSUS, dispDirection dDir = dispDirection::RightToLeft, int nDirections = 8, class ImgL, class ImgR>
HipBridge::DeviceArray<disp_t, 2> censusSgmDisparityShardedOnDevice(ImgL const &img_l, ImgR const &img_r, uint8_t h_radius, uint8_t v_radius, disp_t disp_width,
                                                                    ShardCommunicator const &comm, float P1 = 0.001f, float P2 = 0.01f,
                                                                    Margins const &margins = Margins(), float Pout = 100) {
    static_assert(matchFunc == matchingFunctions::CENSUS || matchFunc == matchingFunctions::HAMMING, "the disparity axis shards for census / Hamming costs only");
    static_assert(nDirections == 0 || nDirections == 4 || nDirections == 8, "0, 4 or 8 SGM directions");
    svh_stereo_params p{};
    p.match_func = static_cast<int>(matchFunc);
    p.disp_direction = dDir == dispDirection::RightToLeft ? SVH_RIGHT_TO_LEFT : SVH_LEFT_TO_RIGHT;
    p.h_radius = h_radius;
    p.v_radius = v_radius;
    p.disp_lower = 0;
    p.disp_count = disp_width;
    p.sgm_directions = nDirections;
    p.P1 = P1;
    p.P2 = P2;
    p.Pout = Pout;
    p.margins[0] = margins.left();
    p.margins[1] = margins.top();
    p.margins[2] = margins.right();
    p.margins[3] = margins.bottom();
    p.refine_kernel = -1;
    const std::array<int, 2> shard = disparityShard(disp_width, comm.rank, comm.nRanks);
    if (shard[1] == 0) throw std::runtime_error("libstevi_hip: more ranks than disparities");
    p.shard_begin = shard[0];
    p.shard_count = shard[1];
    svh_array l = HipBridge::describe(img_l), r = HipBridge::describe(img_r);
    const svh_array &src = dDir == dispDirection::RightToLeft ? r : l;
    const int H = static_cast<int>(src.shape[0]), W = static_cast<int>(src.shape[1]);
    HipBridge::DeviceArray<int32_t, 3> keys({H, W, 2});
    svh_array k = HipBridge::describe(keys);
    if (!HipBridge::check(svh_census_shard_keys(HipBridge::context(), &p, &l, &r, &k))) return HipBridge::DeviceArray<disp_t, 2>();
    if (comm.nRanks > 1 || (comm.exchangeAlways && comm.ncclComm)) {
        const int plane0_only = svh_census_shard_region1_is_global(&p, &l, &r) == 1;
        HipBridge::check(svh_census_exchange_keys(HipBridge::context(), comm.ncclComm, &k, plane0_only));
    }
    p.shard_begin = p.shard_count = 0;
    HipBridge::DeviceArray<disp_t, 2> disp({H, W});
    svh_array d = HipBridge::describe(disp);
    if (!HipBridge::check(svh_census_shard_finish(HipBridge::context(), &p, &l, &r, &k, &d, nullptr))) return HipBridge::DeviceArray<disp_t, 2>();
    return disp;
}

template <matchingFunctions matchFunc = matchingFunctions::CENSUS, dispDirection dDir = dispDirection::RightToLeft, int nDirections = 8, class ImgL, class ImgR>
Multidim::Array<disp_t, 2> censusSgmDisparitySharded(ImgL const &img_l, ImgR const &img_r, uint8_t h_radius, uint8_t v_radius, disp_t disp_width,
                                                     ShardCommunicator const &comm, float P1 = 0.001f, float P2 = 0.01f, Margins const &margins = Margins(),
                                                     float Pout = 100) {
    return censusSgmDisparityShardedOnDevice<matchFunc, dDir, nDirections>(img_l, img_r, h_radius, v_radius, disp_width, comm, P1, P2, margins, Pout).download();
}

} // namespace Correlation
} // namespace StereoVision
