// Drop-in for LibStevi's correlation/sgm.h.
#ifndef STEREOVISION_CORRELATION_SGM_H
#define STEREOVISION_CORRELATION_SGM_H

#include "../utils/margins.h"
#include "./correlation_base.h"

namespace StereoVision {
namespace Correlation {

enum class sgmDirections { // sgm.h:29-46
    Up2Down, Down2Up, Left2Right, Right2Left, UpLeft2DownRight, DownRight2UpLeft, UpRight2DownLeft, DownLeft2UpRight,
    UpLeft2Right, DownRight2Left, UpRight2Left, DownLeft2Right, UpLeft2Down, DownRight2Up, UpRight2Down, DownLeft2Up
};

// sgmCostVolume<nDirections, extractionStrategy, T_CV>(cv_base, P1, P2, margins, Pout), sgm.h:360-404.
// Reproduces the reference as written (which passes really run, min_a_cost = c_score in the Cost branch).  16 directions
// race on sgm_cv in the reference (sgm.h:299 under :336) and are rejected.
template <int nDirections, dispExtractionStartegy extractionStrategy, class T_CV>
Multidim::Array<float, 3> sgmCostVolume(Multidim::Array<T_CV, 3> const &cv_base, float P1, float P2, Margins const &margins, float Pout = 100) {
    static_assert(nDirections == 4 or nDirections == 8 or nDirections == 16, "SGM can only operate with 4, 8 or 16 directions");
    static_assert(nDirections != 16, "libstevi_hip: the reference's 16-direction lines overlap inside one OpenMP loop; its result is not defined");
    static_assert(HipBridge::sgmVolumeTypeOnGpuPath<T_CV>, "libstevi_hip: T_CV is float or an integer type of up to 32 bits (the reference casts those to float as it reads them; double is not taken)");
    auto sgm_cv = HipBridge::makeResult<Multidim::Array<float, 3>>(cv_base.shape());
    if (sgm_cv.empty()) return sgm_cv;
    const int32_t m[4] = {margins.left(), margins.top(), margins.right(), margins.bottom()};
    svh_array in = HipBridge::describe(cv_base), out = HipBridge::describe(sgm_cv);
    HipBridge::check(svh_sgm_cost_volume(HipBridge::context(), nDirections, static_cast<int>(extractionStrategy), &in, P1, P2, m, Pout, &out));
    return sgm_cv;
}

// the same on a cost volume that stays in GPU memory (HipBridge::DeviceArray): the result stays there too
template <int nDirections, dispExtractionStartegy extractionStrategy, class T_CV>
DeviceArray<float, 3> sgmCostVolume(DeviceArray<T_CV, 3> const &cv_base, float P1, float P2, Margins const &margins, float Pout = 100) {
    static_assert(nDirections == 4 or nDirections == 8 or nDirections == 16, "SGM can only operate with 4, 8 or 16 directions");
    static_assert(nDirections != 16, "libstevi_hip: the reference's 16-direction lines overlap inside one OpenMP loop; its result is not defined");
    static_assert(HipBridge::sgmVolumeTypeOnGpuPath<T_CV>, "libstevi_hip: T_CV is float or an integer type of up to 32 bits (the reference casts those to float as it reads them; double is not taken)");
    DeviceArray<float, 3> sgm_cv(cv_base.shape());
    if (sgm_cv.empty()) return sgm_cv;
    const int32_t m[4] = {margins.left(), margins.top(), margins.right(), margins.bottom()};
    svh_array in = HipBridge::describe(cv_base), out = HipBridge::describe(sgm_cv);
    auto known = cv_base.statement(); // one read: the statement is shared by every handle of the memory and may be dropped through another
    // a volume the library wrote and nobody has touched: its regional minima come with it, the probe pass over the volume is skipped
    const bool with_minima = extractionStrategy == dispExtractionStartegy::Cost && known && known->minima;
    svh_array mn{};
    if (with_minima) {
        mn.data = const_cast<float *>(known->minima.get()); // (svh_array has one pointer type; the call only reads the map)
        mn.ndim = 3;
        mn.dtype = SVH_F32;
        mn.memspace = SVH_DEVICE;
        mn.shape[0] = cv_base.shape()[0];
        mn.shape[1] = cv_base.shape()[1];
        mn.shape[2] = 2;
        mn.strides[0] = 2 * static_cast<int64_t>(cv_base.shape()[1]);
        mn.strides[1] = 2;
        mn.strides[2] = 1;
    }
    // ... and the result leaves with a statement of its own: the winner a later extractSelectedIndex<extractionStrategy> would scan the
    // 4 D bytes of every pixel for, which the kernel that writes a pixel's final costs picks while it holds them
    DeviceArray<int32_t, 2> winner({cv_base.shape()[0], cv_base.shape()[1]});
    svh_array wi = HipBridge::describe(winner);
    int written = 0;
    HipBridge::check(svh_sgm_cost_volume_winner(HipBridge::context(), nDirections, static_cast<int>(extractionStrategy), &in, with_minima ? &mn : nullptr,
                                                with_minima ? known->minima_kind : 1, with_minima ? known->max_abs : 0.0f, P1, P2, m, Pout, &out, &wi, &written));
    if (written) {
        auto st = std::make_shared<typename DeviceArray<float, 3>::Statement>();
        st->winner = winner.share();
        st->winner_strategy = static_cast<int>(extractionStrategy);
        sgm_cv.attach(st);
    }
    return sgm_cv;
}

// Extension, not in the reference: the "textbook" mode (all directions fully traversed, neighbour penalties in the Cost strategy);
// see svh_sgm_cost_volume_textbook in include/stevi_hip.h.  sgmCostVolume above stays the reference as written.
template <int nDirections, dispExtractionStartegy extractionStrategy, class T_CV>
Multidim::Array<float, 3> sgmCostVolumeTextbook(Multidim::Array<T_CV, 3> const &cv_base, float P1, float P2, Margins const &margins, float Pout = 100) {
    static_assert(nDirections == 4 or nDirections == 8, "the textbook mode operates with 4 or 8 directions");
    static_assert(HipBridge::sgmVolumeTypeOnGpuPath<T_CV>, "libstevi_hip: T_CV is float or an integer type of up to 32 bits (the reference casts those to float as it reads them; double is not taken)");
    auto sgm_cv = HipBridge::makeResult<Multidim::Array<float, 3>>(cv_base.shape());
    if (sgm_cv.empty()) return sgm_cv;
    const int32_t m[4] = {margins.left(), margins.top(), margins.right(), margins.bottom()};
    svh_array in = HipBridge::describe(cv_base), out = HipBridge::describe(sgm_cv);
    HipBridge::check(svh_sgm_cost_volume_textbook(HipBridge::context(), nDirections, static_cast<int>(extractionStrategy), &in, P1, P2, m, Pout, &out));
    return sgm_cv;
}

} // namespace Correlation
} // namespace StereoVision

#endif // STEREOVISION_CORRELATION_SGM_H
