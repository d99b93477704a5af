// Drop-in for LibStevi's correlation/on_demand_features_volume.h, the part examples/stereo-match uses: the description of an
// on-demand feature volume (an image, a window of (row, column, channel) offsets, a zero-mean / normalising decorator).  The
// features themselves are evaluated on the GPU by the consumers (cachelessPatchMatch, CachelessOnDemandCostVolume); there is no
// host-side getFeatureVec.
#ifndef STEREOVISION_ON_DEMAND_FEATURES_VOLUME_H
#define STEREOVISION_ON_DEMAND_FEATURES_VOLUME_H

#include <array>
#include <vector>

#include "./matching_costs.h"

namespace StereoVision {
namespace Correlation {

// OnDemandFeaturesVolume<T_F, nD, constness, FeaturesDims...>, on_demand_features_volume.h:34-165
template <class T_F, int nD, Multidim::ArrayDataAccessConstness constness, int... FeaturesDims> class OnDemandFeaturesVolume {
  public:
    static_assert(sizeof...(FeaturesDims) < nD);
    using ArrayType = Multidim::Array<T_F, nD, constness>;
    static constexpr int nInDim = nD;
    typedef T_F ScalarT;
    static constexpr int nFeaturesDim = sizeof...(FeaturesDims);
    static constexpr int nOutDim = nInDim - nFeaturesDim;
    using InIndex = std::array<int, nInDim>;
    using OutIndex = std::array<int, nOutDim>;

    OnDemandFeaturesVolume(std::vector<InIndex> const &window, ArrayType const &array) : _window(window), _array(array) {}

    // the shape the corresponding feature volume would have (feature axis last), :103-112
    std::array<int, nOutDim + 1> shape() const {
        static_assert(nD == 3 && nFeaturesDim == 1, "libstevi_hip: (row, column, channel) images with the channel axis as feature dimension");
        return {static_cast<int>(_array.shape()[0]), static_cast<int>(_array.shape()[1]), static_cast<int>(_window.size())};
    }
    std::vector<InIndex> const &window() const { return _window; }
    ArrayType const &array() const { return _array; }

    // The GPU evaluates full rectangular windows in the order examples/stereo-match/main.cpp:150-164 builds them (rows outer,
    // columns, channels innermost over every channel).  Returns false when the window list is anything else.
    bool rectangularWindow(int &v_radius, int &h_radius) const {
        const int C = static_cast<int>(_array.shape()[2]);
        if (_window.empty() || C <= 0 || _window.size() % C != 0) return false;
        const int vr = -_window.front()[0], hr = -_window.front()[1];
        if (vr < 0 || hr < 0 || static_cast<size_t>((2 * vr + 1) * (2 * hr + 1) * C) != _window.size()) return false;
        size_t f = 0;
        for (int i = -vr; i <= vr; i++)
            for (int j = -hr; j <= hr; j++)
                for (int c = 0; c < C; c++, f++)
                    if (_window[f][0] != i || _window[f][1] != j || _window[f][2] != c) return false;
        v_radius = vr;
        h_radius = hr;
        return true;
    }

  protected:
    std::vector<InIndex> _window;
    ArrayType const &_array;
};

// ZNFeaturesVolumeDecorator<ZeroMean, Normalized>, :167-214: mean = sum / nF, norm = sqrt(sum of squares / nF)
template <bool ZeroMean, bool Normalized> struct ZNFeaturesVolumeDecorator {
    static constexpr bool zeroMean = ZeroMean;
    static constexpr bool normalized = Normalized;
};

// OnDemandDecoratedFeaturesVolume<Decorator, T_F, nD, constness, FeaturesDims...>, :216-244
template <class Decorator, class T_F, int nD, Multidim::ArrayDataAccessConstness constness, int... FeaturesDims>
class OnDemandDecoratedFeaturesVolume : public OnDemandFeaturesVolume<T_F, nD, constness, FeaturesDims...> {
    using ParentT = OnDemandFeaturesVolume<T_F, nD, constness, FeaturesDims...>;

  public:
    using DecoratorT = Decorator;
    OnDemandDecoratedFeaturesVolume(std::vector<typename ParentT::InIndex> const &window, typename ParentT::ArrayType const &array) : ParentT(window, array) {}
};

} // namespace Correlation
} // namespace StereoVision

#endif // STEREOVISION_ON_DEMAND_FEATURES_VOLUME_H
