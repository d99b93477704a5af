// Drop-in for LibStevi's interpolation/downsampling.h: DownSampleWindows and averagePoolingDownsample (2-D and 3-D float arrays).
#ifndef STEREOVISION_DOWNSAMPLING_H
#define STEREOVISION_DOWNSAMPLING_H

#include "../correlation/stevi_hip_bridge.h"

namespace StereoVision {
namespace Interpolation {

class DownSampleWindows { // interpolation/downsampling.h:30-65
  public:
    DownSampleWindows(int size) : _horizontal(size), _vertical(size) {}
    DownSampleWindows(int h_size, int v_size) : _horizontal(h_size), _vertical(v_size) {}
    DownSampleWindows(DownSampleWindows const &other) : _horizontal(other._horizontal), _vertical(other._vertical) {}
    inline int horizontal() const { return _horizontal; }
    inline int vertical() const { return _vertical; }

  private:
    int _horizontal;
    int _vertical;
};

// averagePoolingDownsample<T_I, T_O>(input, windows), interpolation/downsampling.h:67-178
template <class T_I, class T_O = float, int nDim, Multidim::ArrayDataAccessConstness Cstn>
Multidim::Array<T_O, nDim> averagePoolingDownsample(Multidim::Array<T_I, nDim, Cstn> const &input, DownSampleWindows const &windows) {
    static_assert(nDim == 2 || nDim == 3, "averagePoolingDownsample takes images with or without a channel axis");
    static_assert(std::is_same_v<std::remove_const_t<T_I>, float> && std::is_same_v<T_O, float>, "libstevi_hip: float images only");
    namespace HipBridge = Correlation::HipBridge;
    auto shape = input.shape();
    shape[0] = (shape[0] + (windows.vertical() - 1)) / windows.vertical(); // :79-80
    shape[1] = (shape[1] + (windows.horizontal() - 1)) / windows.horizontal();
    auto output = StereoVision::Correlation::HipBridge::makeResult<Multidim::Array<T_O, nDim>>(shape);
    if (output.empty()) return output;
    svh_array in = HipBridge::describe(input), out = HipBridge::describe(output);
    HipBridge::check(svh_average_pooling_downsample(HipBridge::context(), &in, windows.horizontal(), windows.vertical(), &out));
    return output;
}

} // namespace Interpolation
} // namespace StereoVision

#endif // STEREOVISION_DOWNSAMPLING_H
