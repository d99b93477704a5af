// The Multidim::Array members SURVEY 8(b) lists from call sites, each used the way the cited reference line uses it (host only: no GPU
// call is made).  Exit code 0 = every check passed; a failed check prints its line.
#include <cstdio>
#include <algorithm>
#include <numeric>
#include <vector>

#include <correlation/stevi_hip_bridge.h>

static int failures = 0;
#define CHECK(cond)                                                   \
    do {                                                              \
        if (!(cond)) {                                                \
            std::fprintf(stderr, "line %d: %s\n", __LINE__, #cond);   \
            failures++;                                               \
        }                                                             \
    } while (0)

namespace HB = StereoVision::Correlation::HipBridge;

template <class T> static Multidim::Array<T, 3> volume(int H, int W, int F) {
    Multidim::Array<T, 3> v(H, W, F);
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++)
            for (int c = 0; c < F; c++) v.template at<Multidim::AccessCheck::Nocheck>(i, j, c) = static_cast<T>(100 * i + 10 * j + c);
    return v;
}

// cross_correlations.h:227-228, hierarchical.h:160: the feature vector of one pixel of a `const&` volume
template <class T_S> static T_S feature_sum(Multidim::Array<T_S, 3> const &source_feature_volume, int i, int j) {
    Multidim::Array<T_S, 1> source_feature_vector = source_feature_volume.subView(Multidim::DimIndex(i), Multidim::DimIndex(j), Multidim::DimSlice());
    T_S s = 0;
    for (int c = 0; c < source_feature_vector.shape()[0]; c++) s += source_feature_vector.valueUnchecked(c);
    return s;
}

// cross_correlations.h:654-659: a const feature volume, NonConstView or ConstView, stored as a ConstView array
template <class FV_T> static float reshaped_first(FV_T const &feature_vol_impl) {
    Multidim::Array<float, 3, Multidim::ArrayDataAccessConstness::ConstView> feature_vol;
    feature_vol = feature_vol_impl.template buildReshapedView<3>(feature_vol_impl.shape(), feature_vol_impl.strides());
    return feature_vol.valueUnchecked(1, 2, 3);
}

int main() {
    constexpr Multidim::AccessCheck Nc = Multidim::AccessCheck::Nocheck;
    const Multidim::Array<float, 3> vol = volume<float>(4, 5, 6);

    // subView with two indices and a slice
    CHECK(feature_sum(vol, 2, 3) == 6 * 230 + 15);
    // subView with two slices: testCorrelationFilters.cpp:303-307 (windows of 2 v_radius + 1 rows, 2 h_radius + 1 columns)
    {
        Multidim::Array<float, 2> img(6, 9);
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 9; j++) img.at(i, j) = 10.f * i + j;
        const int v_radius = 1, h_radius = 2, i = 3;
        Multidim::Array<float, 2> window1 = img.subView(Multidim::DimSlice(0, 2 * v_radius + 1), Multidim::DimSlice(0, 2 * h_radius + 1));
        Multidim::Array<float, 2> window2 = img.subView(Multidim::DimSlice(0, 2 * v_radius + 1), Multidim::DimSlice(i, i + 2 * h_radius + 1));
        CHECK(window1.shape()[0] == 3 && window1.shape()[1] == 5 && window2.shape()[1] == 5);
        CHECK(window1.valueUnchecked(2, 4) == 24.f && window2.valueUnchecked(2, 4) == 27.f);
        window2.at(0, 0) = -1.f; // a view: the write lands in the image
        CHECK(img.valueUnchecked(0, 3) == -1.f);
    }
    // sliceView then subView on the resulting ConstView array: image_based_refinement.h:399, :456
    {
        Multidim::Array<float, 3, Multidim::ConstView> target_feature_volume = vol; // (NonConstView -> ConstView)
        Multidim::Array<float, 2, Multidim::ConstView> line = target_feature_volume.sliceView(0, 2);
        CHECK(line.shape()[0] == 5 && line.shape()[1] == 6 && line.valueUnchecked(4, 5) == 245.f);
        const int minJ = 1, maxJ = 4, f = 2;
        Multidim::Array<float, 1, Multidim::ConstView> featureLine = line.subView(Multidim::DimSlice(minJ, maxJ), Multidim::DimIndex(f));
        CHECK(featureLine.shape()[0] == 3 && featureLine.valueUnchecked(0) == 212.f && featureLine.valueUnchecked(2) == 232.f);
    }
    // buildReshapedView from both kinds of array
    {
        Multidim::Array<float, 3, Multidim::ConstView> cview = vol;
        CHECK(reshaped_first(vol) == 123.f && reshaped_first(cview) == 123.f);
    }
    // indexDimView: on_demand_cost_volume.h:164-172 (the line of features through one pixel)
    {
        const int featureDim = 2;
        Multidim::Array<float, 1, Multidim::ConstView> source_features = vol.indexDimView(featureDim, {3, 1});
        CHECK(source_features.shape()[0] == 6 && source_features.valueUnchecked(5) == 315.f);
    }
    // takePointer: benchmarkStereoMatchingModels.cpp:202-204
    {
        Multidim::Array<int, 2> out(3, 4);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 4; j++) out.at(i, j) = 4 * i + j;
        int *data = out.takePointer();
        Multidim::Array<int, 2> ret(data, {out.shape()[0], out.shape()[1]}, {out.strides()[0], out.strides()[1]}, true);
        CHECK(out.empty() && ret.valueUnchecked(2, 3) == 11);
    }
    // IndexBlock::setZero / moveToNextIndex: io/image_io.h:56-67 (element-wise conversion of any layout)
    {
        Multidim::Array<float, 3> image({2, 3, 4}, {1, 2, 6}); // first index fastest
        typename Multidim::Array<float, 3>::IndexBlock idx;
        idx.setZero();
        int n = 0;
        for (std::size_t e = 0; e < image.flatLenght(); e++) {
            image.at<Nc>(idx) = static_cast<float>(100 * idx[0] + 10 * idx[1] + idx[2]);
            idx.moveToNextIndex(image.shape());
            n++;
        }
        CHECK(n == 24 && idx[0] == 0 && idx[1] == 0 && idx[2] == 0); // every index once, back at the start
        Multidim::Array<int, 3> converted = image.cast<int>();
        CHECK(converted.valueUnchecked(1, 2, 3) == 123 && !image.isDense() && converted.isDense());
        CHECK(&image.atUnchecked(0) == &image.atUnchecked(0, 0, 0)); // io/image_io.h:96: address of the first element
    }
    // IndexConverter: on_demand_cost_volume.h:208-213 (class template argument deduction from the shape) and :497-502
    {
        std::array<int, 3> tcv_shape = {3, 4, 5};
        Multidim::IndexConverter idxConv(tcv_shape);
        std::vector<int> seen(60, 0);
        for (int i = 0; i < idxConv.numberOfPossibleIndices(); i++) {
            std::array<int, 3> tcvid = idxConv.getIndexFromPseudoFlatId(i);
            seen[(tcvid[0] * 4 + tcvid[1]) * 5 + tcvid[2]]++;
        }
        CHECK(idxConv.numberOfPossibleIndices() == 60 && std::accumulate(seen.begin(), seen.end(), 0) == 60 && *std::min_element(seen.begin(), seen.end()) == 1);
        Multidim::IndexConverter<3> partial(tcv_shape, {2});
        CHECK(partial.numberOfPossibleIndices() == 12 && partial.getIndexFromPseudoFlatId(11)[2] == 0);
    }
    // what the shims hand to the C ABI: in place for an array that may be written through (strides as they are), a dense private
    // copy for a ConstView array (kept by the thread while the call runs)
    {
        Multidim::Array<float, 3> strided({2, 3, 4}, {1, 2, 6});
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 3; j++)
                for (int c = 0; c < 4; c++) strided.at(i, j, c) = 100.f * i + 10.f * j + c;
        svh_array d = HB::describe(strided);
        CHECK(d.data == &strided.atUnchecked(0, 0, 0) && d.strides[0] == 1 && d.strides[2] == 6 && d.memspace == SVH_HOST && d.dtype == SVH_F32);
        Multidim::Array<float, 2, Multidim::ConstView> slice = strided.sliceView(1, 2); // (i, c) at j = 2
        svh_array ds = HB::describe(slice);
        const float *p = static_cast<const float *>(ds.data);
        CHECK(ds.shape[0] == 2 && ds.shape[1] == 4 && p[1 * ds.strides[0] + 3 * ds.strides[1]] == 123.f);
        Multidim::Array<float, 2> empty;
        CHECK(HB::describe(empty).data == nullptr && HB::firstElement(empty) == nullptr);
    }
    if (failures) return 1;
    std::puts("multidim compat ok");
    return 0;
}
