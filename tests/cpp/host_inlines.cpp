// The inline HOST functions of the drop-in headers ("Missing 2"): hammingScalar / hammingDistance, dotProduct,
// SumSquareDiff, SumAbsDiff, MatchingFunctionTraits<f>::featureComparison, refineCostTriplet, condImgRef, MatchingFuncComputeTypeInfos,
// optimalDispAndCost / defaultCvValForMatchFunc, Contiguity's tables -- called the way the reference's own callers call them
// (on_demand_cost_volume.h / cross_correlations.h:238-241: featureComparison on two 1-D views of feature vectors).
//
//   host_inlines <H> <W> <F> <nW> <D> <dir>
// reads dir/{fl,fr}.f32 (H, W, F) float feature volumes and dir/{wl,wr}.u32 (H, W, nW) census words (target side already through rule
// E2), evaluates CV(i, j, d) = featureComparison(source(i, j, :), target(i, j + d, :)) for every function with NO GPU call (zeros beyond
// the right edge, cross_correlations.h:230-236), writes dir/cv_<NAME>.f32; reads dir/triplets.f32 (n, 3) and writes dir/ref_<k>.f32.
// tests/test_cpp_host_inlines.py compares everything with the oracle.  Compile-time properties are static_asserts below.
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include "correlation/cost_based_refinement.h"
#include "correlation/cross_correlations.h"
#include "correlation/matching_costs.h"
#include "utils/contiguity.h"

namespace SC = StereoVision::Correlation;
using SC::matchingFunctions;
using StereoVision::Contiguity;

// ---- compile-time surface -------------------------------------------------------------------------------------------------------
static_assert(std::is_same_v<SC::MatchingFuncComputeTypeInfos<matchingFunctions::SAD, float>::FeatureType, float>);
static_assert(std::is_same_v<SC::MatchingFuncComputeTypeInfos<matchingFunctions::SAD, uint8_t>::FeatureType, uint8_t>);   // matching_costs.h:749-757
static_assert(std::is_same_v<SC::MatchingFuncComputeTypeInfos<matchingFunctions::ZSAD, uint8_t>::FeatureType, int16_t>);
static_assert(std::is_same_v<SC::MatchingFuncComputeTypeInfos<matchingFunctions::NCC, uint8_t>::FeatureType, int16_t>);
static_assert(std::is_same_v<SC::MatchingFuncComputeTypeInfos<matchingFunctions::NCC, uint16_t>::FeatureType, int32_t>);  // accumulation_extended_t
static_assert(std::is_same_v<SC::MatchingFuncComputeTypeInfos<matchingFunctions::HAMMING, float>::FeatureType, uint32_t>); // :759-771
static_assert(std::is_same_v<SC::MatchingFuncComputeTypeInfos<matchingFunctions::CENSUS, uint8_t>::FeatureType, uint32_t>);
static_assert(std::is_same_v<SC::FeatureTypeForMatchFunc<matchingFunctions::CENSUS, float>, SC::census_data_t>);
static_assert(std::is_same_v<SC::FeatureTypeForMatchFunc<matchingFunctions::ZNCC, float>, float>);
static_assert(std::is_same_v<StereoVision::TypesManipulations::accumulation_extended_t<uint32_t>, int64_t>);
static_assert(std::is_same_v<StereoVision::TypesManipulations::accumulation_extended_t<int32_t>, int32_t>);
static_assert(std::is_same_v<StereoVision::TypesManipulations::accumulation_extended_t<double>, double>);
static_assert(StereoVision::TypesManipulations::equivalentOneForNormalizing<int16_t>() == 256);
static_assert(StereoVision::TypesManipulations::equivalentOneForNormalizing<int32_t>() == 65536);
static_assert(StereoVision::TypesManipulations::equivalentOneForNormalizing<float>() == 1.0f);
static_assert(SC::defaultCvValForMatchFunc<matchingFunctions::SAD>() == std::numeric_limits<float>::max());
static_assert(SC::defaultCvValForMatchFunc<matchingFunctions::NCC>() == std::numeric_limits<float>::min()); // (sic) the reference's choice
static_assert(std::is_same_v<SC::condImgRef<float, uint8_t, SC::dispDirection::RightToLeft>::T_S, uint8_t>);
static_assert(std::is_same_v<SC::condImgRef<float, uint8_t, SC::dispDirection::RightToLeft>::T_T, float>);
static_assert(std::is_same_v<SC::condImgRef<float, uint8_t, SC::dispDirection::LeftToRight>::T_S, float>);
// utils/contiguity.h:34-190
static_assert(Contiguity::nDirections(Contiguity::Queen) == 8 && Contiguity::nDirections(Contiguity::Rook) == 4 && Contiguity::nDirections(Contiguity::Bishop) == 4);
static_assert(Contiguity::nCornerDirections(Contiguity::Queen) == 3 && Contiguity::nCornerDirections(Contiguity::Rook) == 2 && Contiguity::nCornerDirections(Contiguity::Bishop) == 1);
static_assert(Contiguity::nTilingDirections(Contiguity::Queen) == 4 && Contiguity::nTilingDirections(Contiguity::Bishop) == 2);
static_assert(Contiguity::BidimensionalContiguityTraits<Contiguity::Queen>::nDir == 8 && Contiguity::BidimensionalContiguityTraits<Contiguity::Rook>::nCornerDir == 2 &&
              Contiguity::BidimensionalContiguityTraits<Contiguity::Bishop>::nTilingDir == 2);
constexpr bool same(std::array<int, 2> a, int i, int j) { return a[0] == i && a[1] == j; }
constexpr auto queen = Contiguity::getDirections<Contiguity::Queen>();
static_assert(same(queen[0], 1, 1) && same(queen[1], 1, 0) && same(queen[2], 1, -1) && same(queen[3], 0, 1) && same(queen[4], 0, -1) && same(queen[5], -1, 1) &&
              same(queen[6], -1, 0) && same(queen[7], -1, -1));
constexpr auto rook = Contiguity::getDirections<Contiguity::Rook>();
static_assert(same(rook[0], 1, 0) && same(rook[1], 0, 1) && same(rook[2], 0, -1) && same(rook[3], -1, 0));
constexpr auto bishop = Contiguity::getDirections<Contiguity::Bishop>();
static_assert(same(bishop[0], 1, 1) && same(bishop[1], 1, -1) && same(bishop[2], -1, 1) && same(bishop[3], -1, -1));
constexpr auto queenCorner = Contiguity::getCornerDirections<Contiguity::Queen>();
static_assert(same(queenCorner[0], 1, 1) && same(queenCorner[1], 1, 0) && same(queenCorner[2], 0, 1));
constexpr auto rookCorner = Contiguity::getCornerDirections<Contiguity::Rook>();
static_assert(same(rookCorner[0], 1, 0) && same(rookCorner[1], 0, 1));
static_assert(same(Contiguity::getCornerDirections<Contiguity::Bishop>()[0], 1, 1));
constexpr auto queenTiling = Contiguity::getTilingDirections<Contiguity::Queen>();
static_assert(same(queenTiling[0], 1, 1) && same(queenTiling[1], 1, 0) && same(queenTiling[2], 0, 1) && same(queenTiling[3], 1, -1));
constexpr auto rookTiling = Contiguity::getTilingDirections<Contiguity::Rook>();
static_assert(same(rookTiling[0], 1, 0) && same(rookTiling[1], 0, 1));
constexpr auto bishopTiling = Contiguity::getTilingDirections<Contiguity::Bishop>();
static_assert(same(bishopTiling[0], 1, 1) && same(bishopTiling[1], 1, -1));

template <class T> static std::vector<T> slurp(std::string const &path, std::size_t n) {
    std::vector<T> v(n);
    std::ifstream in(path, std::ios::binary);
    in.read(reinterpret_cast<char *>(v.data()), static_cast<std::streamsize>(n * sizeof(T)));
    if (static_cast<std::size_t>(in.gcount()) != n * sizeof(T)) {
        fprintf(stderr, "short read: %s\n", path.c_str());
        exit(2);
    }
    return v;
}
template <class T> static void dump(std::string const &path, std::vector<T> const &v) {
    std::ofstream(path, std::ios::binary).write(reinterpret_cast<const char *>(v.data()), static_cast<std::streamsize>(v.size() * sizeof(T)));
}

// CV(i, j, d) = featureComparison(source(i, j, :), target(i, j + d, :)), RightToLeft, target = zeros past the right edge
template <matchingFunctions f, class T> static std::vector<float> volume(std::vector<T> &src, std::vector<T> &tgt, int H, int W, int F, int D) {
    using View = Multidim::Array<T, 1, Multidim::ConstView>;
    std::vector<float> cv(static_cast<std::size_t>(H) * W * D);
    std::vector<T> zeros(F, T(0));
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++)
            for (int d = 0; d < D; d++) {
                Multidim::Array<T, 1> s(&src[(static_cast<std::size_t>(i) * W + j) * F], {F}, {1});
                Multidim::Array<T, 1> t(j + d < W ? &tgt[(static_cast<std::size_t>(i) * W + j + d) * F] : zeros.data(), {F}, {1});
                View sv(s), tv(t);
                cv[(static_cast<std::size_t>(i) * W + j) * D + d] = SC::MatchingFunctionTraits<f>::template featureComparison<T, T, float>(sv, tv);
            }
    return cv;
}

int main(int argc, char **argv) {
    if (argc != 7) return 1;
    const int H = atoi(argv[1]), W = atoi(argv[2]), F = atoi(argv[3]), nW = atoi(argv[4]), D = atoi(argv[5]);
    const std::string dir = argv[6];
    auto fl = slurp<float>(dir + "/fl.f32", static_cast<std::size_t>(H) * W * F), fr = slurp<float>(dir + "/fr.f32", static_cast<std::size_t>(H) * W * F);
    auto wl = slurp<uint32_t>(dir + "/wl.u32", static_cast<std::size_t>(H) * W * nW), wr = slurp<uint32_t>(dir + "/wr.u32", static_cast<std::size_t>(H) * W * nW);
    // source = right, target = left (RightToLeft)
    dump(dir + "/cv_CC.f32", volume<matchingFunctions::CC>(fr, fl, H, W, F, D));
    dump(dir + "/cv_NCC.f32", volume<matchingFunctions::NCC>(fr, fl, H, W, F, D)); // (features as given: the dot product is the function's comparison)
    dump(dir + "/cv_SSD.f32", volume<matchingFunctions::SSD>(fr, fl, H, W, F, D));
    dump(dir + "/cv_SAD.f32", volume<matchingFunctions::SAD>(fr, fl, H, W, F, D));
    dump(dir + "/cv_ZSSD.f32", volume<matchingFunctions::ZSSD>(fr, fl, H, W, F, D));
    dump(dir + "/cv_ZSAD.f32", volume<matchingFunctions::ZSAD>(fr, fl, H, W, F, D));
    dump(dir + "/cv_HAMMING.f32", volume<matchingFunctions::HAMMING>(wr, wl, H, W, nW, D));
    dump(dir + "/cv_CENSUS.f32", volume<matchingFunctions::CENSUS>(wr, wl, H, W, nW, D));

    // the free functions on both containers give what featureComparison gave
    {
        std::vector<float> a(fr.begin(), fr.begin() + F), b(fl.begin(), fl.begin() + F);
        Multidim::Array<float, 1> aa(a.data(), {F}, {1}), ba(b.data(), {F}, {1});
        Multidim::Array<float, 1, Multidim::ConstView> av(aa), bv(ba);
        if (SC::dotProduct<float, float>(a, b) != SC::dotProduct<float, float>(av, bv) || SC::SumSquareDiff<float, float>(a, b) != SC::SumSquareDiff<float, float>(av, bv) ||
            SC::SumAbsDiff<float, float>(a, b) != SC::SumAbsDiff<float, float>(av, bv)) {
            fprintf(stderr, "std::vector and Multidim::Array overloads disagree\n");
            return 3;
        }
        std::vector<uint32_t> x(wr.begin(), wr.begin() + nW), y(wl.begin(), wl.begin() + nW);
        Multidim::Array<uint32_t, 1> xa(x.data(), {nW}, {1}), ya(y.data(), {nW}, {1});
        Multidim::Array<uint32_t, 1, Multidim::ConstView> xv(xa), yv(ya);
        if (SC::hammingDistance(x, y) != SC::hammingDistance(xv, yv)) return 3;
        // integer inputs, integer output: uint8 features accumulate in the requested type (dotProduct renormalises a too-narrow one)
        std::vector<uint8_t> p{200, 100, 7}, q{3, 250, 9};
        if (SC::SumAbsDiff<uint8_t, uint8_t, int32_t>(p, q) != 197 + 150 + 2 || SC::SumSquareDiff<uint8_t, uint8_t, int32_t>(p, q) != 197 * 197 + 150 * 150 + 4 ||
            SC::dotProduct<uint8_t, uint8_t, int32_t>(p, q) != 600 + 25000 + 63) {
            fprintf(stderr, "integer accumulations\n");
            return 3;
        }
        if (SC::hammingScalar(uint32_t(0xF0F0F0F0u), uint32_t(0x0F0F0F0Fu)) != 32 || SC::hammingScalar(uint8_t(0x81), uint8_t(0x01)) != 1 ||
            SC::hammingScalar(uint32_t(5), uint32_t(5)) != 0 || SC::hammingScalar(uint16_t(0xFFFF), uint32_t(0)) != 16) {
            fprintf(stderr, "hammingScalar\n");
            return 3;
        }
    }
    // optimalDispAndCost: strict improvement only (the incumbent keeps ties), per strategy
    {
        int d0 = 3, d1 = 9;
        float c0 = 1.5f, c1 = 1.5f, c2 = 1.0f, c3 = 2.0f;
        auto keepC = SC::optimalDispAndCost<matchingFunctions::SAD>(d0, c0, d1, c1);
        auto takeC = SC::optimalDispAndCost<matchingFunctions::SAD>(d0, c0, d1, c2);
        auto keepS = SC::optimalDispAndCost<matchingFunctions::NCC>(d0, c0, d1, c1);
        auto takeS = SC::optimalDispAndCost<matchingFunctions::NCC>(d0, c0, d1, c3);
        if (keepC.first != 3 || takeC.first != 9 || takeC.second != 1.0f || keepS.first != 3 || takeS.first != 9 || takeS.second != 2.0f) {
            fprintf(stderr, "optimalDispAndCost\n");
            return 3;
        }
    }
    // condImgRef hands back the arrays it was given, by role
    {
        Multidim::Array<float, 2> l(2, 3);
        Multidim::Array<uint8_t, 2> r(2, 3);
        SC::condImgRef<float, uint8_t, SC::dispDirection::RightToLeft> r2l(l, r);
        SC::condImgRef<float, uint8_t, SC::dispDirection::LeftToRight> l2r(l, r);
        if (&r2l.source() != &r || &r2l.target() != &l || &l2r.source() != &l || &l2r.target() != &r) {
            fprintf(stderr, "condImgRef\n");
            return 3;
        }
    }
    // refineCostTriplet, the three kernels
    {
        std::ifstream in(dir + "/triplets.f32", std::ios::binary | std::ios::ate);
        const std::size_t n = static_cast<std::size_t>(in.tellg()) / (3 * sizeof(float));
        auto t = slurp<float>(dir + "/triplets.f32", 3 * n);
        std::vector<float> e(n), p(n), g(n);
        for (std::size_t k = 0; k < n; k++) {
            e[k] = SC::refineCostTriplet<SC::InterpolationKernel::Equiangular>(t[3 * k], t[3 * k + 1], t[3 * k + 2]);
            p[k] = SC::refineCostTriplet<SC::InterpolationKernel::Parabola>(t[3 * k], t[3 * k + 1], t[3 * k + 2]);
            g[k] = SC::refineCostTriplet<SC::InterpolationKernel::Gaussian>(t[3 * k], t[3 * k + 1], t[3 * k + 2]);
        }
        dump(dir + "/ref_0.f32", e);
        dump(dir + "/ref_1.f32", p);
        dump(dir + "/ref_2.f32", g);
    }
    return 0;
}
