// Drop-in for LibStevi's correlation/matching_costs.h: the matchingFunctions enum, the compile-time traits the callers read (Name,
// ZeroMean, Normalized, extractionStrategy, isCensusBased), and the per-vector comparison functions as inline HOST functions --
// dotProduct / SumSquareDiff / SumAbsDiff / hammingScalar / hammingDistance and MatchingFunctionTraits<f>::featureComparison
// (matching_costs.h:59-156, :236-263, :419-685) -- for callers that compare two feature vectors they hold themselves (the reference's
// on_demand_cost_volume.h, template_matching.h).  They are a few scalar operations on two short host vectors: there is nothing to
// offload, and they are not a fallback for anything -- every volume-sized comparison runs on the GPU (libstevi_amd/csrc/
// svh_cost_volume*.hip) and no function in this tree routes a volume through them.  Accumulation order is the reference's (increasing
// index, no reassociation), so a vector compared here equals the same voxel of a GPU volume to the last bit for the integer functions
// and within float rounding of the column-sum form for the others (tests/cpp/host_inlines.cpp).
//
// Not provided, by design: MEDAD / ZMEDAD / KERMI (no GPU path: onGpuPath<f>() is false and every entry point static_asserts on it),
// and the barycentricBestApproximation members, which belong to image_based_refinement.h's feature-domain refinement (Eigen).
#ifndef STEREOVISION_MATCHING_COSTS_H
#define STEREOVISION_MATCHING_COSTS_H

#include <cmath>
#include <cstring>
#include <limits>
#include <type_traits>
#include <utility>
#include <vector>

#include "../utils/types_manipulations.h"
#include "./correlation_base.h"

namespace StereoVision {
namespace Correlation {

enum class matchingFunctions { // matching_costs.h:38-53
    None = -1, CC = 0, NCC = 1, SSD = 2, SAD = 3, ZCC = 4, ZNCC = 5, ZSSD = 6, ZSAD = 7, MEDAD = 8, ZMEDAD = 9, HAMMING = 10, CENSUS = 11, KERMI = 12
};

typedef uint16_t hamming_cv_t;  // :234
typedef uint32_t census_data_t; // :773

namespace HipBridge {
namespace detail {
// element i of a feature vector, whichever of the two containers the reference's overloads take
template <class T, Multidim::ArrayDataAccessConstness C> inline T element(Multidim::Array<T, 1, C> const &v, int i) { return v.valueUnchecked(i); }
template <class T> inline T element(std::vector<T> const &v, int i) { return v[static_cast<std::size_t>(i)]; }
template <class T, Multidim::ArrayDataAccessConstness C> inline int length(Multidim::Array<T, 1, C> const &v) { return v.shape()[0]; }
template <class T> inline int length(std::vector<T> const &v) { return static_cast<int>(v.size()); }

template <class T_S, class T_T, class T_O> constexpr void checkOutputType() {
    static_assert((std::is_integral_v<T_S> && std::is_integral_v<T_T>) || !std::is_integral_v<T_O>,
                  "Cannot process floating point inputs for non floating point output"); // matching_costs.h:63-64
}

// score = term(s_0, t_0) + term(s_1, t_1) + ... in this order, in T_O (the reference's loops: one `score +=` per element)
template <class T_O, class VS, class VT, class Term> inline T_O accumulate(VS const &source, VT const &target, Term term) {
    T_O score = 0;
    const int n = length(source);
    for (int i = 0; i < n; i++) score += term(static_cast<T_O>(element(source, i)), static_cast<T_O>(element(target, i)));
    return score;
}

template <class T_O, class VS, class VT> inline T_O dot(VS const &source, VT const &target) {
    using T_S = std::decay_t<decltype(element(source, 0))>;
    using T_T = std::decay_t<decltype(element(target, 0))>;
    checkOutputType<T_S, T_T, T_O>();
    // an integer output too narrow for the full product is renormalised per term (matching_costs.h:69-70)
    constexpr bool renormalise = std::is_integral_v<T_O> && (sizeof(T_O) < 2 * (sizeof(T_S) > sizeof(T_T) ? sizeof(T_S) : sizeof(T_T)));
    return accumulate<T_O>(source, target, [](T_O a, T_O b) -> T_O {
        if constexpr (renormalise) return static_cast<T_O>((a * b) / TypesManipulations::equivalentOneForNormalizing<T_O>());
        else return static_cast<T_O>(a * b);
    });
}
template <class T_O, class VS, class VT> inline T_O squares(VS const &source, VT const &target) {
    checkOutputType<std::decay_t<decltype(element(source, 0))>, std::decay_t<decltype(element(target, 0))>, T_O>();
    return accumulate<T_O>(source, target, [](T_O a, T_O b) -> T_O {
        const T_O d = a - b;
        return static_cast<T_O>(d * d);
    });
}
template <class T_O, class VS, class VT> inline T_O absolutes(VS const &source, VT const &target) {
    checkOutputType<std::decay_t<decltype(element(source, 0))>, std::decay_t<decltype(element(target, 0))>, T_O>();
    return accumulate<T_O>(source, target, [](T_O a, T_O b) -> T_O {
        const T_O d = a - b;
        if constexpr (std::is_integral_v<T_O>) return static_cast<T_O>(d < 0 ? -d : d);
        else return std::fabs(d);
    });
}

// the low 32 bits a feature occupies: integers by value (zero- or sign-extended like a conversion to uint32_t), a 32-bit float by
// its bit pattern (the reference spells this reinterpret_cast<uint32_t>(n), which compiles for integer features only)
template <class T> inline uint32_t bitsOf(T v) {
    static_assert(sizeof(T) <= sizeof(uint32_t), "Cannot process types that do not fit into a 32bit integer"); // matching_costs.h:239-240
    if constexpr (std::is_integral_v<T>) {
        return static_cast<uint32_t>(v);
    } else {
        static_assert(sizeof(T) == sizeof(uint32_t), "libstevi_hip: a non-integer census feature is a 32-bit word");
        uint32_t w;
        std::memcpy(&w, &v, sizeof w);
        return w;
    }
}
inline hamming_cv_t popcount32(uint32_t m) {
    m = m - ((m >> 1) & 0x55555555u);
    m = (m & 0x33333333u) + ((m >> 2) & 0x33333333u);
    return static_cast<hamming_cv_t>((((m + (m >> 4)) & 0x0F0F0F0Fu) * 0x01010101u) >> 24);
}
} // namespace detail
} // namespace HipBridge

// dotProduct, matching_costs.h:59-99
template <class T_S, class T_T, class T_O = float, Multidim::ArrayDataAccessConstness viewConstness = Multidim::ConstView>
inline T_O dotProduct(Multidim::Array<T_S, 1, viewConstness> const &source, Multidim::Array<T_T, 1, viewConstness> const &target) {
    return HipBridge::detail::dot<T_O>(source, target);
}
template <class T_S, class T_T, class T_O = float> inline T_O dotProduct(std::vector<T_S> const &source, std::vector<T_T> const &target) {
    return HipBridge::detail::dot<T_O>(source, target);
}
// SumSquareDiff, matching_costs.h:101-137
template <class T_S, class T_T, class T_O = float, Multidim::ArrayDataAccessConstness viewConstness = Multidim::ConstView>
inline T_O SumSquareDiff(Multidim::Array<T_S, 1, viewConstness> const &source, Multidim::Array<T_T, 1, viewConstness> const &target) {
    return HipBridge::detail::squares<T_O>(source, target);
}
template <class T_S, class T_T, class T_O = float> inline T_O SumSquareDiff(std::vector<T_S> const &source, std::vector<T_T> const &target) {
    return HipBridge::detail::squares<T_O>(source, target);
}
// SumAbsDiff, matching_costs.h:139-183
template <class T_S, class T_T, class T_O = float, Multidim::ArrayDataAccessConstness viewConstness = Multidim::ConstView>
inline T_O SumAbsDiff(Multidim::Array<T_S, 1, viewConstness> const &source, Multidim::Array<T_T, 1, viewConstness> const &target) {
    return HipBridge::detail::absolutes<T_O>(source, target);
}
template <class T_S, class T_T, class T_O = float> inline T_O SumAbsDiff(std::vector<T_S> const &source, std::vector<T_T> const &target) {
    return HipBridge::detail::absolutes<T_O>(source, target);
}

// hammingScalar / hammingDistance, matching_costs.h:236-276: popcount of the XOR, summed in uint16 (rule E3)
template <class T_S, class T_T> inline hamming_cv_t hammingScalar(T_S n1, T_T n2) {
    return HipBridge::detail::popcount32(HipBridge::detail::bitsOf(n1) ^ HipBridge::detail::bitsOf(n2));
}
template <class T_S, class T_T, Multidim::ArrayDataAccessConstness viewConstness = Multidim::ConstView>
inline hamming_cv_t hammingDistance(Multidim::Array<T_S, 1, viewConstness> const &source, Multidim::Array<T_T, 1, viewConstness> const &target) {
    hamming_cv_t score = 0;
    for (int i = 0; i < source.shape()[0]; i++) score += hammingScalar(source.valueUnchecked(i), target.valueUnchecked(i));
    return score;
}
template <class T_S, class T_T> inline hamming_cv_t hammingDistance(std::vector<T_S> const &source, std::vector<T_T> const &target) {
    hamming_cv_t score = 0;
    for (std::size_t i = 0; i < source.size(); i++) score += hammingScalar(source[i], target[i]);
    return score;
}

template <matchingFunctions func> class MatchingFunctionTraits {};

// One traits class per function (matching_costs.h:419-685): the constants, and featureComparison = the vector comparison above that
// the function's name stands for.  (As in the reference, only NCC's featureComparison returns T_O; the others return float.)
#define SVH_MATCH_TRAITS(FUNC, NAME, ZM, NORM, STRAT, CENS, RET, COMPARE)                                                    \
    template <> class MatchingFunctionTraits<matchingFunctions::FUNC> {                                                      \
      public:                                                                                                                \
        static constexpr char Name[] = NAME;                                                                                 \
        static constexpr bool ZeroMean = ZM;                                                                                 \
        static constexpr bool Normalized = NORM;                                                                             \
        static constexpr dispExtractionStartegy extractionStrategy = dispExtractionStartegy::STRAT;                          \
        static constexpr bool isCensusBased = CENS;                                                                          \
        template <class T_S, class T_T, class T_O = float, Multidim::ArrayDataAccessConstness viewConstness = Multidim::ConstView> \
        inline static RET featureComparison(Multidim::Array<T_S, 1, viewConstness> const &source,                            \
                                            Multidim::Array<T_T, 1, viewConstness> const &target) {                          \
            return COMPARE;                                                                                                  \
        }                                                                                                                    \
    };
#define SVH_DOT (dotProduct<T_S, T_T, T_O>(source, target))
#define SVH_SSD (SumSquareDiff<T_S, T_T, T_O>(source, target))
#define SVH_SAD (SumAbsDiff<T_S, T_T, T_O>(source, target))
#define SVH_HAM (hammingDistance(source, target))
SVH_MATCH_TRAITS(NCC, "NCC", false, true, Score, false, T_O, SVH_DOT)
SVH_MATCH_TRAITS(CC, "CC", false, false, Score, false, float, SVH_DOT)
SVH_MATCH_TRAITS(SSD, "SSD", false, false, Cost, false, float, SVH_SSD)
SVH_MATCH_TRAITS(SAD, "SAD", false, false, Cost, false, float, SVH_SAD)
SVH_MATCH_TRAITS(ZCC, "ZCC", true, false, Score, false, float, SVH_DOT)
SVH_MATCH_TRAITS(ZNCC, "ZNCC", true, true, Score, false, float, SVH_DOT)
SVH_MATCH_TRAITS(ZSSD, "ZSSD", true, false, Cost, false, float, SVH_SSD)
SVH_MATCH_TRAITS(ZSAD, "ZSAD", true, false, Cost, false, float, SVH_SAD)
SVH_MATCH_TRAITS(HAMMING, "HAMMING", false, false, Cost, true, float, SVH_HAM)
SVH_MATCH_TRAITS(CENSUS, "CENSUS", false, false, Cost, true, float, SVH_HAM)
#undef SVH_DOT
#undef SVH_SSD
#undef SVH_SAD
#undef SVH_HAM
#undef SVH_MATCH_TRAITS

// defaultCvValForMatchFunc / optimalDispAndCost, matching_costs.h:687-723: the neutral starting value of a search and one step of it
template <matchingFunctions func, class T_CV = float> inline static constexpr T_CV defaultCvValForMatchFunc() {
    // (sic) numeric_limits::min() for scores, as the reference has it: the smallest POSITIVE float
    return MatchingFunctionTraits<func>::extractionStrategy == dispExtractionStartegy::Cost ? std::numeric_limits<T_CV>::max() : std::numeric_limits<T_CV>::min();
}
template <matchingFunctions func, class T_Disp, class T_CV = float>
inline std::pair<T_Disp const &, T_CV const &> optimalDispAndCost(T_Disp const &current_disp, T_CV const &current_cost, T_Disp const &candidate_disp,
                                                                  T_CV const &candidate_cost) {
    constexpr bool cost = MatchingFunctionTraits<func>::extractionStrategy == dispExtractionStartegy::Cost;
    const bool better = cost ? (candidate_cost < current_cost) : (candidate_cost > current_cost); // strict: the incumbent keeps ties
    if (better) return {candidate_disp, candidate_cost};
    return {current_disp, current_cost};
}

// MatchingFuncComputeTypeInfos / FeatureTypeForMatchFunc, matching_costs.h:742-788: the feature element type a function computes on
// for a given image type -- the accumulation-extended image type; for uint8 images the samples themselves unless the function
// normalises or removes the mean (int16); 32-bit words for the census-based functions
template <matchingFunctions func, typename ImType> struct MatchingFuncComputeTypeInfos {
    typedef std::conditional_t<func == matchingFunctions::HAMMING || func == matchingFunctions::CENSUS, uint32_t,
                               std::conditional_t<std::is_same_v<ImType, uint8_t>,
                                                  std::conditional_t<MatchingFunctionTraits<func>::Normalized || MatchingFunctionTraits<func>::ZeroMean, int16_t, uint8_t>,
                                                  TypesManipulations::accumulation_extended_t<ImType>>>
        FeatureType;
};
template <matchingFunctions matchFunc, class T_I>
using FeatureTypeForMatchFunc =
    std::conditional_t<MatchingFunctionTraits<matchFunc>::isCensusBased, census_data_t, typename MatchingFuncComputeTypeInfos<matchFunc, T_I>::FeatureType>;

namespace HipBridge {
template <matchingFunctions f> constexpr bool onGpuPath() {
    return f == matchingFunctions::CC || f == matchingFunctions::NCC || f == matchingFunctions::SSD || f == matchingFunctions::SAD ||
           f == matchingFunctions::ZCC || f == matchingFunctions::ZNCC || f == matchingFunctions::ZSSD || f == matchingFunctions::ZSAD ||
           f == matchingFunctions::HAMMING || f == matchingFunctions::CENSUS;
}
} // namespace HipBridge

} // namespace Correlation
} // namespace StereoVision

#endif // STEREOVISION_MATCHING_COSTS_H
