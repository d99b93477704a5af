// Drop-in for LibStevi's correlation/matching_costs.h (hot-path subset): the matchingFunctions enum and the compile-time
// traits the callers read (Name, ZeroMean, Normalized, extractionStrategy, isCensusBased); the comparison kernels
// themselves run on the GPU (libstevi_amd/csrc/svh_cost_volume.hip).
#ifndef STEREOVISION_MATCHING_COSTS_H
#define STEREOVISION_MATCHING_COSTS_H

#include "./correlation_base.h"

namespace StereoVision {
namespace Correlation {

enum class matchingFunctions { // matching_costs.h:38-53
    None = -1, CC = 0, NCC = 1, SSD = 2, SAD = 3, ZCC = 4, ZNCC = 5, ZSSD = 6, ZSAD = 7, MEDAD = 8, ZMEDAD = 9, HAMMING = 10, CENSUS = 11, KERMI = 12
};

typedef uint16_t hamming_cv_t;  // :234
typedef uint32_t census_data_t; // :773

template <matchingFunctions func> class MatchingFunctionTraits {};

#define SVH_MATCH_TRAITS(FUNC, NAME, ZM, NORM, STRAT, CENS)                                   \
    template <> class MatchingFunctionTraits<matchingFunctions::FUNC> {                       \
      public:                                                                                 \
        static constexpr char Name[] = NAME;                                                  \
        static constexpr bool ZeroMean = ZM;                                                  \
        static constexpr bool Normalized = NORM;                                              \
        static constexpr dispExtractionStartegy extractionStrategy = dispExtractionStartegy::STRAT; \
        static constexpr bool isCensusBased = CENS;                                           \
    };
// matching_costs.h:419-685
SVH_MATCH_TRAITS(NCC, "NCC", false, true, Score, false)
SVH_MATCH_TRAITS(CC, "CC", false, false, Score, false)
SVH_MATCH_TRAITS(SSD, "SSD", false, false, Cost, false)
SVH_MATCH_TRAITS(SAD, "SAD", false, false, Cost, false)
SVH_MATCH_TRAITS(ZCC, "ZCC", true, false, Score, false)
SVH_MATCH_TRAITS(ZNCC, "ZNCC", true, true, Score, false)
SVH_MATCH_TRAITS(ZSSD, "ZSSD", true, false, Cost, false)
SVH_MATCH_TRAITS(ZSAD, "ZSAD", true, false, Cost, false)
SVH_MATCH_TRAITS(HAMMING, "HAMMING", false, false, Cost, true)
SVH_MATCH_TRAITS(CENSUS, "CENSUS", false, false, Cost, true)
#undef SVH_MATCH_TRAITS

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
