// Drop-in for the part of LibStevi's utils/types_manipulations.h the correlation path and io/image_io.h use:
//   accumulation_extended_t<T> and equivalentOneForNormalizing<T>() (reference :45-80) -- which type sums of T are carried in, and what
//       an integer type uses for "1.0" when it is normalised (correlation/matching_costs.h:69, :742-746);
//   dtypeDescr / matchdescr (reference :82-107) -- the "<kind><bits>" tag at the head of a .stevimg file ("f32", "u8", "i32", ...).
#pragma once

#include <cstdint>
#include <string>
#include <type_traits>

namespace StereoVision {
namespace TypesManipulations {

// The type an accumulation of T values is carried in: the next wider SIGNED integer for 8- and 16-bit integers and for uint32,
// the type itself for int32 / 64-bit integers (signed for uint64) and for floating point.
template <class T> struct acc_extended {
    static_assert(std::is_arithmetic_v<T>, "accumulation_extended_t is defined for arithmetic types");
    using type = std::conditional_t<std::is_floating_point_v<T>, T,
                 std::conditional_t<sizeof(T) == 1, int16_t,
                 std::conditional_t<sizeof(T) == 2, int32_t,
                 std::conditional_t<sizeof(T) == 4, std::conditional_t<std::is_signed_v<T>, int32_t, int64_t>, int64_t>>>>;
};
template <class T> using accumulation_extended_t = typename acc_extended<T>::type;

// "1.0" for normalisation in T: 1 for floating point; for an integer type 2^(half its bits) -- a power of two with headroom left
// for the products that follow.
template <class T> inline constexpr T equivalentOneForNormalizing() {
    if constexpr (std::is_integral_v<T>) return static_cast<T>(T(1) << (sizeof(T) * 4));
    else return T(1);
}

// 'u' unsigned integer, 'i' signed integer, 'f' floating point, followed by the width in bits; "" for other types.
template <typename T> inline std::string dtypeDescr() {
    if (!std::is_integral_v<T> && !std::is_floating_point_v<T>) return "";
    const char kind = std::is_floating_point_v<T> ? 'f' : (std::is_signed_v<T> ? 'i' : 'u');
    return std::string(1, kind) + std::to_string(sizeof(T) * 8);
}

template <typename T> inline bool matchdescr(std::string descr) { return descr == dtypeDescr<T>(); }

} // namespace TypesManipulations
} // namespace StereoVision
