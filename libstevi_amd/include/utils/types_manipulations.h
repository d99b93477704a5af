// Element-type tags of the .stevimg raw array format -- compatibility subset.
//
// Mirrors the part of utils/types_manipulations.h (reference :82-107) that io/image_io.h needs: the "<kind><bits>"
// descriptor written at the head of a .stevimg file ("f32", "u8", "i32", "u32", ...).
#pragma once

#include <string>
#include <type_traits>

namespace StereoVision {
namespace TypesManipulations {

// 'u' unsigned integer, 'i' signed integer, 'f' floating point, followed by the width in bits; "" for other types.
template <typename T> inline std::string dtypeDescr() {
    if (!std::is_integral_v<T> && !std::is_floating_point_v<T>) return "";
    const char kind = std::is_floating_point_v<T> ? 'f' : (std::is_signed_v<T> ? 'i' : 'u');
    return std::string(1, kind) + std::to_string(sizeof(T) * 8);
}

template <typename T> inline bool matchdescr(std::string descr) { return descr == dtypeDescr<T>(); }

} // namespace TypesManipulations
} // namespace StereoVision
