// Middlebury optical-flow files (.flo) -- ground truth for the 2-D disparity volumes of the path.
//
// Same name and result as the reference's io/read_flo.h:13-49: magic "PIEH", int32 width, int32 height, then
// height x width (u, v) float pairs, rows first; returned as an H x W x 2 array converted to T.  Anything else
// (missing file, wrong magic, non-positive size, truncated data) gives the empty array.
#pragma once

#include <MultidimArrays/MultidimArrays.h>

#include <cstdint>
#include <cstring>
#include <fstream>
#include <string>

namespace StereoVision {
namespace IO {

template <typename T> Multidim::Array<T, 3> readFloImg(std::string file) {
    std::ifstream in(file, std::ios::binary);
    if (!in.is_open()) return Multidim::Array<T, 3>();
    char magic[4] = {0, 0, 0, 0};
    in.read(magic, 4);
    if (in.gcount() != 4 || std::memcmp(magic, "PIEH", 4) != 0) return Multidim::Array<T, 3>();
    int32_t w = 0, h = 0;
    in.read(reinterpret_cast<char *>(&w), 4);
    in.read(reinterpret_cast<char *>(&h), 4);
    if (!in || w <= 0 || h <= 0) return Multidim::Array<T, 3>();
    Multidim::Array<float, 3> data({h, w, 2}, {2 * w, 2, 1});
    const std::streamsize bytes = static_cast<std::streamsize>(sizeof(float)) * 2 * w * h;
    in.read(reinterpret_cast<char *>(&data.at(0, 0, 0)), bytes); // (the reference: :46)
    if (in.gcount() != bytes) return Multidim::Array<T, 3>();
    return data.template cast<T>();
}

} // namespace IO
} // namespace StereoVision
