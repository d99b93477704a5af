// .stevimg array files -- the data format either side of the correlation/ path (images in, disparity maps, cost
// volumes and fixtures out).
//
// Same names and behaviour as the reference's io/image_io.h for this format (stevImgFileMatchTypeAndDim :15-46,
// writeStevimg :48-108, readStevimg :110-168, readImage / writeImage image_io.cpp:96-225):
//
//     "<dtype> <nDim> <shape_0 .. shape_{n-1}> <stride_0 .. stride_{n-1}>\n"      one text line
//     flatLenght() elements of raw data, exactly as they lie in memory under those strides
//
// so a volume written with the cost-volume strides {W*D, 1, W} comes back with them.  The image codecs behind the
// reference's readImage (CImg, OpenEXR) are third-party and outside the hot path: here readImage serves ".stevimg" and
// ".flo" (read_flo.h), writeImage ".stevimg", and both report failure (empty array / false) for every other extension.
#pragma once

#include "../utils/types_manipulations.h"
#include "read_flo.h"

#include <MultidimArrays/MultidimArrays.h>

#include <algorithm>
#include <cstdio>
#include <fstream>
#include <numeric>
#include <sstream>
#include <string>
#include <vector>

namespace StereoVision {
namespace IO {

namespace detail {

inline bool hasExtension(std::string const &fileName, std::string const &ext) {
    return fileName.size() >= ext.size() && fileName.compare(fileName.size() - ext.size(), ext.size(), ext) == 0;
}
inline bool hasStevimgExtension(std::string const &fileName) { return hasExtension(fileName, ".stevimg"); }

// true when the elements fill a block of flatLenght() elements without holes or overlap (any axis order)
template <typename T, int nDim> bool fillsItsSpan(Multidim::Array<T, nDim> const &a) {
    std::array<int, nDim> order;
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](int x, int y) { return a.strides()[x] < a.strides()[y]; });
    long expected = 1;
    for (int k : order) {
        if (a.shape()[k] == 1) continue;
        if (a.strides()[k] != expected) return false;
        expected *= a.shape()[k];
    }
    return true;
}

} // namespace detail

template <typename ImgType, int nDim> bool stevImgFileMatchTypeAndDim(std::string const &fileName) {
    std::ifstream in(fileName, std::ios_base::in | std::ios_base::binary);
    if (!in.is_open()) return true; // as the reference: only a readable header can contradict the request (:44)
    std::string line;
    std::getline(in, line);
    std::istringstream head(line);
    std::string type;
    int nDimInFile = 0;
    head >> type >> nDimInFile;
    return type == TypesManipulations::dtypeDescr<ImgType>() && nDimInFile <= nDim;
}

template <typename ImgType, typename InType, int nDim> bool writeStevimg(std::string const &fileName, Multidim::Array<InType, nDim> const &image) {
    if constexpr (!std::is_same_v<ImgType, InType>) {
        return writeStevimg<ImgType, ImgType, nDim>(fileName, image.template cast<ImgType>());
    } else {
        if (!image.empty() && !detail::fillsItsSpan(image)) {
            return writeStevimg<ImgType, ImgType, nDim>(fileName, image.template cast<ImgType>()); // dense copy, last index fastest
        }
        std::FILE *out = std::fopen(fileName.c_str(), "wb");
        if (!out) return false;
        std::ostringstream head;
        head << TypesManipulations::dtypeDescr<ImgType>() << ' ' << nDim;
        for (int k = 0; k < nDim; k++) head << ' ' << image.shape()[k];
        for (int k = 0; k < nDim; k++) head << ' ' << image.strides()[k];
        head << '\n';
        const std::string h = head.str();
        bool ok = std::fwrite(h.data(), 1, h.size(), out) == h.size();
        const std::size_t n = image.flatLenght();
        if (n > 0 && !image.empty()) ok = ok && std::fwrite(&const_cast<Multidim::Array<InType, nDim> *>(&image)->atUnchecked(0), sizeof(ImgType), n, out) == n; // (as the reference: :96)
        ok = (std::fclose(out) == 0) && ok;
        return ok;
    }
}

template <typename ImgType, int nDim> Multidim::Array<ImgType, nDim> readStevimg(std::string const &fileName) {
    using ArrayT = Multidim::Array<ImgType, nDim>;
    std::ifstream in(fileName, std::ios_base::in | std::ios_base::binary);
    if (!in.is_open()) return ArrayT();
    std::string line;
    std::getline(in, line);
    std::istringstream head(line);
    std::string type;
    int nDimInFile = 0;
    head >> type >> nDimInFile;
    if (type != TypesManipulations::dtypeDescr<ImgType>() || nDimInFile > nDim || nDimInFile < 0) return ArrayT();
    typename ArrayT::ShapeBlock shape, strides;
    // a file of lower rank is read as an array with trailing axes of extent 1 (reference :139-153)
    for (int k = 0; k < nDim; k++) {
        shape[k] = 1;
        if (k < nDimInFile) head >> shape[k];
    }
    for (int k = 0; k < nDim; k++) {
        strides[k] = 1;
        if (k < nDimInFile) head >> strides[k];
    }
    if (head.fail()) return ArrayT();
    ArrayT img(shape, strides);
    if (img.flatLenght() > 0) {
        in.read(reinterpret_cast<char *>(&img.atUnchecked(0)), static_cast<std::streamsize>(sizeof(ImgType) * img.flatLenght())); // (:160)
        if (static_cast<std::size_t>(in.gcount()) != sizeof(ImgType) * img.flatLenght()) return ArrayT(); // truncated file
    }
    return img;
}

template <typename ImgType> Multidim::Array<ImgType, 3> readImage(std::string const &fileName) {
    if (detail::hasStevimgExtension(fileName)) return readStevimg<ImgType, 3>(fileName);
    if (detail::hasExtension(fileName, ".flo")) return readFloImg<ImgType>(fileName); // image_io.cpp:106-109
    return Multidim::Array<ImgType, 3>(); // codecs are out of scope here
}

template <typename ImgType, typename InType> bool writeImage(std::string const &fileName, Multidim::Array<InType, 3> const &image) {
    if (image.empty()) return false;
    if (detail::hasStevimgExtension(fileName)) return writeStevimg<ImgType, InType, 3>(fileName, image);
    return false;
}

template <typename ImgType, typename InType> bool writeImage(std::string const &fileName, Multidim::Array<InType, 2> const &image) {
    if (image.empty()) return false;
    if (detail::hasStevimgExtension(fileName)) return writeStevimg<ImgType, InType, 2>(fileName, image);
    return false;
}

} // namespace IO
} // namespace StereoVision
