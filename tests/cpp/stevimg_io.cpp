// .stevimg round trips through the drop-in io/image_io.h (host only, no GPU): the call pattern of the reference's
// test/unittests/testImageIO.cpp:40-95 (writeImage -> readImage -> same shape, same elements), plus the strided and
// lower-rank cases of readStevimg.  argv[1] = scratch directory shared with tests/test_stevimg.py:
//   reads  <dir>/from_python_f32.stevimg, <dir>/from_python_cv.stevimg, <dir>/*.flo   (written by libstevi_amd.stevimg)
//   writes <dir>/from_cpp_u16.stevimg, <dir>/from_cpp_cv.stevimg          (read back by the Python side)
#include <io/image_io.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <correlation/stevi_hip_bridge.h>
// raw pointer to an array's elements the way the reference's own code gets one: &a.atUnchecked(0, ...) (io/image_io.h:96, :160)
#define FE(x) StereoVision::Correlation::HipBridge::firstElement(x)


namespace IO = StereoVision::IO;

#define CHECK(cond)                                                                                                    \
    do {                                                                                                               \
        if (!(cond)) {                                                                                                 \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);                                     \
            return 1;                                                                                                  \
        }                                                                                                              \
    } while (0)

template <typename T> static int roundTrip(std::string const &dir, int w, int h, int channels) {
    Multidim::Array<T, 3> img(h, w, channels);
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++)
            for (int c = 0; c < channels; c++) img.atUnchecked(i, j, c) = static_cast<T>((i * 131 + j * 17 + c * 5) % 251);
    const std::string name = dir + "/rt_" + StereoVision::TypesManipulations::dtypeDescr<T>() + "_" + std::to_string(channels) + ".stevimg";
    CHECK((IO::writeImage<T, T>(name, img)));
    CHECK((IO::stevImgFileMatchTypeAndDim<T, 3>(name)));
    Multidim::Array<T, 3> back = IO::readImage<T>(name);
    CHECK(back.shape() == img.shape());
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++)
            for (int c = 0; c < channels; c++) CHECK(back.atUnchecked(i, j, c) == img.atUnchecked(i, j, c));
    return 0;
}

int main(int argc, char **argv) {
    CHECK(argc == 2);
    const std::string dir = argv[1];

    // testImageIO.cpp:112-120 uses 640x480 and 1200x800, 8 and 16 bit, 1 and 3 channels
    CHECK(roundTrip<uint8_t>(dir, 640, 480, 1) == 0);
    CHECK(roundTrip<uint8_t>(dir, 640, 480, 3) == 0);
    CHECK(roundTrip<uint16_t>(dir, 1200, 800, 3) == 0);
    CHECK(roundTrip<float>(dir, 33, 9, 1) == 0);
    CHECK(roundTrip<int32_t>(dir, 7, 5, 2) == 0);

    // element type and rank are checked against the header
    CHECK((IO::readStevimg<float, 3>(dir + "/rt_u8_1.stevimg").empty()));
    CHECK((!IO::stevImgFileMatchTypeAndDim<float, 3>(dir + "/rt_u8_1.stevimg")));
    CHECK((IO::readStevimg<uint8_t, 2>(dir + "/rt_u8_1.stevimg").empty())); // rank 3 file into a rank 2 array
    CHECK((IO::readImage<uint8_t>(dir + "/missing.stevimg").empty()));
    CHECK((IO::readImage<uint8_t>(dir + "/picture.png").empty()));           // codecs: out of scope, reported as failure
    CHECK((!IO::writeImage<uint8_t, uint8_t>(dir + "/picture.png", Multidim::Array<uint8_t, 3>(2, 2, 1))));
    CHECK((!IO::writeImage<uint8_t, uint8_t>(dir + "/empty.stevimg", Multidim::Array<uint8_t, 3>())));

    // conversion on write (ImgType != InType) and a rank 2 file read as rank 3 (trailing axis of extent 1)
    Multidim::Array<float, 2> disp(4, 6);
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 6; j++) disp.atUnchecked(i, j) = static_cast<float>(i * 6 + j) + 0.75f;
    CHECK((IO::writeImage<uint16_t, float>(dir + "/from_cpp_u16.stevimg", disp)));
    Multidim::Array<uint16_t, 3> disp3 = IO::readImage<uint16_t>(dir + "/from_cpp_u16.stevimg");
    CHECK((disp3.shape() == std::array<int, 3>{4, 6, 1}));
    CHECK(disp3.atUnchecked(3, 5, 0) == 23);

    // a cost volume in the reference's layout {W*D, 1, W} (cross_correlations.h:220) keeps it
    const int H = 3, W = 5, D = 4;
    Multidim::Array<float, 3> cv({H, W, D}, {W * D, 1, W});
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++)
            for (int d = 0; d < D; d++) cv.atUnchecked(i, j, d) = static_cast<float>(100 * i + 10 * j + d);
    CHECK((IO::writeStevimg<float, float, 3>(dir + "/from_cpp_cv.stevimg", cv)));
    Multidim::Array<float, 3> cvBack = IO::readStevimg<float, 3>(dir + "/from_cpp_cv.stevimg");
    CHECK(cvBack.strides() == cv.strides());
    CHECK(cvBack.atUnchecked(2, 4, 3) == 243.0f);

    // a view with holes is written as a dense copy
    Multidim::Array<float, 2> holes(FE(cv), {H, W}, {W * D, 2}, false);
    CHECK((IO::writeStevimg<float, float, 2>(dir + "/holes.stevimg", holes)));
    Multidim::Array<float, 2> holesBack = IO::readStevimg<float, 2>(dir + "/holes.stevimg");
    CHECK((holesBack.strides() == std::array<int, 2>{W, 1}));
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) CHECK(holesBack.atUnchecked(i, j) == holes.valueUnchecked(i, j));

    // files written by libstevi_amd.stevimg
    Multidim::Array<float, 3> py = IO::readImage<float>(dir + "/from_python_f32.stevimg");
    CHECK((py.shape() == std::array<int, 3>{5, 7, 1})); // a rank 2 file
    double sum = 0;
    for (int i = 0; i < 5; i++)
        for (int j = 0; j < 7; j++) sum += py.atUnchecked(i, j, 0) * (i + 1) * (j + 2);
    Multidim::Array<float, 3> pycv = IO::readStevimg<float, 3>(dir + "/from_python_cv.stevimg");
    CHECK((pycv.shape() == std::array<int, 3>{H, W, D}));
    CHECK((pycv.strides() == std::array<int, 3>{W * D, 1, W}));
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++)
            for (int d = 0; d < D; d++) CHECK(pycv.atUnchecked(i, j, d) == static_cast<float>(100 * i + 10 * j + d));
    // Middlebury .flo through readImage (image_io.cpp:106-109): H x W x 2, converted to the requested element type
    Multidim::Array<float, 3> flo = IO::readImage<float>(dir + "/from_python.flo");
    CHECK((flo.shape() == std::array<int, 3>{4, 6, 2}));
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 6; j++) {
            CHECK(flo.atUnchecked(i, j, 0) == 0.5f * j - i);
            CHECK(flo.atUnchecked(i, j, 1) == 0.25f * i + j);
        }
    Multidim::Array<int32_t, 3> floInt = IO::readFloImg<int32_t>(dir + "/from_python.flo");
    CHECK(floInt.atUnchecked(3, 5, 1) == 5); // 0.75 + 5 truncated by the cast
    CHECK((IO::readFloImg<float>(dir + "/bad_magic.flo").empty()));
    CHECK((IO::readFloImg<float>(dir + "/truncated.flo").empty()));
    CHECK((IO::readFloImg<float>(dir + "/missing.flo").empty()));
    std::printf("stevimg ok weighted_sum=%.6f\n", sum);
    return 0;
}
