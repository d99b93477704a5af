// Files in, files out: the benchmark chain of the reference (benchmarkCrossCorrelationAlgorithms.cpp:288-294) fed from
// and written to .stevimg files through the drop-in io/image_io.h, the way the reference's tools move arrays around
// (examples/bg_fg_stereo/main.cpp reads its pair with IO::readImage and writes its maps with IO::writeStevimg).
// usage: stevimg_stereo <dir with img_l.stevimg, img_r.stevimg> <out dir> <disp_width>
#include <cstdio>
#include <cstdlib>
#include <string>

#include <correlation/cost_based_refinement.h>
#include <correlation/cross_correlations.h>
#include <correlation/sgm.h>
#include <io/image_io.h>

namespace SC = StereoVision::Correlation;
namespace IO = StereoVision::IO;

int main(int argc, char **argv) {
    if (argc != 4) return 1;
    const std::string in = argv[1], out = argv[2];
    const int D = std::atoi(argv[3]);
    Multidim::Array<float, 3> img_l = IO::readImage<float>(in + "/img_l.stevimg"); // H x W x 1 (rank 2 files)
    Multidim::Array<float, 3> img_r = IO::readImage<float>(in + "/img_r.stevimg");
    if (img_l.empty() || img_r.empty()) return 2;

    constexpr auto matchFunc = SC::matchingFunctions::CENSUS;
    constexpr auto strat = SC::MatchingFunctionTraits<matchFunc>::extractionStrategy;
    Multidim::Array<float, 3> CV = SC::unfoldBasedCostVolume<matchFunc, float, float, 3>(img_l, img_r, 4, 4, D);
    if (CV.empty()) return 3;
    Multidim::Array<float, 3> SGM_CV = SC::sgmCostVolume<8, strat>(CV, 0.001f, 0.01f, StereoVision::Margins(), 100.0f);
    Multidim::Array<SC::disp_t, 2> idx = SC::extractSelectedIndex<strat>(SGM_CV);
    Multidim::Array<float, 3> tcv = SC::truncatedCostVolume(SGM_CV, idx, 4, 4, 1);
    Multidim::Array<float, 2> refined = SC::refineDispCostInterpolation<SC::InterpolationKernel::Parabola>(tcv, idx);

    bool ok = IO::writeStevimg<float, float, 3>(out + "/cost_volume_census.stevimg", CV);
    ok = IO::writeStevimg<float, float, 3>(out + "/sgm8_cost_volume.stevimg", SGM_CV) && ok;
    ok = IO::writeImage<int32_t, SC::disp_t>(out + "/disp_index.stevimg", idx) && ok;
    ok = IO::writeImage<float, float>(out + "/refined_parabola.stevimg", refined) && ok;
    if (!ok) return 4;
    std::printf("ok\n");
    return 0;
}
