// The reference benchmark's call chain written against the drop-in headers, with the reference's own names
// (test/benchmarks/benchmarkCrossCorrelationAlgorithms.cpp:92-96, :288-294; examples/stereo_refine_test/main.cpp:367-384).
// usage: dropin_chain <H> <W> <D> <left.f32> <right.f32> <out_prefix>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include <correlation/census.h>
#include <correlation/cost_based_refinement.h>
#include <correlation/cross_correlations.h>
#include <correlation/hierarchical.h>
#include <correlation/patchmatch.h>
#include <utils/randomcache.h>
#include <correlation/sgm.h>

namespace SC = StereoVision::Correlation;
// raw pointer to an array's elements the way the reference's own code gets one: &a.atUnchecked(0, ...) (io/image_io.h:96, :160)
#define FE(x) StereoVision::Correlation::HipBridge::firstElement(x)


template <class T> static void dump(std::string const &path, T const *p, size_t n) {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f || fwrite(p, sizeof(T), n, f) != n) {
        fprintf(stderr, "cannot write %s\n", path.c_str());
        exit(2);
    }
    fclose(f);
}

static Multidim::Array<float, 2> load(const char *path, int H, int W) {
    Multidim::Array<float, 2> img(H, W);
    FILE *f = fopen(path, "rb");
    if (!f || fread(FE(img), sizeof(float), (size_t)H * W, f) != (size_t)H * W) {
        fprintf(stderr, "cannot read %s\n", path);
        exit(2);
    }
    fclose(f);
    return img;
}

int main(int argc, char **argv) {
    if (argc != 7) return 1;
    const int H = atoi(argv[1]), W = atoi(argv[2]), D = atoi(argv[3]);
    Multidim::Array<float, 2> target = load(argv[4], H, W), source = load(argv[5], H, W);
    const std::string out = argv[6];
    const uint8_t h_r = 4, v_r = 4;
    const float P1 = 0.001f, P2 = 0.01f, Pout = 100;

    { // census + SGM-8 + argmin (Cost)
        constexpr auto matchFunc = SC::matchingFunctions::CENSUS;
        constexpr auto strat = SC::MatchingFunctionTraits<matchFunc>::extractionStrategy;
        Multidim::Array<float, 3> CV = SC::unfoldBasedCostVolume<matchFunc>(target, source, h_r, v_r, D);
        Multidim::Array<float, 3> SGM_CV = SC::sgmCostVolume<8, strat>(CV, P1, P2, StereoVision::Margins(), Pout);
        Multidim::Array<SC::disp_t, 2> disp =
            SC::selectedIndexToDisp<SC::disp_t, SC::dispDirection::RightToLeft>(SC::extractSelectedIndex<strat>(SGM_CV), 0);
        dump(out + "_census_sgm.f32", FE(SGM_CV), SGM_CV.flatLenght());
        dump(out + "_census_disp.i32", FE(disp), disp.flatLenght());
        { // T_CV = int16_t (sgm.h:360 is a template on the volume's element type; the reference casts what it reads to float): same bits
            Multidim::Array<int16_t, 3> CV16(CV.shape());
            for (int i = 0; i < CV.shape()[0]; i++)
                for (int j = 0; j < CV.shape()[1]; j++)
                    for (int d = 0; d < CV.shape()[2]; d++) CV16.atUnchecked(i, j, d) = static_cast<int16_t>(CV.valueUnchecked(i, j, d));
            Multidim::Array<float, 3> SGM16 = SC::sgmCostVolume<8, strat>(CV16, P1, P2, StereoVision::Margins(), Pout);
            if (SGM16.flatLenght() != SGM_CV.flatLenght() || std::memcmp(FE(SGM16), FE(SGM_CV), sizeof(float) * SGM_CV.flatLenght()) != 0) {
                std::fprintf(stderr, "sgmCostVolume<int16_t> differs from sgmCostVolume<float>\n");
                return 3;
            }
        }
        // the same chain with the volumes left in GPU memory (HipBridge::DeviceArray): the reference's names, overloads picked by
        // the argument type; images up once, one disparity map down
        auto dCV = SC::unfoldBasedCostVolumeOnDevice<matchFunc>(target, source, h_r, v_r, D);
        auto dSGM = SC::sgmCostVolume<8, strat>(dCV, P1, P2, StereoVision::Margins(), Pout);
        // dSGM left sgmCostVolume with the winner the kernel that wrote it picked on the way (Statement::winner): extractSelectedIndex
        // returns a copy of that map instead of scanning the volume.  Same indices as the scan, which runs once the statement is gone.
        if (!dSGM.statement() || !dSGM.statement()->winner) {
            fprintf(stderr, "sgmCostVolume<Cost> on a DeviceArray left no winner statement\n");
            return 6;
        }
        auto dIdx = SC::extractSelectedIndex<strat>(dSGM);
        {
            auto dSGMscan = dSGM;
            Multidim::Array<SC::disp_t, 2> byStatement = dIdx.download();
            (void)dSGMscan.data(); // a pointer the volume could be written through: the statement is dropped for every handle of the storage
            if (dSGM.statement()) return 6;
            Multidim::Array<SC::disp_t, 2> byScan = SC::extractSelectedIndex<strat>(dSGM).download();
            for (int i = 0; i < byScan.shape()[0]; i++)
                for (int j = 0; j < byScan.shape()[1]; j++)
                    if (byScan.valueUnchecked(i, j) != byStatement.valueUnchecked(i, j)) {
                        fprintf(stderr, "winner statement differs from the scan at (%d, %d): %d vs %d\n", i, j, byStatement.valueUnchecked(i, j), byScan.valueUnchecked(i, j));
                        return 6;
                    }
        }
        { // a Score-strategy cost volume leaves unfoldBasedCostVolumeOnDevice with ITS winner (svh_unfold_cost_volume_winner): the reference
          // benchmark's own sequence, unfoldBasedCostVolume -> extractSelectedIndex, returns a copy of that map; the scan gives the same indices
            constexpr auto ncc = SC::matchingFunctions::NCC;
            constexpr auto nstrat = SC::MatchingFunctionTraits<ncc>::extractionStrategy;
            auto dN = SC::unfoldBasedCostVolumeOnDevice<ncc>(target, source, h_r, v_r, D);
            if (!dN.statement() || !dN.statement()->winner) {
                fprintf(stderr, "unfoldBasedCostVolumeOnDevice<NCC> left no winner statement\n");
                return 7;
            }
            Multidim::Array<SC::disp_t, 2> byStatement = SC::extractSelectedIndex<nstrat>(dN).download();
            dN.dropStatement();
            Multidim::Array<SC::disp_t, 2> byScan = SC::extractSelectedIndex<nstrat>(dN).download();
            for (int i = 0; i < byScan.shape()[0]; i++)
                for (int j = 0; j < byScan.shape()[1]; j++)
                    if (byScan.valueUnchecked(i, j) != byStatement.valueUnchecked(i, j)) {
                        fprintf(stderr, "cost-volume winner statement differs from the scan at (%d, %d)\n", i, j);
                        return 7;
                    }
        }
        Multidim::Array<SC::disp_t, 2> ddisp = SC::selectedIndexToDisp<SC::disp_t, SC::dispDirection::RightToLeft>(dIdx, 0).download();
        Multidim::Array<float, 2> dref =
            SC::refineDispCostInterpolation<SC::InterpolationKernel::Parabola>(SC::truncatedCostVolume(dSGM, dIdx, h_r, v_r, 1), dIdx).download();
        dump(out + "_census_disp_dev.i32", FE(ddisp), ddisp.flatLenght());
        // dCV came with the library's statement about its contents (regional minima: sgmCostVolume<Cost> skipped its probe pass above); a
        // pointer the volume could be written through drops it -- for EVERY handle of that memory, the copy the pointer was asked of and the
        // original alike (the statement is about the storage) -- and the call takes the probing path: the same map either way.  const
        // access (data() const, share()) keeps it.
        auto dCVt = dCV;
        const bool had = static_cast<bool>(dCVt.statement());
        auto const &dCVc = dCV;
        (void)dCVc.data();
        (void)dCVc.share();
        const bool kept_by_const_access = static_cast<bool>(dCV.statement());
        (void)dCVt.data();
        if (!had || !kept_by_const_access || dCVt.statement() || dCV.statement()) {
            fprintf(stderr, "DeviceArray statement: attached %d, kept by const access %d, after a mutable access through a copy: copy %d original %d\n", (int)had,
                    (int)kept_by_const_access, (int)static_cast<bool>(dCVt.statement()), (int)static_cast<bool>(dCV.statement()));
            return 5;
        }
        Multidim::Array<SC::disp_t, 2> tdisp = SC::selectedIndexToDisp<SC::disp_t, SC::dispDirection::RightToLeft>(
            SC::extractSelectedIndex<strat>(SC::sgmCostVolume<8, strat>(dCVt, P1, P2, StereoVision::Margins(), Pout)), 0).download();
        dump(out + "_census_disp_dev_touched.i32", FE(tdisp), tdisp.flatLenght());
        dump(out + "_census_ref_dev.f32", FE(dref), dref.flatLenght());
        Multidim::Array<float, 2> href = SC::refineDispCostInterpolation<SC::InterpolationKernel::Parabola>(
            SC::truncatedCostVolume(SGM_CV, SC::extractSelectedIndex<strat>(SGM_CV), h_r, v_r, 1), SC::extractSelectedIndex<strat>(SGM_CV));
        dump(out + "_census_ref_host.f32", FE(href), href.flatLenght());
        // re-entrancy: the reference's functions are stateless; here every thread gets a context of its own (stevi_hip_bridge.h).
        // Three threads run the chain at once on the same inputs and must each reproduce the map above.
        int mismatches[3] = {-1, -1, -1};
        std::vector<std::thread> pool;
        for (int t = 0; t < 3; t++)
            pool.emplace_back([&, t] {
                auto cv = SC::unfoldBasedCostVolume<matchFunc>(target, source, h_r, v_r, D);
                auto d = SC::selectedIndexToDisp<SC::disp_t, SC::dispDirection::RightToLeft>(
                    SC::extractSelectedIndex<strat>(SC::sgmCostVolume<8, strat>(cv, P1, P2, StereoVision::Margins(), Pout)), 0);
                int bad = 0;
                for (size_t e = 0; e < d.flatLenght(); e++) bad += FE(d)[e] != FE(disp)[e];
                mismatches[t] = bad;
            });
        for (auto &th : pool) th.join();
        for (int t = 0; t < 3; t++)
            if (mismatches[t] != 0) {
                fprintf(stderr, "thread %d: %d pixels differ from the single-threaded result\n", t, mismatches[t]);
                return 3;
            }
    }
    { // a DeviceArray outlives the thread (and the per-thread context) that made it: a worker hands its volume to the main thread and
      // exits; the main thread finishes the chain on it and lets it go (freed by device, not through the dead context)
        constexpr auto matchFunc = SC::matchingFunctions::CENSUS;
        constexpr auto strat = SC::MatchingFunctionTraits<matchFunc>::extractionStrategy;
        SC::HipBridge::DeviceArray<float, 3> handed;
        std::thread worker([&] { handed = SC::sgmCostVolume<8, strat>(SC::unfoldBasedCostVolumeOnDevice<matchFunc>(target, source, h_r, v_r, D), P1, P2, StereoVision::Margins(), Pout); });
        worker.join(); // the worker's thread-local context is destroyed here
        Multidim::Array<SC::disp_t, 2> late = SC::selectedIndexToDisp<SC::disp_t, SC::dispDirection::RightToLeft>(SC::extractSelectedIndex<strat>(handed), 0).download();
        dump(out + "_census_disp_handed.i32", FE(late), late.flatLenght());
        handed = SC::HipBridge::DeviceArray<float, 3>(); // released on this thread
    }
    { // inputs that are views: a ConstView image (the shims copy what they cannot address: stevi_hip_bridge.h) and a strided view of a
      // larger image (subView of columns: strides travel to the C ABI as they are)
        constexpr auto matchFunc = SC::matchingFunctions::CENSUS;
        Multidim::Array<float, 2, Multidim::ConstView> ctarget = target, csource = source;
        // unfold takes any view constness (unfold.h:247-250), censusFeatures too (census.h:69-70)
        Multidim::Array<float, 3> feat_c = SC::unfold<float, float>(h_r, v_r, ctarget), feat_n = SC::unfold<float, float>(h_r, v_r, target);
        Multidim::Array<float, 3, Multidim::ConstView> cfeat = feat_n;
        Multidim::Array<SC::census_data_t, 3> words_c = SC::censusFeatures(cfeat), words_n = SC::censusFeatures(feat_n);
        size_t bad = words_c.flatLenght() != words_n.flatLenght() || feat_c.flatLenght() != feat_n.flatLenght() || csource.flatLenght() != source.flatLenght();
        for (size_t e = 0; e < words_n.flatLenght() && !bad; e++) bad += FE(words_c)[e] != FE(words_n)[e];
        for (size_t e = 0; e < feat_n.flatLenght() && !bad; e++) bad += FE(feat_c)[e] != FE(feat_n)[e];
        Multidim::Array<float, 2> wide(H, W + 7);
        for (int i = 0; i < H; i++)
            for (int j = 0; j < W + 7; j++) wide.at(i, j) = j >= 3 && j < W + 3 ? source.value(i, j - 3) : -5.f;
        Multidim::Array<float, 2> inner = wide.subView(Multidim::DimSlice(), Multidim::DimSlice(3, W + 3));
        auto cv_view = SC::unfoldBasedCostVolume<matchFunc>(target, inner, h_r, v_r, D);
        auto cv_own = SC::unfoldBasedCostVolume<matchFunc>(target, source, h_r, v_r, D);
        for (size_t e = 0; e < cv_own.flatLenght() && !bad; e++) bad += FE(cv_view)[e] != FE(cv_own)[e];
        if (bad) {
            fprintf(stderr, "view inputs: results differ from the owning arrays'\n");
            return 4;
        }
    }
    { // NCC + SGM-8 (Score) + parabola refinement
        constexpr auto matchFunc = SC::matchingFunctions::NCC;
        constexpr auto strat = SC::MatchingFunctionTraits<matchFunc>::extractionStrategy;
        auto CV = SC::unfoldBasedCostVolume<matchFunc>(target, source, h_r, v_r, D);
        auto SGM_CV = SC::sgmCostVolume<8, strat>(CV, P1, P2, StereoVision::Margins(), Pout);
        auto rawDisp = SC::extractSelectedIndex<strat>(SGM_CV);
        auto tcv = SC::truncatedCostVolume(SGM_CV, rawDisp, h_r, v_r, 1);
        auto refined = SC::refineDispCostInterpolation<SC::InterpolationKernel::Parabola>(tcv, rawDisp);
        dump(out + "_ncc_cv.f32", FE(CV), CV.flatLenght());
        dump(out + "_ncc_idx.i32", FE(rawDisp), rawDisp.flatLenght());
        dump(out + "_ncc_refined.f32", FE(refined), refined.flatLenght());
        // the same on the device: the Score branch's winner (records of the banded sweep from 256 disparities on, a scan of the volume the call
        // has just written below that) becomes the statement: the map must equal the host chain's
        auto dCV = SC::DeviceArray<float, 3>::upload(CV);
        auto dS = SC::sgmCostVolume<8, strat>(dCV, P1, P2, StereoVision::Margins(), Pout);
        Multidim::Array<SC::disp_t, 2> dRaw = SC::extractSelectedIndex<strat>(dS).download();
        for (int i = 0; i < dRaw.shape()[0]; i++)
            for (int j = 0; j < dRaw.shape()[1]; j++)
                if (dRaw.valueUnchecked(i, j) != rawDisp.valueUnchecked(i, j)) {
                    fprintf(stderr, "Score-branch device chain differs from the host chain at (%d, %d)\n", i, j);
                    return 7;
                }
    }
    { // 2-D disparity volume + winner, as test/unittests/testCorrelation2d.cpp:166-175 chains them
        constexpr auto matchFunc = SC::matchingFunctions::ZNCC;
        constexpr auto strat = SC::MatchingFunctionTraits<matchFunc>::extractionStrategy;
        SC::searchOffset<2> searchRange(-1, 2, -2, 3);
        Multidim::Array<float, 4> CV2 = SC::unfoldBased2dDisparityCostVolume<matchFunc>(target, source, 2, 2, searchRange);
        Multidim::Array<SC::disp_t, 3> disp2 = SC::selected2dIndexToDisp(SC::extractSelected2dIndex<strat>(CV2), searchRange);
        auto tcv2 = SC::truncatedBidirectionaCostVolume(CV2, SC::extractSelected2dIndex<strat>(CV2), 1, 1);
        dump(out + "_zncc2d_cv.f32", FE(CV2), CV2.flatLenght());
        dump(out + "_zncc2d_disp.i32", FE(disp2), disp2.flatLenght());
        if (tcv2.shape()[2] != 3 || tcv2.shape()[3] != 3) return 4;
        // feature-volume form, as testOnDemandCostVolume.cpp:139 calls it
        auto fl = SC::unfold<float, float>(2, 2, target), fr = SC::unfold<float, float>(2, 2, source);
        Multidim::Array<float, 4> CV2f = SC::featureVolume2CostVolume<matchFunc, float, float, SC::searchOffset<2>, SC::dispDirection::RightToLeft, float>(fl, fr, searchRange);
        if (CV2f.shape()[2] != CV2.shape()[2] || CV2f.shape()[3] != CV2.shape()[3]) return 6;
        dump(out + "_zncc2d_cv_feat.f32", FE(CV2f), CV2f.flatLenght());
        // the refinement examples/stereo-match --refine applies (main.cpp:198-210), plus the two other 2-D variants
        auto ref_iso = SC::refineDisp2dCostInterpolation<SC::InterpolationKernel::Equiangular>(tcv2, disp2);
        auto ref_aniso = SC::refineDisp2dCostInterpolation<SC::InterpolationKernel::Parabola, SC::IsotropyHypothesis::Anisotropic>(tcv2, disp2);
        auto ref_patch = SC::refineDisp2dCostPatchInterpolation<SC::InterpolationKernel::Parabola>(tcv2, disp2);
        dump(out + "_zncc2d_ref_iso.f32", FE(ref_iso), ref_iso.flatLenght());
        dump(out + "_zncc2d_ref_aniso.f32", FE(ref_aniso), ref_aniso.flatLenght());
        dump(out + "_zncc2d_ref_patch.f32", FE(ref_patch), ref_patch.flatLenght());
    }
    { // benchmarkCrossCorrelationAlgorithms.cpp:198-233: hiearchicalTruncatedCostVolume<matchFunc, depth>(target, source, h_r, v_r, disp_w)
        constexpr auto matchFunc = SC::matchingFunctions::ZNCC;
        SC::OffsetedCostVolume<float> result = SC::hiearchicalTruncatedCostVolume<matchFunc, 2>(target, source, uint8_t(2), uint8_t(2), D);
        if (result.disp_estimate.shape()[0] != H || result.disp_estimate.shape()[1] != W) return 5;
        dump(out + "_hier_disp.i32", FE(result.disp_estimate), result.disp_estimate.flatLenght());
        dump(out + "_hier_tcv.f32", FE(result.truncated_cost_volume), result.truncated_cost_volume.flatLenght());
        Multidim::Array<float, 2> half = StereoVision::Interpolation::averagePoolingDownsample(source, StereoVision::Interpolation::DownSampleWindows(2));
        dump(out + "_half.f32", FE(half), half.flatLenght());
    }
    { // examples/stereo_refine_test/main.cpp:386-398: statistics and feature transforms called directly
        auto feats = SC::unfold<float, float>(1, 1, source);
        Multidim::Array<float, 2> mean = SC::channelsMean<float, float>(feats);
        Multidim::Array<float, 3> zm = SC::zeromeanFeatureVolume<float, float, float>(feats, mean);
        Multidim::Array<float, 3> proc = SC::getFeatureVolumeForMatchFunc<SC::matchingFunctions::ZNCC>(feats);
        Multidim::Array<uint32_t, 3> words = SC::getFeatureVolumeForMatchFunc<SC::matchingFunctions::CENSUS>(SC::unfold<float, float>(3, 3, source));
        Multidim::Array<float, 2> sigma = SC::channelsZeroMeanNorm<float, float, float>(feats, mean);
        if (words.shape()[2] != 2 || sigma.shape()[0] != H) return 7;
        dump(out + "_mean.f32", FE(mean), mean.flatLenght());
        dump(out + "_zm.f32", FE(zm), zm.flatLenght());
        dump(out + "_zncc_feat.f32", FE(proc), proc.flatLenght());
        dump(out + "_words.u32", FE(words), words.flatLenght());
    }
    { // benchmarkCrossCorrelationAlgorithms.cpp:140-160: the compressor overload with one of the shipped masks
        SC::UnFoldCompressor compressor(SC::CompressorGenerators::GrPix17R3Filter());
        if (compressor.nFeatures() != 17 || compressor.width() != 7 || compressor.height() != 7) return 8;
        Multidim::Array<float, 3> CVc = SC::unfoldBasedCostVolume<SC::matchingFunctions::ZNCC>(target, source, compressor, D);
        if (CVc.shape()[0] != H || CVc.shape()[1] != W || CVc.shape()[2] != D) return 9;
        dump(out + "_compressed_cv.f32", FE(CVc), CVc.flatLenght());
    }
    { // examples/stereo-match/main.cpp:140-210 with its own types: on-demand ZN features, cachelessPatchMatch, --refine chain
        using namespace StereoVision::Correlation;
        Multidim::Array<float, 3> img_source(H, W, 1), img_target(H, W, 1);
        for (int i = 0; i < H; i++)
            for (int j = 0; j < W; j++) {
                img_source.atUnchecked(i, j, 0) = source.valueUnchecked(i, j);
                img_target.atUnchecked(i, j, 0) = target.valueUnchecked(i, j);
            }
        const int searchWindowRadius = 2, nChannels = 1;
        std::vector<std::array<int, 3>> featuresWindow;
        for (int i = -searchWindowRadius; i <= searchWindowRadius; i++)
            for (int j = -searchWindowRadius; j <= searchWindowRadius; j++)
                for (int c = 0; c < nChannels; c++) featuresWindow.push_back({i, j, c});
        constexpr matchingFunctions matchFunc = matchingFunctions::ZNCC;
        constexpr bool ZeroMean = MatchingFunctionTraits<matchFunc>::ZeroMean;
        constexpr bool Normalized = MatchingFunctionTraits<matchFunc>::Normalized;
        constexpr Multidim::ArrayDataAccessConstness constness = Multidim::ArrayDataAccessConstness::NonConstView;
        using OnDemandFeaturesT = OnDemandDecoratedFeaturesVolume<ZNFeaturesVolumeDecorator<ZeroMean, Normalized>, float, 3, constness, 2>;
        static_assert(OnDemandFeaturesT::nOutDim == 2);
        searchOffset<2> searchRegion(-2, 2, 0, 12);
        OnDemandFeaturesT features_source(featuresWindow, img_source);
        OnDemandFeaturesT features_target(featuresWindow, img_target);
        HipBridge::patchMatchSeed() = 99; // reproducible run for the checker
        Multidim::Array<disp_t, 3> pm = cachelessPatchMatch<matchFunc, 2>(features_source, features_target, searchRegion, 6, 4, std::nullopt);
        if (pm.shape()[0] != H || pm.shape()[1] != W || pm.shape()[2] != 2) return 10;
        using CostVolT = CachelessOnDemandImageFlowVolume<matchFunc, float, OnDemandFeaturesT, OnDemandFeaturesT>;
        using SearchSpaceT = typename CostVolT::SearchSpaceType;
        SearchSpaceT searchSpace(SearchSpaceBase::SearchDim(searchRegion.template lowerOffset<0>(), searchRegion.template upperOffset<0>()),
                                 SearchSpaceBase::SearchDim(searchRegion.template lowerOffset<1>(), searchRegion.template upperOffset<1>()),
                                 SearchSpaceBase::FeatureDim());
        CostVolT onDemandCV(features_source, features_target, searchSpace);
        Multidim::Array<float, 3> refinedDisp = refineDisp2dCostInterpolation<InterpolationKernel::Equiangular>(onDemandCV.truncatedCostVolume(pm), pm);
        if (refinedDisp.shape()[2] != 2) return 11;
        dump(out + "_pm_disp.i32", FE(pm), pm.flatLenght());
        dump(out + "_pm_refined.f32", FE(refinedDisp), refinedDisp.flatLenght());
    }
    { // benchmarkStereoMatchingModels.cpp:176-206: patchMatch on unfolded images, with a NumbersCache and (second call) an initializer callback
        using namespace StereoVision::Correlation;
        Multidim::Array<float, 3> img_left(H, W, 1), img_right(H, W, 1);
        for (int i = 0; i < H; i++)
            for (int j = 0; j < W; j++) {
                img_left.atUnchecked(i, j, 0) = target.valueUnchecked(i, j);
                img_right.atUnchecked(i, j, 0) = source.valueUnchecked(i, j);
            }
        Multidim::Array<float, 3> fVolLeft = unfold<float, float>(2, 2, img_left), fVolRight = unfold<float, float>(2, 2, img_right);
        std::mt19937 re(5);
        StereoVision::Random::NumbersCache<int> cache(1 << 10, [&re]() { return static_cast<int>(re() >> 1); });
        std::optional<StereoVision::Random::NumbersCache<int>> optCache = cache;
        HipBridge::patchMatchSeed() = 77;
        Multidim::Array<disp_t, 3> out1 = patchMatch<matchingFunctions::ZNCC, 1>(fVolRight, fVolLeft, searchOffset<1>(0, 12), 5, 4, std::nullopt, optCache);
        if (out1.shape()[0] != H || out1.shape()[1] != W || out1.shape()[2] != 1) return 13;
        std::optional<std::function<Multidim::Array<disp_t, 3>(Multidim::Array<float, 3> const &, Multidim::Array<float, 3> const &)>> initializer =
            [](Multidim::Array<float, 3> const &fs, Multidim::Array<float, 3> const &) {
                Multidim::Array<disp_t, 3> d(fs.shape()[0], fs.shape()[1], 1);
                for (int i = 0; i < fs.shape()[0]; i++)
                    for (int j = 0; j < fs.shape()[1]; j++) d.atUnchecked(i, j, 0) = (i + 2 * j) % 13;
                return d;
            };
        Multidim::Array<disp_t, 3> out2 = patchMatch<matchingFunctions::SAD, 1>(fVolRight, fVolLeft, searchOffset<1>(0, 12), 4, 3, initializer, std::nullopt);
        if (out2.shape()[0] != H || out2.shape()[1] != W || out2.shape()[2] != 1) return 14;
        dump(out + "_pmf_fvol_left.f32", FE(fVolLeft), fVolLeft.flatLenght());
        dump(out + "_pmf_fvol_right.f32", FE(fVolRight), fVolRight.flatLenght());
        dump(out + "_pmf_zncc.i32", FE(out1), out1.flatLenght());
        dump(out + "_pmf_sad_init.i32", FE(out2), out2.flatLenght());
    }
    { // uint8 images: the byte versions of the pair (values scaled into 0..255), SAD volume, census words, unfold<uint8, uint8>
        Multidim::Array<uint8_t, 2> target8(H, W), source8(H, W);
        for (int i = 0; i < H; i++)
            for (int j = 0; j < W; j++) {
                target8.atUnchecked(i, j) = static_cast<uint8_t>((target.valueUnchecked(i, j) + 1.0f) * 127.5f);
                source8.atUnchecked(i, j) = static_cast<uint8_t>((source.valueUnchecked(i, j) + 1.0f) * 127.5f);
            }
        Multidim::Array<float, 3> sad8 = SC::unfoldBasedCostVolume<SC::matchingFunctions::SAD>(target8, source8, 2, 2, D);
        Multidim::Array<SC::census_data_t, 3> words8 = SC::censusTransform2D(source8, 3, 3);
        Multidim::Array<uint8_t, 3> unfolded8 = SC::unfold<uint8_t, uint8_t>(1, 2, source8);
        if (sad8.empty() || words8.empty() || unfolded8.empty()) return 12;
        dump(out + "_u8_target.u8", FE(target8), target8.flatLenght());
        dump(out + "_u8_source.u8", FE(source8), source8.flatLenght());
        dump(out + "_u8_sad.f32", FE(sad8), sad8.flatLenght());
        dump(out + "_u8_words.u32", FE(words8), words8.flatLenght());
        dump(out + "_u8_unfold.u8", FE(unfolded8), unfolded8.flatLenght());
    }
    // error behaviour: row mismatch -> empty array (cross_correlations.h:751-753)
    Multidim::Array<float, 2> shorter(H - 1, W);
    auto empty = SC::unfoldBasedCostVolume<SC::matchingFunctions::SAD>(target, shorter, h_r, v_r, D);
    if (!empty.empty()) return 3;
    printf("ok\n");
    return 0;
}
