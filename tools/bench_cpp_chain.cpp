// The reference benchmark's chain through the C++ drop-in headers, timed two ways at 1920 x 1080, D = 256 (census 9x9 + SGM-8):
//   host arrays   every function takes and returns Multidim::Array: each volume (2.1 GB) crosses PCIe down and up again
//   device arrays the same names on HipBridge::DeviceArray: images up once, the disparity map down once
//   g++ -std=c++17 -O2 -I libstevi_amd/include tools/bench_cpp_chain.cpp -o tools/bench_cpp_chain -L libstevi_amd -lstevi_hip \
//       -Wl,-rpath,$PWD/libstevi_amd -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lamdhip64 -lpthread
#include <chrono>
#include <cstdio>
#include <random>

#include <correlation/cross_correlations.h>
#include <correlation/sgm.h>
#include <correlation/stevi_hip_bridge.h>
// raw pointer to an array's elements the way the reference's own code gets one: &a.atUnchecked(0, ...) (io/image_io.h:96, :160)
#define FE(x) StereoVision::Correlation::HipBridge::firstElement(x)


namespace SC = StereoVision::Correlation;

int main() {
    const int H = 1080, W = 1920, D = 256;
    Multidim::Array<float, 2> l(H, W), r(H, W);
    std::mt19937 rng(3);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    for (size_t e = 0; e < l.flatLenght(); e++) {
        FE(l)[e] = u(rng);
        FE(r)[e] = u(rng);
    }
    constexpr auto f = SC::matchingFunctions::CENSUS;
    constexpr auto strat = SC::MatchingFunctionTraits<f>::extractionStrategy;
    auto host_chain = [&] {
        auto cv = SC::unfoldBasedCostVolume<f>(l, r, 4, 4, D);
        auto s = SC::sgmCostVolume<8, strat>(cv, 0.001f, 0.01f, StereoVision::Margins(), 100);
        return SC::selectedIndexToDisp<SC::disp_t>(SC::extractSelectedIndex<strat>(s), 0);
    };
    auto device_chain = [&] {
        auto cv = SC::unfoldBasedCostVolumeOnDevice<f>(l, r, 4, 4, D);
        auto s = SC::sgmCostVolume<8, strat>(cv, 0.001f, 0.01f, StereoVision::Margins(), 100);
        return SC::selectedIndexToDisp<SC::disp_t>(SC::extractSelectedIndex<strat>(s), 0).download();
    };
    auto time_ms = [](auto fn, int reps) {
        auto ref = fn(); // warm-up (workspace, first-touch)
        auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < reps; k++) ref = fn();
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
    };
    auto a = host_chain();
    auto b = device_chain();
    long diff = 0;
    for (size_t e = 0; e < a.flatLenght(); e++) diff += FE(a)[e] != FE(b)[e];
    const double th = time_ms(host_chain, 3), td = time_ms(device_chain, 10);
    // the host chain call by call (one more run): where the time of a chain on host arrays goes
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    auto t0 = now();
    auto cv = SC::unfoldBasedCostVolume<f>(l, r, 4, 4, D);
    auto t1 = now();
    auto s = SC::sgmCostVolume<8, strat>(cv, 0.001f, 0.01f, StereoVision::Margins(), 100);
    auto t2 = now();
    auto idx = SC::extractSelectedIndex<strat>(s);
    auto t3 = now();
    auto dsp = SC::selectedIndexToDisp<SC::disp_t>(idx, 0);
    auto t4 = now();
    const double pcie = 4.0 * 4.0 * H * W * D + (8.0 + 12.0) * H * W;
    printf("{\"chain\": \"unfoldBasedCostVolume -> sgmCostVolume<8> -> extractSelectedIndex -> selectedIndexToDisp (C++ drop-in headers), 1920x1080 D=256 census 9x9\", "
           "\"host_arrays_ms\": %.1f, \"device_arrays_ms\": %.2f, \"pixels_differing\": %ld, \"host_pcie_bytes\": %.0f, \"host_pcie_GBps\": %.1f, "
           "\"host_call_ms\": {\"unfoldBasedCostVolume\": %.1f, \"sgmCostVolume\": %.1f, \"extractSelectedIndex\": %.1f, \"selectedIndexToDisp\": %.2f}, "
           "\"results_page_locked\": %d}\n",
           th, td, diff, pcie, pcie / th / 1e6, ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4), svh_host_is_pinned(FE(cv), cv.flatLenght() * sizeof(float)));
    return diff != 0;
}
