"""north_star: "keeping the StereoVision::Correlation template/API surface so it drops into examples/stereo-match unchanged".

The reference's own examples/stereo-match/main.cpp -- READ FROM /root/reference AT TEST TIME, never copied into this repository and never
sent to the GPU box (the tests skip where that tree is absent) -- compiles with `-I libstevi_amd/include` first on the include path and
links against libstevi_hip.so.  The only file the test supplies besides the product headers is tests/cpp/tclap/CmdLine.h, a tests-only
subset of TCLAP (the image has no TCLAP; it is argument parsing, not part of the path).  WITH_GUI is not defined (Qt display code).

What the translation unit takes from the shim tree (main.cpp:21-31, :150-210): io/image_io.h::readImage, MatchingFunctionTraits<ZNCC>,
OnDemandDecoratedFeaturesVolume<ZNFeaturesVolumeDecorator<...>>, searchOffset<2>, cachelessPatchMatch<ZNCC, 2>,
Contiguity::Queen (utils/contiguity.h through cross_correlations.h), InterpolationKernel, CachelessOnDemandImageFlowVolume + its
SearchSpaceType / SearchSpaceBase::{SearchDim, FeatureDim}, refineDisp2dCostInterpolation<Equiangular>, and the includes of
correlation/hierarchical.h and correlation/image_based_refinement.h (the latter a stand-in: nothing of it is called).

Without a GPU the binary runs as far as its first compute call and stops there with the library's "no CPU fallback" error; the same
chain written with the reference's names runs on the GPU and is compared with the oracle in tests/test_cpp_dropin.py /
tests/test_gpu_patchmatch.py."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE_MAIN = "/root/reference/examples/stereo-match/main.cpp"

pytestmark = pytest.mark.skipif(not os.path.exists(REFERENCE_MAIN), reason="the reference tree is not on this machine")


@pytest.fixture(scope="module")
def stereo_match(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("stereo_match") / "stereo_match")
    # the shim directory FIRST, then the tests-only TCLAP subset; nothing from /root/reference is on the include path
    cmd = ["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "libstevi_amd", "include"), "-I", os.path.join(ROOT, "tests", "cpp"),
           REFERENCE_MAIN, "-o", exe, "-L", os.path.join(ROOT, "libstevi_amd"), "-lstevi_hip", "-Wl,-rpath," + os.path.join(ROOT, "libstevi_amd"),
           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-lpthread"]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-4000:]
    return exe


def test_reference_stereo_match_main_compiles_and_links_unchanged(stereo_match):
    assert os.path.exists(stereo_match)
    # the translation unit is the reference's, byte for byte: nothing was preprocessed in or patched (the compile read it in place)
    deps = subprocess.run(["g++", "-std=c++17", "-MM", "-I", os.path.join(ROOT, "libstevi_amd", "include"), "-I", os.path.join(ROOT, "tests", "cpp"),
                           REFERENCE_MAIN], capture_output=True, text=True)
    assert deps.returncode == 0
    headers = [os.path.normpath(t) for t in deps.stdout.replace("\\\n", " ").split() if t.endswith(".h")]
    assert headers and not [h for h in headers if h.startswith("/root/reference")], "a header came from the reference tree"
    for needed in ("correlation/patchmatch.h", "correlation/hierarchical.h", "correlation/image_based_refinement.h", "correlation/on_demand_features_volume.h",
                   "correlation/on_demand_cost_volume.h", "correlation/cost_based_refinement.h", "utils/contiguity.h", "io/image_io.h"):
        assert any(h.endswith("libstevi_amd/include/" + needed) for h in headers), needed


def test_reference_stereo_match_argument_handling_and_image_loading(stereo_match, tmp_path):
    # no arguments: its own catch block reports the TCLAP error, then the image check stops it (main.cpp:134-142)
    out = subprocess.run([stereo_match], capture_output=True, text=True)
    assert out.returncode == 1 and "Argument error" in out.stderr and "Could not load input images" in out.stderr
    out = subprocess.run([stereo_match, "nowhere_a.stevimg", "nowhere_b.stevimg"], capture_output=True, text=True)
    assert out.returncode == 1 and "Could not load input images" in out.stderr
    # channel mismatch is caught before any compute (main.cpp:147-150): a 1-channel and a 3-channel image
    import numpy as np

    from libstevi_amd import write_stevimg
    write_stevimg(str(tmp_path / "a.stevimg"), np.zeros((6, 8, 1), np.float32))
    write_stevimg(str(tmp_path / "b.stevimg"), np.zeros((6, 8, 3), np.float32))
    out = subprocess.run([stereo_match, str(tmp_path / "a.stevimg"), str(tmp_path / "b.stevimg")], capture_output=True, text=True)
    assert out.returncode == 1 and "Source image: size 6x8x1" in out.stdout and "Inconsistent number of channels" in out.stderr


def test_reference_stereo_match_has_no_cpu_fallback(stereo_match):
    """On a machine without a HIP device the example loads its images through the shim's readImage and then fails loudly at the first
    compute call; with a device (the GPU box never has the reference tree, so this only runs on a developer's GPU machine) it finishes."""
    import libstevi_amd._capi as capi
    pair = os.path.join(ROOT, "tests", "golden", "stereo_pair")
    out = subprocess.run([stereo_match, os.path.join(pair, "img_r.stevimg"), os.path.join(pair, "img_l.stevimg"), "--right-search-delta=8", "--n-iter=3", "--refine"],
                         capture_output=True, text=True)
    assert "Source image: size 24x40x1" in out.stdout
    if capi.load().svh_device_available():
        assert out.returncode == 0 and "Disparity computed" in out.stdout
    else:
        assert out.returncode != 0 and "no CPU fallback" in out.stderr and "Disparity computed" not in out.stdout
