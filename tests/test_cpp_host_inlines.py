"""The inline HOST functions of the drop-in headers (hammingScalar / hammingDistance, dotProduct, SumSquareDiff, SumAbsDiff,
MatchingFunctionTraits<f>::featureComparison, refineCostTriplet, condImgRef, MatchingFuncComputeTypeInfos, Contiguity) against the oracle:
tests/cpp/host_inlines.cpp builds whole cost volumes out of featureComparison calls on 1-D views, the way the reference's aggregateCost
does (cross_correlations.h:230-241), with no GPU call; the oracle's volumes for the same features must come out bit for bit.  CPU only."""
import os
import subprocess

import numpy as np
import pytest

import oracle as so

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("host_inlines") / "host_inlines")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "libstevi_amd", "include"),
                           os.path.join(ROOT, "tests", "cpp", "host_inlines.cpp"), "-o", out,
                           "-L", os.path.join(ROOT, "libstevi_amd"), "-lstevi_hip", "-Wl,-rpath," + os.path.join(ROOT, "libstevi_amd"),
                           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-lpthread"])
    return out


def run(exe, d, fl, fr, wl, wr, D, triplets):
    H, W, F = fl.shape
    fl.astype(np.float32).tofile(d / "fl.f32")
    fr.astype(np.float32).tofile(d / "fr.f32")
    wl.astype(np.uint32).tofile(d / "wl.u32")
    wr.astype(np.uint32).tofile(d / "wr.u32")
    triplets.astype(np.float32).tofile(d / "triplets.f32")
    out = subprocess.run([exe, str(H), str(W), str(F), str(wl.shape[2]), str(D), str(d)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    return {n: np.fromfile(d / f"cv_{n}.f32", np.float32).reshape(H, W, D) for n in ("CC", "NCC", "SSD", "SAD", "ZSSD", "ZSAD", "HAMMING", "CENSUS")}


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_host_inline_functions_reproduce_the_oracle_volumes(exe, tmp_path):
    rng = np.random.default_rng(11)
    H, W, D, r = 9, 23, 7, 2
    img_l = rng.uniform(-1, 1, (H, W)).astype(np.float32)
    img_r = rng.uniform(-1, 1, (H, W)).astype(np.float32)
    raw_l, raw_r = so.unfold(img_l, r, r), so.unfold(img_r, r, r)
    words_l, words_r = so.census_transform(img_l, r, r), so.census_transform(img_r, r, r)
    # rule E2: the TARGET (left) words take the reference's trip through float before they are compared (cross_correlations.h:235-236)
    words_l_rt = np.array([so.round_word_through_float(int(w)) for w in words_l.ravel()], np.uint32).reshape(words_l.shape)
    trip = np.concatenate([rng.uniform(0.1, 5.0, (200, 3)), np.array([[3, 1, 2], [2, 1, 2], [1, 1, 1], [5, 2, 9], [1e-3, 1e-4, 2e-3]])]).astype(np.float32)

    got = run(exe, tmp_path, raw_l, raw_r, words_l_rt, words_r, D, trip)
    for name, func in (("CC", so.CC), ("SSD", so.SSD), ("SAD", so.SAD)):
        assert np.array_equal(bits(got[name]), bits(so.feature_cost_volume(func, raw_l, raw_r, D))), name
    hamming = so.unfold_cost_volume(so.CENSUS, img_l, img_r, r, r, D)
    assert np.array_equal(got["CENSUS"], hamming) and np.array_equal(got["HAMMING"], hamming)

    # normalised features: NCC's comparison is the dot product of what getFeatureVolumeForMatchFunc hands it
    ncc_l, ncc_r = so.feature_volume_for_match_func(so.NCC, raw_l), so.feature_volume_for_match_func(so.NCC, raw_r)
    got = run(exe, tmp_path, ncc_l, ncc_r, words_l_rt, words_r, D, trip)
    assert np.array_equal(bits(got["NCC"]), bits(so.feature_cost_volume(so.NCC, raw_l, raw_r, D)))
    assert np.array_equal(bits(got["NCC"]), bits(got["CC"]))
    # zero-mean features
    z_l, z_r = so.feature_volume_for_match_func(so.ZSSD, raw_l), so.feature_volume_for_match_func(so.ZSSD, raw_r)
    got = run(exe, tmp_path, z_l, z_r, words_l_rt, words_r, D, trip)
    assert np.array_equal(bits(got["ZSSD"]), bits(so.feature_cost_volume(so.ZSSD, raw_l, raw_r, D)))
    assert np.array_equal(bits(got["ZSAD"]), bits(so.feature_cost_volume(so.ZSAD, raw_l, raw_r, D)))

    # refineCostTriplet against the oracle's (cost_based_refinement.h:43-69), NaN for NaN
    for k in (so.EQUIANGULAR, so.PARABOLA, so.GAUSSIAN):
        mine = np.fromfile(tmp_path / f"ref_{k}.f32", np.float32)
        exp = np.array([so.refine_triplet(k, *t) for t in trip], np.float32)
        assert np.array_equal(np.isnan(mine), np.isnan(exp)), k
        ok = ~np.isnan(exp)
        assert np.array_equal(bits(mine[ok]), bits(exp[ok])), k
    assert np.isnan(np.fromfile(tmp_path / "ref_1.f32", np.float32)[202])  # flat triplet: 0/0 as in the reference
