"""tests/golden/: hand_computed.json (known answers recorded from the reference's own headers during the survey, or worked out
by hand from the cited lines) and oracle_regression.npz (the oracle's outputs frozen on seeded inputs, see make_golden.py).
CPU: the oracle against both.  GPU (-m gpu): the HIP path against the same known answers and against the frozen vectors."""
import json
import os

import numpy as np
import pytest

import oracle as so

HERE_TESTS = os.path.dirname(os.path.abspath(__file__))
HERE = os.path.join(HERE_TESTS, "golden")
KNOWN = json.load(open(os.path.join(HERE, "hand_computed.json")))


def fval(x):
    return {"nan": np.nan, "inf": np.inf, "-inf": -np.inf}.get(x, x) if isinstance(x, str) else x


def sgm_constant_expected(case):
    H, W, n = case["H"], case["W"], case["n_dir"]
    if case["expected"] != "formula":
        return np.array(case["expected"], np.float32).reshape(H, W, 1)
    i, j = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    si, sj = (-1.0) ** i, (-1.0) ** j
    S = 1 + si + sj
    if n == 8:
        S = S + (i >= j) * sj + (j >= i) * si + (i + j < W) * si + (i + j < H) * sj
    return S.astype(np.float32).reshape(H, W, 1)


def ramp(kind, H=12, W=14):
    base = np.arange(H * W, dtype=np.float32).reshape(H, W)
    return {"constant": np.full((H, W), 0.5, np.float32), "increasing_ramp": base, "decreasing_ramp": -base}[kind]


def tcv_nan_mask(c):
    H, W, D, h_r, v_r, r = c["H"], c["W"], c["D"], c["h_r"], c["v_r"], c["radius"]
    rng = np.random.default_rng(3)
    idx = rng.integers(0, D, (H, W)).astype(np.int32)
    mask = np.zeros((H, W, 2 * r + 1), bool)
    for i in range(H):
        for j in range(W):
            for k in range(2 * r + 1):
                p = idx[i, j] + k - r
                mask[i, j, k] = p < 0 or p >= D or j < h_r or j + p + h_r >= W or i < v_r or i + v_r >= H
    return idx, mask


class OracleBackend:
    name = "oracle"

    def extract_index(self, cv, strategy): return so.extract_index(cv, strategy)
    def sgm(self, cv, n_dir): return so.sgm(cv, n_dir, so.COST, 0.001, 0.01, (0, 0, 0, 0), 100.0)
    def sgm_literal(self, cv, n_dir): return so.sgm(cv, n_dir, so.COST, 0.001, 0.01, (0, 0, 0, 0), 100.0, variant=0)
    def census(self, img, h_r, v_r): return so.census_transform(img, h_r, v_r)
    def refine(self, tcv, raw, kernel): return so.refine_disp(tcv, raw, kernel)
    def truncated(self, cv, idx, h_r, v_r, r): return so.truncated_cost_volume(cv, idx, h_r, v_r, r)
    def hamming_volume(self, feat_l, feat_r, D): return so.feature_cost_volume(so.CENSUS, feat_l, feat_r, D)


def check_known_answers(be):
    for c in KNOWN["extract_index"]:
        cv = np.array([fval(v) for v in c["values"]], np.float32).reshape(1, 1, -1)
        assert int(be.extract_index(cv, so.COST if c["strategy"] == "cost" else so.SCORE)[0, 0]) == c["index"], c
    for c in KNOWN["sgm_constant_cost"]["cases"]:
        cv = np.ones((c["H"], c["W"], 1), np.float32)
        assert np.array_equal(be.sgm(cv, c["n_dir"]), sgm_constant_expected(c)), c
    for c in KNOWN["census_words"]:
        w = be.census(ramp(c["image"]), c["h_r"], c["v_r"])
        exp = [int(x, 16) for x in c["interior_words"]]
        inner = w[c["v_r"]:-c["v_r"], c["h_r"]:-c["h_r"]]
        assert inner.shape[2] == len(exp) and all(np.all(inner[:, :, k] == e) for k, e in enumerate(exp)), c
    kern = {"parabola": so.PARABOLA, "equiangular": so.EQUIANGULAR, "gaussian": so.GAUSSIAN}
    for c in KNOWN["refine_triplet"]:
        got = be.refine(np.array(c["c"], np.float32).reshape(1, 1, 3), np.zeros((1, 1), np.int32), kern[c["kernel"]])[0, 0]
        exp = fval(c["delta"])
        assert (np.isnan(got) if isinstance(exp, float) and np.isnan(exp) else abs(got - exp) < 1e-6), c
    t = KNOWN["truncated_cost_volume_nan_rule"]
    idx, mask = tcv_nan_mask(t)
    cv = np.random.default_rng(4).uniform(0, 1, (t["H"], t["W"], t["D"])).astype(np.float32)
    assert np.array_equal(np.isnan(be.truncated(cv, idx, t["h_r"], t["v_r"], t["radius"])), mask)
    # finding F7: the Hamming distance of a census feature vector to itself is not always 0.  33 channels whose comparisons spell
    # the word 0x01000001 (bits 0 and 24 set): channel 0 is the reference value, channel c sets bit c - 1 when ref > value
    h = KNOWN["hamming_self_distance"]
    word = int(h["feature_word"], 16)
    feat = np.ones((1, 1, 33), np.float32)
    for b in range(32):
        feat[0, 0, 1 + b] = 0.0 if (word >> b) & 1 else 2.0
    assert int(be.hamming_volume(feat, feat, 1)[0, 0, 0]) == h["distance_to_itself"]


def test_oracle_matches_known_answers():
    for c in KNOWN["round_word_through_float"]:
        assert so.round_word_through_float(int(c["word"], 16)) == int(c["rounded"], 16), c
    check_known_answers(OracleBackend())
    # the literal O(D^2) loops give the same constant-cost pattern
    for c in KNOWN["sgm_constant_cost"]["cases"]:
        assert np.array_equal(OracleBackend().sgm_literal(np.ones((c["H"], c["W"], 1), np.float32), c["n_dir"]), sgm_constant_expected(c))


def test_oracle_regression_vectors_are_unchanged():
    """oracle_regression.npz was generated by the oracle itself: this guards against accidental changes of the restatement"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    frozen = np.load(os.path.join(HERE, "oracle_regression.npz"))
    now = mod.build()
    assert set(frozen.files) == set(now)
    for k in frozen.files:
        a, b = frozen[k], now[k]
        assert a.shape == b.shape and a.dtype == b.dtype, k
        assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b), k


@pytest.mark.gpu
def test_hip_path_matches_known_answers_and_frozen_vectors():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import libstevi_amd as sv
    MF = sv.matchingFunctions
    dev = torch.device("cuda:0")
    d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
    h = lambda x: x.cpu().numpy()  # noqa: E731

    class Hip:
        name = "hip"
        def extract_index(self, cv, strategy): return h(sv.extractSelectedIndex(strategy, d(cv)))
        def sgm(self, cv, n_dir): return h(sv.sgmCostVolume(n_dir, so.COST, d(cv), 0.001, 0.01, sv.Margins(), 100.0))
        def census(self, img, h_r, v_r): return h(sv.censusTransform2D(d(img), h_r, v_r))
        def refine(self, tcv, raw, kernel): return h(sv.refineDispCostInterpolation(kernel, d(tcv), d(raw)))
        def truncated(self, cv, idx, h_r, v_r, r): return h(sv.truncatedCostVolume(d(cv), d(idx), h_r, v_r, r))
        def hamming_volume(self, fl, fr, D): return h(sv.featureVolume2CostVolume(MF.CENSUS, d(fl), d(fr), D))

    check_known_answers(Hip())
    g = np.load(os.path.join(HERE, "oracle_regression.npz"))
    left, right = g["in_left"], g["in_right"]
    bits = lambda a: np.ascontiguousarray(a).view(np.uint32)  # noqa: E731
    assert np.array_equal(h(sv.censusTransform2D(d(left), 4, 4)), g["census_4_4"])
    assert np.array_equal(h(sv.unfoldBasedCostVolume(MF.CENSUS, d(left), d(right), 4, 4, 9)), g["cv_census_r2l"])
    for name, func in (("sad", MF.SAD), ("ssd", MF.SSD), ("ncc", MF.NCC), ("zncc", MF.ZNCC), ("zsad", MF.ZSAD)):
        got = h(sv.unfoldBasedCostVolume(func, d(left), d(right), 2, 2, 9))
        assert np.max(np.abs(got - g[f"cv_{name}_r2l"])) <= 1e-4
        got = h(sv.unfoldBasedCostVolume(func, d(left), d(right), 2, 2, sv.searchOffset1(-2, 4), sv.dispDirection.LeftToRight))
        assert np.max(np.abs(got - g[f"cv_{name}_l2r_off"])) <= 1e-4
    assert np.array_equal(bits(h(sv.sgmCostVolume(8, so.COST, d(g["cv_census_r2l"]), 0.001, 0.01, sv.Margins(), 100.0))), bits(g["sgm8_cost_census"]))
    assert np.array_equal(bits(h(sv.sgmCostVolume(4, so.COST, d(g["cv_sad_r2l"]), 0.5, 2.0, sv.Margins(1, 2, 0, 1), 3.5))), bits(g["sgm4_cost_sad_margins"]))
    assert np.array_equal(bits(h(sv.sgmCostVolume(8, so.SCORE, d(g["cv_zncc_r2l"]), 0.001, 0.01, sv.Margins(), 100.0))), bits(g["sgm8_score_zncc"]))
    assert np.array_equal(bits(h(sv.sgmCostVolume(8, so.SCORE, d(g["cv_zncc_r2l"]), 0.3, 0.1, sv.Margins(1, 0, 2, 0), 1.0))), bits(g["sgm8_score_zncc_p1_gt_p2"]))
    assert np.array_equal(h(sv.extractSelectedIndex(so.COST, d(g["sgm8_cost_census"]))), g["idx_census"])
    for name, mode in (("same", sv.truncatedCostVolumeDirection.Same), ("reversed", sv.truncatedCostVolumeDirection.Reversed), ("both", sv.truncatedCostVolumeDirection.Both)):
        got = h(sv.truncatedCostVolume(d(g["sgm8_score_zncc"]), d(g["idx_zncc"]), 2, 2, 1, sdir=mode))
        assert np.array_equal(np.isnan(got), np.isnan(g[f"tcv_{name}"])) and np.array_equal(got[~np.isnan(got)], g[f"tcv_{name}"][~np.isnan(got)])
    assert np.array_equal(bits(h(sv.averagePoolingDownsample(d(left), 2))), bits(g["downsample"]))
    assert np.array_equal(bits(h(sv.unfoldCompressed(sv.UnFoldCompressor(g["compressor_mask"]), d(left)))), bits(g["unfold_compressed"]))
    tb = h(sv.sgmCostVolume(8, so.COST, d(g["cv_sad_r2l"]), 0.5, 2.0, sv.Margins(), 3.0, semantics="textbook"))
    assert np.array_equal(tb, g["sgm8_textbook_cost"])
    assert np.array_equal(bits(h(sv.onDemandFeatures(MF.ZNCC, d(g["in_rgb"]), 1))), bits(g["on_demand_features_zncc"]))
    pm = sv.cachelessPatchMatch(MF.ZNCC, d(g["in_rgb"]), d(np.roll(g["in_rgb"], (1, -1), (0, 1)).copy()), 1, sv.searchOffset2(-2, 2, -2, 2), 4, 3, seed=7)
    assert np.array_equal(h(pm), g["patch_match"])


# ---- the .stevimg fixture pair (tests/golden/stereo_pair, written by make_golden.py in the reference's array file format) ----
PAIR = os.path.join(HERE, "stereo_pair")


def load_pair():
    from libstevi_amd.stevimg import read_stevimg
    return {os.path.splitext(f)[0]: read_stevimg(os.path.join(PAIR, f)) for f in sorted(os.listdir(PAIR))}


def check_pair_outputs(fx, cv, sg, idx, refined):
    assert np.array_equal(cv, fx["cost_volume_census"])
    assert np.array_equal(sg.view(np.uint32), fx["sgm8_cost_volume"].view(np.uint32))
    assert np.array_equal(idx, fx["disp_index"])
    exp = fx["refined_parabola"]
    assert np.array_equal(np.isnan(refined), np.isnan(exp))
    ok = ~np.isnan(exp)
    assert np.max(np.abs(refined[ok] - exp[ok])) <= 1e-4  # float refinement: the north star's tolerance


def test_stevimg_pair_oracle():
    fx = load_pair()
    assert fx["img_l"].dtype == np.float32 and fx["disp_index"].dtype == np.int32
    H, W, D = fx["cost_volume_census"].shape
    assert fx["cost_volume_census"].strides == (4 * W * D, 4, 4 * W)  # the reference's cost-volume layout survives the file
    cv = so.unfold_cost_volume(so.CENSUS, fx["img_l"], fx["img_r"], 4, 4, D)
    sg = so.sgm(cv, 8, so.COST, 0.001, 0.01, (0, 0, 0, 0), 100.0)
    idx = so.extract_index(sg, so.COST)
    check_pair_outputs(fx, cv, sg, idx, so.refine_disp(so.truncated_cost_volume(sg, idx, 4, 4, 1, so.TCV_SAME), idx, so.PARABOLA))


@pytest.mark.gpu
def test_stevimg_pair_hip_python_and_cpp(tmp_path):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import subprocess
    import libstevi_amd as sv
    from libstevi_amd.stevimg import read_stevimg
    fx = load_pair()
    D = fx["cost_volume_census"].shape[2]
    dev = torch.device("cuda:0")
    res = sv.stereoMatch(sv.matchingFunctions.CENSUS, torch.from_numpy(fx["img_l"].copy()).to(dev), torch.from_numpy(fx["img_r"].copy()).to(dev), 4, 4, D,
                         sgmDirections=8, P1=0.001, P2=0.01, Pout=100.0, refineKernel=sv.InterpolationKernel.Parabola, refine_h_radius=4,
                         refine_v_radius=4, want_cv=True, want_sgm_cv=True)
    check_pair_outputs(fx, res["cv"].cpu().numpy(), res["sgm_cv"].cpu().numpy(), res["disp"].cpu().numpy(), res["refined"].cpu().numpy())
    # the same through the C++ drop-in headers, files in and files out
    root = os.path.dirname(HERE_TESTS)
    exe = str(tmp_path / "stevimg_stereo")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(root, "libstevi_amd", "include"),
                           os.path.join(HERE_TESTS, "cpp", "stevimg_stereo.cpp"), "-o", exe, "-L", os.path.join(root, "libstevi_amd"), "-lstevi_hip",
                           "-Wl,-rpath," + os.path.join(root, "libstevi_amd"), "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-lpthread"])
    run = subprocess.run([exe, PAIR, str(tmp_path), str(D)], capture_output=True, text=True)
    assert run.returncode == 0, run.stderr
    got = {k: read_stevimg(tmp_path / f"{k}.stevimg") for k in ("cost_volume_census", "sgm8_cost_volume", "disp_index", "refined_parabola")}
    check_pair_outputs(fx, got["cost_volume_census"], got["sgm8_cost_volume"], got["disp_index"], got["refined_parabola"])
