"""Hand-computed / first-principles checks for the oracle rows the reference never tests
(census, Hamming, float round trip, SGM, argmin tie rule, truncated volume, refinement kernels).
Pure-Python loops follow the cited reference lines literally on tiny inputs."""
import math

import numpy as np
import pytest

import oracle as so


def py_census(img, h_r, v_r):
    """census.h:69-115 on top of unfold.h:247-291, literal loops."""
    H, W = img.shape
    h, v = 2 * h_r + 1, 2 * v_r + 1
    F = h * v
    nW = (F - 1) // 32 + 1
    out = np.zeros((H, W, nW), np.uint32)
    for i in range(H):
        for j in range(W):
            feat = []
            for k in range(v):
                for l in range(h):
                    ii, jj = i - v_r + k, j - h_r + l
                    feat.append(img[ii, jj] if 0 <= ii < H and 0 <= jj < W else np.float32(0))
            ref, d, b, ch = feat[0], 0, 0, 0
            for c in range(1, F):
                d |= (1 if ref > feat[c] else 0) << b
                b += 1
                if b >= 32:
                    out[i, j, ch] = d
                    ch, d, b = ch + 1, 0, 0
    return out


@pytest.mark.parametrize("h_r,v_r", [(1, 1), (2, 2), (3, 3), (4, 4), (5, 2), (4, 3)])
def test_census_bits(rng, h_r, v_r):
    img = rng.uniform(-1, 1, (9, 11)).astype(np.float32)
    img[2, 3] = np.nan  # strict '>' makes every comparison with NaN false
    got = so.census_transform(img, h_r, v_r)
    assert np.array_equal(got, py_census(img, h_r, v_r))
    F = (2 * h_r + 1) * (2 * v_r + 1)
    # F6 / E1: words beyond floor((F-1)/32) are never written -> 0
    assert np.all(got[:, :, (F - 1) // 32:] == 0)


def test_census_reference_pixel_is_top_left():
    """census.h:89 with unfold.h:180: the comparisons are against window element (k=0,l=0)."""
    img = np.zeros((9, 9), np.float32)
    img[0, 0] = 5.0  # top-left of the 9x9 window centred at (4,4)
    w = so.census_transform(img, 4, 4)
    assert w[4, 4, 0] == 0xFFFFFFFF and w[4, 4, 1] == 0xFFFFFFFF and w[4, 4, 2] == 0
    assert w[3, 3, 0] == 0  # window top-left is outside the image -> ref = 0, nothing is smaller


def test_census_needs_two_channels():
    assert so.census_features(np.zeros((2, 2, 1), np.float32)).size == 0  # census.h:76-78


@pytest.mark.parametrize("w,expected", [
    (0, 0), (1, 1), (0x00FFFFFF, 0x00FFFFFF), (0x01000001, 0x01000000), (0x01000003, 0x01000004),
    (0x7FFFFFFF, 0x80000000), (0xFFFFFF7F, 0xFFFFFF00), (0xFFFFFF80, 0xFFFFFFFF), (0xFFFFFFFF, 0xFFFFFFFF),
    (0x80000080, 0x80000000), (0x80000180, 0x80000200),
])
def test_round_word_through_float(w, expected):
    """Rule E2: round-to-nearest-even to 24 significant bits, 2^32 saturates."""
    assert so.round_word_through_float(w) == expected


E2_WORDS = [0xFFFFFFFF, 0xFFFFFF80, 0xFFFFFF7F, 0xFFFFFF00, 0xFFFFFFC1, 0x80000000, 0x01000001, 0x7FFFFFFF, 5]


def test_round_word_through_float_zero_mode():
    """Rule E2, second behaviour: a word that rounds to 2^32 becomes 0; every other word is unchanged by the switch."""
    try:
        so.set_float_overflow(True)
        assert so.get_float_overflow() == 1
        assert [so.round_word_through_float(w) for w in (0xFFFFFF80, 0xFFFFFFC1, 0xFFFFFFFF)] == [0, 0, 0]
        assert so.round_word_through_float(0xFFFFFF7F) == 0xFFFFFF00 and so.round_word_through_float(0x01000003) == 0x01000004
    finally:
        so.set_float_overflow(False)
    assert so.round_word_through_float(0xFFFFFFFF) == 0xFFFFFFFF


def _host_round_trip(flags, tmp_path):
    """`float t = word; uint32_t back = t;` (cross_correlations.h:235-236) compiled by this host's gcc with `flags` and executed."""
    import subprocess
    src = tmp_path / "e2.c"
    src.write_text("#include <stdio.h>\n#include <stdint.h>\n#include <stdlib.h>\nint main(int n, char **a) { for (int i = 1; i < n; i++) {"
                   " volatile uint32_t w = (uint32_t)strtoul(a[i], 0, 16); volatile float t = (float)w; volatile uint32_t b = (uint32_t)t;"
                   " printf(\"%08x\\n\", b); } return 0; }\n")
    exe = tmp_path / ("e2_" + str(abs(hash(flags))))
    subprocess.check_call(["gcc", "-O3"] + flags.split() + [str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)] + [f"{w:08x}" for w in E2_WORDS], text=True)
    return [int(x, 16) for x in out.split()]


def test_e2_overflow_matches_this_hosts_conversions(tmp_path):
    """Pins both modes of rule E2 to what x86-64 code generation does with the reference's statement, by running it here: with the
    flags the reference adds on x86 (-mavx -mavx2 -mfma, CMakeLists.txt:44-58) the conversion is vcvttss2si r64 + truncation, 2^32 -> 0
    (mode 1); with AVX-512 code generation (what -march=native, CMakeLists.txt:41, gives on such a host) it is vcvttss2usi, 2^32 ->
    0xFFFFFFFF (mode 0, the default).  Every other word agrees in both."""
    import platform
    import shutil
    if platform.machine() != "x86_64" or shutil.which("gcc") is None:
        pytest.skip("needs gcc on x86-64")
    got = _host_round_trip("-mavx -mavx2 -mfma", tmp_path)
    try:
        so.set_float_overflow(True)
        assert got == [so.round_word_through_float(w) for w in E2_WORDS]
    finally:
        so.set_float_overflow(False)
    with open("/proc/cpuinfo") as f:
        if "avx512f" not in f.read():
            return
    got = _host_round_trip("-mavx512f", tmp_path)
    assert got == [so.round_word_through_float(w) for w in E2_WORDS]


def test_hamming_volume_by_hand(rng):
    """aggregateCost<CENSUS> (cross_correlations.h:194-249) with hammingDistance (matching_costs.h:236-263)."""
    H, W, D = 5, 13, 6
    left = rng.uniform(-1, 1, (H, W)).astype(np.float32)
    right = rng.uniform(-1, 1, (H, W)).astype(np.float32)
    cv = so.unfold_cost_volume(so.CENSUS, left, right, 4, 4, D)
    wl, wr = py_census(left, 4, 4), py_census(right, 4, 4)
    for i in range(H):
        for j in range(W):
            for d in range(D):
                exp = 0
                for k in range(3):
                    t = so.round_word_through_float(int(wl[i, j + d, k])) if j + d < W else 0
                    exp += bin(int(wr[i, j, k]) ^ t).count("1")
                assert cv[i, j, d] == exp
    assert np.array_equal(cv, so.unfold_cost_volume(so.HAMMING, left, right, 4, 4, D))


def test_hamming_self_distance_not_zero_for_big_words(rng):
    """F7: target words lose their low bits, so Hamming(x, x) > 0 when a word needs > 24 bits;
    an all-ones word rounds to 2^32 and saturates back to itself (rule E2)."""
    img = rng.uniform(-1, 1, (12, 12)).astype(np.float32)
    cv = so.unfold_cost_volume(so.CENSUS, img, img, 4, 4, 1)
    assert (cv[4:8, 4:8, 0] > 0).any()
    ramp = np.tile(np.arange(12, 0, -1, dtype=np.float32), (12, 1)) + np.arange(12, 0, -1, dtype=np.float32)[:, None]
    assert np.all(so.census_transform(ramp, 4, 4)[6, 6, :2] == 0xFFFFFFFF)
    assert so.unfold_cost_volume(so.CENSUS, ramp, ramp, 4, 4, 1)[6, 6, 0] == 0


def test_cost_volume_out_of_range_target_is_zero_vector(rng):
    """cross_correlations.h:235: columns outside the target image contribute an all-zero vector."""
    left = rng.uniform(-1, 1, (6, 8)).astype(np.float32)
    right = rng.uniform(-1, 1, (6, 8)).astype(np.float32)
    D = 12  # larger than the width
    fr = so.unfold(right, 1, 1)
    sad = so.unfold_cost_volume(so.SAD, left, right, 1, 1, D)
    ssd = so.unfold_cost_volume(so.SSD, left, right, 1, 1, D)
    ncc = so.unfold_cost_volume(so.NCC, left, right, 1, 1, D)
    for j in range(8):
        for d in range(D):
            if j + d >= 8:
                assert np.allclose(sad[:, j, d], np.abs(fr[:, j]).sum(-1), rtol=1e-6)
                assert np.allclose(ssd[:, j, d], (fr[:, j] ** 2).sum(-1), rtol=1e-6)
                assert np.all(ncc[:, j, d] == 0)
    # LeftToRight looks at j - d
    sad_l2r = so.unfold_cost_volume(so.SAD, left, right, 1, 1, D, ddir=so.LEFT_TO_RIGHT)
    fl = so.unfold(left, 1, 1)
    assert np.allclose(sad_l2r[:, 2, 5], np.abs(fl[:, 2]).sum(-1), rtol=1e-6)
    assert np.allclose(sad_l2r[:, 5, 2], np.abs(fl[:, 5] - fr[:, 3]).sum(-1), rtol=1e-5)


def test_cost_volume_shape_errors():
    a = np.zeros((4, 6), np.float32)
    b = np.zeros((5, 6), np.float32)
    assert so.unfold_cost_volume(so.SAD, a, b, 1, 1, 3).size == 0  # cross_correlations.h:751-753
    c = np.zeros((4, 9), np.float32)
    assert so.unfold_cost_volume(so.SAD, a, c, 1, 1, 3).shape == (4, 9, 3)  # widths may differ; source = right


def test_search_offset_range(rng):
    """aggregateCost(searchOffset<1>) (cross_correlations.h:251-308): disp = lower + idx."""
    left = rng.uniform(-1, 1, (5, 12)).astype(np.float32)
    right = rng.uniform(-1, 1, (5, 12)).astype(np.float32)
    full = so.unfold_cost_volume(so.SSD, left, right, 1, 1, 6)
    off = so.unfold_cost_volume(so.SSD, left, right, 1, 1, 5, disp_lower=-2)
    assert np.array_equal(off[:, :, 2:], full[:, :, :3])
    fl, fr = so.unfold(left, 1, 1), so.unfold(right, 1, 1)
    assert np.allclose(off[:, 2:, 0], ((fr[:, 2:] - fl[:, :-2]) ** 2).sum(-1), rtol=1e-5)
    assert np.allclose(off[:, :2, 0], (fr[:, :2] ** 2).sum(-1), rtol=1e-5)  # j - 2 < 0 -> zero vector


# ---------------------------------------------------------------- SGM ---------------------------------------
def visits_8(H, W, margins):
    """n(p) for the effective 8-direction set (F5)."""
    l, t, r, b = margins
    n = np.zeros((H, W), np.int32)
    Hp, Wp = H - t - b, W - l - r
    for i in range(t, H - b):
        for j in range(l, W - r):
            ip, jp = i - t, j - l
            n[i, j] = 2 + 1 + (ip == jp) + (ip + jp < Wp) + (ip + jp < Hp)
    return n


@pytest.mark.parametrize("strategy", [so.COST, so.SCORE])
@pytest.mark.parametrize("margins", [(0, 0, 0, 0), (1, 2, 3, 1)])
@pytest.mark.parametrize("integer", [True, False])
@pytest.mark.parametrize("n_dir", [4, 8])
def test_sgm_linear_equals_literal(rng, strategy, margins, integer, n_dir):
    H, W, D = 11, 14, 9
    cv = rng.integers(0, 64, (H, W, D)).astype(np.float32) if integer else rng.uniform(-1, 1, (H, W, D)).astype(np.float32)
    for P1, P2, Pout in [(0.001, 0.01, 100.0), (2.0, 7.0, 3.0), (5.0, 1.0, 0.5)]:
        a = so.sgm(cv, n_dir, strategy, P1, P2, margins, Pout, variant=0)
        b = so.sgm(cv, n_dir, strategy, P1, P2, margins, Pout, variant=1)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_sgm_nonfinite_costs(rng):
    cv = rng.uniform(-1, 1, (6, 7, 5)).astype(np.float32)
    cv[2, 3, 1] = np.inf
    cv[4, 1, :] = np.nan
    cv[1, 5, 2] = -np.inf
    for strategy in (so.COST, so.SCORE):
        a = so.sgm(cv, 8, strategy, 0.5, 1.5, variant=0)
        b = so.sgm(cv, 8, strategy, 0.5, 1.5, variant=1)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("margins", [(0, 0, 0, 0), (2, 1, 1, 3)])
@pytest.mark.parametrize("shape", [(9, 13), (13, 9), (10, 10)])
def test_sgm_coverage_and_cost_identity(rng, margins, shape):
    """F4/F5: in the Cost branch with integer costs,
    S = (1+n) C + n Pout [j+d>=W] - sum of per-pass minima, so argmin S = argmin of the first two terms."""
    H, W = shape
    D = 6
    cv = rng.integers(0, 40, (H, W, D)).astype(np.float32)
    Pout = 100.0
    S = so.sgm(cv, 8, so.COST, 0.001, 0.01, margins, Pout)
    n = visits_8(H, W, margins)
    oob = (np.arange(W)[:, None] + np.arange(D)[None, :] >= W).astype(np.float32)
    t = (1 + n)[:, :, None] * cv + n[:, :, None] * Pout * oob[None]
    resid = S - t
    # the residual is a per-pixel scalar (same for every d)
    assert np.all(resid == resid[:, :, :1])
    assert np.array_equal(so.extract_index(S, so.COST), so.extract_index(t, so.COST))
    # pixels outside the margin box are untouched
    assert np.array_equal(S[n == 0], cv[n == 0])


def test_sgm_constant_cost_alternates():
    """F4 verified by hand: constant cost 1, Pout unused -> contributions +1, -1, +1 ... along a line."""
    cv = np.ones((1, 6, 3), np.float32)
    S = so.sgm(cv, 4, so.COST, 0.5, 0.5, (0, 0, 0, 0), 0.0)
    # Up2Down: every column is a 1-pixel line: contribution 2c - c = +1; Left2Right: +1,-1,+1,...
    expect = 1 + 1 + np.array([1, -1, 1, -1, 1, -1], np.float32)
    assert np.array_equal(S[0, :, 0], expect)


def test_sgm_score_by_hand():
    """Score branch, one line of 2 pixels, D=3 (sgm.h:218-255)."""
    cv = np.array([[[1, 5, 2], [3, 0, 4]]], np.float32)  # H=1, W=2
    P1, P2, Pout = np.float32(1), np.float32(3), np.float32(10)
    S = so.sgm(cv, 4, so.SCORE, P1, P2, (0, 0, 0, 0), Pout)
    # Up2Down (1-pixel lines): prev = 0 -> max_p = 0, max_a = 0 (prev[nd]); oob subtracts Pout
    up = np.zeros((2, 3), np.float32)
    for j in range(2):
        for d in range(3):
            up[j, d] = -Pout if j + d >= 2 else 0
    # Left2Right pixel 0: same as above with actual = c + a ; pixel 1 uses prev = actual(pixel 0)
    a0 = cv[0, 0] + up[0]
    max_p = a0.max()
    l2r1 = np.zeros(3, np.float32)
    for nd in range(3):
        cands = [a0[od] - (0 if od == nd else P1 if abs(od - nd) == 1 else P2) for od in range(3)]
        a = max(cands) - (Pout if 1 + nd >= 2 else 0)
        l2r1[nd] = (cv[0, 1, nd] + (a - max_p)) - cv[0, 1, nd]
    assert np.allclose(S[0, 0], cv[0, 0] + up[0] + up[0])
    assert np.allclose(S[0, 1], cv[0, 1] + up[1] + l2r1)


def test_sgm_16_runs_and_extends_8(rng):
    cv = rng.integers(0, 9, (7, 8, 4)).astype(np.float32)
    s8 = so.sgm(cv, 8, so.COST, 1, 2)
    s16 = so.sgm(cv, 16, so.COST, 1, 2)
    assert s16.shape == s8.shape and np.all(np.isfinite(s16))
    with pytest.raises(ValueError):
        so.sgm(cv, 5, so.COST, 1, 2)


# ------------------------------------------------------- winner / truncation / refinement --------------------
def test_extract_index_rules():
    """correlation_base.h:441-455: '<=' / '>=' scan -> last extremum wins; NaN never replaces the incumbent."""
    cv = np.array([[[3, 1, 1, 2], [5, 5, 5, 5], [np.nan, 1, 0, 2], [2, np.nan, 2, 3], [0.0, -0.0, 1, 1]]], np.float32)
    assert so.extract_index(cv, so.COST).tolist() == [[2, 3, 0, 2, 1]]
    assert so.extract_index(cv, so.SCORE).tolist() == [[0, 3, 0, 3, 3]]


def test_index_to_disp():
    idx = np.array([[0, 3], [7, 1]], np.int32)
    assert so.index_to_disp(idx, so.RIGHT_TO_LEFT, 2).tolist() == [[2, 5], [9, 3]]
    assert so.index_to_disp(idx, so.LEFT_TO_RIGHT, 2).tolist() == [[2, -1], [-5, 1]]


def test_truncated_cost_volume_borders(rng):
    """correlation_base.h:601-613 (Same): NaN when p<0, p>=D, j<h_r, j+p+h_r>=W, i<v_r, i+v_r>=H."""
    H, W, D, h_r, v_r, r = 7, 12, 5, 2, 1, 1
    cv = rng.uniform(0, 1, (H, W, D)).astype(np.float32)
    idx = rng.integers(0, D, (H, W)).astype(np.int32)
    t = so.truncated_cost_volume(cv, idx, h_r, v_r, r)
    assert t.shape == (H, W, 3)
    for i in range(H):
        for j in range(W):
            for k in range(3):
                p = idx[i, j] + k - r
                bad = p < 0 or p >= D or j < h_r or j + p + h_r >= W or i < v_r or i + v_r >= H
                if bad:
                    assert math.isnan(t[i, j, k])
                else:
                    assert t[i, j, k] == cv[i, j, p]
    both = so.truncated_cost_volume(cv, idx, h_r, v_r, r, sdir=so.TCV_BOTH)
    assert both.shape == (H, W, 5)
    same_slots = [0, 2, 3]  # d_d for d = 0, 1, 2 with r = 1
    assert np.array_equal(np.isnan(both[:, :, same_slots]), np.isnan(t))
    ok = ~np.isnan(t)
    assert np.array_equal(both[:, :, same_slots][ok], t[ok])
    rev = so.truncated_cost_volume(cv, idx, h_r, v_r, r, sdir=so.TCV_REVERSED)
    for i in range(H):
        for j in range(W):
            for k in range(3):
                p = idx[i, j] + k - r
                jp = j - (k - r)  # RightToLeft -> sgn = -1 (correlation_base.h:618-621)
                bad = p < 0 or p >= D or min(jp, j) < h_r or max(jp, j) + h_r >= W or i < v_r or i + v_r >= H
                if bad:
                    assert math.isnan(rev[i, j, k])
                    if k != r:
                        assert math.isnan(both[i, j, 2 * k + 1 - (k > r)])
                else:
                    assert rev[i, j, k] == cv[i, jp, p]
                    if k != r:
                        assert both[i, j, 2 * k + 1 - (k > r)] == cv[i, jp, p]


def test_refine_kernels():
    """cost_based_refinement.h:43-69."""
    f = np.float32
    cm1, c0, c1 = f(3.0), f(1.0), f(2.0)
    assert so.refine_triplet(so.PARABOLA, cm1, c0, c1) == f((cm1 - c1) / (2 * (c1 - 2 * c0 + cm1)))
    alpha = math.copysign(1.0, c0 - cm1) * max(abs(c0 - cm1), abs(c1 - c0))
    assert so.refine_triplet(so.EQUIANGULAR, cm1, c0, c1) == f((c1 - cm1) / f(2 * alpha))
    g = (math.log(3) - math.log(2)) / (2 * (math.log(2) - 2 * math.log(1) + math.log(3)))
    assert abs(so.refine_triplet(so.GAUSSIAN, cm1, c0, c1) - g) < 1e-6
    # no clamp, NaN propagates (cost_based_refinement.h:150-156)
    tcv = np.array([[[np.nan, 1, 2], [1, 1, 1]]], np.float32)
    out = so.refine_disp(tcv, np.array([[4, 2]], np.int32))
    assert math.isnan(out[0, 0]) and math.isnan(out[0, 1])  # 0/0


# ------------------------------------------------- second, independent restatement of sgm.h in pure Python ---------------------
def py_sgm(cv, n_dir, strategy, P1, P2, margins, Pout):
    """sgm.h:186-404 line by line with numpy float32 scalars (no shortcuts: the O(D^2) double loop, the start-position
    rule of :329-354 including the 'EndPos' starts that fall outside the loop guard)."""
    f = np.float32
    H, W, D = cv.shape
    left, top, right, bottom = margins
    P1, P2, Pout = f(P1), f(P2), f(Pout)
    steps = {  # directionTraits, :61-155: (stepsVertical, stepsHorizontal)
        "Up2Down": ((1, 1), (0, 0)), "Down2Up": ((-1, -1), (0, 0)), "Left2Right": ((0, 0), (1, 1)), "Right2Left": ((0, 0), (-1, -1)),
        "UpLeft2DownRight": ((1, 1), (1, 1)), "DownRight2UpLeft": ((-1, -1), (-1, -1)), "UpRight2DownLeft": ((1, 1), (-1, -1)),
        "DownLeft2UpRight": ((-1, -1), (1, 1)),
    }

    def start_pos(s):  # startPostInfos, :162-184
        if s[0] == 0 and s[1] == 0:
            return "NoStart"
        if s[0] >= 0 and s[1] >= 0:
            return "ZeroPos"
        return "EndPos"

    sgm = cv.copy()

    def traverse(name, start_i, start_j):
        sv_, sh_ = steps[name]
        prev = np.zeros(D, np.float32)
        c, i, j = 0, start_i, start_j
        while top <= i < H - bottom and left <= j < W - right:
            act = np.zeros(D, np.float32)
            fin = np.isfinite(prev)
            if strategy == so.SCORE:
                max_p = prev[fin].max() if fin.any() else f(-np.inf)
                for nd in range(D):
                    max_a = f(-np.inf)
                    for od in range(D):
                        p = prev[od]
                        if abs(od - nd) == 1:
                            p = f(p - P1)
                        if abs(od - nd) > 1:
                            p = f(p - P2)
                        if p > max_a and np.isfinite(p):
                            max_a = p
                    if j + nd >= W:
                        max_a = f(max_a - Pout)
                    act[nd] = cv[i, j, nd]
                    if np.isfinite(max_a) and np.isfinite(max_p):
                        act[nd] = f(act[nd] + f(max_a - max_p))
            else:
                min_p = prev[fin].min() if fin.any() else f(np.inf)
                for nd in range(D):
                    min_a = f(np.inf)
                    c_score = cv[i, j, nd]
                    for od in range(D):
                        p = prev[od]
                        if abs(od - nd) == 1:
                            p = f(p + P1)
                        if abs(od - nd) > 1:
                            p = f(p + P2)
                        if p < min_a and np.isfinite(p):
                            min_a = c_score  # sic, sgm.h:281-283
                    if j + nd >= W:
                        min_a = f(min_a + Pout)
                    act[nd] = c_score
                    if np.isfinite(min_a) and np.isfinite(min_p):
                        act[nd] = f(act[nd] + f(min_a - min_p))
            sgm[i, j, :] = (sgm[i, j, :] + (act - cv[i, j, :])).astype(np.float32)
            prev = act
            i += sv_[c % 2]
            j += sh_[c % 2]
            c += 1

    def add_direction(name):  # addDirectionalCost, :313-356
        sv_, sh_ = steps[name]
        col_start, row_start = start_pos(sv_), start_pos(sh_)
        if row_start != "NoStart":
            start_j = left if row_start == "ZeroPos" else W - right
            for start_i in range(top, H - bottom):
                traverse(name, start_i, start_j)
        if col_start != "NoStart":
            start_i = top if col_start == "ZeroPos" else H - bottom
            for start_j in range(left, W - right):
                traverse(name, start_i, start_j)

    with np.errstate(invalid="ignore", over="ignore"):
        for name in ("Up2Down", "Down2Up", "Left2Right", "Right2Left"):
            add_direction(name)
        if n_dir >= 8:
            for name in ("UpLeft2DownRight", "DownRight2UpLeft", "UpRight2DownLeft", "DownLeft2UpRight"):
                add_direction(name)
    return sgm


@pytest.mark.parametrize("strategy", [so.COST, so.SCORE])
@pytest.mark.parametrize("n_dir,margins", [(8, (0, 0, 0, 0)), (8, (1, 0, 2, 1)), (4, (0, 0, 0, 0))])
def test_oracle_sgm_equals_python_restatement(rng, strategy, n_dir, margins):
    for shape, integer in (((5, 7, 4), True), ((6, 5, 5), False), ((4, 4, 3), False)):
        cv = (rng.integers(0, 20, shape) if integer else rng.uniform(-1, 1, shape)).astype(np.float32)
        for P1, P2, Pout in ((0.001, 0.01, 100.0), (3.0, 1.0, 0.5)):
            exp = py_sgm(cv, n_dir, strategy, P1, P2, margins, Pout)
            for variant in (0, 1):
                got = so.sgm(cv, n_dir, strategy, P1, P2, margins, Pout, variant=variant)
                assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))


# ---- "textbook" SGM: an explicit second mode, not the reference's behaviour --------------
def py_sgm_textbook(cv, n_dir, strategy, P1, P2, margins, Pout):
    """Literal loops of the definition in oracle/stevi_oracle.c (so_sgm_textbook): O(D^2) candidate scan per pixel, float32
    arithmetic, the eight directions in the call order of sgm.h:379-388, every line of the margin box once."""
    f = np.float32
    H, W, D = cv.shape
    out = cv.copy()
    left, top, right, bottom = margins
    Hp, Wp = H - top - bottom, W - left - right
    if Hp <= 0 or Wp <= 0:
        return out
    cost = strategy == so.COST
    worst = f(np.inf) if cost else f(-np.inf)
    better = (lambda a, b: a if a < b else b) if cost else (lambda a, b: a if a > b else b)
    steps = [(1, 0), (-1, 0), (0, 1), (0, -1), (1, 1), (-1, -1), (1, -1), (-1, 1)][:n_dir]
    for di, dj in steps:
        starts = [(i, j) for i in range(Hp) for j in range(Wp) if not (0 <= i - di < Hp and 0 <= j - dj < Wp)]
        for (i, j) in starts:
            prev = np.zeros(D, f)
            while 0 <= i < Hp and 0 <= j < Wp:
                ii, jj = top + i, left + j
                c = cv[ii, jj]
                fin = [p for p in prev if np.isfinite(p)]
                ext = worst
                for p in fin:
                    ext = better(ext, p)
                act = c.copy()
                for nd in range(D):
                    a = worst
                    for od in range(D):
                        if not np.isfinite(prev[od]):
                            continue
                        pen = f(0) if od == nd else (f(P1) if abs(od - nd) == 1 else f(P2))
                        v = f(prev[od] + pen) if cost else f(prev[od] - pen)
                        if np.isfinite(v):
                            a = better(a, v)
                    if jj + nd >= W:
                        a = f(a + f(Pout)) if cost else f(a - f(Pout))
                    if np.isfinite(a) and np.isfinite(ext):
                        act[nd] = f(c[nd] + f(a - ext))
                out[ii, jj] += act - c
                prev = act
                i, j = i + di, j + dj
    return out


@pytest.mark.parametrize("strategy", [so.COST, so.SCORE])
@pytest.mark.parametrize("n_dir", [4, 8])
def test_textbook_sgm_oracle_equals_literal_loops(rng, strategy, n_dir):
    for shape, margins, Pout in (((5, 7, 6), (0, 0, 0, 0), 100.0), ((6, 4, 9), (1, 0, 0, 2), 3.5), ((1, 5, 3), (0, 0, 0, 0), 0.0)):
        cv = rng.uniform(0, 4, shape).astype(np.float32)
        cv[0, 0, 1] = np.inf
        if shape[0] > 2:
            cv[2, 1, 0] = np.nan
        got = so.sgm_textbook(cv, n_dir, strategy, 0.75, 2.5, margins, Pout)
        with np.errstate(invalid="ignore"):
            exp = py_sgm_textbook(cv, n_dir, strategy, 0.75, 2.5, margins, Pout)
        assert np.array_equal(np.isnan(got), np.isnan(exp))
        ok = ~np.isnan(exp)
        assert np.array_equal(got[ok], exp[ok])


def test_textbook_sgm_properties(rng):
    cv = rng.uniform(0, 10, (9, 11, 7)).astype(np.float32)
    # Cost and Score modes are mirror images
    assert np.array_equal(so.sgm_textbook(cv, 8, so.COST, 0.5, 2.0), -so.sgm_textbook(-cv, 8, so.SCORE, 0.5, 2.0))
    # a constant volume has nothing to propagate (Pout 0)
    one = np.ones((4, 5, 3), np.float32)
    assert np.array_equal(so.sgm_textbook(one, 8, so.COST, 1.0, 2.0, Pout=0.0), one)
    # unlike the reference as written (finding F4), the penalties matter in Cost mode ...
    assert not np.array_equal(so.sgm_textbook(cv, 8, so.COST, 0.5, 2.0), so.sgm_textbook(cv, 8, so.COST, 1.5, 2.0))
    assert np.array_equal(so.sgm(cv, 8, so.COST, 0.5, 2.0), so.sgm(cv, 8, so.COST, 1.5, 2.0))
    # ... and the result is invariant under a 180 degree rotation of the problem (the reference's direction set is not, F5); only
    # compare where Pout cannot act (Pout = 0) and up to float summation order
    a = so.sgm_textbook(cv, 8, so.COST, 0.5, 2.0, Pout=0.0)
    b = so.sgm_textbook(cv[::-1, ::-1].copy(), 8, so.COST, 0.5, 2.0, Pout=0.0)[::-1, ::-1]
    assert np.allclose(a, b, rtol=1e-5, atol=1e-4)


# ---- on-demand cost volumes and PatchMatch: literal Python restatements ------------------
def py_on_demand_features(func, img, h_r, v_r):
    """on_demand_features_volume.h:115-214 in plain loops: clamped window gather, then mean = sum / nF, v -= mean,
    norm = sqrt(sum v^2 / nF), v /= norm, all in float32 in the reference's order"""
    f = np.float32
    H, W, C = img.shape
    zm = func in (so.ZCC, so.ZNCC, so.ZSSD, so.ZSAD)
    nrm = func in (so.NCC, so.ZNCC)
    nF = (2 * v_r + 1) * (2 * h_r + 1) * C
    out = np.empty((H, W, nF), f)
    for i in range(H):
        for j in range(W):
            v = [img[min(H - 1, max(0, i + di)), min(W - 1, max(0, j + dj)), c] for di in range(-v_r, v_r + 1) for dj in range(-h_r, h_r + 1) for c in range(C)]
            if zm:
                mean = f(0)
                for x in v:
                    mean = f(mean + x)
                mean = f(mean / f(nF))
                v = [f(x - mean) for x in v]
            if nrm:
                norm = f(0)
                for x in v:
                    norm = f(norm + f(x * x))
                norm = f(np.sqrt(f(norm / f(nF))))
                v = [f(x / norm) for x in v]
            out[i, j] = v
    return out


def py_cost(func, fs, ft, i, j, disp, lower, upper):
    """costValue, on_demand_cost_volume.h:409-468; None when there is no value"""
    f = np.float32
    nd = len(disp)
    if any(d < lo or d > hi for d, lo, hi in zip(disp, lower, upper)):
        return None
    ti, tj = (i + disp[0], j + disp[1]) if nd == 2 else (i, j + disp[0])
    if not (0 <= ti < ft.shape[0] and 0 <= tj < ft.shape[1]):
        return None
    acc = f(0)
    for a, b in zip(fs[i, j], ft[ti, tj]):
        if func in (so.SSD, so.ZSSD):
            acc = f(acc + f(f(a - b) * f(a - b)))
        elif func in (so.SAD, so.ZSAD):
            acc = f(acc + abs(f(a - b)))
        else:
            acc = f(acc + f(a * b))
    return acc


@pytest.mark.parametrize("func", [so.ZNCC, so.NCC, so.ZSAD, so.SSD])
def test_on_demand_features_and_truncated_volume_literal(rng, func):
    img = rng.uniform(-1, 1, (5, 6, 2)).astype(np.float32)
    other = rng.uniform(-1, 1, (5, 6, 2)).astype(np.float32)
    with np.errstate(all="ignore"):
        fs, ft = py_on_demand_features(func, img, 1, 2), py_on_demand_features(func, other, 1, 2)
    assert np.array_equal(so.on_demand_features(func, img, 1, 2).view(np.uint32), fs.view(np.uint32))
    cost = so.func_strategy(func) == so.COST
    default = np.float32(np.finfo(np.float32).max if cost else np.finfo(np.float32).tiny)
    for (lower, upper) in (((0, 0), (2, 3)), ((-1, -2), (1, 2))):
        disp = np.stack([rng.integers(lower[0], upper[0] + 1, (5, 6)), rng.integers(lower[1], upper[1] + 1, (5, 6))], axis=-1).astype(np.int32)
        tcv = so.on_demand_truncated_cv(func, img, other, 1, 2, ((lower[0], upper[0]), (lower[1], upper[1])), disp, 1)
        for i in range(5):
            for j in range(6):
                for a in range(3):
                    for b in range(3):
                        # as written (:513-514): the argument of costValue is tap - radius + (disparity - lowerOffset)
                        arg = (a - 1 + disp[i, j, 0] - lower[0], b - 1 + disp[i, j, 1] - lower[1])
                        c = py_cost(func, fs, ft, i, j, arg, lower, upper)
                        exp = default if c is None else c
                        assert tcv[i, j, a, b].view(np.uint32) == np.float32(exp).view(np.uint32)


def py_pm_random(seed, it, i, j, k, dim):
    M = (1 << 64) - 1
    z = (seed + 0x9E3779B97F4A7C15 * (((it << 40) ^ (i << 20) ^ j ^ (k << 56) ^ (dim << 60) ^ 0x632BE59BD9B4E019) & M)) & M
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
    z = z ^ (z >> 31)
    v = z >> 32
    return v - (1 << 32) if v >= (1 << 31) else v


def py_in_range(v, lower, upper):
    rng_ = upper - lower + 1
    m = abs(v) % rng_  # |v % range| with C's truncating %: the remainder's magnitude is |v| mod range
    return m + lower


def py_patch_match(func, nd, img_s, img_t, h_r, v_r, lower, upper, n_iter, n_random, seed):
    """patchmatch.h:61-493 in plain loops (see the comment of so_cacheless_patch_match for the random stream)"""
    with np.errstate(all="ignore"):
        fs, ft = py_on_demand_features(func, img_s, h_r, v_r), py_on_demand_features(func, img_t, h_r, v_r)
    H, W = img_s.shape[:2]
    score = so.func_strategy(func) == so.SCORE
    sol = [[[py_in_range(py_pm_random(seed, 0xFFFFFFFF, i, j, 0, s), lower[s], upper[s]) for s in range(nd)] for j in range(W)] for i in range(H)]

    def test(i, j, cand):
        c_new = py_cost(func, fs, ft, i, j, cand, lower, upper)
        if c_new is None:
            return 0
        c_old = py_cost(func, fs, ft, i, j, sol[i][j], lower, upper)
        keep = (c_old is None or c_new >= c_old) if score else (c_old is not None and c_new <= c_old)
        if keep:
            sol[i][j] = list(cand)
        return int(keep)

    iterations = 0
    for it in range(n_iter):
        iterations += 1
        changes = 0
        inc0, inc1 = (1 if it % 4 < 2 else -1), (1 if it % 2 == 0 else -1)
        for i in range(H):
            for j in (range(0, W) if inc1 > 0 else range(W - 1, 0, -1)):
                if 0 <= j - inc1 < W:
                    changes += test(i, j, sol[i][j - inc1])
        for j in range(W):
            for i in (range(0, H) if inc0 > 0 else range(H - 1, 0, -1)):
                if 0 <= i - inc0 < H:
                    changes += test(i, j, sol[i - inc0][j])
        for i in range(H):
            for j in range(W):
                base = list(sol[i][j])
                n_chang = 0
                for k in range(n_random):
                    draw = [py_in_range(py_pm_random(seed, it, i, j, k, s), lower[s], upper[s]) for s in range(nd)]
                    delta = [int((draw[s] - base[s]) * (k + 1) / (n_random + 1)) for s in range(nd)]  # C division truncates towards zero
                    cand = [base[s] + delta[s] for s in range(nd)]
                    if cand == base:
                        cand = [b + 1 for b in base]
                    n_chang = test(i, j, cand)
                changes += n_chang
        if changes == 0:
            break
    return np.array(sol, np.int32), iterations


@pytest.mark.parametrize("func,nd", [(so.ZNCC, 2), (so.SAD, 1), (so.ZSSD, 2)])
def test_patch_match_oracle_equals_literal_loops(rng, func, nd):
    src = rng.uniform(0, 1, (7, 9, 2)).astype(np.float32)
    tgt = np.roll(src, (1, -2), (0, 1)) + rng.uniform(0, 0.01, src.shape).astype(np.float32)
    lower, upper = ((-2, -3), (2, 3)) if nd == 2 else ((-3,), (3,))
    search = ((lower[0], upper[0]), (lower[1], upper[1])) if nd == 2 else (lower[0], upper[0])
    for seed, n_iter, n_random in ((3, 4, 3), (2 ** 40 + 17, 2, 1)):
        assert so.pm_random(seed, 1, 2, 3, 1, 1) == py_pm_random(seed, 1, 2, 3, 1, 1)
        got, its = so.cacheless_patch_match(func, nd, src, tgt, 1, 1, search, n_iter, n_random, seed)
        exp, eits = py_patch_match(func, nd, src, tgt, 1, 1, lower, upper, n_iter, n_random, seed)
        assert np.array_equal(got, exp) and its == eits
