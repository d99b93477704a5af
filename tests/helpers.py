"""Shared test helpers: seeded synthetic stereo pairs and naive window formulas.

The pair generator restates the recipe of the reference's generateParallaxSquareImage
(test/test_correlation_utils.h:324-391) with a seeded numpy generator in place of
std::random_device, as SURVEY.md section 8(d) prescribes.
"""
import numpy as np


def parallax_pair(height, width, square, v_pos, h_pos, bg_parallax, sq_parallax, seed):
    """Returns (source, target, gt_disp): float32 U(-1,1) images, background and a square
    copied from source into target with the given parallaxes."""
    rng = np.random.default_rng(seed)
    height = max(height, square + v_pos)
    width = max(width, square + h_pos + max(abs(bg_parallax), abs(sq_parallax)))
    both = rng.uniform(-1.0, 1.0, size=(height, width, 2)).astype(np.float32)
    source = np.ascontiguousarray(both[:, :, 0])
    target = np.ascontiguousarray(both[:, :, 1])
    gt = np.full((height, width), bg_parallax, np.int32)
    jj = np.arange(width)
    in_sq_rows = (np.arange(height) >= v_pos) & (np.arange(height) < v_pos + square)
    in_sq_cols = (jj >= h_pos) & (jj < h_pos + square)
    sq_mask = in_sq_rows[:, None] & in_sq_cols[None, :]
    # background first, then the square (same write order as the reference)
    for mask, par in ((~sq_mask, bg_parallax), (sq_mask, sq_parallax)):
        ii, jj2 = np.nonzero(mask)
        ok = (jj2 + par >= 0) & (jj2 + par < width)
        target[ii[ok], jj2[ok] + par] = source[ii[ok], jj2[ok]]
    gt[sq_mask] = sq_parallax
    return source, target, gt


def naive_window_cost(name, w1, w2):
    """Formulas of the reference's in-test naive oracles (test/test_correlation_utils.h:9-310),
    evaluated in float64 (the reference tests compare at 1e-3)."""
    w1 = w1.astype(np.float64)
    w2 = w2.astype(np.float64)
    if name.startswith("Z"):
        w1 = w1 - w1.mean()
        w2 = w2 - w2.mean()
        name = name[1:]
    if name == "CC":
        return float((w1 * w2).sum())
    if name == "NCC":
        return float((w1 * w2).sum() / (np.sqrt((w1 * w1).sum()) * np.sqrt((w2 * w2).sum())))
    if name == "SSD":
        return float(((w1 - w2) ** 2).sum())
    if name == "SAD":
        return float(np.abs(w1 - w2).sum())
    raise ValueError(name)


# ---- synthetic quadric cost patches, as testCostRefinement.cpp builds them ------------------------------------------
def isotropic_patch(s, ex, ey, n=3):
    """cost(dx, dy) = s ((dx - ex)^2 + (dy - ey)^2) on an n x n grid centred on 0 (testCostRefinement.cpp:82-100);
    returns a (1, 1, n, n) float32 truncated volume, axis 2 = dx."""
    r = n // 2
    t = np.zeros((1, 1, n, n), np.float32)
    for a in range(n):
        for b in range(n):
            cx, cy = np.float32(a - r) - np.float32(ex), np.float32(b - r) - np.float32(ey)
            t[0, 0, a, b] = np.float32(s) * (cx * cx + cy * cy)
    return t


def rotated_quadric_patch(d1, d2, alpha, b, n):
    """cost(dx, dy) = p^T A p with p = (dx - b0, dy - b1), A = R(alpha) diag(d1, d2) R(alpha)^T
    (testCostRefinement.cpp:130-160, :176-208)."""
    R = np.array([[np.cos(alpha), -np.sin(alpha)], [np.sin(alpha), np.cos(alpha)]], np.float32)
    A = (R @ np.diag(np.array([d1, d2], np.float32)) @ R.T).astype(np.float32)
    r = n // 2
    t = np.zeros((1, 1, n, n), np.float32)
    for i in range(n):
        for j in range(n):
            p = np.array([i - r - b[0], j - r - b[1]], np.float32)
            t[0, 0, i, j] = np.float32(p @ (A @ p))
    return t


def refined_2d_mismatch(got, want, raw, tol=1e-4):
    """Fraction of pixels whose refined 2-D disparities differ by more than tol, not counting the pixels where one side
    zeroed its deltas (|delta| > 1 or NaN, cost_based_refinement.h:362-366) and the other did not: the rule is a hard
    threshold on a float result, so two correctly rounded evaluations can fall on either side of it."""
    got, want, raw = np.asarray(got), np.asarray(want), np.asarray(raw, np.float32)
    bad = np.any(np.abs(got - want) > tol, axis=-1)
    zero_g = np.all(got == raw, axis=-1)
    zero_w = np.all(want == raw, axis=-1)
    flipped = zero_g != zero_w
    return float(np.mean(bad & ~flipped)), float(np.mean(flipped))


# ---- hierarchical matching: the scene and the acceptance rule of testCorrelationHierarchical.cpp:27-132 -------------
def hierarchical_scene(depth, img_height=48, img_width=64, h_radius=2, v_radius=2, disp_w=16, seed=0):
    disp_square = (disp_w & ~7) // 2
    disp_bg = disp_square // 2
    square_size = max(h_radius, v_radius) * (2 << max(3, depth)) + 1
    v_pos, h_pos = img_height // 2 - square_size // 2, img_width // 2 - square_size // 2
    src, tgt, _ = parallax_pair(img_height, img_width, square_size, v_pos, h_pos, disp_bg, disp_square, seed)
    return dict(source=src, target=tgt, disp_square=disp_square, square=(v_pos, h_pos, square_size), disp_w=disp_w, h_radius=h_radius, v_radius=v_radius)


def hierarchical_acceptance(scene, depth, disp, disp_rev, upscale_disp_radius=2):
    """range check on both directions (:84-113) and >= 95 % exact matches inside the square (:115-140)"""
    H, W = disp.shape
    mo = 0
    for _ in range(depth):
        mo = mo * 2 + upscale_disp_radius
    mo *= 2
    assert disp.min() >= -mo and disp.max() <= scene["disp_w"] + mo
    assert disp_rev.min() >= -mo and disp_rev.max() <= scene["disp_w"] + mo
    v_pos, h_pos, size = scene["square"]
    vr, hr = scene["v_radius"], scene["h_radius"]
    rows = range(v_pos + 2 * depth * vr, min(v_pos + size - 3 * depth * vr, H - 2 * depth * vr))
    cols = range(h_pos + 2 * depth * hr, min(h_pos + size - 3 * depth * hr, W - 2 * depth * hr))
    expected = len(rows) * len(cols)
    count = sum(int(disp[i, j] == scene["disp_square"]) for i in rows for j in cols)
    assert expected > 0 and expected - count < 0.05 * expected, (count, expected)
