#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_regression.npz.

What these vectors are -- and are not.  The reference cannot be built in this image (MultidimArrays, Eigen, FFTW and Qt are not
here and there is no network: DESIGN.md section 2), so nothing in this directory was produced by running LibStevi.  The file
freezes the outputs of the CPU restatement (oracle/stevi_oracle.c) on small seeded inputs at the time it was generated: a later
change of the oracle that alters any of them fails tests/test_golden.py and has to be deliberate.  The vectors that do come from
the reference are in hand_computed.json: the cases SURVEY.md section 8(c) records from executing the reference's own
correlation_base.h / sgm.h during the survey, and cases worked out by hand from the cited lines.

    python tests/golden/make_golden.py        # rewrites oracle_regression.npz (inputs are derived from the fixed seed)

It also writes stereo_pair/*.stevimg: one small image pair with the frozen results of the census + SGM-8 chain, in the
reference's own array file format (io/image_io.h:48-168; the cost volume in the layout {W*D, 1, W} of
cross_correlations.h:220), so that the fixtures are readable by the reference's readStevimg as well.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as so  # noqa: E402

SEED = 20260401


def build():
    rng = np.random.default_rng(SEED)
    out = {}
    left = rng.uniform(-1, 1, (13, 17)).astype(np.float32)
    right = rng.uniform(-1, 1, (13, 17)).astype(np.float32)
    rgb = rng.uniform(-1, 1, (9, 11, 3)).astype(np.float32)
    out["in_left"], out["in_right"], out["in_rgb"] = left, right, rgb
    out["unfold_2_1"] = so.unfold(left, 2, 1)
    out["unfold_rgb_1_1_pad"] = so.unfold(rgb, 1, 1, (0, 2, 1, 0))
    out["census_4_4"] = so.census_transform(left, 4, 4)
    out["census_3_3"] = so.census_transform(left, 3, 3)
    for name, func in (("census", so.CENSUS), ("sad", so.SAD), ("ssd", so.SSD), ("ncc", so.NCC), ("zncc", so.ZNCC), ("zsad", so.ZSAD)):
        hr = 4 if func == so.CENSUS else 2
        out[f"cv_{name}_r2l"] = so.unfold_cost_volume(func, left, right, hr, hr, 9)
        out[f"cv_{name}_l2r_off"] = so.unfold_cost_volume(func, left, right, hr, hr, 7, so.LEFT_TO_RIGHT, -2)
    cvc, cvn = out["cv_census_r2l"], out["cv_zncc_r2l"]
    out["sgm8_cost_census"] = so.sgm(cvc, 8, so.COST, 0.001, 0.01, (0, 0, 0, 0), 100.0)
    out["sgm8_cost_census_literal"] = so.sgm(cvc, 8, so.COST, 0.001, 0.01, (0, 0, 0, 0), 100.0, variant=0)
    out["sgm4_cost_sad_margins"] = so.sgm(out["cv_sad_r2l"], 4, so.COST, 0.5, 2.0, (1, 2, 0, 1), 3.5)
    out["sgm8_score_zncc"] = so.sgm(cvn, 8, so.SCORE, 0.001, 0.01, (0, 0, 0, 0), 100.0)
    out["sgm8_score_zncc_p1_gt_p2"] = so.sgm(cvn, 8, so.SCORE, 0.3, 0.1, (1, 0, 2, 0), 1.0)
    idx = so.extract_index(out["sgm8_cost_census"], so.COST)
    out["idx_census"] = idx
    out["disp_census"] = so.index_to_disp(idx)
    sidx = so.extract_index(out["sgm8_score_zncc"], so.SCORE)
    out["idx_zncc"] = sidx
    for name, mode in (("same", so.TCV_SAME), ("reversed", so.TCV_REVERSED), ("both", so.TCV_BOTH)):
        out[f"tcv_{name}"] = so.truncated_cost_volume(out["sgm8_score_zncc"], sidx, 2, 2, 1, mode)
    for name, k in (("parabola", so.PARABOLA), ("equiangular", so.EQUIANGULAR), ("gaussian", so.GAUSSIAN)):
        out[f"refined_{name}"] = so.refine_disp(np.abs(out["tcv_same"]) + np.float32(0.05), sidx, k)
    cv2 = so.unfold_cost_volume_2d(so.ZNCC, left, right, 2, 2, (-1, 2), (-2, 3))
    out["cv2d_zncc"] = cv2
    idx2 = so.extract_index_2d(cv2, so.SCORE)
    out["idx2d_zncc"] = idx2
    t2 = so.truncated_bidirectional_cv(cv2, idx2, 1, 1)
    out["tcv2d"] = t2
    out["refined2d_iso"] = so.refine_disp_2d(t2, idx2, so.EQUIANGULAR, so.ISOTROPIC)
    out["refined2d_patch"] = so.refine_disp_2d_patch(t2, idx2, so.PARABOLA)
    out["downsample"] = so.average_pooling_downsample(left, 2)
    ht, hd = so.hierarchical_truncated_cv(so.ZNCC, 2, left, right, 1, 1, 8, 2)
    out["hier_tcv"], out["hier_disp"] = ht, hd
    mask = np.array([[1, 1, 2], [3, 0, 2], [3, 4, 4]], np.int32)
    out["compressor_mask"] = mask
    out["unfold_compressed"] = so.unfold_compressed(left, mask)
    out["sgm8_textbook_cost"] = so.sgm_textbook(out["cv_sad_r2l"], 8, so.COST, 0.5, 2.0, (0, 0, 0, 0), 3.0)
    out["on_demand_features_zncc"] = so.on_demand_features(so.ZNCC, rgb, 1, 1)
    pm, its = so.cacheless_patch_match(so.ZNCC, 2, rgb, np.roll(rgb, (1, -1), (0, 1)).copy(), 1, 1, ((-2, 2), (-2, 2)), 4, 3, seed=7)
    out["patch_match"], out["patch_match_iterations"] = pm, np.array([its], np.int32)
    return out


def build_stereo_pair():
    """(target, source) as (img_l, img_r), like benchmarkCrossCorrelationAlgorithms.cpp:93; census 9x9, D = 16, SGM-8."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import parallax_pair
    src, tgt, _ = parallax_pair(24, 40, 8, 6, 12, 2, 7, seed=SEED % 1000)
    cv = so.unfold_cost_volume(so.CENSUS, tgt, src, 4, 4, 16)
    sg = so.sgm(cv, 8, so.COST, 0.001, 0.01, (0, 0, 0, 0), 100.0)
    idx = so.extract_index(sg, so.COST)
    tcv = so.truncated_cost_volume(sg, idx, 4, 4, 1, so.TCV_SAME)
    H, W, D = cv.shape
    cv_ref_layout = np.ascontiguousarray(cv.transpose(0, 2, 1)).transpose(0, 2, 1)  # strides {W*D, 1, W}
    return {"img_l": tgt, "img_r": src, "cost_volume_census": cv_ref_layout, "sgm8_cost_volume": sg, "disp_index": idx,
            "refined_parabola": so.refine_disp(tcv, idx, so.PARABOLA)}


if __name__ == "__main__":
    from libstevi_amd.stevimg import write_stevimg
    pair_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stereo_pair")
    os.makedirs(pair_dir, exist_ok=True)
    for name, arr in build_stereo_pair().items():
        write_stevimg(os.path.join(pair_dir, name + ".stevimg"), arr)
    print(f"{pair_dir}: {sorted(os.listdir(pair_dir))}")
    data = build()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_regression.npz")
    np.savez_compressed(path, **data)
    print(f"{path}: {len(data)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")
