"""ctypes loader for the CPU oracle (oracle/stevi_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (libstevi_amd) never imports this module.

All functions take / return dense numpy arrays laid out as the C file documents
(images [H][W] or [H][W][C], volumes [H][W][D], words [H][W][nW]).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("STEVI_ORACLE_LIB") or os.path.join(_HERE, "libstevi_oracle.so")  # STEVI_ORACLE_LIB: e.g. the sanitizer build

# enum values (reference: correlation/matching_costs.h:38-53, correlation_base.h:31-45,
# cost_based_refinement.h:30-35)
CC, NCC, SSD, SAD, ZCC, ZNCC, ZSSD, ZSAD, HAMMING, CENSUS = 0, 1, 2, 3, 4, 5, 6, 7, 10, 11
COST, SCORE = 0, 1
LEFT_TO_RIGHT, RIGHT_TO_LEFT = 0, 1
TCV_SAME, TCV_REVERSED, TCV_BOTH = 0, 1, 2
EQUIANGULAR, PARABOLA, GAUSSIAN = 0, 1, 2


def build(force=False):
    if os.environ.get("STEVI_ORACLE_LIB"):
        return _LIB_PATH
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "stevi_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def granted_cpus():
    """Host CPUs this process may really use: the scheduler affinity, cut down to the cgroup's CPU quota (cpu.max).  The GPU boxes show
    256 CPUs and grant 16: an OpenMP region on 128 threads runs into CFS throttling there, which also stalls whatever the process times
    next by one 25 - 100 ms period (tools/stall_probe.py, profiles/r05a_stall_probe.jsonl)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                fields = f.read().split()
            if path.endswith("cpu.max"):
                quota, period = fields[0], int(fields[1])
            else:
                quota = fields[0]
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = int(f.read())
            if quota not in ("max", "-1") and int(quota) > 0 and period > 0:
                n = min(n, max(1, int(quota) // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        if "OMP_NUM_THREADS" not in os.environ:  # (an explicit request stands)
            _lib.so_set_num_threads(min(int(_lib.so_num_threads()), granted_cpus()))
        _lib.so_round_word_through_float.restype = C.c_uint32
        _lib.so_round_word_through_float.argtypes = [C.c_uint32]
        _lib.so_refine_triplet.restype = C.c_float
        _lib.so_refine_triplet.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float]
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _pad(pad):
    if pad is None:
        return None
    return (C.c_int * 4)(*[int(x) for x in pad])


def _img3(img):
    img = _f32(img)
    if img.ndim == 2:
        img = img[:, :, None]
    return np.ascontiguousarray(img)


def num_threads():
    return int(lib().so_num_threads())


def set_num_threads(n):
    lib().so_set_num_threads(int(n))


def func_strategy(func):
    return int(lib().so_func_strategy(int(func)))


def census_words(F):
    return (F - 1) // 32 + 1


ROTATE0, ROTATE90, ROTATE180, ROTATE270 = 0, 1, 2, 3


def unfold(img, h_r, v_r, pad=None, orientation=ROTATE0):
    """pad = (left, top, right, bottom) or None for the reference's auto padding."""
    img = _img3(img)
    H, W, Cc = img.shape
    Ho, Wo, F = C.c_int(), C.c_int(), C.c_int()
    lib().so_unfold_shape(H, W, Cc, h_r, v_r, _pad(pad), C.byref(Ho), C.byref(Wo), C.byref(F))
    out = np.empty((max(Ho.value, 0), max(Wo.value, 0), F.value), np.float32)
    if out.size:
        if orientation == ROTATE0:
            lib().so_unfold(_p(img), H, W, Cc, h_r, v_r, _pad(pad), _p(out))
        else:
            lib().so_unfold_oriented(_p(img), H, W, Cc, h_r, v_r, _pad(pad), int(orientation), _p(out))
    return out


def census_features(feat):
    feat = _f32(feat)
    H, W, F = feat.shape
    if F <= 1:
        return np.empty((0, 0, 0), np.uint32)
    out = np.empty((H, W, census_words(F)), np.uint32)
    rc = lib().so_census_features(_p(feat), H, W, F, _p(out))
    assert rc == 0
    return out


def census_transform(img, h_r, v_r, pad=None):
    return census_features(unfold(img, h_r, v_r, pad))


def set_float_overflow(zero):
    """Rule E2 when a target word rounds to 2^32: False / 0 = saturate to 0xFFFFFFFF (default), True / 1 = 0 (x86-64 without AVX-512)."""
    lib().so_set_float_overflow(int(bool(zero)))


def get_float_overflow():
    return int(lib().so_get_float_overflow())


def round_word_through_float(w):
    return int(lib().so_round_word_through_float(int(w) & 0xFFFFFFFF))


def channels_mean(feat):
    feat = _f32(feat)
    H, W, F = feat.shape
    out = np.empty((H, W), np.float32)
    lib().so_channels_mean(_p(feat), H, W, F, _p(out))
    return out


def channels_norm(feat):
    feat = _f32(feat)
    H, W, F = feat.shape
    out = np.empty((H, W), np.float32)
    lib().so_channels_norm(_p(feat), H, W, F, _p(out))
    return out


def channels_zeromean_norm(feat):
    feat = _f32(feat)
    H, W, F = feat.shape
    out = np.empty((H, W), np.float32)
    lib().so_channels_zeromean_norm(_p(feat), H, W, F, _p(out))
    return out


def feature_cost_volume(func, feat_l, feat_r, D, ddir=RIGHT_TO_LEFT, disp_lower=0):
    feat_l, feat_r = _f32(feat_l), _f32(feat_r)
    H, Wl, F = feat_l.shape
    Hr, Wr, Fr = feat_r.shape
    if H != Hr or F != Fr:
        return np.empty((0, 0, 0), np.float32)
    Ws = Wr if ddir == RIGHT_TO_LEFT else Wl
    cv = np.empty((H, Ws, D), np.float32)
    rc = lib().so_feature_cost_volume(int(func), _p(feat_l), _p(feat_r), H, Wl, Wr, F, int(ddir), int(disp_lower), int(D), _p(cv))
    if rc:
        return np.empty((0, 0, 0), np.float32)
    return cv


def unfold_cost_volume(func, img_l, img_r, h_r, v_r, D, ddir=RIGHT_TO_LEFT, disp_lower=0):
    img_l, img_r = _img3(img_l), _img3(img_r)
    Hl, Wl, Cc = img_l.shape
    Hr, Wr, Cr = img_r.shape
    if Hl != Hr or Cc != Cr:
        return np.empty((0, 0, 0), np.float32)
    Ws = Wr if ddir == RIGHT_TO_LEFT else Wl
    cv = np.empty((Hl, Ws, D), np.float32)
    rc = lib().so_unfold_cost_volume(int(func), _p(img_l), _p(img_r), Hl, Wl, Hr, Wr, Cc, int(h_r), int(v_r),
                                     int(ddir), int(disp_lower), int(D), _p(cv))
    if rc:
        return np.empty((0, 0, 0), np.float32)
    return cv


def sgm(cv, n_dir, strategy, P1, P2, margins=(0, 0, 0, 0), Pout=100.0, variant=1):
    """margins = (left, top, right, bottom); variant 0 = literal O(D^2) loops, 1 = O(D)."""
    cv = _f32(cv)
    H, W, D = cv.shape
    out = np.empty_like(cv)
    m = (C.c_int * 4)(*[int(x) for x in margins])
    rc = lib().so_sgm(int(n_dir), int(strategy), _p(cv), H, W, D, C.c_float(P1), C.c_float(P2), m, C.c_float(Pout), _p(out), int(variant))
    if rc:
        raise ValueError("unsupported number of directions")
    return out


def sgm_add_direction(sgm_cv, cv, direction, strategy, P1, P2, margins=(0, 0, 0, 0), Pout=100.0, variant=1):
    cv = _f32(cv)
    assert sgm_cv.dtype == np.float32 and sgm_cv.flags.c_contiguous and sgm_cv.shape == cv.shape
    H, W, D = cv.shape
    m = (C.c_int * 4)(*[int(x) for x in margins])
    rc = lib().so_sgm_add_direction(int(direction), int(strategy), _p(cv), H, W, D, C.c_float(P1), C.c_float(P2), m,
                                    C.c_float(Pout), _p(sgm_cv), int(variant))
    assert rc == 0
    return sgm_cv


def extract_index(cv, strategy):
    cv = _f32(cv)
    H, W, D = cv.shape
    idx = np.empty((H, W), np.int32)
    lib().so_extract_index(int(strategy), _p(cv), H, W, D, _p(idx))
    return idx


def index_to_disp(idx, ddir=RIGHT_TO_LEFT, offset=0):
    idx = _i32(idx)
    H, W = idx.shape
    out = np.empty_like(idx)
    lib().so_index_to_disp(int(ddir), _p(idx), H, W, int(offset), _p(out))
    return out


def truncated_cost_volume(cv, idx, h_r, v_r, r, sdir=TCV_SAME, ddir=RIGHT_TO_LEFT):
    cv, idx = _f32(cv), _i32(idx)
    H, W, D = cv.shape
    T = int(lib().so_truncated_cv_depth(int(sdir), int(r)))
    out = np.empty((H, W, T), np.float32)
    lib().so_truncated_cost_volume(int(sdir), int(ddir), _p(cv), _p(idx), H, W, D, int(h_r), int(v_r), int(r), _p(out))
    return out


def refine_triplet(kernel, cm1, c0, c1):
    return float(lib().so_refine_triplet(int(kernel), C.c_float(cm1), C.c_float(c0), C.c_float(c1)))


def refine_disp(tcv, raw, kernel=PARABOLA):
    tcv, raw = _f32(tcv), _i32(raw)
    H, W, T = tcv.shape
    out = np.empty((H, W), np.float32)
    rc = lib().so_refine_disp(int(kernel), _p(tcv), _p(raw), H, W, T, _p(out))
    if rc:
        return np.empty((0, 0), np.float32)
    return out


# ---- 2-D disparity volumes ---------------------------------------------------------
def unfold_cost_volume_2d(func, img_l, img_r, h_r, v_r, range0, range1, ddir=RIGHT_TO_LEFT):
    """range0 = (lower, upper) vertical offsets, range1 = horizontal; returns (H, W, Dh, Dw) or an empty array."""
    img_l, img_r = _img3(img_l), _img3(img_r)
    Hl, Wl, Cc = img_l.shape
    Hr, Wr, Cr = img_r.shape
    Dh, Dw = range0[1] - range0[0] + 1, range1[1] - range1[0] + 1
    if Hl != Hr or Wl != Wr or Cc != Cr or Dh <= 0 or Dw <= 0:
        return np.empty((0, 0, 0, 0), np.float32)
    cv = np.empty((Hl, Wl, Dh, Dw), np.float32)
    rc = lib().so_unfold_cost_volume_2d(int(func), _p(img_l), _p(img_r), Hl, Wl, Hr, Wr, Cc, int(h_r), int(v_r), int(ddir),
                                        int(range0[0]), int(range0[1]), int(range1[0]), int(range1[1]), _p(cv))
    return cv if rc == 0 else np.empty((0, 0, 0, 0), np.float32)


def extract_index_2d(cv, strategy):
    cv = _f32(cv)
    H, W, D1, D2 = cv.shape
    idx = np.empty((H, W, 2), np.int32)
    lib().so_extract_index_2d(int(strategy), _p(cv), H, W, D1, D2, _p(idx))
    return idx


def index_2d_to_disp(idx, lower0, lower1):
    idx = _i32(idx)
    out = np.empty_like(idx)
    lib().so_index_2d_to_disp(_p(idx), idx.shape[0], idx.shape[1], int(lower0), int(lower1), _p(out))
    return out


def truncated_bidirectional_cv(cv, idx, r0, r1):
    cv, idx = _f32(cv), _i32(idx)
    H, W, D1, D2 = cv.shape
    out = np.empty((H, W, 2 * r0 + 1, 2 * r1 + 1), np.float32)
    lib().so_truncated_bidirectional_cv(_p(cv), _p(idx), H, W, D1, D2, int(r0), int(r1), _p(out))
    return out


# ---- 2-D cost-based refinement (what stereo-match --refine calls) ---------------------
ISOTROPIC, ANISOTROPIC = 0, 1


def refine_disp_2d(tcv, raw, kernel=PARABOLA, isotropy=ISOTROPIC):
    tcv, raw = _f32(tcv), _i32(raw)
    H, W, T0, T1 = tcv.shape
    out = np.empty((H, W, 2), np.float32)
    rc = lib().so_refine_disp_2d(int(kernel), int(isotropy), _p(tcv), _p(raw), H, W, T0, T1, _p(out))
    return out if rc == 0 else np.empty((0, 0, 0), np.float32)


def refine_disp_2d_patch(tcv, raw, kernel=PARABOLA):
    tcv, raw = _f32(tcv), _i32(raw)
    H, W, T0, T1 = tcv.shape
    out = np.empty((H, W, 2), np.float32)
    rc = lib().so_refine_disp_2d_patch(int(kernel), _p(tcv), _p(raw), H, W, T0, T1, _p(out))
    if rc == 2:
        raise ValueError("patch refinement supports the Parabola and Gaussian kernels only")
    return out if rc == 0 else np.empty((0, 0, 0), np.float32)


# ---- hierarchical matching ----------------------------------------------------------
def average_pooling_downsample(img, win_h=2, win_v=None):
    """Interpolation::averagePoolingDownsample (interpolation/downsampling.h:67-178); 2-D or 3-D float image."""
    win_v = win_h if win_v is None else win_v
    x = _img3(img)
    H, W, Cc = x.shape
    Ho, Wo = C.c_int(), C.c_int()
    lib().so_downsample_shape(H, W, int(win_h), int(win_v), C.byref(Ho), C.byref(Wo))
    out = np.empty((Ho.value, Wo.value, Cc), np.float32)
    lib().so_average_pooling_downsample(_p(x), H, W, Cc, int(win_h), int(win_v), _p(out))
    return out if np.ndim(img) == 3 else out[:, :, 0]


def match_features(func, img, h_r, v_r):
    """unfold + getFeatureVolumeForMatchFunc: float (H,W,F) or uint32 census words (H,W,nW)."""
    x = _img3(img)
    H, W, Cc = x.shape
    F = (2 * h_r + 1) * (2 * v_r + 1) * Cc
    census = int(func) in (HAMMING, CENSUS)
    n = census_words(F) if census else F
    out = np.empty((H, W, n), np.uint32 if census else np.float32)
    lib().so_match_features(int(func), _p(x), H, W, Cc, int(h_r), int(v_r), _p(out))
    return out


def guided_cv(func, feat_l, feat_r, guide, radius, ddir=RIGHT_TO_LEFT):
    """computeGuidedCV (hierarchical.h:74-229) on feature volumes; returns (tcv (H,Ws,2r+1), disp_estimate (H,Ws))."""
    census = int(func) in (HAMMING, CENSUS)
    dt = np.uint32 if census else np.float32
    feat_l, feat_r = np.ascontiguousarray(feat_l, dt), np.ascontiguousarray(feat_r, dt)
    guide = _i32(guide)
    H, Wl, F = feat_l.shape
    Wr = feat_r.shape[1]
    Ws = Wr if ddir == RIGHT_TO_LEFT else Wl
    tcv = np.empty((H, Ws, 2 * radius + 1), np.float32)
    disp = np.empty((H, Ws), np.int32)
    rc = lib().so_guided_cv(int(func), int(census), _p(feat_l), _p(feat_r), H, Wl, Wr, F, int(ddir), _p(guide), guide.shape[0], guide.shape[1],
                            int(radius), _p(tcv), _p(disp))
    if rc:
        raise ValueError("guide must be at least 2x2")
    return tcv, disp


def hierarchical_truncated_cv(func, depth, img_l, img_r, h_radii, v_radii, disp_width, radius=2, ddir=RIGHT_TO_LEFT):
    """hiearchicalTruncatedCostVolume<matchFunc, depth> (hierarchical.h:232-319); h_radii / v_radii: int or depth+1 ints."""
    img_l, img_r = _img3(img_l), _img3(img_r)
    H, Wl, Cc = img_l.shape
    Wr = img_r.shape[1]
    if img_r.shape[0] != H or img_r.shape[2] != Cc:
        return np.empty((0, 0, 0), np.float32), np.empty((0, 0), np.int32)
    hr = [int(h_radii)] * (depth + 1) if np.isscalar(h_radii) else [int(x) for x in h_radii]
    vr = [int(v_radii)] * (depth + 1) if np.isscalar(v_radii) else [int(x) for x in v_radii]
    Ws = Wr if ddir == RIGHT_TO_LEFT else Wl
    tcv = np.empty((H, Ws, 2 * radius + 1), np.float32)
    disp = np.empty((H, Ws), np.int32)
    rc = lib().so_hierarchical_truncated_cv(int(func), int(depth), _p(img_l), _p(img_r), H, Wl, Wr, Cc, (C.c_int * len(hr))(*hr), (C.c_int * len(vr))(*vr),
                                            int(disp_width), int(radius), int(ddir), _p(tcv), _p(disp))
    if rc:
        return np.empty((0, 0, 0), np.float32), np.empty((0, 0), np.int32)
    return tcv, disp


def feature_cost_volume_2d(func, feat_l, feat_r, range0, range1, ddir=RIGHT_TO_LEFT):
    """featureVolume2CostVolume with a searchOffset<2> on raw feature volumes; (H, Ws, Dh, Dw) or an empty array."""
    feat_l, feat_r = _f32(feat_l), _f32(feat_r)
    H, Wl, F = feat_l.shape
    Dh, Dw = range0[1] - range0[0] + 1, range1[1] - range1[0] + 1
    if feat_r.shape[0] != H or feat_r.shape[2] != F or Dh <= 0 or Dw <= 0:
        return np.empty((0, 0, 0, 0), np.float32)
    Wr = feat_r.shape[1]
    cv = np.empty((H, Wr if ddir == RIGHT_TO_LEFT else Wl, Dh, Dw), np.float32)
    rc = lib().so_feature_cost_volume_2d(int(func), _p(feat_l), _p(feat_r), H, Wl, Wr, F, int(ddir), int(range0[0]), int(range0[1]), int(range1[0]),
                                         int(range1[1]), _p(cv))
    return cv if rc == 0 else np.empty((0, 0, 0, 0), np.float32)


# ---- A7 / A8 as stand-alone functions ----------------------------------------------------------------------------
def channels_zeromean_norm_given(feat, mean):
    feat, mean = _f32(feat), _f32(mean)
    H, W, F = feat.shape
    out = np.empty((H, W), np.float32)
    lib().so_channels_zeromean_norm_given(_p(feat), _p(mean), H, W, F, _p(out))
    return out


def affine_feature_volume(feat, mean=None, norm=None):
    """zeromeanFeatureVolume (mean only), normalizedFeatureVolume (norm only), zeromeanNormalizedFeatureVolume (both)."""
    feat = _f32(feat)
    H, W, F = feat.shape
    mean = None if mean is None else _f32(mean)
    norm = None if norm is None else _f32(norm)
    out = np.empty_like(feat)
    lib().so_affine_feature_volume(_p(feat), None if mean is None else _p(mean), None if norm is None else _p(norm), H, W, F, _p(out))
    return out


def feature_volume_for_match_func(func, feat):
    feat = _f32(feat)
    H, W, F = feat.shape
    census = int(func) in (HAMMING, CENSUS)
    out = np.empty((H, W, census_words(F) if census else F), np.uint32 if census else np.float32)
    lib().so_feature_volume_for_match_func(int(func), _p(feat), H, W, F, _p(out))
    return out


# ---- UnFoldCompressor ---------------------------------------------------------------
def unfold_compressed(img, mask, pad=None):
    """unfold(UnFoldCompressor(mask), img, padding) -- correlation/unfold.h:47-121, :346-471."""
    x = _img3(img)
    H, W, Cc = x.shape
    mask = _i32(mask)
    Ho, Wo, F = C.c_int(), C.c_int(), C.c_int()
    lib().so_unfold_compressed_shape(H, W, Cc, _p(mask), mask.shape[0], mask.shape[1], _pad(pad), C.byref(Ho), C.byref(Wo), C.byref(F))
    out = np.empty((max(Ho.value, 0), max(Wo.value, 0), F.value), np.float32)
    if out.size:
        lib().so_unfold_compressed(_p(x), H, W, Cc, _p(mask), mask.shape[0], mask.shape[1], _pad(pad), _p(out))
    return out


# ---- "textbook" SGM: NOT the reference's behaviour, see the C comment ------------------
def sgm_textbook(cv, n_dir, strategy, P1, P2, margins=(0, 0, 0, 0), Pout=100.0):
    cv = _f32(cv)
    H, W, D = cv.shape
    out = np.empty_like(cv)
    m = (C.c_int * 4)(*[int(x) for x in margins])
    rc = lib().so_sgm_textbook(int(n_dir), int(strategy), _p(cv), H, W, D, C.c_float(P1), C.c_float(P2), m, C.c_float(Pout), _p(out))
    if rc:
        raise ValueError("unsupported number of directions")
    return out


# ---- on-demand (cacheless) cost volumes and PatchMatch (what examples/stereo-match runs) -----
def on_demand_features(func, img, h_r, v_r):
    x = _img3(img)
    H, W, Cc = x.shape
    out = np.empty((H, W, (2 * h_r + 1) * (2 * v_r + 1) * Cc), np.float32)
    lib().so_on_demand_features(int(func), _p(x), H, W, Cc, int(h_r), int(v_r), _p(out))
    return out


def _od_range(search_dims, search_range):
    """search_range: (lower, upper) for stereo, ((lower0, upper0), (lower1, upper1)) for flow -> C array lower0, upper0, lower1, upper1"""
    if search_dims == 1:
        return (C.c_int * 4)(0, 0, int(search_range[0]), int(search_range[1]))
    return (C.c_int * 4)(int(search_range[0][0]), int(search_range[0][1]), int(search_range[1][0]), int(search_range[1][1]))


def on_demand_truncated_cv(func, img_s, img_t, h_r, v_r, search_range, disp, radius=1):
    s, t = _img3(img_s), _img3(img_t)
    disp = _i32(disp)
    nd = disp.shape[2]
    T = 2 * radius + 1
    out = np.empty(s.shape[:2] + (T,) * nd, np.float32)
    rc = lib().so_on_demand_truncated_cv(int(func), nd, _p(s), s.shape[0], s.shape[1], _p(t), t.shape[0], t.shape[1], s.shape[2], int(h_r), int(v_r),
                                         _od_range(nd, search_range), _p(disp), int(radius), _p(out))
    if rc:
        raise ValueError("unsupported on-demand configuration")
    return out


def pm_random(seed, it, i, j, k, dim):
    lib().so_pm_random.restype = C.c_int32
    return int(lib().so_pm_random(C.c_uint64(seed), C.c_uint32(it & 0xFFFFFFFF), C.c_uint32(i), C.c_uint32(j), C.c_uint32(k), C.c_uint32(dim)))


def cacheless_patch_match(func, search_dims, img_s, img_t, h_r, v_r, search_range, n_iter=5, n_random=4, seed=0, init=None):
    """init: (H, W, search_dims) int32 initial solution in place of the random draw (the reference's `initializer` callback's map)."""
    s, t = _img3(img_s), _img3(img_t)
    sol = np.empty(s.shape[:2] + (search_dims,), np.int32)
    its = C.c_int(0)
    ini = None if init is None else _i32(init)
    rc = lib().so_cacheless_patch_match_init(int(func), int(search_dims), _p(s), s.shape[0], s.shape[1], _p(t), t.shape[0], t.shape[1], s.shape[2], int(h_r),
                                             int(v_r), _od_range(search_dims, search_range), int(n_iter), int(n_random), C.c_uint64(seed),
                                             None if ini is None else _p(ini), _p(sol), C.byref(its))
    if rc:
        return np.empty((0, 0, 0), np.int32), 0
    return sol, its.value


def patch_match(func, search_dims, feat_s, feat_t, search_range, n_iter=5, n_random=4, seed=0, init=None):
    """patchMatch (patchmatch.h:496-558) on feature volumes (H, W, F) float32."""
    s, t = np.ascontiguousarray(feat_s, np.float32), np.ascontiguousarray(feat_t, np.float32)
    if s.shape[2] != t.shape[2]:
        return np.empty((0, 0, 0), np.int32), 0
    sol = np.empty(s.shape[:2] + (search_dims,), np.int32)
    its = C.c_int(0)
    ini = None if init is None else _i32(init)
    rc = lib().so_patch_match(int(func), int(search_dims), _p(s), s.shape[0], s.shape[1], _p(t), t.shape[0], t.shape[1], s.shape[2],
                              _od_range(search_dims, search_range), int(n_iter), int(n_random), C.c_uint64(seed), None if ini is None else _p(ini),
                              _p(sol), C.byref(its))
    if rc:
        return np.empty((0, 0, 0), np.int32), 0
    return sol, its.value
