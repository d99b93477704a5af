"""numpy restatement of the disparity-shard protocol (test infrastructure, mirrors svh_census_shard_keys /
svh_census_shard_finish): regional winner keys per shard, MIN reduction, line recurrences, winner."""
import numpy as np

IDX_BITS = 12
IDX_MASK = (1 << IDX_BITS) - 1
KEY_NONE = 0x7FFFFFFF


def shard_keys(cv_shard, begin, W, cv_full=None):
    """cv_shard: (H, W, Dl) integer Hamming costs of global indices begin..begin+Dl-1 -> (H, W, 2) int32.
    cv_full (the whole range): plane 1 is the Pout region's winner over ALL shards instead of the shard's own -- what
    svh_census_shard_keys writes when svh_census_shard_region1_is_global() holds (every paying disparity sees the zero vector)."""
    if cv_full is not None:
        return np.stack([shard_keys(cv_shard, begin, W)[:, :, 0], shard_keys(cv_full, 0, W)[:, :, 1]], axis=2)
    H, Wc, Dl = cv_shard.shape
    d = begin + np.arange(Dl)
    key = (cv_shard.astype(np.int64) << IDX_BITS) | (IDX_MASK - d)[None, None, :]
    oob = (np.arange(Wc)[:, None] + d[None, :] >= W)[None]  # (1, W, Dl)
    k0 = np.where(oob, KEY_NONE, key).min(axis=2)
    k1 = np.where(oob, key, KEY_NONE).min(axis=2)
    return np.stack([k0, k1], axis=2).astype(np.int32)


def finish(keys, n_dir, Pout, margins=(0, 0, 0, 0)):
    """reduced keys -> selected index map, following sgm.h's effective passes in integer arithmetic."""
    H, W, _ = keys.shape
    k0, k1 = keys[:, :, 0].astype(np.int64), keys[:, :, 1].astype(np.int64)
    c0, c1 = k0 >> IDX_BITS, k1 >> IDX_BITS
    g = np.minimum(np.where(k0 == KEY_NONE, 1 << 24, 2 * c0), np.where(k1 == KEY_NONE, 1 << 24, 2 * c1 + int(Pout)))
    left, top, right, bottom = margins
    Hp, Wp = H - top - bottom, W - left - right
    msum = np.zeros((H, W), np.int64)
    nvis = np.zeros((H, W), np.int64)
    passes = []
    if n_dir >= 4 and Hp > 0 and Wp > 0:
        passes += [[(top, left + l, 1, 0) for l in range(Wp)], [(top + l, left, 0, 1) for l in range(Hp)]]
    if n_dir >= 8 and Hp > 0 and Wp > 0:
        passes += [[(top + l, left, 1, 1) for l in range(Hp)], [(top, left + l, 1, 1) for l in range(Wp)],
                   [(top, left + l, 1, -1) for l in range(Wp)], [(top + l, left, -1, 1) for l in range(Hp)]]
    for lines in passes:
        for (i, j, di, dj) in lines:
            mp = 0
            while top <= i < H - bottom and left <= j < W - right:
                msum[i, j] += mp
                nvis[i, j] += 1
                mp = g[i, j] - mp
                i += di
                j += dj
    mul = 1 + nvis
    v0 = mul * c0 - msum
    v1 = mul * c1 + nvis * int(Pout) - msum
    take1 = (k1 != KEY_NONE) & ((k0 == KEY_NONE) | (v1 <= v0))
    return np.where(take1, IDX_MASK - (k1 & IDX_MASK), IDX_MASK - (k0 & IDX_MASK)).astype(np.int32)


def band_winner(img_l, img_r, h_r, v_r, D, rows, n_dir, Pout, margins=(0, 0, 0, 0), oracle=None):
    """numpy restatement of svh_census_band_match (RightToLeft): rows (begin, count) of the selected-index map computed from the
    band's own rows plus v_r halo rows only.  The winner of a pixel is argmin_d [(1 + n) c + n Pout (j + d >= W)], ties to the
    larger index, n = number of effective SGM passes visiting the pixel's position in the WHOLE image (sgm.h:329-354)."""
    so = oracle
    H, W = img_r.shape[0], img_r.shape[1]
    begin, count = rows
    a, b = max(0, begin - v_r), min(H, begin + count + v_r)
    cv = so.unfold_cost_volume(so.CENSUS, np.ascontiguousarray(img_l[a:b]), np.ascontiguousarray(img_r[a:b]), h_r, v_r, D)  # (b - a, W, D)
    cv = cv[begin - a:begin - a + count].astype(np.int64)
    left, top, right, bottom = margins
    Hp, Wp = H - top - bottom, W - left - right
    i = np.arange(begin, begin + count)[:, None] - top
    j = np.arange(W)[None, :] - left
    inside = (i >= 0) & (i < Hp) & (j >= 0) & (j < Wp)
    n6 = 2 + (i >= j) + (j >= i) + (i + j < Wp) + (i + j < Hp)
    n = np.where(inside, n6 if n_dir >= 8 else (2 if n_dir >= 4 else 0), 0) if Hp > 0 and Wp > 0 else np.zeros((count, W), np.int64)
    d = np.arange(D)
    oob = (np.arange(W)[:, None] + d[None, :] >= W)[None]
    S = (1 + n)[:, :, None] * cv + n[:, :, None] * int(Pout) * oob
    return (D - 1 - np.argmin(S[:, :, ::-1], axis=2)).astype(np.int32)  # last minimum
