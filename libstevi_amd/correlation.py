"""Host-side mirror of the StereoVision::Correlation functions on the hot path.

Same names, argument meaning and error behaviour as the reference templates (template parameters become
leading keyword-free arguments).  Arrays are either numpy arrays (host memory: copied in and out by the C
library) or torch CUDA tensors (used in place on the tensor's device, enqueued on torch's current stream);
results come back as the same kind.  Where the reference returns an empty Multidim::Array (shape mismatch,
bad truncated-volume depth, single-channel census) an empty array is returned.

All compute goes through the C ABI of libstevi_hip.so (libstevi_amd/_capi.py); nothing here computes on the CPU.
"""
import ctypes as C
import enum

import numpy as np

from . import _capi


class matchingFunctions(enum.IntEnum):  # correlation/matching_costs.h:38-53
    CC = 0
    NCC = 1
    SSD = 2
    SAD = 3
    ZCC = 4
    ZNCC = 5
    ZSSD = 6
    ZSAD = 7
    HAMMING = 10
    CENSUS = 11


class dispExtractionStartegy(enum.IntEnum):  # correlation/correlation_base.h:31-34 (spelling as in the reference)
    Cost = 0
    Score = 1


class dispDirection(enum.IntEnum):  # correlation_base.h:36-39
    LeftToRight = 0
    RightToLeft = 1


class truncatedCostVolumeDirection(enum.IntEnum):  # correlation_base.h:41-45
    Same = 0
    Reversed = 1
    Both = 2


class InterpolationKernel(enum.IntEnum):  # correlation/cost_based_refinement.h:30-35
    Equiangular = 0
    Parabola = 1
    Gaussian = 2


class IsotropyHypothesis(enum.IntEnum):  # correlation/cost_based_refinement.h:37-41
    Isotropic = 0
    Anisotropic = 1


class Margins:  # utils/margins.h:24-92
    def __init__(self, *a):
        if len(a) == 0:
            l = t = r = b = 0
        elif len(a) == 1:
            l = t = r = b = a[0]
        elif len(a) == 2:
            l = r = a[0]
            t = b = a[1]
        elif len(a) == 4:
            l, t, r, b = a
        else:
            raise TypeError("Margins takes 0, 1, 2 or 4 integers")
        self._v = (int(l), int(t), int(r), int(b))

    def left(self): return self._v[0]
    def top(self): return self._v[1]
    def right(self): return self._v[2]
    def bottom(self): return self._v[3]
    def as_tuple(self): return self._v


class PaddingMargins(Margins):  # utils/margins.h:95-163: no argument = automatic padding
    def __init__(self, *a):
        super().__init__(*a)
        self._auto = len(a) == 0

    def isAuto(self): return self._auto


class searchOffset1:  # searchOffset<1>, correlation_base.h:288-409
    def __init__(self, lower, upper):
        self.lower, self.upper = int(lower), int(upper)

    def dimRange(self): return self.upper - self.lower + 1


def matchFuncStrategy(matchFunc):
    """MatchingFunctionTraits<f>::extractionStrategy (matching_costs.h:419-685)."""
    return dispExtractionStartegy.Score if int(matchFunc) in (0, 1, 4, 5) else dispExtractionStartegy.Cost


# ------------------------------------------------------------------------------------------------ plumbing
_NP_DTYPES = {np.dtype(np.float32): _capi.F32, np.dtype(np.int32): _capi.I32, np.dtype(np.uint32): _capi.U32,
              np.dtype(np.uint8): _capi.U8, np.dtype(np.uint64): _capi.U64, np.dtype(np.int16): _capi.I16, np.dtype(np.uint16): _capi.U16}


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _torch_code(t):
    import torch
    table = {torch.float32: _capi.F32, torch.int32: _capi.I32, torch.uint8: _capi.U8, torch.int16: _capi.I16}
    for name, code in (("uint32", _capi.U32), ("uint64", _capi.U64), ("uint16", _capi.U16)):
        if hasattr(torch, name):
            table[getattr(torch, name)] = code
    if t.dtype == torch.int64:  # 8-byte keys travel as int64 tensors (torch.distributed has no uint64 reductions)
        return _capi.U64
    return table[t.dtype]


class _Ctx:
    """One svh_context per (device, stream)."""
    _cache = {}

    @classmethod
    def get(cls, device_index=None):
        lib = _capi.load()
        stream = None
        if device_index is None:
            key = ("default",)
        else:
            import torch
            stream = torch.cuda.current_stream(device_index).cuda_stream
            key = (device_index, stream)
        h = cls._cache.get(key)
        if h is None:
            h = C.c_void_p()
            if device_index is not None:
                import torch
                with torch.cuda.device(device_index):
                    st = lib.svh_context_create(C.byref(h), device_index, C.c_void_p(stream))
            else:
                st = lib.svh_context_create(C.byref(h), -1, None)
            if st != _capi.OK:
                raise _capi.SvhError(st, lib.svh_status_string(st).decode() + " (the HIP path has no CPU fallback)")
            cls._cache[key] = h
        return h


def context_for(x=None):
    """The svh_context handle used for array x (exposed for bench.py's profiling calls)."""
    if x is not None and _is_torch(x):
        if not x.is_cuda:
            raise TypeError("torch tensors must live on a HIP device; pass numpy arrays for host memory")
        return _Ctx.get(x.device.index if x.device.index is not None else 0)
    return _Ctx.get(None)


def _desc(x):
    a = _capi.SvhArray()
    if _is_torch(x):
        a.data = x.data_ptr()
        a.dtype = _torch_code(x)
        a.memspace = _capi.DEVICE
        shape, strides = tuple(x.shape), tuple(x.stride())
    else:
        if x.dtype not in _NP_DTYPES:
            raise TypeError(f"unsupported dtype {x.dtype}")
        a.data = x.ctypes.data
        a.dtype = _NP_DTYPES[x.dtype]
        a.memspace = _capi.HOST
        shape = x.shape
        strides = tuple(s // x.itemsize for s in x.strides)
        if any(s % x.itemsize for s in x.strides):
            raise ValueError("strides must be multiples of the item size")
    a.ndim = len(shape)
    for k in range(len(shape)):
        a.shape[k] = shape[k]
        a.strides[k] = strides[k]
    return a


def _like(x, shape, dtype):
    """dtype: 'f32' | 'i32' | 'u32' | 'u64'"""
    if _is_torch(x):
        import torch
        td = {"f32": torch.float32, "i32": torch.int32, "u32": getattr(torch, "uint32", torch.int32), "u64": torch.int64}[dtype]
        return torch.empty(shape, dtype=td, device=x.device)
    nd = {"f32": np.float32, "i32": np.int32, "u32": np.uint32, "u64": np.uint64}[dtype]
    return host_empty(shape, nd)


class _PinnedBlock:
    """A block of svh_host_alloc (page-locked host memory) that a numpy array is laid over; released with the last view of it."""

    def __init__(self, nbytes):
        p = C.c_void_p()
        st = _capi.load().svh_host_alloc(nbytes, C.byref(p))
        if st != _capi.OK or not p.value:
            raise MemoryError(f"svh_host_alloc({nbytes})")
        self.ptr = p.value
        self.__array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 3}

    def __del__(self):
        try:
            _capi.load().svh_host_free(C.c_void_p(self.ptr))
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass


PINNED_RESULTS_FROM = 1 << 20  # bytes


def host_empty(shape, dtype):
    """np.empty for the host arrays this module returns: arrays of a megabyte and more are laid over page-locked memory
    (svh_host_alloc), which the DMA engines write at the link's rate -- and read at that rate when the array is passed to the next
    function of a chain (unfoldBasedCostVolume -> sgmCostVolume -> extractSelectedIndex on numpy arrays moves 8.5 GB at 1080p x 256).
    An ordinary numpy array in every other respect (the block goes back to the library's cache with the last view of it)."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    if n < PINNED_RESULTS_FROM:
        return np.empty(shape, dtype)
    try:
        block = _PinnedBlock(n)
    except (MemoryError, ImportError):
        return np.empty(shape, dtype)
    return np.asarray(block)[:n].view(dtype).reshape(shape)


def _empty_like(x, ndim, dtype):
    return _like(x, (0,) * ndim, dtype)


def _prep(x, dtype=None):
    """numpy: make sure the dtype is right; torch: must already be a CUDA tensor of the right dtype."""
    if _is_torch(x):
        return x
    x = np.asarray(x)
    if dtype is not None and x.dtype != dtype:
        x = x.astype(dtype)
    return x


_VOLUME_DTYPES = (np.float32, np.uint8, np.int16, np.uint16, np.int32, np.uint32)


def _prep_volume(x):
    """sgmCostVolume's T_CV: float32 or an integer type the reference casts to float as it reads (converted on the device);
    any other numpy dtype is converted to float32 on the host."""
    if _is_torch(x):
        return x
    x = np.asarray(x)
    return x if x.dtype in [np.dtype(t) for t in _VOLUME_DTYPES] else _prep(x, np.float32)


def _prep_image(x):
    """Images are float32 or uint8 (svh_array dtype SVH_U8: widened on the device, exact for the functions that accept it);
    any other numpy dtype is converted to float32 on the host."""
    if _is_torch(x):
        return x
    x = np.asarray(x)
    return x if x.dtype == np.uint8 else _prep(x, np.float32)


def _check(ctx, status):
    if status in (_capi.OK, _capi.EMPTY_RESULT):
        return status
    lib = _capi.load()
    msg = lib.svh_last_error(ctx).decode() or lib.svh_status_string(status).decode()
    raise _capi.SvhError(status, msg)


def _pad_arg(padding):
    if padding is None or (isinstance(padding, PaddingMargins) and padding.isAuto()):
        return None
    v = padding.as_tuple() if isinstance(padding, Margins) else tuple(padding)
    return (C.c_int32 * 4)(*v)


def _search_range(r):
    if isinstance(r, searchOffset1):
        return r.lower, r.dimRange()
    if isinstance(r, (tuple, list)):
        return int(r[0]), int(r[1]) - int(r[0]) + 1
    return 0, int(r)


# ------------------------------------------------------------------------------------------------ functions
class UnfoldPatchOrientation(enum.IntEnum):  # correlation/unfold.h:139-144
    Rotate0 = 0
    Rotate90 = 1
    Rotate180 = 2
    Rotate270 = 3


def unfold(h_radius, v_radius, in_data, padding=None, orientation=UnfoldPatchOrientation.Rotate0):
    """unfold<T,T>(h_radius, v_radius, in_data, padding, orientation) -- correlation/unfold.h:247-344; with an UnFoldCompressor
    as first argument (unfold(compressor, in_data, padding), :346-471) it forwards to unfoldCompressed."""
    if isinstance(h_radius, UnFoldCompressor):
        return unfoldCompressed(h_radius, v_radius, in_data)
    lib = _capi.load()
    x = _prep_image(in_data)
    ctx = context_for(x)
    d = _desc(x)
    shp = (C.c_int64 * 3)()
    if lib.svh_unfold_shape(C.byref(d), h_radius, v_radius, _pad_arg(padding), shp) != _capi.OK:
        raise ValueError("bad unfold arguments")
    if shp[0] <= 0 or shp[1] <= 0:
        return _empty_like(x, 3, "f32")
    out = _like(x, (shp[0], shp[1], shp[2]), "f32")
    _check(ctx, lib.svh_unfold_oriented(ctx, C.byref(d), h_radius, v_radius, _pad_arg(padding), int(orientation), C.byref(_desc(out))))
    return out


def censusFeatures(baseFeatures):
    """censusFeatures -- correlation/census.h:69-115."""
    lib = _capi.load()
    x = _prep(baseFeatures, np.float32)
    ctx = context_for(x)
    H, W, F = x.shape
    if F <= 1:
        return _empty_like(x, 3, "u32")
    out = _like(x, (H, W, (F - 1) // 32 + 1), "u32")
    st = _check(ctx, lib.svh_census_features(ctx, C.byref(_desc(x)), C.byref(_desc(out))))
    return out if st == _capi.OK else _empty_like(x, 3, "u32")


def censusTransform2D(input, h_radius, v_radius, padding=None):
    """censusTransform2D -- correlation/census.h:117-131."""
    lib = _capi.load()
    x = _prep_image(input)
    ctx = context_for(x)
    d = _desc(x)
    shp = (C.c_int64 * 3)()
    if lib.svh_unfold_shape(C.byref(d), h_radius, v_radius, _pad_arg(padding), shp) != _capi.OK:
        raise ValueError("bad census arguments")
    if shp[0] <= 0 or shp[1] <= 0 or shp[2] <= 1:
        return _empty_like(x, 3, "u32")
    out = _like(x, (shp[0], shp[1], (shp[2] - 1) // 32 + 1), "u32")
    st = _check(ctx, lib.svh_census_transform(ctx, C.byref(d), h_radius, v_radius, _pad_arg(padding), C.byref(_desc(out))))
    return out if st == _capi.OK else _empty_like(x, 3, "u32")


def featureVolume2CostVolume(matchFunc, feature_vol_l, feature_vol_r, searchRange, dDir=dispDirection.RightToLeft):
    """featureVolume2CostVolume<matchFunc,...,dDir,float> -- correlation/cross_correlations.h:724-738.
    searchRange: int / searchOffset1 -> (H,W,D) volume; searchOffset2 -> (H,W,Dh,Dw) volume (aggregateCost :310-374)."""
    lib = _capi.load()
    l, r = _prep(feature_vol_l, np.float32), _prep(feature_vol_r, np.float32)
    ctx = context_for(l)
    src = r if int(dDir) == dispDirection.RightToLeft else l
    if isinstance(searchRange, searchOffset2):
        o = searchRange
        Dh, Dw = o.upper0 - o.lower0 + 1, o.upper1 - o.lower1 + 1
        if l.shape[0] != r.shape[0] or Dh <= 0 or Dw <= 0:
            return _empty_like(l, 4, "f32")
        out = _like(l, (src.shape[0], src.shape[1], Dh, Dw), "f32")
        st = _check(ctx, lib.svh_feature_cost_volume_2d(ctx, int(matchFunc), int(dDir), C.byref(_desc(l)), C.byref(_desc(r)), o.lower0, o.upper0,
                                                        o.lower1, o.upper1, C.byref(_desc(out))))
        return out if st == _capi.OK else _empty_like(l, 4, "f32")
    lower, D = _search_range(searchRange)
    if l.shape[0] != r.shape[0]:
        return _empty_like(l, 3, "f32")
    out = _like(l, (src.shape[0], src.shape[1], D), "f32")
    st = _check(ctx, lib.svh_feature_cost_volume(ctx, int(matchFunc), int(dDir), C.byref(_desc(l)), C.byref(_desc(r)), lower, D,
                                                 C.byref(_desc(out))))
    return out if st == _capi.OK else _empty_like(l, 3, "f32")


def unfoldBasedCostVolume(matchFunc, img_l, img_r, h_radius, v_radius, disp_width, dDir=dispDirection.RightToLeft, keep_minima=False, keep_winner=False):
    """unfoldBasedCostVolume<matchFunc,...> -- correlation/cross_correlations.h:740-765.
    disp_width: an int (disp_t overload) or a searchOffset1 / (lower, upper) pair.

    keep_minima (device tensors, CENSUS / HAMMING only; off by default): also keep the volume's per-pixel regional minima with the returned
    tensor, so that a later sgmCostVolume<Cost> on it skips its probing read of the volume (2.1 GB at 1080p x 256).  The statement is
    checked against the tensor's storage pointer, shape, strides and version counter, which sees every in-place torch operation on the tensor
    or on a view of it -- and nothing else.  Writes that torch's version counter does not see make sgmCostVolume trust stale minima and
    return a wrong volume without notice: through `cv.data` (a tensor with a counter of its own), through another framework's kernel handed
    `cv.data_ptr()` / a DLPack or __cuda_array_interface__ export, through a raw C-ABI call.  Ask for it only for a volume you do not write
    to by such routes; `dropMinima(cv)` withdraws it.

    Exactly ONE statement is attached: with both flags a Cost-strategy function keeps the minima, a Score-strategy one the winner (what
    the C++ drop-in headers do on DeviceArray); either flag with host (numpy) images raises TypeError.

    keep_winner (device tensors; off by default; the same conditions): keep the index map a later extractSelectedIndex<strategy of
    matchFunc> on the untouched tensor would scan the volume for (svh_unfold_cost_volume_winner: the reference benchmark's own sequence
    is unfoldBasedCostVolume -> extractSelectedIndex): that call then returns a copy of the map."""
    lib = _capi.load()
    l, r = _prep_image(img_l), _prep_image(img_r)
    if (keep_minima or keep_winner) and not _is_torch(l):
        raise TypeError("keep_minima / keep_winner attach a statement to a device tensor; host (numpy) volumes carry none")
    if keep_minima and keep_winner and int(matchFuncStrategy(matchFunc)) != dispExtractionStartegy.Cost:
        keep_minima = False  # exactly ONE statement travels with the volume, chosen by the function's strategy (the C++ headers do the same):
    elif keep_minima and keep_winner:  # Cost-strategy functions keep the minima (for sgmCostVolume), Score-strategy ones the winner
        keep_winner = False
    ctx = context_for(l)
    lower, D = _search_range(disp_width)
    if l.shape[0] != r.shape[0] or (l.ndim == 3 and l.shape[2] != r.shape[2]):
        return _empty_like(l, 3, "f32")
    src = r if int(dDir) == dispDirection.RightToLeft else l
    out = _like(l, (src.shape[0], src.shape[1], D), "f32")
    if keep_minima and _is_torch(out) and int(matchFuncStrategy(matchFunc)) == dispExtractionStartegy.Cost:
        # device volume of integer costs, at the caller's request: keep what a later sgmCostVolume<Cost> on it would otherwise re-read the
        # volume for (svh_unfold_cost_volume_minima), tied to this tensor's storage and version counter
        minima = _like(l, (src.shape[0], src.shape[1], 2), "f32")
        written = C.c_int(0)
        st = _check(ctx, lib.svh_unfold_cost_volume_minima(ctx, int(matchFunc), int(dDir), C.byref(_desc(l)), C.byref(_desc(r)), h_radius, v_radius,
                                                           lower, D, C.byref(_desc(out)), C.byref(_desc(minima)), C.byref(written)))
        if st == _capi.OK and written.value:  # 1: integer costs (census / Hamming), 2: float costs inside the regime (column-sum kernel)
            nWw = ((2 * h_radius + 1) * (2 * v_radius + 1) * (l.shape[2] if l.ndim == 3 else 1) - 1) // 32
            out._svh_minima = (minima, float(32 * nWw) if written.value == 1 else 1e30, out.data_ptr(), out._version, tuple(out.shape), int(written.value))
        return out if st == _capi.OK else _empty_like(l, 3, "f32")
    if keep_winner and _is_torch(out):
        widx = _like(l, (src.shape[0], src.shape[1]), "i32")
        written = C.c_int(0)
        st = _check(ctx, lib.svh_unfold_cost_volume_winner(ctx, int(matchFunc), int(dDir), C.byref(_desc(l)), C.byref(_desc(r)), h_radius, v_radius,
                                                           lower, D, C.byref(_desc(out)), C.byref(_desc(widx)), C.byref(written)))
        if st == _capi.OK and written.value:
            out._svh_winner = (widx, int(matchFuncStrategy(matchFunc)), out.data_ptr(), out._version, tuple(out.shape))
        return out if st == _capi.OK else _empty_like(l, 3, "f32")
    st = _check(ctx, lib.svh_unfold_cost_volume(ctx, int(matchFunc), int(dDir), C.byref(_desc(l)), C.byref(_desc(r)), h_radius, v_radius,
                                                lower, D, C.byref(_desc(out))))
    return out if st == _capi.OK else _empty_like(l, 3, "f32")


def dropMinima(cv):
    """Withdraw the statements unfoldBasedCostVolume(..., keep_minima=True) / sgmCostVolume(..., keep_winner=True) left with `cv` (call it
    after writing to the volume by a route torch's version counter does not see)."""
    if getattr(cv, "_svh_minima", None) is not None:
        cv._svh_minima = None
    if getattr(cv, "_svh_winner", None) is not None:
        cv._svh_winner = None


def _volume_minima(cv):
    """The regional minima unfoldBasedCostVolume(..., keep_minima=True) left with a device volume, if the tensor is still the one it wrote
    (same storage, same shape, version counter untouched: no in-place torch operation since).  See that function for what this cannot see."""
    hint = getattr(cv, "_svh_minima", None) if _is_torch(cv) else None
    if hint is None:
        return None
    minima, max_abs, ptr, version, shape, kind = hint
    if cv.data_ptr() != ptr or cv._version != version or tuple(cv.shape) != shape or not cv.is_contiguous():
        return None
    return minima, max_abs, kind


def sgmCostVolume(nDirections, extractionStrategy, cv_base, P1, P2, margins=None, Pout=100.0, semantics="reference", keep_winner=False):
    """sgmCostVolume<nDirections, strategy>(cv, P1, P2, margins, Pout) -- correlation/sgm.h:360-404, as written.
    semantics="textbook" selects the explicit second mode (every direction fully traversed, neighbour penalties in the Cost
    strategy): NOT the reference's result, see svh_sgm_cost_volume_textbook in include/stevi_hip.h.

    keep_winner (device tensors; off by default): also keep, with the returned tensor, the index map a later
    extractSelectedIndex(strategy, ...) on it returns -- the kernel that writes a pixel's final aggregated costs picks it while it holds
    them (svh_sgm_cost_volume_winner) -- so that call hands back a copy of the map instead of reading the volume again.  Like the minima
    of unfoldBasedCostVolume(keep_minima=True) the statement is checked against the tensor's storage pointer, shape, strides and version
    counter and cannot see writes that bypass torch's counter (`.data`, other frameworks' kernels on data_ptr(), DLPack, raw C-ABI calls):
    dropMinima(out) withdraws it."""
    lib = _capi.load()
    cv = _prep_volume(cv_base)
    ctx = context_for(cv)
    m = (margins or Margins()).as_tuple()
    out = _like(cv, tuple(cv.shape), "f32")
    if semantics not in ("reference", "textbook"):
        raise ValueError("semantics is 'reference' or 'textbook'")
    hint = _volume_minima(cv_base) if (semantics == "reference" and int(extractionStrategy) == dispExtractionStartegy.Cost) else None
    if semantics == "reference" and keep_winner and _is_torch(out):
        widx = _like(cv, tuple(cv.shape[:2]), "i32")
        written = C.c_int(0)
        _check(ctx, lib.svh_sgm_cost_volume_winner(ctx, int(nDirections), int(extractionStrategy), C.byref(_desc(cv)),
                                                   C.byref(_desc(hint[0])) if hint is not None else None, hint[2] if hint is not None else 1,
                                                   C.c_float(hint[1] if hint is not None else 0.0), P1, P2,
                                                   (C.c_int32 * 4)(*m), Pout, C.byref(_desc(out)), C.byref(_desc(widx)), C.byref(written)))
        if written.value:
            out._svh_winner = (widx, int(extractionStrategy), out.data_ptr(), out._version, tuple(out.shape))
        return out
    if hint is not None and hint[2] == 2:  # float costs inside the regime: the minima go in through the winner entry point (no map asked for)
        _check(ctx, lib.svh_sgm_cost_volume_winner(ctx, int(nDirections), int(extractionStrategy), C.byref(_desc(cv)), C.byref(_desc(hint[0])), 2, C.c_float(hint[1]),
                                                   P1, P2, (C.c_int32 * 4)(*m), Pout, C.byref(_desc(out)), None, None))
        return out
    if hint is not None:  # same bits, one read of the volume less (svh_sgm_cost_volume_minima)
        _check(ctx, lib.svh_sgm_cost_volume_minima(ctx, int(nDirections), int(extractionStrategy), C.byref(_desc(cv)), C.byref(_desc(hint[0])),
                                                   C.c_float(hint[1]), P1, P2, (C.c_int32 * 4)(*m), Pout, C.byref(_desc(out))))
        return out
    fn = lib.svh_sgm_cost_volume if semantics == "reference" else lib.svh_sgm_cost_volume_textbook
    _check(ctx, fn(ctx, int(nDirections), int(extractionStrategy), C.byref(_desc(cv)), P1, P2, (C.c_int32 * 4)(*m), Pout, C.byref(_desc(out))))
    return out


def extractSelectedIndex(strategy, costVolume):
    """extractSelectedIndex<strategy> -- correlation/correlation_base.h:427-464."""
    lib = _capi.load()
    hint = getattr(costVolume, "_svh_winner", None) if _is_torch(costVolume) else None
    if hint is not None:  # the volume came from sgmCostVolume(keep_winner=True) and has not been written to since: its winner travels with it
        widx, strat, ptr, version, shape = hint
        if (strat == int(strategy) and costVolume.data_ptr() == ptr and costVolume._version == version and tuple(costVolume.shape) == shape and
                costVolume.is_contiguous()):
            return widx.clone()
    cv = _prep(costVolume, np.float32)
    ctx = context_for(cv)
    out = _like(cv, tuple(cv.shape[:2]), "i32")
    _check(ctx, lib.svh_extract_selected_index(ctx, int(strategy), C.byref(_desc(cv)), C.byref(_desc(out))))
    return out


def selectedIndexToDisp(selectedIndex, disp_offset=0, dDir=dispDirection.RightToLeft):
    """selectedIndexToDisp<disp_t,dDir> -- correlation_base.h:511-532."""
    lib = _capi.load()
    idx = _prep(selectedIndex, np.int32)
    ctx = context_for(idx)
    out = _like(idx, tuple(idx.shape), "i32")
    _check(ctx, lib.svh_selected_index_to_disp(ctx, int(dDir), C.byref(_desc(idx)), int(disp_offset), C.byref(_desc(out))))
    return out


def selectedCost(costVolume, selectedIndex):
    """selectedCost -- correlation_base.h:557-577."""
    lib = _capi.load()
    cv, idx = _prep(costVolume, np.float32), _prep(selectedIndex, np.int32)
    ctx = context_for(cv)
    out = _like(cv, tuple(cv.shape[:2]), "f32")
    _check(ctx, lib.svh_selected_cost(ctx, C.byref(_desc(cv)), C.byref(_desc(idx)), C.byref(_desc(out))))
    return out


def truncatedCostVolume(costVolume, selectedIndex, h_radius, v_radius, cost_vol_radius, dir=dispDirection.RightToLeft,
                        sdir=truncatedCostVolumeDirection.Same):
    """truncatedCostVolume<T_CV,dir,sdir> -- correlation_base.h:579-674."""
    lib = _capi.load()
    cv, idx = _prep(costVolume, np.float32), _prep(selectedIndex, np.int32)
    ctx = context_for(cv)
    T = 4 * cost_vol_radius + 1 if int(sdir) == truncatedCostVolumeDirection.Both else 2 * cost_vol_radius + 1
    out = _like(cv, (cv.shape[0], cv.shape[1], T), "f32")
    _check(ctx, lib.svh_truncated_cost_volume(ctx, int(sdir), int(dir), C.byref(_desc(cv)), C.byref(_desc(idx)), h_radius, v_radius,
                                              cost_vol_radius, C.byref(_desc(out))))
    return out


def refineDispCostInterpolation(kernel, truncatedCostVolume, rawDisparity):
    """refineDispCostInterpolation<kernel> -- correlation/cost_based_refinement.h:128-163."""
    lib = _capi.load()
    tcv, raw = _prep(truncatedCostVolume, np.float32), _prep(rawDisparity, np.int32)
    ctx = context_for(tcv)
    out = _like(tcv, tuple(raw.shape), "f32")
    st = _check(ctx, lib.svh_refine_disp_cost_interpolation(ctx, int(kernel), C.byref(_desc(tcv)), C.byref(_desc(raw)), C.byref(_desc(out))))
    return out if st == _capi.OK else _empty_like(tcv, 2, "f32")


def _stereo_params(matchFunc, h_radius, v_radius, disp_width, dDir, sgmDirections, P1, P2, Pout, margins, refineKernel, refine_h_radius,
                   refine_v_radius, shard):
    lower, D = _search_range(disp_width)
    p = _capi.SvhStereoParams()
    p.match_func, p.disp_direction = int(matchFunc), int(dDir)
    p.h_radius, p.v_radius = int(h_radius), int(v_radius)
    p.disp_lower, p.disp_count = lower, D
    p.sgm_directions = int(sgmDirections)
    p.P1, p.P2, p.Pout = P1, P2, Pout
    for k, v in enumerate((margins or Margins()).as_tuple()):
        p.margins[k] = v
    p.refine_kernel = -1 if refineKernel is None else int(refineKernel)
    p.refine_h_radius, p.refine_v_radius = int(refine_h_radius), int(refine_v_radius)
    if shard is not None:
        p.shard_begin, p.shard_count = int(shard[0]), int(shard[1])
    return p, D


def stereoMatch(matchFunc, img_l, img_r, h_radius, v_radius, disp_width, dDir=dispDirection.RightToLeft, sgmDirections=0, P1=0.001,
                P2=0.01, Pout=100.0, margins=None, refineKernel=None, refine_h_radius=0, refine_v_radius=0, want_cv=False,
                want_sgm_cv=False, want_keys=False, shard=None):
    """Fused device pipeline (svh_stereo_match): unfoldBasedCostVolume -> [sgmCostVolume] -> extractSelectedIndex ->
    selectedIndexToDisp -> [truncatedCostVolume(Same, radius 1) -> refineDispCostInterpolation].
    Returns a dict with 'disp' and, when requested, 'refined', 'cv', 'sgm_cv', 'keys'."""
    lib = _capi.load()
    l, r = _prep_image(img_l), _prep_image(img_r)
    ctx = context_for(l)
    p, D = _stereo_params(matchFunc, h_radius, v_radius, disp_width, dDir, sgmDirections, P1, P2, Pout, margins, refineKernel,
                          refine_h_radius, refine_v_radius, shard)
    Dl = D if shard is None else int(shard[1])
    src = r if int(dDir) == dispDirection.RightToLeft else l
    H, W = src.shape[0], src.shape[1]
    res = {"disp": _like(l, (H, W), "i32")}
    if refineKernel is not None:
        res["refined"] = _like(l, (H, W), "f32")
    if want_cv:
        res["cv"] = _like(l, (H, W, Dl), "f32")
    if want_sgm_cv:
        res["sgm_cv"] = _like(l, (H, W, Dl), "f32")
    if want_keys:
        res["keys"] = _like(l, (H, W), "u64")
    descs = {k: _desc(v) for k, v in res.items()}

    def ref(name):
        return C.byref(descs[name]) if name in descs else None

    st = _check(ctx, lib.svh_stereo_match(ctx, C.byref(p), C.byref(_desc(l)), C.byref(_desc(r)), ref("disp"), ref("refined"), ref("cv"),
                                          ref("sgm_cv"), ref("keys")))
    if st != _capi.OK:
        return {k: _empty_like(l, v.ndim, "f32") for k, v in res.items()}
    return res


def censusShardKeys(img_l, img_r, h_radius, v_radius, disp_width, shard, dDir=dispDirection.RightToLeft, sgmDirections=8, P1=0.001,
                    P2=0.01, Pout=100.0, margins=None, matchFunc=matchingFunctions.CENSUS):
    """svh_census_shard_keys: (H, W, 2) int32 regional winner keys of the disparity shard (begin, count)."""
    lib = _capi.load()
    l, r = _prep_image(img_l), _prep_image(img_r)
    ctx = context_for(l)
    p, _ = _stereo_params(matchFunc, h_radius, v_radius, disp_width, dDir, sgmDirections, P1, P2, Pout, margins, None, 0, 0, shard)
    src = r if int(dDir) == dispDirection.RightToLeft else l
    keys = _like(l, (src.shape[0], src.shape[1], 2), "i32")
    st = _check(ctx, lib.svh_census_shard_keys(ctx, C.byref(p), C.byref(_desc(l)), C.byref(_desc(r)), C.byref(_desc(keys))))
    # row / channel mismatch, single-channel census: the reference returns an empty array (census.h:76-78); never hand
    # uninitialised keys to the all-reduce
    return keys if st == _capi.OK else _empty_like(l, 3, "i32")


def censusBandMatch(img_l, img_r, h_radius, v_radius, disp_width, rows, dDir=dispDirection.RightToLeft, sgmDirections=8, P1=0.001, P2=0.01,
                    Pout=100.0, margins=None, matchFunc=matchingFunctions.CENSUS, out=None):
    """svh_census_band_match: rows (begin, count) of the disparity map of the WHOLE images, (count, W) int32, bit-identical to the
    same rows of stereoMatch(...)["disp"].  Census costs, integer-exact regime, matrix-core sweep geometries."""
    lib = _capi.load()
    l, r = _prep_image(img_l), _prep_image(img_r)
    ctx = context_for(l)
    p, _ = _stereo_params(matchFunc, h_radius, v_radius, disp_width, dDir, sgmDirections, P1, P2, Pout, margins, None, 0, 0, None)
    src = r if int(dDir) == dispDirection.RightToLeft else l
    begin, count = int(rows[0]), int(rows[1])
    band = out if out is not None else _like(l, (count, src.shape[1]), "i32")
    st = _check(ctx, lib.svh_census_band_match(ctx, C.byref(p), C.byref(_desc(l)), C.byref(_desc(r)), begin, count, C.byref(_desc(band))))
    return band if st == _capi.OK else _empty_like(l, 2, "i32")


def censusShardRegion1IsGlobal(img_l, img_r, disp_width, dDir=dispDirection.RightToLeft):
    """svh_census_shard_region1_is_global: True when keys[..., 1] of censusShardKeys is already the winner over all shards (only
    keys[..., 0] needs the MIN reduction).  Reads shapes only."""
    lib = _capi.load()
    l, r = _prep_image(img_l), _prep_image(img_r)
    p, _ = _stereo_params(matchingFunctions.CENSUS, 1, 1, disp_width, dDir, 0, 0.0, 0.0, 0.0, None, None, 0, 0, None)
    return bool(lib.svh_census_shard_region1_is_global(C.byref(p), C.byref(_desc(l)), C.byref(_desc(r))))


def censusShardFinish(img_l, img_r, keys, h_radius, v_radius, disp_width, dDir=dispDirection.RightToLeft, sgmDirections=8, P1=0.001,
                      P2=0.01, Pout=100.0, margins=None, refineKernel=None, refine_h_radius=0, refine_v_radius=0,
                      matchFunc=matchingFunctions.CENSUS):
    """svh_census_shard_finish on keys already MIN-reduced over all shards -> {'disp'[, 'refined']}."""
    lib = _capi.load()
    l, r = _prep_image(img_l), _prep_image(img_r)
    ctx = context_for(l)
    p, _ = _stereo_params(matchFunc, h_radius, v_radius, disp_width, dDir, sgmDirections, P1, P2, Pout, margins, refineKernel,
                          refine_h_radius, refine_v_radius, None)
    src = r if int(dDir) == dispDirection.RightToLeft else l
    H, W = src.shape[0], src.shape[1]
    res = {"disp": _like(l, (H, W), "i32")}
    if refineKernel is not None:
        res["refined"] = _like(l, (H, W), "f32")
    dd = _desc(res["disp"])
    dr = _desc(res["refined"]) if "refined" in res else None
    st = _check(ctx, lib.svh_census_shard_finish(ctx, C.byref(p), C.byref(_desc(l)), C.byref(_desc(r)), C.byref(_desc(keys)), C.byref(dd),
                                                 C.byref(dr) if dr is not None else None))
    if st != _capi.OK:
        return {k: _empty_like(l, 2, "i32" if k == "disp" else "f32") for k in res}
    return res


def keysToIndex(strategy, keys, disp_count):
    lib = _capi.load()
    ctx = context_for(keys)
    out = _like(keys, tuple(keys.shape), "i32")
    _check(ctx, lib.svh_keys_to_index(ctx, int(strategy), C.byref(_desc(keys)), int(disp_count), C.byref(_desc(out))))
    return out


class searchOffset2:  # searchOffset<2>, correlation_base.h:288-409
    def __init__(self, lower0, upper0, lower1, upper1):
        self.lower0, self.upper0, self.lower1, self.upper1 = int(lower0), int(upper0), int(lower1), int(upper1)


def unfoldBased2dDisparityCostVolume(matchFunc, img_l, img_r, h_radius, v_radius, searchWindows, dDir=dispDirection.RightToLeft):
    """unfoldBased2dDisparityCostVolume -- correlation/cross_correlations.h:794-822; (H, W, Dh, Dw) or an empty array."""
    lib = _capi.load()
    l, r = _prep_image(img_l), _prep_image(img_r)
    ctx = context_for(l)
    sw = searchWindows
    Dh, Dw = sw.upper0 - sw.lower0 + 1, sw.upper1 - sw.lower1 + 1
    if l.shape[0] != r.shape[0] or l.shape[1] != r.shape[1] or (l.ndim == 3 and l.shape[2] != r.shape[2]) or Dh <= 0 or Dw <= 0:
        return _empty_like(l, 4, "f32")
    out = _like(l, (l.shape[0], l.shape[1], Dh, Dw), "f32")
    st = _check(ctx, lib.svh_unfold_cost_volume_2d(ctx, int(matchFunc), int(dDir), C.byref(_desc(l)), C.byref(_desc(r)), h_radius, v_radius,
                                                   sw.lower0, sw.upper0, sw.lower1, sw.upper1, C.byref(_desc(out))))
    return out if st == _capi.OK else _empty_like(l, 4, "f32")


def extractSelected2dIndex(strategy, costVolume):
    """extractSelected2dIndex<strategy> -- correlation_base.h:466-509."""
    lib = _capi.load()
    cv = _prep(costVolume, np.float32)
    ctx = context_for(cv)
    out = _like(cv, (cv.shape[0], cv.shape[1], 2), "i32")
    _check(ctx, lib.svh_extract_selected_2d_index(ctx, int(strategy), C.byref(_desc(cv)), C.byref(_desc(out))))
    return out


def selected2dIndexToDisp(selectedIndex, offset):
    """selected2dIndexToDisp(idx, searchOffset<2>) -- correlation_base.h:534-555."""
    lib = _capi.load()
    idx = _prep(selectedIndex, np.int32)
    ctx = context_for(idx)
    out = _like(idx, tuple(idx.shape), "i32")
    _check(ctx, lib.svh_selected_2d_index_to_disp(ctx, C.byref(_desc(idx)), offset.lower0, offset.lower1, C.byref(_desc(out))))
    return out


def truncatedBidirectionaCostVolume(costVolume, selectedIndex, cost_vol_radius0, cost_vol_radius1):
    """truncatedBidirectionaCostVolume -- correlation_base.h:677-725 (explicit radii)."""
    lib = _capi.load()
    cv, idx = _prep(costVolume, np.float32), _prep(selectedIndex, np.int32)
    ctx = context_for(cv)
    out = _like(cv, (cv.shape[0], cv.shape[1], 2 * cost_vol_radius0 + 1, 2 * cost_vol_radius1 + 1), "f32")
    _check(ctx, lib.svh_truncated_bidirectional_cost_volume(ctx, C.byref(_desc(cv)), C.byref(_desc(idx)), cost_vol_radius0, cost_vol_radius1,
                                                            C.byref(_desc(out))))
    return out


def refineDisp2dCostInterpolation(kernel, truncatedCostVolume, rawDisparity, isotropHypothesis=IsotropyHypothesis.Isotropic):
    """refineDisp2dCostInterpolation<kernel, isotropHypothesis> -- correlation/cost_based_refinement.h:165-376."""
    lib = _capi.load()
    tcv, raw = _prep(truncatedCostVolume, np.float32), _prep(rawDisparity, np.int32)
    ctx = context_for(tcv)
    out = _like(tcv, tuple(raw.shape), "f32")
    st = _check(ctx, lib.svh_refine_disp_2d_cost_interpolation(ctx, int(kernel), int(isotropHypothesis), C.byref(_desc(tcv)), C.byref(_desc(raw)),
                                                               C.byref(_desc(out))))
    return out if st == _capi.OK else _empty_like(tcv, 3, "f32")


def refineDisp2dCostPatchInterpolation(kernel, truncatedCostVolume, rawDisparity):
    """refineDisp2dCostPatchInterpolation<kernel> -- correlation/cost_based_refinement.h:378-436 (Parabola / Gaussian)."""
    lib = _capi.load()
    tcv, raw = _prep(truncatedCostVolume, np.float32), _prep(rawDisparity, np.int32)
    ctx = context_for(tcv)
    out = _like(tcv, tuple(raw.shape), "f32")
    st = _check(ctx, lib.svh_refine_disp_2d_cost_patch_interpolation(ctx, int(kernel), C.byref(_desc(tcv)), C.byref(_desc(raw)), C.byref(_desc(out))))
    return out if st == _capi.OK else _empty_like(tcv, 3, "f32")


# ---- UnFoldCompressor features ---------------------------------------------------------
class UnFoldCompressor:
    """UnFoldCompressor(mask) -- correlation/unfold.h:36-137.  mask: 2-D int array, positive labels = superpixels of the window."""

    class pixelIndex:
        def __init__(self, verticalShift, horizontalShift, featureIndex, weight):
            self.verticalShift, self.horizontalShift, self.featureIndex, self.weight = verticalShift, horizontalShift, featureIndex, weight

    def __init__(self, mask):
        self.mask = np.ascontiguousarray(np.asarray(mask), dtype=np.int32)
        if self.mask.ndim != 2:
            raise ValueError("the mask is a 2-D array of labels")
        mh, mw = self.mask.shape
        ii, jj = np.nonzero(self.mask > 0)
        dv, dh = ii - mh // 2, jj - mw // 2
        self._minH, self._maxH = min(0, int(dv.min(initial=0))), max(0, int(dv.max(initial=0)))  # the box always holds the centre (:57-60)
        self._minW, self._maxW = min(0, int(dh.min(initial=0))), max(0, int(dh.max(initial=0)))
        labels, counts = np.unique(self.mask[self.mask > 0], return_counts=True)
        self._nFeatures = int(labels.size)
        self._indices = []
        for f, (lab, cnt) in enumerate(zip(labels, counts)):  # features in increasing label order, entries row-major (:103-120)
            for i, j in zip(*np.nonzero(self.mask == lab)):
                self._indices.append(UnFoldCompressor.pixelIndex(int(i) - mh // 2, int(j) - mw // 2, f, float(np.float32(1.0 / cnt))))

    def nFeatures(self): return self._nFeatures
    def width(self): return self._maxW - self._minW + 1
    def height(self): return self._maxH - self._minH + 1
    def margins(self): return PaddingMargins(-self._minW, -self._minH, self._maxW, self._maxH)
    def indices(self): return list(self._indices)


class CompressorGenerators:
    """correlation/unfold.h:475-693: the two 17-superpixel masks the reference ships (radius 3 and 4 windows)."""

    @staticmethod
    def GrPix17R3Filter():
        return np.array([[14, 14, 10, 10, 10, 16, 16],
                         [14, 14, 6, 4, 7, 16, 16],
                         [11, 6, 6, 4, 7, 7, 13],
                         [11, 2, 2, 1, 3, 3, 13],
                         [11, 8, 8, 5, 9, 9, 13],
                         [15, 15, 8, 5, 9, 17, 17],
                         [15, 15, 12, 12, 12, 17, 17]], np.int32)

    @staticmethod
    def GrPix17R4Filter():
        return np.array([[14, 14, 14, 10, 10, 10, 16, 16, 16],
                         [14, 14, 14, 10, 10, 10, 16, 16, 16],
                         [14, 14, 6, 6, 4, 7, 7, 16, 16],
                         [11, 11, 6, 6, 4, 7, 7, 13, 13],
                         [11, 11, 2, 2, 1, 3, 3, 13, 13],
                         [11, 11, 8, 8, 5, 9, 9, 13, 13],
                         [15, 15, 8, 8, 5, 9, 9, 17, 17],
                         [15, 15, 15, 12, 12, 12, 17, 17, 17],
                         [15, 15, 15, 12, 12, 12, 17, 17, 17]], np.int32)


def unfoldCompressed(compressor, in_data, padding=None):
    """unfold(compressor, in_data, padding) -- correlation/unfold.h:346-471."""
    lib = _capi.load()
    x = _prep_image(in_data)
    ctx = context_for(x)
    d = _desc(x)
    m = compressor.mask
    mp = m.ctypes.data_as(C.POINTER(C.c_int32))
    shp = (C.c_int64 * 3)()
    if lib.svh_unfold_compressed_shape(C.byref(d), mp, m.shape[0], m.shape[1], _pad_arg(padding), shp) != _capi.OK:
        raise ValueError("bad unfold arguments")
    if shp[0] <= 0 or shp[1] <= 0 or shp[2] <= 0:
        return _empty_like(x, 3, "f32")
    out = _like(x, (shp[0], shp[1], shp[2]), "f32")
    _check(ctx, lib.svh_unfold_compressed(ctx, C.byref(d), mp, m.shape[0], m.shape[1], _pad_arg(padding), C.byref(_desc(out))))
    return out


def unfoldBasedCostVolumeCompressed(matchFunc, img_l, img_r, compressor, disp_width, dDir=dispDirection.RightToLeft):
    """unfoldBasedCostVolume(img_l, img_r, compressor, disp_width) / unfoldBased2dDisparityCostVolume(img_l, img_r, compressor,
    searchOffset<2>) -- correlation/cross_correlations.h:767-791, :824-851: compressed unfold, then featureVolume2CostVolume."""
    l, r = _prep_image(img_l), _prep_image(img_r)
    two_d = isinstance(disp_width, searchOffset2)
    if l.shape[0] != r.shape[0] or (two_d and l.shape[1] != r.shape[1]) or (l.ndim == 3 and l.shape[2] != r.shape[2]):
        return _empty_like(l, 4 if two_d else 3, "f32")
    return featureVolume2CostVolume(matchFunc, unfoldCompressed(compressor, l), unfoldCompressed(compressor, r), disp_width, dDir)


# ---- A7 / A8: per-pixel statistics and feature-volume transforms as stand-alone functions -------------------------------
def _map_call(fn, feat, *maps):
    lib = _capi.load()
    f = _prep(feat, np.float32)
    ms = [None if m is None else _prep(m, np.float32) for m in maps]
    ctx = context_for(f)
    out = _like(f, (f.shape[0], f.shape[1]), "f32")
    _check(ctx, getattr(lib, fn)(ctx, C.byref(_desc(f)), *[None if m is None else C.byref(_desc(m)) for m in ms], C.byref(_desc(out))))
    return out


def _volume_call(fn, feat, *maps):
    lib = _capi.load()
    f = _prep(feat, np.float32)
    ms = [_prep(m, np.float32) for m in maps]
    ctx = context_for(f)
    out = _like(f, tuple(f.shape), "f32")
    _check(ctx, getattr(lib, fn)(ctx, C.byref(_desc(f)), *[C.byref(_desc(m)) for m in ms], C.byref(_desc(out))))
    return out


def channelsMean(in_data):
    """channelsMean -- correlation/correlation_base.h:1100-1136."""
    return _map_call("svh_channels_mean", in_data)


def channelsNorm(in_data):
    """channelsNorm -- correlation/cross_correlations.h:149-191."""
    return _map_call("svh_channels_norm", in_data)


def channelsZeroMeanNorm(in_data, mean=None):
    """channelsZeroMeanNorm(in_data[, mean]) -- correlation/cross_correlations.h:61-122."""
    return _map_call("svh_channels_zero_mean_norm", in_data, mean)


def zeromeanFeatureVolume(feature_vol, mean):
    """zeromeanFeatureVolume -- correlation/cross_correlations.h:570-594."""
    return _volume_call("svh_zeromean_feature_volume", feature_vol, mean)


def normalizedFeatureVolume(feature_vol, norm):
    """normalizedFeatureVolume -- correlation/cross_correlations.h:504-550."""
    return _volume_call("svh_normalized_feature_volume", feature_vol, norm)


def zeromeanNormalizedFeatureVolume(feature_vol, mean, norm):
    """zeromeanNormalizedFeatureVolume -- correlation/cross_correlations.h:416-462."""
    return _volume_call("svh_zeromean_normalized_feature_volume", feature_vol, mean, norm)


def getFeatureVolumeForMatchFunc(matchFunc, feature_vol):
    """getFeatureVolumeForMatchFunc<matchFunc> -- correlation/cross_correlations.h:645-722: float32 volume, or uint32 census words."""
    lib = _capi.load()
    f = _prep(feature_vol, np.float32)
    ctx = context_for(f)
    census = int(matchFunc) in (matchingFunctions.CENSUS, matchingFunctions.HAMMING)
    H, W, F = f.shape
    if census and F <= 1:
        return _empty_like(f, 3, "u32")
    out = _like(f, (H, W, (F - 1) // 32 + 1), "u32") if census else _like(f, (H, W, F), "f32")
    st = _check(ctx, lib.svh_feature_volume_for_match_func(ctx, int(matchFunc), C.byref(_desc(f)), C.byref(_desc(out))))
    return out if st == _capi.OK else _empty_like(f, 3, "u32" if census else "f32")


# ---- hierarchical matching ----------------------------------------------------------
def averagePoolingDownsample(input, windows):
    """Interpolation::averagePoolingDownsample(input, DownSampleWindows) -- interpolation/downsampling.h:67-178.
    windows: int (square) or (horizontal, vertical)."""
    lib = _capi.load()
    x = _prep(input, np.float32)
    ctx = context_for(x)
    wh, wv = (int(windows), int(windows)) if np.isscalar(windows) else (int(windows[0]), int(windows[1]))
    shp = ((x.shape[0] + wv - 1) // wv, (x.shape[1] + wh - 1) // wh) + tuple(x.shape[2:])
    out = _like(x, shp, "f32")
    _check(ctx, lib.svh_average_pooling_downsample(ctx, C.byref(_desc(x)), wh, wv, C.byref(_desc(out))))
    return out


class OffsetedCostVolume:  # correlation/hierarchical.h:33-37
    def __init__(self, truncated_cost_volume, disp_estimate):
        self.truncated_cost_volume = truncated_cost_volume
        self.disp_estimate = disp_estimate


def computeGuidedCV(matchFunc, feature_vol_l, feature_vol_r, disp_guide, upscale_disp_radius, dDir=dispDirection.RightToLeft):
    """computeGuidedCV<matchFunc, ..., dDir> -- correlation/hierarchical.h:74-229, on the feature volumes
    getFeatureVolumeForMatchFunc returns (float32, or uint32 census words)."""
    lib = _capi.load()
    ft = np.uint32 if int(matchFunc) in (matchingFunctions.CENSUS, matchingFunctions.HAMMING) else np.float32
    l, r, g = _prep(feature_vol_l, ft), _prep(feature_vol_r, ft), _prep(disp_guide, np.int32)
    ctx = context_for(l)
    src = r if int(dDir) == dispDirection.RightToLeft else l
    T = 2 * int(upscale_disp_radius) + 1
    tcv, disp = _like(l, (src.shape[0], src.shape[1], T), "f32"), _like(l, (src.shape[0], src.shape[1]), "i32")
    st = _check(ctx, lib.svh_guided_cost_volume(ctx, int(matchFunc), int(dDir), C.byref(_desc(l)), C.byref(_desc(r)), C.byref(_desc(g)),
                                                int(upscale_disp_radius), C.byref(_desc(tcv)), C.byref(_desc(disp))))
    if st != _capi.OK:
        return OffsetedCostVolume(_empty_like(l, 3, "f32"), _empty_like(l, 2, "i32"))
    return OffsetedCostVolume(tcv, disp)


def hiearchicalTruncatedCostVolume(matchFunc, depth, img_l, img_r, h_radiuses, v_radiuses, disp_width, upscale_disp_radius=2,
                                   dDir=dispDirection.RightToLeft):
    """hiearchicalTruncatedCostVolume<matchFunc, depth, ..., dDir> -- correlation/hierarchical.h:232-319.  h_radiuses /
    v_radiuses: one radius for every level or depth + 1 radii, coarsest level first."""
    lib = _capi.load()
    l, r = _prep(img_l, np.float32), _prep(img_r, np.float32)
    ctx = context_for(l)
    hr = [int(h_radiuses)] * (depth + 1) if np.isscalar(h_radiuses) else [int(v) for v in h_radiuses]
    vr = [int(v_radiuses)] * (depth + 1) if np.isscalar(v_radiuses) else [int(v) for v in v_radiuses]
    if len(hr) != depth + 1 or len(vr) != depth + 1:
        raise ValueError("radii lists must hold depth + 1 entries")
    src = r if int(dDir) == dispDirection.RightToLeft else l
    T = 2 * int(upscale_disp_radius) + 1
    tcv, disp = _like(l, (src.shape[0], src.shape[1], T), "f32"), _like(l, (src.shape[0], src.shape[1]), "i32")
    st = _check(ctx, lib.svh_hierarchical_truncated_cost_volume(ctx, int(matchFunc), int(dDir), int(depth), C.byref(_desc(l)), C.byref(_desc(r)),
                                                                (C.c_int32 * len(hr))(*hr), (C.c_int32 * len(vr))(*vr), int(disp_width),
                                                                int(upscale_disp_radius), C.byref(_desc(tcv)), C.byref(_desc(disp))))
    if st != _capi.OK:
        return OffsetedCostVolume(_empty_like(l, 3, "f32"), _empty_like(l, 2, "i32"))
    return OffsetedCostVolume(tcv, disp)


# ---- on-demand (cacheless) cost volumes and PatchMatch (what examples/stereo-match runs) --------
def _on_demand_params(matchFunc, radius, searchRange):
    p = _capi.SvhOnDemandParams()
    p.match_func = int(matchFunc)
    p.h_radius, p.v_radius = (int(radius), int(radius)) if np.isscalar(radius) else (int(radius[0]), int(radius[1]))
    if isinstance(searchRange, searchOffset2):
        p.search_dims = 2
        p.lower0, p.upper0, p.lower1, p.upper1 = searchRange.lower0, searchRange.upper0, searchRange.lower1, searchRange.upper1
    else:
        lo, hi = (searchRange.lower, searchRange.upper) if isinstance(searchRange, searchOffset1) else (int(searchRange[0]), int(searchRange[1]))
        p.search_dims = 1
        p.lower0 = p.upper0 = 0
        p.lower1, p.upper1 = lo, hi
    return p


def onDemandFeatures(matchFunc, img, radius):
    """getFeatureVec of OnDemandDecoratedFeaturesVolume<ZNFeaturesVolumeDecorator<ZeroMean, Normalized>, ...> at every pixel, over
    the full (2r+1)^2 x C window of examples/stereo-match/main.cpp:150-164 -- correlation/on_demand_features_volume.h:34-214."""
    lib = _capi.load()
    x = _prep(img, np.float32)
    ctx = context_for(x)
    hr, vr = (int(radius), int(radius)) if np.isscalar(radius) else (int(radius[0]), int(radius[1]))
    Cc = x.shape[2] if x.ndim == 3 else 1
    out = _like(x, (x.shape[0], x.shape[1], (2 * hr + 1) * (2 * vr + 1) * Cc), "f32")
    _check(ctx, lib.svh_on_demand_features(ctx, int(matchFunc), C.byref(_desc(x)), hr, vr, C.byref(_desc(out))))
    return out


def cachelessPatchMatch(matchFunc, img_source, img_target, radius, searchOffset, nIter=5, nRandomSearch=4, seed=0, return_iterations=False, initial=None):
    """cachelessPatchMatch<matchFunc, searchSpaceDim>(OnDemand features of img_source / img_target, searchOffset, nIter, nRandomSearch, initializer)
    -- correlation/patchmatch.h:560-621.  searchOffset: searchOffset2 (flow, disp (H,W,2)) or searchOffset1 / (lower, upper)
    (stereo, disp (H,W,1)).  The random stream is a function of `seed` (the reference's is not reproducible).
    initial: an (H, W, search_dims) int32 array in the images' memory space, used instead of the random draw (what the reference's
    `initializer` callback returns, :598-605)."""
    lib = _capi.load()
    s, t = _prep(img_source, np.float32), _prep(img_target, np.float32)
    ctx = context_for(s)
    p = _on_demand_params(matchFunc, radius, searchOffset)
    out = _like(s, (s.shape[0], s.shape[1], p.search_dims), "i32")
    its = C.c_int32(0)
    ini = None if initial is None else _prep(initial, np.int32)
    st = _check(ctx, lib.svh_cacheless_patch_match_init(ctx, C.byref(p), C.byref(_desc(s)), C.byref(_desc(t)), int(nIter), int(nRandomSearch), C.c_uint64(seed),
                                                        None if ini is None else C.byref(_desc(ini)), C.byref(_desc(out)), C.byref(its)))
    if st != _capi.OK:
        out = _empty_like(s, 3, "i32")
    return (out, its.value) if return_iterations else out


def patchMatch(matchFunc, feature_vol_s, feature_vol_t, searchOffset, nIter=5, nRandomSearch=4, seed=0, return_iterations=False, initial=None):
    """patchMatch<matchFunc, searchSpaceDim>(feature_vol_s, feature_vol_t, searchOffset, nIter, nRandomSearch, initializer, randcache) --
    correlation/patchmatch.h:496-558: PatchMatch on feature volumes (H, W, F) the caller built (benchmarkStereoMatchingModels.cpp:187-199
    passes unfolded images), the reference's cached cost volume behind it (the cache changes no value).  Zero-mean / normalised functions
    process the vectors twice, as the reference does.  Float matching functions.  `initial` as in cachelessPatchMatch; the reference's
    `randcache` has no counterpart (the random stream is a function of `seed`)."""
    lib = _capi.load()
    s, t = _prep(feature_vol_s, np.float32), _prep(feature_vol_t, np.float32)
    ctx = context_for(s)
    p = _on_demand_params(matchFunc, 0, searchOffset)
    out = _like(s, (s.shape[0], s.shape[1], p.search_dims), "i32")
    its = C.c_int32(0)
    ini = None if initial is None else _prep(initial, np.int32)
    st = _check(ctx, lib.svh_patch_match(ctx, C.byref(p), C.byref(_desc(s)), C.byref(_desc(t)), int(nIter), int(nRandomSearch), C.c_uint64(seed),
                                         None if ini is None else C.byref(_desc(ini)), C.byref(_desc(out)), C.byref(its)))
    if st != _capi.OK:
        out = _empty_like(s, 3, "i32")
    return (out, its.value) if return_iterations else out

def onDemandTruncatedCostVolume(matchFunc, img_source, img_target, radius, searchOffset, disp, cv_radius=1):
    """CachelessOnDemand{ImageFlow,Stereo}CostVolume(features_source, features_target, searchSpace).truncatedCostVolume(disp, radius)
    -- correlation/on_demand_cost_volume.h:474-596, as written (window centred on disparity - lowerOffset)."""
    lib = _capi.load()
    s, t, d = _prep(img_source, np.float32), _prep(img_target, np.float32), _prep(disp, np.int32)
    ctx = context_for(s)
    p = _on_demand_params(matchFunc, radius, searchOffset)
    T = 2 * int(cv_radius) + 1
    out = _like(s, (s.shape[0], s.shape[1]) + (T,) * p.search_dims, "f32")
    st = _check(ctx, lib.svh_on_demand_truncated_cost_volume(ctx, C.byref(p), C.byref(_desc(s)), C.byref(_desc(t)), C.byref(_desc(d)), int(cv_radius),
                                                             C.byref(_desc(out))))
    return out if st == _capi.OK else _empty_like(s, 2 + p.search_dims, "f32")


def set_option(x, name, value):
    """svh_context_set_option on the context used for array x: "census_float_overflow", "census_winner_shortcut", "census_sweep",
    "sgm_score_fused", "literal_cost_volumes" (include/stevi_hip.h)."""
    ctx = context_for(x)
    _check(ctx, _capi.load().svh_context_set_option(ctx, name.encode(), int(value)))


def set_test_option(x, name, value):
    """svh_test_set_option (include/stevi_hip_test.h): the A/B switches the parity tests cross-check -- no part of the product surface."""
    ctx = context_for(x)
    _check(ctx, _capi.load().svh_test_set_option(ctx, name.encode(), int(value)))


# ------------------------------------------------------------------------------------------------ profiling
def profile_enable(x, on=True, only=None, every=1):
    """hipEvents around every kernel launch (only=None) or around launches of the kernel named `only`; every=n brackets only
    every n-th of them (an event pair costs about 5 us of stream time)."""
    lib = _capi.load()
    lib.svh_profile_filter(context_for(x), only.encode() if only else None)
    lib.svh_profile_sampling(context_for(x), int(every))
    lib.svh_profile_enable(context_for(x), 1 if on else 0)


def profile_reset(x):
    _capi.load().svh_profile_reset(context_for(x))


def profile_collect(x):
    """-> {kernel name: (total_ms, launches)} accumulated since the last reset."""
    lib = _capi.load()
    ctx = context_for(x)
    _check(ctx, lib.svh_profile_collect(ctx))
    out = {}
    for k in range(lib.svh_profile_count(ctx)):
        name = C.create_string_buffer(64)
        ms, n = C.c_double(), C.c_int64()
        lib.svh_profile_get(ctx, k, name, 64, C.byref(ms), C.byref(n))
        out[name.value.decode()] = (ms.value, n.value)
    return out
