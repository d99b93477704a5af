"""`.stevimg` array files: the reference's raw array format (io/image_io.h:48-168), the data format either side of
the correlation/ path -- image pairs in, disparity maps / cost volumes / fixtures out.

    "<dtype> <nDim> <shape...> <strides...>\\n"       one text line; dtype = f32, f64, u8, u16, u32, i32, ...
    raw elements exactly as they lie in memory under those (element) strides

Arrays keep their strides through a round trip, so a cost volume stored with the reference's {W*D, 1, W} layout
(cross_correlations.h:220) comes back as a numpy view with that layout.

Also the Middlebury `.flo` optical-flow files the reference reads as ground truth for 2-D disparities (io/read_flo.h:13-49).
"""
import numpy as np

_KINDS = {"f": "f", "u": "u", "i": "i"}


def dtype_descr(dtype):
    """'f32', 'u8', 'i32', ... (utils/types_manipulations.h:82-102); '' for anything but plain numbers."""
    dt = np.dtype(dtype)
    if dt.kind not in _KINDS:
        return ""
    return f"{dt.kind}{dt.itemsize * 8}"


def _dtype_from_descr(descr):
    if len(descr) < 2 or descr[0] not in _KINDS or not descr[1:].isdigit():
        raise ValueError(f"not a .stevimg element type: {descr!r}")
    bits = int(descr[1:])
    if bits % 8:
        raise ValueError(f"not a .stevimg element type: {descr!r}")
    return np.dtype(f"{descr[0]}{bits // 8}")


def _fills_its_span(arr):
    """True when the elements occupy arr.size consecutive slots (any axis order), i.e. the memory can be written as is."""
    expected = arr.itemsize
    for k in sorted(range(arr.ndim), key=lambda k: arr.strides[k]):
        if arr.shape[k] == 1:
            continue
        if arr.strides[k] != expected:
            return False
        expected *= arr.shape[k]
    return True


def write_stevimg(path, arr, dtype=None):
    """Write `arr` (converted to `dtype` when given).  Memory that fills its span is written with its own strides;
    anything else (slices with holes, negative or zero strides) as a dense last-index-fastest copy, like the reference
    (:66-69)."""
    arr = np.asarray(arr)
    if dtype is not None and np.dtype(dtype) != arr.dtype:
        arr = arr.astype(dtype)
    descr = dtype_descr(arr.dtype)
    if not descr:
        raise TypeError(f"{arr.dtype} has no .stevimg element type")
    if arr.size and not _fills_its_span(arr):
        arr = np.ascontiguousarray(arr)
    strides = [1] * arr.ndim
    if arr.size:
        strides = [s // arr.itemsize for s in arr.strides]
    else:  # no memory behind it: dense strides of the shape
        for k in range(arr.ndim - 2, -1, -1):
            strides[k] = strides[k + 1] * max(arr.shape[k + 1], 1)
    head = " ".join([descr, str(arr.ndim)] + [str(int(s)) for s in arr.shape] + [str(int(s)) for s in strides]) + "\n"
    with open(path, "wb") as f:
        f.write(head.encode("ascii"))
        if arr.size:
            # the block in memory order: walk the axes from the slowest stride to the fastest
            order = sorted(range(arr.ndim), key=lambda k: -arr.strides[k])
            f.write(np.ascontiguousarray(arr.transpose(order)).tobytes())


def read_stevimg(path, dtype=None, ndim=None):
    """Read an array.  `dtype` / `ndim` play the role of the reference's template arguments (:110): a different element
    type or a file of higher rank gives None (the reference returns an empty array), a file of lower rank gains trailing
    axes of extent 1."""
    with open(path, "rb") as f:
        head = f.readline().decode("ascii").split()
        if len(head) < 2:
            return None
        file_dtype = _dtype_from_descr(head[0])
        n = int(head[1])
        if len(head) != 2 + 2 * n:
            raise ValueError(f"{path}: header announces {n} dimensions but holds {len(head) - 2} numbers")
        if dtype is not None and np.dtype(dtype) != file_dtype:
            return None
        if ndim is not None and n > ndim:
            return None
        shape = [int(v) for v in head[2:2 + n]]
        strides = [int(v) for v in head[2 + n:2 + 2 * n]]
        count = int(np.prod(shape)) if n else 1
        blob = f.read(count * file_dtype.itemsize)
    if len(blob) != count * file_dtype.itemsize:
        return None  # truncated file
    raw = np.frombuffer(blob, dtype=file_dtype)
    if ndim is not None:
        shape += [1] * (ndim - n)
        strides += [1] * (ndim - n)
    if count == 0:
        return np.zeros(shape, dtype=file_dtype)
    span = 1 + sum((s - 1) * st for s, st in zip(shape, strides))
    if span != count or any(st <= 0 for s, st in zip(shape, strides) if s > 1):
        raise ValueError(f"{path}: strides {strides} do not describe a dense block of shape {shape}")
    return np.lib.stride_tricks.as_strided(raw.copy(), shape=shape, strides=[st * file_dtype.itemsize for st in strides])


def read_flo(path, dtype=np.float32):
    """Middlebury .flo -> (H, W, 2) array of (u, v) pairs converted to `dtype`; None where the reference returns an empty
    array (missing file, wrong magic, non-positive size, truncated data)."""
    try:
        with open(path, "rb") as f:
            if f.read(4) != b"PIEH":
                return None
            blob = f.read(8)
            size = np.frombuffer(blob, dtype="<i4") if len(blob) == 8 else np.zeros(2, "<i4")
            if size[0] <= 0 or size[1] <= 0:
                return None
            w, h = int(size[0]), int(size[1])
            blob = f.read(8 * w * h)
    except OSError:
        return None
    if len(blob) != 8 * w * h:
        return None
    return np.frombuffer(blob, dtype="<f4").reshape(h, w, 2).astype(dtype)


def write_flo(path, flow):
    """(H, W, 2) -> Middlebury .flo (the reference only reads the format; this writes test inputs and results)."""
    flow = np.asarray(flow)
    if flow.ndim != 3 or flow.shape[2] != 2 or flow.shape[0] <= 0 or flow.shape[1] <= 0:
        raise ValueError("a flow field is a non-empty (H, W, 2) array")
    with open(path, "wb") as f:
        f.write(b"PIEH")
        f.write(np.array([flow.shape[1], flow.shape[0]], dtype="<i4").tobytes())
        f.write(np.ascontiguousarray(flow, dtype="<f4").tobytes())
