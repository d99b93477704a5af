"""The .stevimg array format (reference io/image_io.h:48-168) through the Python reader / writer and the drop-in C++
header, and across the two.  Host only."""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from libstevi_amd.stevimg import dtype_descr, read_flo, read_stevimg, write_flo, write_stevimg  # noqa: E402


def test_header_line_is_the_reference_layout(tmp_path):
    a = np.arange(24, dtype=np.float32).reshape(2, 3, 4)
    p = tmp_path / "a.stevimg"
    write_stevimg(p, a)
    raw = p.read_bytes()
    head, body = raw.split(b"\n", 1)
    assert head == b"f32 3 2 3 4 12 4 1"          # dtype, nDim, shape, element strides
    assert body == a.tobytes()
    assert [dtype_descr(t) for t in (np.uint8, np.uint16, np.int32, np.uint32, np.float32, np.float64, np.int8)] == [
        "u8", "u16", "i32", "u32", "f32", "f64", "i8"]
    assert dtype_descr(np.complex64) == ""


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.int32, np.uint32, np.float32, np.float64])
@pytest.mark.parametrize("shape", [(7,), (5, 9), (4, 6, 3), (0, 5), (3, 1, 4)])
def test_round_trip(tmp_path, dtype, shape):
    rng = np.random.default_rng(3)
    a = (rng.random(shape) * 200).astype(dtype)
    p = tmp_path / "a.stevimg"
    write_stevimg(p, a)
    b = read_stevimg(p)
    assert b.dtype == a.dtype and b.shape == a.shape and np.array_equal(a, b)
    assert read_stevimg(p, dtype=dtype, ndim=len(shape)).shape == a.shape
    other = np.float32 if dtype != np.float32 else np.int32
    assert read_stevimg(p, dtype=other) is None                      # element type mismatch: the reference's empty array
    if len(shape) > 1:
        assert read_stevimg(p, ndim=len(shape) - 1) is None          # file of higher rank than asked for
    assert read_stevimg(p, ndim=len(shape) + 1).shape == tuple(shape) + (1,)  # lower rank: trailing axis of extent 1


def test_strided_layouts_survive(tmp_path):
    H, W, D = 3, 5, 4
    dense = np.arange(H * W * D, dtype=np.float32).reshape(H, D, W)
    cv = dense.transpose(0, 2, 1)                                    # (H, W, D) with strides {W*D, 1, W}: cross_correlations.h:220
    p = tmp_path / "cv.stevimg"
    write_stevimg(p, cv)
    assert p.read_bytes().split(b"\n", 1)[0] == f"f32 3 {H} {W} {D} {W * D} 1 {W}".encode()
    back = read_stevimg(p)
    assert np.array_equal(back, cv) and back.strides == cv.strides
    # holes (a strided slice) and reversed axes are written as a dense copy
    for view in (dense[:, ::2, :], dense[::-1], np.broadcast_to(np.float32(2.0), (3, 4))):
        write_stevimg(p, view)
        back = read_stevimg(p)
        assert np.array_equal(back, view) and back.flags["C_CONTIGUOUS"]
    # conversion on write
    write_stevimg(p, np.array([[1.75, 2.25]]), dtype=np.uint16)
    assert read_stevimg(p).tolist() == [[1, 2]]


def test_malformed_files(tmp_path):
    p = tmp_path / "bad.stevimg"
    p.write_bytes(b"f32 2 3 3 3 1\n" + np.zeros(4, np.float32).tobytes())
    assert read_stevimg(p) is None                                   # truncated data
    p.write_bytes(b"f32 2 3 3\n")
    with pytest.raises(ValueError):
        read_stevimg(p)
    p.write_bytes(b"q32 1 1 1\n\0\0\0\0")
    with pytest.raises(ValueError):
        read_stevimg(p)
    p.write_bytes(b"f32 2 2 2 4 1\n" + np.zeros(4, np.float32).tobytes())
    with pytest.raises(ValueError):
        read_stevimg(p)                                              # strides with holes cannot come from the writer
    with pytest.raises(TypeError):
        write_stevimg(p, np.zeros(3, np.complex64))


def test_cpp_header_and_python_agree(tmp_path):
    """tests/cpp/stevimg_io.cpp: the reference's testImageIO round trips through the drop-in io/image_io.h, and files
    exchanged with the Python module in both directions."""
    rng = np.random.default_rng(11)
    img = rng.random((5, 7)).astype(np.float32)
    write_stevimg(tmp_path / "from_python_f32.stevimg", img)
    H, W, D = 3, 5, 4
    i, j, d = np.meshgrid(np.arange(H), np.arange(W), np.arange(D), indexing="ij")
    cv = np.ascontiguousarray((100 * i + 10 * j + d).astype(np.float32).transpose(0, 2, 1)).transpose(0, 2, 1)
    write_stevimg(tmp_path / "from_python_cv.stevimg", cv)
    i2, j2 = np.meshgrid(np.arange(4), np.arange(6), indexing="ij")
    flow = np.stack([0.5 * j2 - i2, 0.25 * i2 + j2], axis=2).astype(np.float32)
    write_flo(tmp_path / "from_python.flo", flow)
    raw = (tmp_path / "from_python.flo").read_bytes()
    (tmp_path / "bad_magic.flo").write_bytes(b"HEIP" + raw[4:])
    (tmp_path / "truncated.flo").write_bytes(raw[:-4])
    exe = tmp_path / "stevimg_io"
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "libstevi_amd", "include"),
                    os.path.join(HERE, "cpp", "stevimg_io.cpp"), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe), str(tmp_path)], check=True, capture_output=True, text=True).stdout
    weights = np.outer(np.arange(1, 6), np.arange(2, 9)).astype(np.float64)
    assert abs(float(out.split("weighted_sum=")[1]) - float((img.astype(np.float64) * weights).sum())) < 1e-4
    u16 = read_stevimg(tmp_path / "from_cpp_u16.stevimg")
    assert u16.dtype == np.uint16 and u16.shape == (4, 6) and np.array_equal(u16, np.arange(24).reshape(4, 6))
    back = read_stevimg(tmp_path / "from_cpp_cv.stevimg")
    assert back.strides == cv.strides and np.array_equal(back, cv)
    holes = read_stevimg(tmp_path / "holes.stevimg")                 # every second element of the volume's memory rows
    expected = np.array([[100 * a + 10 * ((2 * b) % W) + (2 * b) // W for b in range(W)] for a in range(H)], dtype=np.float32)
    assert holes.flags["C_CONTIGUOUS"] and np.array_equal(holes, expected)


def test_flo_round_trip(tmp_path):
    """Middlebury .flo (reference io/read_flo.h:13-49): "PIEH", width, height, rows of (u, v) float pairs."""
    rng = np.random.default_rng(5)
    flow = rng.normal(size=(7, 9, 2)).astype(np.float32)
    p = tmp_path / "f.flo"
    write_flo(p, flow)
    raw = p.read_bytes()
    assert raw[:4] == b"PIEH" and np.frombuffer(raw[4:12], "<i4").tolist() == [9, 7] and len(raw) == 12 + flow.nbytes
    assert np.array_equal(read_flo(p), flow)
    assert read_flo(p, np.int32).dtype == np.int32 and np.array_equal(read_flo(p, np.int32), flow.astype(np.int32))
    p.write_bytes(b"HEIP" + raw[4:])
    assert read_flo(p) is None
    p.write_bytes(raw[:-1])
    assert read_flo(p) is None
    p.write_bytes(b"PIEH" + np.array([0, 3], "<i4").tobytes())
    assert read_flo(p) is None
    assert read_flo(tmp_path / "missing.flo") is None
    with pytest.raises(ValueError):
        write_flo(p, np.zeros((3, 3)))
