"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol the header
declares, and compute calls fail loudly (no CPU fallback) when no HIP device is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from libstevi_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions(names=("stevi_hip.h", "stevi_hip_test.h")):
    """every function any header under include/ declares (stevi_hip.h: the product surface; stevi_hip_test.h: the tests' A/B switches)"""
    found = set()
    for name in names:
        text = open(os.path.join(ROOT, "include", name)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        found |= set(re.findall(r"\b(svh_[a-z0-9_]+)\s*\(", text))
    return sorted(found)


def test_headers_under_include_are_the_ones_checked():
    assert sorted(os.listdir(os.path.join(ROOT, "include"))) == ["stevi_hip.h", "stevi_hip_test.h"]


def test_public_header_documents_at_most_eight_options():
    """the product surface carries the options a caller chooses between; the tests' A/B switches live elsewhere"""
    text = open(os.path.join(ROOT, "include", "stevi_hip.h")).read()
    doc = text[text.index("/* Options of a context"):text.index("int svh_context_set_option")]
    options = re.findall(r'^ \* "([a-z_0-9]+)" \(default', doc, flags=re.M)
    assert 1 <= len(options) <= 8, options
    assert "svh_test_set_option" not in header_functions(("stevi_hip.h",))


def test_library_exports_every_declared_symbol():
    lib = _capi.load()
    declared = header_functions()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in include/stevi_hip.h but not exported"
    assert sorted(_capi.EXPORTS) == declared


def test_struct_layouts_match_header():
    # svh_array: pointer + 4 x int32 + 2 x int64[4]; svh_stereo_params: 19 x 4-byte fields
    assert C.sizeof(_capi.SvhArray) == 8 + 16 + 64
    assert C.sizeof(_capi.SvhStereoParams) == 19 * 4


def test_status_strings():
    lib = _capi.load()
    assert lib.svh_status_string(_capi.OK) == b"ok"
    assert b"empty" in lib.svh_status_string(_capi.EMPTY_RESULT)


def test_shape_helper_runs_without_gpu():
    lib = _capi.load()
    img = np.zeros((7, 9), np.float32)
    from libstevi_amd.correlation import _desc
    shp = (C.c_int64 * 3)()
    assert lib.svh_unfold_shape(C.byref(_desc(img)), 2, 1, None, shp) == _capi.OK
    assert list(shp) == [7, 9, 15]
    pad = (C.c_int32 * 4)(0, 0, 0, 0)
    assert lib.svh_unfold_shape(C.byref(_desc(img)), 2, 1, pad, shp) == _capi.OK
    assert list(shp) == [5, 5, 15]


def test_no_cpu_fallback():
    lib = _capi.load()
    if lib.svh_device_available():
        pytest.skip("a HIP device is present")
    import libstevi_amd as sv
    with pytest.raises(_capi.SvhError) as e:
        sv.unfoldBasedCostVolume(sv.matchingFunctions.SAD, np.zeros((4, 6), np.float32), np.zeros((4, 6), np.float32), 1, 1, 3)
    assert e.value.status == _capi.ERR_NO_DEVICE
