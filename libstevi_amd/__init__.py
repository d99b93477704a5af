"""libstevi_amd -- MI355X (gfx950) implementation of LibStevi's correlation/ hot path.

csrc/            hand-written HIP kernels + the C ABI (include/stevi_hip.h) -> libstevi_hip.so
include/         C++ drop-in headers with the reference's names (StereoVision::Correlation)
correlation.py   Python mirror of the same functions over the C ABI (numpy = host arrays, torch = device arrays)
sharded.py       disparity sharding over the GPUs of a node (torch.distributed / RCCL all-reduce of winner keys)
stevimg.py       the reference's .stevimg array files (io/image_io.h), the data format either side of the path
"""
from . import _capi  # noqa: F401
from .correlation import *  # noqa: F401,F403
from .correlation import (Margins, PaddingMargins, searchOffset1, searchOffset2, matchFuncStrategy, context_for, profile_enable,  # noqa: F401
                          profile_reset, profile_collect, set_option, set_test_option)
from .stevimg import read_flo, read_stevimg, write_flo, write_stevimg  # noqa: F401,E402
