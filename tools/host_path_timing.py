#!/usr/bin/env python3
"""PCIe-inclusive rate of the hot path: svh_stereo_match called with HOST arrays (numpy), i.e. what the C++ drop-in
headers do for a caller that keeps Multidim::Array data in host memory.  Never the headline value (bench.py times
HBM-resident inputs); DESIGN.md quotes this number."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402,F401

import libstevi_amd as sv  # noqa: E402
from helpers import parallax_pair  # noqa: E402

src, tgt, _ = parallax_pair(1080, 1920, 320, 320, 380, 8, 64, 3)
for _ in range(3):
    sv.stereoMatch(sv.matchingFunctions.CENSUS, tgt, src, 4, 4, 256, sgmDirections=8)
n = 20
t0 = time.perf_counter()
for _ in range(n):
    out = sv.stereoMatch(sv.matchingFunctions.CENSUS, tgt, src, 4, 4, 256, sgmDirections=8)
dt = (time.perf_counter() - t0) / n
print(f"host arrays in / host disparity out: {dt * 1e3:.3f} ms per 1920x1080xD256 pair = {1920 * 1080 * 256 / dt / 1e6:.0f} Mdisparities/s "
      f"(H2D 2 x 8.3 MB, D2H 8.3 MB, pageable memory), checksum {int(out['disp'].sum())}")
