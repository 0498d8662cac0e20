#!/usr/bin/env python3
"""tests/golden/cornell_gt_400.npz from the reference's only reference-held render, tools/cornell-gt.exr
(400 x 400, HALF, channels B G R, uncompressed; provenance and settings undocumented).
Run in the build container (where /root/reference exists); the fixture holds pixel values only:
    rgb   float16 (400, 400, 3), row 0 = top scanline, channels R G B, columns as stored in the file
          (horizontally MIRRORED relative to the current camera convention: the test flips them)
The file is decoded with this repository's own EXR reader (pathed_amd/host/exr.cpp)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pathed_amd import _capi

source = "/root/reference/tools/cornell-gt.exr"
host = _capi.load_host()
w, h = C.c_int(), C.c_int()
assert host.pathed_host_read_exr_rgba(source.encode(), C.byref(w), C.byref(h), None, 0) == 0, host.pathed_host_last_error()
data = np.zeros((h.value, w.value, 4), dtype=np.float32)
assert host.pathed_host_read_exr_rgba(source.encode(), C.byref(w), C.byref(h), data.ctypes.data_as(C.POINTER(C.c_float)), data.size) == 0
rgb = data[..., :3].astype(np.float16)
assert np.array_equal(rgb.astype(np.float32), data[..., :3])   # the file stores HALF: nothing is lost
out = os.path.join(ROOT, "tests", "golden", "cornell_gt_400.npz")
np.savez_compressed(out, rgb=rgb)
print(out, rgb.shape, os.path.getsize(out), "bytes; mean rgb", data[..., :3].mean(axis=(0, 1)))
