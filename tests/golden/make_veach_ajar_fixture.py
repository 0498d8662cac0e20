"""Builds tests/golden/veach_ajar_tungsten_blocks.npz from the render the reference ships beside its scene file
(scenes/veach-ajar/TungstenRender.exr, 1280x720, PIZ; a third-party render of the same scene made with Tungsten).

    python tests/golden/make_veach_ajar_fixture.py [/root/reference]

Only VALUES go into the fixture: the means of the 16 x 16 pixel blocks (45 x 80 x 3 floats, top row first, as the EXR
stores them).  Run in the build container, where the reference is mounted; the fixture travels to the GPU box."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pathed_amd import _capi  # noqa: E402

BLOCK = 16


def read_exr(path):
    host = _capi.load_host()
    w, h = C.c_int(), C.c_int()
    if host.pathed_host_read_exr_rgba(path.encode(), C.byref(w), C.byref(h), None, 0) != 0:
        raise RuntimeError(host.pathed_host_last_error().decode())
    data = np.zeros((h.value, w.value, 4), dtype=np.float32)
    if host.pathed_host_read_exr_rgba(path.encode(), C.byref(w), C.byref(h), data.ctypes.data_as(C.POINTER(C.c_float)), data.size) != 0:
        raise RuntimeError(host.pathed_host_last_error().decode())
    return data


def block_means(image):
    h, w = image.shape[:2]
    return image[:h - h % BLOCK, :w - w % BLOCK, :3].reshape(h // BLOCK, BLOCK, w // BLOCK, BLOCK, 3).mean(axis=(1, 3), dtype=np.float64).astype(np.float32)


def main():
    reference = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    image = read_exr(os.path.join(reference, "scenes", "veach-ajar", "TungstenRender.exr"))
    blocks = block_means(image)
    out = os.path.join(ROOT, "tests", "golden", "veach_ajar_tungsten_blocks.npz")
    np.savez_compressed(out, blocks=blocks, width=image.shape[1], height=image.shape[0], block=BLOCK)
    print("wrote %s: %s, mean rgb %s" % (out, blocks.shape, blocks.mean(axis=(0, 1))))


if __name__ == "__main__":
    main()
