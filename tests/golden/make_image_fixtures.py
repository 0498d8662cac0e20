#!/usr/bin/env python3
"""Writes tests/golden/images.npz: small per-pixel radiance SUMS rendered by the CPU oracle.

They pin the oracle's estimator against regressions and give the GPU tests a committed target
that does not depend on the oracle library being rebuilt identically.  Regenerate with
    python tests/golden/make_image_fixtures.py
(The function-level fixtures come from the reference's object code instead: see
oracle/ref_driver.cpp and oracle/Makefile.ref.)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle_lib  # noqa: E402
from pathed_amd.scene import LoadedScene  # noqa: E402

CASES = {
    # name: (scene, width, height, seed, spp_begin, spp_count, start_bounce, last_bounce)
    "cornell_32": ("scenes/cornell.json", 32, 32, 1, 0, 4, 0, 10),
    "cornell_window_2_3": ("scenes/cornell.json", 24, 24, 5, 3, 3, 2, 3),
    "cornell_glass_24": ("scenes/cornell-glass.json", 24, 24, 2, 0, 4, 0, 6),
    "cornell_glossy_24": ("scenes/cornell-glossy.json", 24, 24, 2, 0, 4, 0, 6),
    "oren_nayar_24": ("scenes/cornell-oren-nayar.json", 24, 24, 3, 0, 4, 0, 5),
    "ggx_24": ("scenes/cornell-ggx.json", 24, 24, 8, 0, 4, 0, 5),
    "mis_32x24": ("scenes/mis-pbrt.json", 32, 24, 4, 0, 4, 0, 4),
    "teapot_32x24": ("scenes/teapot.json", 32, 24, 6, 0, 3, 0, 8),
    "env_sampling_24": ("test_scenes/environment_map_sampling.json", 24, 24, 7, 0, 8, 0, 3),
}


def main():
    arrays = {}
    for name, (path, w, h, seed, begin, count, sb, lb) in CASES.items():
        scene = LoadedScene(path, w, h)
        oracle = oracle_lib.OracleScene(scene.desc)
        image, stats = oracle.render(w, h, seed, begin, count, sb, lb, threads=1)
        arrays[name] = image
        print(name, image.reshape(-1, 3).mean(0) / count, stats["dropped"])
    np.savez_compressed(os.path.join(HERE, "images.npz"), **arrays)


if __name__ == "__main__":
    main()
