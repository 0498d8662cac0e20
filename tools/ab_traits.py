#!/usr/bin/env python3
"""A/B of the scene-specialised instantiations of the fused path kernel (shading.h: SceneTraits) against the generic one
(PATHED_NO_SCENE_TRAITS=1, read at scene creation), interleaved repeats in one process, images compared bit for bit.
Usage: ab_traits.py [--spp 256] [--repeats 3] [--scenes ON,GL,GGX,GLASS]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from pathed_amd.integrator import HipScene
from pathed_amd.scene import LoadedScene

SCENES = {
    "C2": ("scenes/cornell.json", 1024, 1024),
    "C3": ("scenes/mis-pbrt.json", 1024, 1024),
    "ON": ("scenes/cornell-oren-nayar.json", 1024, 1024),
    "GL": ("scenes/cornell-glossy.json", 1024, 1024),
    "GGX": ("scenes/cornell-ggx.json", 1024, 1024),
    "GLASS": ("scenes/cornell-glass.json", 1024, 1024),
}


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--spp", type=int, default=256)
    parser.add_argument("--repeats", type=int, default=3)
    parser.add_argument("--scenes", default="ON,GL,GGX,GLASS")
    args = parser.parse_args()
    for name in args.scenes.split(","):
        path, w, h = SCENES[name]
        scene = LoadedScene(path, w, h)
        variants = {}
        os.environ["PATHED_NO_SCENE_TRAITS"] = "1"
        variants["generic"] = HipScene(scene.desc, device=0)
        del os.environ["PATHED_NO_SCENE_TRAITS"]
        variants["narrowed"] = HipScene(scene.desc, device=0)
        accum = {k: torch.zeros((h, w, 3), dtype=torch.float32, device="cuda") for k in variants}
        for k, gpu in variants.items():
            gpu.render_device(1, 0, 16, 0, 10, accum[k].data_ptr())
            accum[k].zero_()
        rates = {k: [] for k in variants}
        for _ in range(args.repeats):
            for k, gpu in variants.items():
                accum[k].zero_()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                gpu.render_device(1, 0, args.spp, 0, 10, accum[k].data_ptr())
                torch.cuda.synchronize()
                rates[k].append(w * h * args.spp / (time.perf_counter() - t0) / 1e6)
        same = bool(torch.equal(accum["generic"], accum["narrowed"]))
        print("%s %s %dx%d x %d spp (%d triangles): generic %s  narrowed %s  Msamples/s  (best %.0f -> %.0f, %+.1f %%)  images identical: %s" % (
            name, path, w, h, args.spp, scene.n_triangles, " ".join("%.0f" % r for r in rates["generic"]), " ".join("%.0f" % r for r in rates["narrowed"]),
            max(rates["generic"]), max(rates["narrowed"]), 100.0 * (max(rates["narrowed"]) / max(rates["generic"]) - 1.0), same), flush=True)


if __name__ == "__main__":
    main()
