#!/usr/bin/env python3
"""A/B of the fused kernel's phase 1: pairs of triangles (exact test) against item records with parallelograms
(pathed_amd/csrc/small_items.h), interleaved repeats on one box, images compared bit for bit.
Usage: ab_quads.py [--spp 256] [--repeats 3] [--scenes C2,C3,ON]"""
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
    "C1": ("scenes/cornell.json", 256, 256),
    "C2": ("scenes/cornell.json", 1024, 1024),
    "C3": ("scenes/mis-pbrt.json", 1024, 1024),
    "ON": ("scenes/cornell-oren-nayar.json", 1024, 1024),
    "GL": ("scenes/cornell-glossy.json", 1024, 1024),
}


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--spp", type=int, default=256)
    parser.add_argument("--repeats", type=int, default=3)
    parser.add_argument("--scenes", default="C2,C3,ON")
    args = parser.parse_args()
    for name in args.scenes.split(","):
        path, w, h = SCENES[name]
        scene = LoadedScene(path, w, h)
        variants = {}
        os.environ["PATHED_NO_QUADS"] = "1"          # read when the scene's phase-1 records are built
        variants["pairs"] = HipScene(scene.desc, device=0)
        del os.environ["PATHED_NO_QUADS"]
        variants["items"] = HipScene(scene.desc, device=0)
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
        same = bool(torch.equal(accum["pairs"], accum["items"]))
        print("%s %s %dx%d x %d spp (%d triangles): pairs %s  items %s  Msamples/s  (best %.0f -> %.0f, %+.1f %%)  images identical: %s" % (
            name, path, w, h, args.spp, scene.n_triangles, " ".join("%.0f" % r for r in rates["pairs"]), " ".join("%.0f" % r for r in rates["items"]),
            max(rates["pairs"]), max(rates["items"]), 100.0 * (max(rates["items"]) / max(rates["pairs"]) - 1.0), same), flush=True)


if __name__ == "__main__":
    main()
