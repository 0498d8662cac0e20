#!/usr/bin/env python3
"""A/B of the fused kernel's phase 1: VALU (packed Moeller-Trumbore) against the matrix pipe (mfma_candidates.h).
Interleaved repeats on one box, images compared bit for bit.  Usage: ab_phase1.py [--spp 256] [--repeats 3] [--scenes C2,C3]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from pathed_amd.integrator import HipScene
from pathed_amd.scene import LoadedScene

SCENES = {
    "C2": ("scenes/cornell.json", 1024, 1024, "PathTracer"),
    "C3": ("scenes/mis-pbrt.json", 1024, 1024, "PathTracer"),
    "ON": ("scenes/cornell-oren-nayar.json", 1024, 1024, "PathTracer"),
}


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--spp", type=int, default=256)
    parser.add_argument("--repeats", type=int, default=3)
    parser.add_argument("--scenes", default="C2,C3")
    args = parser.parse_args()
    for name in args.scenes.split(","):
        path, w, h, integrator = SCENES[name]
        scene = LoadedScene(path, w, h)
        variants = {}
        for phase1 in ("valu", "mfma"):
            gpu = HipScene(scene.desc, device=0, small_phase1=phase1)
            gpu.set_integrator(integrator)
            variants[phase1] = gpu
        accum = {k: torch.zeros((h, w, 3), dtype=torch.float32, device="cuda") for k in variants}
        for k, gpu in variants.items():
            gpu.render_device(1, 0, 16, 0, 10, accum[k].data_ptr())
            accum[k].zero_()
        rates = {k: [] for k in variants}
        for repeat in range(args.repeats):
            for k, gpu in variants.items():
                accum[k].zero_()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                gpu.render_device(1, 0, args.spp, 0, 10, accum[k].data_ptr())
                torch.cuda.synchronize()
                rates[k].append(w * h * args.spp / (time.perf_counter() - t0) / 1e6)
        same = bool(torch.equal(accum["valu"], accum["mfma"]))
        print("%s %dx%d x %d spp: valu %s  mfma %s  Msamples/s  (best %.0f -> %.0f, %+.1f %%)  images identical: %s" % (
            name, w, h, args.spp, " ".join("%.0f" % r for r in rates["valu"]), " ".join("%.0f" % r for r in rates["mfma"]),
            max(rates["valu"]), max(rates["mfma"]), 100.0 * (max(rates["mfma"]) / max(rates["valu"]) - 1.0), same), flush=True)


if __name__ == "__main__":
    main()
