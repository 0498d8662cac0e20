#!/usr/bin/env python3
"""k_path_hybrid's burst scheduling (PathedSceneOptions.wave_stragglers / wave_refill), one process, images compared:
tools/sweep_hybrid.py [--spp 128]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pathed_amd.integrator import HipScene
from pathed_amd.scene import LoadedScene
parser = argparse.ArgumentParser()
parser.add_argument("--spp", type=int, default=128)
args = parser.parse_args()
for path in ("scenes/cornell-glossy.json", "scenes/cornell-glass.json"):
    scene = LoadedScene(path, 1024, 1024)
    reference = None
    for stragglers in (-1, 4, 8, 12, 16, 24, 32, 48):
        for refill in (24, 40, 56):
            gpu = HipScene(scene.desc, device=0, wave_stragglers=stragglers, wave_refill=refill)
            accum = torch.zeros((1024, 1024, 3), dtype=torch.float32, device="cuda")
            gpu.render_device(1, 0, 8, 0, 10, accum.data_ptr())
            accum.zero_()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            gpu.render_device(1, 0, args.spp, 0, 10, accum.data_ptr())
            torch.cuda.synchronize(); rate = 1024 * 1024 * args.spp / (time.perf_counter() - t0) / 1e6
            if reference is None: reference = accum.clone()
            print("%s stragglers %3d refill %2d: %7.1f Msamples/s  identical %s" % (path, stragglers, refill, rate, bool(torch.equal(accum, reference))), flush=True)
            gpu.close()
