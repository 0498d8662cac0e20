#!/usr/bin/env python3
"""k_path_hybrid's burst scheduling (PathedSceneOptions.hybrid_batch / hybrid_ready / wave_stragglers), one process, images compared:
tools/sweep_hybrid.py [--spp 128] [--scenes GL,GLASS,C4,C5]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pathed_amd.integrator import HipScene
from pathed_amd.scene import LoadedScene
SCENES = {"GL": ("scenes/cornell-glossy.json", 1024, 1024), "GLASS": ("scenes/cornell-glass.json", 1024, 1024),
          "C4": ("scenes/teapot.json", 1024, 1024), "C5": ("scenes/dragon-standin.json", 1920, 1080)}
parser = argparse.ArgumentParser()
parser.add_argument("--spp", type=int, default=128)
parser.add_argument("--scenes", default="GL,GLASS,C4,C5")
parser.add_argument("--batches", default="1,24,40,56,80")
parser.add_argument("--readies", default="-1,16,28,40")
parser.add_argument("--stragglers", default="16")
args = parser.parse_args()
for key in args.scenes.split(","):
    path, w, h = SCENES[key]
    scene = LoadedScene(path, w, h)
    builder = "ploc" if scene.n_triangles > 1000000 else "sah"
    reference = None
    for stragglers in (int(v) for v in args.stragglers.split(",")):
        for batch in (int(v) for v in args.batches.split(",")):
            for ready in (int(v) for v in args.readies.split(",")):
                gpu = HipScene(scene.desc, device=0, bvh_builder=builder, shade_kernel="hybrid", wave_stragglers=stragglers, hybrid_batch=batch, hybrid_ready=ready)
                accum = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
                gpu.render_device(1, 0, 8, 0, 10, accum.data_ptr())
                accum.zero_()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                gpu.render_device(1, 0, args.spp, 0, 10, accum.data_ptr())
                torch.cuda.synchronize(); rate = w * h * args.spp / (time.perf_counter() - t0) / 1e6
                if reference is None: reference = accum.clone()
                print("%-5s stragglers %3d batch %3d ready %3d: %7.1f Msamples/s  identical %s" % (key, stragglers, batch, ready, rate, bool(torch.equal(accum, reference))), flush=True)
                gpu.close()
