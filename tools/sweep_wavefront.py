#!/usr/bin/env python3
"""The wavefront's occupancy split after the local rays (k_trace has less to do): trace blocks per CU x pools, with the
per-launch kernel times (HIP events around every 8th launch).  tools/sweep_wavefront.py [--spp 256] [--scenes C4,C5]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pathed_amd.integrator import HipScene
from pathed_amd.scene import LoadedScene
SCENES = {"C4": ("scenes/teapot.json", 1024, 1024), "C5": ("scenes/dragon-standin.json", 1920, 1080), "C5close": ("scenes/dragon-standin-close.json", 1920, 1080)}
parser = argparse.ArgumentParser()
parser.add_argument("--spp", type=int, default=256)
parser.add_argument("--scenes", default="C4,C5")
parser.add_argument("--blocks", default="0,1,2,3")
parser.add_argument("--pools", default="1,2,3")
parser.add_argument("--local", default="0")
args = parser.parse_args()
for key in args.scenes.split(","):
    path, w, h = SCENES[key]
    scene = LoadedScene(path, w, h)
    builder = "ploc" if scene.n_triangles > 1000000 else "sah"
    reference = None
    for local in (int(v) for v in args.local.split(",")):
        for pools in (int(v) for v in args.pools.split(",")):
            for blocks in (int(v) for v in args.blocks.split(",")):
                gpu = HipScene(scene.desc, device=0, bvh_builder=builder, shade_kernel="per-slot", pools=pools, trace_blocks_per_cu=blocks, local_rays=local)
                accum = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
                gpu.render_device(1, 0, 16, 0, 10, accum.data_ptr())
                accum.zero_()
                gpu.set_stats_mode(time_sampled=True)
                gpu.reset_stats()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                gpu.render_device(1, 0, args.spp, 0, 10, accum.data_ptr())
                torch.cuda.synchronize(); rate = w * h * args.spp / (time.perf_counter() - t0) / 1e6
                stats = gpu.stats()
                launches = max(stats["trace_launches"], 1)
                if reference is None: reference = accum.clone()
                print("%-7s local_rays %d pools %d trace blocks per CU %d: %7.1f Msamples/s  trace %6.1f us  shade %6.1f us per launch (%d launches)  identical %s" % (
                    key, local, pools, blocks, rate, 1e3 * stats["trace_ms"] / launches, 1e3 * stats["shade_ms"] / launches, stats["trace_launches_all"], bool(torch.equal(accum, reference))), flush=True)
                gpu.close()
