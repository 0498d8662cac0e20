#!/usr/bin/env python3
"""What ONE GPU says about the 2 / 4 / 8-GPU curve (samples shard over the GPUs, one RCCL reduce at the end).

bench.py --gpus N splits every step's samples over the ranks (strong scaling): a rank of N renders spp_per_step / N samples
per call.  The scene is replicated and nothing is exchanged inside the timed region but ONE reduce of 3*W*H floats, so the
N-GPU rate is N x the rate ONE GPU reaches on calls of that size, minus the reduce: this tool times exactly those calls --
`steps` calls of spp_per_step / N samples after `warmup` of them, as bench.py's loop makes them -- and prints
predicted_efficiency_N = rate(share) / rate(full step) per configuration, with the reduce charged at a stated rate.

    tools/predict_scaling.py [--configs C2,C3,C4,C5] [--steps 4] [--warmup 2]
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pathed_amd.integrator import HipScene
from pathed_amd.scene import LoadedScene

CONFIGS = {   # scene, width, height, spp per step of the single-GPU run (bench.py default: 1024)
    "C2": ("scenes/cornell.json", 1024, 1024, 1024),
    "C3": ("scenes/mis-pbrt.json", 1024, 1024, 1024),
    "C4": ("scenes/teapot.json", 1024, 1024, 1024),
    "C5": ("scenes/dragon-standin.json", 1920, 1080, 1024),
}
REDUCE_GBS = 100.0   # what a 12.6 - 24.9 MB ncclReduce over xGMI is charged at (7 links x ~153 GB/s per GPU; a fraction of that, to be safe)


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--configs", default="C2,C3,C4,C5")
    parser.add_argument("--steps", type=int, default=4)
    parser.add_argument("--warmup", type=int, default=2)
    args = parser.parse_args()
    for key in args.configs.split(","):
        path, w, h, step_spp = CONFIGS[key]
        scene = LoadedScene(path, w, h)
        gpu = HipScene(scene.desc, device=0, bvh_builder="ploc" if scene.n_triangles > 1000000 else "sah")
        accum = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
        row = {"config": key, "scene": path, "res": "%dx%d" % (w, h), "spp_per_step": step_spp, "steps": args.steps, "ranks": {}}
        full = None
        for ranks in (1, 2, 4, 8):
            share = step_spp // ranks
            for index in range(args.warmup):
                gpu.render_device(1, index * step_spp, share, 0, 10, accum.data_ptr())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for index in range(args.steps):
                gpu.render_device(1, (args.warmup + index) * step_spp, share, 0, 10, accum.data_ptr())
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0
            rate = w * h * share * args.steps / elapsed / 1e6
            reduce_s = 12.0 * w * h / (REDUCE_GBS * 1e9) if ranks > 1 else 0.0
            with_reduce = w * h * share * args.steps / (elapsed + reduce_s) / 1e6
            if full is None:
                full = rate
            row["ranks"][str(ranks)] = {"spp_per_call": share, "samples_per_call": w * h * share, "path_kernel": gpu.stats()["path_kernel"],
                                        "Msamples_per_s_per_gpu": round(rate, 1), "predicted_efficiency": round(with_reduce / full, 3),
                                        "predicted_Msamples_per_s": round(with_reduce * ranks, 0)}
        print(json.dumps(row), flush=True)
        gpu.close()


if __name__ == "__main__":
    sys.exit(main())
