#!/usr/bin/env python3
"""GPU PathTracer against the reference's own converged render (tools/cornell-gt.exr, committed as the value
fixture tests/golden/cornell_gt_400.npz): MSE / AE / MRSE exactly as the reference's tools/error_reports.py:13-23
at 4^k spp, the 1/spp law of the error, the floor it settles on, block-mean (bias) and z-score statistics.
    python tools/gt_compare.py [--max-spp 16384] [--json out.json]
pathed_amd.gt_metrics holds the arithmetic (shared with tests/test_gpu_gt.py)."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pathed_amd import gt_metrics
from pathed_amd.integrator import HipScene
from pathed_amd.scene import LoadedScene

parser = argparse.ArgumentParser()
parser.add_argument("--max-spp", type=int, default=16384)
parser.add_argument("--last-bounce", type=int, default=gt_metrics.GT_LAST_BOUNCE)
parser.add_argument("--json", default="")
args = parser.parse_args()

gt = gt_metrics.load_gt()
scene = LoadedScene(gt_metrics.GT_SCENE, 400, 400)
gpu = HipScene(scene.desc, device=0)
report = gt_metrics.compare(lambda seed, begin, count: gpu.render(seed, begin, count, 0, args.last_bounce), gt, args.max_spp)
for row in report["levels"]:
    print("spp %6d  MSE %.6e  AE %.6e  MRSE %.6e   (no-light pixels: MSE %.6e)" % (row["spp"], row["mse"], row["ae"], row["mrse"], row["mse_dim"]))
for key in ("law_ratios", "floor_mse_dim", "gt_noise_mse_dim", "variance_per_sample_dim", "gt_equivalent_spp", "block_rel_p50", "block_rel_p95", "block_rel_max",
            "block_z_median", "block_z_p90", "block_z_over_4", "pixel_z_median", "pixel_z_p90", "pixel_z_over_4", "mean_rgb", "gt_mean_rgb", "energy_ratio"):
    print(key, report[key])
if args.json:
    json.dump(report, open(args.json, "w"), indent=1)
