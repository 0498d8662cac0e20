#!/usr/bin/env python3
"""Runs the five BASELINE.json configurations (stand-in assets where the reference ships none) on
one GPU at the target resolution with reduced spp, plus a small-size parity check against the CPU
oracle and the oracle's own rate on the host cores.  Output: one table row per config."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from pathed_amd.integrator import HipScene
from pathed_amd.scene import LoadedScene
import oracle_lib

CONFIGS = [
    # name, scene, width, height, gpu spp, cpu spp (timing), note
    ("C1", "scenes/cornell.json", 256, 256, 16, 16, "real scene"),
    ("C2", "scenes/cornell.json", 1024, 1024, 1024, 8, "real scene; Lambertian"),
    ("C3", "scenes/mis-pbrt.json", 1024, 1024, 256, 8, "plates/floor synthetic; plastic+Beckmann, sphere lights"),
    ("C4", "scenes/teapot.json", 1024, 1024, 256, 4, "mesh + env synthetic; glass, checkerboard, env light"),
    ("C5", "scenes/dragon-standin.json", 1920, 1080, 64, 2, "procedural mesh + env synthetic; plastic, env light"),
    # not a BASELINE configuration: the reference's own participating-media scene under its VolumePathTracer
    ("VOL", "scenes/cornell-medium.json", 1024, 1024, 64, 4, "container / frame OBJs synthetic; homogeneous gas, glass sphere", "VolumePathTracer"),
]


def relative_l2(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--dragon", type=int, default=9)
    parser.add_argument("--only", default="")
    parser.add_argument("--full", action="store_true", help="render the spp BASELINE.json names (C2 4096, C3 1024, C4 2048, C5 8192) on this one GPU")
    parser.add_argument("--builder", default="sah", choices=["sah", "lbvh", "ploc"])
    parser.add_argument("--parity-full", type=int, default=0, metavar="SPP",
                        help="also compare GPU and CPU oracle at the configuration's FULL resolution with this many spp")
    args = parser.parse_args()
    full_spp = {"C1": 16, "C2": 4096, "C3": 1024, "C4": 2048, "C5": 8192, "VOL": 1024}
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_assets.py"), "--dragon", str(args.dragon)], check=True,
                   stdout=subprocess.DEVNULL)
    cores = oracle_lib.host_threads()   # what the job may run at once (oracle_lib), not the machine's 256 hardware threads
    rows = []
    for name, path, w, h, spp, cpu_spp, note, *integrator in CONFIGS:
        integrator = integrator[0] if integrator else "PathTracer"
        if args.only and name not in args.only.split(","):
            continue
        if args.full:
            spp = full_spp[name]
        scene = LoadedScene(path, w, h)
        t0 = time.perf_counter()
        gpu = HipScene(scene.desc, device=0, bvh_builder=args.builder)
        setup = time.perf_counter() - t0
        gpu.set_integrator(integrator)
        accum = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
        gpu.render_device(1, 0, min(spp, 128), 0, 10, accum.data_ptr())  # warm-up (long enough for the clocks: the first call of a process runs 3 - 8 % low)
        accum.zero_()
        gpu.set_stats_mode(count=False, time_sampled=True)   # HIP events around every 8th launch
        gpu.reset_stats()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gpu.render_device(1, 0, spp, 0, 10, accum.data_ptr())
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        timed = gpu.stats()
        if timed["trace_launches"]:
            # extrapolate the sampled launches to all of them
            scale = timed["trace_launches_all"] / timed["trace_launches"]
            timed["trace_ms"] *= scale
            timed["shade_ms"] *= scale
        gpu.set_stats_mode(count=True)
        gpu.reset_stats()
        scratch = torch.zeros_like(accum)
        count_spp = min(spp, 16)
        gpu.render_device(1, 0, count_spp, 0, 10, scratch.data_ptr())
        counted = gpu.stats()
        gpu.set_stats_mode(count=False)
        rays = counted["closest_rays"] + counted["shadow_rays"]
        alg = 48 * counted["closest_rays"] + 36 * counted["shadow_rays"] + 32 * counted["nodes_visited"] + 48 * counted["tris_tested"]
        alg_total = alg * spp / count_spp
        # what the TIMED call ran (count mode never takes k_path_wave; PathedStats.path_kernel is the last render call's)
        fused = timed["path_kernel"] in (3, 4, 6, 7)   # one kernel carries the whole path: there is no separate trace kernel to rate
        gbs = alg_total / (timed["trace_ms"] * 1e-3) / 1e9 if timed["trace_ms"] and not fused else 0.0

        # CPU oracle rate at the same resolution
        oracle = oracle_lib.OracleScene(scene.desc)
        oracle.set_integrator(integrator)
        t0 = time.perf_counter()
        oracle.render(w, h, 1, 0, cpu_spp, 0, 10, threads=cores)
        cpu_elapsed = time.perf_counter() - t0

        # parity at a size the oracle finishes quickly
        pw, ph = (96, 96) if w == h else (128, 72)
        small = LoadedScene(path, pw, ph)
        g2, o2 = HipScene(small.desc, device=0), oracle_lib.OracleScene(small.desc)
        g2.set_integrator(integrator)
        o2.set_integrator(integrator)
        image = g2.render(1, 0, 16, 0, 10)
        expected, _ = o2.render(pw, ph, 1, 0, 16, 0, 10, threads=cores)
        full_parity = None
        if args.parity_full > 0:
            n = args.parity_full
            image_full = gpu.render(1, 0, n, 0, 10)
            expected_full, _ = oracle.render(w, h, 1, 0, n, 0, 10, threads=cores)
            difference = np.abs(image_full - expected_full)
            full_parity = {
                "spp": n, "relL2": "%.2e" % relative_l2(image_full, expected_full),
                "pixels_bit_identical": round(float((image_full == expected_full).all(axis=2).mean()), 4),
                "pixels_off_by_more_than_1_percent": round(float((difference > 1e-2 * np.maximum(np.abs(expected_full), 1e-3)).any(axis=2).mean()), 6),
            }
        rows.append({
            "config": name, "scene": path, "integrator": integrator, "res": "%dx%d" % (w, h), "spp": spp, "note": note,
            "triangles": scene.n_triangles, "scene_create_s": round(setup, 2), "bvh_builder": args.builder,
            "bvh_build_ms": round(timed["bvh_build_ms"], 1), "render_s": round(elapsed, 3),
            "gpu_Msamples_s": round(w * h * spp / elapsed / 1e6, 1),
            "rays_per_sample": round(rays / (w * h * count_spp), 2),
            "trace_Grays_s": round(rays * spp / count_spp / timed["trace_ms"] / 1e6, 2) if timed["trace_ms"] and not fused else None,
            # (a tree that fits the 4 MiB L2 of an XCD is read from there: its node bytes per second are not an HBM figure)
            "trace_algorithmic_GBs": round(gbs, 0) if not fused else None,
            "frac_of_8TBs": round(gbs / 8000.0, 3) if not fused and counted["bvh_bytes"] > (4 << 20) else None,
            "bvh_bytes": counted["bvh_bytes"],
            "intersector": ["BVH in HBM" if counted["bvh_bytes"] > (4 << 20) else "BVH in L2 (%.2f MB)" % (counted["bvh_bytes"] / 1e6), "BVH in LDS", "all triangles (scalar loads)"][counted["scene_in_lds"]],
            "path_kernel": ["", "wavefront: k_trace + k_shade", "wavefront: k_trace + k_shade_staged", "fused: k_path_small", "volume: k_path_volume", "wavefront: k_trace + k_vertex + k_regen", "wave: k_path_wave", "hybrid: k_path_hybrid"][timed["path_kernel"]],
            "cpu_oracle_Msamples_s": round(w * h * cpu_spp / cpu_elapsed / 1e6, 2), "cpu_cores": cores,
            "relL2_vs_oracle_%dx%d_16spp" % (pw, ph): "%.2e" % relative_l2(image, expected),
            "mean_rgb": [round(float(v), 4) for v in (accum / spp).mean(dim=(0, 1)).tolist()],
            "parity_at_full_resolution": full_parity,
        })
        print(json.dumps(rows[-1]), flush=True)
    if not args.only or "GT" in args.only.split(","):
        # the reference's own error metrics (tools/error_reports.py:13-23) against its only rendered image
        from pathed_amd import gt_metrics
        gt = gt_metrics.load_gt()
        scene = LoadedScene(gt_metrics.GT_SCENE, 400, 400)
        gpu = HipScene(scene.desc, device=0)
        report = gt_metrics.compare(lambda seed, begin, count: gpu.render(seed, begin, count, 0, gt_metrics.GT_LAST_BOUNCE), gt, 16384)
        top = report["levels"][-1]
        rows.append({"config": "GT", "scene": gt_metrics.GT_SCENE + " vs tools/cornell-gt.exr (400x400)", "spp": top["spp"],
                     "MSE": "%.4e" % top["mse"], "AE": "%.4e" % top["ae"], "MRSE": "%.4e" % top["mrse"],
                     "MSE_without_light_pixels": "%.3e" % top["mse_dim"], "ground_truth_noise": "%.3e" % report["gt_noise_mse_dim"],
                     "one_over_spp_law_ratios": [round(r, 2) for r in report["law_ratios"]],
                     "energy_ratio": round(report["energy_ratio"], 4), "block_rel_p50": round(report["block_rel_p50"], 4),
                     "block_rel_p95": round(report["block_rel_p95"], 4)})
        print(json.dumps(rows[-1]), flush=True)
    return rows


if __name__ == "__main__":
    main()
