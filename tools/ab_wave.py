#!/usr/bin/env python3
"""A/B of the BVH scenes' path tracer: the wavefront kernels (k_trace + k_shade over slot state in HBM) against k_path_wave
(pathed_amd/csrc/path_wave.h: paths in registers, the wave's rays shared through LDS; PATHED_SHADE_KERNEL=wave).  One child process
per variant and repeat, interleaved on one box; images compared by hash.
Usage: ab_wave.py [--spp 64] [--repeats 2] [--scenes C4,C5] [--dragon 9] [--stragglers 24,...] [--refill 48,...]  (variants: wave/<stragglers>/<refill>)"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SCENES = {
    "C4": ("scenes/teapot.json", 1024, 1024),
    "C4s": ("scenes/teapot.json", 256, 256),
    "C5": ("scenes/dragon-standin.json", 1920, 1080),
    "C5s": ("scenes/dragon-standin.json", 320, 180),
    "GL": ("scenes/cornell-glossy.json", 1024, 1024),
    "GLASS": ("scenes/cornell-glass.json", 1024, 1024),
    "BUNNY": ("scenes/cornell-bunny.json", 1024, 1024),
    "C4m": ("scenes/teapot.json", 512, 512),
    "C5m": ("scenes/dragon-standin.json", 960, 540),
}


def child(name, spp):
    import torch
    from pathed_amd.integrator import HipScene
    from pathed_amd.scene import LoadedScene
    path, w, h = SCENES[name]
    scene = LoadedScene(path, w, h)
    gpu = HipScene(scene.desc, device=0)
    accum = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    gpu.render_device(1, 0, min(16, spp), 0, 10, accum.data_ptr())
    best = 0.0
    for _ in range(2):
        accum.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gpu.render_device(1, 0, spp, 0, 10, accum.data_ptr())
        torch.cuda.synchronize()
        best = max(best, w * h * spp / (time.perf_counter() - t0) / 1e6)
    digest = hashlib.sha256(accum.cpu().numpy().tobytes()).hexdigest()[:16]
    print(json.dumps({"rate": best, "digest": digest, "mean": float(accum.mean().item() / spp)}), flush=True)


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--spp", type=int, default=64)
    parser.add_argument("--repeats", type=int, default=2)
    parser.add_argument("--scenes", default="C4,C5")
    parser.add_argument("--dragon", type=int, default=9)
    parser.add_argument("--stragglers", default="24")
    parser.add_argument("--refill", default="48")
    parser.add_argument("--block", default="", help="shade-ready thresholds of the block-ring variant, e.g. 32,40,48 (variants block/<stragglers>/<ready>)")
    parser.add_argument("--child", default=None)
    args = parser.parse_args()
    if args.child:
        child(args.child, args.spp)
        return
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_assets.py"), "--dragon", str(args.dragon)], check=True, stdout=subprocess.DEVNULL)
    variants = {"wavefront": {"PATHED_SHADE_KERNEL": "per-slot"}}
    for value in args.stragglers.split(","):
        for refill in args.refill.split(","):
            variants["wave/%s/%s" % (value, refill)] = {"PATHED_SHADE_KERNEL": "wave", "PATHED_WAVE_STRAGGLERS": value, "PATHED_WAVE_REFILL": refill}
    for ready in [v for v in args.block.split(",") if v]:
        for value in args.stragglers.split(","):
            variants["block/%s/%s" % (value, ready)] = {"PATHED_SHADE_KERNEL": "wave", "PATHED_WAVE_BLOCK": "1", "PATHED_WAVE_STRAGGLERS": value,
                                                        "PATHED_WAVE_SHADE_READY": ready}
    for name in args.scenes.split(","):
        rates = {k: [] for k in variants}
        digests = {k: set() for k in variants}
        means = {}
        for _ in range(args.repeats):
            for k, extra in variants.items():
                env = dict(os.environ, **extra)
                run = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", name, "--spp", str(args.spp)],
                                     env=env, capture_output=True, text=True, timeout=900)
                if run.returncode != 0:
                    print("%s %s FAILED: %s" % (name, k, run.stderr[-600:]), flush=True)
                    continue
                result = json.loads(run.stdout.strip().splitlines()[-1])
                rates[k].append(result["rate"])
                digests[k].add(result["digest"])
                means[k] = result["mean"]
        base = max(rates["wavefront"]) if rates["wavefront"] else 1.0
        reference = digests["wavefront"]
        print("%s %s %dx%d x %d spp, Msamples/s: %s" % (
            name, SCENES[name][0], SCENES[name][1], SCENES[name][2], args.spp,
            "  ".join("%s %s (%+.1f %%, identical %s, mean %.6f)" % (k, " ".join("%.0f" % r for r in v), 100.0 * (max(v) / base - 1.0) if v else 0.0,
                                                                      digests[k] == reference, means.get(k, 0.0)) for k, v in rates.items())), flush=True)


if __name__ == "__main__":
    main()
