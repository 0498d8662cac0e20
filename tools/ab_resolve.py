#!/usr/bin/env python3
"""A/B of k_path_small's phase 2 (kernels.h smallResolveShared): every lane resolves its own candidates (a library built
with -DPATHED_RESOLVE_SHARED=0), the wave shares the left-over candidates out (=1), and the shadow ray's candidates as well
(=2, the product).  One child process per library and repeat, interleaved on one box; images compared by hash.
Build the other two first:
  hipcc $(HIPFLAGS) -DPATHED_RESOLVE_SHARED=0 -shared -o pathed_amd/lib/libpathed_hip_inplace.so pathed_amd/csrc/*.hip   (=1: ..._share1.so)
Usage: ab_resolve.py [--libs name=path,...] [--spp 256] [--repeats 3] [--scenes C2,C3,ON]"""
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
    "C1": ("scenes/cornell.json", 256, 256),
    "C2": ("scenes/cornell.json", 1024, 1024),
    "C3": ("scenes/mis-pbrt.json", 1024, 1024),
    "ON": ("scenes/cornell-oren-nayar.json", 1024, 1024),
    "GL": ("scenes/cornell-glossy.json", 1024, 1024),
    "VOL": ("scenes/cornell-medium.json", 1024, 1024),
}


def child(name, spp):
    import torch
    from pathed_amd.integrator import HipScene
    from pathed_amd.scene import LoadedScene
    path, w, h = SCENES[name]
    scene = LoadedScene(path, w, h)
    gpu = HipScene(scene.desc, device=0)
    accum = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    gpu.render_device(1, 0, 16, 0, 10, accum.data_ptr())
    best = 0.0
    for _ in range(2):
        accum.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gpu.render_device(1, 0, spp, 0, 10, accum.data_ptr())
        torch.cuda.synchronize()
        best = max(best, w * h * spp / (time.perf_counter() - t0) / 1e6)
    digest = hashlib.sha256(accum.cpu().numpy().tobytes()).hexdigest()[:16]
    print(json.dumps({"rate": best, "digest": digest}), flush=True)


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--libs", default="in place=pathed_amd/lib/libpathed_hip_inplace.so,left-overs=pathed_amd/lib/libpathed_hip_share1.so,"
                                          "left-overs + shadow=pathed_amd/lib/libpathed_hip.so", help="name=library,... (the first is the base)")
    parser.add_argument("--spp", type=int, default=256)
    parser.add_argument("--repeats", type=int, default=3)
    parser.add_argument("--scenes", default="C2,C3,ON")
    parser.add_argument("--child", default=None)
    args = parser.parse_args()
    if args.child:
        child(args.child, args.spp)
        return
    libs = {}
    for entry in args.libs.split(","):
        name, path = entry.split("=")
        libs[name.strip()] = os.path.join(ROOT, path.strip())
    for name in args.scenes.split(","):
        rates = {k: [] for k in libs}
        digests = set()
        for _ in range(args.repeats):
            for k, lib in libs.items():
                env = dict(os.environ, PATHED_HIP_LIB=lib)
                out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", name, "--spp", str(args.spp)],
                                     env=env, check=True, capture_output=True, text=True, timeout=600).stdout
                result = json.loads(out.strip().splitlines()[-1])
                rates[k].append(result["rate"])
                digests.add(result["digest"])
        base = max(rates[next(iter(libs))])
        print("%s %s x %d spp, Msamples/s: %s  images identical: %s" % (
            name, SCENES[name][0], args.spp,
            "  ".join("%s %s (best %.0f, %+.1f %%)" % (k, " ".join("%.0f" % r for r in v), max(v), 100.0 * (max(v) / base - 1.0)) for k, v in rates.items()),
            len(digests) == 1), flush=True)


if __name__ == "__main__":
    main()
