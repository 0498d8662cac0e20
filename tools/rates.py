#!/usr/bin/env python3
"""Msamples/s of a list of scenes through the C ABI, one process, interleaved repeats; variants are PathedSceneOptions
(the product library reads no environment variable), images of the variants of a scene compared bit for bit.

    tools/rates.py [--spp 256] [--repeats 3] [--scenes C2,ON,GGX,GL,GLASS,C3,C4,C5,VOL] [--variants default,generic]
                   [--lib other.so]   # the second column from another build of the library (a child process per library)

Variants: name or name=opt:value+opt:value, e.g. "wave=shade_kernel:wave", "front=shade_kernel:per-slot".
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SCENES = {
    "C1": ("scenes/cornell.json", 256, 256, "PathTracer"),
    "C2": ("scenes/cornell.json", 1024, 1024, "PathTracer"),
    "C3": ("scenes/mis-pbrt.json", 1024, 1024, "PathTracer"),
    "C4": ("scenes/teapot.json", 1024, 1024, "PathTracer"),
    "C5": ("scenes/dragon-standin.json", 1920, 1080, "PathTracer"),
    "C5close": ("scenes/dragon-standin-close.json", 1920, 1080, "PathTracer"),
    "ON": ("scenes/cornell-oren-nayar.json", 1024, 1024, "PathTracer"),
    "GGX": ("scenes/cornell-ggx.json", 1024, 1024, "PathTracer"),
    "GL": ("scenes/cornell-glossy.json", 1024, 1024, "PathTracer"),
    "GLASS": ("scenes/cornell-glass.json", 1024, 1024, "PathTracer"),
    "VOL": ("scenes/cornell-medium.json", 1024, 1024, "VolumePathTracer"),
}
PRESETS = {
    "default": {},
    "generic": {"generic_kernels": 1},
    "wave": {"shade_kernel": "wave"},
    "front": {"shade_kernel": "per-slot"},
    "bvh": {"intersector": "bvh"},
}


def parse_variant(text):
    name, _, rest = text.partition("=")
    options = dict(PRESETS.get(name, {}))
    if rest:
        for item in rest.split("+"):
            key, _, value = item.partition(":")
            options[key] = int(value) if value.lstrip("-").isdigit() else value
    return name, options


def run(args):
    import torch
    from pathed_amd.integrator import HipScene
    from pathed_amd.scene import LoadedScene
    rows = []
    for key in args.scenes.split(","):
        path, w, h, integrator = SCENES[key]
        scene = LoadedScene(path, w, h)
        builder = args.builder or ("ploc" if scene.n_triangles > 1000000 else "sah")
        variants = {}
        for text in args.variants.split(","):
            name, options = parse_variant(text)
            try:
                variants[name] = HipScene(scene.desc, device=0, bvh_builder=builder, **options)
                variants[name].set_integrator(integrator)
            except Exception as error:   # a variant that does not apply to this scene
                print("%s %s: %s" % (key, name, error), flush=True)
        accum = {k: torch.zeros((h, w, 3), dtype=torch.float32, device="cuda") for k in variants}
        for k, gpu in variants.items():
            gpu.render_device(1, 0, min(args.spp, 16), 0, 10, accum[k].data_ptr())
        rates = {k: [] for k in variants}
        for _ in range(args.repeats):
            for k, gpu in variants.items():
                accum[k].zero_()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                gpu.render_device(1, 0, args.spp, 0, 10, accum[k].data_ptr())
                torch.cuda.synchronize()
                rates[k].append(w * h * args.spp / (time.perf_counter() - t0) / 1e6)
        names = list(variants)
        same = all(bool(torch.equal(accum[names[0]], accum[k])) for k in names[1:])
        row = {"scene": key, "path": path, "res": "%dx%d" % (w, h), "spp": args.spp, "triangles": scene.n_triangles,
               "rates": {k: [round(r, 1) for r in rates[k]] for k in names}, "best": {k: round(max(rates[k]), 1) for k in names},
               "path_kernel": {k: variants[k].stats()["path_kernel"] for k in names}, "images_identical": same if len(names) > 1 else None,
               "mean_rgb": [round(float(v), 5) for v in (accum[names[0]] / args.spp).mean(dim=(0, 1)).tolist()],
               "checksum": float(accum[names[0]].double().sum().item())}
        rows.append(row)
        print(json.dumps(row), flush=True)
        for gpu in variants.values():
            gpu.close()
    return rows


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--spp", type=int, default=256)
    parser.add_argument("--repeats", type=int, default=3)
    parser.add_argument("--scenes", default="C2,ON,GGX,GL,GLASS,C3")
    parser.add_argument("--variants", default="default")
    parser.add_argument("--builder", default="", help="sah / lbvh / ploc (default: ploc beyond a million triangles, else sah)")
    parser.add_argument("--lib", default="", help="also run with PATHED_HIP_LIB=<this library> in a child process and print both")
    args = parser.parse_args()
    if args.lib:
        base = [sys.executable, os.path.abspath(__file__), "--spp", str(args.spp), "--repeats", str(args.repeats), "--scenes", args.scenes, "--variants", args.variants] + (["--builder", args.builder] if args.builder else [])
        for label, env in (("this", dict(os.environ)), (args.lib, dict(os.environ, PATHED_HIP_LIB=args.lib))):
            print("== library: %s" % label, flush=True)
            subprocess.run(base, env=env, check=False)
        return 0
    run(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
