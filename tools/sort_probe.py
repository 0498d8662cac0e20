#!/usr/bin/env python3
"""What sorting the ray queue would buy the PRODUCTION trace kernel (experiments build: pathed_hip.hip sortProbe).
At iteration PATHED_SORT_PROBE of a wavefront render the library copies pool 0's rays and times k_trace over them in slot
order and sorted by (direction octant, Morton code of the origin); it prints to stderr.

    PATHED_HIP_LIB=pathed_amd/lib/libpathed_hip_experiments.so PATHED_SORT_PROBE=24 python tools/sort_probe.py [C4,C5,C5close]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import torch
from pathed_amd.integrator import HipScene
from pathed_amd.scene import LoadedScene
SCENES = {"C4": ("scenes/teapot.json", 1024, 1024, 256), "C5": ("scenes/dragon-standin.json", 1920, 1080, 256), "C5close": ("scenes/dragon-standin-close.json", 1920, 1080, 256)}
for key in (sys.argv[1] if len(sys.argv) > 1 else "C4,C5,C5close").split(","):
    path, w, h, spp = SCENES[key]
    scene = LoadedScene(path, w, h)
    gpu = HipScene(scene.desc, device=0, bvh_builder="sah", shade_kernel="per-slot")
    accum = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    print("== %s %dx%d x %d spp, %d triangles" % (key, w, h, spp, scene.n_triangles), file=sys.stderr, flush=True)
    gpu.render_device(1, 0, spp, 0, 10, accum.data_ptr())
    torch.cuda.synchronize()
    gpu.close()
