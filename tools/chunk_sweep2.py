"""Samples per unit (pathed_hip_set_samples_per_unit) x samples per render call, Cornell (fused) and the 5.2 M-triangle
mesh (wavefront): rate of the best of 3 calls.  A unit is the grain of the work queue: the last units of a call are
what its drain waits for."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pathed_amd.integrator import HipScene
from pathed_amd.scene import LoadedScene
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_assets.py"), "--dragon", "9"], check=True, stdout=subprocess.DEVNULL)
for path, w, h in (("scenes/cornell.json", 1024, 1024), ("scenes/teapot.json", 1024, 1024), ("scenes/dragon-standin.json", 1920, 1080)):
    scene = LoadedScene(path, w, h)
    gpu = HipScene(scene.desc, device=0)
    accum = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    for chunk in [int(v) for v in os.environ.get("CHUNKS", "4,2,1,4").split(",")]:
        gpu.set_samples_per_unit(chunk)
        gpu.render_device(1, 0, 32, 0, 10, accum.data_ptr())
        rates = []
        for spp in (64, 256, 1024):
            best = None
            for rep in range(2 if spp == 1024 else 3):
                torch.cuda.synchronize(); t = time.perf_counter()
                gpu.render_device(1, 0, spp, 0, 10, accum.data_ptr())
                torch.cuda.synchronize(); e = time.perf_counter() - t
                best = e if best is None else min(best, e)
            rates.append("%d spp: %.1f" % (spp, w * h * spp / best / 1e6))
        print("%-28s samples per unit %d: %s Msamples/s" % (path, chunk, "  ".join(rates)), flush=True)
