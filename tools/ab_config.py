"""A/B timing of one scene: python tools/ab_config.py scene w h spp  (library picked by PATHED_HIP_LIB)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pathed_amd.integrator import HipScene
from pathed_amd.scene import LoadedScene

path, w, h, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
scene = LoadedScene(path, w, h)
gpu = HipScene(scene.desc, device=0)
accum = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
gpu.render_device(1, 0, min(spp, 16), 0, 10, accum.data_ptr())
best = None
for rep in range(3):
    gpu.set_stats_mode(count=False, time_kernels=(rep == 2)); gpu.reset_stats()
    torch.cuda.synchronize(); t = time.perf_counter()
    gpu.render_device(1, 0, spp, 0, 10, accum.data_ptr())
    torch.cuda.synchronize(); e = time.perf_counter() - t
    best = e if best is None else min(best, e)
s = gpu.stats()
print("%s %dx%d x %d: best %.1f ms = %.1f Msamples/s; timed pass: trace %.1f ms shade %.1f ms launches %d" % (
    os.environ.get("PATHED_HIP_LIB", "current"), w, h, spp, best * 1e3, w * h * spp / best / 1e6, s["trace_ms"], s["shade_ms"], s["trace_launches"]))
