"""Minimal large-BVH render for profiling: scenes/dragon-standin.json 1920x1080, 8 spp."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pathed_amd.scene import LoadedScene
from pathed_amd.integrator import HipScene
spp = int(os.environ.get("DRAGON_SPP", "8"))
scene = LoadedScene("scenes/dragon-standin.json", 1920, 1080)
gpu = HipScene(scene.desc, device=0)
accum = torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda")
gpu.set_stats_mode(time_kernels=True)
torch.cuda.synchronize(); t = time.time()
gpu.render_device(1, 0, spp, 0, 10, accum.data_ptr())
torch.cuda.synchronize(); dt = time.time() - t
s = gpu.stats()
print("dragon render %d spp: %.3fs %.1f Msamples/s trace %.1f ms / %d launches shade %.1f ms" % (spp, dt, 1920 * 1080 * spp / dt / 1e6, s["trace_ms"], s["trace_launches"], s["shade_ms"]))
