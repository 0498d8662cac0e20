"""The volume integrator's kernel against the path tracer's kernels on scenes WITHOUT media (same sums, bit for bit):
what the per-lane tree walk of k_path_volume costs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pathed_amd.integrator import HipScene
from pathed_amd.scene import LoadedScene
for path, spp in (("scenes/cornell.json", 256), ("scenes/cornell-glass.json", 128), ("scenes/teapot.json", 128), ("scenes/cornell-medium.json", 64)):
    scene = LoadedScene(path, 1024, 1024)
    accum = torch.zeros((1024, 1024, 3), dtype=torch.float32, device="cuda")
    for name in ("PathTracer", "VolumePathTracer"):
        gpu = HipScene(scene.desc, device=0)
        gpu.set_integrator(name)
        try:
            gpu.render_device(1, 0, 16, 0, 10, accum.data_ptr())
        except Exception as error:
            print("%-28s %-16s %s" % (path, name, str(error)[-70:])); continue
        best = None
        for rep in range(2):
            torch.cuda.synchronize(); t = time.perf_counter()
            gpu.render_device(1, 0, spp, 0, 10, accum.data_ptr())
            torch.cuda.synchronize(); e = time.perf_counter() - t
            best = e if best is None else min(best, e)
        print("%-28s %-16s %.1f Msamples/s" % (path, name, 1024 * 1024 * spp / best / 1e6), flush=True)
