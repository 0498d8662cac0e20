import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from pathed_amd.integrator import HipScene
from pathed_amd.scene import LoadedScene
scene = LoadedScene("scenes/cornell.json", 1024, 1024)
gpu = HipScene(scene.desc, device=0)
accum = torch.zeros((1024, 1024, 3), dtype=torch.float32, device="cuda")
for chunk in (4, 2, 8, 16, 32, 4):
    gpu.set_samples_per_unit(chunk)
    gpu.render_device(1, 0, 64, 0, 10, accum.data_ptr())
    best = None
    for rep in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        gpu.render_device(1, 0, 256, 0, 10, accum.data_ptr())
        torch.cuda.synchronize(); e = time.perf_counter() - t
        best = e if best is None else min(best, e)
    print("samples per unit %2d: %.1f ms = %.1f Msamples/s" % (chunk, best * 1e3, 1024 * 1024 * 256 / best / 1e6), flush=True)
