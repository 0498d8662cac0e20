import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from pathed_amd.scene import LoadedScene
from pathed_amd.integrator import HipScene
for path, w, h, spp in (("scenes/cornell.json", 1024, 1024, 2048), ("scenes/dragon-standin.json", 1920, 1080, 2048)):
    scene = LoadedScene(path, w, h)
    gpu = HipScene(scene.desc, device=0)
    accum = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    for chunks in (64, 256, 512, 64, 256):
        os.environ["PATHED_CHUNKS_PER_PASS"] = str(chunks)
        gpu.render_device(1, 0, 64, 0, 10, accum.data_ptr())
        torch.cuda.synchronize(); t = time.perf_counter()
        gpu.render_device(1, 0, spp, 0, 10, accum.data_ptr())
        torch.cuda.synchronize(); e = time.perf_counter() - t
        print("%s %d spp, %4d chunks (%d spp) per pass: %.1f ms = %.1f Msamples/s" % (path, spp, chunks, 4 * chunks, e * 1e3, w * h * spp / e / 1e6), flush=True)
