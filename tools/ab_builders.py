import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from pathed_amd.scene import LoadedScene
from pathed_amd.integrator import HipScene
for path, w, h, spp in (("scenes/dragon-standin.json", 1920, 1080, 256), ("scenes/teapot.json", 1024, 1024, 512)):
    scene = LoadedScene(path, w, h)
    gpus = {b: HipScene(scene.desc, device=0, bvh_builder=b) for b in ("sah", "ploc")}
    accum = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    for b in gpus: gpus[b].render_device(1, 0, 32, 0, 10, accum.data_ptr())
    times = {b: [] for b in gpus}
    for rep in range(4):
        for b in ("sah", "ploc"):
            torch.cuda.synchronize(); t = time.perf_counter()
            gpus[b].render_device(1, 0, spp, 0, 10, accum.data_ptr())
            torch.cuda.synchronize(); times[b].append(time.perf_counter() - t)
    for b in gpus:
        best = min(times[b]); print("%s %s: best %.1f ms = %.1f Msamples/s (all: %s)" % (path, b, best * 1e3, w * h * spp / best / 1e6, ["%.1f" % (t * 1e3) for t in times[b]]), flush=True)
