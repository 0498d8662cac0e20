"""Slots of the wavefront's render state (PathedSceneOptions.max_slots, split over the two pools) x samples per call."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pathed_amd.integrator import HipScene
from pathed_amd.scene import LoadedScene
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_assets.py"), "--dragon", "9"], check=True, stdout=subprocess.DEVNULL)
sizes = [int(v) for v in os.environ.get("SLOTS", "4194304,8388608,16777216,33554432,4194304").split(",")]
for path, w, h in (("scenes/teapot.json", 1024, 1024), ("scenes/dragon-standin.json", 1920, 1080), ("scenes/dragon-standin-close.json", 1920, 1080)):
    scene = LoadedScene(path, w, h)
    accum = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    for slots in sizes:
        gpu = HipScene(scene.desc, device=0, max_slots=slots)
        gpu.render_device(1, 0, 256, 0, 10, accum.data_ptr())
        rates = []
        for spp in (16, 64, 256, 1024):
            best = None
            for rep in range(2 if spp == 1024 else 3):
                torch.cuda.synchronize(); t = time.perf_counter()
                gpu.render_device(1, 0, spp, 0, 10, accum.data_ptr())
                torch.cuda.synchronize(); e = time.perf_counter() - t
                best = e if best is None else min(best, e)
            rates.append("%d spp: %.1f" % (spp, w * h * spp / best / 1e6))
        print("%-34s %9d slots: %s Msamples/s" % (path, slots, "  ".join(rates)), flush=True)
        gpu.close()
