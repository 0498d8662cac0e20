import os, subprocess, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch
from pathed_amd.integrator import HipScene
from pathed_amd.scene import LoadedScene
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_assets.py"), "--dragon", "9"], check=True, stdout=subprocess.DEVNULL)
w, h = 1920, 1080
scene = LoadedScene("scenes/dragon-standin.json", w, h)
gpu = HipScene(scene.desc, device=0)
accum = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
gpu.render_device(1, 0, 32, 0, 10, accum.data_ptr())
for spp in (256, 512, 517, 1024, 1034, 2048, 512, 1024):
    times = []
    for rep in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        gpu.render_device(1, 0, spp, 0, 10, accum.data_ptr())
        torch.cuda.synchronize(); times.append(time.perf_counter() - t)
    print("spp %5d: %s ms -> best %.1f Msamples/s" % (spp, " ".join("%.1f" % (t * 1e3) for t in times), w * h * spp / min(times) / 1e6), flush=True)
