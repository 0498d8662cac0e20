import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from pathed_amd.integrator import HipScene
from pathed_amd.scene import LoadedScene
scene = LoadedScene("scenes/cornell-medium.json", 1024, 1024)
v = {}
os.environ["PATHED_NO_QUADS"] = "1"; v["pairs"] = HipScene(scene.desc, device=0); del os.environ["PATHED_NO_QUADS"]
v["items"] = HipScene(scene.desc, device=0)
acc = {k: torch.zeros((1024, 1024, 3), dtype=torch.float32, device="cuda") for k in v}
for k, g in v.items():
    g.set_integrator("VolumePathTracer"); g.render_device(1, 0, 8, 0, 10, acc[k].data_ptr()); acc[k].zero_()
rates = {k: [] for k in v}
for r in range(3):
    for k, g in v.items():
        acc[k].zero_(); torch.cuda.synchronize(); t = time.perf_counter()
        g.render_device(1, 0, 128, 0, 10, acc[k].data_ptr()); torch.cuda.synchronize()
        rates[k].append(1024 * 1024 * 128 / (time.perf_counter() - t) / 1e6)
print("VOL cornell-medium: pairs", rates["pairs"], "items", rates["items"], "identical", bool(torch.equal(acc["pairs"], acc["items"])))
