import sys, json, time, torch
sys.path.insert(0, ".")
from pathed_amd.integrator import HipScene
from pathed_amd.scene import LoadedScene
scene = LoadedScene("scenes/teapot.json", 1024, 1024)
for name, opts in (("chain", {}), ("nochain", {"shade_chain": 1})):
    gpu = HipScene(scene.desc, device=0, bvh_builder="sah", shade_kernel="per-slot", **opts)
    img = torch.zeros((1024, 1024, 3), dtype=torch.float32, device="cuda")
    gpu.render_device(1, 0, 16, 0, 10, img.data_ptr()); torch.cuda.synchronize()
    gpu.reset_stats()
    t0 = time.perf_counter(); gpu.render_device(1, 0, 64, 0, 10, img.data_ptr()); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s = gpu.stats(); print(name, "plain", round(dt*1e3,1), "ms", {k: s[k] for k in ("iterations","path_kernel")}, flush=True)
    gpu.set_stats_mode(count=True); gpu.reset_stats()
    t0 = time.perf_counter(); gpu.render_device(1, 0, 64, 0, 10, img.data_ptr()); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s = gpu.stats(); print(name, "counted", round(dt*1e3,1), "ms", json.dumps(s), flush=True)
