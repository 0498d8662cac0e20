import sys, torch
sys.path.insert(0, ".")
from pathed_amd.integrator import HipScene
from pathed_amd.scene import LoadedScene
scene = LoadedScene("scenes/teapot.json", 1024, 1024)
opts = {"shade_chain": int(sys.argv[1])}
gpu = HipScene(scene.desc, device=0, bvh_builder="sah", shade_kernel="per-slot", **opts)
img = torch.zeros((1024, 1024, 3), dtype=torch.float32, device="cuda")
gpu.render_device(1, 0, 64, 0, 10, img.data_ptr()); torch.cuda.synchronize()
