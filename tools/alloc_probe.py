"""How long a large device allocation takes here (hipMalloc through ctypes and through torch), and what the first
large render call of a scene pays for its partial-sum buffer."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
hip = ctypes.CDLL("libamdhip64.so")
for gib in (4, 17, 34):
    pointer = ctypes.c_void_p()
    t = time.perf_counter(); code = hip.hipMalloc(ctypes.byref(pointer), ctypes.c_size_t(gib << 30)); a = time.perf_counter() - t
    t = time.perf_counter(); hip.hipMemset(pointer, 0, ctypes.c_size_t(gib << 30)); hip.hipDeviceSynchronize(); b = time.perf_counter() - t
    t = time.perf_counter(); hip.hipFree(pointer); c = time.perf_counter() - t
    print("hipMalloc %2d GiB: code %d, %.3f s; memset %.3f s; hipFree %.3f s" % (gib, code, a, b, c), flush=True)
from pathed_amd.integrator import HipScene
from pathed_amd.scene import LoadedScene
scene = LoadedScene("scenes/cornell.json", 1024, 1024)
gpu = HipScene(scene.desc, device=0)
accum = torch.zeros((1024, 1024, 3), dtype=torch.float32, device="cuda")
gpu.render_device(1, 0, 16, 0, 10, accum.data_ptr())
for spp in (256, 256, 1024, 1024):
    torch.cuda.synchronize(); t = time.perf_counter()
    gpu.render_device(1, 0, spp, 0, 10, accum.data_ptr())
    torch.cuda.synchronize(); print("cornell 1024^2 x %d spp call: %.3f s" % (spp, time.perf_counter() - t), flush=True)
