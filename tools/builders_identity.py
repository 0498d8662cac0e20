"""At scale: the 5.2 M-triangle stand-in rendered at 1920x1080 over the host SAH tree and the two device-built
trees must give bit-identical radiance sums (hits do not depend on the tree)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pathed_amd.scene import LoadedScene
from pathed_amd.integrator import HipScene
spp = int(os.environ.get("IDENTITY_SPP", "64"))
scene = LoadedScene("scenes/dragon-standin.json", 1920, 1080)
images = {}
for builder in ("sah", "ploc", "lbvh"):
    gpu = HipScene(scene.desc, device=0, bvh_builder=builder)
    accum = torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize(); t = time.time()
    gpu.render_device(1, 0, spp, 0, 10, accum.data_ptr())
    torch.cuda.synchronize()
    images[builder] = accum
    print("%-5s %d triangles, %d spp: %.3f s, mean %s" % (builder, scene.n_triangles, spp, time.time() - t, [round(v, 5) for v in (accum / spp).mean(dim=(0, 1)).tolist()]), flush=True)
    del gpu
for builder in ("ploc", "lbvh"):
    same = torch.equal(images["sah"], images[builder])
    print("sah vs %s: %s" % (builder, "bit-identical (%d floats)" % images["sah"].numel() if same else "DIFFERENT"))
    assert same
