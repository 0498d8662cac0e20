"""Where k_path_hybrid's waves spend their time (needs a -DPATHED_SHADE_PROFILE build of the library):
PATHED_HIP_LIB=pathed_amd/lib/libpathed_hip_profile.so python tools/hybrid_profile.py [scene] [w h spp]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
from pathed_amd.integrator import HipScene
from pathed_amd.scene import LoadedScene
path = sys.argv[1] if len(sys.argv) > 1 else "scenes/cornell-glossy.json"
w, h, spp = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (1024, 1024, 32)
scene = LoadedScene(path, w, h)
gpu = HipScene(scene.desc, device=0, shade_kernel=os.environ.get("HYBRID_PROFILE_KERNEL", "auto"), bvh_builder="ploc" if scene.n_triangles > 1000000 else "sah")
gpu.render(1, 0, 2, 0, 10)
gpu.reset_stats()
gpu.render(1, 2, spp, 0, 10)
print(path, w, h, spp, "path_kernel", gpu.stats()["path_kernel"], flush=True)
