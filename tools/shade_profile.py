"""Lane utilisation per region of k_shade (needs a -DPATHED_SHADE_PROFILE build, see tools/README in DESIGN.md):
PATHED_HIP_LIB=.../libpathed_hip_profile.so PATHED_SHADE_PROFILE=1 python tools/shade_profile.py scene w h spp"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pathed_amd.integrator import HipScene
from pathed_amd.scene import LoadedScene
path, w, h, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
scene = LoadedScene(path, w, h)
gpu = HipScene(scene.desc, device=0)
gpu.render(1, 0, 4, 0, 10)
gpu.reset_stats()
gpu.render(1, 4, spp, 0, 10)
print(path, w, h, spp, flush=True)
gpu.stats()
