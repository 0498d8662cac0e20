#!/usr/bin/env python3
"""Box and triangle tests per ray of the three node formats (count mode), teapot and the stand-in mesh:
    python tools/node_format_counts.py [--dragon 9]"""
import argparse, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
parser = argparse.ArgumentParser()
parser.add_argument("--dragon", type=int, default=9)
args = parser.parse_args()
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_assets.py"), "--dragon", str(args.dragon)], check=True, stdout=subprocess.DEVNULL)
from pathed_amd.integrator import HipScene
from pathed_amd.scene import LoadedScene
for path, w, h in (("scenes/teapot.json", 1024, 1024), ("scenes/dragon-standin.json", 1920, 1080)):
    scene = LoadedScene(path, w, h)
    for node_format in ("wide", "compressed", "compressed8"):
        gpu = HipScene(scene.desc, device=0, node_format=node_format)
        gpu.set_stats_mode(count=True)
        gpu.render(1, 0, 16, 0, 10)
        s = gpu.stats()
        rays = s["closest_rays"] + s["shadow_rays"]
        print("%-28s %-12s boxes/ray %6.2f  triangles/ray %5.2f  max boxes of one ray %5d  rays %d" % (
            path, node_format, s["nodes_visited"] / rays, s["tris_tested"] / rays, s["max_boxes_per_ray"], rays), flush=True)
        gpu.close()
