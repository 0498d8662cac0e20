#!/usr/bin/env python3
"""Five refits of the 5.2 M-triangle stand-in (pathed_hip_scene_refit): prints the device time of each; under
rocprofv3 --kernel-trace --stats the per-kernel split (k_build_tri_shade, k_refit_pass)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def main():
    level = int(sys.argv[1]) if len(sys.argv) > 1 else 9
    from pathed_amd.integrator import HipScene
    from pathed_amd.scene import LoadedScene
    from test_gpu_refit import _deform, _positions
    scene = LoadedScene("assets/dragon-standin-%d.json" % level, 64, 36)
    gpu = HipScene(scene.desc, device=0, bvh_builder="ploc", refittable=1)
    original = _positions(scene).copy()
    times = [gpu.refit(_deform(original, 1.0 + 0.1 * k)) for k in range(6)]
    stats = gpu.stats()
    print("refit of %d triangles, %d nodes, depth %d: %s ms" % (scene.n_triangles, stats["bvh_nodes"], stats["bvh_max_depth"],
                                                                 " ".join("%.3f" % t for t in times)))


if __name__ == "__main__":
    if "--generate" in sys.argv:   # a child process, before any GPU call
        sys.argv.remove("--generate")
        level = sys.argv[1] if len(sys.argv) > 1 else "9"
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_assets.py"), "--dragon-variants", level], check=True)
    main()
