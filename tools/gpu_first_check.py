"""First-light check on a GPU box: HIP path vs the CPU oracle (used while developing)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from pathed_amd.scene import LoadedScene
from pathed_amd.integrator import HipScene
import oracle_lib

def compare(scene_path, w, h, spp, lb=10):
    s = LoadedScene(scene_path, w, h)
    g = HipScene(s.desc, device=0)
    o = oracle_lib.OracleScene(s.desc)
    rng = np.random.default_rng(1)
    n = 20000
    rays = np.zeros((n, 8), dtype=np.float32)
    rays[:, 0:3] = rng.uniform(-1, 1, (n, 3)) * [1, 1, 1] + [0, 1, 0]
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays[:, 4:7] = d
    rays[:, 3] = 1e-3; rays[:, 7] = 1e5
    hg = g.trace(rays); ho = o.trace(rays)
    pg = hg[:, 3].view(np.int32); po = ho[:, 3].view(np.int32)
    print(scene_path, "trace prim mismatch:", int((pg != po).sum()), "of", n, " t bitwise mismatch:", int((hg[:, 0].view(np.int32) != ho[:, 0].view(np.int32)).sum()),
          "uv mismatch", int((hg[:, 1:3].view(np.int32) != ho[:, 1:3].view(np.int32)).sum()))
    og = g.trace(rays, any_hit=True); oo = o.trace(rays, any_hit=True)
    print("  anyhit mismatch:", int((og != oo).sum()))
    t = time.time(); ig = g.render(1, 0, spp, 0, lb); tg = time.time() - t
    t = time.time(); io, st = o.render(w, h, 1, 0, spp, 0, lb, threads=os.cpu_count()); to = time.time() - t
    diff = np.abs(ig - io)
    rel = np.linalg.norm(ig - io) / max(np.linalg.norm(io), 1e-30)
    bad = (diff > 1e-2 * np.maximum(np.abs(io), 1e-3)).any(axis=2).mean()
    print("  render %dx%d spp=%d: gpu %.3fs cpu %.3fs  relL2=%.3e maxabs=%.3e bitexact_pixels=%.4f badfrac=%.5f mean=%s" % (
        w, h, spp, tg, to, rel, diff.max(), (ig.view(np.int32) == io.view(np.int32)).all(axis=2).mean(), bad, (ig / spp).reshape(-1, 3).mean(0)))
    print("  stats", g.stats())
    return ig, io

if __name__ == "__main__":
    compare("scenes/cornell.json", 64, 64, 4)
    compare("scenes/cornell.json", 256, 256, 16)
    compare("scenes/cornell-glass.json", 128, 128, 8)
    compare("scenes/cornell-glossy.json", 128, 128, 8)
