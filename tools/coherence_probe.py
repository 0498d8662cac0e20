"""How much would sorting the ray queue buy on the large-BVH scene?  Traces the same 4 M secondary-like
rays (origins on the mesh, random directions) through pathed_hip_trace three ways: in slot order (what the
wavefront produces), sorted by (direction octant, Morton code of the origin), and sorted by origin only.
Run under rocprofv3 --kernel-trace: the k_trace_rays launches appear in that order (after one warm-up)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pathed_amd.scene import LoadedScene
from pathed_amd.integrator import HipScene

n = int(os.environ.get("PROBE_RAYS", str(1 << 22)))
scene = LoadedScene("scenes/dragon-standin.json", 2048, 2048)
gpu = HipScene(scene.desc, device=0)
camera = scene.desc.contents.camera
rng = np.random.default_rng(1)
origin = np.array(list(camera.origin), dtype=np.float32)
target = np.array(list(camera.target), dtype=np.float32)
forward = (target - origin) / np.linalg.norm(target - origin)
primary = np.zeros((n, 8), dtype=np.float32)
primary[:, 0:3] = origin
jitter = rng.normal(size=(n, 3)).astype(np.float32) * 0.25
direction = forward + jitter - (jitter @ forward)[:, None] * forward
primary[:, 4:7] = direction / np.linalg.norm(direction, axis=1, keepdims=True)
primary[:, 3] = 1e-3
primary[:, 7] = 1e5
hits = gpu.trace(primary)                     # warm-up launch, and the surface points
hit = hits[:, 3].view(np.int32) >= 0
points = primary[hit, 0:3] + primary[hit, 4:7] * hits[hit, 0:1]
m = points.shape[0]
rays = np.zeros((m, 8), dtype=np.float32)
rays[:, 0:3] = points
d = rng.normal(size=(m, 3)).astype(np.float32)
rays[:, 4:7] = d / np.linalg.norm(d, axis=1, keepdims=True)
rays[:, 3] = 1e-3 * max(1.0, float(np.abs(points).max()) * 1e-3)
rays[:, 7] = 1e5
shuffled = rays[rng.permutation(m)]           # slot order of a wavefront in steady state: no spatial order


def morton(p):
    lo, hi = p.min(axis=0), p.max(axis=0)
    q = np.clip(((p - lo) / np.maximum(hi - lo, 1e-30) * 1023.0), 0, 1023).astype(np.uint64)
    code = np.zeros(p.shape[0], dtype=np.uint64)
    for bit in range(10):
        for axis in range(3):
            code |= ((q[:, axis] >> np.uint64(bit)) & np.uint64(1)) << np.uint64(3 * bit + (2 - axis))
    return code


octant = ((shuffled[:, 4] < 0).astype(np.uint64) << np.uint64(2)) | ((shuffled[:, 5] < 0).astype(np.uint64) << np.uint64(1)) | (shuffled[:, 6] < 0).astype(np.uint64)
code = morton(shuffled[:, 0:3])
orders = {
    "slot order (unsorted)": np.arange(m),
    "sorted by octant, then origin Morton code": np.argsort((octant << np.uint64(30)) | code, kind="stable"),
    "sorted by origin Morton code": np.argsort(code, kind="stable"),
}
reference = None
for name, order in orders.items():
    batch = np.ascontiguousarray(shuffled[order])
    t = time.time()
    result = gpu.trace(batch)
    elapsed = time.time() - t
    restored = np.zeros_like(result)
    restored[order] = result
    if reference is None:
        reference = restored
    assert np.array_equal(restored.view(np.int32), reference.view(np.int32))
    print("%-45s %d rays, hit fraction %.3f (host round trip %.3f s)" % (name, m, (result[:, 3].view(np.int32) >= 0).mean(), elapsed))
