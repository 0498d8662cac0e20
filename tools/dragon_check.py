"""Large-BVH stand-in (config 5): build, parity spot-check and timing on a GPU box."""
import argparse, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

parser = argparse.ArgumentParser()
parser.add_argument("--subdiv", type=int, default=9)
parser.add_argument("--width", type=int, default=1920)
parser.add_argument("--height", type=int, default=1080)
parser.add_argument("--spp", type=int, default=16)
parser.add_argument("--no-oracle", action="store_true")
parser.add_argument("--builder", default="sah", choices=["sah", "lbvh", "ploc"], help="host binned-SAH build or the on-GPU LBVH build")
args = parser.parse_args()

t = time.time()
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_assets.py"), "--dragon", str(args.subdiv)], check=True)
print("generate %.1fs" % (time.time() - t))

from pathed_amd.scene import LoadedScene
from pathed_amd.integrator import HipScene
import oracle_lib

t = time.time(); scene = LoadedScene("scenes/dragon-standin.json", args.width, args.height); print("load %.1fs tris=%d" % (time.time() - t, scene.n_triangles))
t = time.time(); gpu = HipScene(scene.desc, device=0, bvh_builder=args.builder); print("scene_create (%s BVH build + upload) %.2fs, of which build %.1f ms" % (args.builder, time.time() - t, gpu.stats()["bvh_build_ms"]))
print(gpu.stats())

if not args.no_oracle:
    t = time.time(); cpu = oracle_lib.OracleScene(scene.desc); print("oracle build %.1fs" % (time.time() - t))
    rng = np.random.default_rng(4)
    n = 200000
    rays = np.zeros((n, 8), dtype=np.float32)
    rays[:, 0:3] = rng.normal(size=(n, 3)) * 120 + [0, 0, 25]
    target = rng.normal(size=(n, 3)) * 30 + [0, 0, 25]
    d = target - rays[:, 0:3]; d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays[:, 4:7] = d; rays[:, 3] = 1e-3; rays[:, 7] = 1e5
    t = time.time(); hg = gpu.trace(rays); tg = time.time() - t
    t = time.time(); hc = cpu.trace(rays); tc = time.time() - t
    same = (hg.view(np.int32) == hc.view(np.int32)).all(axis=1)
    print("trace parity: %d rays, hit frac %.3f, bit-identical %.6f (gpu %.2fs cpu %.2fs)" % (n, (hc[:, 3].view(np.int32) >= 0).mean(), same.mean(), tg, tc))
    if not same.all():
        bad = np.where(~same)[0][:5]
        print("  first mismatches:", hg[bad], hc[bad])
    w, h = 160, 90
    small = LoadedScene("scenes/dragon-standin.json", w, h)
    g2, c2 = HipScene(small.desc, device=0), oracle_lib.OracleScene(small.desc)
    ig = g2.render(1, 0, 8, 0, 10); ic, _ = c2.render(w, h, 1, 0, 8, 0, 10, threads=os.cpu_count())
    print("render parity %dx%d 8spp relL2 %.3e mean %s" % (w, h, np.linalg.norm(ig - ic) / np.linalg.norm(ic), (ig / 8).reshape(-1, 3).mean(0)))

import torch
accum = torch.zeros((args.height, args.width, 3), dtype=torch.float32, device="cuda")
gpu.render_device(1, 0, 4, 0, 10, accum.data_ptr())
gpu.set_stats_mode(count=True); gpu.reset_stats()
gpu.render_device(1, 100, args.spp, 0, 10, accum.data_ptr())
counted = gpu.stats()
gpu.set_stats_mode(count=False, time_kernels=True); gpu.reset_stats()
torch.cuda.synchronize(); t = time.time()
gpu.render_device(1, 100, args.spp, 0, 10, accum.data_ptr())
torch.cuda.synchronize(); elapsed = time.time() - t
timed = gpu.stats()
samples = args.width * args.height * args.spp
alg = 48 * counted["closest_rays"] + 36 * counted["shadow_rays"] + 32 * counted["nodes_visited"] + 48 * counted["tris_tested"]
rays = counted["closest_rays"] + counted["shadow_rays"]
print("render %dx%d x %d spp: %.3fs = %.1f Msamples/s; rays/sample %.2f; boxes/ray %.1f tris/ray %.1f" % (
    args.width, args.height, args.spp, elapsed, samples / elapsed / 1e6, rays / samples, counted["nodes_visited"] / rays, counted["tris_tested"] / rays))
print("trace: %.1f ms over %d launches; %.2f Grays/s; algorithmic %.1f GB -> %.0f GB/s (%.1f%% of 8 TB/s); shade %.1f ms" % (
    timed["trace_ms"], timed["trace_launches"], rays / timed["trace_ms"] / 1e6, alg / 1e9, alg / timed["trace_ms"] / 1e6,
    alg / timed["trace_ms"] / 1e6 / 80.0, timed["shade_ms"]))
print("bvh: nodes %d bytes %.1f MB depth %d in_lds %d max_boxes_per_ray %d" % (timed["bvh_nodes"], timed["bvh_bytes"] / 1e6, timed["bvh_max_depth"], timed["scene_in_lds"], counted["max_boxes_per_ray"]))
