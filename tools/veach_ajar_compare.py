"""scenes/veach-ajar-available.json (the reference's veach-ajar scene without the two meshes its repository lacks) against the
Tungsten render the reference ships (tests/golden/veach_ajar_tungsten_blocks.npz: 16 x 16 block means).

    python tools/veach_ajar_compare.py [--spp N] [--last-bounce B] [--write-mask]

Prints the energy ratio and the distribution of per-block relative differences; --write-mask stores the blocks that differ by
more than a factor 1.5 (after a one-block dilation) as tests/golden/veach_ajar_mask.npz: the teapots the scene lacks."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pathed_amd.gt_metrics import veach_ajar_blocks, veach_ajar_compare  # noqa: E402

parser = argparse.ArgumentParser()
parser.add_argument("--spp", type=int, default=2048)
parser.add_argument("--last-bounce", type=int, default=12)
parser.add_argument("--write-mask", action="store_true")
args = parser.parse_args()

from pathed_amd.integrator import HipScene
from pathed_amd.scene import LoadedScene
scene = LoadedScene("scenes/veach-ajar-available.json", 1280, 720)
gpu = HipScene(scene.desc, device=0)
import time
t = time.perf_counter()
image = gpu.render(1, 0, args.spp, 0, args.last_bounce) / float(args.spp)
elapsed = time.perf_counter() - t
print("rendered 1280x720 x %d spp, lastBounce %d in %.1f s (%.0f Msamples/s), dropped %d" % (
    args.spp, args.last_bounce, elapsed, 1280 * 720 * args.spp / elapsed / 1e6, gpu.stats()["dropped_samples"]))
ours = veach_ajar_blocks(image)
fixture = np.load(os.path.join(ROOT, "tests", "golden", "veach_ajar_tungsten_blocks.npz"))
theirs = fixture["blocks"]
mask_path = os.path.join(ROOT, "tests", "golden", "veach_ajar_mask.npz")
mask = np.load(mask_path)["mask"] if os.path.exists(mask_path) and not args.write_mask else np.zeros(theirs.shape[:2], dtype=bool)
report = veach_ajar_compare(ours, theirs, mask)
for key, value in report.items():
    print("%-28s %s" % (key, value))
ratio = (ours.sum(axis=2) + 1e-4) / (theirs.sum(axis=2) + 1e-4)
for row in ratio[::2]:
    print(" ".join("%4.1f" % min(v, 99.9) for v in row[::2]))
if args.write_mask:
    off = np.abs(np.log(ratio)) > np.log(1.5)
    grown = off.copy()
    grown[1:] |= off[:-1]; grown[:-1] |= off[1:]; grown[:, 1:] |= off[:, :-1]; grown[:, :-1] |= off[:, 1:]
    np.savez_compressed(mask_path, mask=grown)
    print("mask: %d of %d blocks" % (grown.sum(), grown.size))
