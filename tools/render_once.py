"""One render call of one scene, for profiling runs (rocprofv3 ... -- python3 tools/render_once.py ...):
    python tools/render_once.py --scene scenes/cornell-medium.json --integrator VolumePathTracer --spp 64 [--width 1024 --height 1024]
Prints one line: samples, seconds, Msamples/s."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import torch
from pathed_amd.scene import LoadedScene
from pathed_amd.integrator import HipScene
parser = argparse.ArgumentParser()
parser.add_argument("--scene", default="scenes/cornell-medium.json")
parser.add_argument("--integrator", default="PathTracer")
parser.add_argument("--spp", type=int, default=64)
parser.add_argument("--width", type=int, default=1024)
parser.add_argument("--height", type=int, default=1024)
parser.add_argument("--last-bounce", type=int, default=10)
parser.add_argument("--options", default="", help="PathedSceneOptions as opt:value+opt:value")
args = parser.parse_args()
scene = LoadedScene(args.scene, args.width, args.height)
options = {}
for item in filter(None, args.options.split("+")):
    key, _, value = item.partition(":")
    options[key] = int(value) if value.lstrip("-").isdigit() else value
gpu = HipScene(scene.desc, device=0, **options)
gpu.set_integrator(args.integrator)
accum = torch.zeros((args.height, args.width, 3), dtype=torch.float32, device="cuda")
torch.cuda.synchronize(); start = time.perf_counter()
gpu.render_device(1, 0, args.spp, 0, args.last_bounce, accum.data_ptr())
torch.cuda.synchronize(); elapsed = time.perf_counter() - start
print("%s %s %dx%d x %d spp: %d samples in %.3f s = %.1f Msamples/s" % (
    args.scene, args.integrator, args.width, args.height, args.spp, args.width * args.height * args.spp, elapsed,
    args.width * args.height * args.spp / elapsed / 1e6))
