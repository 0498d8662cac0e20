"""A/B of the shade kernels inside ONE process / one gpurun call (boxes of the pool differ by +-5 %):
    python tools/ab_shade.py [--scenes cornell,teapot,dragon] [--variants per-slot,staged:512,staged:1024] [--spp N]
Per scene and variant: best-of-3 Msamples/s untimed, then one pass with sampled HIP-event timing for the
average trace / shade launch duration.  PATHED_HIP_LIB picks another build of the library."""
import argparse, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pathed_amd.integrator import HipScene
from pathed_amd.scene import LoadedScene

SCENES = {
    "cornell": ("scenes/cornell.json", 1024, 1024, 256),
    "mis": ("scenes/mis-pbrt.json", 1024, 1024, 256),
    "teapot": ("scenes/teapot.json", 1024, 1024, 256),
    "dragon": ("scenes/dragon-standin.json", 1920, 1080, 128),
}
parser = argparse.ArgumentParser()
parser.add_argument("--scenes", default="cornell,teapot,dragon")
parser.add_argument("--variants", default="per-slot,staged:1024,fused")
parser.add_argument("--spp", type=int, default=0)
parser.add_argument("--pools", type=int, default=0)
parser.add_argument("--opt", action="append", default=[], help="extra scene option name=integer, e.g. suspend_lanes=-1")
args = parser.parse_args()

for name in args.scenes.split(","):
    path, w, h, spp = SCENES[name]
    spp = args.spp or spp
    if name == "dragon" and not os.path.exists(os.path.join(ROOT, "assets", "dragon-standin.ply")):
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_assets.py"), "--dragon", "9"], check=True, stdout=subprocess.DEVNULL)
    scene = LoadedScene(path, w, h)
    accum = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    reference = None
    for variant in args.variants.split(","):
        # variant = kernel[:stage slots][/pools][@node format], e.g. staged:1024/1, per-slot@compressed
        variant_spec, _, node_format = variant.partition("@")
        spec, _, pools = variant_spec.partition("/")
        kernel, _, slots = spec.partition(":")
        options = {"shade_kernel": kernel}
        if node_format:
            options["node_format"] = node_format
        if slots:
            options["stage_slots"] = int(slots)
        if pools or args.pools:
            options["pools"] = int(pools or args.pools)
        for extra in args.opt:
            key, _, value = extra.partition("=")
            options[key] = int(value)
        try:
            gpu = HipScene(scene.desc, device=0, **options)
        except Exception as error:
            print("%-8s %-12s not available: %s" % (name, variant, str(error)[-80:]))
            continue
        gpu.render_device(1, 0, 16, 0, 10, accum.data_ptr())
        best = None
        for rep in range(3):
            accum.zero_()
            gpu.set_stats_mode(count=False); gpu.reset_stats()
            torch.cuda.synchronize(); t = time.perf_counter()
            gpu.render_device(1, 0, spp, 0, 10, accum.data_ptr())
            torch.cuda.synchronize(); e = time.perf_counter() - t
            best = e if best is None else min(best, e)
        image = accum.clone()
        if reference is None:
            reference = image
        identical = bool(torch.equal(image, reference))
        gpu.set_stats_mode(count=False, time_sampled=True); gpu.reset_stats()
        gpu.render_device(1, 0, spp, 0, 10, accum.data_ptr())
        torch.cuda.synchronize()
        s = gpu.stats()
        n = max(s["trace_launches"], 1)
        print("%-8s %-20s %7.1f Msamples/s  trace %6.1f us  shade %6.1f us per launch (%d launches)  image identical to first variant: %s" % (
            name, variant, w * h * spp / best / 1e6, s["trace_ms"] / n * 1e3, s["shade_ms"] / n * 1e3, s["trace_launches_all"], identical), flush=True)
        gpu.close()
