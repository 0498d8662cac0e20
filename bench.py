#!/usr/bin/env python3
"""Headline benchmark: Msamples/s of the GPU PathTracer on scenes/cornell.json, 1024x1024.

A "step" renders `--spp-per-step` camera samples (default 1024: one internal pass of the library) for
every pixel through the C ABI (pathed_hip_render_device) into device-resident radiance sums; 4 steps are
the 4096-spp configuration BASELINE.json quotes.

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU.  Under torch.distributed.run (WORLD_SIZE in the environment) this process IS
a rank; started plainly with --gpus N it spawns the N ranks itself, as children of a parent that never
touches the GPU.  Scaling is STRONG by default (north star: Cornell 1024^2 x 4096 spp split over the
GPUs): the samples of every step are split over the ranks (pathed_amd/parallel.strong_range), each rank
adds into its own sums, and ONE RCCL reduce to rank 0 closes the timed region.  --scaling weak fixes the
per-GPU work instead.

Rank 0 prints ONE JSON line (contract in the task statement) with, besides the headline value:
  roofline      what bounds the pipeline on this workload, with a denominator measured on this box
  cpu_baseline  the CPU oracle (a port of the reference estimator; the reference binary cannot be
                built) timed on the host cores, bounded sample   [N = 1 only]
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

REPO_ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO_ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
VALU_PEAK_GINSTR = 1228.8  # G wave-instr/s: 1 024 SIMDs x 2.4 GHz / 2 cycles per v_fma_f32 (same guide)
PATHED_FUSED_WAVES_PER_SIMD = 4   # k_path_small's occupancy (kernels.h: PATHED_FUSED_WAVES): which row of the VALU probe prices its mix
STATE_BYTES_PER_VERTEX = 112 + 128   # k_shade's path-state streams: read + written per shaded vertex


def parse_args():
    parser = argparse.ArgumentParser()
    parser.add_argument("--gpus", type=int, default=1)
    parser.add_argument("--steps", type=int, default=4)
    parser.add_argument("--warmup", type=int, default=2)
    parser.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                        help="strong: the samples of a step are split over the GPUs (total work fixed); weak: every GPU renders a full step")
    parser.add_argument("--spp-per-step", type=int, default=1024)
    parser.add_argument("--width", type=int, default=1024)
    parser.add_argument("--height", type=int, default=1024)
    parser.add_argument("--scene", default="scenes/cornell.json")
    parser.add_argument("--last-bounce", type=int, default=10)
    parser.add_argument("--seed", type=int, default=1)
    parser.add_argument("--bvh-builder", default="auto", choices=["auto", "sah", "lbvh", "ploc"],
                        help="auto: the host binned-SAH build on one GPU, the on-GPU PLOC build when several ranks each need the tree of a "
                             "mesh of more than a million triangles (N host builds side by side would be seconds of serial time per rank); "
                             "only matters for scenes of more than 64 triangles")
    parser.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                        help="collective backend: nccl (= RCCL over xGMI); gloo stages the reduce through host memory (rehearsals)")
    parser.add_argument("--share-gpu", action="store_true",
                        help="rehearsal on a one-GPU box: every rank renders on device 0 (use with --backend gloo: RCCL refuses two ranks on one device)")
    parser.add_argument("--dist-single", action="store_true",
                        help="run the single-GPU bench as a ONE-rank process group: init, barrier, reduce and all_gather of the N > 1 path execute (RCCL rehearsal on a one-GPU box)")
    parser.add_argument("--no-cpu-baseline", action="store_true")
    parser.add_argument("--no-large-bvh", action="store_true",
                        help="skip the second, BVH-traversal-bound workload (scenes/dragon-standin.json) behind roofline.large_bvh")
    parser.add_argument("--large-bvh-subdiv", type=int, default=10, help="icosphere subdivisions of the stand-in mesh: 9 = 5.2 M, 10 = 21 M triangles")
    parser.add_argument("--time-every-launch", action="store_true",
                        help="HIP events around every trace / shade launch instead of every 8th")
    parser.add_argument("--no-kernel-timing", action="store_true",
                        help="no HIP-event timing of kernel launches (the roofline block becomes null)")
    parser.add_argument("--allow-overrides", action="store_true",
                        help="run although PATHED_* variables are set in the environment (they are listed in the line either way: "
                             "PATHED_HIP_LIB selects another library, the experiments build honours tuning variables)")
    return parser.parse_args()


def environment_overrides():
    """Every PATHED_* variable of this process's environment.  The product library reads none that changes a kernel, a slot
    count or a builder (include/pathed_hip.h: PathedSceneOptions), but PATHED_HIP_LIB swaps the library itself and the
    experiments build listens to dozens: a bench line must say so, and refuses to be produced silently."""
    return {name: value for name, value in sorted(os.environ.items()) if name.startswith("PATHED_")}


# --------------------------------------------------------------------------- launcher (no GPU call)

def visible_gpus():
    """GPUs this process may use, counted WITHOUT a HIP call or torch: the KFD topology lists every node, the ones
    with SIMDs are GPUs; HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES narrow the set."""
    for name in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        value = os.environ.get(name)
        if value is not None:
            return len([item for item in value.split(",") if item.strip() != ""])
    count = 0
    nodes = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in os.listdir(nodes):
            try:
                with open(os.path.join(nodes, node, "properties")) as handle:
                    fields = dict(line.split()[:2] for line in handle if len(line.split()) >= 2)
                if int(fields.get("simd_count", "0")) > 0:
                    count += 1
            except (OSError, ValueError):
                continue
    except OSError:
        return None   # no KFD here: let the ranks find out
    return count or None   # a sysfs view without GPU nodes (a restricted container) says nothing either


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes.  The parent imports
    neither torch nor HIP (devices are counted from sysfs) and never re-executes itself."""
    visible = visible_gpus()
    if visible is not None and visible < args.gpus and not args.share_gpu:
        raise SystemExit("bench.py --gpus %d: this machine shows %d GPU(s)" % (args.gpus, visible))
    with socket.socket() as probe:
        probe.bind(("127.0.0.1", 0))
        port = probe.getsockname()[1]
    children = []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        children.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    code = 0
    try:
        pending = list(children)
        while pending:
            for child in list(pending):
                status = child.poll()
                if status is None:
                    continue
                pending.remove(child)
                if status != 0:
                    code = code or status
                    for other in pending:   # a rank died: the others would wait in a collective forever
                        other.terminate()
            time.sleep(0.05)
    finally:
        for child in children:
            if child.poll() is None:
                child.kill()
    return code


# --------------------------------------------------------------------------- measurement helpers

def algorithmic_bytes(stats):
    """per ray 32 B of ray + S_hit (16 closest / 4 any-hit) + 32 B per child box
    tested + 48 B per leaf triangle tested."""
    return (48 * stats["closest_rays"] + 36 * stats["shadow_rays"]
            + 32 * stats["nodes_visited"] + 48 * stats["tris_tested"])


def kernel_sources_digest():
    """Identifies the kernel sources a committed PMC summary was collected on."""
    digest = hashlib.sha1()
    directory = os.path.join(REPO_ROOT, "pathed_amd", "csrc")
    for name in sorted(os.listdir(directory)):
        with open(os.path.join(directory, name), "rb") as handle:
            digest.update(handle.read())
    return digest.hexdigest()[:16]


def pmc_per_sample(workload):
    """Per-camera-sample counter totals of a workload from the committed rocprofv3 --pmc passes
    (tools/pmc_per_sample.sh -> profiles/pmc_per_sample.json); None when there is no such pass."""
    path = os.path.join(REPO_ROOT, "profiles", "pmc_per_sample.json")
    if not os.path.exists(path):
        return None
    with open(path) as handle:
        entry = json.load(handle).get(workload)
    if entry is not None:
        entry = dict(entry, stale=(entry.get("kernel_sources") != kernel_sources_digest()))
    return entry


def static_mix():
    """The trip-count-weighted static instruction mix of the benchmarked instantiation of the fused kernel
    (tools/static_mix.py --bench-json profiles/static_mix.json); None when it has not been generated."""
    path = os.path.join(REPO_ROOT, "profiles", "static_mix.json")
    if not os.path.exists(path):
        return None
    with open(path) as handle:
        entry = json.load(handle)
    return dict(entry, stale=(entry.get("kernel_sources") != kernel_sources_digest()))


def mix_aware_bound(mix, probes_at_occupancy):
    """The issue rate THIS instruction mix could reach on this box if nothing ever stalled: every class at the rate the
    box's own probe reaches for it at the kernel's occupancy (pathed_hip_measure_valu_modes: v_fma_f32 on three VGPRs for the
    plain class, v_pk_fma_f32 for the packed one, the rcp / sqrt rate recovered from the 6 + 2 mix, v_mul_lo_u32 for the wide
    integer class).  In wave-instructions per second."""
    fma_scalar_operand, mixed, fma, packed, wide = probes_at_occupancy
    slow = 8.0 / mixed - 6.0 / fma_scalar_operand          # seconds per (rcp + sqrt) pair per G instructions
    transcendental = 2.0 / slow if slow > 0 else fma / 4.0
    rates = {"plain": fma, "packed": packed, "transcendental": transcendental, "wide": wide}
    classes = mix["classes"]
    total = float(sum(classes.values()))
    seconds = sum(classes[name] / rates[name] for name in classes)
    return total / seconds, {name: rates[name] / 1e9 for name in rates}


def scene_description(scene, stats):
    """What the workload is, read off the loaded scene (not a hard-coded string)."""
    desc = scene.desc.contents
    kinds = ["Lambertian", "Oren-Nayar", "Microfacet", "Plastic", "Glass", "Mirror"]
    used = sorted({desc.materials[i].type for i in range(desc.n_materials)})
    lights = "environment light" if bool(desc.env) else "area lights"
    intersector = ["4-wide BVH in HBM", "4-wide BVH staged in LDS", "all-triangles kernel (<= 64 triangles)"][stats["scene_in_lds"]]
    return "%d triangles, %d spheres, materials {%s}, %s; %s" % (
        desc.n_triangles, desc.n_spheres, ", ".join(kinds[k] for k in used), lights, intersector)


def host_cores():
    """What this process may run at once, and where each number comes from: the scheduler affinity, the cgroup's CPU quota
    (a GPU box hands a job a share of its cores; os.cpu_count() reports the whole machine, and 256 threads on a 16-core
    share is what made earlier rounds' baseline scale 10x on "256 threads"), and the smaller of the two."""
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = os.cpu_count() or 1
    quota, source = None, None
    try:
        with open("/sys/fs/cgroup/cpu.max") as handle:   # cgroup v2: "<quota|max> <period>"
            first, period = handle.read().split()[:2]
            source = "/sys/fs/cgroup/cpu.max = %s %s" % (first, period)
            if first != "max":
                quota = float(first) / float(period)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as q, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as p:
                value, period = float(q.read()), float(p.read())
                source = "cpu.cfs_quota_us / cpu.cfs_period_us = %g / %g" % (value, period)
                if value > 0:
                    quota = value / period
        except (OSError, ValueError):
            pass
    usable = affinity if quota is None else max(1, min(affinity, int(quota + 0.999)))
    return {"machine_threads": os.cpu_count(), "affinity": affinity, "cgroup_quota_cores": quota, "cgroup_source": source, "usable": usable}


def usable_cores():
    return host_cores()["usable"]


def cpu_baseline(scene, args):
    """Time the CPU oracle on the same workload at reduced spp (rate is spp-independent).  Every thread count tried is
    reported with its rate; `value` is the fastest, `cores` the threads it ran on."""
    sys.path.insert(0, os.path.join(REPO_ROOT, "tests"))
    import oracle_lib  # the checker; used here only as the reported CPU baseline

    host = host_cores()
    allowed = min(host["usable"], oracle_lib.host_threads())   # oracle_lib never starts more than that
    oracle = oracle_lib.OracleScene(scene.desc)
    pixels = args.width * args.height
    oracle.render(args.width, args.height, args.seed, 0, 1, 0, args.last_bounce, threads=allowed)   # warm-up: the BVH, the thread pool
    # the thread count that is fastest HERE: a box may report more hardware threads than the share of it this job runs on.
    # Two samples per pixel each (a few tenths of a second to a few seconds), sample indices disjoint from the timed ones.
    tried = []
    for candidate in sorted({max(1, min(count, allowed)) for count in (allowed // 2, allowed, 8, 16, 24, 32)}):
        t0 = time.perf_counter()
        oracle.render(args.width, args.height, args.seed, 1000, 2, 0, args.last_bounce, threads=candidate)
        took = time.perf_counter() - t0
        tried.append({"threads": candidate, "Msamples_per_s": 2 * pixels / took / 1e6, "spp": 2, "seconds": took})
    best = max(tried, key=lambda row: row["Msamples_per_s"])
    cores = best["threads"]
    spp = max(2, min(64, int(15.0 * best["Msamples_per_s"] * 1e6 / pixels)))
    t0 = time.perf_counter()
    oracle.render(args.width, args.height, args.seed, 1, spp, 0, args.last_bounce, threads=cores)
    elapsed = time.perf_counter() - t0
    samples = pixels * spp
    # one thread on a 256 x 256 rendering of the same scene: the scalar rate the reference's per-core code is comparable to
    from pathed_amd.scene import LoadedScene
    small = LoadedScene(args.scene, 256, 256)
    small_oracle = oracle_lib.OracleScene(small.desc)
    t0 = time.perf_counter()
    small_oracle.render(256, 256, args.seed, 0, 2, 0, args.last_bounce, threads=1)
    single = time.perf_counter() - t0
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    rate, single_rate = samples / elapsed / 1e6, 256 * 256 * 2 / single / 1e6
    return {
        "value": rate,
        "unit": "Msamples/s",
        "cores": cores,
        "threads": cores,
        "host": host,
        "oracle_thread_cap": oracle_lib.host_threads(),
        "thread_counts_tried": tried,
        "machine_threads": host["machine_threads"],
        "cpu_model": model,
        "single_thread": single_rate,
        "parallel_efficiency": rate / (cores * single_rate),
        "kind": "port",
        "sample": "%s %dx%d, %d spp, lastBounce %d, OpenMP over rows (dynamic) on %d threads -- the fastest of %s threads at 2 spp each; "
                  "the job may use %d cores (affinity %d, cgroup quota %s), the machine has %s hardware threads; %.1f s" % (
            args.scene, args.width, args.height, spp, args.last_bounce, cores, " / ".join(str(row["threads"]) for row in tried),
            host["usable"], host["affinity"], "none" if host["cgroup_quota_cores"] is None else "%.1f" % host["cgroup_quota_cores"],
            host["machine_threads"], elapsed),
    }


def kernel_rates(counted, counted_samples, timed, timed_samples):
    """Per-kernel algorithmic rates of a timed region.  `counted`: statistics of a counting pass over
    counted_samples camera samples (exact ray / box / triangle counts; never part of a timed region);
    `timed`: statistics of the timed region (HIP-event time of the sampled launches, launch counts)."""
    if not timed["trace_launches"] or timed["trace_ms"] <= 0:
        return None
    scale = timed_samples / float(counted_samples)
    launches = timed["trace_launches_all"]
    trace_avg_ms = timed["trace_ms"] / timed["trace_launches"]
    shade_avg_ms = timed["shade_ms"] / timed["trace_launches"]
    trace_bytes = algorithmic_bytes(counted) * scale / launches
    # every closest-hit result is one slot visit of k_shade: state in (112 B), state out (128 B) -- whether the trace kernel found
    # it or, for a ray that cannot meet the mesh, the shade kernel itself (local rays, counted apart)
    local_closest, local_shadow = counted.get("local_closest_rays", 0), counted.get("local_shadow_rays", 0)
    vertices = counted["closest_rays"] + local_closest
    shade_bytes = STATE_BYTES_PER_VERTEX * vertices * scale / launches
    return {
        "launches": launches,
        "timed_launches": timed["trace_launches"],
        "trace": {"algorithmic_bytes_per_launch": trace_bytes, "avg_launch_us": trace_avg_ms * 1e3,
                  "achieved": trace_bytes / (trace_avg_ms * 1e-3) / 1e9, "frac": trace_bytes / (trace_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
        "shade": {"algorithmic_bytes_per_launch": shade_bytes, "avg_launch_us": shade_avg_ms * 1e3,
                  "achieved": shade_bytes / (shade_avg_ms * 1e-3) / 1e9, "frac": shade_bytes / (shade_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
        "rays_per_sample": (vertices + counted["shadow_rays"] + local_shadow) / float(counted_samples),
        "traced_rays_per_sample": (counted["closest_rays"] + counted["shadow_rays"]) / float(counted_samples),
        "local_rays_per_sample": (local_closest + local_shadow) / float(counted_samples),
        "vertices_per_sample": vertices / float(counted_samples),
        "algorithmic_bytes_per_sample": {"trace": algorithmic_bytes(counted) / float(counted_samples),
                                         "shade": STATE_BYTES_PER_VERTEX * vertices / float(counted_samples)},
    }


def ensure_large_bvh_mesh(args):
    """The stand-in mesh is generated on the box (the 100 MB file does not travel with the snapshot) -- by a child
    process, so this runs BEFORE this process makes its first GPU call."""
    subprocess.run([sys.executable, os.path.join(REPO_ROOT, "tools", "make_assets.py"), "--dragon", str(args.large_bvh_subdiv)],
                   check=True, stdout=subprocess.DEVNULL)   # a no-op when the file already has that many faces


def large_bvh_leg(args, torch, stream, measured_copy_gbs=None):
    """The workload the north-star roofline target is about: BVH traversal over a tree that does not fit the
    256 MB Infinity Cache (scenes/dragon-standin.json: procedural mesh, 21 M triangles by default, 1920x1080), timed in
    this very run.  Two rows over ONE upload of the scene: the reference's camera (scenes/dragon.json:2-11; most camera
    rays miss the mesh and die on the environment) and a close-up that fills the frame with it
    (scenes/dragon-standin-close.json's camera, set with pathed_hip_scene_set_camera)."""
    import copy
    import ctypes
    from pathed_amd.integrator import HipScene
    from pathed_amd.scene import LoadedScene

    width, height = 1920, 1080
    # timed: 1024 spp = four internal passes of 256, as the 8192-spp configuration runs them
    # (the warm-up is one full pass too: it sizes the partial-sum buffer, and fresh VRAM costs ~40 ms per GB once)
    count_spp, warm_spp, timed_spp = 16, 256, 1024
    t0 = time.perf_counter()
    scene = LoadedScene("scenes/dragon-standin.json", width, height)
    loaded_s = time.perf_counter() - t0
    gpu = HipScene(scene.desc, device=torch.cuda.current_device(), bvh_builder=args.bvh_builder)
    setup_s = time.perf_counter() - t0
    accum = torch.zeros((height, width, 3), dtype=torch.float32, device="cuda")
    pmc = pmc_per_sample("large_bvh")

    def row(label, seed_offset):
        gpu.set_stats_mode(count=False)
        gpu.render_device(args.seed, seed_offset, warm_spp, 0, args.last_bounce, accum.data_ptr(), stream)
        gpu.set_stats_mode(count=True)
        gpu.reset_stats()
        gpu.render_device(args.seed, seed_offset + 1000, count_spp, 0, args.last_bounce, accum.data_ptr(), stream)
        torch.cuda.synchronize()
        counted = gpu.stats()
        gpu.set_stats_mode(count=False, time_sampled=True)
        gpu.reset_stats()
        accum.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gpu.render_device(args.seed, seed_offset + 2000, timed_spp, 0, args.last_bounce, accum.data_ptr(), stream)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        timed = gpu.stats()
        rates = kernel_rates(counted, width * height * count_spp, timed, width * height * timed_spp)
        traffic = None
        if pmc and rates and label == "reference camera":
            traffic = {name: pmc["hbm_bytes_per_sample"][name] * width * height * timed_spp / rates["launches"]
                       for name in ("trace", "shade") if name in pmc.get("hbm_bytes_per_sample", {})}
            # the same kernels against the bytes the COUNTERS saw cross the fabric (2 x FETCH_SIZE + WRITE_SIZE, committed
            # PMC passes): what is left of the algorithmic figure once L1 / L2 / Infinity Cache have served the top of the tree
            for name in traffic:
                kernel = rates[name]
                kernel["counter_bytes_per_launch"] = traffic[name]
                kernel["counter_achieved"] = traffic[name] / (kernel["avg_launch_us"] * 1e-6) / 1e9
                kernel["counter_frac"] = kernel["counter_achieved"] / HBM_PEAK_GBS
        pipeline = None
        if pmc and label == "reference camera" and "hbm_bytes_per_sample" in pmc:
            # the whole pipeline against the memory system: every byte the counters saw cross the fabric per camera sample
            # (trace + shade-side kernels) x the rate of this row
            per_sample = sum(pmc["hbm_bytes_per_sample"].values())
            achieved = per_sample * width * height * timed_spp / elapsed / 1e9
            pipeline = {"counter_bytes_per_sample": per_sample, "achieved": achieved, "unit": "GB/s", "frac_of_peak": achieved / HBM_PEAK_GBS,
                        "frac_of_measured_copy": (achieved / measured_copy_gbs) if measured_copy_gbs else None,
                        "measured_copy": measured_copy_gbs,
                        "note": "2 x FETCH_SIZE + WRITE_SIZE of the committed PMC passes (one 64-spp call) x this row's Msamples/s; "
                                "measured_copy = what a plain read + write stream reaches on this box (pathed_hip_measure_bandwidth)"}
        return counted, timed, {
            "camera": label,
            "pipeline_hbm": pipeline,
            "msamples_per_s": width * height * timed_spp / elapsed / 1e6,
            # the pipeline figures FIRST: k_trace of one pool shares the chip with k_shade of the other
            "k_trace": rates["trace"] if rates else None,
            "k_shade": rates["shade"] if rates else None,
            "rays_per_sample": rates["rays_per_sample"] if rates else None,
            "traced_rays_per_sample": rates["traced_rays_per_sample"] if rates else None,
            "local_rays_per_sample": rates["local_rays_per_sample"] if rates else None,
            "vertices_per_sample": rates["vertices_per_sample"] if rates else None,
            "mrays_per_s": (rates["rays_per_sample"] * width * height * timed_spp / elapsed / 1e6) if rates else None,
            "traffic": traffic,
            "image_mean_rgb": (accum / float(timed_spp)).mean(dim=(0, 1)).tolist(),
        }

    counted, timed, reference_row = row("reference camera", 0)
    close_camera = copy.copy(scene.desc.contents.camera)
    close_camera.origin = (ctypes.c_float * 3)(117.4, -127.3, 143.7)   # scenes/dragon-standin-close.json
    close_camera.target = (ctypes.c_float * 3)(0.0, 0.0, 25.0)
    gpu.set_camera(close_camera)
    _, _, close_row = row("close-up: the mesh fills the frame", 100000)
    gpu.set_camera(scene.desc.contents.camera)

    result = {
        "workload": "scenes/dragon-standin.json %dx%d, %d spp timed per row (%d spp counted), bounces 0..%d; %s" % (
            width, height, timed_spp, count_spp, args.last_bounce, scene_description(scene, timed)),
        "bvh_bytes": timed["bvh_bytes"], "bvh_builder": args.bvh_builder, "bvh_build_ms": timed["bvh_build_ms"],
        "scene_load_s": loaded_s, "setup_s": setup_s,
        "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "note": "k_trace / k_shade: algorithmic bytes per launch (SURVEY.md 8d) / HIP-event time of the launch AS THE PIPELINE RUNS "
                "(two pools: a trace launch shares the chip with the other pool's shade launch); one_pool below is a diagnostic.  "
                "Since round 5 the shade kernel answers the queries that cannot meet the mesh itself (local_rays_per_sample): k_trace's "
                "launches carry the tree-walking rays only, fewer and longer ones",
        # headline row = the reference's camera; the same keys as round 2 so the lines stay comparable
        "msamples_per_s": reference_row["msamples_per_s"],
        "k_trace": reference_row["k_trace"], "k_shade": reference_row["k_shade"],
        "rays_per_sample": reference_row["rays_per_sample"],
        "traffic": reference_row["traffic"],
        "traffic_source": None if not pmc else {"file": "profiles/pmc_per_sample.json", "stale": pmc["stale"]},
        "image_mean_rgb": reference_row["image_mean_rgb"],
        "rows": [reference_row, close_row],
    }
    # [r5] The same pipeline seen from the instruction issue side: what its two kernels issue per camera sample (a committed
    # counter pass over ONE 64-spp call of this workload: ramp-up and drain included) x this run's rate.  The HBM fractions above
    # are the contract's; this says how much of the chip's VALU issue the pipeline takes while it runs.
    by_class = (pmc or {}).get("valu_by_class")
    if by_class and "trace" in by_class and "shade" in by_class:
        per_sample = by_class["trace"]["wave_instructions_per_sample"] + by_class["shade"]["wave_instructions_per_sample"]
        issued = per_sample * reference_row["msamples_per_s"] * 1e6 / 1e9
        result["valu"] = {
            "wave_instructions_per_sample": {k: by_class[k]["wave_instructions_per_sample"] for k in ("trace", "shade")},
            "lane_utilisation": {k: by_class[k]["lane_utilisation"] for k in ("trace", "shade")},
            "achieved": issued, "peak": VALU_PEAK_GINSTR, "unit": "G wave-instr/s", "frac": issued / VALU_PEAK_GINSTR,
            "source": {"file": "profiles/pmc_per_sample.json", "stale": pmc["stale"]},
            "note": "k_trace + k_shade VALU wave-instructions per camera sample (counter pass, 64-spp call) x the reference-camera row's Msamples/s, "
                    "against the guide's issue peak: the two kernels overlap on two streams and TOGETHER issue this much",
        }
    gpu.close()
    # Diagnostic: one pool -- the launches alternate, HIP events around every launch time each kernel on its own
    alone = HipScene(scene.desc, device=torch.cuda.current_device(), bvh_builder=args.bvh_builder, pools=1)
    alone.render_device(args.seed, 0, warm_spp, 0, args.last_bounce, accum.data_ptr(), stream)
    alone.set_stats_mode(count=False, time_kernels=True)
    alone.reset_stats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    alone.render_device(args.seed, 2000, 256, 0, args.last_bounce, accum.data_ptr(), stream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    serial = alone.stats()
    if serial["trace_launches"]:   # one k_shade launch follows every k_trace launch
        result["one_pool"] = {
            "note": "diagnostic, PathedSceneOptions.pools = 1: no overlap, every launch timed with HIP events, 256 spp, reference camera",
            "msamples_per_s": width * height * 256 / elapsed / 1e6,
            "k_trace_avg_launch_us": 1e3 * serial["trace_ms"] / serial["trace_launches"],
            "k_shade_avg_launch_us": 1e3 * serial["shade_ms"] / serial["trace_launches"],
            "k_trace_ms": serial["trace_ms"], "k_shade_ms": serial["shade_ms"], "launches": serial["trace_launches"],
        }
        own = kernel_rates(counted, width * height * count_spp, serial, width * height * 256)
        if own:   # the same algorithmic bytes against each kernel's own time
            result["one_pool"]["k_trace"] = own["trace"]
            result["one_pool"]["k_shade"] = own["shade"]
    alone.close()
    return result


# --------------------------------------------------------------------------- one rank

def run_rank(args):
    import torch
    import torch.distributed as dist

    if args.dist_single and "WORLD_SIZE" not in os.environ:
        with socket.socket() as probe:
            probe.bind(("127.0.0.1", 0))
            port = probe.getsockname()[1]
        os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world_size > 1 or args.dist_single
    wants_large_bvh = (world_size == 1 and not args.no_large_bvh and not args.no_kernel_timing and args.scene == "scenes/cornell.json")
    if wants_large_bvh and rank == 0:
        ensure_large_bvh_mesh(args)   # a child process: before anything here touches the GPU
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    staged = args.backend == "gloo"   # collectives on host copies
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if staged:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        world_size = dist.get_world_size()   # the ranks the backend actually sees

    from pathed_amd import _capi, parallel
    from pathed_amd.integrator import HipScene, measure_bandwidth
    from pathed_amd.scene import LoadedScene

    setup_begin = time.perf_counter()
    scene = LoadedScene(args.scene, args.width, args.height)
    if args.bvh_builder == "auto":
        args.bvh_builder = "ploc" if (world_size > 1 and scene.n_triangles > 1000000) else "sah"
    gpu = HipScene(scene.desc, device=local_rank, bvh_builder=args.bvh_builder)
    setup_s = time.perf_counter() - setup_begin   # scene file -> flat description -> BVH -> upload, this rank
    stream = torch.cuda.current_stream().cuda_stream

    accum = torch.zeros((args.height, args.width, 3), dtype=torch.float32, device="cuda")
    spp = args.spp_per_step
    strong = args.scaling == "strong"

    def step_range(index):
        """Sample indices this rank renders in step `index` (warm-up steps first)."""
        if strong:   # the step's samples [index * spp, (index + 1) * spp) split over the ranks
            return parallel.strong_range(rank, world_size, index * spp, spp)
        # weak: every rank renders spp samples of its own, disjoint stream
        return parallel.weak_range(rank, (args.warmup + args.steps) * spp, first=index * spp)[0], spp

    def step(index):
        begin, count = step_range(index)
        if count > 0:
            gpu.render_device(args.seed, begin, count, 0, args.last_bounce, accum.data_ptr(), stream)

    # one untimed, counting pass over the first timed step's samples: exact ray / box / triangle counts
    counted = None
    counted_samples = 0
    timing = not args.no_kernel_timing
    if rank == 0 and timing:   # (a PMC run passes --no-kernel-timing: nothing but the timed path's kernels then)
        gpu.set_stats_mode(count=True)
        gpu.reset_stats()
        scratch = torch.zeros_like(accum)
        begin, count = step_range(args.warmup)
        count = max(count, 1)
        gpu.render_device(args.seed, begin, count, 0, args.last_bounce, scratch.data_ptr(), stream)
        torch.cuda.synchronize()
        counted = gpu.stats()
        counted_samples = args.width * args.height * count
        del scratch
    gpu.set_stats_mode(count=False)

    for index in range(args.warmup):
        step(index)
    if distributed:
        # warm the collective itself: RCCL sets up the channels of an operation on its first use, which would otherwise
        # land inside the timed region (the reduce there is 12.6 MB: a fraction of a millisecond once the links are up)
        if staged:
            dist.reduce(accum.cpu(), dst=0, op=dist.ReduceOp.SUM)
        else:
            dist.reduce(accum.clone(), dst=0, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
    accum.zero_()

    # HIP-event pairs around every 8th launch of each pool: timing every launch keeps a pool's kernels from
    # running back to back and costs ~6 % of the rate (--time-every-launch restores it)
    gpu.set_stats_mode(count=False, time_kernels=timing and args.time_every_launch,
                       time_sampled=timing and not args.time_every_launch)
    gpu.reset_stats()

    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for index in range(args.steps):
        step(args.warmup + index)
    torch.cuda.synchronize()
    rendered = time.perf_counter() - t0
    if distributed:
        # the path's one exchange step: per-GPU radiance sums -> rank 0 (RCCL reduce over xGMI)
        if staged:
            host_sums = accum.cpu()
            dist.reduce(host_sums, dst=0, op=dist.ReduceOp.SUM)
            if rank == 0:
                accum.copy_(host_sums)
        else:
            dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize()
    reduced = time.perf_counter() - t0
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    per_rank = [[rendered, reduced - rendered, elapsed, setup_s]]
    if distributed:
        mine = torch.tensor(per_rank[0], dtype=torch.float64, device="cpu" if staged else "cuda")
        gathered = [torch.zeros_like(mine) for _ in range(world_size)]
        dist.all_gather(gathered, mine)
        per_rank = [g.tolist() for g in gathered]
        elapsed = max(row[2] for row in per_rank)   # MAX over ranks

    timed = gpu.stats()

    if rank == 0:
        pixels = args.width * args.height
        total_spp = spp * args.steps * (1 if strong else world_size)
        total_samples = pixels * total_spp
        value = total_samples / elapsed / 1e6
        my_samples = pixels * sum(step_range(args.warmup + index)[1] for index in range(args.steps))

        roofline = None
        if timing and counted is not None:
            fused = counted["path_kernel"] == 3
            rates = None if fused else kernel_rates(counted, counted_samples, timed, my_samples)
            measured_read, measured_copy = measure_bandwidth(gib=2.0, repeats=10)
            workload_key = "cornell_1024" if (args.scene, args.width, args.height) == ("scenes/cornell.json", 1024, 1024) else None
            pmc = pmc_per_sample(workload_key) if workload_key else None
            launches = timed["trace_launches_all"]
            traffic = None
            if pmc and launches and "hbm_bytes_per_sample" in pmc:
                traffic = sum(pmc["hbm_bytes_per_sample"].values()) * my_samples / launches
            hbm = None
            if rates:
                # the kernel that takes the larger share of the timed region is the dominant one
                dominant = "shade" if timed["shade_ms"] >= timed["trace_ms"] else "trace"
                hbm = {
                    "dominant_kernel": {"shade": "k_shade (path-state streams, 112 B read + 128 B written per shaded vertex)",
                                        "trace": "k_trace (BVH traversal + triangle/sphere intersect)"}[dominant],
                    "k_shade": rates["shade"], "k_trace": rates["trace"] if counted["scene_in_lds"] == 0 else None,
                    "peak_measured": {"stream_read": measured_read, "stream_copy": measured_copy, "unit": "GB/s",
                                      "note": "2 GiB probe, 16 B per lane, HIP events (pathed_hip_measure_bandwidth)"},
                    "k_shade_frac_of_measured_copy": rates["shade"]["achieved"] / measured_copy,
                    "trace_ms_timed": timed["trace_ms"], "shade_ms_timed": timed["shade_ms"],
                    "rays_per_sample": rates["rays_per_sample"], "vertices_per_sample": rates["vertices_per_sample"],
                }
            if counted["scene_in_lds"] == 2:
                # All-triangles scenes (<= 64 triangles: every ray query runs out of registers and scalar loads):
                # neither HBM nor MFMA bounds the path, VALU issue does.  Denominator: the best rate any occupancy
                # reaches on THIS box with independent v_fma_f32 on VGPR operands (pathed_hip_measure_valu_modes);
                # numerator: SQ_INSTS_VALU per camera sample from the committed PMC pass x the samples of the timed
                # region / the kernels' own time, measured live with HIP events (fused kernel: every launch).
                from pathed_amd.integrator import VALU_MODES, measure_valu_modes
                probes = {waves: measure_valu_modes(waves, repeats=5) for waves in (1, 4, 8)}
                measured = max(row[2] for row in probes.values())
                # the denominator is the GUIDE's: MI355X_MICROARCH.md constants table, v_fma_f32 wave64 2 cycles, 1 024 SIMDs, 2.4 GHz.
                # What this box reaches with independent v_fma_f32 (it clocks ~2.2 GHz under the load and issues every ~2.5
                # cycles) is reported beside it as peak_measured / frac_of_measured.
                peak_guide = 1228.8e9
                issued = pmc["valu_wave_instructions_per_sample"] * my_samples if pmc else None
                kernel_s = (timed["trace_ms"] * 1e-3) if fused and timed["trace_ms"] > 0 else rendered
                if hbm is None and traffic is not None and launches:
                    # "report HBM fraction anyway and say the kernel is ALU-bound" -- the counter bytes of a launch
                    # (partial sums out, scene records in) over the launch's own time
                    per_launch_s = kernel_s / launches
                    hbm = {"achieved": traffic / per_launch_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": traffic / per_launch_s / 1e9 / HBM_PEAK_GBS, "bytes_per_launch": traffic,
                           "note": "the fused kernel keeps whole paths in registers: HBM sees the partial sums and the scene records only; "
                                   "VALU issue bounds it, not memory"}
                lanes = pmc.get("valu_lane_utilisation") if pmc else None
                mix = static_mix()
                mix_block = None
                if mix and issued:
                    bound, class_rates = mix_aware_bound(mix, probes[PATHED_FUSED_WAVES_PER_SIMD])
                    mix_block = {
                        "bound": bound / 1e9, "unit": "G wave-instr/s", "frac": issued / kernel_s / bound,
                        "classes_per_iteration": mix["classes"], "class_rates": class_rates, "waves_per_simd": PATHED_FUSED_WAVES_PER_SIMD,
                        "source": {"file": "profiles/static_mix.json", "stale": mix["stale"], "method": mix["method"]},
                        "note": "what this kernel's own instruction mix could issue on this box with no stall at all (every class at the box's probed "
                                "rate at four waves per SIMD); frac = achieved / bound: what stalls, dependencies and LDS / scalar traffic cost on top of the mix"}
                roofline = {
                    "bound": "valu",
                    "kernel": "k_path_small (fused: camera ray .. termination in registers, one persistent launch per pass)" if fused
                              else "whole wavefront pipeline (k_trace_small + shade kernel, both pools)",
                    "achieved": (issued / kernel_s / 1e9) if issued else None,
                    "peak": peak_guide / 1e9,
                    "unit": "G wave-instr/s",
                    "frac": (issued / kernel_s / peak_guide) if issued else None,
                    # frac counts instructions ISSUED, whatever share of the 64 lanes they worked for: useful_frac = frac x lane
                    # utilisation (SQ_THREAD_CYCLES_VALU / 64 SQ_ACTIVE_INST_VALU) is the share of the chip's lane-slots that did work
                    "lane_utilisation": lanes,
                    "useful_frac": (issued / kernel_s / peak_guide * lanes) if (issued and lanes) else None,
                    "mix_aware": mix_block,
                    "valu_wave_instructions_per_sample": pmc["valu_wave_instructions_per_sample"] if pmc else None,
                    "peak_measured": measured / 1e9,
                    "frac_of_measured": (issued / kernel_s / measured) if issued else None,
                    "traffic": traffic,
                    "kernel_seconds": kernel_s, "launches": launches,
                    "rays_per_sample": (counted["closest_rays"] + counted["shadow_rays"]) / float(counted_samples),
                    "peak_note": "peak = the guide's 2 cycles per v_fma_f32 per SIMD at 2.4 GHz; peak_measured = best rate of "
                                 "pathed_hip_measure_valu_modes on this box (v_fma_f32, three VGPR operands, 8 chains, 1 / 4 / 8 waves per SIMD)",
                    "peak_probe": {"modes": list(VALU_MODES),
                                   "waves_per_simd": {str(waves): [rate / 1e9 for rate in row] for waves, row in probes.items()}},
                    "instructions_source": None if not pmc else {
                        "file": "profiles/pmc_per_sample.json", "valu_wave_instructions_per_sample": pmc["valu_wave_instructions_per_sample"],
                        "collected": "a committed rocprofv3 --pmc pass over this kernel (tools/pmc_per_sample.sh), NOT counters of this run: "
                                     "a process cannot read its own PMC counters; `stale` says whether the kernel sources have changed since",
                        "stale": pmc["stale"]},
                    "hbm": hbm,
                }
            elif rates:
                dominant = "shade" if timed["shade_ms"] >= timed["trace_ms"] else "trace"
                roofline = {
                    "bound": "hbm", "kernel": hbm["dominant_kernel"],
                    "achieved": rates[dominant]["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": rates[dominant]["frac"],
                    "traffic": traffic, "hbm": hbm,
                }
            if roofline is not None and wants_large_bvh:
                roofline["large_bvh"] = large_bvh_leg(args, torch, stream, measured_copy)

        baseline = None
        # the CPU oracle is timed on rank 0 of the single-GPU run only
        if not args.no_cpu_baseline and world_size == 1:
            baseline = cpu_baseline(scene, args)

        mean = (accum / float(total_spp)).mean(dim=(0, 1)).tolist()
        line = {
            "metric": "Msamples/s (pixels x spp / s), PathTracer radiance loop",
            "value": value,
            "unit": "Msamples/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic camera samples (counter-based random stream, seed %d) on %s, the reference's own scene file: %s" % (
                args.seed, args.scene, scene_description(scene, timed)),
            "config": {
                "workload": "%s %dx%d, %d spp per step x %d steps = %d spp%s, bounces 0..%d, seed %d" % (
                    args.scene, args.width, args.height, spp, args.steps, total_spp,
                    " (the 4096-spp configuration = 4 steps of 1024)" if (spp, args.steps) == (1024, 4) and strong else "",
                    args.last_bounce, args.seed),
                "spp_per_step": spp,
                "total_samples": total_samples,
                "parallelism": ("the samples of every step split over %d GPU(s)" if strong else "a full step on each of %d GPU(s)") % world_size
                               + (", one %s of 3*W*H fp32 sums to rank 0 inside the timed region" % (
                                   "reduce staged through host memory (gloo)" if staged else "RCCL reduce") if distributed else ", no exchange"),
                "collective": None if not distributed else "%s, world size %d" % ("gloo (host-staged)" if staged else "nccl (RCCL)", world_size),
            },
            "per_rank_s": {"render": [row[0] for row in per_rank], "reduce": [row[1] for row in per_rank], "total": [row[2] for row in per_rank],
                           "setup": [row[3] for row in per_rank]},
            "bvh_builder": args.bvh_builder,
            "roofline": roofline,
            "cpu_baseline": baseline,
            "image_mean_rgb": mean,
            "dropped_samples": timed["dropped_samples"],
            "environment_overrides": environment_overrides(),
            "library": {"path": os.path.relpath(_capi.hip_library_path(), REPO_ROOT), "experiments_build": bool(_capi.load_hip().pathed_hip_has_experiments())},
        }
        print(json.dumps(line), flush=True)

    if distributed:
        dist.destroy_process_group()
    return 0


def main():
    args = parse_args()
    overrides = environment_overrides()
    if overrides and not args.allow_overrides:
        sys.stderr.write("bench.py: PATHED_* variables are set (%s): unset them, or pass --allow-overrides to run anyway "
                         "(they are reported in the line's \"environment_overrides\")\n" % ", ".join("%s=%s" % item for item in overrides.items()))
        return 3
    if "WORLD_SIZE" in os.environ or args.gpus <= 1:
        return run_rank(args)      # a rank of torch.distributed.run / of our own launcher, or the single-GPU run
    return launch_ranks(args)


if __name__ == "__main__":
    sys.exit(main())
