#!/usr/bin/env python3
"""Headline benchmark: Msamples/s of the GPU PathTracer on scenes/cornell.json, 1024x1024.

A "step" renders `--spp-per-step` camera samples (default 1024: one internal pass of the library) for every
pixel through the C ABI (pathed_hip_render_device) into a device-resident radiance-sum buffer; 4 steps are
the 4096-spp configuration BASELINE.json quotes.  With N ranks each rank renders its own,
disjoint range of sample indices (weak scaling: per-GPU work is fixed) and the sums are
reduced to rank 0 over RCCL once, inside the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline      algorithmic bytes of the BVH-traversal kernel / its HIP-event time
  cpu_baseline  the CPU oracle (a port of the reference estimator; the reference binary
                cannot be built) timed on the host cores, bounded sample
"""
import argparse
import json
import os
import sys
import time

REPO_ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO_ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse_args():
    parser = argparse.ArgumentParser()
    parser.add_argument("--gpus", type=int, default=1)
    parser.add_argument("--steps", type=int, default=4)
    parser.add_argument("--warmup", type=int, default=2)
    parser.add_argument("--spp-per-step", type=int, default=1024)
    parser.add_argument("--width", type=int, default=1024)
    parser.add_argument("--height", type=int, default=1024)
    parser.add_argument("--scene", default="scenes/cornell.json")
    parser.add_argument("--last-bounce", type=int, default=10)
    parser.add_argument("--seed", type=int, default=1)
    parser.add_argument("--bvh-builder", default="sah", choices=["sah", "lbvh", "ploc"],
                        help="host binned-SAH build (default) or an on-GPU build; only matters for scenes of more than 64 triangles")
    parser.add_argument("--no-cpu-baseline", action="store_true")
    parser.add_argument("--time-every-launch", action="store_true",
                        help="HIP events around every trace / shade launch instead of every 8th")
    parser.add_argument("--no-kernel-timing", action="store_true",
                        help="skip the HIP-event timing of every trace launch (roofline.achieved becomes null)")
    return parser.parse_args()


def algorithmic_bytes(stats):
    """per ray 32 B of ray + S_hit (16 closest / 4 any-hit) + 32 B per child box
    tested + 48 B per leaf triangle tested."""
    return (48 * stats["closest_rays"] + 36 * stats["shadow_rays"]
            + 32 * stats["nodes_visited"] + 48 * stats["tris_tested"])


def cpu_baseline(scene, args):
    """Time the CPU oracle on the same workload at reduced spp (rate is spp-independent)."""
    sys.path.insert(0, os.path.join(REPO_ROOT, "tests"))
    import oracle_lib  # the checker; used here only as the reported CPU baseline

    cores = os.cpu_count() or 1
    oracle = oracle_lib.OracleScene(scene.desc)
    t0 = time.perf_counter()
    oracle.render(args.width, args.height, args.seed, 0, 1, 0, args.last_bounce, threads=cores)
    one = time.perf_counter() - t0
    spp = max(1, min(64, int(15.0 / max(one, 1e-3))))
    t0 = time.perf_counter()
    oracle.render(args.width, args.height, args.seed, 1, spp, 0, args.last_bounce, threads=cores)
    elapsed = time.perf_counter() - t0
    samples = args.width * args.height * spp
    return {
        "value": samples / elapsed / 1e6,
        "unit": "Msamples/s",
        "cores": cores,
        "kind": "port",
        "sample": "%s %dx%d, %d spp, lastBounce %d, OpenMP over rows, %.1f s" % (
            args.scene, args.width, args.height, spp, args.last_bounce, elapsed),
    }


def main():
    args = parse_args()

    import torch
    import torch.distributed as dist

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world_size > 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from pathed_amd.integrator import HipScene
    from pathed_amd.scene import LoadedScene

    scene = LoadedScene(args.scene, args.width, args.height)
    gpu = HipScene(scene.desc, device=local_rank, bvh_builder=args.bvh_builder)
    stream = torch.cuda.current_stream().cuda_stream

    accum = torch.zeros((args.height, args.width, 3), dtype=torch.float32, device="cuda")
    spp = args.spp_per_step
    # rank r owns sample indices [r * (W+K) * spp, (r+1) * (W+K) * spp): disjoint streams
    base = rank * (args.warmup + args.steps) * spp

    def step(index):
        gpu.render_device(args.seed, base + index * spp, spp, 0, args.last_bounce, accum.data_ptr(), stream)

    # one untimed, counting pass over the first step's samples: exact ray / box / triangle counts
    per_step_stats = None
    if rank == 0:
        gpu.set_stats_mode(count=True)
        gpu.reset_stats()
        scratch = torch.zeros_like(accum)
        gpu.render_device(args.seed, base + args.warmup * spp, spp, 0, args.last_bounce, scratch.data_ptr(), stream)
        torch.cuda.synchronize()
        per_step_stats = gpu.stats()
        del scratch
    gpu.set_stats_mode(count=False)

    for index in range(args.warmup):
        step(index)
    accum.zero_()

    # HIP-event pairs around every 8th launch of each pool: timing every launch keeps a pool's kernels from
    # running back to back and costs ~6 % of the rate (--time-every-launch restores it)
    gpu.set_stats_mode(count=False, time_kernels=(not args.no_kernel_timing) and args.time_every_launch,
                       time_sampled=(not args.no_kernel_timing) and not args.time_every_launch)
    gpu.reset_stats()

    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for index in range(args.steps):
        step(args.warmup + index)
    if distributed:
        # the path's one exchange step: per-GPU radiance sums -> rank 0 (RCCL reduce over xGMI)
        dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    if distributed:
        slowest = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(slowest, op=dist.ReduceOp.MAX)
        elapsed = float(slowest.item())

    timed_stats = gpu.stats()

    if rank == 0:
        samples_per_rank = args.width * args.height * spp * args.steps
        total_samples = samples_per_rank * world_size
        value = total_samples / elapsed / 1e6

        roofline = None
        if per_step_stats is not None:
            bytes_per_step = algorithmic_bytes(per_step_stats)
            timed_launches = timed_stats["trace_launches"]          # launches bracketed by HIP events
            launches = timed_stats["trace_launches_all"]            # all trace launches of the timed region
            trace_ms = timed_stats["trace_ms"]
            total_bytes = bytes_per_step * args.steps
            # bytes per launch / average duration of the timed launches
            achieved = ((total_bytes / launches) / (trace_ms / timed_launches * 1e-3) / 1e9) if trace_ms > 0 and launches else None
            traffic = None
            # PMC passes exist for two workloads (tools/profile_bench.sh): the default one and the large-BVH scene
            traffic_files = {("scenes/cornell.json", 2): "hbm_traffic.json", ("scenes/dragon-standin.json", 0): "r1s2dragon_hbm_traffic.json"}
            traffic_name = traffic_files.get((args.scene, per_step_stats["scene_in_lds"]))
            if traffic_name and os.path.exists(os.path.join(REPO_ROOT, "profiles", traffic_name)):
                with open(os.path.join(REPO_ROOT, "profiles", traffic_name)) as handle:
                    traffic = json.load(handle).get("trace_hbm_bytes_per_launch")
            # what a plain streaming kernel reaches on this box: the second denominator
            from pathed_amd.integrator import measure_bandwidth
            measured_read, measured_copy = measure_bandwidth(gib=2.0, repeats=10)
            # Secondary bound for the all-triangles scenes ("FP32 VALU issue rate"):
            # wave instructions per launch from the committed PMC pass of this very workload
            # (profiles/r1s2_pmc_cornell.json, SQ_INSTS_VALU), launches and time measured live.
            # A wave64 VALU instruction occupies its SIMD for 4 cycles; 4 SIMDs per CU.
            valu = None
            pmc_path = os.path.join(REPO_ROOT, "profiles", "r1s2_pmc_cornell.json")
            if os.path.exists(pmc_path) and per_step_stats["scene_in_lds"] == 2 and args.scene == "scenes/cornell.json" \
                    and (args.width, args.height) == (1024, 1024) and spp >= 256:
                with open(pmc_path) as handle:
                    pmc = json.load(handle)["kernels"]
                trace_instructions = pmc["void pathed::k_trace_small<false>"]["SQ_INSTS_VALU"]
                shade_instructions = pmc["void pathed::k_shade<true>"]["SQ_INSTS_VALU"]
                simds = 4 * torch.cuda.get_device_properties(local_rank).multi_processor_count
                clock_hz = 2.4e9  # MI355X peak engine clock, /opt/skills/guides/MI355X_MICROARCH.md
                issued = launches * (trace_instructions + shade_instructions)
                valu = {
                    "bound": "valu issue",
                    "wave_instructions_per_launch": {"k_trace_small": trace_instructions, "k_shade": shade_instructions},
                    "achieved": issued / elapsed / 1e9, "peak": simds * clock_hz / 4 / 1e9, "unit": "G wave-instructions/s",
                    "frac": issued * 4 / (simds * clock_hz * elapsed),
                    "note": "whole pipeline (both kernels, both pools) over the timed region; per-launch counts from the PMC pass at 256 spp per call (profiles/r1s2_pmc_cornell.json)",
                }
            # k_shade is bound by its state streams: HBM bytes per launch from the same PMC passes,
            # launch duration live, against the read-plus-write rate the bandwidth probe reaches on this box
            shade_streams = None
            shade_launches = timed_stats["trace_launches"]   # one k_shade per timed k_trace launch
            if traffic_name and shade_launches and timed_stats["shade_ms"] > 0:
                with open(os.path.join(REPO_ROOT, "profiles", traffic_name)) as handle:
                    kernels = json.load(handle).get("kernels", {})
                shade_entry = next((v for k, v in kernels.items() if "k_shade" in k), None)
                if shade_entry and "FETCH_SIZE_KiB_avg" in shade_entry and "WRITE_SIZE_KiB_avg" in shade_entry:
                    shade_bytes = (2.0 * shade_entry["FETCH_SIZE_KiB_avg"] + shade_entry["WRITE_SIZE_KiB_avg"]) * 1024.0
                    shade_avg_ms = timed_stats["shade_ms"] / shade_launches
                    shade_rate = shade_bytes / (shade_avg_ms * 1e-3) / 1e9
                    shade_streams = {
                        "kernel": "k_shade (path state streams: read 8 x 16 B, write 6 x 16 B per slot)",
                        "hbm_bytes_per_launch": shade_bytes, "avg_launch_ms": shade_avg_ms, "achieved": shade_rate, "unit": "GB/s",
                        "frac_of_measured_copy": shade_rate / measured_copy,
                        "note": "while sharing the chip with the other pool's trace kernel; alone it reaches 3.6 TB/s",
                    }
            roofline = {
                "bound": "hbm",
                "kernel": "k_trace (BVH traversal + triangle/sphere intersect, closest + any-hit)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                "peak_measured": {"stream_read": measured_read, "stream_copy": measured_copy, "unit": "GB/s",
                                  "note": "2 GiB probe, 16 B per lane, HIP events (pathed_hip_measure_bandwidth)"},
                "frac_of_measured_read": (achieved / measured_read) if achieved else None,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": (total_bytes / launches) if launches else None,
                "algorithmic_bytes_per_sample": bytes_per_step / (args.width * args.height * spp),
                "avg_launch_ms": (trace_ms / timed_launches) if timed_launches else None,
                "launches": launches,
                "timed_launches": timed_launches,
                "rays_per_sample": (per_step_stats["closest_rays"] + per_step_stats["shadow_rays"])
                / (args.width * args.height * spp),
                "bvh_resident": ["HBM", "LDS", "none (<= 64 triangles: every ray tests all, scalar loads)"][per_step_stats["scene_in_lds"]],
                "trace_ms_timed": trace_ms,
                "shade_ms_timed": timed_stats["shade_ms"],
                "valu": valu,
                "shade_streams": shade_streams,
            }

        baseline = None
        # the CPU oracle is timed on rank 0 of the single-GPU run only
        if not args.no_cpu_baseline and world_size == 1:
            baseline = cpu_baseline(scene, args)

        mean = (accum / float(spp * args.steps * world_size)).mean(dim=(0, 1)).tolist()
        line = {
            "metric": "Msamples/s (pixels x spp / s), PathTracer radiance loop",
            "value": value,
            "unit": "Msamples/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic camera samples on scenes/cornell.json (the reference's own scene file)",
            "config": {
                "workload": "%s %dx%d, %d spp per step x %d steps per GPU (the 4096-spp configuration = 4 steps of 1024), "
                            "Lambertian, bounces 0..%d, seed %d" % (
                                args.scene, args.width, args.height, spp, args.steps, args.last_bounce, args.seed),
                "spp_per_step": spp,
                "samples_per_gpu": samples_per_rank,
                "parallelism": "spp sharded over %d GPU(s), one RCCL reduce of 3*W*H fp32" % world_size,
            },
            "roofline": roofline,
            "cpu_baseline": baseline,
            "image_mean_rgb": mean,
            "dropped_samples": timed_stats["dropped_samples"],
        }
        print(json.dumps(line), flush=True)

    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
