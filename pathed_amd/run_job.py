#!/usr/bin/env python3
"""Multi-GPU job runner: `pathed <job.json>` with the samples of every pixel sharded over the GPUs
of one node (one process per GPU, RCCL reduce of the radiance sums to rank 0).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        -m pathed_amd.run_job job.json
    python -m pathed_amd.run_job job.json            # single GPU

Same job.json keys, output files and log lines as the reference (src/job.cpp:33-63,
src/integrator.cpp:69-102): <output_directory>/report.json, auto.exr, auto-%05dspp.exr at every
power-of-two sample count (and only there, like the reference), "[<outdir>/] sample: i/N (Xs elapsed)".
Every batch of samples is split over the ranks (contiguous shares, pathed_amd/parallel.py) and each rank
keeps adding into its own sums; they are reduced only when a checkpoint is due, so the image at every
checkpoint holds exactly the samples [0, n): the single-GPU image up to fp32 summation order.
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    job_path = argv[0] if argv else "job.json"
    asset_root = argv[1] if len(argv) > 1 else None

    import torch
    import torch.distributed as dist

    from . import _capi, parallel
    from .integrator import BounceController, HipScene
    from .scene import LoadedScene

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("pathed_amd.run_job needs a GPU: there is no CPU path")
    torch.cuda.set_device(local_rank)
    if world_size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    with open(job_path) as handle:
        job = json.load(handle)
    width, height = job["width"], job["height"]
    spp = job["spp"] if job["spp"] > 0 else 9999999
    seed = int(job.get("seed", 1))
    bounces = BounceController(job["startBounce"], job["lastBounce"])
    if job["integrator"] not in ("PathTracer", "DataParallelIntegrator", "VolumePathTracer"):
        raise SystemExit("Unimplemented")  # the reference throws "Unimplemented" (src/job.cpp:96)
    out_dir = job["output_directory"] + "/"
    # keys of the C++ host this runner does not implement are refused, not ignored: a job file must not behave
    # differently between the two runners (here the GPUs are the ranks of torch.distributed.run, and there is no auto.state)
    if job.get("resume", False):
        raise SystemExit("pathed_amd.run_job: \"resume\" is a key of the C++ host (pathed <job.json>); this runner keeps no auto.state")
    if "gpus" in job:
        raise SystemExit("pathed_amd.run_job: \"gpus\" is a key of the C++ host; start this runner under torch.distributed.run, one rank per GPU")

    # rank 0 decides whether the job may run (output directory rules, src/job.cpp:33-63) and tells the
    # others BEFORE anyone builds a scene or enters a collective: every rank leaves with the same code
    go = 1
    if rank == 0:
        if os.path.isdir(out_dir):
            print("Output directory already exists: %s" % out_dir)
            if not job.get("force", False):
                go = 0
        if go:
            os.makedirs(out_dir, exist_ok=True)
            with open(os.path.join(out_dir, "report.json"), "w") as handle:
                json.dump(job, handle, indent=4)
    if world_size > 1:
        decision = torch.tensor([go], dtype=torch.int32, device="cuda")
        dist.broadcast(decision, src=0)
        go = int(decision.item())
    if not go:
        if world_size > 1:
            dist.destroy_process_group()
        return 1

    spp_per_launch = int(job.get("spp_per_launch", 1024))
    if spp_per_launch < 1:
        raise SystemExit("spp_per_launch must be >= 1")

    scene = LoadedScene(job["scene"], width, height, asset_root if asset_root is not None else job.get("asset_root"))
    gpu = HipScene(scene.desc, device=local_rank, bvh_builder=(lambda name: ("ploc" if world_size > 1 and scene.n_triangles > 1000000 else "sah") if name == "auto" else name)(job.get("bvh_builder", "auto")))
    gpu.set_integrator(job["integrator"])
    host = _capi.load_host()
    stream = torch.cuda.current_stream().cuda_stream

    # this rank's sums; they keep growing, and are reduced only when an image is due: at the power-of-two
    # checkpoints (src/integrator.cpp:87-92) and at the end.  The union over ranks of what has been rendered is always
    # exactly the samples [0, done).  Like the reference (src/integrator.cpp:42) and like the C++ host
    # (pathed_amd/host/integrator.cpp) the loop renders ALL `spp` samples; numbered files appear at the powers of two only,
    # and when spp is not one the end of the run refreshes auto.exr with all of them -- the same files, sample counts and
    # timings whichever launcher ran the job.
    accum = torch.zeros((height, width, 3), dtype=torch.float32, device="cuda")
    total = torch.zeros_like(accum)
    done = 0
    while done < spp:
        next_power = 1
        while next_power <= done:
            next_power *= 2
        count = min(next_power, spp, done + spp_per_launch * world_size) - done
        begin, mine = parallel.strong_range(rank, world_size, done, count)
        t0 = time.perf_counter()
        if mine > 0:
            gpu.render_device(seed, begin, mine, bounces.start_bounce, bounces.last_bounce, accum.data_ptr(), stream)
        done += count
        checkpoint = (done & (done - 1)) == 0
        if checkpoint or done == spp:   # numbered images at the power-of-two sample counts only (src/integrator.cpp:87-92)
            total.copy_(accum)
            parallel.reduce_to_root(total, root=0)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if rank == 0:
            if checkpoint or done == spp:
                mean = (total / float(done)).cpu().numpy()
                pointer = mean.ctypes.data_as(C.POINTER(C.c_float))
                for name in (("auto.exr", "auto-%05dspp.exr" % done) if checkpoint else ("auto.exr",)):
                    path = os.path.join(out_dir, name)
                    if host.pathed_host_write_exr_half_bgr(path.encode(), width, height, pointer) != 0:
                        raise RuntimeError(host.pathed_host_last_error().decode())
                    print("Saved exr file. [ %s ] " % path)
            print("[%s] sample: %d/%d (%.1fs elapsed)" % (out_dir, done, spp, elapsed), flush=True)

    if world_size > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
