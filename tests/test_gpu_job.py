"""BASELINE config 1 plumbing through the C++ host: job.json -> report.json + auto-000NNspp.exr,
with the reference's file conventions (src/job.cpp:33-63, src/integrator.cpp:87-92,
src/image.cpp:21-35, 80-154)."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _read_exr(path):
    from pathed_amd import _capi
    host = _capi.load_host()
    w, h = C.c_int(), C.c_int()
    assert host.pathed_host_read_exr_rgba(path.encode(), C.byref(w), C.byref(h), None, 0) == 0, host.pathed_host_last_error()
    data = np.zeros((h.value, w.value, 4), dtype=np.float32)
    assert host.pathed_host_read_exr_rgba(path.encode(), C.byref(w), C.byref(h), data.ctypes.data_as(C.POINTER(C.c_float)), data.size) == 0
    return data


def test_pathed_executable_runs_config_1(tmp_path):
    import oracle_lib
    from pathed_amd import _capi
    from pathed_amd.scene import LoadedScene

    job = json.load(open(os.path.join(_capi.REPO_ROOT, "jobs", "cornell-c1.json")))
    out_dir = str(tmp_path / "cornell-render")
    job["output_directory"] = out_dir
    job["width"] = job["height"] = 96
    job["spp_per_launch"] = 16
    job_path = str(tmp_path / "job.json")
    json.dump(job, open(job_path, "w"))

    exe = os.path.join(_capi.REPO_ROOT, "pathed_amd", "bin", "pathed")
    result = subprocess.run([exe, job_path, _capi.REPO_ROOT], capture_output=True, text=True, cwd=str(tmp_path))
    assert result.returncode == 0, result.stdout + result.stderr
    # the reference's log line: "[<outdir>/] sample: i/N (X.Xs elapsed)"
    assert "[%s/] sample: 16/16" % out_dir in result.stdout

    report = json.load(open(os.path.join(out_dir, "report.json")))
    assert report["integrator"] == "PathTracer" and report["spp"] == 16
    for spp in (1, 2, 4, 8, 16):  # checkpoints at powers of two
        assert os.path.exists(os.path.join(out_dir, "auto-%05dspp.exr" % spp))
    assert os.path.exists(os.path.join(out_dir, "auto.exr"))

    image = _read_exr(os.path.join(out_dir, "auto-00016spp.exr"))[..., :3]
    blob = open(os.path.join(out_dir, "auto-00016spp.exr"), "rb").read(400)
    assert b"B\x00\x01\x00\x00\x00" in blob and b"G\x00\x01\x00\x00\x00" in blob and b"R\x00\x01\x00\x00\x00" in blob  # HALF channels

    scene = LoadedScene("scenes/cornell.json", 96, 96)
    expected, _ = oracle_lib.OracleScene(scene.desc).render(96, 96, 1, 0, 16, 0, 10, threads=os.cpu_count())
    expected = (expected / 16)[::-1]  # Image::set flips: EXR row 0 is the top scanline
    half = expected.astype(np.float16).astype(np.float32)
    assert np.allclose(image, half, rtol=2e-3, atol=2e-3)
    assert np.mean(np.abs(image - half) > 1e-2 * np.maximum(half, 1e-3)) < 2e-3


def test_python_job_runner_matches_the_executable(tmp_path):
    """pathed_amd.run_job (the multi-GPU entry point, here with one rank) writes the same
    checkpoints as the C++ executable."""
    import sys
    from pathed_amd import _capi

    job = json.load(open(os.path.join(_capi.REPO_ROOT, "jobs", "cornell-c1.json")))
    job["width"] = job["height"] = 64
    job["spp"] = 8
    outputs = {}
    for name in ("cpp", "py"):
        out_dir = str(tmp_path / name)
        job["output_directory"] = out_dir
        job_path = str(tmp_path / (name + ".json"))
        json.dump(job, open(job_path, "w"))
        if name == "cpp":
            command = [os.path.join(_capi.REPO_ROOT, "pathed_amd", "bin", "pathed"), job_path, _capi.REPO_ROOT]
        else:
            command = [sys.executable, "-m", "pathed_amd.run_job", job_path, _capi.REPO_ROOT]
        result = subprocess.run(command, capture_output=True, text=True, cwd=_capi.REPO_ROOT)
        assert result.returncode == 0, result.stdout + result.stderr
        assert "sample: 8/8" in result.stdout
        outputs[name] = _read_exr(os.path.join(out_dir, "auto-00008spp.exr"))
    # the executable renders 8 samples as 1+1+2+4 launches, the runner as 1+1+2+4 too: same sums
    assert np.allclose(outputs["cpp"], outputs["py"], rtol=1e-3, atol=1e-4)


def test_both_launchers_render_every_sample_of_a_count_that_is_no_power_of_two(tmp_path):
    """spp = 11: the reference runs all 11 waves and saves numbered files at 1, 2, 4, 8 (src/integrator.cpp:42, :87-92).
    Both launchers do that, and both refresh auto.exr with all 11 samples at the end: same files, same sample counts."""
    import sys
    from pathed_amd import _capi

    job = json.load(open(os.path.join(_capi.REPO_ROOT, "jobs", "cornell-c1.json")))
    job["width"] = job["height"] = 48
    job["spp"] = 11
    final, numbered = {}, {}
    for name in ("cpp", "py"):
        out_dir = str(tmp_path / name)
        job["output_directory"] = out_dir
        job_path = str(tmp_path / (name + ".json"))
        json.dump(job, open(job_path, "w"))
        command = ([os.path.join(_capi.REPO_ROOT, "pathed_amd", "bin", "pathed")] if name == "cpp" else [sys.executable, "-m", "pathed_amd.run_job"]) + [job_path, _capi.REPO_ROOT]
        result = subprocess.run(command, capture_output=True, text=True, cwd=_capi.REPO_ROOT)
        assert result.returncode == 0, result.stdout + result.stderr
        assert "sample: 11/11" in result.stdout
        assert sorted(f for f in os.listdir(out_dir) if f.endswith("spp.exr")) == ["auto-%05dspp.exr" % n for n in (1, 2, 4, 8)]
        final[name] = _read_exr(os.path.join(out_dir, "auto.exr"))[..., :3]
        numbered[name] = _read_exr(os.path.join(out_dir, "auto-00008spp.exr"))[..., :3]
    assert np.allclose(final["cpp"], final["py"], rtol=1e-3, atol=1e-4)
    # auto.exr holds 11 samples, not the 8 of the last numbered file
    assert not np.allclose(final["cpp"], numbered["cpp"], rtol=1e-3, atol=1e-4)
    # ... namely the mean of samples [0, 11) (HALF precision in the file)
    from pathed_amd.integrator import HipScene
    from pathed_amd.scene import LoadedScene
    scene = LoadedScene(job["scene"], 48, 48)
    mean = HipScene(scene.desc, device=0).render(job.get("seed", 1), 0, 11, job["startBounce"], job["lastBounce"]) / 11.0
    half = mean[::-1].astype(np.float16).astype(np.float32)   # Image::set flips: EXR row 0 is the top scanline
    assert np.allclose(final["cpp"], half, rtol=2e-3, atol=2e-3)


def test_bvh_builder_job_key_changes_the_build_not_the_image(tmp_path):
    """job.json "bvh_builder": the on-GPU PLOC / LBVH builds give the checkpoint the host SAH build gives,
    byte for byte (hits do not depend on the tree); an unknown name is an error."""
    from pathed_amd import _capi
    job = json.load(open(os.path.join(_capi.REPO_ROOT, "jobs", "cornell-c1.json")))
    job["scene"] = "scenes/cornell-glossy.json"   # > 64 triangles: a BVH is built and walked
    job["width"] = job["height"] = 48
    job["spp"] = 4
    exe = os.path.join(_capi.REPO_ROOT, "pathed_amd", "bin", "pathed")
    files = {}
    for builder in ("sah", "ploc", "lbvh", "octree"):
        out_dir = str(tmp_path / builder)
        job["output_directory"] = out_dir
        job["bvh_builder"] = builder
        job_path = str(tmp_path / (builder + ".json"))
        json.dump(job, open(job_path, "w"))
        result = subprocess.run([exe, job_path, _capi.REPO_ROOT], capture_output=True, text=True, cwd=str(tmp_path))
        if builder == "octree":
            assert result.returncode != 0 and "bvh_builder" in (result.stdout + result.stderr)
            continue
        assert result.returncode == 0, result.stdout + result.stderr
        files[builder] = open(os.path.join(out_dir, "auto-00004spp.exr"), "rb").read()
    assert files["sah"] == files["ploc"] == files["lbvh"]


def _read_state(path):
    """<outdir>/auto.state: 48-byte header (magic, w, h, done, startBounce, lastBounce, pad, seed, job digest) + fp32 sums."""
    blob = open(path, "rb").read()
    assert blob[:8] == b"PATHEDS3"
    width, height, done, start, last = np.frombuffer(blob, dtype="<i4", count=5, offset=8)
    sums = np.frombuffer(blob, dtype="<f4", offset=48).reshape(height, width, 3)
    return int(done), sums


def _run_job(tmp_path, name, job):
    from pathed_amd import _capi
    out_dir = str(tmp_path / name)
    job = dict(job, output_directory=out_dir)
    job_path = str(tmp_path / (name + ".json"))
    json.dump(job, open(job_path, "w"))
    exe = os.path.join(_capi.REPO_ROOT, "pathed_amd", "bin", "pathed")
    result = subprocess.run([exe, job_path, _capi.REPO_ROOT], capture_output=True, text=True, cwd=str(tmp_path))
    return out_dir, result


def test_gpus_job_key_fans_the_samples_out_and_sums_them_back(tmp_path):
    """job.json "gpus": the C++ host renders every batch on several scene replicas (one worker thread
    each) and sums their buffers on replica 0.  Two replicas on the one GPU of this box ([0, 0])
    against one: the same samples [0, n) at every checkpoint, so the fp32 sums agree to summation order;
    three replicas leave a ragged split (16 = 6 + 5 + 5)."""
    from pathed_amd import _capi
    job = json.load(open(os.path.join(_capi.REPO_ROOT, "jobs", "cornell-c1.json")))
    job["scene"] = "scenes/cornell-glossy.json"
    job["width"], job["height"], job["spp"] = 80, 64, 16
    single_dir, result = _run_job(tmp_path, "single", job)
    assert result.returncode == 0, result.stdout + result.stderr
    done, single = _read_state(os.path.join(single_dir, "auto.state"))
    assert done == 16
    for name, gpus in (("two", [0, 0]), ("three", [0, 0, 0])):
        out_dir, result = _run_job(tmp_path, name, dict(job, gpus=gpus))
        assert result.returncode == 0, result.stdout + result.stderr
        done, several = _read_state(os.path.join(out_dir, "auto.state"))
        assert done == 16
        rel = np.linalg.norm(several - single) / np.linalg.norm(single)
        assert rel < 1e-6, (name, rel)
        for spp in (1, 2, 4, 8, 16):
            assert os.path.exists(os.path.join(out_dir, "auto-%05dspp.exr" % spp))
        metrics = json.load(open(os.path.join(out_dir, "metrics.json")))
        assert metrics["devices"] == gpus and len(metrics["replica_seconds"]) == len(gpus)
        assert metrics["last_sample"] == 16 and metrics["msamples_per_second"] > 0 and metrics["reduces"] == 5
        # two replicas on ONE device: RCCL takes one rank per device, the host says so and sums through peer copies
        assert metrics["reduce_method"] == "peer-copy" and "RCCL reduce unavailable" in result.stdout
        # what the reference prints per render (src/rtc_manager.cpp:94-115): rays, as rates
        assert metrics["rays_per_sample"] > 2 and metrics["mrays_per_second"] > 0 and 0 < metrics["hbm_roofline_fraction"] < 1
    # a device this box does not have is an error, not a crash
    _, result = _run_job(tmp_path, "absent", dict(job, gpus=[0, 63]))
    assert result.returncode != 0 and "device" in (result.stdout + result.stderr)


def test_resumed_job_continues_bit_identically(tmp_path):
    """"resume": true reloads <outdir>/auto.state (fp32 sums + sample count) and goes on from there.  The random
    stream is a function of (seed, pixel, sample, dimension), so 8 samples now + 8 later are the 16-sample
    run bit for bit; a state file from another seed is refused."""
    from pathed_amd import _capi
    job = json.load(open(os.path.join(_capi.REPO_ROOT, "jobs", "cornell-c1.json")))
    job["width"], job["height"] = 72, 56
    straight_dir, result = _run_job(tmp_path, "straight", dict(job, spp=16))
    assert result.returncode == 0, result.stdout + result.stderr
    first_dir, result = _run_job(tmp_path, "resumed", dict(job, spp=8))
    assert result.returncode == 0, result.stdout + result.stderr
    assert _read_state(os.path.join(first_dir, "auto.state"))[0] == 8
    second_dir, result = _run_job(tmp_path, "resumed", dict(job, spp=16, resume=True))
    assert result.returncode == 0, result.stdout + result.stderr
    assert "resuming at sample 8/16" in result.stdout and "sample: 16/16" in result.stdout and "sample: 4/16" not in result.stdout
    done, resumed = _read_state(os.path.join(second_dir, "auto.state"))
    done_straight, straight = _read_state(os.path.join(straight_dir, "auto.state"))
    assert done == done_straight == 16
    assert np.array_equal(resumed, straight)
    assert open(os.path.join(second_dir, "auto-00016spp.exr"), "rb").read() == open(os.path.join(straight_dir, "auto-00016spp.exr"), "rb").read()
    _, result = _run_job(tmp_path, "resumed", dict(job, spp=32, resume=True, seed=5))
    assert result.returncode != 0 and "seed" in (result.stdout + result.stderr)
    # another scene of the same resolution, or another integrator, is not what the sums hold
    _, result = _run_job(tmp_path, "resumed", dict(job, spp=32, resume=True, scene="scenes/cornell-glossy.json"))
    assert result.returncode != 0 and "another scene" in (result.stdout + result.stderr)
    # a state file that already holds every sample: nothing is rendered, the image is the state's
    before = open(os.path.join(second_dir, "auto.exr"), "rb").read()
    _, result = _run_job(tmp_path, "resumed", dict(job, spp=16, resume=True))
    assert result.returncode == 0 and "nothing to render" in result.stdout
    assert open(os.path.join(second_dir, "auto.exr"), "rb").read() == before
    _, result = _run_job(tmp_path, "bad-launch", dict(job, spp_per_launch=0))
    assert result.returncode != 0 and "spp_per_launch" in (result.stdout + result.stderr)


def test_rccl_reduce_through_the_c_abi():
    """pathed_hip_comm_*: librccl loaded on demand, ncclCommInitAll over the devices of one process, ONE ncclReduce
    (sum, fp32, root = device_ids[0]) -- here with the one device of this box, so RCCL itself has executed on hardware
    before the 8-GPU node does it (reference src/integrator.cpp:42-51).  Two ranks on one device are
    refused with a message (the host then falls back to peer copies)."""
    import torch
    from pathed_amd import _capi
    lib = _capi.load_hip()
    devices = (C.c_int * 1)(0)
    comm = C.c_void_p()
    assert lib.pathed_hip_comm_init(1, devices, C.byref(comm)) == 0, lib.pathed_hip_last_error()
    send = torch.arange(3 * 64 * 48, dtype=torch.float32, device="cuda") * 0.25
    total = torch.zeros_like(send)
    torch.cuda.synchronize()
    pointers = (C.c_void_p * 1)(send.data_ptr())
    assert lib.pathed_hip_comm_reduce(comm, pointers, total.data_ptr(), send.numel()) == 0, lib.pathed_hip_last_error()
    assert torch.equal(total, send)
    # in place on the root
    assert lib.pathed_hip_comm_reduce(comm, pointers, send.data_ptr(), send.numel()) == 0, lib.pathed_hip_last_error()
    assert torch.equal(total, send)
    lib.pathed_hip_comm_destroy(comm)
    twice = (C.c_int * 2)(0, 0)
    assert lib.pathed_hip_comm_init(2, twice, C.byref(comm)) == -4 and b"one rank per device" in lib.pathed_hip_last_error()
    absent = (C.c_int * 1)(63)
    assert lib.pathed_hip_comm_init(1, absent, C.byref(comm)) == -1


def test_bench_runs_as_a_one_rank_rccl_process_group():
    """bench.py --dist-single: the N > 1 code path of the benchmark -- init_process_group("nccl"), the barrier, the
    reduce to rank 0 inside the timed region, the all_gather of the per-rank times -- executed with world size 1, so
    the driver's 8-GPU scaling run is not the first time RCCL and that path run."""
    import sys
    from pathed_amd import _capi
    command = [sys.executable, os.path.join(_capi.REPO_ROOT, "bench.py"), "--dist-single", "--steps", "2", "--warmup", "1",
               "--spp-per-step", "8", "--width", "96", "--height", "64", "--no-cpu-baseline", "--no-large-bvh"]
    result = subprocess.run(command, capture_output=True, text=True, timeout=600)
    assert result.returncode == 0, result.stdout + result.stderr
    line = json.loads(result.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["collective"] == "nccl (RCCL), world size 1"
    assert len(line["per_rank_s"]["reduce"]) == 1 and 0.05 < line["image_mean_rgb"][0] < 0.4


def test_bench_rehearses_many_ranks_on_one_gpu():
    """The driver's 2/4/8-GPU scaling run, rehearsed with what one box allows: `bench.py --gpus 5 --share-gpu --backend gloo`
    starts five child ranks on device 0 (a box admits six processes on its card and this test process is one of them, so
    five is the most a test may start), every step's samples are split five ways (ragged: 1024 = 4 x 205 + 204), the sums are
    reduced to rank 0 inside the timed region and the per-rank times gathered.  The reduced image is the single-rank image
    up to fp32 summation order (additive waves: reference src/integrator.cpp:42-51)."""
    import sys
    from pathed_amd import _capi
    common = ["--steps", "1", "--warmup", "1", "--spp-per-step", "1024", "--width", "64", "--height", "48",
              "--no-cpu-baseline", "--no-large-bvh", "--no-kernel-timing"]
    bench = os.path.join(_capi.REPO_ROOT, "bench.py")
    many = subprocess.run([sys.executable, bench, "--gpus", "5", "--share-gpu", "--backend", "gloo"] + common,
                          capture_output=True, text=True, timeout=900)
    assert many.returncode == 0, many.stdout + many.stderr
    line = json.loads(many.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 5 and line["scaling"] == "strong" and line["config"]["collective"] == "gloo (host-staged), world size 5"
    for key in ("render", "reduce", "total", "setup"):
        assert len(line["per_rank_s"][key]) == 5 and all(value >= 0 for value in line["per_rank_s"][key]), key
    assert line["config"]["total_samples"] == 64 * 48 * 1024
    one = subprocess.run([sys.executable, bench] + common, capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stdout + one.stderr
    single = json.loads(one.stdout.strip().splitlines()[-1])
    for a, b in zip(line["image_mean_rgb"], single["image_mean_rgb"]):
        assert abs(a - b) <= 1e-5 * abs(b), (line["image_mean_rgb"], single["image_mean_rgb"])


def test_bench_line_carries_the_contract_fields():
    """The ONE JSON line of bench.py on the configuration it is quoted on (scenes/cornell.json 1024 x 1024; two short steps here):
    the driver's keys, a `roofline` for the dominant kernel -- bound, achieved, peak, unit, frac, traffic, and since round 5
    `useful_frac` (x the lane utilisation of the counter pass) and `mix_aware` (against what the kernel's own instruction mix can
    issue) -- and a `cpu_baseline` that states its cores, the host's quota and the rate at every thread count it tried."""
    import sys
    from pathed_amd import _capi
    command = [sys.executable, os.path.join(_capi.REPO_ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--spp-per-step", "64", "--no-large-bvh"]
    result = subprocess.run(command, capture_output=True, text=True, timeout=900)
    assert result.returncode == 0, result.stdout + result.stderr
    lines = [text for text in result.stdout.strip().splitlines() if text.startswith("{")]
    assert len(lines) == 1                                               # ONE line
    line = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in line, key
    assert line["unit"] == "Msamples/s" and line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1
    assert line["higher_is_better"] is True and line["vs_baseline"] is None and line["dtype"] == "f32"
    assert "workload" in line["config"] and "model" not in line["config"]
    assert line["environment_overrides"] == {} and line["library"]["experiments_build"] is False
    roofline = line["roofline"]
    assert roofline["bound"] == "valu" and roofline["unit"] == "G wave-instr/s" and roofline["peak"] == 1228.8
    assert abs(roofline["frac"] - roofline["achieved"] / roofline["peak"]) < 1e-9 and 0.2 < roofline["frac"] < 1.0
    assert abs(roofline["useful_frac"] - roofline["frac"] * roofline["lane_utilisation"]) < 1e-9
    assert 0.5 < roofline["mix_aware"]["frac"] <= 1.05 and roofline["mix_aware"]["bound"] < roofline["peak"]
    assert roofline["traffic"] is None or roofline["traffic"] > 0
    # (whether the committed counter pass / static mix are of THESE kernel sources is stated in the line, not hidden)
    assert roofline["instructions_source"]["stale"] in (True, False) and roofline["mix_aware"]["source"]["stale"] in (True, False)
    cpu = line["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["value"] > 0 and cpu["unit"] == "Msamples/s" and cpu["cores"] >= 1 and cpu["sample"]
    assert cpu["cores"] == cpu["threads"] <= cpu["host"]["usable"] and len(cpu["thread_counts_tried"]) >= 1
    assert all(entry["spp"] >= 2 for entry in cpu["thread_counts_tried"])
    assert max(entry["Msamples_per_s"] for entry in cpu["thread_counts_tried"]) > 0
