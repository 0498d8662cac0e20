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
    expected, _ = oracle_lib.OracleScene(scene.desc).render(96, 96, 1, 0, 16, 0, 10, threads=os.cpu_count(), chunk=4)
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
