"""The CPU half of the build under sanitizers: the scene / OBJ / MTL / PLY / image loaders,
the host BVH builder (single- and multi-threaded) and the CPU oracle, compiled with AddressSanitizer + UBSan and, in a
second build, ThreadSanitizer (the builder's six threads), over the scenes of the eight image
fixtures (tests/golden/make_image_fixtures.py).  GPU sanitizers are not available on this pool: the kernels are covered by
the bit-exact parity tests instead.  The reference has no sanitizer run and has real data races (one RandomGenerator shared
by all OpenMP threads, src/random_generator.cpp:4-6; std::rand() in Camera::generateRay, src/camera.cpp:51-52)."""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = ["scenes/cornell.json", "scenes/cornell-glass.json", "scenes/cornell-glossy.json", "scenes/cornell-oren-nayar.json",
          "scenes/cornell-ggx.json", "scenes/mis-pbrt.json", "scenes/teapot.json", "test_scenes/environment_map_sampling.json",
          "scenes/cornell-medium.json", "scenes/veach-ajar-available.json"]
SOURCES = ["tools/sanitize_host.cpp", "oracle/oracle.cpp"] + sorted(
    path for path in glob.glob(os.path.join(ROOT, "pathed_amd", "host", "*.cpp"))
    if os.path.basename(path) in ("scene_loader.cpp", "json.cpp", "image_decode.cpp", "exr.cpp", "image.cpp"))


def _build(tmp_path, name, flags):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    binary = str(tmp_path / name)
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fwrapv", "-ffp-contract=off", "-fopenmp", "-Iinclude"] + flags
                           + ["-o", binary] + SOURCES + ["-lz", "-lpthread"], cwd=ROOT, capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr and "cannot find" in build.stderr:
        pytest.skip("sanitizer runtime not available: " + build.stderr[-200:])
    assert build.returncode == 0, build.stderr[-4000:]
    return binary


def _run(binary, env=None):
    result = subprocess.run([binary, ROOT, "20", "16", "2"] + SCENES, capture_output=True, text=True, timeout=900,
                            env=dict(os.environ, OMP_NUM_THREADS="2", **(env or {})))
    assert result.returncode == 0, result.stdout[-2000:] + result.stderr[-6000:]
    assert "sanitize_host: done" in result.stdout and "threaded build identical" in result.stdout
    assert "runtime error" not in result.stderr and "ERROR: AddressSanitizer" not in result.stderr and "WARNING: ThreadSanitizer" not in result.stderr, result.stderr[-6000:]
    return result.stdout


def test_loader_builder_and_oracle_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    binary = _build(tmp_path, "sanitize_asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"])
    output = _run(binary, {"ASAN_OPTIONS": "detect_leaks=1", "UBSAN_OPTIONS": "print_stacktrace=1"})
    assert output.count("checksum") == len(SCENES)


def test_loader_builder_and_oracle_under_thread_sanitizer(tmp_path):
    binary = _build(tmp_path, "sanitize_tsan", ["-fsanitize=thread"])
    # what ThreadSanitizer can judge here is the builder's std::thread pool (six workers over a 257 K-triangle mesh, and
    # whatever the scenes trigger); libgomp is not built with its annotations -- the barrier that ends a parallel region
    # looks like a race to the tool -- so the oracle renders on one OpenMP thread in this build
    _run(binary, {"SANITIZE_ORACLE_THREADS": "1", "TSAN_OPTIONS": "history_size=4"})
