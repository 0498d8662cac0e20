"""The host's file decoders take untrusted input (textures, environment maps): a short mutation-fuzz run
under AddressSanitizer + UBSan on the CPU (tools/fuzz_decoders.cpp) must neither crash nor trip a sanitizer."""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_decoders_survive_mutated_files(tmp_path):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    binary = str(tmp_path / "fuzz_decoders")
    build = subprocess.run(
        ["g++", "-std=c++17", "-O1", "-g", "-fwrapv", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
         "-Iinclude", "-o", binary, "tools/fuzz_decoders.cpp", "pathed_amd/host/image_decode.cpp", "pathed_amd/host/exr.cpp", "-lz"],
        cwd=ROOT, capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("sanitizer runtime not available: " + build.stderr[-200:])
    assert build.returncode == 0, build.stderr
    files = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "textures", "*")))
    result = subprocess.run([binary, "60"] + files, capture_output=True, text=True, timeout=600)
    assert result.returncode == 0, result.stdout[-2000:] + result.stderr[-4000:]
    assert "no crash" in result.stdout
    decoded = int(result.stdout.split()[1].rstrip(","))
    assert decoded > 50          # some mutations leave a decodable file
