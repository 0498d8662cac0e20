"""The C-ABI shared library: loads, exports every symbol include/pathed_hip.h declares, and
fails loudly (no fallback) when there is no GPU.  No compute calls here."""
import ctypes as C
import os
import re

import pytest

from pathed_amd import _capi


def _declared_symbols():
    header = open(os.path.join(_capi.REPO_ROOT, "include", "pathed_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    return sorted(set(re.findall(r"\b(pathed_hip_[a-z_]+)\s*\(", header)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(_capi.HIP_SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(_capi.hip_library_path())
    for name in _declared_symbols():
        assert hasattr(lib, name), "libpathed_hip.so does not export %s" % name


def test_struct_sizes_match_the_header():
    # sizes the C compiler gives the header's structs (checked here so the ctypes mirror cannot drift)
    import subprocess
    import tempfile
    source = (
        '#include "pathed_hip.h"\n#include <stdio.h>\n'
        'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(PathedCamera), sizeof(PathedMaterial),'
        ' sizeof(PathedSphere), sizeof(PathedGeom), sizeof(PathedEnvLight), sizeof(PathedSceneDesc), sizeof(PathedStats),'
        ' sizeof(PathedSceneOptions));return 0;}\n'
    )
    with tempfile.TemporaryDirectory() as tmp:
        c_file = os.path.join(tmp, "sizes.c")
        open(c_file, "w").write(source)
        exe = os.path.join(tmp, "sizes")
        subprocess.run(["gcc", "-I", os.path.join(_capi.REPO_ROOT, "include"), c_file, "-o", exe], check=True)
        sizes = [int(x) for x in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()]
    expected = [C.sizeof(t) for t in (_capi.PathedCamera, _capi.PathedMaterial, _capi.PathedSphere, _capi.PathedGeom,
                                      _capi.PathedEnvLight, _capi.PathedSceneDesc, _capi.PathedStats, _capi.PathedSceneOptions)]
    assert sizes == expected


def test_version_string():
    # the string carries the header's ABI number: a host that checks it cannot accept a library of another struct layout
    lib = _capi.load_hip()
    lib.pathed_hip_version.restype = C.c_char_p
    version = lib.pathed_hip_version()
    header = open(os.path.join(_capi.REPO_ROOT, "include", "pathed_hip.h")).read()
    abi = int(re.search(r"#define PATHED_ABI_VERSION (\d+)", header).group(1))
    assert b"gfx950" in version and ("abi %d)" % abi).encode() in version
    assert abi == _capi.PATHED_ABI_VERSION


def test_no_gpu_means_a_loud_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from pathed_amd.integrator import HipScene, PathedError
    from pathed_amd.scene import LoadedScene
    scene = LoadedScene("scenes/cornell.json", 8, 8)
    with pytest.raises(PathedError, match="no HIP device|pathed_hip"):
        HipScene(scene.desc, device=0)


def test_product_never_imports_the_oracle():
    # the oracle is test infrastructure: nothing under pathed_amd/ may reference it
    offenders = []
    for root, _, files in os.walk(os.path.join(_capi.REPO_ROOT, "pathed_amd")):
        for name in files:
            if name.endswith((".py", ".h", ".cpp", ".hip")):
                text = open(os.path.join(root, name), errors="ignore").read()
                if re.search(r"oracle_lib|liboracle|oracle/|#include\s+\"oracle", text):
                    offenders.append(os.path.join(root, name))
    assert offenders == []


def test_product_library_reads_no_tuning_variable_from_the_environment():
    """35 getenv("PATHED_*") in the product library let a stray variable on a bench box change kernels, slot counts
    or the builder silently.  Every setting travels in PathedSceneOptions now; the environment is read by the experiments
    build only (tuningEnv in pathed_hip.hip), apart from two debug prints that change no result."""
    import re
    from pathed_amd import _capi
    source = open(os.path.join(_capi.REPO_ROOT, "pathed_amd", "csrc", "pathed_hip.hip")).read()
    assert source.count("getenv(") <= 3
    blob = open(_capi.hip_library_path(), "rb").read()
    if b"libpathed_hip_experiments" in os.path.basename(_capi.hip_library_path()).encode():
        return
    names = set(re.findall(rb"PATHED_[A-Z0-9_]+", blob))
    assert names <= {b"PATHED_DEBUG_ALLOC", b"PATHED_DEBUG_STATS"}, names


def test_bench_refuses_to_run_with_overrides_in_the_environment():
    import subprocess
    import sys
    from pathed_amd import _capi
    env = dict(os.environ, PATHED_MAX_SLOTS="4096")
    result = subprocess.run([sys.executable, os.path.join(_capi.REPO_ROOT, "bench.py"), "--steps", "1"], env=env, capture_output=True, text=True)
    assert result.returncode == 3 and "PATHED_MAX_SLOTS" in result.stderr and "--allow-overrides" in result.stderr
