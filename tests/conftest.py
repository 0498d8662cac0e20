import os
import sys

import pytest

REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO_ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    config.addinivalue_line("markers", "experiments: needs libpathed_hip_experiments.so (`make experiments`, then "
                            "PATHED_HIP_LIB=pathed_amd/lib/libpathed_hip_experiments.so pytest -m gpu); skipped on the product library")


def has_experiments():
    """True when the HIP library this process loads is the experiments build (include/pathed_hip.h: pathed_hip_has_experiments)."""
    from pathed_amd import _capi
    try:
        return bool(_capi.load_hip().pathed_hip_has_experiments())
    except (OSError, AttributeError):
        return False


def pytest_runtest_setup(item):
    if item.get_closest_marker("experiments") is not None and not has_experiments():
        pytest.skip("the product library carries no experiments (make experiments; PATHED_HIP_LIB=.../libpathed_hip_experiments.so)")


@pytest.fixture(scope="session")
def repo_root():
    return REPO_ROOT


@pytest.fixture(scope="session", autouse=True)
def generated_assets(request):
    """GPU runs: the sized variants of the procedural stand-in mesh (assets/dragon-standin-<n>.ply/.json) are generated
    HERE, by a child process, before any test of the session makes this process's first GPU call."""
    # whenever a GPU test is among the selected ones (with -m gpu, -k ..., or a file name on the command line)
    if any(item.get_closest_marker("gpu") is not None for item in request.session.items):
        import subprocess
        subprocess.run([sys.executable, os.path.join(REPO_ROOT, "tools", "make_assets.py"), "--dragon-variants", "6,7,9,10"],
                       check=True, stdout=subprocess.DEVNULL)
    yield
