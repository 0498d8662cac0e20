import os
import sys

import pytest

REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO_ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


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
        subprocess.run([sys.executable, os.path.join(REPO_ROOT, "tools", "make_assets.py"), "--dragon-variants", "6,7,9"],
                       check=True, stdout=subprocess.DEVNULL)
    yield
