"""bench.py's launcher side, which must work without a GPU and without touching one: device counting from the environment
and from sysfs, the argument contract the driver relies on."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_visible_gpus_reads_the_environment_before_sysfs(monkeypatch):
    import bench
    for name in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(name, raising=False)
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2,3")
    assert bench.visible_gpus() == 4
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_gpus() == 0
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "2")
    assert bench.visible_gpus() == 1
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES")
    count = bench.visible_gpus()          # sysfs, or None where there is no KFD (this container)
    assert count is None or count >= 1


def test_the_launcher_parent_never_imports_torch():
    """`bench.py --gpus N` starts its ranks from a parent that makes no GPU call: it must not even import torch (whose
    device_count goes through amdsmi / HIP depending on the build).  Asking for more GPUs than the box shows is refused
    before anything is started."""
    code = ("import sys, bench; sys.argv = ['bench.py', '--gpus', '64']; "
            "import os; os.environ['HIP_VISIBLE_DEVICES'] = '0'\n"
            "try:\n    bench.main()\nexcept SystemExit as error:\n    print('exit:', error)\n"
            "print('torch imported:', 'torch' in sys.modules)")
    env = {key: value for key, value in os.environ.items() if key not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    result = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, env=env, timeout=120)
    assert result.returncode == 0, result.stderr
    assert "this machine shows 1 GPU(s)" in result.stdout and "torch imported: False" in result.stdout


def test_default_arguments_are_the_baseline_configuration():
    import bench
    sys_argv = sys.argv
    try:
        sys.argv = ["bench.py"]
        args = bench.parse_args()
    finally:
        sys.argv = sys_argv
    assert (args.gpus, args.scene, args.width, args.height, args.spp_per_step, args.steps) == (1, "scenes/cornell.json", 1024, 1024, 1024, 4)
    assert args.scaling == "strong" and args.backend == "nccl" and args.large_bvh_subdiv == 10
