import os, sys
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import test_gpu_fuzz
bad = []
for seed in range(6, 126):
    try:
        test_gpu_fuzz.test_random_scene_parity(seed)
    except AssertionError as e:
        bad.append((seed, str(e)[:200]))
        print("seed", seed, "FAILED", str(e)[:200], flush=True)
print("seeds 6..125:", "all passed" if not bad else "%d failed" % len(bad))
