"""Many random tiny scenes (<= 64 triangles): the all-triangles intersector's images against the tree walk's, bit for bit
(tests/test_gpu_fuzz.py runs a dozen seeds; this runs a few hundred):  python tools/fuzz_tiny_scenes.py [first] [count]"""
import os, sys
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import test_gpu_fuzz
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
bad = []
for seed in range(first, first + count):
    try:
        test_gpu_fuzz.test_random_tiny_scene_all_triangles_intersector_equals_tree_walk(seed)
    except AssertionError as error:
        bad.append(seed)
        print("seed", seed, "FAILED", str(error)[:300], flush=True)
    if (seed - first) % 50 == 49:
        print("... %d seeds done, %d failed" % (seed - first + 1, len(bad)), flush=True)
print("seeds %d..%d:" % (first, first + count - 1), "all passed" if not bad else "%d failed: %s" % (len(bad), bad))
