"""Phase 1 of the all-triangles intersector on the matrix pipe (pathed_amd/csrc/mfma_candidates.h).

The pass replaces the VALU's packed Moeller-Trumbore inside test by v_mfma_f32_32x32x2_f32 over Pluecker rows; it only has
to be CONSERVATIVE -- phase 2 (trace.h: intersectTriangle + testLeafTriangle) decides, so hits and images stay those of
the tree walk bit for bit (the stand-in for rtcIntersect1 / rtcOccluded1, reference src/scene.cpp:113,374).  Two checks:
the candidate sets contain everything phase 2 accepts, on rays aimed at the cases where rounding decides (edges, corners,
in-plane directions, origins on triangles and at the camera); and the fused kernel renders the same floats either way.
"""
import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.experiments]   # mfma_candidates.h is compiled into libpathed_hip_experiments.so only


@pytest.fixture(scope="module")
def libs():
    import oracle_lib
    from pathed_amd.integrator import HipScene
    from pathed_amd.scene import LoadedScene
    return oracle_lib, HipScene, LoadedScene


def adversarial_rays(rng, tris, camera, n):
    """(n, 10): origins on triangles (a fifth on an edge or corner) or at the camera; directions random, aimed at points
    of triangles (half of those at edges / corners) or inside the plane of the origin's triangle; shadow rays to triangle points."""
    v0 = tris[:, 0:3].astype(np.float64); e1 = tris[:, 4:7].astype(np.float64); e2 = tris[:, 8:11].astype(np.float64)
    count = tris.shape[0]

    def points(which, edge_share):
        r1 = rng.random(n); r2 = rng.random(n)
        on_edge = rng.random(n) < edge_share
        r2 = np.where(on_edge, 0.0, r2)
        r1 = np.where(on_edge & (rng.random(n) < 0.3), rng.choice([0.0, 1.0], size=n), r1)
        a = 1 - np.sqrt(r1); b = np.sqrt(r1) * (1 - r2)
        return v0[which] + e1[which] * b[:, None] + e2[which] * (1 - a - b)[:, None]

    start = rng.integers(0, count, size=n)
    origin = points(start, 0.2)
    origin = np.where((rng.random(n) < 0.15)[:, None], np.asarray(camera, dtype=np.float64)[None], origin).astype(np.float32)

    def directions(aim_share):
        direction = rng.normal(size=(n, 3))
        target = points(rng.integers(0, count, size=n), 0.5)
        aimed = rng.random(n) < aim_share
        direction = np.where(aimed[:, None], target - origin, direction)
        inplane = e1[start] * rng.normal(size=(n, 1)) + e2[start] * rng.normal(size=(n, 1))
        direction = np.where((rng.random(n) < 0.05)[:, None], inplane, direction)
        length = np.linalg.norm(direction, axis=1, keepdims=True)
        direction = np.where(length > 0, direction / np.maximum(length, 1e-30), [[0.0, 0.0, 1.0]])
        distance = np.linalg.norm(target - origin, axis=1)
        return direction.astype(np.float32), np.where(aimed, distance - 1e-3, rng.choice([1e4, 3e38, 2.0], size=n))

    rays = np.zeros((n, 10), dtype=np.float32)
    rays[:, 0:3] = origin
    rays[:, 3:6], _ = directions(0.5)
    rays[:, 6:9], tfar = directions(0.8)
    rays[:, 9] = np.maximum(tfar, 0.0)
    return rays


def bits(words):
    return int(np.unpackbits(np.ascontiguousarray(words).view(np.uint8)).sum())


def check_conservative(gpu, camera, seed, n=60000):
    _, tris = gpu.export_bvh()
    rays = adversarial_rays(np.random.default_rng(seed), tris, camera, n)
    out = gpu.small_candidates(rays)
    valu_a, valu_b, mfma_a, mfma_b, accept_a, accept_b = (out[:, i] for i in range(6))   # bit p = primitive id p
    assert accept_a.any() and accept_b.any()
    # nothing phase 2 accepts is missing from either phase 1
    assert not (accept_a & ~valu_a).any() and not (accept_b & ~valu_b).any()
    assert not (accept_a & ~mfma_a).any(), int(np.count_nonzero(accept_a & ~mfma_a))
    assert not (accept_b & ~mfma_b).any(), int(np.count_nonzero(accept_b & ~mfma_b))
    return bits(mfma_a) / n, bits(valu_a) / n, bits(mfma_b) / n, bits(valu_b) / n


def test_matrix_pipe_candidates_contain_every_accepted_hit_cornell(libs):
    _, HipScene, LoadedScene = libs
    scene = LoadedScene("scenes/cornell.json", 32, 32)
    gpu = HipScene(scene.desc, device=0)
    counts = check_conservative(gpu, (0.0, 1.0, 6.8), 11)
    # on these adversarial rays the tolerant form keeps more than the VALU form, but not the whole scene
    assert counts[0] < 2.0 * counts[1] + 1.0 and counts[2] < 2.0 * counts[3] + 1.0, counts


@pytest.mark.parametrize("seed", range(300, 308))
def test_matrix_pipe_candidates_contain_every_accepted_hit_fuzz(seed):
    from pathed_amd.integrator import HipScene
    from test_gpu_fuzz import build_scene
    built, desc, scale = build_scene(seed, tiny=True)
    gpu = HipScene(desc, device=0)
    check_conservative(gpu, (0.0, 0.3 * scale, 3.0 * scale), seed)


@pytest.mark.parametrize("scene_path,size,spp,last_bounce", [
    ("scenes/cornell.json", 96, 8, 10),
    ("scenes/mis-pbrt.json", 96, 8, 6),
    ("scenes/cornell-oren-nayar.json", 64, 6, 3),
])
def test_fused_kernel_renders_the_same_floats_with_either_phase_1(libs, scene_path, size, spp, last_bounce):
    _, HipScene, LoadedScene = libs
    scene = LoadedScene(scene_path, size, size)
    valu = HipScene(scene.desc, device=0, small_phase1="valu")
    mfma = HipScene(scene.desc, device=0, small_phase1="mfma")
    assert valu.stats()["path_kernel"] == 3 and mfma.stats()["path_kernel"] == 3
    expected = valu.render(7, 3, spp, 0, last_bounce)
    assert expected.any()
    assert np.array_equal(mfma.render(7, 3, spp, 0, last_bounce), expected)
    generic = HipScene(scene.desc, device=0, small_phase1="mfma", generic_kernels=1)
    assert np.array_equal(generic.render(7, 3, spp, 0, last_bounce), expected)
    mfma.set_samples_per_unit(4)
    valu.set_samples_per_unit(4)
    assert np.array_equal(mfma.render(2, 0, 5, 1, 2), valu.render(2, 0, 5, 1, 2))


@pytest.mark.parametrize("seed", range(200, 206))
def test_fused_kernel_fuzz_scenes_same_floats_with_either_phase_1(seed):
    from pathed_amd.integrator import HipScene
    from test_gpu_fuzz import build_scene
    built, desc, scale = build_scene(seed, tiny=True)
    valu = HipScene(desc, device=0, small_phase1="valu")
    mfma = HipScene(desc, device=0, small_phase1="mfma")
    assert np.array_equal(mfma.render(7, 0, 16, 0, 6), valu.render(7, 0, 16, 0, 6))
