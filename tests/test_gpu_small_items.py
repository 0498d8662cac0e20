"""Phase 1 of the all-triangles intersector as the fused kernel runs it (pathed_amd/csrc/small_items.h, kernels.h:
smallCandidatesItems): two triangles that form a parallelogram -- the two halves of a quad, which is what the scenes of this
size are made of (reference src/quad.cpp:27-151, the (0,1,2),(0,2,3) split of src/obj_parser.cpp) -- are ONE
Moeller-Trumbore evaluation with tolerances, the rest keep the exact pair-of-triangles test.  It only has to be
CONSERVATIVE (phase 2 decides; stands in for rtcIntersect1 / rtcOccluded1, reference src/scene.cpp:113,374): everything
phase 2 accepts must be a candidate, on rays aimed at the cases where rounding decides."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def libs():
    import oracle_lib
    from pathed_amd.integrator import HipScene
    from pathed_amd.scene import LoadedScene
    return oracle_lib, HipScene, LoadedScene


def check(gpu, camera, seed, n=60000):
    from test_gpu_mfma_phase1 import adversarial_rays
    _, tris = gpu.export_bvh()
    rays = adversarial_rays(np.random.default_rng(seed), tris, camera, n)
    out = gpu.small_candidates(rays)
    pairs_a, pairs_b, accept_a, accept_b, items_a, items_b = out[:, 0], out[:, 1], out[:, 4], out[:, 5], out[:, 6], out[:, 7]
    assert accept_a.any() and accept_b.any()
    assert not (accept_a & ~items_a).any(), int(np.count_nonzero(accept_a & ~items_a))
    assert not (accept_b & ~items_b).any(), int(np.count_nonzero(accept_b & ~items_b))
    count = lambda words: int(np.unpackbits(np.ascontiguousarray(words).view(np.uint8)).sum())
    return count(items_a) / n, count(pairs_a) / n, count(items_b) / n, count(pairs_b) / n


@pytest.mark.parametrize("scene_path,camera,quads", [
    ("scenes/cornell.json", (0.0, 1.0, 6.8), 18),
    ("scenes/mis-pbrt.json", None, None),
    ("scenes/cornell-medium.json", None, None),
])
def test_item_candidates_contain_every_accepted_hit(libs, scene_path, camera, quads):
    _, HipScene, LoadedScene = libs
    scene = LoadedScene(scene_path, 32, 32)
    gpu = HipScene(scene.desc, device=0)
    if camera is None:
        camera = tuple(scene.desc.contents.camera.origin)
    counts = check(gpu, camera, 5)
    # the tolerant form keeps more than the exact one on these adversarial rays (a tenth of them graze the quad they leave,
    # half of the shadow rays end 1e-3 short of a triangle), never the whole scene
    assert counts[0] <= 1.5 * counts[1] + 1.0 and counts[2] <= 1.5 * counts[3] + 1.0, counts
    # ... and from another camera position (the tolerances scale with the distance a ray may start at)
    import copy
    import ctypes as C
    far = copy.copy(scene.desc.contents.camera)
    far.origin = (C.c_float * 3)(camera[0] * 3.0 + 10.0, camera[1] - 20.0, camera[2] * 5.0)
    gpu.set_camera(far)
    check(gpu, tuple(far.origin), 6)


@pytest.mark.parametrize("seed", range(400, 410))
def test_item_candidates_on_random_scenes_with_quads(seed):
    """Random triangle soups (needles, duplicates, zero-area triangles, scales 0.1 .. 100) with random parallelograms mixed
    in -- some exact, some trapezoids (planar, a few per cent off a parallelogram, like the walls of the Cornell box), some a few
    ulps off, some folded so that they must NOT pair."""
    from pathed_amd import _capi
    from pathed_amd.integrator import HipScene
    from scene_builder import BuiltScene
    rng = np.random.default_rng(seed)
    scale = float(10.0 ** rng.uniform(-1.0, 2.0))
    built = BuiltScene(40, 40, (0, 0.3 * scale, 3.0 * scale), (0, 0, 0), fov_degrees=50.0)
    grey = built.material(_capi.MAT_LAMBERTIAN, diffuse=(0.5, 0.5, 0.5))
    light = built.material(_capi.MAT_LAMBERTIAN, diffuse=(0, 0, 0), emit=(5.0, 5.0, 5.0))
    n_quads = int(rng.integers(4, 14))
    for k in range(n_quads):
        p0 = rng.normal(size=3) * scale
        a1 = rng.normal(size=3) * scale * 10.0 ** rng.uniform(-1.5, 0.0)
        a2 = rng.normal(size=3) * scale * 10.0 ** rng.uniform(-1.5, 0.0)
        p1, p3 = p0 + a1, p0 + a2
        p2 = p0 + a1 + a2
        if k % 5 == 2:
            p2 = p0 + a1 * float(rng.uniform(0.97, 1.03)) + a2 * float(rng.uniform(0.97, 1.03))   # a planar quad that is no parallelogram
        if k % 5 == 3:
            p2 = p2 + rng.normal(size=3) * 3e-7 * scale          # a few ulps off a parallelogram
        if k % 5 == 4:
            p3 = p0 + a1 * 0.3 - a2                               # folded: two triangles on one side of the diagonal
        order = [(0, 1, 2), (0, 2, 3)] if k % 2 == 0 else [(1, 2, 0), (3, 0, 2)]   # any vertex may be a triangle's v0
        built.mesh(np.array([p0, p1, p2, p3]), order, grey)
    n_lone = int(rng.integers(4, 62 - 2 * n_quads))
    corners = rng.normal(size=(n_lone, 3, 3)) * scale * 0.5
    corners[0, 2] = corners[0, 1]
    if n_lone > 3:
        corners[2] = corners[3]
    built.mesh(corners.reshape(-1, 3), np.arange(3 * n_lone).reshape(-1, 3), grey)
    top = 2.5 * scale
    built.quad([(-scale, top, -scale), (scale, top, -scale), (scale, top, scale), (-scale, top, scale)], light)
    desc = built.finish()
    assert desc.contents.n_triangles <= 64
    gpu = HipScene(desc, device=0)
    assert gpu.stats()["scene_in_lds"] == 2 and gpu.stats()["path_kernel"] == 3
    check(gpu, (0, 0.3 * scale, 3.0 * scale), seed, n=40000)
    # the fused kernel over items = the fused kernel over lone triangles only (generic_kernels) = the tree walk, bit for bit
    image = gpu.render(7, 0, 16, 0, 6)
    assert np.array_equal(HipScene(desc, device=0, generic_kernels=1).render(7, 0, 16, 0, 6), image)
    assert np.array_equal(HipScene(desc, device=0, intersector="bvh").render(7, 0, 16, 0, 6), image)


@pytest.mark.parametrize("sheets,gap", [(31, 0.02), (12, 0.05), (4, 0.1)])
def test_stacked_sheets_fill_and_overflow_the_shared_resolve(sheets, gap):
    """Every camera ray crosses a stack of parallel quads: with 31 sheets each lane holds dozens of phase-2 candidates, so the
    wave's list of left-overs (kernels.h smallResolveShared, 128 items) overflows and the owners resolve in place; with 12 it
    takes two helping turns; with 4 one.  Same image as the generic kernels (lone triangles, in-place resolve) and the tree walk."""
    from pathed_amd import _capi
    from pathed_amd.integrator import HipScene
    from scene_builder import BuiltScene
    built = BuiltScene(48, 48, (0.1, 0.2, 4.0), (0, 0, 0), fov_degrees=40.0)
    light = built.material(_capi.MAT_LAMBERTIAN, diffuse=(0, 0, 0), emit=(8.0, 8.0, 8.0))
    for k in range(sheets):
        z = -gap * k
        shade = 0.2 + 0.6 * ((k * 7) % 10) / 10.0
        sheet = built.material(_capi.MAT_LAMBERTIAN, diffuse=(shade, 0.5, 1.0 - shade))
        built.quad([(-1.0, -1.0, z), (1.0, -1.0, z), (1.0, 1.0, z), (-1.0, 1.0, z)], sheet)
    built.quad([(-3.0, -3.0, 6.0), (-3.0, 3.0, 6.0), (3.0, 3.0, 6.0), (3.0, -3.0, 6.0)], light)   # behind the camera, facing the stack
    desc = built.finish()
    assert desc.contents.n_triangles == 2 * sheets + 2 <= 64
    gpu = HipScene(desc, device=0)
    assert gpu.stats()["scene_in_lds"] == 2 and gpu.stats()["path_kernel"] == 3
    image = gpu.render(11, 0, 24, 0, 5)
    assert image.any()
    assert np.array_equal(HipScene(desc, device=0, generic_kernels=1).render(11, 0, 24, 0, 5), image)
    assert np.array_equal(HipScene(desc, device=0, intersector="bvh", shade_kernel="per-slot").render(11, 0, 24, 0, 5), image)
    assert np.array_equal(HipScene(desc, device=0, intersector="bvh", shade_kernel="wave").render(11, 0, 24, 0, 5), image)


@pytest.mark.parametrize("seed,n_spheres", [(1, 16), (2, 15), (3, 7), (4, 1)])
def test_sphere_candidates_of_the_fused_kernel_lose_no_hit(seed, n_spheres):
    """Spheres of the fused kernel: a packed line-misses-sphere test picks the candidates (kernels.h smallSphereCandidates), the
    exact tests run on the wave's shared list.  Spheres from a thousandth of the scene to one that holds the camera,
    overlapping, emissive and not, glass and mirror among them (rays start ON spheres, inside them, graze them): the image of
    the tree walk, whose leaves go through testSphere alone, bit for bit."""
    from pathed_amd import _capi
    from pathed_amd.integrator import HipScene
    from scene_builder import BuiltScene
    rng = np.random.default_rng(seed)
    built = BuiltScene(56, 40, (0.2, 0.8, 5.0), (0, 0.5, 0), fov_degrees=45.0)
    grey = built.material(_capi.MAT_LAMBERTIAN, diffuse=(0.6, 0.6, 0.6))
    kinds = [built.material(_capi.MAT_LAMBERTIAN, diffuse=(0.7, 0.3, 0.2)), built.material(_capi.MAT_GLASS, ior=1.5), built.material(_capi.MAT_MIRROR),
             built.material(_capi.MAT_PLASTIC, diffuse=(0.2, 0.4, 0.6), alpha=0.1),
             built.material(_capi.MAT_LAMBERTIAN, diffuse=(0, 0, 0), emit=(9.0, 8.0, 6.0))]
    built.quad([(-4, 0, 4), (4, 0, 4), (4, 0, -4), (-4, 0, -4)], grey)
    built.quad([(-4, 0, -4), (4, 0, -4), (4, 5, -4), (-4, 5, -4)], grey)
    built.quad([(-1, 4.5, 1), (-1, 4.5, -1), (1, 4.5, -1), (1, 4.5, 1)], kinds[4])
    for k in range(n_spheres):
        if k == 0 and n_spheres > 8:
            centre, radius, material = (0.2, 0.8, 5.0), 1.5, kinds[1]       # the camera sits inside a glass sphere
        elif k == 1 and n_spheres > 8:
            centre, radius, material = (0.0, 1.0, 0.0), 4e-3, kinds[4]      # a speck of light
        else:
            centre = tuple(rng.uniform(-2.5, 2.5, 3) * (1.0, 0.5, 1.0) + (0.0, 1.2, 0.0))
            radius = float(10.0 ** rng.uniform(-1.5, 0.1))
            material = kinds[int(rng.integers(0, len(kinds)))]
        built.sphere(centre, radius, material)
    desc = built.finish()
    fused = HipScene(desc, device=0)
    assert fused.stats()["scene_in_lds"] == 2 and fused.stats()["path_kernel"] == 3
    image = fused.render(3, 0, 24, 0, 8)
    assert image.any()
    assert np.array_equal(HipScene(desc, device=0, intersector="bvh", shade_kernel="per-slot").render(3, 0, 24, 0, 8), image)
    per_slot = HipScene(desc, device=0, shade_kernel="per-slot")
    assert np.array_equal(per_slot.render(3, 0, 24, 0, 8), image)
    # (a camera inside glass, a speck of light: some samples are not finite and are dropped -- by every kernel alike)
    assert fused.stats()["dropped_samples"] == per_slot.stats()["dropped_samples"]
