"""k_path_hybrid (pathed_amd/csrc/path_hybrid.h): scenes of 65 .. 4096 triangles -- the reference's cornell-glossy / cornell-glass
(1 112 triangles) are the ones in its repository -- split into a DIRECT set of at most 64 large triangles, tested by the
all-items intersector of the <= 64-triangle kernel, and a TREE part with a BVH of its own that a ray walks only when its
segment meets the part's box.  Every triangle still goes through intersectTriangle with the ray's own origin and direction
and the acceptance rule does not depend on the order candidates arrive in, so the test is the strongest there is: the image
is the BVH kernels' BIT FOR BIT (whose hits are pinned bit-exact against the oracle, tests/test_gpu_parity.py), whatever the
split.  Stands in for rtcIntersect1 / rtcOccluded1 (reference src/scene.cpp:113, :374)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(desc, seed, spp, last_bounce=8, **options):
    from pathed_amd.integrator import HipScene
    hybrid = HipScene(desc, device=0, **options)
    walk = HipScene(desc, device=0, shade_kernel="per-slot", intersector="bvh")
    image = hybrid.render(seed, 0, spp, 0, last_bounce)
    expected = walk.render(seed, 0, spp, 0, last_bounce)
    return hybrid, walk, image, expected


@pytest.mark.parametrize("scene_path,size,spp", [
    ("scenes/cornell-glossy.json", 160, 12),     # mirror ball + mirror cube in the box
    ("scenes/cornell-glass.json", 160, 12),      # glass ball
    ("scenes/cornell-glossy.json", 17, 3),       # ragged: fewer paths than a block has lanes
])
def test_hybrid_kernel_renders_the_tree_walks_image_bit_for_bit(scene_path, size, spp):
    from pathed_amd.scene import LoadedScene
    scene = LoadedScene(scene_path, size, size)
    hybrid, walk, image, expected = _both(scene.desc, 5, spp)
    assert hybrid.stats()["path_kernel"] == 7 and walk.stats()["path_kernel"] == 1     # the default on such a scene
    assert expected.any() and np.array_equal(image, expected)
    assert hybrid.stats()["dropped_samples"] == 0
    # batches continue the sum; a window of bounces; the unbounded path length
    assert np.array_equal(hybrid.render(5, spp, 4, 0, 8), walk.render(5, spp, 4, 0, 8))
    assert np.array_equal(hybrid.render(9, 0, 4, 2, 3), walk.render(9, 0, 4, 2, 3))
    assert np.array_equal(hybrid.render(9, 0, 2, 0, -1), walk.render(9, 0, 2, 0, -1))
    # counting is the wavefront kernels': same image, their counters
    hybrid.set_stats_mode(count=True)
    hybrid.reset_stats()
    assert np.array_equal(hybrid.render(5, 0, spp, 0, 8), expected)
    stats = hybrid.stats()
    assert stats["path_kernel"] == 1 and stats["closest_rays"] > 0


def _soup(seed, n_small, n_big, with_quads=True):
    """A few large triangles / quads around a cloud of small ones: the shape the hybrid split is made for; and the awkward
    cases mixed in (duplicates across the two parts, zero-area triangles, triangles the size of the threshold)."""
    from pathed_amd import _capi
    from scene_builder import BuiltScene
    rng = np.random.default_rng(seed)
    built = BuiltScene(72, 56, (0.2, 1.0, 6.5), (0, 0.9, 0), fov_degrees=38.0)
    grey = built.material(_capi.MAT_LAMBERTIAN, diffuse=(0.7, 0.7, 0.65))
    red = built.material(_capi.MAT_OREN_NAYAR, diffuse=(0.6, 0.1, 0.1), sigma=0.4)
    shiny = built.material(_capi.MAT_PLASTIC, diffuse=(0.1, 0.3, 0.5), alpha=0.1)
    glass = built.material(_capi.MAT_GLASS, ior=1.5)
    mirror = built.material(_capi.MAT_MIRROR)
    light = built.material(_capi.MAT_LAMBERTIAN, diffuse=(0, 0, 0), emit=(14, 11, 6))
    if with_quads:
        built.quad([(-2, 0, -2), (2, 0, -2), (2, 0, 2), (-2, 0, 2)], grey)                 # floor
        built.quad([(-2, 0, -2), (-2, 2.4, -2), (2, 2.4, -2), (2, 0, -2)], red)           # back wall
        built.quad([(-2, 0, 2), (-2, 2.4, 2), (-2, 2.4, -2), (-2, 0, -2)], shiny)         # left wall
        built.quad([(-0.5, 2.39, -0.5), (0.5, 2.39, -0.5), (0.5, 2.39, 0.5), (-0.5, 2.39, 0.5)], light)
    else:
        built.quad([(-3, 3.5, -3), (3, 3.5, -3), (3, 3.5, 3), (-3, 3.5, 3)], light)
    # big lone triangles of random sizes (some end up direct, some in the tree)
    if n_big:
        corners = rng.normal(size=(n_big, 3, 3)) * rng.uniform(0.3, 1.5, size=(n_big, 1, 1)) + rng.normal(size=(n_big, 1, 3)) * 0.8 + (0, 1, 0)
        built.mesh(corners.reshape(-1, 3).astype(np.float32), np.arange(3 * n_big).reshape(-1, 3), mirror if seed % 2 else grey)
    # the cloud
    centres = rng.normal(size=(n_small, 1, 3)) * 0.45 + (0.3, 0.9, 0.2)
    corners = centres + rng.normal(size=(n_small, 3, 3)) * 10.0 ** rng.uniform(-2.2, -0.8, size=(n_small, 1, 1))
    corners[0:3, 2] = corners[0:3, 1]                           # zero-area
    if n_small > 40:
        corners[10:20] = corners[20:30]                         # exact duplicates inside the tree part (lower primitive id wins)
    vertices = corners.reshape(-1, 3).astype(np.float32)
    normals = np.zeros_like(vertices)
    smooth = rng.random(n_small) < 0.4
    face_normal = np.cross(corners[:, 1] - corners[:, 0], corners[:, 2] - corners[:, 0])
    for k in np.nonzero(smooth)[0]:
        normals[3 * k:3 * k + 3] = face_normal[k] / max(np.linalg.norm(face_normal[k]), 1e-30) + rng.normal(size=(3, 3)) * 0.1
    third = n_small // 3
    for first, last, material in ((0, third, glass), (third, 2 * third, shiny), (2 * third, n_small, light if seed % 3 == 0 else red)):
        picked = np.arange(3 * first, 3 * last)
        if picked.size:
            built.mesh(vertices[picked], np.arange(picked.size).reshape(-1, 3), material, normals=normals[picked])
    return built, built.finish()


@pytest.mark.parametrize("seed,n_small,n_big,with_quads", [
    (1, 60, 0, True),        # 68 triangles: just over the all-triangles kernel's 64
    (2, 100, 12, True),
    (3, 700, 30, True),      # emissive triangles inside the tree part
    (4, 1500, 80, True),     # more large triangles than the direct set holds
    (5, 400, 0, False),      # nearly everything in the tree
    (6, 3000, 40, True),
    (7, 57, 6, False),       # 65 triangles
])
def test_random_splits_give_the_tree_walks_image(seed, n_small, n_big, with_quads):
    built, desc = _soup(seed, n_small, n_big, with_quads)
    hybrid, walk, image, expected = _both(desc, 20 + seed, 8)
    assert hybrid.stats()["path_kernel"] == 7
    assert expected.any() and np.array_equal(image, expected), (seed, float(np.abs(image - expected).max()))
    assert hybrid.stats()["dropped_samples"] == walk.stats()["dropped_samples"]
    # the generic instantiation of the same kernel
    generic, _, image, _ = _both(desc, 20 + seed, 8, generic_kernels=1)
    assert generic.stats()["path_kernel"] == 7 and np.array_equal(image, expected)


def test_hybrid_kernel_follows_a_new_camera_and_is_refused_where_it_does_not_apply():
    import copy
    import ctypes
    from pathed_amd.integrator import HipScene, PathedError
    from pathed_amd.scene import LoadedScene
    scene = LoadedScene("scenes/cornell-glossy.json", 96, 96)
    hybrid = HipScene(scene.desc, device=0, shade_kernel="hybrid")
    walk = HipScene(scene.desc, device=0, shade_kernel="per-slot", intersector="bvh")
    camera = copy.copy(scene.desc.contents.camera)
    camera.origin = (ctypes.c_float * 3)(1.5, 1.6, 5.0)
    camera.target = (ctypes.c_float * 3)(0.2, 0.6, 0.0)
    hybrid.set_camera(camera)
    walk.set_camera(camera)
    assert np.array_equal(hybrid.render(3, 0, 6, 0, 6), walk.render(3, 0, 6, 0, 6))
    assert hybrid.stats()["path_kernel"] == 7
    # <= 64 triangles: the all-triangles kernels; forced tree walk: no split; a refittable scene keeps ONE tree that refits
    for path, options in (("scenes/cornell.json", {"shade_kernel": "hybrid"}),
                          ("scenes/cornell-glossy.json", {"shade_kernel": "hybrid", "intersector": "bvh"}),
                          ("scenes/mis-pbrt.json", {"shade_kernel": "hybrid", "intersector": "bvh"})):
        other = LoadedScene(path, 32, 32)
        with pytest.raises(PathedError):
            HipScene(other.desc, device=0, **options)
    refittable = HipScene(scene.desc, device=0, refittable=1)
    refittable.render(1, 0, 2, 0, 4)
    assert refittable.stats()["path_kernel"] in (1, 6)
    teapot = LoadedScene("scenes/teapot.json", 32, 32)     # 11 234 triangles: beyond the hybrid kernel's range
    big = HipScene(teapot.desc, device=0)
    big.render(1, 0, 2, 0, 4)
    assert big.stats()["path_kernel"] == 6
