"""On-GPU BVH build (pathed_amd/csrc/lbvh.hip) against the host SAH builder.

The contract: the device-built tree has the same node format, is a valid bounding hierarchy over
exactly the input triangles, is the same tree every run, and — because the intersector's
acceptance rule does not depend on traversal order — gives bit-identical hits and bit-identical
images to the host-built tree (and so to the CPU oracle within the stated image tolerance).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMPTY = np.int32(-2 ** 31)


@pytest.fixture(scope="module")
def libs():
    import oracle_lib
    from pathed_amd.integrator import HipScene
    from pathed_amd.scene import LoadedScene
    return oracle_lib, HipScene, LoadedScene


def _rays(n, seed, centre, extent):
    rng = np.random.default_rng(seed)
    rays = np.zeros((n, 8), dtype=np.float32)
    rays[:, 0:3] = rng.uniform(-extent, extent, (n, 3)) + np.asarray(centre)
    d = rng.normal(size=(n, 3))
    rays[:, 4:7] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays[:, 3] = 1e-3
    rays[:, 7] = 1e5
    return rays


def check_tree(nodes, tris, n_triangles, reported_depth):
    """Structural validity of an exported 4-wide tree (include/pathed_hip.h: export_bvh layout)."""
    assert tris.shape[0] == n_triangles
    prims = np.sort(tris[:, 3].view(np.int32))
    assert np.array_equal(prims, np.arange(n_triangles))            # a permutation of the input
    v0, e1, e2 = tris[:, 0:3], tris[:, 4:7], tris[:, 8:11]
    corners = np.stack([v0, v0 + e1, v0 + e2])
    # e1 / e2 are rounded differences: allow the corners an ulp or two
    slack = 4e-6 * np.maximum(1.0, np.abs(corners).max(axis=0))
    tri_lo, tri_hi = corners.min(axis=0) - slack, corners.max(axis=0) + slack

    lo = nodes[:, 0:12].reshape(-1, 3, 4)
    hi = nodes[:, 12:24].reshape(-1, 3, 4)
    refs = nodes[:, 24:28].view(np.int32)
    n_nodes = nodes.shape[0]
    covered = np.zeros(n_triangles, dtype=np.int32)
    visited = np.zeros(n_nodes, dtype=np.int32)
    # level by level from the root; each node's child box must contain what hangs below it
    level, depth = np.array([0]), 0
    subtree_lo = np.full((n_nodes, 3), np.inf)
    subtree_hi = np.full((n_nodes, 3), -np.inf)
    order = []
    while level.size:
        depth += 1
        visited[level] += 1
        order.append(level)
        next_level = []
        for k in range(4):
            r = refs[level, k]
            inner = r >= 0
            assert (r[inner] < n_nodes).all()
            next_level.append(r[inner])
            leaf = (r <= -2) & (r != EMPTY)
            code = -r[leaf].astype(np.int64) - 1
            first, count = code >> 3, code & 7
            assert (count >= 1).all() and (count <= 4).all() and (first + count <= n_triangles).all()
            for j in range(4):
                has = count > j
                index = first[has] + j
                covered[index] += 1
                box_lo, box_hi = lo[level[leaf][has], :, k], hi[level[leaf][has], :, k]
                assert (box_lo <= tri_lo[index]).all() and (box_hi >= tri_hi[index]).all()
                np.minimum.at(subtree_lo, level[leaf][has], tri_lo[index])
                np.maximum.at(subtree_hi, level[leaf][has], tri_hi[index])
        level = np.concatenate(next_level) if next_level else np.array([], dtype=np.int64)
    assert (visited == 1).all()                 # every node reachable exactly once
    assert (covered == 1).all()                 # every triangle in exactly one leaf
    assert depth == reported_depth
    # bottom-up: an inner child's box contains the boxes stored in that child
    for level in reversed(order):
        own_lo = np.where(refs[level][:, None, :] != EMPTY, lo[level], np.inf).min(axis=2)
        own_hi = np.where(refs[level][:, None, :] != EMPTY, hi[level], -np.inf).max(axis=2)
        subtree_lo[level] = np.minimum(subtree_lo[level], own_lo)
        subtree_hi[level] = np.maximum(subtree_hi[level], own_hi)
        for k in range(4):
            r = refs[level, k]
            inner = r >= 0
            child = r[inner]
            assert (lo[level[inner], :, k] <= subtree_lo[child] + 1e-4 * np.abs(subtree_lo[child])).all()
            assert (hi[level[inner], :, k] >= subtree_hi[child] - 1e-4 * np.abs(subtree_hi[child])).all()


DEVICE_BUILDERS = [("lbvh", 1), ("ploc", 2)]


@pytest.mark.parametrize("builder,builder_code", DEVICE_BUILDERS)
@pytest.mark.parametrize("scene_path,centre,extent", [
    ("scenes/cornell-glossy.json", (0, 1, 0), 1.0),
    ("scenes/teapot.json", (0, 4, 0), 9.0),
])
def test_device_built_tree_is_valid_and_gives_identical_hits_and_images(libs, scene_path, centre, extent, builder, builder_code):
    oracle_lib, HipScene, LoadedScene = libs
    scene = LoadedScene(scene_path, 48, 48)
    assert scene.n_triangles > 64
    sah = HipScene(scene.desc, device=0)
    lbvh = HipScene(scene.desc, device=0, bvh_builder=builder)
    assert sah.stats()["bvh_builder"] == 0 and lbvh.stats()["bvh_builder"] == builder_code
    assert lbvh.stats()["bvh_build_ms"] > 0
    check_tree(*sah.export_bvh(), scene.n_triangles, sah.stats()["bvh_max_depth"])   # the checker itself, on the host tree
    nodes, tris = lbvh.export_bvh()
    check_tree(nodes, tris, scene.n_triangles, lbvh.stats()["bvh_max_depth"])
    # the same tree every run (ids come from a scan, not from atomics)
    nodes_again, tris_again = HipScene(scene.desc, device=0, bvh_builder=builder).export_bvh()
    assert np.array_equal(nodes.view(np.int32), nodes_again.view(np.int32))
    assert np.array_equal(tris.view(np.int32), tris_again.view(np.int32))

    rays = _rays(60000, 11, centre, extent)
    hits = lbvh.trace(rays)
    assert np.array_equal(hits.view(np.int32), sah.trace(rays).view(np.int32))
    assert np.array_equal(hits.view(np.int32), oracle_lib.OracleScene(scene.desc).trace(rays).view(np.int32))
    assert np.array_equal(lbvh.trace(rays, any_hit=True), sah.trace(rays, any_hit=True))
    assert np.array_equal(lbvh.render(5, 0, 8, 0, 10), sah.render(5, 0, 8, 0, 10))


@pytest.mark.parametrize("builder,builder_code", DEVICE_BUILDERS)
def test_device_build_on_a_large_mesh(libs, builder, builder_code):
    """The stand-in dragon at ~330 K triangles: valid tree, hits and image identical to the SAH
    tree's, and the counting kernel says what the cheaper build costs in traversal work."""
    _, HipScene, LoadedScene = libs
    scene = LoadedScene("assets/dragon-standin-7.json", 96, 54)   # generated by tests/conftest.py before the first GPU call
    assert scene.n_triangles >= 200000
    sah = HipScene(scene.desc, device=0)
    lbvh = HipScene(scene.desc, device=0, bvh_builder=builder)
    nodes, tris = lbvh.export_bvh()
    check_tree(nodes, tris, scene.n_triangles, lbvh.stats()["bvh_max_depth"])
    rng = np.random.default_rng(4)
    n = 100000
    rays = np.zeros((n, 8), dtype=np.float32)
    rays[:, 0:3] = rng.normal(size=(n, 3)) * 120 + [0, 0, 25]
    target = rng.normal(size=(n, 3)) * 30 + [0, 0, 25]
    direction = target - rays[:, 0:3]
    rays[:, 4:7] = direction / np.linalg.norm(direction, axis=1, keepdims=True)
    rays[:, 3] = 1e-3
    rays[:, 7] = 1e5
    assert np.array_equal(lbvh.trace(rays).view(np.int32), sah.trace(rays).view(np.int32))
    work = {}
    for name, gpu in (("sah", sah), ("lbvh", lbvh)):
        gpu.set_stats_mode(count=True)
        gpu.reset_stats()
        work[name] = (gpu.render(1, 0, 4, 0, 10), gpu.stats())
    assert np.array_equal(work["sah"][0], work["lbvh"][0])
    boxes = {name: entry[1]["nodes_visited"] for name, entry in work.items()}
    print("build ms: sah %.1f (host) %s %.2f (device); child boxes tested: sah %d %s %d (x%.2f)" % (
        sah.stats()["bvh_build_ms"], builder, lbvh.stats()["bvh_build_ms"], boxes["sah"], builder, boxes["lbvh"], boxes["lbvh"] / boxes["sah"]))
    assert boxes["lbvh"] < 3 * boxes["sah"]
    assert lbvh.stats()["bvh_build_ms"] < sah.stats()["bvh_build_ms"]


def test_degenerate_inputs(libs):
    """Coincident triangles (equal Morton codes: ties fall back to sorted position), the
    smallest mesh the device builder takes, and a flat mesh (zero extent on one axis)."""
    oracle_lib, HipScene, _ = libs
    from scene_builder import BuiltScene
    rng = np.random.default_rng(2)

    def build(vertices, faces):
        built = BuiltScene(24, 24, (0, 0, 6), (0, 0, 0))
        grey = built.material(diffuse=(0.6, 0.6, 0.6))
        light = built.material(emit=(5, 5, 5))
        built.mesh(vertices, faces, grey)
        built.quad([(-4, 5, -4), (4, 5, -4), (4, 5, 4), (-4, 5, 4)], light)
        return built, built.finish()

    one = np.array([(-1, -1, 0), (1, -1, 0), (0, 1, 0)], dtype=np.float32)
    cases = {
        "coincident": (np.tile(one, (70, 1)), [(3 * k, 3 * k + 1, 3 * k + 2) for k in range(70)]),
        "flat": (np.concatenate([rng.uniform(-2, 2, (300, 2)), np.zeros((300, 1))], axis=1).astype(np.float32),
                 [(k, k + 1, k + 2) for k in range(0, 297, 3)]),
        "random": (rng.uniform(-2, 2, (600, 3)).astype(np.float32), [(k, k + 1, k + 2) for k in range(0, 597, 3)]),
    }
    for name, (vertices, faces) in cases.items():
        built, desc = build(vertices, faces)
        n_triangles = len(faces) + 2
        assert n_triangles > 64
        for builder, builder_code in DEVICE_BUILDERS:
            sah = HipScene(desc, device=0)
            lbvh = HipScene(desc, device=0, bvh_builder=builder)
            assert lbvh.stats()["bvh_builder"] == builder_code, name
            nodes, tris = lbvh.export_bvh()
            check_tree(nodes, tris, n_triangles, lbvh.stats()["bvh_max_depth"])
            rays = _rays(20000, 5, (0, 0, 0), 3.0)
            hits = lbvh.trace(rays)
            assert np.array_equal(hits.view(np.int32), sah.trace(rays).view(np.int32)), (name, builder)
            assert np.array_equal(hits.view(np.int32), oracle_lib.OracleScene(desc).trace(rays).view(np.int32)), (name, builder)
            assert np.array_equal(lbvh.render(3, 0, 4, 0, 6), sah.render(3, 0, 4, 0, 6)), (name, builder)


def test_tiny_meshes_keep_the_host_path(libs):
    _, HipScene, LoadedScene = libs
    scene = LoadedScene("scenes/cornell.json", 16, 16)
    gpu = HipScene(scene.desc, device=0, bvh_builder="lbvh")
    assert gpu.stats()["bvh_builder"] == 0 and gpu.stats()["scene_in_lds"] == 2
    assert gpu.render(1, 0, 2, 0, 4).any()


@pytest.mark.parametrize("builder", ["lbvh", "ploc"])
def test_spheres_are_leaves_of_the_device_built_tree(builder):
    """Spheres go to the device builders as primitives of their own (the host builder's rule; the reference hands each one to
    Embree as a geometry, src/sphere.cpp:16-48): a scene with far more than 16 spheres stays on the device, every sphere is
    the single content of exactly one leaf reference, and closest hits, occlusion and image are the host SAH tree's and the
    oracle's bit for bit -- sphere hits included (primitive ids >= the triangle count)."""
    import oracle_lib
    from pathed_amd import _capi
    from pathed_amd.integrator import HipScene
    from scene_builder import BuiltScene
    rng = np.random.default_rng(41)
    built = BuiltScene(48, 32, (0.0, 2.0, 14.0), (0.0, 1.0, 0.0), fov_degrees=40.0)
    grey = built.material(_capi.MAT_LAMBERTIAN, diffuse=(0.6, 0.6, 0.55))
    glass = built.material(_capi.MAT_GLASS, ior=1.5)
    mirror = built.material(_capi.MAT_MIRROR)
    light = built.material(_capi.MAT_LAMBERTIAN, diffuse=(0, 0, 0), emit=(9.0, 8.0, 7.0))
    # a bumpy floor of a few hundred triangles, a light, and 150 spheres of three materials scattered over it (some overlapping)
    grid = 14
    xs, zs = np.meshgrid(np.linspace(-6, 6, grid), np.linspace(-6, 6, grid), indexing="ij")
    ys = 0.3 * np.sin(xs) * np.cos(zs)
    vertices = np.stack([xs, ys, zs], axis=-1).reshape(-1, 3)
    faces = []
    for i in range(grid - 1):
        for j in range(grid - 1):
            a, b, c, d = i * grid + j, (i + 1) * grid + j, (i + 1) * grid + j + 1, i * grid + j + 1
            faces += [(a, c, b), (a, d, c)]
    built.mesh(vertices, faces, grey)
    built.quad([(-3, 7, -3), (3, 7, -3), (3, 7, 3), (-3, 7, 3)], light)
    n_spheres = 150
    for k in range(n_spheres):
        centre = (float(rng.uniform(-5.5, 5.5)), float(rng.uniform(0.4, 3.5)), float(rng.uniform(-5.5, 5.5)))
        built.sphere(centre, float(rng.uniform(0.08, 0.6)), (grey, glass, mirror)[k % 3])
    desc = built.finish()
    n_triangles = desc.contents.n_triangles
    assert n_triangles > 64 and desc.contents.n_spheres == n_spheres
    host = HipScene(desc, device=0, bvh_builder="sah")
    device = HipScene(desc, device=0, bvh_builder=builder)
    assert device.stats()["bvh_builder"] == {"lbvh": 1, "ploc": 2}[builder]      # not redirected to the host any more
    nodes, tris = device.export_bvh()
    refs = nodes[:, 24:28].view(np.int32).reshape(-1)
    leaves = refs[(refs <= -2) & (refs != EMPTY)]
    codes = -leaves.astype(np.int64) - 1
    sphere_leaves = np.sort((codes[(codes & 7) == 0] >> 3) - 1)
    assert np.array_equal(sphere_leaves, np.arange(n_spheres))                   # each sphere once, in a leaf of its own
    triangle_leaves = codes[(codes & 7) != 0]
    assert int((triangle_leaves & 7).sum()) == n_triangles and tris.shape[0] == n_triangles
    rays = _rays(60000, 3, (0.0, 2.0, 0.0), 7.0)
    cpu = oracle_lib.OracleScene(desc)
    expected = cpu.trace(rays)
    assert (expected[:, 3].view(np.int32) >= n_triangles).mean() > 0.05           # a fair share of sphere hits
    assert np.array_equal(device.trace(rays).view(np.int32), expected.view(np.int32))
    assert np.array_equal(host.trace(rays).view(np.int32), expected.view(np.int32))
    assert np.array_equal(device.trace(rays, any_hit=True), cpu.trace(rays, any_hit=True))
    assert np.array_equal(device.render(5, 0, 8, 0, 8), host.render(5, 0, 8, 0, 8))
