"""The wavefront's LOCAL RAYS (pathed_amd/csrc/kernels.h: k_shade, RenderParams::localTris).  In "an object on a floor under a
sky" most rays never come near the object: the shade kernel tests a ray against the bounds of everything but the scene's few
large triangles, and a ray that cannot meet them is resolved right there against those triangles -- the tree walk's
intersector and acceptance rule -- instead of going through the trace kernel.  Scheduling only: the image is the same bits
with the shortcut on or off, and every query is accounted for (Scene::testIntersect / testOcclusion, reference
src/scene.cpp:91-223, 355-381)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scene_path,width,height,spp,builder", [
    ("scenes/teapot.json", 256, 192, 8, "sah"),                     # C4's scene: the checkerboard quad is the large pair
    ("assets/dragon-standin-9.json", 320, 180, 4, "ploc"),          # C5's: 5.2 M triangles on a floor quad
    ("scenes/teapot.json", 33, 19, 3, "lbvh"),
])
def test_local_rays_change_no_bit_and_lose_no_query(scene_path, width, height, spp, builder):
    from pathed_amd.integrator import HipScene
    from pathed_amd.scene import LoadedScene
    scene = LoadedScene(scene_path, width, height)
    on = HipScene(scene.desc, device=0, bvh_builder=builder, shade_kernel="per-slot")
    off = HipScene(scene.desc, device=0, bvh_builder=builder, shade_kernel="per-slot", local_rays=1)
    expected = off.render(3, 0, spp, 0, 10)
    assert expected.any() and np.array_equal(on.render(3, 0, spp, 0, 10), expected)
    assert np.array_equal(on.render(3, spp, 2, 1, 4), off.render(3, spp, 2, 1, 4))          # a batch further on, a bounce window
    # the counting instantiations: the same image again, and the books balance
    for gpu in (on, off):
        gpu.set_stats_mode(count=True)
        gpu.reset_stats()
        assert np.array_equal(gpu.render(3, 0, spp, 0, 10), expected)
    a, b = on.stats(), off.stats()
    assert b["local_closest_rays"] == 0 and b["local_shadow_rays"] == 0
    assert a["local_closest_rays"] > 0.2 * b["closest_rays"]                                  # a good part of the queries never reach the tree
    assert a["closest_rays"] + a["local_closest_rays"] == b["closest_rays"]
    assert a["shadow_rays"] + a["local_shadow_rays"] == b["shadow_rays"]
    assert a["nodes_visited"] < b["nodes_visited"] and a["dropped_samples"] == b["dropped_samples"] == 0
    # several shade launches per trace launch (the slots whose rays were all local advance, the others wait): scheduling only
    for launches in (1, 3, 5):
        other = HipScene(scene.desc, device=0, bvh_builder=builder, shade_kernel="per-slot", shade_launches=launches)
        assert np.array_equal(other.render(3, 0, spp, 0, 10), expected), launches
    # [r5] k_shade_env (environment-lit scenes: a sample that ends starts the next one in the same launch, and a local camera
    # ray that hits a large triangle has its first vertex shaded there) against k_shade<ENV_ONLY> (shade_chain=1): scheduling
    # only -- the same bits, the same queries, with few slots (many launches, slots re-used often) and with many
    for slots in (0, 4096):
        chained = HipScene(scene.desc, device=0, bvh_builder=builder, shade_kernel="per-slot", max_slots=slots)
        plain = HipScene(scene.desc, device=0, bvh_builder=builder, shade_kernel="per-slot", max_slots=slots, shade_chain=1)
        assert np.array_equal(chained.render(3, 0, spp, 0, 10), expected), slots
        assert np.array_equal(plain.render(3, 0, spp, 0, 10), expected), slots
        for gpu in (chained, plain):
            gpu.set_stats_mode(count=True)
            gpu.reset_stats()
            assert np.array_equal(gpu.render(3, 0, spp, 0, 10), expected)
        c, q = chained.stats(), plain.stats()
        assert c["closest_rays"] + c["local_closest_rays"] == q["closest_rays"] + q["local_closest_rays"] == b["closest_rays"]
        assert c["shadow_rays"] + c["local_shadow_rays"] == q["shadow_rays"] + q["local_shadow_rays"] == b["shadow_rays"]
        assert c["iterations"] <= q["iterations"]          # fewer slot visits per sample: never more launches
    # the wave kernel and the default dispatch agree with both
    assert np.array_equal(HipScene(scene.desc, device=0, bvh_builder=builder).render(3, 0, spp, 0, 10), expected)


def test_local_rays_stay_off_where_they_do_not_apply():
    from pathed_amd.integrator import HipScene
    from pathed_amd.scene import LoadedScene
    for path in ("scenes/cornell-glossy.json",     # 24 large triangles: more than the shade kernel would test itself
                 "scenes/mis-pbrt.json"):           # sphere primitives (with intersector="bvh": the tree walk)
        scene = LoadedScene(path, 64, 64)
        gpu = HipScene(scene.desc, device=0, shade_kernel="per-slot", intersector="bvh")
        gpu.set_stats_mode(count=True)
        gpu.render(1, 0, 4, 0, 6)
        stats = gpu.stats()
        assert stats["closest_rays"] > 0 and stats["local_closest_rays"] == 0 and stats["local_shadow_rays"] == 0
