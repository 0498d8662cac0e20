"""k_path_wave (pathed_amd/csrc/path_wave.h): the path tracer over a BVH with the paths in registers and the wave's rays shared
through LDS -- what BVH scenes run for calls of fewer than 48 Mi camera samples (include/pathed_hip.h: shade_kernel).  Its
arithmetic is the wavefront kernels' (k_trace's traversal, k_shade's vertex code, the same unit decomposition), so the tests
ask for the same image bit for bit; oracle parity of the default path is the parity suite's (tests/test_gpu_parity.py)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scene_path,width,height,spp,options", [
    ("scenes/teapot.json", 160, 120, 8, {}),                               # environment light only: the narrowed instantiation
    ("scenes/teapot.json", 17, 5, 3, {}),                                  # ragged: 85 pixels, fewer paths than one block has lanes
    ("assets/dragon-standin-9.json", 128, 72, 8, {}),                      # 5.2 M triangles (tests/conftest.py generates them): a deep tree
    ("scenes/cornell.json", 96, 96, 12, {"intersector": "bvh"}),           # triangle lights, the generic instantiation
    ("scenes/mis-pbrt.json", 96, 96, 12, {"intersector": "bvh"}),          # sphere lights as leaves of the tree
    ("scenes/cornell-glass.json", 64, 64, 12, {"intersector": "bvh"}),     # delta BSDF
])
def test_wave_kernel_renders_the_wavefront_image_bit_for_bit(scene_path, width, height, spp, options):
    from pathed_amd.integrator import HipScene
    from pathed_amd.scene import LoadedScene
    scene = LoadedScene(scene_path, width, height)
    wavefront = HipScene(scene.desc, device=0, shade_kernel="per-slot", **options)
    wave = HipScene(scene.desc, device=0, shade_kernel="wave", **options)
    expected = wavefront.render(5, 0, spp, 0, 8)
    image = wave.render(5, 0, spp, 0, 8)
    assert expected.any() and np.array_equal(image, expected)
    assert wavefront.stats()["path_kernel"] == 1 and wave.stats()["path_kernel"] == 6
    # a second batch continues the sum; a window of bounces
    assert np.array_equal(wave.render(5, spp, 4, 0, 8), wavefront.render(5, spp, 4, 0, 8))
    assert np.array_equal(wave.render(9, 0, 4, 2, 3), wavefront.render(9, 0, 4, 2, 3))
    assert wave.stats()["dropped_samples"] == 0


def test_call_size_picks_the_kernel_and_the_straggler_bound_changes_nothing(monkeypatch):
    from pathed_amd.integrator import HipScene
    from pathed_amd.scene import LoadedScene
    scene = LoadedScene("scenes/teapot.json", 128, 96)
    automatic = HipScene(scene.desc, device=0)
    expected = automatic.render(3, 0, 6, 0, 10)
    assert automatic.stats()["path_kernel"] == 6          # 74 k camera samples: far below the 48 Mi where the wavefront takes over
    long_call = HipScene(scene.desc, device=0, wave_max_ksamples=1)   # ... unless the scene says 1 024
    assert np.array_equal(long_call.render(3, 0, 6, 0, 10), expected)
    assert long_call.stats()["path_kernel"] == 1
    # the product library reads no tuning variable from the environment (include/pathed_hip.h: PathedSceneOptions)
    from conftest import has_experiments
    if not has_experiments():
        monkeypatch.setenv("PATHED_WAVE_MAX_SAMPLES", "1000")
        monkeypatch.setenv("PATHED_SHADE_KERNEL", "per-slot")
        deaf = HipScene(scene.desc, device=0)
        assert np.array_equal(deaf.render(3, 0, 6, 0, 10), expected) and deaf.stats()["path_kernel"] == 6
        monkeypatch.delenv("PATHED_WAVE_MAX_SAMPLES")
        monkeypatch.delenv("PATHED_SHADE_KERNEL")
    # counting is the wavefront kernels': stats mode renders there, with the same image
    counting = HipScene(scene.desc, device=0)
    counting.set_stats_mode(count=True)
    assert np.array_equal(counting.render(3, 0, 6, 0, 10), expected)
    stats = counting.stats()
    assert stats["path_kernel"] == 1 and stats["closest_rays"] > 0
    for stragglers, refill in ((-1, 0), (1, 16), (64, 64)):   # never leave rays in flight / ... / leave whatever is in flight once the list is dealt
        gpu = HipScene(scene.desc, device=0, shade_kernel="wave", wave_stragglers=stragglers, wave_refill=refill)
        assert np.array_equal(gpu.render(3, 0, 6, 0, 10), expected), stragglers


def test_wave_kernel_is_refused_where_it_does_not_apply():
    from pathed_amd.integrator import HipScene, PathedError
    from pathed_amd.scene import LoadedScene
    scene = LoadedScene("scenes/cornell.json", 32, 32)
    with pytest.raises(PathedError):
        HipScene(scene.desc, device=0, shade_kernel="wave")   # 36 triangles: the all-triangles kernels serve it


def test_wave_kernel_follows_a_refit():
    """The refitted tree is the one both organisations walk (pathed_hip_scene_refit)."""
    from pathed_amd.integrator import HipScene
    from pathed_amd.scene import LoadedScene
    scene = LoadedScene("scenes/teapot.json", 96, 72)
    wave = HipScene(scene.desc, device=0, shade_kernel="wave", bvh_builder="lbvh", refittable=1)
    wavefront = HipScene(scene.desc, device=0, shade_kernel="per-slot", bvh_builder="lbvh", refittable=1)
    n = scene.desc.contents.n_vertices
    positions = np.ctypeslib.as_array(scene.desc.contents.positions, shape=(n, 3)).copy()
    positions[:, 1] += (0.02 * np.abs(positions).max() * np.sin(3.0 * positions[:, 0])).astype(np.float32)
    wave.refit(positions)
    wavefront.refit(positions)
    image = wave.render(2, 0, 6, 0, 8)
    assert image.any() and np.array_equal(image, wavefront.render(2, 0, 6, 0, 8))
