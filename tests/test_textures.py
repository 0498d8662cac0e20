"""Image-texture albedo (reference src/texture.cpp, scene_parser.cpp:624-648).

Pinned against the reference's own object code: oracle/_ref/refdump decodes the fixture files
under tests/golden/textures with the reference's vendored stb_image and calls Texture::lookup
(records `texture_image` / `texture_lookup` of tests/golden/reference_functions.jsonl).
  - the host library's PNG / PNM decoder must return stb_image's bytes exactly,
  - the oracle's lookup (wrap, v flip, nearest texel, powf(x / 255, 2.2)) must match the reference's,
  - the scene loader must accept the reference's "texture" key for lambertian and plastic.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_lib
from pathed_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "reference_functions.jsonl")
TEXTURES = os.path.join(ROOT, "tests", "golden", "textures")
FILES = ["rgb8_7x5.png", "rgba16_4x3.png", "greyalpha8_3x4.png", "grey2_5x6.png", "palette4_5x4.png",
         "rgb_2x3.ppm", "wood_48x32.png"]
# JPEG files written by Pillow / libjpeg (tests/golden/make_texture_fixtures.py); record index = 100 + position
JPEG_FILES = ["jpeg_444_37x29.jpg", "jpeg_420_37x29.jpg", "jpeg_422_37x29.jpg", "jpeg_progressive_420_37x29.jpg",
              "jpeg_progressive_444_37x29.jpg", "jpeg_noise_420_26x21.jpg", "jpeg_grey_37x29.jpg", "jpeg_420_1x9.jpg",
              "jpeg_restart_420_100x70.jpg",
              # Adam7-interlaced PNGs written by tests/golden/make_texture_fixtures.py (records 109...)
              "adam7_rgb8_13x11.png", "adam7_grey4_10x9.png", "adam7_rgba16_3x2.png"]


def _records():
    images, lookups = {}, []
    with open(GOLDEN) as handle:
        for line in handle:
            record = json.loads(line)
            if record["fn"] == "texture_image":
                out = record["out"]
                width, height = int(out[0]), int(out[1])
                images[int(record["in"][0])] = np.array(out[2:], dtype=np.uint8).reshape(height, width, 3)
            elif record["fn"] == "texture_lookup":
                lookups.append((int(record["in"][0]), np.float32(record["in"][1]), np.float32(record["in"][2]),
                                np.array(record["out"], dtype=np.float64)))
    return images, lookups


IMAGES, LOOKUPS = _records()


def load_image(path):
    host = _capi.load_host()
    width, height = C.c_int(0), C.c_int(0)
    code = host.pathed_host_load_image_rgb8(path.encode(), C.byref(width), C.byref(height), None, 0)
    if code != 0:
        raise RuntimeError(host.pathed_host_last_error().decode())
    data = np.zeros((height.value, width.value, 3), dtype=np.uint8)
    code = host.pathed_host_load_image_rgb8(path.encode(), C.byref(width), C.byref(height),
                                            data.ctypes.data_as(C.POINTER(C.c_uint8)), data.size)
    assert code == 0
    return data


def test_golden_file_has_the_texture_records():
    assert sorted(IMAGES) == list(range(len(FILES))) + [100 + k for k in range(len(JPEG_FILES))]
    assert len(LOOKUPS) >= 200


@pytest.mark.parametrize("index", range(len(FILES)))
def test_decoder_returns_the_bytes_stb_image_returns(index):
    decoded = load_image(os.path.join(TEXTURES, FILES[index]))
    assert decoded.shape == IMAGES[index].shape
    assert np.array_equal(decoded, IMAGES[index])


@pytest.mark.parametrize("index", range(len(JPEG_FILES)))
def test_jpeg_decoder_returns_the_bytes_stb_image_returns(index):
    """A JPEG's decoded bytes depend on the decoder (IDCT, chroma upsampling, colour conversion): the host
    decoder follows stb_image's arithmetic and must reproduce its output exactly — baseline 4:4:4 / 4:2:2 /
    4:2:0, progressive, greyscale, optimised Huffman tables, restart intervals, one-pixel-wide images."""
    decoded = load_image(os.path.join(TEXTURES, JPEG_FILES[index]))
    assert decoded.shape == IMAGES[100 + index].shape
    assert np.array_equal(decoded, IMAGES[100 + index])


def test_lookup_matches_the_reference():
    worst = 0.0
    for index, u, v, expected in LOOKUPS:
        image = IMAGES[index]
        inputs = np.concatenate([[image.shape[1], image.shape[0], u, v], image.reshape(-1).astype(np.float32)]).astype(np.float32)
        actual = oracle_lib.evaluate("texture_lookup", inputs, n_out=3)
        # same glibc powf on the same bytes: exact up to the 9 digits the dump prints
        assert np.allclose(actual, expected, rtol=2e-7, atol=1e-9), (index, u, v, actual, expected)
        worst = max(worst, float(np.abs(actual - expected).max()))
    assert worst < 1e-7


def test_decoder_errors_are_loud(tmp_path):
    jpeg = tmp_path / "a.jpg"
    jpeg.write_bytes(open(os.path.join(TEXTURES, JPEG_FILES[1]), "rb").read()[:300])
    with pytest.raises(RuntimeError, match="jpeg"):
        load_image(str(jpeg))
    unknown = tmp_path / "a.gif"
    unknown.write_bytes(b"GIF89a" + b"\0" * 32)
    with pytest.raises(RuntimeError, match="unknown image format"):
        load_image(str(unknown))
    broken = tmp_path / "b.png"
    broken.write_bytes(open(os.path.join(TEXTURES, FILES[0]), "rb").read()[:60])
    with pytest.raises(RuntimeError, match="png"):
        load_image(str(broken))
    with pytest.raises(RuntimeError, match="cannot open"):
        load_image(str(tmp_path / "missing.png"))


SCENE = {
    "sensor": {"lookAt": {"origin": ["0", "1", "4"], "target": ["0", "1", "0"], "up": ["0", "1", "0"]}, "fov": "40"},
    "models": [
        {"type": "quad", "transform": {"legacy": True, "scale": ["2", "2", "2"], "rotate": ["0", "0", "0"], "translate": ["0", "0", "0"]},
         "bsdf": {"type": "lambertian", "diffuseReflectance": ["0.9", "0.1", "0.1"], "texture": "TEXTURE"}},
        {"type": "quad", "transform": {"legacy": True, "scale": ["1", "1", "1"], "rotate": ["180", "0", "0"], "translate": ["0", "3", "0"]},
         "bsdf": {"type": "lambertian", "diffuseReflectance": ["0", "0", "0"], "emit": ["8", "8", "8"]}},
    ],
}


def test_loader_reads_the_texture_key(tmp_path):
    from pathed_amd.scene import LoadedScene
    scene = json.loads(json.dumps(SCENE).replace("TEXTURE", os.path.join(TEXTURES, "wood_48x32.png")))
    scene["models"].append({
        "type": "quad", "transform": {"legacy": True, "scale": ["0.5", "0.5", "0.5"], "rotate": ["0", "0", "0"], "translate": ["-1", "1", "0"]},
        "bsdf": {"type": "plastic", "diffuseReflectance": ["0.2", "0.2", "0.2"], "texture": os.path.join(TEXTURES, "wood_48x32.png"),
                 "distribution": {"type": "beckmann", "alpha": "0.1"}}})
    path = tmp_path / "textured.json"
    path.write_text(json.dumps(scene))
    loaded = LoadedScene(str(path), 16, 16)
    desc = loaded.desc.contents
    assert desc.n_textures == 1                      # the same file is loaded once
    texture = desc.textures[0]
    assert (texture.width, texture.height) == (48, 32)
    data = np.ctypeslib.as_array(texture.rgb, shape=(32, 48, 3))
    assert np.array_equal(data, IMAGES[6])
    kinds = [(desc.materials[i].type, desc.materials[i].albedo_type, desc.materials[i].texture) for i in range(desc.n_materials)]
    assert (_capi.MAT_LAMBERTIAN, _capi.ALBEDO_TEXTURE, 0) in kinds
    assert (_capi.MAT_PLASTIC, _capi.ALBEDO_TEXTURE, 0) in kinds
    # a textured Lambertian's constant colour is unused (reference lambertian.cpp:12-14 zeroes it)
    textured = [desc.materials[i] for i in range(desc.n_materials) if desc.materials[i].albedo_type == _capi.ALBEDO_TEXTURE]
    assert all(list(m.diffuse) == [0.0, 0.0, 0.0] for m in textured)

    missing = json.loads(json.dumps(SCENE).replace("TEXTURE", "no/such/file.png"))
    bad = tmp_path / "missing.json"
    bad.write_text(json.dumps(missing))
    with pytest.raises(RuntimeError, match="Error loading texture"):
        LoadedScene(str(bad), 16, 16)


def test_oracle_renders_the_texture(tmp_path):
    """The textured floor's colour shows up in the oracle's image (sanity of the plumbing)."""
    from pathed_amd.scene import LoadedScene
    scene = json.loads(json.dumps(SCENE).replace("TEXTURE", os.path.join(TEXTURES, "wood_48x32.png")))
    path = tmp_path / "textured.json"
    path.write_text(json.dumps(scene))
    loaded = LoadedScene(str(path), 24, 24)
    image, _ = oracle_lib.OracleScene(loaded.desc).render(24, 24, 1, 0, 16, 0, 4, threads=2)
    floor = image[:10].reshape(-1, 3).mean(axis=0)
    assert floor[0] > floor[2] > 0.0                 # the fixture is red-brown: r > g,b and lit


@pytest.mark.gpu
def test_gpu_matches_the_oracle_on_a_textured_scene(tmp_path):
    from pathed_amd.integrator import HipScene
    from pathed_amd.scene import LoadedScene
    scene = json.loads(json.dumps(SCENE).replace("TEXTURE", os.path.join(TEXTURES, "wood_48x32.png")))
    scene["models"].append({
        "type": "quad", "transform": {"legacy": True, "scale": ["0.7", "0.7", "0.7"], "rotate": ["0", "0", "0"], "translate": ["-0.8", "0.5", "0"]},
        "bsdf": {"type": "plastic", "diffuseReflectance": ["0.2", "0.2", "0.2"], "texture": os.path.join(TEXTURES, "rgb8_7x5.png"),
                 "distribution": {"type": "beckmann", "alpha": "0.2"}}})
    path = tmp_path / "textured.json"
    path.write_text(json.dumps(scene))
    loaded = LoadedScene(str(path), 64, 64)
    gpu = HipScene(loaded.desc, device=0)
    image = gpu.render(3, 0, 32, 0, 6)
    expected, _ = oracle_lib.OracleScene(loaded.desc).render(64, 64, 3, 0, 32, 0, 6, threads=os.cpu_count())
    rel = float(np.linalg.norm(image - expected) / np.linalg.norm(expected))
    bad = float((np.abs(image - expected) > 1e-2 * np.maximum(np.abs(expected), 1e-3)).any(axis=2).mean())
    assert rel <= 2e-3 and bad <= 1e-3, (rel, bad)   # SURVEY.md §8d tolerance
    # the texture really is on the path: a constant-colour floor gives a different image
    plain = json.loads(json.dumps(SCENE).replace(', "texture": "TEXTURE"', ""))
    plain_path = tmp_path / "plain.json"
    plain_path.write_text(json.dumps(plain))
    plain_loaded = LoadedScene(str(plain_path), 64, 64)   # owns the description: keep it alive
    other = HipScene(plain_loaded.desc, device=0).render(3, 0, 32, 0, 6)
    assert np.linalg.norm(other - image) / np.linalg.norm(image) > 0.05


@pytest.mark.gpu
def test_gpu_rejects_bad_texture_descriptions():
    from pathed_amd.integrator import HipScene, PathedError
    from scene_builder import BuiltScene
    built = BuiltScene(8, 8, (0, 0, 4), (0, 0, 0))
    texture = built.texture(np.zeros((2, 2, 3), dtype=np.uint8))
    material = built.material(texture=texture)
    built.materials[material].texture = 5            # out of range
    built.quad([(-1, -1, 0), (1, -1, 0), (1, 1, 0), (-1, 1, 0)], material)
    with pytest.raises(PathedError, match="texture"):
        HipScene(built.finish(), device=0)
