"""C++ host side (libpathed_host.so): scene / OBJ / MTL / PLY / quad readers, EXR, job files.

Format rules are the reference's (src/scene_parser.cpp, src/obj_parser.cpp, src/mtl_parser.cpp,
src/ply_parser.cpp, src/quad.cpp, src/image.cpp); see pathed_amd/host/scene_loader.h.
"""
import ctypes as C
import json
import os
import struct

import numpy as np
import pytest

from pathed_amd import _capi
from pathed_amd.scene import LoadedScene


def _arrays(scene):
    d = scene.desc.contents
    positions = np.ctypeslib.as_array(d.positions, shape=(d.n_vertices, 3)).copy()
    normals = np.ctypeslib.as_array(d.normals, shape=(d.n_vertices, 3)).copy()
    uvs = np.ctypeslib.as_array(d.uvs, shape=(d.n_vertices, 2)).copy()
    indices = np.ctypeslib.as_array(d.indices, shape=(d.n_triangles, 3)).copy()
    tri_material = np.ctypeslib.as_array(d.tri_material, shape=(d.n_triangles,)).copy()
    return d, positions, normals, uvs, indices, tri_material


def test_cornell_scene_matches_reference_parse():
    scene = LoadedScene("scenes/cornell.json", 256, 256)
    d, positions, normals, uvs, indices, tri_material = _arrays(scene)
    # 18 quads "f -4 -3 -2 -1" -> 36 triangles split (0,1,2),(0,2,3) (obj_parser.cpp:377-396)
    assert d.n_triangles == 36 and d.n_vertices == 72
    assert indices[0].tolist() == [0, 1, 2] and indices[1].tolist() == [0, 2, 3]
    # no vn / vt in the file: attributes zero-filled (geometry_parser.cpp:68-89)
    assert not normals.any() and not uvs.any()
    # floor quad winding: (v1-v0)x(v2-v0) points into the box (+y), SURVEY.md App. C
    e1, e2 = positions[1] - positions[0], positions[2] - positions[0]
    assert np.cross(e1, e2)[1] > 0
    # camera: fov given in degrees as a string
    assert abs(d.camera.vertical_fov - np.float32(19.5 / 180 * np.pi)) < 1e-7
    assert tuple(d.camera.origin) == (0.0, 1.0, np.float32(6.8))
    # exactly the two light triangles are emissive with Ke = 17 12 4
    emissive = [i for i in range(36) if tuple(d.materials[tri_material[i]].emit) != (0.0, 0.0, 0.0)]
    assert len(emissive) == 2
    assert tuple(d.materials[tri_material[emissive[0]]].emit) == (17.0, 12.0, 4.0)
    light = positions[indices[emissive[0]]]
    assert np.cross(light[1] - light[0], light[2] - light[0])[1] < 0  # light faces down
    assert d.n_geoms == 1 and d.geoms[0].count == 36
    assert d.env is None or not bool(d.env)


def test_triplet_faces_normals_and_transform():
    # cornell-glass: ball.obj uses v/vt/vn triplets with negative indices + translate (0,0,1)
    scene = LoadedScene("scenes/cornell-glass.json", 64, 64)
    d, positions, normals, uvs, indices, tri_material = _arrays(scene)
    assert d.n_geoms == 3
    ball = d.geoms[2]
    assert ball.count == 1088
    ball_material = d.materials[tri_material[ball.first]]
    assert ball_material.type == _capi.MAT_GLASS and abs(ball_material.ior - 1.4) < 1e-7
    box_material = d.materials[tri_material[d.geoms[1].first]]
    assert box_material.type == _capi.MAT_MIRROR
    used = np.unique(indices[ball.first:ball.first + ball.count])
    ball_normals = normals[used]
    lengths = np.linalg.norm(ball_normals, axis=1)
    assert np.all(np.abs(lengths - 1) < 1e-3)
    # smooth normals of a sphere point away from its centre; the centre moved by the translate
    centre = positions[used].mean(axis=0)
    reference = LoadedScene("scenes/cornell-glossy.json", 64, 64)
    d2, positions2, _, _, indices2, _ = _arrays(reference)
    used2 = np.unique(indices2[d2.geoms[2].first:d2.geoms[2].first + d2.geoms[2].count])
    centre2 = positions2[used2].mean(axis=0)
    assert np.allclose(centre - centre2, [0, 0, 1], atol=1e-5)
    outward = positions[used] - centre
    outward /= np.linalg.norm(outward, axis=1, keepdims=True)
    assert np.mean(np.sum(outward * ball_normals, axis=1)) > 0.99


def test_obj_face_syntaxes_and_cube_normal_split(tmp_path):
    obj = tmp_path / "mixed.obj"
    obj.write_text(
        "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0 0 1\n"
        "vn 0 0 1\nvn 0 1 0\n"
        "vt 0.25 0.75\n"
        "g a\n"
        "f 1 2 3 4\n"            # quad, geometry only
        "f 1//1 2//1 3//1\n"     # v//vn
        "f 1//2 2//2 5//2\n"     # same vertices, other normal -> duplicated vertices
        "f -5/1/1 -4/1/1 -3/1/1\n"  # triplets with negative indices
        "usemtl hidden\nf 1 2 3\n"  # skipped (obj_parser.cpp:151)
    )
    scene_file = tmp_path / "scene.json"
    scene_file.write_text(json.dumps({
        "sensor": {"lookAt": {"origin": ["0", "0", "5"], "target": ["0", "0", "0"], "up": ["0", "1", "0"]}, "fov": "40"},
        "models": [{"type": "obj", "filename": str(obj), "bsdf": {"type": "mirror"}}],
    }))
    scene = LoadedScene(str(scene_file), 8, 8, asset_root="")
    d, positions, normals, uvs, indices, tri_material = _arrays(scene)
    assert d.n_triangles == 5
    assert indices[0].tolist() == [0, 1, 2] and indices[1].tolist() == [0, 2, 3]
    # face 3 re-uses vertices 0,1 with normal index 1 after they were seen with -1/0:
    # the reference duplicates them at the back of the vertex list (obj_parser.cpp:60-117)
    assert d.n_vertices > 5
    assert all(d.materials[m].type == _capi.MAT_MIRROR for m in tri_material)
    assert uvs[0].tolist() == [0.25, 0.75]


def test_material_lookup_precedence_and_reference(tmp_path):
    scene = LoadedScene("scenes/cornell-oren-nayar.json", 16, 16)
    d, _, _, _, _, tri_material = _arrays(scene)
    types = [d.materials[m].type for m in tri_material]
    # JSON materials keyed by OBJ group beat the MTL entries (obj_parser.cpp:241-252) ...
    assert types[0] == _capi.MAT_OREN_NAYAR
    assert _capi.MAT_MICROFACET in types and _capi.MAT_PLASTIC in types
    # ... and the group key beats the usemtl key: in CornellBox-Original.obj the tall box's faces
    # are still in group "shortBox" (its `g tallBox` line comes after them), so BOTH boxes
    # resolve to the JSON material named "shortBox": 12 + 12 = 24 plastic triangles
    assert types.count(_capi.MAT_PLASTIC) == 24
    # ... while the light group has no JSON entry and keeps its MTL emission
    assert sum(1 for m in tri_material if tuple(d.materials[m].emit) == (17.0, 12.0, 4.0)) == 2


def test_quad_sphere_ply_and_env(tmp_path):
    scene = LoadedScene("scenes/mis-pbrt.json", 32, 32)
    d, positions, normals, uvs, indices, tri_material = _arrays(scene)
    assert d.n_spheres == 5 and d.n_geoms == 10
    assert [g.type for g in d.geoms[0:5]] == [_capi.GEOM_SPHERE] * 5
    assert abs(d.spheres[2].radius - 0.03333) < 1e-7
    assert d.camera.flip_handedness == 1
    plate = d.materials[tri_material[d.geoms[5].first]]
    assert plate.type == _capi.MAT_PLASTIC and abs(plate.alpha - 0.005) < 1e-9

    teapot = LoadedScene("scenes/teapot.json", 32, 32)
    d, positions, normals, uvs, indices, tri_material = _arrays(teapot)
    quad = d.geoms[2]
    assert quad.count == 2
    checker = d.materials[tri_material[quad.first]]
    assert checker.albedo_type == _capi.ALBEDO_CHECKERBOARD and tuple(checker.checker_res) == (20.0, 20.0)
    assert tuple(checker.diffuse) == (0.0, 0.0, 0.0)  # Lambertian(albedo, emit) ctor zeroes m_diffuse
    quad_vertices = positions[np.unique(indices[quad.first:quad.first + 2])]
    # legacy transform: scale 113.071, rotate x by -90 deg -> the z-up quad becomes the y = 0 plane
    assert np.allclose(quad_vertices[:, 1], 0, atol=1e-3)
    assert abs(np.abs(quad_vertices).max() - 113.071 * np.sqrt(2)) < 1e-2
    quad_normal = normals[indices[quad.first][0]]
    assert np.allclose(np.abs(quad_normal), [0, 1, 0], atol=1e-5)
    assert bool(d.env) and d.env.contents.width == 256 and d.env.contents.height == 128


def test_scene_errors_are_loud(tmp_path):
    bad = tmp_path / "bad.json"
    bad.write_text(json.dumps({
        "sensor": {"lookAt": {"origin": ["0", "0", "5"], "target": ["0", "0", "0"], "up": ["0", "1", "0"]}, "fov": "40"},
        "models": [{"type": "quad", "bsdf": {"type": "velvet"}}],
    }))
    with pytest.raises(RuntimeError, match="Unimplemented material: velvet"):
        LoadedScene(str(bad), 8, 8, asset_root="")
    with pytest.raises(RuntimeError):
        LoadedScene("scenes/does-not-exist.json", 8, 8)


def test_exr_round_trip(tmp_path):
    host = _capi.load_host()
    rng = np.random.default_rng(3)
    rgba = rng.uniform(0, 4, size=(5, 7, 4)).astype(np.float32)
    path = str(tmp_path / "x.exr").encode()
    assert host.pathed_host_write_exr_float_rgba(path, 7, 5, rgba.ctypes.data_as(C.POINTER(C.c_float))) == 0
    w, h = C.c_int(), C.c_int()
    out = np.zeros_like(rgba)
    assert host.pathed_host_read_exr_rgba(path, C.byref(w), C.byref(h), out.ctypes.data_as(C.POINTER(C.c_float)), out.size) == 0
    assert (w.value, h.value) == (7, 5)
    assert np.array_equal(out, rgba)
    # header: magic, version 2, scanline, uncompressed
    with open(path, "rb") as handle:
        blob = handle.read()
    assert blob[:4] == bytes([0x76, 0x2F, 0x31, 0x01]) and blob[4:8] == bytes([2, 0, 0, 0])
    assert b"compression\x00compression\x00\x01\x00\x00\x00\x00" in blob


def test_one_pixel_env_map_matches_reference_decode():
    # tests/golden "env_image": the reference's tinyexr decode of test_scenes/1_pixel_test.exr
    host = _capi.load_host()
    w, h = C.c_int(), C.c_int()
    path = os.path.join(_capi.REPO_ROOT, "test_scenes", "1_pixel_test.exr").encode()
    assert host.pathed_host_read_exr_rgba(path, C.byref(w), C.byref(h), None, 0) == 0
    assert (w.value, h.value) == (1000, 500)
    data = np.zeros((500, 1000, 4), dtype=np.float32)
    assert host.pathed_host_read_exr_rgba(path, C.byref(w), C.byref(h), data.ctypes.data_as(C.POINTER(C.c_float)), data.size) == 0
    nonzero = np.argwhere(data[..., :3].sum(-1) != 0)
    assert nonzero.tolist() == [[239, 753]]
    assert data[239, 753, :3].tolist() == [10000.0, 10000.0, 10000.0]


@pytest.mark.parametrize("index,name", [(0, "piz_float_45x70.exr"), (1, "piz_half_33x40.exr")])
def test_piz_compressed_exr_matches_reference_decode(index, name):
    """PIZ (range bitmap + wavelet + Huffman) environment maps: the fixture files were written by the
    reference's vendored tinyexr through oracle/_ref/refdump, and `exr_piz_image` holds what its LoadEXR
    returned for them (FLOAT channels spanning > 2 blocks with the 16-bit wavelet; HALF channels with few
    distinct values, i.e. the 14-bit wavelet).  This reader must return the same floats exactly."""
    import json
    expected = None
    with open(os.path.join(_capi.REPO_ROOT, "tests", "golden", "reference_functions.jsonl")) as handle:
        for line in handle:
            if '"exr_piz_image"' in line:
                record = json.loads(line)
                if int(record["in"][0]) == index:
                    out = record["out"]
                    expected = np.array(out[2:], dtype=np.float64).reshape(int(out[1]), int(out[0]), 4)
    assert expected is not None
    host = _capi.load_host()
    w, h = C.c_int(), C.c_int()
    path = os.path.join(_capi.REPO_ROOT, "tests", "golden", "textures", name).encode()
    assert host.pathed_host_read_exr_rgba(path, C.byref(w), C.byref(h), None, 0) == 0, host.pathed_host_last_error()
    assert (h.value, w.value) == expected.shape[:2]
    data = np.zeros((h.value, w.value, 4), dtype=np.float32)
    assert host.pathed_host_read_exr_rgba(path, C.byref(w), C.byref(h), data.ctypes.data_as(C.POINTER(C.c_float)), data.size) == 0
    # the dump prints 9 significant digits: float32 round-trips exactly
    assert np.array_equal(data, expected.astype(np.float32))
    assert data[..., :3].max() > (1000.0 if index == 0 else 1.5)


def test_truncated_piz_file_is_rejected(tmp_path):
    host = _capi.load_host()
    blob = open(os.path.join(_capi.REPO_ROOT, "tests", "golden", "textures", "piz_half_33x40.exr"), "rb").read()
    broken = tmp_path / "broken.exr"
    broken.write_bytes(blob[: len(blob) - 200])
    w, h = C.c_int(), C.c_int()
    assert host.pathed_host_read_exr_rgba(str(broken).encode(), C.byref(w), C.byref(h), None, 0) != 0


def test_bmp_preview_is_what_stb_image_write_writes(tmp_path):
    """Image::write (reference src/image.cpp:156-161) goes through stbi_write_bmp; `bmp_bytes` records hold
    the files stb wrote for three small images (row padding 1, 0 and 3 bytes)."""
    import json
    host = _capi.load_host()
    seen = 0
    with open(os.path.join(_capi.REPO_ROOT, "tests", "golden", "reference_functions.jsonl")) as handle:
        for line in handle:
            if '"bmp_bytes"' not in line:
                continue
            record = json.loads(line)
            width, height = int(record["in"][0]), int(record["in"][1])
            pixels = np.array(record["in"][2:], dtype=np.uint8)
            path = str(tmp_path / ("preview%d.bmp" % seen))
            assert host.pathed_host_write_bmp_rgb8(path.encode(), width, height, pixels.ctypes.data_as(C.POINTER(C.c_uint8))) == 0
            assert open(path, "rb").read() == bytes(int(v) for v in record["out"])
            seen += 1
    assert seen == 3


def test_deeply_nested_json_is_an_error_not_a_stack_overflow(tmp_path):
    bad = tmp_path / "deep.json"
    bad.write_text("[" * 200000)
    with pytest.raises(RuntimeError, match="nesting"):
        LoadedScene(str(bad), 8, 8, asset_root="")


def _text_records():
    import json
    path = os.path.join(os.path.dirname(__file__), "golden", "reference_functions.jsonl")
    records = {"ltrim": [], "tokenize": [], "mtl_parse": []}
    for line in open(path):
        record = json.loads(line)
        if record["fn"] in records:
            records[record["fn"]].append(record)
    return records


def test_tokenizer_and_mtl_reader_match_the_reference_object_code():
    """The text layer under the OBJ / MTL readers against records dumped from the reference's own string_util.o and
    mtl_parser.o (oracle/ref_driver.cpp): its known answers (test/string_util_test.cpp:9-37) first, then tabs,
    repeated blanks, the two material libraries the reference ships and two fixture files with the corner cases
    (values ahead of any newmtl, a material defined twice, unknown statements, extra tokens, a missing file)."""
    import ctypes as C
    from pathed_amd import _capi
    host = _capi.load_host()
    host.pathed_host_tokenize.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
    host.pathed_host_ltrim.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
    host.pathed_host_parse_mtl.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
    records = _text_records()
    assert len(records["ltrim"]) >= 9 and len(records["tokenize"]) >= 8 and len(records["mtl_parse"]) == 5
    buffer = C.create_string_buffer(1 << 16)
    for record in records["ltrim"]:
        assert host.pathed_host_ltrim(record["text"].encode(), buffer, len(buffer)) == len(record["result"])
        assert buffer.value.decode() == record["result"], record
    for record in records["tokenize"]:
        count = host.pathed_host_tokenize(record["text"].encode(), buffer, len(buffer))
        tokens = buffer.value.decode().split("\n") if count else []
        assert count == len(record["tokens"]) and tokens == record["tokens"], record
    for record in records["mtl_parse"]:
        path = os.path.join(_capi.REPO_ROOT, record["file"])
        count = host.pathed_host_parse_mtl(path.encode(), buffer, len(buffer))
        assert count == len(record["materials"]) == record["baked"], record["file"]
        lines = [line for line in buffer.value.decode().split("\n") if line or count == 1][:count]
        for line, expected in zip(lines, record["materials"]):
            name, kd, ke = line.split("\t")
            assert name == expected["name"]
            assert [np.float32(v) for v in kd.split()[1:]] == [np.float32(v) for v in expected["Kd"]], (record["file"], name)
            assert [np.float32(v) for v in ke.split()[1:]] == [np.float32(v) for v in expected["Ke"]], (record["file"], name)
    # the Cornell library defines its light through Ke (the scene's only emitter)
    cornell = [r for r in records["mtl_parse"] if r["file"] == "scenes/CornellBox-Original.mtl"][0]
    assert [m["Ke"] for m in cornell["materials"] if m["name"] == "light"] == [[17.0, 12.0, 4.0]]


def test_every_in_scope_reference_scene_file_parses(tmp_path):
    """scenes/*.json holds the reference's scene descriptions unmodified; the meshes / textures most of them point at
    are git-ignored upstream.  With one-triangle / 37x29-texel stand-ins at the paths they name, every scene file that
    is in scope (no voxel media, no curves: SURVEY.md §2) must parse: material `reference`s, the scene-level
    `materials` table, MTL look-ups, `legacy` transforms, checkerboards, image textures, media."""
    import glob
    import re
    import shutil
    triangle = "v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nvt 0 0\nvt 1 0\nvt 0 1\ng a\nf 1/1/1 2/2/1 3/3/1\n"
    textures = os.path.join(_capi.REPO_ROOT, "tests", "golden", "textures")
    root = str(tmp_path)
    for directory in ("scenes", "assets", "test_scenes"):     # what the repository does ship stays in place
        shutil.copytree(os.path.join(_capi.REPO_ROOT, directory), os.path.join(root, directory),
                        ignore=shutil.ignore_patterns("dragon-standin.ply"))
    expected = {"bunny.json": (0, 1), "cornell-bunny.json": (0, 0), "material-ball.json": (0, 0), "staircase2.json": (3, 0),
                "teapot-full.json": (0, 0), "veach-ajar.json": (3, 0), "cornell-medium.json": (0, 1)}
    for name, (n_textures, n_spheres) in sorted(expected.items()):
        path = os.path.join(_capi.REPO_ROOT, "scenes", name)
        for _ in range(100):
            try:
                scene = LoadedScene(path, 32, 32, asset_root=root)
                break
            except RuntimeError as error:
                missing = re.search(r"cannot open (\S.*)$", str(error)) or re.search(r"exr (\S.*): cannot open", str(error))
                assert missing, "%s: %s" % (name, error)
                full = missing.group(1).strip()
                assert full.startswith(root), full
                os.makedirs(os.path.dirname(full), exist_ok=True)
                if full.endswith(".obj"):
                    open(full, "w").write(triangle)
                elif full.endswith((".jpg", ".jpeg")):
                    shutil.copy(os.path.join(textures, "jpeg_420_37x29.jpg"), full)
                elif full.endswith(".png"):
                    shutil.copy(os.path.join(textures, "adam7_rgb8_13x11.png"), full)
                elif full.endswith(".mtl"):
                    open(full, "w").write("newmtl a\nKd 0.5 0.5 0.5\n")
                elif full.endswith(".exr"):
                    shutil.copy(os.path.join(root, "assets", "teapot", "envmap.exr"), full)
                else:
                    raise AssertionError("%s wants %s" % (name, full))
        else:
            raise AssertionError(name + ": still missing assets after 100 stand-ins")
        d = scene.desc.contents
        assert d.n_triangles > 0 and d.n_materials > 0, name
        assert (d.n_textures, d.n_spheres) == (n_textures, n_spheres), (name, d.n_textures, d.n_spheres)
        if name in ("teapot-full.json", "cornell-medium.json"):
            assert d.n_media == 1 and any(d.geoms[i].medium == 0 for i in range(d.n_geoms)), name
    # out of scope, refused with a message that says so
    for name, reason in (("heterogeneous", "heterogeneous"), ("pbrt-curve", "pbrt-curve")):
        body = {"sensor": {"lookAt": {"origin": ["0", "0", "5"], "target": ["0", "0", "0"], "up": ["0", "1", "0"]}, "fov": "30"}, "models": []}
        if name == "heterogeneous":
            body["media"] = [{"name": "smoke", "type": "heterogeneous", "filename": "x.vol"}]
        else:
            body["models"] = [{"type": "pbrt-curve", "filename": "x.pbrt"}]
        scene_path = os.path.join(root, name + ".json")
        json.dump(body, open(scene_path, "w"))
        with pytest.raises(RuntimeError, match=reason):
            LoadedScene(scene_path, 8, 8, asset_root=root)
