#!/usr/bin/env python3
"""Generates the stand-in assets the reference's scene files point at but does not ship.

The reference's `assets/` directory is git-ignored upstream: the Veach-MIS
plates, the teapot meshes + envmap.exr, dragon.obj + its HDR are absent.  Everything written
here is SYNTHETIC (labelled so in DESIGN.md); file formats are the reference's own
(binary-LE PLY as src/ply_parser.cpp reads it, OBJ, scanline EXR).

  assets/mis-pbrt/geometry/{plate1..4,floor}.ply   Veach-MIS layout (publicly known numbers)
  assets/teapot/{Mesh000,Mesh001}.obj              closed smooth-shaded glass stand-ins
  assets/teapot/envmap.exr                         procedural sky, 256x128 float RGBA
  assets/dragon.obj -> assets/dragon.ply is NOT used: scenes/dragon.json reads an OBJ, the
      large-BVH stand-in is scenes/dragon-standin.json + assets/dragon-standin.ply (--dragon N)
  assets/20060807_wells6_hd.exr                    procedural sky for scenes/dragon*.json
  assets/cornell-volume-caustic/{bounds,CornellBox-Frame}.obj   for scenes/cornell-medium.json: the gas container (a box
                                                   around the glass sphere, below the light) and the Cornell room
                                                   without its two boxes (cut from scenes/CornellBox-Original.obj)
  test_scenes/1_pixel_test.exr                     1000x500, one texel (col 753,row 239)=1e4,
                                                   as decoded from the reference's file
"""
import argparse
import ctypes as C
import math
import os
import struct
import sys

import numpy as np

REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO_ROOT)


def write_ply(path, vertices, faces):
    vertices = np.asarray(vertices, dtype="<f4")
    faces = np.asarray(faces, dtype="<i4")
    header = (
        "ply\nformat binary_little_endian 1.0\nelement vertex %d\n"
        "property float x\nproperty float y\nproperty float z\n"
        "element face %d\nproperty list uchar int vertex_indices\nend_header\n" % (len(vertices), len(faces))
    )
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "wb") as handle:
        handle.write(header.encode())
        handle.write(vertices.tobytes())
        records = np.zeros(len(faces), dtype=[("n", "u1"), ("i", "<i4", 3)])
        records["n"] = 3
        records["i"] = faces
        handle.write(records.tobytes())


def quad_ply(path, corners):
    """corners: 4 points counter-clockwise seen from the side the normal should face."""
    write_ply(path, corners, [[0, 1, 2], [0, 2, 3]])


def make_mis():
    root = os.path.join(REPO_ROOT, "assets", "mis-pbrt", "geometry")
    # (y, z) of the far and near edge of each plate; x spans [-4, 4]
    plates = [
        ((-2.70651, 0.25609), (-2.08375, -0.526323)),
        ((-3.28825, 1.36972), (-2.83856, 0.476536)),
        ((-3.73096, 2.70046), (-3.43378, 1.74564)),
        ((-3.99615, 4.0667), (-3.82069, 3.08221)),
    ]
    for index, ((y0, z0), (y1, z1)) in enumerate(plates, start=1):
        # near edge (larger z) first so the normal (e1 x e2) faces up / towards the lights
        corners = [(-4, y0, z0), (4, y0, z0), (4, y1, z1), (-4, y1, z1)]
        quad_ply(os.path.join(root, "plate%d.ply" % index), corners)
    y = -4.14615
    floor = [(-10, y, 10), (10, y, 10), (10, y, -10), (-10, y, -10)]
    wall = [(-10, y, -10), (10, y, -10), (10, y + 20, -10), (-10, y + 20, -10)]
    vertices = floor + wall
    write_ply(os.path.join(root, "floor.ply"), vertices, [[0, 1, 2], [0, 2, 3], [4, 5, 6], [4, 6, 7]])


def uv_sphere(center, radii, stacks, slices):
    """Closed smooth mesh with per-vertex normals: returns (vertices, normals, faces)."""
    vertices, normals, faces = [], [], []
    for i in range(stacks + 1):
        theta = math.pi * i / stacks
        for j in range(slices):
            phi = 2 * math.pi * j / slices
            n = (math.sin(theta) * math.cos(phi), math.cos(theta), math.sin(theta) * math.sin(phi))
            vertices.append((center[0] + radii[0] * n[0], center[1] + radii[1] * n[1], center[2] + radii[2] * n[2]))
            # normal of an ellipsoid: gradient direction
            g = (n[0] / radii[0], n[1] / radii[1], n[2] / radii[2])
            length = math.sqrt(sum(c * c for c in g))
            normals.append(tuple(c / length for c in g))
    for i in range(stacks):
        for j in range(slices):
            a = i * slices + j
            b = i * slices + (j + 1) % slices
            c = (i + 1) * slices + j
            d = (i + 1) * slices + (j + 1) % slices
            if i > 0:
                faces.append((a, b, c))
            if i < stacks - 1:
                faces.append((b, d, c))
    return vertices, normals, faces


def write_obj(path, vertices, normals, faces):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as handle:
        handle.write("# synthetic stand-in written by tools/make_assets.py\n")
        for v in vertices:
            handle.write("v %.6f %.6f %.6f\n" % v)
        for n in normals:
            handle.write("vn %.6f %.6f %.6f\n" % n)
        handle.write("g standin\n")
        for a, b, c in faces:
            handle.write("f %d//%d %d//%d %d//%d\n" % (a + 1, a + 1, b + 1, b + 1, c + 1, c + 1))


def sky(width, height, sun_dir=(0.4, 0.7, 0.3), sun_power=60.0):
    """Lat-long RGBA float sky: gradient + a soft sun; row 0 = theta 0 (+y)."""
    sun = np.array(sun_dir, dtype=np.float64)
    sun /= np.linalg.norm(sun)
    theta = (np.arange(height) + 0.5) / height * math.pi
    phi = (np.arange(width) + 0.5) / width * 2 * math.pi
    t, p = np.meshgrid(theta, phi, indexing="ij")
    d = np.stack([np.sin(t) * np.cos(p), np.cos(t), np.sin(t) * np.sin(p)], axis=-1)
    up = np.clip(d[..., 1], -1, 1)
    horizon = np.exp(-8 * np.abs(up))
    base = np.stack([0.25 + 0.35 * horizon, 0.35 + 0.35 * horizon, 0.65 + 0.25 * horizon], axis=-1)
    base *= np.where(up[..., None] > 0, 1.0, 0.25)
    cosine = np.clip((d * sun).sum(-1), 0, 1)
    glow = sun_power * np.power(cosine, 400.0) + 2.0 * np.power(cosine, 20.0)
    rgb = base + glow[..., None] * np.array([1.0, 0.9, 0.7])
    rgba = np.concatenate([rgb, np.ones_like(rgb[..., :1])], axis=-1)
    return np.ascontiguousarray(rgba, dtype=np.float32)


def write_exr(path, rgba):
    from pathed_amd import _capi

    host = _capi.load_host()
    os.makedirs(os.path.dirname(path), exist_ok=True)
    height, width = rgba.shape[:2]
    code = host.pathed_host_write_exr_float_rgba(path.encode(), width, height, rgba.ctypes.data_as(C.POINTER(C.c_float)))
    if code != 0:
        raise RuntimeError(host.pathed_host_last_error().decode())


def make_teapot():
    root = os.path.join(REPO_ROOT, "assets", "teapot")
    # body and lid knob of a "teapot": two closed ellipsoids above the checkerboard plane y = 0
    v, n, f = uv_sphere((0.0, 3.2, 0.0), (5.0, 3.2, 5.0), 48, 96)
    write_obj(os.path.join(root, "Mesh000.obj"), v, n, f)
    v, n, f = uv_sphere((0.0, 7.3, 0.0), (1.1, 0.9, 1.1), 24, 48)
    write_obj(os.path.join(root, "Mesh001.obj"), v, n, f)
    write_exr(os.path.join(root, "envmap.exr"), sky(256, 128))


def make_env_test():
    rgba = np.zeros((500, 1000, 4), dtype=np.float32)
    rgba[..., 3] = 1.0
    rgba[239, 753, 0:3] = 10000.0
    write_exr(os.path.join(REPO_ROOT, "test_scenes", "1_pixel_test.exr"), rgba)


def make_cornell_medium():
    """scenes/cornell-medium.json (reference, unmodified) reads two OBJs the reference does not ship."""
    root = os.path.join(REPO_ROOT, "assets", "cornell-volume-caustic")
    os.makedirs(root, exist_ok=True)
    lines = open(os.path.join(REPO_ROOT, "scenes", "CornellBox-Original.obj")).read().split("\n")
    begin = next(i for i, line in enumerate(lines) if line.startswith("## Object shortBox"))
    end = next(i for i, line in enumerate(lines) if line.startswith("## Object light"))
    with open(os.path.join(root, "CornellBox-Frame.obj"), "w") as handle:
        handle.write("# synthetic stand-in written by tools/make_assets.py: CornellBox-Original.obj without shortBox / tallBox\n")
        handle.write("\n".join(lines[:begin] + lines[end:]))
    lo, hi = (-0.9, 0.1, -0.9), (0.9, 1.9, 0.9)
    corners = [(x, y, z) for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])]
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    with open(os.path.join(root, "bounds.obj"), "w") as handle:
        handle.write("# synthetic stand-in written by tools/make_assets.py: the gas container of scenes/cornell-medium.json\n")
        for corner in corners:
            handle.write("v %.6f %.6f %.6f\n" % corner)
        handle.write("g bounds\n")
        for quad in quads:
            handle.write("f %d %d %d %d\n" % tuple(index + 1 for index in quad))


def parallel_chunks(function, array, min_chunk=1 << 18):
    """function over chunks of `array` on a thread pool (numpy releases the GIL inside its loops), results concatenated."""
    workers = max(1, min(64, os.cpu_count() or 1))
    if len(array) < 2 * min_chunk or workers == 1:
        return function(array)
    from concurrent.futures import ThreadPoolExecutor
    pieces = np.array_split(array, max(workers, min(4 * workers, len(array) // min_chunk)))
    with ThreadPoolExecutor(workers) as pool:
        return np.concatenate(list(pool.map(function, pieces)), axis=0)


def icosphere(subdivisions):
    t = (1.0 + math.sqrt(5.0)) / 2.0
    verts = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
             (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    faces = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2),
             (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5),
             (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = np.array(verts, dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array(faces, dtype=np.int64)
    for _ in range(subdivisions):
        # every edge once: (lo, hi) packed into one int64 key (a 1-D unique sorts integers, several times faster than
        # unique rows; same order as the row-wise unique, so the mesh is the one earlier versions of this script wrote)
        edges = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], axis=0)
        lo, hi = edges.min(axis=1), edges.max(axis=1)
        # (the sort of np.unique without return_inverse is numpy's vectorised one; the inverse is a binary search per edge,
        # spread over the host's cores: the same arrays np.unique(..., return_inverse=True) returns, in a third of the time)
        all_keys = lo * np.int64(len(v)) + hi
        keys = np.unique(all_keys)
        inverse = parallel_chunks(lambda chunk: np.searchsorted(keys, chunk), all_keys)
        unique = np.stack([keys // len(v), keys % len(v)], axis=1)
        mid = v[unique[:, 0]] + v[unique[:, 1]]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        base = len(v)
        v = np.concatenate([v, mid], axis=0)
        n = len(f)
        ab, bc, ca = base + inverse[:n], base + inverse[n:2 * n], base + inverse[2 * n:]
        a, b, c = f[:, 0], f[:, 1], f[:, 2]
        f = np.concatenate([
            np.stack([a, ab, ca], 1), np.stack([b, bc, ab], 1), np.stack([c, ca, bc], 1), np.stack([ab, bc, ca], 1)
        ], axis=0)
    return v, f


def value_noise(points, seed):
    """Cheap deterministic multi-octave lattice noise on unit vectors."""
    rng = np.random.default_rng(seed)
    octaves = []
    amplitude, frequency = 1.0, 2.0
    for _ in range(6):
        direction = rng.normal(size=(8, 3))
        phase = rng.uniform(0, 2 * math.pi, size=8)
        octaves.append((amplitude, frequency, direction, phase))
        amplitude *= 0.55
        frequency *= 2.1

    def noise(chunk):   # elementwise in the points: chunks on a thread pool give the array the whole-array call gives
        total = np.zeros(len(chunk))
        for amplitude, frequency, direction, phase in octaves:
            # written out component by component: a BLAS product rounds differently for different chunk sizes
            projected = (chunk[:, 0:1] * direction[None, :, 0] + chunk[:, 1:2] * direction[None, :, 1]) + chunk[:, 2:3] * direction[None, :, 2]
            total += amplitude * np.sin(projected * frequency + phase).mean(axis=1)
        return total
    return parallel_chunks(noise, points)


def make_dragon(subdivisions, path=None):
    """Displaced icosphere: 20 * 4^subdivisions triangles (10 -> 21 M, 9 -> 5.2 M, 8 -> 1.3 M)."""
    v, f = icosphere(subdivisions)
    radius = 60.0 * (1.0 + 0.18 * value_noise(v, 7))
    vertices = v * radius[:, None] + np.array([0.0, 0.0, 25.0])
    path = path or os.path.join(REPO_ROOT, "assets", "dragon-standin.ply")
    write_ply(path, vertices, f)
    return len(f)


def ply_faces(path):
    """Face count a PLY header declares, 0 when the file is missing."""
    if not os.path.exists(path):
        return 0
    with open(path, "rb") as handle:
        for line in handle.read(400).split(b"\n"):
            if line.startswith(b"element face"):
                return int(line.split()[2])
    return 0


def make_dragon_variant(subdivisions):
    """assets/dragon-standin-<n>.ply + assets/dragon-standin-<n>.json (scenes/dragon-standin.json pointed at it): sized
    variants of the stand-in with names of their own, so that tests which want different sizes do not regenerate one file
    in turn -- and can all be generated before a test process makes its first GPU call."""
    import json
    mesh = os.path.join(REPO_ROOT, "assets", "dragon-standin-%d.ply" % subdivisions)
    if ply_faces(mesh) != 20 * 4 ** subdivisions:
        make_dragon(subdivisions, mesh)
    scene = json.load(open(os.path.join(REPO_ROOT, "scenes", "dragon-standin.json")))
    for model in scene["models"]:
        if model.get("type") == "ply":
            model["filename"] = "assets/dragon-standin-%d.ply" % subdivisions
    with open(os.path.join(REPO_ROOT, "assets", "dragon-standin-%d.json" % subdivisions), "w") as handle:
        json.dump(scene, handle, indent=2)


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--dragon", type=int, default=-1, help="icosphere subdivisions of the large-BVH stand-in (off by default)")
    parser.add_argument("--dragon-variants", default="", help="comma-separated subdivision levels: writes assets/dragon-standin-<n>.ply / .json each")
    parser.add_argument("--force", action="store_true")
    args = parser.parse_args()

    marker = os.path.join(REPO_ROOT, "assets", ".generated-v2")
    if args.force or not os.path.exists(marker):
        make_mis()
        make_teapot()
        make_env_test()
        make_cornell_medium()
        write_exr(os.path.join(REPO_ROOT, "assets", "20060807_wells6_hd.exr"), sky(256, 128, sun_dir=(0.3, 0.2, 0.8)))
        with open(marker, "w") as handle:
            handle.write("tools/make_assets.py\n")
        print("assets written under assets/ and test_scenes/")
    if args.dragon >= 0:
        if ply_faces(os.path.join(REPO_ROOT, "assets", "dragon-standin.ply")) != 20 * 4 ** args.dragon or args.force:
            make_dragon(args.dragon)
        print("dragon stand-in: %d triangles" % (20 * 4 ** args.dragon))
    for level in [int(item) for item in args.dragon_variants.split(",") if item.strip()]:
        make_dragon_variant(level)


if __name__ == "__main__":
    main()
