#!/usr/bin/env python3
"""Writes the small PNG / PNM texture files the texture tests use (tests/golden/textures/).

The files are inputs only: the expected texels and lookups come from the reference's own
stb_image + Texture::lookup through oracle/_ref/refdump (records `texture_image`,
`texture_lookup` of reference_functions.jsonl).  Every PNG scanline-filter type, every colour
type and the sub-byte / 16-bit depths appear at least once.
"""
import os
import struct
import zlib

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "textures")


def chunk(kind, data):
    return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)


def paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def filter_rows(rows, bpp, filters):
    """rows: list of bytes per scanline (unfiltered); filters: filter type per row."""
    out = bytearray()
    previous = bytes(len(rows[0]))
    for row, kind in zip(rows, filters):
        line = bytearray(len(row))
        for i, value in enumerate(row):
            left = row[i - bpp] if i >= bpp else 0
            up = previous[i]
            up_left = previous[i - bpp] if i >= bpp else 0
            predicted = [0, left, up, (left + up) >> 1, paeth(left, up, up_left)][kind]
            line[i] = (value - predicted) & 0xFF
        out += bytes([kind]) + line
        previous = row
    return bytes(out)


def write_png(name, width, height, depth, color_type, rows, bpp, palette=None):
    filters = [(y + 1) % 5 for y in range(height)]   # 1, 2, 3, 4, 0, 1, ...
    data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, depth, color_type, 0, 0, 0))
    if palette is not None:
        data += chunk(b"PLTE", bytes(palette))
    body = zlib.compress(filter_rows(rows, bpp, filters), 9)
    half = len(body) // 2   # two IDAT chunks: decoders must concatenate them
    data += chunk(b"IDAT", body[:half]) + chunk(b"IDAT", body[half:]) + chunk(b"IEND", b"")
    with open(os.path.join(OUT, name), "wb") as handle:
        handle.write(data)


def write_png_adam7(name, width, height, depth, color_type, pixel_rows, channels, palette=None):
    """pixel_rows[y][x] = tuple of samples; writes the seven Adam7 passes, each filtered on its own."""
    x0, y0 = [0, 4, 0, 2, 0, 1, 0], [0, 0, 4, 0, 2, 0, 1]
    dx, dy = [8, 8, 4, 4, 2, 2, 1], [8, 8, 8, 4, 4, 2, 2]
    bpp = max(1, channels * depth // 8)
    stream = b""
    for k in range(7):
        columns = list(range(x0[k], width, dx[k]))
        lines = list(range(y0[k], height, dy[k]))
        if not columns or not lines:
            continue
        rows = []
        for y in lines:
            samples = [s for x in columns for s in pixel_rows[y][x]]
            if depth == 8:
                rows.append(bytes(samples))
            elif depth == 16:
                rows.append(b"".join(struct.pack(">H", s) for s in samples))
            else:
                rows.append(pack_bits(samples, depth))
        stream += filter_rows(rows, bpp, [(y + k) % 5 for y in range(len(rows))])
    data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, depth, color_type, 0, 0, 1))
    if palette is not None:
        data += chunk(b"PLTE", bytes(palette))
    data += chunk(b"IDAT", zlib.compress(stream, 9)) + chunk(b"IEND", b"")
    with open(os.path.join(OUT, name), "wb") as handle:
        handle.write(data)


def pack_bits(values, depth):
    out, acc, bits = bytearray(), 0, 0
    for value in values:
        acc = (acc << depth) | value
        bits += depth
        if bits == 8:
            out.append(acc)
            acc, bits = 0, 0
    if bits:
        out.append(acc << (8 - bits))
    return bytes(out)


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(7)

    rgb = rng.integers(0, 256, (5, 7, 3), dtype=np.uint8)
    write_png("rgb8_7x5.png", 7, 5, 8, 2, [rgb[y].tobytes() for y in range(5)], 3)

    rgba16 = rng.integers(0, 65536, (3, 4, 4), dtype=np.uint16)
    write_png("rgba16_4x3.png", 4, 3, 16, 6, [rgba16[y].astype(">u2").tobytes() for y in range(3)], 8)

    grey_alpha = rng.integers(0, 256, (4, 3, 2), dtype=np.uint8)
    write_png("greyalpha8_3x4.png", 3, 4, 8, 4, [grey_alpha[y].tobytes() for y in range(4)], 2)

    grey2 = rng.integers(0, 4, (6, 5), dtype=np.uint8)
    write_png("grey2_5x6.png", 5, 6, 2, 0, [pack_bits(grey2[y].tolist(), 2) for y in range(6)], 1)

    palette = rng.integers(0, 256, 16 * 3, dtype=np.uint8).tolist()
    indices = rng.integers(0, 16, (4, 5), dtype=np.uint8)
    write_png("palette4_5x4.png", 5, 4, 4, 3, [pack_bits(indices[y].tolist(), 4) for y in range(4)], 1, palette)

    ppm = rng.integers(0, 256, (3, 2, 3), dtype=np.uint8)
    with open(os.path.join(OUT, "rgb_2x3.ppm"), "wb") as handle:
        handle.write(b"P6\n# a comment\n2 3\n255\n" + ppm.tobytes())

    # a larger smooth image for the rendered-scene parity test
    y, x = np.mgrid[0:32, 0:48]
    wood = np.stack([128 + 100 * np.sin(x / 3.0 + y / 11.0), 90 + 60 * np.cos(y / 2.0), 40 + 30 * np.sin((x + y) / 5.0)], axis=2)
    wood = np.clip(wood, 0, 255).astype(np.uint8)
    write_png("wood_48x32.png", 48, 32, 8, 2, [wood[r].tobytes() for r in range(32)], 3)


if __name__ == "__main__":
    main()


def adam7_fixtures():
    rng = np.random.default_rng(23)
    rgb = rng.integers(0, 256, (11, 13, 3), dtype=np.uint8)
    write_png_adam7("adam7_rgb8_13x11.png", 13, 11, 8, 2, [[tuple(int(v) for v in rgb[y, x]) for x in range(13)] for y in range(11)], 3)
    grey = rng.integers(0, 16, (9, 10), dtype=np.uint8)
    write_png_adam7("adam7_grey4_10x9.png", 10, 9, 4, 0, [[(int(grey[y, x]),) for x in range(10)] for y in range(9)], 1)
    tiny = rng.integers(0, 65536, (2, 3, 4), dtype=np.uint16)   # fewer pixels than passes
    write_png_adam7("adam7_rgba16_3x2.png", 3, 2, 16, 6, [[tuple(int(v) for v in tiny[y, x]) for x in range(3)] for y in range(2)], 4)


def jpeg_fixtures():
    """JPEG files written by Pillow (libjpeg): the sampling factors, scan types and edge cases the decoder
    must reproduce stb_image's bytes for."""
    from PIL import Image
    rng = np.random.default_rng(11)
    y, x = np.mgrid[0:29, 0:37]
    smooth = np.stack([128 + 100 * np.sin(x / 3.0 + y / 7.0), 100 + 80 * np.cos(y / 2.5), 60 + 50 * np.sin((x - y) / 4.0)], axis=2)
    smooth = np.clip(smooth + rng.normal(0, 6, smooth.shape), 0, 255).astype(np.uint8)
    noisy = rng.integers(0, 256, (21, 26, 3), dtype=np.uint8)          # clamping, large coefficients
    noisy[5:12, 4:15] = [255, 0, 0]
    colour = Image.fromarray(smooth, "RGB")
    colour.save(os.path.join(OUT, "jpeg_444_37x29.jpg"), quality=92, subsampling=0)
    colour.save(os.path.join(OUT, "jpeg_420_37x29.jpg"), quality=75, subsampling=2)
    colour.save(os.path.join(OUT, "jpeg_422_37x29.jpg"), quality=80, subsampling=1, optimize=True)
    colour.save(os.path.join(OUT, "jpeg_progressive_420_37x29.jpg"), quality=85, subsampling=2, progressive=True)
    colour.save(os.path.join(OUT, "jpeg_progressive_444_37x29.jpg"), quality=60, subsampling=0, progressive=True, optimize=True)
    Image.fromarray(noisy, "RGB").save(os.path.join(OUT, "jpeg_noise_420_26x21.jpg"), quality=95, subsampling=2)
    Image.fromarray(smooth[:, :, 0], "L").save(os.path.join(OUT, "jpeg_grey_37x29.jpg"), quality=80)
    Image.fromarray(smooth[:9, :1], "RGB").save(os.path.join(OUT, "jpeg_420_1x9.jpg"), quality=85, subsampling=2)
    big = np.kron(smooth, np.ones((3, 3, 1), dtype=np.uint8))[:70, :100]
    try:
        Image.fromarray(big, "RGB").save(os.path.join(OUT, "jpeg_restart_420_100x70.jpg"), quality=70, subsampling=2, restart_marker_blocks=3)
    except TypeError:
        Image.fromarray(big, "RGB").save(os.path.join(OUT, "jpeg_restart_420_100x70.jpg"), quality=70, subsampling=2)


if __name__ == "__main__":
    jpeg_fixtures()
    adam7_fixtures()
