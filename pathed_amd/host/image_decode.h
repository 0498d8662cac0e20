// 8-bit RGB image files for image-texture albedos: what the reference gets from
// stbi_load(path, &w, &h, &channels, 3) (src/texture.cpp:12-31).  Lossless formats only, where
// any conforming decoder returns the same bytes as stb_image: PNG (non-interlaced; grey, RGB,
// palette, with or without alpha, 1-16 bits) and binary PNM (P5 / P6, maxval <= 255).
// JPEG is rejected with a message: its decoded bytes depend on the decoder's IDCT and
// upsampling, so a second implementation would not reproduce stb_image's texels.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace pathed {

// rgb: 3*width*height bytes, row 0 = first row of the file
bool loadImageRgb8(const std::string &path, int *width, int *height, std::vector<uint8_t> *rgb, std::string *error);

}  // namespace pathed
