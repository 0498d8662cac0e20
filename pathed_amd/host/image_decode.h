// 8-bit RGB image files for image-texture albedos: what the reference gets from
// stbi_load(path, &w, &h, &channels, 3) (src/texture.cpp:12-31).
//   PNG   grey, RGB, palette, with or without alpha, 1-16 bits, plain or Adam7-interlaced
//   PNM   binary P5 / P6, maxval <= 255
//   JPEG  baseline and progressive, 8-bit, greyscale or three components.  A JPEG's decoded bytes depend
//         on the decoder's IDCT, chroma upsampling and colour conversion, so those follow stb_image's
//         arithmetic; the output is checked byte for byte against stb_image (tests/test_textures.py).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace pathed {

// rgb: 3*width*height bytes, row 0 = first row of the file
bool loadImageRgb8(const std::string &path, int *width, int *height, std::vector<uint8_t> *rgb, std::string *error);

}  // namespace pathed
