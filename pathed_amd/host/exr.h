// OpenEXR scanline reader/writer, just enough for the hot path's two file uses.
//   write: reference Image::save (src/image.cpp:80-154) — 3 channels named B,G,R
//          stored HALF, uncompressed, increasing-Y scanlines.
//   read:  reference EnvironmentLight ctor (src/environment_light.cpp:14-28) uses
//          tinyexr LoadEXR -> float RGBA, row 0 = first scanline.  Supported here:
//          scanline files with NONE / ZIPS / ZIP / PIZ compression and HALF / FLOAT
//          channels (tiled, deep and the lossy codecs are rejected with a clear error).
#pragma once

#include <string>
#include <vector>

namespace pathed {

// planarRGB: r, g, b planes of width*height floats, row 0 = TOP scanline
bool writeExrHalfBGR(
    const std::string &path,
    int width, int height,
    const float *r, const float *g, const float *b,
    std::string *error
);

// writes float RGBA (4 channels A,B,G,R stored FLOAT, uncompressed); used for the
// synthetic environment maps of the stand-in scenes
bool writeExrFloatRGBA(
    const std::string &path,
    int width, int height,
    const float *rgba,
    std::string *error
);

// out: 4*width*height floats RGBA, missing channels = 0 (A = 1), row 0 = first scanline
bool readExrRGBA(
    const std::string &path,
    int *width, int *height,
    std::vector<float> *rgba,
    std::string *error
);

unsigned short floatToHalf(float value);
float halfToFloat(unsigned short half);

}  // namespace pathed
