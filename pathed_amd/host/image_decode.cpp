#include "image_decode.h"

#include <zlib.h>

#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>

namespace pathed {

namespace {

uint32_t bigEndian32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int paeth(int a, int b, int c)
{
    const int p = a + b - c;
    const int pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    if (pa <= pb && pa <= pc) { return a; }
    if (pb <= pc) { return b; }
    return c;
}

bool decodePng(const std::vector<uint8_t> &file, int *width, int *height, std::vector<uint8_t> *rgb, std::string *error)
{
    auto fail = [&](const std::string &message) { *error = "png: " + message; return false; };
    size_t at = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, colorType = -1, interlace = 0;
    std::vector<uint8_t> palette, compressed;
    bool sawHeader = false, sawEnd = false;
    while (!sawEnd) {
        if (at + 8 > file.size()) { return fail("truncated file"); }
        const uint32_t length = bigEndian32(&file[at]);
        const std::string type(reinterpret_cast<const char *>(&file[at + 4]), 4);
        if ((uint64_t)at + 12 + length > file.size()) { return fail("truncated chunk " + type); }
        const uint8_t *data = &file[at + 8];
        if (type == "IHDR") {
            if (length != 13) { return fail("bad IHDR"); }
            w = bigEndian32(data);
            h = bigEndian32(data + 4);
            depth = data[8];
            colorType = data[9];
            interlace = data[12];
            if (data[10] != 0 || data[11] != 0) { return fail("unknown compression / filter method"); }
            sawHeader = true;
        } else if (type == "PLTE") {
            palette.assign(data, data + length);
        } else if (type == "IDAT") {
            compressed.insert(compressed.end(), data, data + length);
        } else if (type == "IEND") {
            sawEnd = true;
        }
        at += 12 + (size_t)length;
    }
    if (!sawHeader) { return fail("no IHDR"); }
    if (w == 0 || h == 0 || w > 65535 || h > 65535) { return fail("image size out of range (1..65535)"); }
    if ((uint64_t)w * h > (1ull << 28)) { return fail("image larger than 2^28 pixels"); }
    if (interlace != 0) { return fail("interlaced (Adam7) files are not supported: re-save without interlacing"); }
    int channels;
    switch (colorType) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: return fail("unknown colour type");
    }
    const bool depthOk = (colorType == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16))
        || (colorType == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8))
        || ((colorType == 2 || colorType == 4 || colorType == 6) && (depth == 8 || depth == 16));
    if (!depthOk) { return fail("bit depth not allowed for the colour type"); }
    if (colorType == 3 && palette.size() < 3) { return fail("palette image without PLTE"); }

    const size_t rowBytes = ((size_t)w * channels * depth + 7) / 8;
    const size_t pixelBytes = (size_t)(channels * depth + 7) / 8;   // filter distance, >= 1
    std::vector<uint8_t> raw((rowBytes + 1) * h);
    uLongf rawSize = (uLongf)raw.size();
    const int status = uncompress(raw.data(), &rawSize, compressed.data(), (uLong)compressed.size());
    if (status != Z_OK || rawSize != raw.size()) { return fail("corrupt image data (zlib)"); }

    // undo the scanline filters in place
    std::vector<uint8_t> zeroRow(rowBytes, 0);
    for (uint32_t y = 0; y < h; y++) {
        uint8_t *row = &raw[(rowBytes + 1) * y + 1];
        const uint8_t *above = y ? &raw[(rowBytes + 1) * (y - 1) + 1] : zeroRow.data();
        const int filter = row[-1];
        for (size_t i = 0; i < rowBytes; i++) {
            const int left = i >= pixelBytes ? row[i - pixelBytes] : 0;
            const int up = above[i];
            const int upLeft = i >= pixelBytes ? above[i - pixelBytes] : 0;
            int predicted;
            switch (filter) {
            case 0: predicted = 0; break;
            case 1: predicted = left; break;
            case 2: predicted = up; break;
            case 3: predicted = (left + up) >> 1; break;
            case 4: predicted = paeth(left, up, upLeft); break;
            default: return fail("unknown scanline filter");
            }
            row[i] = (uint8_t)(row[i] + predicted);
        }
    }

    // expand to 8-bit RGB the way stb_image does for req_comp = 3: 16-bit samples keep their
    // high byte, 1/2/4-bit grey is scaled to 0..255, palette indices are looked up, alpha is dropped
    rgb->resize((size_t)3 * w * h);
    static const int depthScale[9] = { 0, 0xff, 0x55, 0, 0x11, 0, 0, 0, 0x01 };
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t *row = &raw[(rowBytes + 1) * y + 1];
        for (uint32_t x = 0; x < w; x++) {
            uint8_t sample[4] = { 0, 0, 0, 0 };
            for (int c = 0; c < channels; c++) {
                const size_t index = (size_t)x * channels + c;
                if (depth == 16) { sample[c] = row[2 * index]; }
                else if (depth == 8) { sample[c] = row[index]; }
                else {
                    const size_t bit = index * depth;
                    const int shift = 8 - depth - (int)(bit & 7);
                    sample[c] = (uint8_t)((row[bit >> 3] >> shift) & ((1 << depth) - 1));
                }
            }
            uint8_t *out = &(*rgb)[3 * ((size_t)y * w + x)];
            if (colorType == 3) {
                const size_t entry = sample[0];
                if (3 * entry + 2 >= palette.size()) { return fail("palette index out of range"); }
                out[0] = palette[3 * entry + 0];
                out[1] = palette[3 * entry + 1];
                out[2] = palette[3 * entry + 2];
            } else if (channels <= 2) {
                const uint8_t grey = depth < 8 ? (uint8_t)(sample[0] * depthScale[depth]) : sample[0];
                out[0] = out[1] = out[2] = grey;
            } else {
                out[0] = sample[0];
                out[1] = sample[1];
                out[2] = sample[2];
            }
        }
    }
    *width = (int)w;
    *height = (int)h;
    return true;
}

bool decodePnm(const std::vector<uint8_t> &file, int *width, int *height, std::vector<uint8_t> *rgb, std::string *error)
{
    auto fail = [&](const std::string &message) { *error = "pnm: " + message; return false; };
    const int channels = file[1] == '6' ? 3 : 1;
    size_t at = 2;
    long values[3] = { 0, 0, 0 };
    for (int k = 0; k < 3; k++) {
        // whitespace and comments
        while (at < file.size()) {
            if (file[at] == '#') { while (at < file.size() && file[at] != '\n') { at++; } }
            else if (file[at] == ' ' || file[at] == '\t' || file[at] == '\n' || file[at] == '\r') { at++; }
            else { break; }
        }
        if (at >= file.size() || file[at] < '0' || file[at] > '9') { return fail("malformed header"); }
        while (at < file.size() && file[at] >= '0' && file[at] <= '9') { values[k] = values[k] * 10 + (file[at] - '0'); at++; if (values[k] > 70000) { return fail("header value out of range"); } }
    }
    at++;  // the single whitespace byte after maxval
    if (values[0] < 1 || values[1] < 1 || values[0] > 65535 || values[1] > 65535) { return fail("image size out of range (1..65535)"); }
    if (values[2] < 1 || values[2] > 255) { return fail("only maxval <= 255 is supported"); }
    const size_t count = (size_t)values[0] * values[1];
    if (count > ((size_t)1 << 28)) { return fail("image larger than 2^28 pixels"); }
    if (at + count * channels > file.size()) { return fail("truncated pixel data"); }
    rgb->resize(3 * count);
    for (size_t k = 0; k < count; k++) {
        for (int c = 0; c < 3; c++) { (*rgb)[3 * k + c] = file[at + k * channels + (channels == 3 ? c : 0)]; }
    }
    *width = (int)values[0];
    *height = (int)values[1];
    return true;
}

}  // namespace

bool loadImageRgb8(const std::string &path, int *width, int *height, std::vector<uint8_t> *rgb, std::string *error)
{
    std::ifstream stream(path, std::ios::binary);
    if (!stream) { *error = "cannot open " + path; return false; }
    const std::vector<uint8_t> file((std::istreambuf_iterator<char>(stream)), std::istreambuf_iterator<char>());
    static const uint8_t pngSignature[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    bool ok;
    if (file.size() >= 8 && std::memcmp(file.data(), pngSignature, 8) == 0) {
        ok = decodePng(file, width, height, rgb, error);
    } else if (file.size() >= 3 && file[0] == 'P' && (file[1] == '5' || file[1] == '6')) {
        ok = decodePnm(file, width, height, rgb, error);
    } else if (file.size() >= 2 && file[0] == 0xFF && file[1] == 0xD8) {
        *error = "JPEG textures are not supported (the decoded texels depend on the decoder): convert the file to PNG";
        ok = false;
    } else {
        *error = "unknown image format (PNG and binary PNM are supported)";
        ok = false;
    }
    if (!ok) { *error = path + ": " + *error; }
    return ok;
}

}  // namespace pathed
