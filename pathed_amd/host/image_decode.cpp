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
    if (interlace > 1) { return fail("unknown interlace method"); }
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

    const size_t pixelBytes = (size_t)(channels * depth + 7) / 8;   // filter distance, >= 1
    auto rowBytesFor = [&](uint32_t pixels) { return ((size_t)pixels * channels * depth + 7) / 8; };

    // Adam7: seven reduced images, each filtered on its own (PNG specification, section 8.2)
    static const int xOrigin[7] = { 0, 4, 0, 2, 0, 1, 0 }, yOrigin[7] = { 0, 0, 4, 0, 2, 0, 1 };
    static const int xSpacing[7] = { 8, 8, 4, 4, 2, 2, 1 }, ySpacing[7] = { 8, 8, 8, 4, 4, 2, 2 };
    struct Pass { uint32_t w, h; int x0, y0, dx, dy; };
    std::vector<Pass> passes;
    if (interlace == 0) {
        passes.push_back({ w, h, 0, 0, 1, 1 });
    } else {
        for (int k = 0; k < 7; k++) {
            const uint32_t pw = (w - (uint32_t)xOrigin[k] + (uint32_t)xSpacing[k] - 1) / (uint32_t)xSpacing[k];
            const uint32_t ph = (h - (uint32_t)yOrigin[k] + (uint32_t)ySpacing[k] - 1) / (uint32_t)ySpacing[k];
            if ((uint32_t)xOrigin[k] < w && (uint32_t)yOrigin[k] < h && pw && ph) { passes.push_back({ pw, ph, xOrigin[k], yOrigin[k], xSpacing[k], ySpacing[k] }); }
        }
    }
    size_t rawSizeWanted = 0;
    for (const Pass &pass : passes) { rawSizeWanted += (rowBytesFor(pass.w) + 1) * pass.h; }
    std::vector<uint8_t> raw(rawSizeWanted);
    uLongf rawSize = (uLongf)raw.size();
    const int status = uncompress(raw.data(), &rawSize, compressed.data(), (uLong)compressed.size());
    if (status != Z_OK || rawSize != raw.size()) { return fail("corrupt image data (zlib)"); }

    // expand to 8-bit RGB the way stb_image does for req_comp = 3: 16-bit samples keep their
    // high byte, 1/2/4-bit grey is scaled to 0..255, palette indices are looked up, alpha is dropped
    rgb->resize((size_t)3 * w * h);
    static const int depthScale[9] = { 0, 0xff, 0x55, 0, 0x11, 0, 0, 0, 0x01 };
    size_t passStart = 0;
    for (const Pass &pass : passes) {
        const size_t rowBytes = rowBytesFor(pass.w);
        // undo the scanline filters in place
        std::vector<uint8_t> zeroRow(rowBytes, 0);
        for (uint32_t y = 0; y < pass.h; y++) {
            uint8_t *row = &raw[passStart + (rowBytes + 1) * y + 1];
            const uint8_t *above = y ? &raw[passStart + (rowBytes + 1) * (y - 1) + 1] : zeroRow.data();
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
        for (uint32_t y = 0; y < pass.h; y++) {
            const uint8_t *row = &raw[passStart + (rowBytes + 1) * y + 1];
            for (uint32_t x = 0; x < pass.w; x++) {
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
                const size_t outX = (size_t)x * pass.dx + pass.x0, outY = (size_t)y * pass.dy + pass.y0;
                uint8_t *out = &(*rgb)[3 * (outY * w + outX)];
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
        passStart += (rowBytes + 1) * pass.h;
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


// ---- JPEG -------------------------------------------------------------------------------------
// The reference's textured scenes (scenes/staircase2.json, scenes/veach-ajar.json) use .jpg files
// and read them with stb_image.  A JPEG's decoded bytes depend on the decoder's arithmetic, so this
// one follows stb_image's: coefficients wrap to 16 bits on dequantisation, the integer IDCT after
// libjpeg's jidctint (12-bit constants, two extra bits kept between the passes), chroma upsampling
// with the 3:1 triangle filter, fixed-point YCbCr -> RGB with the green chroma term truncated to 16
// bits.  Entropy decoding is the standard's (Annex F / G).  Baseline and progressive, 8-bit, one or
// three components.  Checked byte for byte against stb_image's output (tests/test_textures.py).

struct JpegHuffman {
    int count[17] = { 0 };
    uint8_t values[256] = { 0 };
    int minCode[18] = { 0 }, maxCode[18] = { 0 }, firstIndex[18] = { 0 };
    bool present = false;

    void build()
    {
        int code = 0, index = 0;
        for (int length = 1; length <= 16; length++) {
            firstIndex[length] = index;
            minCode[length] = code;
            code += count[length];
            index += count[length];
            maxCode[length] = count[length] ? code - 1 : -1;
            code <<= 1;
        }
        present = true;
    }
};

struct JpegBits {
    const uint8_t *at, *end;
    uint32_t hold = 0;
    int held = 0;
    int marker = 0;     // a marker met inside the entropy-coded data (bits after it read as zero)

    void fill()
    {
        while (held <= 24) {
            uint32_t byte = 0;
            if (!marker && at < end) {
                byte = *at++;
                if (byte == 0xFF) {
                    uint32_t next = at < end ? *at++ : 0xD9u;
                    while (next == 0xFF && at < end) { next = *at++; }   // fill bytes
                    if (next != 0) { marker = (int)next; byte = 0; }
                }
            }
            hold |= byte << (24 - held);
            held += 8;
        }
    }
    int bit()
    {
        if (held < 1) { fill(); }
        const int value = (int)(hold >> 31);
        hold <<= 1;
        held--;
        return value;
    }
    int bits(int n)
    {
        int value = 0;
        for (int k = 0; k < n; k++) { value = (value << 1) | bit(); }
        return value;
    }
    void reset() { hold = 0; held = 0; marker = 0; }
};

// -1: no such code
int jpegDecodeSymbol(JpegBits &bits, const JpegHuffman &table)
{
    int code = 0;
    for (int length = 1; length <= 16; length++) {
        code = (code << 1) | bits.bit();
        if (table.maxCode[length] >= 0 && code <= table.maxCode[length] && code >= table.minCode[length]) {
            return table.values[table.firstIndex[length] + code - table.minCode[length]];
        }
    }
    return -1;
}

// the standard's RECEIVE + EXTEND
int jpegReceiveExtend(JpegBits &bits, int n)
{
    if (n == 0) { return 0; }
    const int value = bits.bits(n);
    return value < (1 << (n - 1)) ? value - (1 << n) + 1 : value;
}

const uint8_t kJpegZigzag[64] = {
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
};

#define JPEG_FIX(x) ((int)(((x) * 4096 + 0.5)))

// one 8-point pass of the jidctint-derived IDCT on inputs s0..s7; outputs the even part x0..x3 and
// the odd part t0..t3, both scaled by 4096
inline void jpegIdct1d(int s0, int s1, int s2, int s3, int s4, int s5, int s6, int s7, int *x, int *t)
{
    int p2 = s2, p3 = s6;
    int p1 = (p2 + p3) * JPEG_FIX(0.5411961f);
    const int e2 = p1 + p3 * JPEG_FIX(-1.847759065f);
    const int e3 = p1 + p2 * JPEG_FIX(0.765366865f);
    p2 = s0;
    p3 = s4;
    const int e0 = (p2 + p3) * 4096;
    const int e1 = (p2 - p3) * 4096;
    x[0] = e0 + e3;
    x[3] = e0 - e3;
    x[1] = e1 + e2;
    x[2] = e1 - e2;
    int t0 = s7, t1 = s5, t2 = s3, t3 = s1;
    p3 = t0 + t2;
    int p4 = t1 + t3;
    p1 = t0 + t3;
    p2 = t1 + t2;
    const int p5 = (p3 + p4) * JPEG_FIX(1.175875602f);
    t0 = t0 * JPEG_FIX(0.298631336f);
    t1 = t1 * JPEG_FIX(2.053119869f);
    t2 = t2 * JPEG_FIX(3.072711026f);
    t3 = t3 * JPEG_FIX(1.501321110f);
    p1 = p5 + p1 * JPEG_FIX(-0.899976223f);
    p2 = p5 + p2 * JPEG_FIX(-2.562915447f);
    p3 = p3 * JPEG_FIX(-1.961570560f);
    p4 = p4 * JPEG_FIX(-0.390180644f);
    t[3] = t3 + p1 + p4;
    t[2] = t2 + p2 + p3;
    t[1] = t1 + p2 + p4;
    t[0] = t0 + p1 + p3;
}

inline uint8_t jpegClamp(int value) { return (uint8_t)(value < 0 ? 0 : value > 255 ? 255 : value); }

void jpegIdctBlock(uint8_t *out, int stride, const int16_t *data)
{
    int columns[64];
    for (int i = 0; i < 8; i++) {
        const int16_t *d = data + i;
        int *v = columns + i;
        if (d[8] == 0 && d[16] == 0 && d[24] == 0 && d[32] == 0 && d[40] == 0 && d[48] == 0 && d[56] == 0) {
            const int flat = d[0] * 4;
            for (int k = 0; k < 8; k++) { v[8 * k] = flat; }
        } else {
            int x[4], t[4];
            jpegIdct1d(d[0], d[8], d[16], d[24], d[32], d[40], d[48], d[56], x, t);
            for (int k = 0; k < 4; k++) { x[k] += 512; }   // 12 bits of scale down to 2
            v[0] = (x[0] + t[3]) >> 10;  v[56] = (x[0] - t[3]) >> 10;
            v[8] = (x[1] + t[2]) >> 10;  v[48] = (x[1] - t[2]) >> 10;
            v[16] = (x[2] + t[1]) >> 10; v[40] = (x[2] - t[1]) >> 10;
            v[24] = (x[3] + t[0]) >> 10; v[32] = (x[3] - t[0]) >> 10;
        }
    }
    for (int i = 0; i < 8; i++) {
        const int *v = columns + 8 * i;
        uint8_t *o = out + (size_t)stride * i;
        int x[4], t[4];
        jpegIdct1d(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], x, t);
        for (int k = 0; k < 4; k++) { x[k] += 65536 + (128 << 17); }   // rounding of the 17-bit scale, level shift
        o[0] = jpegClamp((x[0] + t[3]) >> 17); o[7] = jpegClamp((x[0] - t[3]) >> 17);
        o[1] = jpegClamp((x[1] + t[2]) >> 17); o[6] = jpegClamp((x[1] - t[2]) >> 17);
        o[2] = jpegClamp((x[2] + t[1]) >> 17); o[5] = jpegClamp((x[2] - t[1]) >> 17);
        o[3] = jpegClamp((x[3] + t[0]) >> 17); o[4] = jpegClamp((x[3] - t[0]) >> 17);
    }
}

struct JpegComponent {
    int id = 0, h = 1, v = 1, tq = 0, hd = 0, ha = 0, dcPred = 0;
    int x = 0, y = 0, w2 = 0, h2 = 0;
    std::vector<uint8_t> data;       // w2 x h2 samples
    std::vector<int16_t> coeff;      // progressive: (w2 / 8) x (h2 / 8) blocks of 64
};

struct JpegState {
    int width = 0, height = 0, componentCount = 0;
    bool progressive = false, jfif = false;
    int adobeTransform = -1, rgbIds = 0;
    int hMax = 1, vMax = 1, mcuX = 0, mcuY = 0;
    int restartInterval = 0;
    uint16_t dequant[4][64] = { { 0 } };
    JpegHuffman dc[4], ac[4];
    JpegComponent component[4];
    // scan
    int scanCount = 0, order[4] = { 0 }, specStart = 0, specEnd = 63, succHigh = 0, succLow = 0, eobRun = 0;
};

bool jpegBaselineBlock(JpegState &z, JpegBits &bits, JpegComponent &c, int16_t *data, std::string *why)
{
    const JpegHuffman &dcTable = z.dc[c.hd], &acTable = z.ac[c.ha];
    const uint16_t *dequant = z.dequant[c.tq];
    std::memset(data, 0, 64 * sizeof(int16_t));
    const int t = jpegDecodeSymbol(bits, dcTable);
    if (t < 0 || t > 15) { *why = "bad huffman code"; return false; }
    c.dcPred += jpegReceiveExtend(bits, t);
    data[0] = (int16_t)(c.dcPred * dequant[0]);
    for (int k = 1; k < 64;) {
        const int rs = jpegDecodeSymbol(bits, acTable);
        if (rs < 0) { *why = "bad huffman code"; return false; }
        const int size = rs & 15, run = rs >> 4;
        if (size == 0) {
            if (rs != 0xF0) { break; }
            k += 16;
        } else {
            k += run;
            if (k > 63) { *why = "coefficient index out of range"; return false; }
            const int zig = kJpegZigzag[k++];
            data[zig] = (int16_t)(jpegReceiveExtend(bits, size) * dequant[zig]);
        }
    }
    return true;
}

bool jpegProgressiveDc(JpegState &z, JpegBits &bits, JpegComponent &c, int16_t *data, std::string *why)
{
    if (z.specEnd != 0) { *why = "progressive scan mixes dc and ac"; return false; }
    if (z.succHigh == 0) {
        std::memset(data, 0, 64 * sizeof(int16_t));
        const int t = jpegDecodeSymbol(bits, z.dc[c.hd]);
        if (t < 0 || t > 15) { *why = "bad huffman code"; return false; }
        c.dcPred += jpegReceiveExtend(bits, t);
        data[0] = (int16_t)(c.dcPred << z.succLow);
    } else if (bits.bit()) {
        data[0] = (int16_t)(data[0] + (int16_t)(1 << z.succLow));
    }
    return true;
}

bool jpegProgressiveAc(JpegState &z, JpegBits &bits, JpegComponent &c, int16_t *data, std::string *why)
{
    if (z.specStart == 0) { *why = "progressive scan mixes dc and ac"; return false; }
    const JpegHuffman &table = z.ac[c.ha];
    if (z.succHigh == 0) {
        if (z.eobRun) { z.eobRun--; return true; }
        for (int k = z.specStart; k <= z.specEnd;) {
            const int rs = jpegDecodeSymbol(bits, table);
            if (rs < 0) { *why = "bad huffman code"; return false; }
            const int size = rs & 15, run = rs >> 4;
            if (size == 0) {
                if (run < 15) {
                    z.eobRun = (1 << run);
                    if (run) { z.eobRun += bits.bits(run); }
                    z.eobRun--;
                    break;
                }
                k += 16;
            } else {
                k += run;
                if (k > 63) { *why = "coefficient index out of range"; return false; }
                data[kJpegZigzag[k++]] = (int16_t)(jpegReceiveExtend(bits, size) << z.succLow);
            }
        }
        return true;
    }
    // refinement of coefficients that already have their leading bits
    const int16_t bit = (int16_t)(1 << z.succLow);
    auto refine = [&](int16_t *p) {
        if (bits.bit() && (*p & bit) == 0) { *p = (int16_t)(*p > 0 ? *p + bit : *p - bit); }
    };
    if (z.eobRun) {
        z.eobRun--;
        for (int k = z.specStart; k <= z.specEnd; k++) {
            int16_t *p = &data[kJpegZigzag[k]];
            if (*p != 0) { refine(p); }
        }
        return true;
    }
    int k = z.specStart;
    do {
        const int rs = jpegDecodeSymbol(bits, table);
        if (rs < 0) { *why = "bad huffman code"; return false; }
        int size = rs & 15, run = rs >> 4;
        if (size == 0) {
            if (run < 15) {
                z.eobRun = (1 << run) - 1;
                if (run) { z.eobRun += bits.bits(run); }
                run = 64;   // nothing new in this block: refine to its end
            }
        } else {
            if (size != 1) { *why = "bad huffman code"; return false; }
            size = bits.bit() ? bit : -bit;
        }
        while (k <= z.specEnd) {
            int16_t *p = &data[kJpegZigzag[k++]];
            if (*p != 0) {
                refine(p);
            } else {
                if (run == 0) { *p = (int16_t)size; break; }
                run--;
            }
        }
    } while (k <= z.specEnd);
    return true;
}

bool jpegScan(JpegState &z, JpegBits &bits, std::string *why)
{
    auto restart = [&]() {
        bits.reset();
        for (int i = 0; i < 4; i++) { z.component[i].dcPred = 0; }
        z.eobRun = 0;
    };
    restart();
    int todo = z.restartInterval ? z.restartInterval : 0x7FFFFFFF;
    // true: go on; false: the scan ends here (a restart marker is missing: keep what was decoded)
    auto mcuDone = [&]() {
        if (--todo > 0) { return true; }
        if (bits.held < 24) { bits.fill(); }
        if (bits.marker < 0xD0 || bits.marker > 0xD7) { return false; }
        restart();
        todo = z.restartInterval ? z.restartInterval : 0x7FFFFFFF;
        return true;
    };
    int16_t block[64];
    if (z.scanCount == 1) {
        JpegComponent &c = z.component[z.order[0]];
        const int blocksWide = (c.x + 7) >> 3, blocksHigh = (c.y + 7) >> 3;
        for (int j = 0; j < blocksHigh; j++) {
            for (int i = 0; i < blocksWide; i++) {
                if (!z.progressive) {
                    if (!jpegBaselineBlock(z, bits, c, block, why)) { return false; }
                    jpegIdctBlock(c.data.data() + (size_t)c.w2 * j * 8 + i * 8, c.w2, block);
                } else {
                    int16_t *data = c.coeff.data() + 64 * ((size_t)i + (size_t)j * (c.w2 / 8));
                    if (z.specStart == 0 ? !jpegProgressiveDc(z, bits, c, data, why) : !jpegProgressiveAc(z, bits, c, data, why)) { return false; }
                }
                if (!mcuDone()) { return true; }
            }
        }
        return true;
    }
    for (int j = 0; j < z.mcuY; j++) {
        for (int i = 0; i < z.mcuX; i++) {
            for (int k = 0; k < z.scanCount; k++) {
                JpegComponent &c = z.component[z.order[k]];
                for (int y = 0; y < c.v; y++) {
                    for (int x = 0; x < c.h; x++) {
                        const int bx = i * c.h + x, by = j * c.v + y;
                        if (!z.progressive) {
                            if (!jpegBaselineBlock(z, bits, c, block, why)) { return false; }
                            jpegIdctBlock(c.data.data() + (size_t)c.w2 * by * 8 + bx * 8, c.w2, block);
                        } else {
                            int16_t *data = c.coeff.data() + 64 * ((size_t)bx + (size_t)by * (c.w2 / 8));
                            if (!jpegProgressiveDc(z, bits, c, data, why)) { return false; }
                        }
                    }
                }
            }
            if (!mcuDone()) { return true; }
        }
    }
    return true;
}

// chroma rows to full width, as the reference's decoder does it
const uint8_t *jpegResampleRow(uint8_t *out, const uint8_t *nearRow, const uint8_t *farRow, int w, int hs, int vs)
{
    if (hs == 1 && vs == 1) { return nearRow; }
    if (hs == 1 && vs == 2) {
        for (int i = 0; i < w; i++) { out[i] = (uint8_t)((3 * nearRow[i] + farRow[i] + 2) >> 2); }
        return out;
    }
    if (hs == 2 && vs == 1) {
        const uint8_t *in = nearRow;
        if (w == 1) { out[0] = out[1] = in[0]; return out; }
        out[0] = in[0];
        out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
        int i = 1;
        for (; i < w - 1; i++) {
            const int n = 3 * in[i] + 2;
            out[2 * i] = (uint8_t)((n + in[i - 1]) >> 2);
            out[2 * i + 1] = (uint8_t)((n + in[i + 1]) >> 2);
        }
        out[2 * i] = (uint8_t)((in[w - 2] * 3 + in[w - 1] + 2) >> 2);
        out[2 * i + 1] = in[w - 1];
        return out;
    }
    if (hs == 2 && vs == 2) {
        if (w == 1) { out[0] = out[1] = (uint8_t)((3 * nearRow[0] + farRow[0] + 2) >> 2); return out; }
        int t1 = 3 * nearRow[0] + farRow[0];
        out[0] = (uint8_t)((t1 + 2) >> 2);
        for (int i = 1; i < w; i++) {
            const int t0 = t1;
            t1 = 3 * nearRow[i] + farRow[i];
            out[2 * i - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
            out[2 * i] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
        }
        out[2 * w - 1] = (uint8_t)((t1 + 2) >> 2);
        return out;
    }
    for (int i = 0; i < w; i++) {   // other factors: nearest neighbour
        for (int j = 0; j < hs; j++) { out[i * hs + j] = nearRow[i]; }
    }
    return out;
}

bool decodeJpeg(const std::vector<uint8_t> &file, int *width, int *height, std::vector<uint8_t> *rgb, std::string *error)
{
    auto fail = [&](const std::string &message) { *error = "jpeg: " + message; return false; };
    JpegState z;
    size_t at = 2;
    bool sawFrame = false;
    auto be16 = [&](size_t p) { return (int)((file[p] << 8) | file[p + 1]); };
    JpegBits bits { nullptr, nullptr };
    int pendingMarker = 0;
    while (true) {
        int marker;
        if (pendingMarker) {
            marker = pendingMarker;
            pendingMarker = 0;
        } else {
            if (at + 2 > file.size()) { return fail("truncated file"); }
            if (file[at] != 0xFF) { return fail("expected a marker"); }
            while (at < file.size() && file[at] == 0xFF) { at++; }
            if (at >= file.size()) { return fail("truncated file"); }
            marker = file[at++];
        }
        if (marker == 0xD9) { break; }                     // EOI
        if (at + 2 > file.size()) { return fail("truncated segment"); }
        const int length = be16(at);
        if (length < 2 || at + (size_t)length > file.size()) { return fail("bad segment length"); }
        const size_t body = at + 2, bodyEnd = at + (size_t)length;
        if (marker == 0xC0 || marker == 0xC1 || marker == 0xC2) {
            if (sawFrame) { return fail("more than one frame"); }
            if (length < 11) { return fail("bad SOF length"); }
            if (file[body] != 8) { return fail("only 8-bit samples are supported"); }
            z.height = be16(body + 1);
            z.width = be16(body + 3);
            z.componentCount = file[body + 5];
            z.progressive = marker == 0xC2;
            if (z.width == 0 || z.height == 0) { return fail("zero image size"); }
            if (z.componentCount != 1 && z.componentCount != 3) { return fail("only greyscale and three-component files are supported"); }
            if (length != 8 + 3 * z.componentCount) { return fail("bad SOF length"); }
            static const uint8_t rgbIds[3] = { 'R', 'G', 'B' };
            for (int i = 0; i < z.componentCount; i++) {
                JpegComponent &c = z.component[i];
                c.id = file[body + 6 + 3 * i];
                if (z.componentCount == 3 && c.id == rgbIds[i]) { z.rgbIds++; }
                c.h = file[body + 7 + 3 * i] >> 4;
                c.v = file[body + 7 + 3 * i] & 15;
                c.tq = file[body + 8 + 3 * i];
                if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) { return fail("bad component description"); }
                z.hMax = c.h > z.hMax ? c.h : z.hMax;
                z.vMax = c.v > z.vMax ? c.v : z.vMax;
            }
            if ((uint64_t)z.width * z.height > (1ull << 28)) { return fail("image larger than 2^28 pixels"); }
            z.mcuX = (z.width + z.hMax * 8 - 1) / (z.hMax * 8);
            z.mcuY = (z.height + z.vMax * 8 - 1) / (z.vMax * 8);
            for (int i = 0; i < z.componentCount; i++) {
                JpegComponent &c = z.component[i];
                c.x = (z.width * c.h + z.hMax - 1) / z.hMax;
                c.y = (z.height * c.v + z.vMax - 1) / z.vMax;
                c.w2 = z.mcuX * c.h * 8;
                c.h2 = z.mcuY * c.v * 8;
                c.data.assign((size_t)c.w2 * c.h2, 0);
                if (z.progressive) { c.coeff.assign((size_t)c.w2 * c.h2, 0); }
            }
            sawFrame = true;
        } else if (marker == 0xDB) {
            size_t p = body;
            while (p < bodyEnd) {
                const int q = file[p++];
                const int wide = q >> 4, table = q & 15;
                if (wide > 1 || table > 3) { return fail("bad quantisation table"); }
                if (p + (size_t)(wide ? 128 : 64) > bodyEnd) { return fail("truncated quantisation table"); }
                for (int i = 0; i < 64; i++) {
                    z.dequant[table][kJpegZigzag[i]] = (uint16_t)(wide ? be16(p + 2 * (size_t)i) : file[p + (size_t)i]);
                }
                p += wide ? 128 : 64;
            }
        } else if (marker == 0xC4) {
            size_t p = body;
            while (p < bodyEnd) {
                if (p + 17 > bodyEnd) { return fail("truncated huffman table"); }
                const int q = file[p++];
                const int kind = q >> 4, index = q & 15;
                if (kind > 1 || index > 3) { return fail("bad huffman table header"); }
                JpegHuffman &table = kind == 0 ? z.dc[index] : z.ac[index];
                int total = 0;
                for (int i = 1; i <= 16; i++) { table.count[i] = file[p++]; total += table.count[i]; }
                if (total > 256 || p + (size_t)total > bodyEnd) { return fail("bad huffman table"); }
                for (int i = 0; i < total; i++) { table.values[i] = file[p++]; }
                table.build();
            }
        } else if (marker == 0xDD) {
            if (length != 4) { return fail("bad DRI length"); }
            z.restartInterval = be16(body);
        } else if (marker == 0xDA) {
            if (!sawFrame) { return fail("scan before frame header"); }
            z.scanCount = file[body];
            if (z.scanCount < 1 || z.scanCount > z.componentCount || length != 6 + 2 * z.scanCount) { return fail("bad scan header"); }
            for (int i = 0; i < z.scanCount; i++) {
                const int id = file[body + 1 + 2 * i], tables = file[body + 2 + 2 * i];
                int which = 0;
                while (which < z.componentCount && z.component[which].id != id) { which++; }
                if (which == z.componentCount) { return fail("scan names an unknown component"); }
                z.component[which].hd = tables >> 4;
                z.component[which].ha = tables & 15;
                if (z.component[which].hd > 3 || z.component[which].ha > 3) { return fail("bad huffman table index"); }
                z.order[i] = which;
            }
            z.specStart = file[body + 1 + 2 * z.scanCount];
            z.specEnd = file[body + 2 + 2 * z.scanCount];
            z.succHigh = file[body + 3 + 2 * z.scanCount] >> 4;
            z.succLow = file[body + 3 + 2 * z.scanCount] & 15;
            if (z.progressive) {
                if (z.specStart > 63 || z.specEnd > 63 || z.specStart > z.specEnd || z.succHigh > 13 || z.succLow > 13) { return fail("bad progressive scan"); }
                if (z.scanCount > 1 && z.specStart != 0) { return fail("interleaved ac scan"); }
            } else {
                if (z.specStart != 0 || z.succHigh != 0 || z.succLow != 0) { return fail("bad baseline scan"); }
                z.specEnd = 63;
            }
            bits = JpegBits { file.data() + bodyEnd, file.data() + file.size() };
            std::string why;
            if (!jpegScan(z, bits, &why)) { return fail(why); }
            // continue after the entropy-coded data: at the marker the bit reader met, or the next one in the file
            if (bits.marker == 0) {
                const uint8_t *p = bits.at;
                while (p + 1 < bits.end && !(p[0] == 0xFF && p[1] != 0 && p[1] != 0xFF)) { p++; }
                if (p + 1 >= bits.end) { return fail("no end-of-image marker"); }
                bits.marker = p[1];
                bits.at = p + 2;
            }
            at = (size_t)(bits.at - file.data());
            pendingMarker = bits.marker;
            // restart markers left over when a scan ended early are skipped
            if (pendingMarker >= 0xD0 && pendingMarker <= 0xD7) { pendingMarker = 0; }
            continue;
        } else if ((marker >= 0xE0 && marker <= 0xEF) || marker == 0xFE) {
            if (marker == 0xE0 && length - 2 >= 5 && std::memcmp(&file[body], "JFIF\0", 5) == 0) { z.jfif = true; }
            if (marker == 0xEE && length - 2 >= 12 && std::memcmp(&file[body], "Adobe\0", 6) == 0) { z.adobeTransform = file[body + 11]; }
        } else if (marker == 0xDC) {
            // DNL: the height was in the frame header already
        } else {
            return fail("unsupported marker");
        }
        at = bodyEnd;
    }
    if (!sawFrame) { return fail("no frame"); }

    if (z.progressive) {
        for (int n = 0; n < z.componentCount; n++) {
            JpegComponent &c = z.component[n];
            const int blocksWide = (c.x + 7) >> 3, blocksHigh = (c.y + 7) >> 3;
            for (int j = 0; j < blocksHigh; j++) {
                for (int i = 0; i < blocksWide; i++) {
                    int16_t *data = c.coeff.data() + 64 * ((size_t)i + (size_t)j * (c.w2 / 8));
                    for (int k = 0; k < 64; k++) { data[k] = (int16_t)(data[k] * z.dequant[c.tq][k]); }
                    jpegIdctBlock(c.data.data() + (size_t)c.w2 * j * 8 + i * 8, c.w2, data);
                }
            }
        }
    }

    // to RGB: upsample the components row by row, then convert
    const bool isRgb = z.componentCount == 3 && (z.rgbIds == 3 || (z.adobeTransform == 0 && !z.jfif));
    rgb->assign((size_t)3 * z.width * z.height, 0);
    struct Resample { int hs, vs, ystep, lowWidth, ypos; const uint8_t *line0, *line1; std::vector<uint8_t> buffer; };
    Resample state[3];
    for (int k = 0; k < z.componentCount; k++) {
        Resample &r = state[k];
        r.hs = z.hMax / z.component[k].h;
        r.vs = z.vMax / z.component[k].v;
        r.ystep = r.vs >> 1;
        r.lowWidth = (z.width + r.hs - 1) / r.hs;
        r.ypos = 0;
        r.line0 = r.line1 = z.component[k].data.data();
        r.buffer.assign((size_t)z.width + 2 * r.hs + 8, 0);
    }
    const int fixR = ((int)(1.40200f * 4096.0f + 0.5f)) << 8, fixG1 = ((int)(0.71414f * 4096.0f + 0.5f)) << 8;
    const int fixG2 = ((int)(0.34414f * 4096.0f + 0.5f)) << 8, fixB = ((int)(1.77200f * 4096.0f + 0.5f)) << 8;
    for (int j = 0; j < z.height; j++) {
        const uint8_t *rows[3] = { nullptr, nullptr, nullptr };
        for (int k = 0; k < z.componentCount; k++) {
            Resample &r = state[k];
            const bool bottom = r.ystep >= (r.vs >> 1);
            rows[k] = jpegResampleRow(r.buffer.data(), bottom ? r.line1 : r.line0, bottom ? r.line0 : r.line1, r.lowWidth, r.hs, r.vs);
            if (++r.ystep >= r.vs) {
                r.ystep = 0;
                r.line0 = r.line1;
                if (++r.ypos < z.component[k].y) { r.line1 += z.component[k].w2; }
            }
        }
        uint8_t *out = rgb->data() + (size_t)3 * z.width * j;
        for (int i = 0; i < z.width; i++, out += 3) {
            if (z.componentCount == 1) {
                out[0] = out[1] = out[2] = rows[0][i];
            } else if (isRgb) {
                out[0] = rows[0][i]; out[1] = rows[1][i]; out[2] = rows[2][i];
            } else {
                const int yFixed = (rows[0][i] << 20) + (1 << 19);
                const int cr = rows[2][i] - 128, cb = rows[1][i] - 128;
                int r = yFixed + cr * fixR;
                int g = yFixed + cr * -fixG1 + (int)((unsigned int)(cb * -fixG2) & 0xFFFF0000u);
                int b = yFixed + cb * fixB;
                r >>= 20; g >>= 20; b >>= 20;
                out[0] = jpegClamp(r); out[1] = jpegClamp(g); out[2] = jpegClamp(b);
            }
        }
    }
    *width = z.width;
    *height = z.height;
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
    } else if (file.size() >= 4 && file[0] == 0xFF && file[1] == 0xD8) {
        ok = decodeJpeg(file, width, height, rgb, error);
    } else {
        *error = "unknown image format (PNG, JPEG and binary PNM are supported)";
        ok = false;
    }
    if (!ok) { *error = path + ": " + *error; }
    return ok;
}

}  // namespace pathed
