#include "exr.h"

#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>

namespace pathed {

unsigned short floatToHalf(float value)
{
    uint32_t bits;
    std::memcpy(&bits, &value, 4);
    const uint32_t sign = (bits >> 16) & 0x8000u;
    const int32_t exponent = (int32_t)((bits >> 23) & 0xFF) - 127 + 15;
    uint32_t mantissa = bits & 0x7FFFFFu;

    if (((bits >> 23) & 0xFF) == 0xFF) {  // inf / nan
        return (unsigned short)(sign | 0x7C00u | (mantissa ? 0x200u : 0u));
    }
    if (exponent >= 31) { return (unsigned short)(sign | 0x7C00u); }  // overflow -> inf
    if (exponent <= 0) {
        if (exponent < -10) { return (unsigned short)sign; }  // underflow -> 0
        mantissa |= 0x800000u;
        const int shift = 14 - exponent;
        uint32_t half = mantissa >> shift;
        const uint32_t remainder = mantissa & ((1u << shift) - 1u);
        const uint32_t halfway = 1u << (shift - 1);
        if (remainder > halfway || (remainder == halfway && (half & 1u))) { half++; }
        return (unsigned short)(sign | half);
    }
    uint32_t half = ((uint32_t)exponent << 10) | (mantissa >> 13);
    const uint32_t remainder = mantissa & 0x1FFFu;
    if (remainder > 0x1000u || (remainder == 0x1000u && (half & 1u))) { half++; }
    return (unsigned short)(sign | half);
}

float halfToFloat(unsigned short half)
{
    const uint32_t sign = ((uint32_t)half & 0x8000u) << 16;
    uint32_t exponent = (half >> 10) & 0x1Fu;
    uint32_t mantissa = half & 0x3FFu;
    uint32_t bits;
    if (exponent == 0) {
        if (mantissa == 0) {
            bits = sign;
        } else {
            exponent = 127 - 15 + 1;
            while (!(mantissa & 0x400u)) { mantissa <<= 1; exponent--; }
            mantissa &= 0x3FFu;
            bits = sign | (exponent << 23) | (mantissa << 13);
        }
    } else if (exponent == 31) {
        bits = sign | 0x7F800000u | (mantissa << 13);
    } else {
        bits = sign | ((exponent + 127 - 15) << 23) | (mantissa << 13);
    }
    float value;
    std::memcpy(&value, &bits, 4);
    return value;
}

namespace {

void putBytes(std::vector<unsigned char> &out, const void *data, size_t size)
{
    const unsigned char *bytes = (const unsigned char *)data;
    out.insert(out.end(), bytes, bytes + size);
}

void putString(std::vector<unsigned char> &out, const char *text)
{
    putBytes(out, text, std::strlen(text) + 1);
}

void putInt(std::vector<unsigned char> &out, int32_t value) { putBytes(out, &value, 4); }
void putFloat(std::vector<unsigned char> &out, float value) { putBytes(out, &value, 4); }

void putAttribute(
    std::vector<unsigned char> &out,
    const char *name, const char *type,
    const std::vector<unsigned char> &value
) {
    putString(out, name);
    putString(out, type);
    putInt(out, (int32_t)value.size());
    putBytes(out, value.data(), value.size());
}

struct ChannelSpec {
    const char *name;
    int pixelType;  // 1 = HALF, 2 = FLOAT
    const float *plane;
    int stride;     // floats between consecutive pixels of the plane
};

bool writeScanlineExr(
    const std::string &path,
    int width, int height,
    const std::vector<ChannelSpec> &channels,  // must be in alphabetical order
    std::string *error
) {
    std::vector<unsigned char> out;
    const unsigned char magic[4] = { 0x76, 0x2f, 0x31, 0x01 };
    putBytes(out, magic, 4);
    const unsigned char version[4] = { 2, 0, 0, 0 };
    putBytes(out, version, 4);

    {
        std::vector<unsigned char> chlist;
        for (const ChannelSpec &channel : channels) {
            putString(chlist, channel.name);
            putInt(chlist, channel.pixelType);
            const unsigned char linearAndReserved[4] = { 0, 0, 0, 0 };
            putBytes(chlist, linearAndReserved, 4);
            putInt(chlist, 1);
            putInt(chlist, 1);
        }
        chlist.push_back(0);
        putAttribute(out, "channels", "chlist", chlist);
    }
    putAttribute(out, "compression", "compression", { 0 });
    {
        std::vector<unsigned char> box;
        putInt(box, 0); putInt(box, 0); putInt(box, width - 1); putInt(box, height - 1);
        putAttribute(out, "dataWindow", "box2i", box);
        putAttribute(out, "displayWindow", "box2i", box);
    }
    putAttribute(out, "lineOrder", "lineOrder", { 0 });
    {
        std::vector<unsigned char> one;
        putFloat(one, 1.f);
        putAttribute(out, "pixelAspectRatio", "float", one);
        std::vector<unsigned char> center;
        putFloat(center, 0.f); putFloat(center, 0.f);
        putAttribute(out, "screenWindowCenter", "v2f", center);
        putAttribute(out, "screenWindowWidth", "float", one);
    }
    out.push_back(0);

    size_t bytesPerLine = 0;
    for (const ChannelSpec &channel : channels) {
        bytesPerLine += (size_t)width * (channel.pixelType == 1 ? 2 : 4);
    }

    const size_t tableStart = out.size();
    const size_t dataStart = tableStart + (size_t)height * 8;
    for (int y = 0; y < height; y++) {
        uint64_t offset = dataStart + (uint64_t)y * (8 + bytesPerLine);
        putBytes(out, &offset, 8);
    }

    out.reserve(dataStart + (size_t)height * (8 + bytesPerLine));
    std::vector<unsigned short> halfLine((size_t)width);
    std::vector<float> floatLine((size_t)width);
    for (int y = 0; y < height; y++) {
        putInt(out, y);
        putInt(out, (int32_t)bytesPerLine);
        for (const ChannelSpec &channel : channels) {
            const float *row = channel.plane + (size_t)y * width * channel.stride;
            if (channel.pixelType == 1) {
                for (int x = 0; x < width; x++) { halfLine[x] = floatToHalf(row[(size_t)x * channel.stride]); }
                putBytes(out, halfLine.data(), (size_t)width * 2);
            } else {
                for (int x = 0; x < width; x++) { floatLine[x] = row[(size_t)x * channel.stride]; }
                putBytes(out, floatLine.data(), (size_t)width * 4);
            }
        }
    }

    std::ofstream file(path, std::ios::binary);
    if (!file) {
        if (error) { *error = "cannot open " + path + " for writing"; }
        return false;
    }
    file.write((const char *)out.data(), (std::streamsize)out.size());
    return (bool)file;
}

}  // namespace

bool writeExrHalfBGR(
    const std::string &path,
    int width, int height,
    const float *r, const float *g, const float *b,
    std::string *error
) {
    std::vector<ChannelSpec> channels = {
        { "B", 1, b, 1 },
        { "G", 1, g, 1 },
        { "R", 1, r, 1 },
    };
    return writeScanlineExr(path, width, height, channels, error);
}

bool writeExrFloatRGBA(
    const std::string &path,
    int width, int height,
    const float *rgba,
    std::string *error
) {
    std::vector<ChannelSpec> channels = {
        { "A", 2, rgba + 3, 4 },
        { "B", 2, rgba + 2, 4 },
        { "G", 2, rgba + 1, 4 },
        { "R", 2, rgba + 0, 4 },
    };
    return writeScanlineExr(path, width, height, channels, error);
}

namespace {

struct Reader {
    const std::vector<unsigned char> &data;
    size_t pos = 0;
    bool ok = true;

    explicit Reader(const std::vector<unsigned char> &d) : data(d) {}

    bool take(void *out, size_t size)
    {
        if (pos + size > data.size()) { ok = false; return false; }
        std::memcpy(out, data.data() + pos, size);
        pos += size;
        return true;
    }

    std::string takeString()
    {
        std::string s;
        while (pos < data.size() && data[pos] != 0) { s += (char)data[pos++]; }
        if (pos >= data.size()) { ok = false; return s; }
        pos++;
        return s;
    }
};

struct ChannelInfo {
    std::string name;
    int pixelType;
};

// ---- PIZ (OpenEXR compression 4): range compaction by bitmap, 2-D Haar-like wavelet on 16-bit
// words, canonical Huffman with a run-length symbol.  Needed for environment maps written by tools
// whose default is PIZ; the reference reads them through tinyexr (src/environment_light.cpp:14-28).
// Restated from the published format; checked against a file written by the reference's tinyexr
// (tests/golden/env_piz.exr).

struct PizBits {
    const unsigned char *at, *end;
    uint64_t hold = 0;
    int held = 0;

    // n <= 32; reads past the end yield zero bits
    uint32_t take(int n)
    {
        while (held < n) {
            hold = (hold << 8) | (at < end ? *at++ : 0u);
            held += 8;
        }
        held -= n;
        return (uint32_t)((hold >> held) & ((1ull << n) - 1ull));
    }
};

bool pizHuffmanDecode(const unsigned char *in, size_t inLength, std::vector<uint16_t> *out, std::string *why)
{
    const int kSymbols = (1 << 16) + 1;   // 16-bit literals plus the run-length symbol
    if (inLength < 20) { *why = "piz: truncated huffman header"; return false; }
    uint32_t header[5];
    std::memcpy(header, in, 20);
    const uint32_t first = header[0], last = header[1], bitCount = header[3];
    if (first >= (uint32_t)kSymbols || last >= (uint32_t)kSymbols || first > last) { *why = "piz: bad symbol range"; return false; }

    // code lengths: 6 bits each, 59..62 = a run of 2..5 zero lengths, 63 = 8 more bits, run of 6..261
    std::vector<unsigned char> length((size_t)kSymbols, 0);
    PizBits table { in + 20, in + inLength };
    for (uint32_t symbol = first; symbol <= last; symbol++) {
        const uint32_t l = table.take(6);
        if (l == 63u) {
            const uint32_t run = table.take(8) + 6u;
            if (symbol + run > last + 1u) { *why = "piz: bad zero run"; return false; }
            symbol += run - 1u;
        } else if (l >= 59u) {
            const uint32_t run = l - 59u + 2u;
            if (symbol + run > last + 1u) { *why = "piz: bad zero run"; return false; }
            symbol += run - 1u;
        } else {
            length[symbol] = (unsigned char)l;
        }
    }
    const unsigned char *stream = table.at;   // the bit stream starts on the next byte
    if (stream > in + inLength || (uint64_t)bitCount > 8ull * (uint64_t)(in + inLength - stream)) { *why = "piz: truncated huffman data"; return false; }

    // canonical codes: within a length symbols take consecutive codes in symbol order; the longest
    // codes are numerically lowest (base[l] = (base[l + 1] + count[l + 1]) >> 1)
    uint64_t count[59] = { 0 }, base[60] = { 0 };
    for (int symbol = 0; symbol < kSymbols; symbol++) { count[length[(size_t)symbol]]++; }
    {
        uint64_t code = 0;
        for (int l = 58; l > 0; l--) {
            base[l] = code;
            code = (code + count[l]) >> 1;
        }
    }
    // a consistent length table never assigns a code that does not fit its length
    for (int l = 1; l <= 58; l++) {
        if (count[l] && base[l] + count[l] > (1ull << l)) { *why = "piz: inconsistent huffman code lengths"; return false; }
    }
    std::vector<uint32_t> start(60, 0), ordered;   // symbols sorted by (length, symbol)
    {
        uint32_t total = 0;
        for (int l = 1; l <= 58; l++) { start[(size_t)l] = total; total += (uint32_t)count[l]; }
        ordered.resize(total);
        std::vector<uint32_t> cursor(start);
        for (int symbol = 0; symbol < kSymbols; symbol++) {
            const int l = length[(size_t)symbol];
            if (l > 0) { ordered[cursor[(size_t)l]++] = (uint32_t)symbol; }
        }
    }
    // short codes resolve with one table access
    const int kFast = 12;
    std::vector<uint32_t> fast((size_t)1 << kFast, 0);   // symbol << 6 | length, 0 = longer code
    for (int l = 1; l <= kFast; l++) {
        for (uint64_t k = 0; k < count[l]; k++) {
            const uint64_t code = base[l] + k;
            const uint32_t entry = (ordered[start[(size_t)l] + (uint32_t)k] << 6) | (uint32_t)l;
            const uint64_t low = code << (kFast - l);
            for (uint64_t fill = 0; fill < (1ull << (kFast - l)); fill++) { fast[(size_t)(low + fill)] = entry; }
        }
    }

    const uint32_t runSymbol = last;
    PizBits bits { stream, in + inLength };
    uint64_t remaining = bitCount;
    size_t produced = 0;
    const size_t wanted = out->size();
    while (remaining > 0) {
        uint32_t symbol = 0;
        int used = 0;
        // peek up to kFast bits without consuming more than remain
        const int peek = remaining < (uint64_t)kFast ? (int)remaining : kFast;
        while (bits.held < peek) {
            bits.hold = (bits.hold << 8) | (bits.at < bits.end ? *bits.at++ : 0u);
            bits.held += 8;
        }
        const uint32_t window = (uint32_t)((bits.hold >> (bits.held - peek)) & ((1ull << peek) - 1ull)) << (kFast - peek);
        const uint32_t entry = fast[window];
        if (entry != 0 && (int)(entry & 63u) <= peek) {
            used = (int)(entry & 63u);
            symbol = entry >> 6;
            bits.held -= used;
        } else {
            // longer than the table: extend bit by bit
            uint64_t code = 0;
            bool found = false;
            while ((uint64_t)used < remaining && used < 58) {
                code = (code << 1) | bits.take(1);
                used++;
                if (count[used] && code >= base[used] && code - base[used] < count[used]) {
                    symbol = ordered[start[(size_t)used] + (uint32_t)(code - base[used])];
                    found = true;
                    break;
                }
            }
            if (!found) { *why = "piz: invalid huffman code"; return false; }
        }
        remaining -= (uint64_t)used;
        if (symbol == runSymbol) {
            if (remaining < 8 || produced == 0) { *why = "piz: bad run"; return false; }
            const uint32_t run = bits.take(8);
            remaining -= 8;
            if (produced + run > wanted) { *why = "piz: run past the end"; return false; }
            const uint16_t value = (*out)[produced - 1];
            for (uint32_t k = 0; k < run; k++) { (*out)[produced++] = value; }
        } else {
            if (produced >= wanted) { *why = "piz: too much data"; return false; }
            (*out)[produced++] = (uint16_t)symbol;
        }
    }
    if (produced != wanted) { *why = "piz: not enough data"; return false; }
    return true;
}

// inverse of the two wavelet butterflies (14-bit exact variant, 16-bit modulo variant)
inline void pizUnbutterfly(bool exact14, uint16_t low, uint16_t high, uint16_t *a, uint16_t *b)
{
    if (exact14) {
        const int l = (int16_t)low, h = (int16_t)high;
        const int first = l + (h & 1) + (h >> 1);
        *a = (uint16_t)(int16_t)first;
        *b = (uint16_t)(int16_t)(first - h);
    } else {
        const int m = low, d = high;
        const int second = (m - (d >> 1)) & 0xFFFF;
        *a = (uint16_t)((d + second - 0x8000) & 0xFFFF);
        *b = (uint16_t)second;
    }
}

// in-place inverse 2-D wavelet over an nx x ny grid with element strides sx, sy
void pizWaveletDecode(uint16_t *data, int nx, int sx, int ny, int sy, uint16_t maxValue)
{
    const bool exact14 = maxValue < (1 << 14);
    const int n = nx < ny ? nx : ny;
    int step = 1;
    while (step <= n) { step <<= 1; }
    step >>= 1;           // coarsest level: distance between the pairs' first elements is 2 * half
    int half = step >> 1;
    for (; half >= 1; step = half, half >>= 1) {
        const int rowLimit = ny - step, columnLimit = nx - step;
        int y = 0;
        for (; y <= rowLimit; y += step) {
            uint16_t *row = data + (size_t)y * sy;
            int x = 0;
            for (; x <= columnLimit; x += step) {
                uint16_t *p00 = row + (size_t)x * sx;
                uint16_t *p01 = p00 + (size_t)half * sx;
                uint16_t *p10 = p00 + (size_t)half * sy;
                uint16_t *p11 = p10 + (size_t)half * sx;
                uint16_t a, b, c, d;
                pizUnbutterfly(exact14, *p00, *p10, &a, &c);
                pizUnbutterfly(exact14, *p01, *p11, &b, &d);
                pizUnbutterfly(exact14, a, b, p00, p01);
                pizUnbutterfly(exact14, c, d, p10, p11);
            }
            if (nx & half) {   // odd column: vertical pair only
                uint16_t *p00 = row + (size_t)x * sx;
                uint16_t *p10 = p00 + (size_t)half * sy;
                uint16_t a;
                pizUnbutterfly(exact14, *p00, *p10, &a, p10);
                *p00 = a;
            }
        }
        if (ny & half) {       // odd row: horizontal pairs only
            uint16_t *row = data + (size_t)y * sy;
            for (int x = 0; x <= columnLimit; x += step) {
                uint16_t *p00 = row + (size_t)x * sx;
                uint16_t *p01 = p00 + (size_t)half * sx;
                uint16_t a;
                pizUnbutterfly(exact14, *p00, *p01, &a, p01);
                *p00 = a;
            }
        }
    }
}

// one PIZ block -> the raw scanline layout (per line, per channel, little-endian samples)
bool pizDecodeBlock(const unsigned char *in, size_t inLength, const std::vector<ChannelInfo> &channels, int width, int lines,
                    std::vector<unsigned char> *raw, std::string *why)
{
    if (inLength < 4) { *why = "piz: truncated block"; return false; }
    uint16_t minNonZero, maxNonZero;
    std::memcpy(&minNonZero, in, 2);
    std::memcpy(&maxNonZero, in + 2, 2);
    size_t at = 4;
    std::vector<unsigned char> bitmap(8192, 0);
    if (maxNonZero >= 8192) { *why = "piz: bad bitmap range"; return false; }
    if (minNonZero <= maxNonZero) {
        const size_t bytes = (size_t)maxNonZero - minNonZero + 1;
        if (at + bytes > inLength) { *why = "piz: truncated bitmap"; return false; }
        std::memcpy(bitmap.data() + minNonZero, in + at, bytes);
        at += bytes;
    }
    // the values present (plus zero), in order: word k of the coded data stands for table[k]
    std::vector<uint16_t> table(65536, 0);
    int present = 0;
    for (int value = 0; value < 65536; value++) {
        if (value == 0 || (bitmap[(size_t)value >> 3] & (1 << (value & 7)))) { table[(size_t)present++] = (uint16_t)value; }
    }
    const uint16_t maxValue = (uint16_t)(present - 1);

    if (at + 4 > inLength) { *why = "piz: truncated block"; return false; }
    int32_t huffmanLength;
    std::memcpy(&huffmanLength, in + at, 4);
    at += 4;
    if (huffmanLength < 0 || at + (size_t)huffmanLength > inLength) { *why = "piz: bad huffman length"; return false; }

    size_t words = 0;
    for (const ChannelInfo &channel : channels) { words += (size_t)width * lines * (channel.pixelType == 1 ? 1 : 2); }
    std::vector<uint16_t> data(words);
    if (!pizHuffmanDecode(in + at, (size_t)huffmanLength, &data, why)) { return false; }

    // channel planes follow each other; a 32-bit sample is two interleaved 16-bit planes
    size_t plane = 0;
    std::vector<size_t> planeStart;
    for (const ChannelInfo &channel : channels) {
        const int wordsPerSample = channel.pixelType == 1 ? 1 : 2;
        planeStart.push_back(plane);
        for (int part = 0; part < wordsPerSample; part++) {
            pizWaveletDecode(data.data() + plane + part, width, wordsPerSample, lines, width * wordsPerSample, maxValue);
        }
        plane += (size_t)width * lines * wordsPerSample;
    }
    for (uint16_t &word : data) { word = table[word]; }

    raw->resize(words * 2);
    size_t cursor = 0;
    for (int line = 0; line < lines; line++) {
        for (size_t c = 0; c < channels.size(); c++) {
            const size_t rowWords = (size_t)width * (channels[c].pixelType == 1 ? 1 : 2);
            std::memcpy(raw->data() + cursor, data.data() + planeStart[c] + (size_t)line * rowWords, rowWords * 2);
            cursor += rowWords * 2;
        }
    }
    return true;
}

}  // namespace

bool readExrRGBA(
    const std::string &path,
    int *width, int *height,
    std::vector<float> *rgba,
    std::string *error
) {
    auto fail = [&](const std::string &what) {
        if (error) { *error = "exr " + path + ": " + what; }
        return false;
    };

    std::ifstream file(path, std::ios::binary);
    if (!file) { return fail("cannot open"); }
    std::vector<unsigned char> data((std::istreambuf_iterator<char>(file)), std::istreambuf_iterator<char>());

    Reader reader(data);
    unsigned char magic[4];
    unsigned char version[4];
    if (!reader.take(magic, 4) || !reader.take(version, 4)) { return fail("truncated"); }
    if (magic[0] != 0x76 || magic[1] != 0x2f || magic[2] != 0x31 || magic[3] != 0x01) { return fail("bad magic"); }
    if (version[1] & 0x02) { return fail("tiled files unsupported"); }
    if (version[1] & 0x18) { return fail("deep / multipart files unsupported"); }

    std::vector<ChannelInfo> channels;
    int compression = -1;
    int32_t window[4] = { 0, 0, -1, -1 };
    int lineOrder = 0;

    while (reader.ok) {
        std::string name = reader.takeString();
        if (name.empty()) { break; }
        std::string type = reader.takeString();
        int32_t size = 0;
        reader.take(&size, 4);
        if (!reader.ok || size < 0 || reader.pos + (size_t)size > data.size()) { return fail("bad attribute"); }
        const size_t valueStart = reader.pos;
        if (name == "channels") {
            Reader sub(data);
            sub.pos = valueStart;
            while (sub.pos < valueStart + (size_t)size) {
                std::string channelName = sub.takeString();
                if (channelName.empty()) { break; }
                int32_t pixelType = 0;
                unsigned char skip[4];
                int32_t xs = 0, ys = 0;
                sub.take(&pixelType, 4);
                sub.take(skip, 4);
                sub.take(&xs, 4);
                sub.take(&ys, 4);
                if (xs != 1 || ys != 1) { return fail("subsampled channels unsupported"); }
                channels.push_back({ channelName, pixelType });
            }
        } else if (name == "compression") {
            compression = data[valueStart];
        } else if (name == "dataWindow") {
            std::memcpy(window, data.data() + valueStart, 16);
        } else if (name == "lineOrder") {
            lineOrder = data[valueStart];
        }
        reader.pos = valueStart + (size_t)size;
    }
    if (!reader.ok) { return fail("truncated header"); }
    if (channels.empty()) { return fail("no channels"); }
    if (compression != 0 && compression != 2 && compression != 3 && compression != 4) {
        return fail("compression type " + std::to_string(compression) + " unsupported (NONE / ZIPS / ZIP / PIZ only)");
    }
    (void)lineOrder;

    const int w = window[2] - window[0] + 1;
    const int h = window[3] - window[1] + 1;
    if (w <= 0 || h <= 0) { return fail("bad data window"); }
    if ((uint64_t)w * (uint64_t)h > (1ull << 28)) { return fail("image larger than 2^28 pixels"); }

    size_t bytesPerLine = 0;
    for (const ChannelInfo &channel : channels) {
        if (channel.pixelType != 1 && channel.pixelType != 2) { return fail("UINT channels unsupported"); }
        bytesPerLine += (size_t)w * (channel.pixelType == 1 ? 2 : 4);
    }

    const int linesPerBlock = (compression == 4) ? 32 : (compression == 3) ? 16 : 1;
    const int blockCount = (h + linesPerBlock - 1) / linesPerBlock;

    std::vector<uint64_t> offsets((size_t)blockCount);
    if (!reader.take(offsets.data(), (size_t)blockCount * 8)) { return fail("truncated offset table"); }

    rgba->assign((size_t)4 * w * h, 0.f);
    for (size_t i = 0; i < (size_t)w * h; i++) { (*rgba)[4 * i + 3] = 1.f; }

    std::vector<unsigned char> raw;
    std::vector<unsigned char> scratch;
    for (int block = 0; block < blockCount; block++) {
        size_t pos = (size_t)offsets[(size_t)block];
        if (pos > data.size() || data.size() - pos < 8) { return fail("bad block offset"); }   // no wrap for offsets near 2^64
        int32_t y, size;
        std::memcpy(&y, data.data() + pos, 4);
        std::memcpy(&size, data.data() + pos + 4, 4);
        pos += 8;
        if (size < 0 || (size_t)size > data.size() - pos) { return fail("bad block size"); }

        const int firstLine = y - window[1];
        const int lines = std::min(linesPerBlock, h - firstLine);
        if (firstLine < 0 || lines <= 0) { return fail("bad block y"); }
        const size_t expected = bytesPerLine * (size_t)lines;

        if (compression == 0 || (size_t)size == expected) {
            raw.assign(data.begin() + (long)pos, data.begin() + (long)(pos + (size_t)size));
            if (raw.size() != expected) { return fail("bad uncompressed block"); }
        } else if (compression == 4) {
            std::string why;
            if (!pizDecodeBlock(data.data() + pos, (size_t)size, channels, w, lines, &raw, &why)) { return fail(why); }
            if (raw.size() != expected) { return fail("piz: block size mismatch"); }
        } else {
            scratch.resize(expected);
            uLongf destLength = (uLongf)expected;
            int code = uncompress(scratch.data(), &destLength, data.data() + pos, (uLong)size);
            if (code != Z_OK || destLength != expected) { return fail("zlib failure"); }
            // undo the predictor ...
            for (size_t i = 1; i < expected; i++) {
                scratch[i] = (unsigned char)(scratch[i - 1] + scratch[i] - 128);
            }
            // ... and the even/odd byte split
            raw.resize(expected);
            const size_t half = (expected + 1) / 2;
            size_t a = 0, b = half;
            for (size_t i = 0; i < expected; ) {
                raw[i++] = scratch[a++];
                if (i < expected) { raw[i++] = scratch[b++]; }
            }
        }

        size_t cursor = 0;
        for (int line = 0; line < lines; line++) {
            const int row = firstLine + line;
            for (const ChannelInfo &channel : channels) {
                int component = -1;
                if (channel.name == "R") { component = 0; }
                else if (channel.name == "G") { component = 1; }
                else if (channel.name == "B") { component = 2; }
                else if (channel.name == "A") { component = 3; }
                else if (channel.name == "Y" && channels.size() == 1) { component = 4; }
                for (int x = 0; x < w; x++) {
                    float value;
                    if (channel.pixelType == 1) {
                        unsigned short half;
                        std::memcpy(&half, raw.data() + cursor, 2);
                        cursor += 2;
                        value = halfToFloat(half);
                    } else {
                        std::memcpy(&value, raw.data() + cursor, 4);
                        cursor += 4;
                    }
                    float *pixel = rgba->data() + 4 * ((size_t)row * w + x);
                    if (component == 4) { pixel[0] = pixel[1] = pixel[2] = value; }
                    else if (component >= 0) { pixel[component] = value; }
                }
            }
        }
    }

    *width = w;
    *height = h;
    return true;
}

}  // namespace pathed
