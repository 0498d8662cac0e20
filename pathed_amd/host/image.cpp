#include "image.h"

#include "exr.h"

#include <cmath>
#include <cstdio>

namespace pathed {

Image::Image(int width, int height, const std::string &outputDirectory)
    : m_height(height),
      m_width(width),
      m_spp(0),
      m_outputDirectory(outputDirectory),
      m_data((size_t)3 * height * width),
      m_raw((size_t)3 * height * width)
{}

// src/image.cpp:21-35
void Image::set(int row, int col, float r, float g, float b)
{
    const size_t flipped = (size_t)3 * ((size_t)(m_height - row - 1) * m_width + col);
    m_raw[flipped + 0] = r;
    m_raw[flipped + 1] = g;
    m_raw[flipped + 2] = b;

    const size_t index = (size_t)3 * ((size_t)row * m_width + col);
    m_data[index + 0] = (unsigned char)(fminf(powf(r, 1 / 2.2), 1.f) * 255);
    m_data[index + 1] = (unsigned char)(fminf(powf(g, 1 / 2.2), 1.f) * 255);
    m_data[index + 2] = (unsigned char)(fminf(powf(b, 1 / 2.2), 1.f) * 255);
}

void Image::save(const std::string &filestem) { save(filestem, false); }
void Image::saveCheckpoint(const std::string &filestem) { save(filestem, true); }

// src/image.cpp:80-154: <outdir>/<stem>.exr and, for checkpoints, <outdir>/<stem>-%05dspp.exr
void Image::save(const std::string &filestem, bool saveCheckpoint)
{
    const size_t pixels = (size_t)m_width * m_height;
    std::vector<float> planes[3];
    for (int c = 0; c < 3; c++) { planes[c].resize(pixels); }
    for (size_t i = 0; i < pixels; i++) {
        planes[0][i] = m_raw[3 * i + 0];
        planes[1][i] = m_raw[3 * i + 1];
        planes[2][i] = m_raw[3 * i + 2];
    }

    const std::string outputExr = m_outputDirectory + filestem + ".exr";
    std::string error;
    if (!writeExrHalfBGR(outputExr, m_width, m_height, planes[0].data(), planes[1].data(), planes[2].data(), &error)) {
        fprintf(stderr, "Save EXR err: %s\n", error.c_str());
        return;
    }
    printf("Saved exr file. [ %s ] \n", outputExr.c_str());

    if (saveCheckpoint) {
        char suffix[64];
        snprintf(suffix, sizeof suffix, "-%05dspp.exr", m_spp);
        const std::string outputSppExr = m_outputDirectory + filestem + suffix;
        if (!writeExrHalfBGR(outputSppExr, m_width, m_height, planes[0].data(), planes[1].data(), planes[2].data(), &error)) {
            fprintf(stderr, "Save EXR err: %s\n", error.c_str());
            return;
        }
        printf("Saved exr file. [ %s ] \n", outputSppExr.c_str());
    }
}

// src/image.cpp:156-161 -> stbi_write_bmp(path, w, h, 3, data): 14 + 40 byte headers, rows bottom-up,
// pixels B G R, rows padded to four bytes
bool writeBmpRgb8(const std::string &path, int width, int height, const unsigned char *rgb)
{
    FILE *file = fopen(path.c_str(), "wb");
    if (!file) { return false; }
    const int pad = (-width * 3) & 3;
    auto put16 = [&](unsigned int v) { fputc(v & 0xFF, file); fputc((v >> 8) & 0xFF, file); };
    auto put32 = [&](unsigned int v) { put16(v & 0xFFFF); put16(v >> 16); };
    fputc('B', file); fputc('M', file);
    put32(14u + 40u + (unsigned int)(width * 3 + pad) * (unsigned int)height);
    put16(0); put16(0); put32(14u + 40u);
    put32(40u); put32((unsigned int)width); put32((unsigned int)height); put16(1); put16(24);
    put32(0); put32(0); put32(0); put32(0); put32(0); put32(0);
    for (int row = height - 1; row >= 0; row--) {
        const unsigned char *line = rgb + (size_t)3 * row * width;
        for (int col = 0; col < width; col++) {
            fputc(line[3 * col + 2], file);
            fputc(line[3 * col + 1], file);
            fputc(line[3 * col + 0], file);
        }
        for (int k = 0; k < pad; k++) { fputc(0, file); }
    }
    return fclose(file) == 0;
}

void Image::write(const std::string &filename)
{
    writeBmpRgb8(m_outputDirectory + filename, m_width, m_height, m_data.data());
}

}  // namespace pathed
