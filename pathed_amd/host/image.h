// Image: float accumulation target + EXR checkpoints, reference include/image.h,
// src/image.cpp:14-161.  set() flips vertically (row 0 of the integrator is the bottom
// scanline, EXR rows run top-down); save() writes B,G,R HALF uncompressed.
#pragma once

#include <mutex>
#include <string>
#include <vector>

namespace pathed {

class Image {
public:
    Image(int width, int height, const std::string &outputDirectory);

    void set(int row, int col, float r, float g, float b);
    void setSpp(int spp) { m_spp = spp; }

    void save(const std::string &filestem);
    void saveCheckpoint(const std::string &filestem);
    // <outdir>/<filename>: the 8-bit preview as a 24-bit BMP, byte for byte what stbi_write_bmp
    // writes (reference src/image.cpp:156-161)
    void write(const std::string &filename);

    std::mutex &getLock() { return m_lock; }
    const std::vector<unsigned char> &data() const { return m_data; }
    const std::vector<float> &raw() const { return m_raw; }
    int width() const { return m_width; }
    int height() const { return m_height; }

private:
    void save(const std::string &filestem, bool saveCheckpoint);

    int m_height, m_width;
    int m_spp;
    std::string m_outputDirectory;
    std::vector<unsigned char> m_data;  // 8-bit gamma preview (the reference's UI texture)
    std::vector<float> m_raw;           // interleaved RGB, top scanline first
    std::mutex m_lock;
};

// 24-bit BMP of top-down interleaved RGB, the layout stbi_write_bmp produces
bool writeBmpRgb8(const std::string &path, int width, int height, const unsigned char *rgb);

}  // namespace pathed
