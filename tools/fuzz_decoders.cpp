// Mutation fuzz of the host's file decoders (PNG / PNM / JPEG in image_decode.cpp, EXR incl. ZIP and PIZ in
// exr.cpp) under AddressSanitizer + UBSan on the CPU: every fixture file is truncated or has bytes replaced /
// bits flipped ROUNDS times (argv[1]) and decoded; the decoders must reject or decode, never touch memory out of bounds.
//   g++ -std=c++17 -O1 -g -fwrapv -fsanitize=address,undefined -fno-sanitize-recover=undefined -Iinclude \
//       -o /tmp/fuzz_decoders tools/fuzz_decoders.cpp pathed_amd/host/image_decode.cpp pathed_amd/host/exr.cpp -lz
//   /tmp/fuzz_decoders 400 tests/golden/textures/* test_scenes/1_pixel_test.exr
#include "../pathed_amd/host/image_decode.h"
#include "../pathed_amd/host/exr.h"
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>
int main(int argc, char **argv) {
    unsigned int seed = 1;
    long ok = 0, failed = 0;
    const int rounds = argc > 1 ? atoi(argv[1]) : 100;
    const std::string scratch = std::string("/tmp/pathed_fuzz_case_") + std::to_string((long)getpid()) + ".bin";
    for (int a = 2; a < argc; a++) {
        std::ifstream in(argv[a], std::ios::binary);
        std::vector<unsigned char> original((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        const bool exr = std::string(argv[a]).find(".exr") != std::string::npos;
        for (int round = 0; round < rounds; round++) {
            std::vector<unsigned char> data = original;
            seed = seed * 1664525u + 1013904223u;
            const int mode = (seed >> 28) & 3;
            if (mode == 0 && data.size() > 8) { data.resize(8 + (seed >> 4) % (data.size() - 8)); }            // truncate
            const int flips = 1 + (seed >> 8) % 6;
            for (int f = 0; f < flips && mode != 0; f++) {
                seed = seed * 1664525u + 1013904223u;
                const size_t at = (seed >> 3) % data.size();
                data[at] = mode == 1 ? (unsigned char)(seed >> 20) : (unsigned char)(data[at] ^ (1u << ((seed >> 13) & 7)));
            }
            const char *path = scratch.c_str();
            { std::ofstream out(path, std::ios::binary); out.write((const char *)data.data(), (std::streamsize)data.size()); }
            std::string error;
            int w = 0, h = 0;
            bool good;
            if (exr) { std::vector<float> rgba; good = pathed::readExrRGBA(path, &w, &h, &rgba, &error); }
            else { std::vector<uint8_t> rgb; good = pathed::loadImageRgb8(path, &w, &h, &rgb, &error); }
            good ? ok++ : failed++;
        }
    }
    // directed case: every 8-byte field in the head of an EXR file (among them the entries of the block-offset
    // table) replaced by a value just below 2^64 -- an offset check written `pos + 8 > size` wraps and passes
    for (int a = 2; a < argc; a++) {
        if (std::string(argv[a]).find(".exr") == std::string::npos) { continue; }
        std::ifstream in(argv[a], std::ios::binary);
        const std::vector<unsigned char> original((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        if (original.size() > (1u << 20)) { continue; }
        const size_t span = original.size() < 8 ? 0 : (original.size() - 8 < 1024 ? original.size() - 8 : 1024);
        for (size_t at = 0; at < span; at++) {
            std::vector<unsigned char> data = original;
            const unsigned char huge[8] = { 0xFA, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF };   // little-endian 0xFFFFFFFFFFFFFFFA
            for (int k = 0; k < 8; k++) { data[at + k] = huge[k]; }
            { std::ofstream out(scratch.c_str(), std::ios::binary); out.write((const char *)data.data(), (std::streamsize)data.size()); }
            std::string error;
            int w = 0, h = 0;
            std::vector<float> rgba;
            pathed::readExrRGBA(scratch.c_str(), &w, &h, &rgba, &error) ? ok++ : failed++;
        }
    }
    remove(scratch.c_str());
    printf("decoded %ld, rejected %ld, no crash\n", ok, failed);
    return 0;
}
