// CPU sanitizer driver (tests/test_sanitizers.py): the host loaders, the host BVH builder and the CPU oracle under
// AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY.md §5 planned it; the reference has none and has real races,
// src/random_generator.cpp:4-6, src/camera.cpp:51-52).  Loads every scene on the command line through the product's
// loader, renders it with the oracle (two OpenMP threads; one in the ThreadSanitizer build, see below), builds the 4-wide tree with the host builder -- single- and multi-threaded, which must
// agree -- and prints one line per scene: "<scene> <w>x<h> checksum <sum of the radiance sums>".
//   usage: sanitize_host <asset root> <width> <height> <spp> <scene.json>...
#include "../oracle/oracle.h"
#include "../pathed_amd/csrc/bvh_build.h"
#include "../pathed_amd/host/scene_loader.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

int main(int argc, char **argv)
{
    if (argc < 6) { fprintf(stderr, "usage: %s <asset root> <width> <height> <spp> <scene.json>...\n", argv[0]); return 2; }
    const std::string root = argv[1];
    const int width = atoi(argv[2]), height = atoi(argv[3]), spp = atoi(argv[4]);
    // the oracle's OpenMP threads: 1 under ThreadSanitizer (libgomp's barriers are not annotated, so the tool reports the
    // hand-over at the end of a parallel region as a race)
    const int oracleThreads = getenv("SANITIZE_ORACLE_THREADS") ? atoi(getenv("SANITIZE_ORACLE_THREADS")) : 2;
    for (int i = 5; i < argc; i++) {
        try {
            pathed::FlatScene flat = pathed::loadScene(argv[i], width, height, root);
            const PathedSceneDesc desc = flat.desc();
            OracleScene *oracle = oracle_scene_create(&desc);
            if (!oracle) { fprintf(stderr, "%s: %s\n", argv[i], oracle_last_error()); return 1; }
            std::vector<float> sums((size_t)3 * width * height, 0.f);
            uint64_t stats[8] = { 0 };
            if (oracle_render(oracle, 1, 0, (uint32_t)spp, 0, 8, sums.data(), oracleThreads, stats) != 0) { fprintf(stderr, "%s: %s\n", argv[i], oracle_last_error()); return 1; }
            oracle_scene_destroy(oracle);
            double checksum = 0.0;
            for (float value : sums) { checksum += value; }
            // the product's host builder on the same soup, one thread and several: the same tree
            std::vector<float> spheres;
            for (uint32_t s = 0; s < desc.n_spheres; s++) {
                for (int a = 0; a < 3; a++) { spheres.push_back(desc.spheres[s].center_world[a]); }
                spheres.push_back(desc.spheres[s].radius);
            }
            const pathed::FlatBvh one = pathed::buildBvh(desc.positions, desc.indices, desc.n_triangles, spheres.data(), desc.n_spheres, 1);
            const pathed::FlatBvh many = pathed::buildBvh(desc.positions, desc.indices, desc.n_triangles, spheres.data(), desc.n_spheres, 4);
            if (one.nodes.size() != many.nodes.size() || std::memcmp(one.nodes.data(), many.nodes.data(), one.nodes.size() * sizeof(float)) != 0) {
                fprintf(stderr, "%s: the threaded build differs\n", argv[i]);
                return 1;
            }
            printf("%s %dx%d checksum %.6e nodes %d depth %d\n", argv[i], width, height, checksum, one.nodeCount, one.maxDepth);
        } catch (const std::exception &error) {
            fprintf(stderr, "%s: %s\n", argv[i], error.what());
            return 1;
        }
    }
    // a mesh large enough for the builder's threaded path (>= 200 K primitives): a displaced grid
    {
        const int n = 360;   // 2 * 359 * 359 = 257 762 triangles
        std::vector<float> positions;
        std::vector<uint32_t> indices;
        for (int y = 0; y < n; y++) {
            for (int x = 0; x < n; x++) {
                positions.push_back((float)x);
                positions.push_back(3.f * (float)(((x * 7919 + y * 104729) % 97)) / 97.f);
                positions.push_back((float)y);
            }
        }
        for (int y = 0; y + 1 < n; y++) {
            for (int x = 0; x + 1 < n; x++) {
                const uint32_t a = (uint32_t)(y * n + x), b = a + 1, c = a + (uint32_t)n, d = c + 1;
                indices.insert(indices.end(), { a, b, d, a, d, c });
            }
        }
        const uint32_t triangles = (uint32_t)(indices.size() / 3);
        const pathed::FlatBvh one = pathed::buildBvh(positions.data(), indices.data(), triangles, nullptr, 0, 1);
        const pathed::FlatBvh many = pathed::buildBvh(positions.data(), indices.data(), triangles, nullptr, 0, 6);
        if (one.nodes.size() != many.nodes.size() || std::memcmp(one.nodes.data(), many.nodes.data(), one.nodes.size() * sizeof(float)) != 0
            || std::memcmp(one.leafTris.data(), many.leafTris.data(), one.leafTris.size() * sizeof(float)) != 0) {
            fprintf(stderr, "grid mesh: the threaded build differs\n");
            return 1;
        }
        printf("grid mesh %u triangles: nodes %d depth %d, threaded build identical\n", triangles, one.nodeCount, one.maxDepth);
    }
    printf("sanitize_host: done\n");
    return 0;
}
