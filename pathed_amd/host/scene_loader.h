// Scene JSON / OBJ / MTL / PLY / quad / sphere readers producing the flat
// PathedSceneDesc the C ABI consumes.
//
// The reference builds an object graph whose BSDF parameters are private and whose
// per-vertex attributes live only in Embree buffers, so a drop-in
// has to read the same files itself.  Format rules followed here:
//   scene JSON   reference src/scene_parser.cpp:140-200, 251-291, 574-667, 690-850
//   OBJ          reference src/obj_parser.cpp:50-515
//   MTL          reference src/mtl_parser.cpp:40-113
//   PLY          reference src/ply_parser.cpp:29-151
//   quad         reference src/quad.cpp:7-151
//   sphere       reference src/sphere.cpp:16-48, src/scene_parser.cpp:484-512
//   textures     reference src/texture.cpp:12-31 (PNG / PNM here, see image_decode.h)
#pragma once

#include "pathed_hip.h"

#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace pathed {

struct FlatScene {
    PathedCamera camera;

    std::vector<float> positions;
    std::vector<float> normals;
    std::vector<float> uvs;
    std::vector<uint32_t> indices;
    std::vector<int32_t> triMaterial;

    std::vector<PathedSphere> spheres;
    std::vector<PathedGeom> geoms;
    std::vector<PathedMaterial> materials;
    std::vector<PathedMedium> media;

    bool hasEnv = false;
    PathedEnvLight env;
    std::vector<float> envData;

    // image textures (reference src/texture.cpp): 8-bit RGB, one entry per distinct file
    std::vector<PathedTexture> textures;
    std::vector<std::vector<uint8_t>> textureData;
    std::map<std::string, int> textureByPath;

    // valid as long as this FlatScene is alive and unmodified
    PathedSceneDesc desc() const;
};

struct SceneLoadError : std::runtime_error {
    explicit SceneLoadError(const std::string &what) : std::runtime_error(what) {}
};

// assetRoot plays the role of the reference's working directory after its
// chdir("..") (app/main.cpp:60): every filename inside the scene is relative to it.
FlatScene loadScene(
    const std::string &scenePath,
    int width, int height,
    const std::string &assetRoot
);

PathedMaterial makeLambertian(const float diffuse[3], const float emit[3]);

// The text layer under the OBJ / MTL readers, exposed so the tests can hold it against the reference's own
// tokenizer and MtlParser (src/string_util.cpp:7-38, src/mtl_parser.cpp:16-113; test/string_util_test.cpp:9-37).
struct MtlMaterial {
    std::string name;
    float diffuse[3] = { 0.f, 0.f, 0.f };
    float emit[3] = { 0.f, 0.f, 0.f };
};
std::vector<std::string> tokenizeLine(const std::string &line);
std::string leftTrim(const std::string &token);
std::vector<MtlMaterial> parseMtlFile(const std::string &path);

}  // namespace pathed
