#include "scene_loader.h"

#include "image_decode.h"

#include "exr.h"
#include "json.h"
#include "transform.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <utility>

namespace pathed {

namespace {

using MaterialMap = std::map<std::string, int>;

struct LoaderContext {
    FlatScene *scene;
    std::string assetRoot;
    MaterialMap materialLookup;
    std::map<std::string, int> mediumLookup;   // scene JSON "media" by name -> index into FlatScene::media

    std::string resolve(const std::string &filename) const
    {
        if (!filename.empty() && filename[0] == '/') { return filename; }
        if (assetRoot.empty()) { return filename; }
        return assetRoot + "/" + filename;
    }

    int addMaterial(const PathedMaterial &material)
    {
        scene->materials.push_back(material);
        return (int)scene->materials.size() - 1;
    }

    // Texture::load, reference src/texture.cpp:12-31 (throws "Error loading texture")
    int addTexture(const std::string &filename)
    {
        const std::string path = resolve(filename);
        auto found = scene->textureByPath.find(path);
        if (found != scene->textureByPath.end()) { return found->second; }
        PathedTexture texture;
        std::vector<uint8_t> data;
        std::string error;
        if (!loadImageRgb8(path, &texture.width, &texture.height, &data, &error)) {
            throw SceneLoadError("Error loading texture: " + error);
        }
        texture.rgb = nullptr;  // set by FlatScene::desc()
        scene->textures.push_back(texture);
        scene->textureData.push_back(std::move(data));
        const int index = (int)scene->textures.size() - 1;
        scene->textureByPath[path] = index;
        return index;
    }
};

// ---- scalar helpers: scene-JSON numbers are strings (scene_parser.cpp:814-817) ----

bool checkFloat(const Json &value, float *out)
{
    if (value.isString()) {
        *out = std::stof(value.asString());
        return true;
    }
    if (value.isNumber()) {  // superset: plain numbers are accepted too
        *out = (float)value.asNumber();
        return true;
    }
    return false;
}

float parseFloat(const Json &value, const char *what)
{
    float out;
    if (!checkFloat(value, &out)) {
        throw SceneLoadError(std::string("scene: expected a number (string) for ") + what);
    }
    return out;
}

float parseFloatDefault(const Json &value, float defaultValue)
{
    float out;
    return checkFloat(value, &out) ? out : defaultValue;
}

bool parseBool(const Json &value, bool defaultValue)
{
    return value.isBool() ? value.asBool() : defaultValue;
}

void parseTriple(const Json &value, float out[3], const char *what)
{
    if (!value.isArray() || value.size() < 3) {
        throw SceneLoadError(std::string("scene: expected a 3-array for ") + what);
    }
    for (int i = 0; i < 3; i++) { out[i] = parseFloat(value[(size_t)i], what); }
}

// parseColor(json, defaultColor): array -> colour, anything else -> default
void parseColor(const Json &value, const float defaultColor[3], float out[3])
{
    if (value.isArray()) {
        parseTriple(value, out, "color");
    } else {
        for (int i = 0; i < 3; i++) { out[i] = defaultColor[i]; }
    }
}

const float kBlack[3] = { 0.f, 0.f, 0.f };
const float kWhite[3] = { 1.f, 1.f, 1.f };

// ---- transforms (scene_parser.cpp:690-793) ----------------------------------------

Transform parseTransform(const Json &json)
{
    Transform transform;
    if (!json.isObject()) { return transform; }

    const bool legacyMode = parseBool(json["legacy"], false);

    float scaleX = 1.f, scaleY = 1.f, scaleZ = 1.f;
    const Json &scale = json["scale"];
    if (scale.isArray()) {
        scaleX = parseFloat(scale[0], "scale");
        scaleY = parseFloat(scale[1], "scale");
        scaleZ = parseFloat(scale[2], "scale");
    }

    float rotateX = 0.f, rotateY = 0.f, rotateZ = 0.f;
    const Json &rotate = json["rotate"];
    if (rotate.isArray()) {
        // float * M_PI / 180.f evaluates in double and narrows on assignment
        rotateX = (float)((double)parseFloat(rotate[0], "rotate") * M_PI / (double)180.f);
        rotateY = (float)((double)parseFloat(rotate[1], "rotate") * M_PI / (double)180.f);
        rotateZ = (float)((double)parseFloat(rotate[2], "rotate") * M_PI / (double)180.f);
        if (legacyMode) {
            rotateX *= -1;
            rotateZ *= -1;
        } else {
            rotateY *= -1;
        }
    }

    float translateX = 0.f, translateY = 0.f, translateZ = 0.f;
    const Json &translate = json["translate"];
    if (translate.isArray()) {
        translateX = parseFloat(translate[0], "translate");
        translateY = parseFloat(translate[1], "translate");
        translateZ = parseFloat(translate[2], "translate");
    }

    Mat4 &matrix = transform.matrix;
    matrix.preScale(scaleX, scaleY, scaleZ);
    if (legacyMode) {
        matrix.preRotateX(rotateX);
        matrix.preRotateY(rotateY);
        matrix.preRotateZ(rotateZ);
    } else {
        matrix.preRotateZ(rotateZ);
        matrix.preRotateX(rotateX);
        matrix.preRotateY(rotateY);
    }
    matrix.preTranslate(translateX, translateY, translateZ);

    Mat4 &inverse = transform.inverse;
    inverse.preTranslate(-translateX, -translateY, -translateZ);
    if (legacyMode) {
        inverse.preRotateZ(-rotateZ);
        inverse.preRotateY(-rotateY);
        inverse.preRotateX(-rotateX);
    } else {
        inverse.preRotateY(-rotateY);
        inverse.preRotateX(-rotateX);
        inverse.preRotateZ(-rotateZ);
    }
    inverse.preScale(1.f / scaleX, 1.f / scaleY, 1.f / scaleZ);

    return transform;
}

// ---- materials (scene_parser.cpp:555-683) ------------------------------------------

PathedMaterial blankMaterial(int type)
{
    PathedMaterial material;
    std::memset(&material, 0, sizeof material);
    material.type = type;
    material.albedo_type = PATHED_ALBEDO_CONSTANT;
    material.ior = 1.4f;
    material.distribution = PATHED_DIST_BECKMANN;
    return material;
}

void parseDistribution(const Json &json, PathedMaterial *material)
{
    material->alpha = parseFloat(json["alpha"], "distribution.alpha");
    const std::string type = json["type"].isString() ? json["type"].asString() : "<missing>";
    if (type == "beckmann") {
        material->distribution = PATHED_DIST_BECKMANN;
    } else if (type == "ggx") {
        material->distribution = PATHED_DIST_GGX;
    } else {
        throw SceneLoadError("Unimplemented distribution: " + type);
    }
}

// returns a material index, or -1 for "no bsdf object" (reference returns nullptr)
int parseMaterial(const Json &json, LoaderContext &context)
{
    if (!json.isObject()) { return -1; }

    const std::string type = json["type"].isString() ? json["type"].asString() : "<missing>";

    if (type == "reference") {
        const std::string name = json["name"].asString();
        auto found = context.materialLookup.find(name);
        if (found == context.materialLookup.end()) {
            throw SceneLoadError("scene: unknown material reference: " + name);
        }
        return found->second;
    } else if (type == "mirror") {
        return context.addMaterial(blankMaterial(PATHED_MAT_MIRROR));
    } else if (type == "passthrough") {
        // the boundary of a participating medium, scene_parser.cpp:593-594
        return context.addMaterial(blankMaterial(PATHED_MAT_PASSTHROUGH));
    } else if (type == "glass") {
        PathedMaterial material = blankMaterial(PATHED_MAT_GLASS);
        float ior;
        if (checkFloat(json["ior"], &ior)) { material.ior = ior; }
        return context.addMaterial(material);
    } else if (type == "oren-nayar") {
        PathedMaterial material = blankMaterial(PATHED_MAT_OREN_NAYAR);
        parseColor(json["diffuseReflectance"], kWhite, material.diffuse);
        material.sigma = parseFloat(json["sigma"], "oren-nayar sigma");
        return context.addMaterial(material);
    } else if (type == "microfacet") {
        PathedMaterial material = blankMaterial(PATHED_MAT_MICROFACET);
        parseDistribution(json["distribution"], &material);
        return context.addMaterial(material);
    } else if (type == "plastic") {
        PathedMaterial material = blankMaterial(PATHED_MAT_PLASTIC);
        parseColor(json["diffuseReflectance"], kBlack, material.diffuse);
        parseDistribution(json["distribution"], &material);
        if (json["texture"].isString()) {
            // Plastic(Lambertian(texture, 0), distribution), scene_parser.cpp:624-636
            material.albedo_type = PATHED_ALBEDO_TEXTURE;
            material.texture = context.addTexture(json["texture"].asString());
            for (int i = 0; i < 3; i++) { material.diffuse[i] = 0.f; }
        }
        return context.addMaterial(material);
    } else if (type == "lambertian") {
        PathedMaterial material = blankMaterial(PATHED_MAT_LAMBERTIAN);
        parseColor(json["diffuseReflectance"], kBlack, material.diffuse);
        parseColor(json["emit"], kBlack, material.emit);
        const Json &albedo = json["albedo"];
        if (json["texture"].isString()) {
            // Lambertian(texture, emit), scene_parser.cpp:641-648; takes precedence over "albedo"
            material.albedo_type = PATHED_ALBEDO_TEXTURE;
            material.texture = context.addTexture(json["texture"].asString());
            for (int i = 0; i < 3; i++) { material.diffuse[i] = 0.f; }
        } else if (albedo.isObject() && albedo["type"].isString() && albedo["type"].asString() == "checkerboard") {
            material.albedo_type = PATHED_ALBEDO_CHECKERBOARD;
            parseColor(albedo["onColor"], kBlack, material.checker_on);
            parseColor(albedo["offColor"], kBlack, material.checker_off);
            material.checker_res[0] = parseFloat(albedo["resolution"]["u"], "checkerboard resolution");
            material.checker_res[1] = parseFloat(albedo["resolution"]["v"], "checkerboard resolution");
            // the reference's Lambertian(albedo, emit) ctor zeroes m_diffuse (lambertian.cpp:12-14)
            for (int i = 0; i < 3; i++) { material.diffuse[i] = 0.f; }
        }
        return context.addMaterial(material);
    } else if (type == "phong" || type == "passthrough" || type == "perfect-transmission"
               || type == "ptex" || type == "disney") {
        throw SceneLoadError("Unsupported material (outside hot-path scope, SURVEY.md §2 #15): " + type);
    }
    throw SceneLoadError("Unimplemented material: " + type);
}

// ---- geometry accumulation ---------------------------------------------------------

struct MeshBuffers {
    std::vector<float> positions;   // 3 per vertex
    std::vector<float> normals;     // 3 per vertex (zero = none)
    std::vector<float> uvs;         // 2 per vertex
    std::vector<uint32_t> indices;  // 3 per face
    std::vector<int32_t> faceMaterial;
};

void appendMesh(FlatScene &scene, const MeshBuffers &mesh, int medium)
{
    const uint32_t vertexBase = (uint32_t)(scene.positions.size() / 3);
    const size_t vertexCount = mesh.positions.size() / 3;

    scene.positions.insert(scene.positions.end(), mesh.positions.begin(), mesh.positions.end());

    // processRTCGeometry pads missing attributes with zeros (geometry_parser.cpp:68-89)
    for (size_t i = 0; i < vertexCount; i++) {
        for (int c = 0; c < 3; c++) {
            const size_t k = 3 * i + c;
            scene.normals.push_back(k < mesh.normals.size() ? mesh.normals[k] : 0.f);
        }
        for (int c = 0; c < 2; c++) {
            const size_t k = 2 * i + c;
            scene.uvs.push_back(k < mesh.uvs.size() ? mesh.uvs[k] : 0.f);
        }
    }

    PathedGeom geom;
    geom.type = PATHED_GEOM_MESH;
    geom.first = (int32_t)(scene.indices.size() / 3);
    geom.count = (int32_t)(mesh.indices.size() / 3);
    geom.medium = medium;
    scene.geoms.push_back(geom);

    for (uint32_t index : mesh.indices) { scene.indices.push_back(vertexBase + index); }
    scene.triMaterial.insert(scene.triMaterial.end(), mesh.faceMaterial.begin(), mesh.faceMaterial.end());
}

// ---- MTL (mtl_parser.cpp) -----------------------------------------------------------

std::string lTrim(const std::string &token)
{
    const size_t first = token.find_first_not_of(" \t");
    if (first == std::string::npos) { return ""; }
    return token.substr(first);
}

std::vector<std::string> tokenize(const std::string &line)
{
    std::vector<std::string> tokens;
    std::string remaining = lTrim(line);
    while (!remaining.empty()) {
        const size_t end = remaining.find_first_of(" \t");
        if (end == std::string::npos) {
            tokens.push_back(remaining);
            break;
        }
        tokens.push_back(remaining.substr(0, end));
        remaining = lTrim(remaining.substr(end));
    }
    return tokens;
}

std::string stripCarriageReturn(std::string line)
{
    while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) { line.pop_back(); }
    return line;
}

// MtlParser::parse, src/mtl_parser.cpp:16-113: newmtl / Kd / Ke only, a later newmtl of the same name starts
// over, Kd / Ke ahead of any newmtl land on the material named "".  Entries in std::map (name) order.
std::vector<MtlMaterial> readMtlFile(const std::string &path)
{
    std::map<std::string, MtlMaterial> entries;
    std::string current;

    std::ifstream file(path);
    // like the reference, a missing library silently yields no materials
    std::string line;
    while (std::getline(file, line)) {
        std::vector<std::string> tokens = tokenize(stripCarriageReturn(line));
        if (tokens.empty()) { continue; }
        const std::string &command = tokens[0];
        if (command == "newmtl" && tokens.size() >= 2) {
            current = tokens[1];
            entries[current] = MtlMaterial();
            entries[current].name = current;
        } else if (command == "Kd" && tokens.size() >= 4) {
            entries[current].name = current;
            for (int i = 0; i < 3; i++) { entries[current].diffuse[i] = std::stof(tokens[(size_t)i + 1]); }
        } else if (command == "Ke" && tokens.size() >= 4) {
            entries[current].name = current;
            for (int i = 0; i < 3; i++) { entries[current].emit[i] = std::stof(tokens[(size_t)i + 1]); }
        }
    }
    std::vector<MtlMaterial> materials;
    for (const auto &item : entries) { materials.push_back(item.second); }
    return materials;
}

std::map<std::string, int> parseMtl(const std::string &path, LoaderContext &context)
{
    std::map<std::string, int> lookup;
    for (const MtlMaterial &material : readMtlFile(path)) {
        lookup[material.name] = context.addMaterial(makeLambertian(material.diffuse, material.emit));
    }
    return lookup;
}

// ---- OBJ (obj_parser.cpp) -----------------------------------------------------------

struct VertexRef {
    int vertexIndex;
    int normalIndex;
    int uvIndex;
};

struct FaceRef {
    VertexRef vertices[3];
};

class ObjReader {
public:
    ObjReader(LoaderContext &context, const Transform &transform, const std::string &prefix, int defaultMaterial)
        : m_context(context), m_transform(transform), m_prefix(prefix), m_defaultMaterial(defaultMaterial)
    {
        if (m_defaultMaterial < 0) {
            // obj_parser.cpp:40-45: red Lambertian when the model has no bsdf
            const float red[3] = { 1.f, 0.f, 0.f };
            m_defaultMaterial = m_context.addMaterial(makeLambertian(red, kBlack));
        }
    }

    MeshBuffers parse(const std::string &path)
    {
        std::ifstream file(path);
        if (!file) { throw SceneLoadError("obj: cannot open " + path); }
        std::string line;
        while (std::getline(file, line)) { parseLine(stripCarriageReturn(line)); }
        return finish();
    }

private:
    LoaderContext &m_context;
    Transform m_transform;
    std::string m_prefix;
    int m_defaultMaterial;

    std::string m_currentGroup;
    std::string m_currentMaterialName;
    std::map<std::string, int> m_mtlLookup;

    std::vector<float> m_vertices;       // 3 per vertex, transformed
    std::vector<float> m_normals;        // 3 per vn, transformed (not renormalised)
    std::vector<float> m_objUVs;         // 2 per vt
    std::vector<FaceRef> m_faces;
    std::vector<int32_t> m_faceMaterial;
    std::vector<float> m_vertexUVs;      // 2 per vertex, last writer wins

    void parseLine(const std::string &line)
    {
        if (line.empty()) { return; }
        const size_t space = line.find_first_of(" \t");
        if (space == std::string::npos) { return; }
        const std::string command = line.substr(0, space);
        if (command.empty() || command[0] == '#') { return; }
        const std::string rest = lTrim(line.substr(space + 1));

        if (command == "v") {
            float p[3], q[3];
            readFloats(rest, p, 3);
            m_transform.matrix.applyPoint(p, q);
            m_vertices.insert(m_vertices.end(), q, q + 3);
        } else if (command == "vn") {
            float n[3], q[3];
            readFloats(rest, n, 3);
            m_transform.matrix.applyVector(n, q);
            m_normals.insert(m_normals.end(), q, q + 3);
        } else if (command == "vt") {
            float uv[2];
            readFloats(rest, uv, 2);
            m_objUVs.insert(m_objUVs.end(), uv, uv + 2);
        } else if (command == "g") {
            m_currentGroup = lTrim(rest);
        } else if (command == "f") {
            if (m_currentMaterialName == "hidden") { return; }
            processFace(rest);
        } else if (command == "mtllib") {
            m_mtlLookup = parseMtl(m_context.resolve(rest), m_context);
        } else if (command == "usemtl") {
            m_currentMaterialName = rest;
        }
    }

    static void readFloats(const std::string &text, float *out, int count)
    {
        const char *cursor = text.c_str();
        for (int i = 0; i < count; i++) {
            char *end = nullptr;
            out[i] = std::strtof(cursor, &end);
            if (end == cursor) { throw SceneLoadError("obj: bad number in '" + text + "'"); }
            cursor = end;
        }
    }

    int currentMaterial() const
    {
        // precedence: obj_parser.cpp:241-252
        const std::string groupKey = m_prefix + m_currentGroup;
        const std::string mtlKey = m_prefix + m_currentMaterialName;
        auto byGroup = m_context.materialLookup.find(groupKey);
        if (byGroup != m_context.materialLookup.end()) { return byGroup->second; }
        auto byName = m_context.materialLookup.find(mtlKey);
        if (byName != m_context.materialLookup.end()) { return byName->second; }
        auto byMtl = m_mtlLookup.find(m_currentMaterialName);
        if (byMtl != m_mtlLookup.end()) { return byMtl->second; }
        return m_defaultMaterial;
    }

    static int correctIndex(size_t count, int index)
    {
        // obj_parser.cpp:221-228: negative = relative to the current end, else 1-based
        return index < 0 ? index + (int)count : index - 1;
    }

    void addTriangle(const VertexRef raw[3], bool hasNormals, bool hasUVs)
    {
        FaceRef face;
        for (int i = 0; i < 3; i++) {
            face.vertices[i].vertexIndex = correctIndex(m_vertices.size() / 3, raw[i].vertexIndex);
            face.vertices[i].normalIndex = hasNormals ? correctIndex(m_normals.size() / 3, raw[i].normalIndex) : -1;
            face.vertices[i].uvIndex = hasUVs ? correctIndex(m_objUVs.size() / 2, raw[i].uvIndex) : -1;
            const int v = face.vertices[i].vertexIndex;
            if (v < 0 || (size_t)v >= m_vertices.size() / 3) { throw SceneLoadError("obj: vertex index out of range"); }
            if (hasNormals) {
                const int n = face.vertices[i].normalIndex;
                if (n < 0 || (size_t)n >= m_normals.size() / 3) { throw SceneLoadError("obj: normal index out of range"); }
            }
            if (hasUVs) {
                const int t = face.vertices[i].uvIndex;
                if (t < 0 || (size_t)t >= m_objUVs.size() / 2) { throw SceneLoadError("obj: uv index out of range"); }
            }
        }

        if (hasUVs) {
            // obj_parser.cpp:292-295: per-vertex uv table sized to the vertices seen so far
            m_vertexUVs.resize(2 * (m_vertices.size() / 3), 0.f);
            for (int i = 0; i < 3; i++) {
                const int v = face.vertices[i].vertexIndex;
                const int t = face.vertices[i].uvIndex;
                m_vertexUVs[2 * (size_t)v + 0] = m_objUVs[2 * (size_t)t + 0];
                m_vertexUVs[2 * (size_t)v + 1] = m_objUVs[2 * (size_t)t + 1];
            }
        }

        m_faces.push_back(face);
        m_faceMaterial.push_back(currentMaterial());
    }

    static bool parseCorner(const std::string &token, VertexRef *out, int *form)
    {
        // form: 0 = "v", 1 = "v/t/n", 2 = "v//n"
        const char *cursor = token.c_str();
        char *end = nullptr;
        long v = std::strtol(cursor, &end, 10);
        if (end == cursor) { return false; }
        out->vertexIndex = (int)v;
        out->normalIndex = 0;
        out->uvIndex = 0;
        if (*end == '\0') { *form = 0; return true; }
        if (*end != '/') { return false; }
        cursor = end + 1;
        if (*cursor == '/') {
            cursor++;
            long n = std::strtol(cursor, &end, 10);
            if (end == cursor || *end != '\0') { return false; }
            out->normalIndex = (int)n;
            *form = 2;
            return true;
        }
        long t = std::strtol(cursor, &end, 10);
        if (end == cursor || *end != '/') { return false; }
        cursor = end + 1;
        long n = std::strtol(cursor, &end, 10);
        if (end == cursor || *end != '\0') { return false; }
        out->uvIndex = (int)t;
        out->normalIndex = (int)n;
        *form = 1;
        return true;
    }

    void processFace(const std::string &args)
    {
        std::vector<std::string> tokens = tokenize(args);
        if (tokens.size() < 3) { throw SceneLoadError("obj: face with fewer than 3 corners"); }

        VertexRef corners[4];
        int form = -1;
        const size_t count = tokens.size() > 4 ? 4 : tokens.size();
        for (size_t i = 0; i < count; i++) {
            int cornerForm;
            if (!parseCorner(tokens[i], &corners[i], &cornerForm)) {
                throw SceneLoadError("obj: unsupported face syntax '" + args + "'");
            }
            if (i == 0) { form = cornerForm; }
            else if (cornerForm != form) { throw SceneLoadError("obj: mixed face syntax '" + args + "'"); }
        }

        const bool hasNormals = (form != 0);
        const bool hasUVs = (form == 1);

        // the reference accepts quads only as "v v v v" and "v//n v//n v//n v//n"
        // (obj_parser.cpp:377-396, 455-489); quads split (0,1,2),(0,2,3)
        if (tokens.size() == 4 && form != 1) {
            const VertexRef first[3] = { corners[0], corners[1], corners[2] };
            const VertexRef second[3] = { corners[0], corners[2], corners[3] };
            addTriangle(first, hasNormals, hasUVs);
            addTriangle(second, hasNormals, hasUVs);
        } else if (tokens.size() == 3 || form == 0) {
            const VertexRef only[3] = { corners[0], corners[1], corners[2] };
            addTriangle(only, hasNormals, hasUVs);
        } else {
            throw SceneLoadError("obj: unsupported face syntax '" + args + "'");
        }
    }

    MeshBuffers finish()
    {
        // "cube-normal" correction, obj_parser.cpp:60-117: a vertex re-used with a
        // different normal index is duplicated at the back of the vertex list
        std::map<int, int> normalLookup;
        std::map<std::pair<int, int>, int> correctionLookup;

        for (FaceRef &face : m_faces) {
            for (int j = 0; j < 3; j++) {
                VertexRef &ref = face.vertices[j];
                auto seen = normalLookup.find(ref.vertexIndex);
                if (seen == normalLookup.end()) {
                    normalLookup[ref.vertexIndex] = ref.normalIndex;
                } else if (seen->second != ref.normalIndex) {
                    const std::pair<int, int> key(ref.vertexIndex, ref.normalIndex);
                    auto corrected = correctionLookup.find(key);
                    int correctedIndex;
                    if (corrected == correctionLookup.end()) {
                        for (int c = 0; c < 3; c++) {
                            const float value = m_vertices[3 * (size_t)ref.vertexIndex + c];
                            m_vertices.push_back(value);
                        }
                        correctedIndex = (int)(m_vertices.size() / 3) - 1;
                        correctionLookup[key] = correctedIndex;
                    } else {
                        correctedIndex = corrected->second;
                    }
                    ref.vertexIndex = correctedIndex;
                }
            }
        }

        MeshBuffers mesh;
        mesh.positions = m_vertices;
        mesh.normals.assign(m_vertices.size(), 0.f);
        for (const FaceRef &face : m_faces) {
            for (int j = 0; j < 3; j++) {
                const VertexRef &ref = face.vertices[j];
                if (ref.normalIndex != -1) {
                    for (int c = 0; c < 3; c++) {
                        mesh.normals[3 * (size_t)ref.vertexIndex + c] = m_normals[3 * (size_t)ref.normalIndex + c];
                    }
                }
                mesh.indices.push_back((uint32_t)ref.vertexIndex);
            }
        }
        mesh.uvs = m_vertexUVs;  // duplicated vertices keep uv (0,0), as in the reference
        mesh.faceMaterial = m_faceMaterial;
        return mesh;
    }
};

// ---- PLY (ply_parser.cpp:29-151) ------------------------------------------------------

MeshBuffers parsePly(const std::string &path, const Transform &transform, int material)
{
    std::ifstream file(path, std::ios::binary);
    if (!file) { throw SceneLoadError("ply: cannot open " + path); }

    auto nextLine = [&]() {
        std::string line;
        std::getline(file, line);
        return stripCarriageReturn(line);
    };
    auto expectLine = [&](const std::string &expected) {
        const std::string line = nextLine();
        if (line != expected) { throw SceneLoadError("ply: expected '" + expected + "', got '" + line + "'"); }
    };
    auto countAfter = [&](const std::string &prefix) {
        const std::string line = nextLine();
        if (line.compare(0, prefix.size(), prefix) != 0) {
            throw SceneLoadError("ply: expected '" + prefix + "N', got '" + line + "'");
        }
        return std::stoi(line.substr(prefix.size()));
    };

    expectLine("ply");
    expectLine("format binary_little_endian 1.0");
    const int vertexCount = countAfter("element vertex ");
    expectLine("property float x");
    expectLine("property float y");
    expectLine("property float z");
    const int faceCount = countAfter("element face ");
    const std::string listLine = nextLine();
    if (listLine != "property list uint8 int vertex_indices" && listLine != "property list uchar int vertex_indices") {
        throw SceneLoadError("ply: unsupported face property '" + listLine + "'");
    }
    expectLine("end_header");

    MeshBuffers mesh;
    mesh.positions.resize(3 * (size_t)vertexCount);
    for (int i = 0; i < vertexCount; i++) {
        float p[3], q[3];
        file.read((char *)p, 12);
        transform.matrix.applyPoint(p, q);
        for (int c = 0; c < 3; c++) { mesh.positions[3 * (size_t)i + c] = q[c]; }
    }
    mesh.indices.reserve(3 * (size_t)faceCount);
    for (int i = 0; i < faceCount; i++) {
        unsigned char faceSize = 0;
        file.read((char *)&faceSize, 1);
        if (faceSize != 3) { throw SceneLoadError("ply: only triangles are supported"); }
        int32_t index[3];
        file.read((char *)index, 12);
        for (int j = 0; j < 3; j++) {
            if (index[j] < 0 || index[j] >= vertexCount) { throw SceneLoadError("ply: index out of range"); }
            mesh.indices.push_back((uint32_t)index[j]);
        }
    }
    if (!file) { throw SceneLoadError("ply: truncated file " + path); }
    mesh.faceMaterial.assign((size_t)faceCount, material);
    return mesh;
}

// ---- quad (quad.cpp:7-151) -------------------------------------------------------------

MeshBuffers makeQuad(const Transform &transform, int material, bool zUp)
{
    static const float yUpPoints[6][3] = {
        { -1.f, 0.f, -1.f }, { -1.f, 0.f, 1.f }, { 1.f, 0.f, -1.f },
        { -1.f, 0.f, 1.f }, { 1.f, 0.f, 1.f }, { 1.f, 0.f, -1.f },
    };
    static const float zUpPoints[6][3] = {
        { -1.f, -1.f, 0.f }, { 1.f, -1.f, 0.f }, { -1.f, 1.f, 0.f },
        { -1.f, 1.f, 0.f }, { 1.f, -1.f, 0.f }, { 1.f, 1.f, 0.f },
    };
    static const float quadUVs[6][2] = {
        { 0.f, 0.f }, { 1.f, 0.f }, { 0.f, 1.f },
        { 0.f, 1.f }, { 1.f, 0.f }, { 1.f, 1.f },
    };

    MeshBuffers mesh;
    const float up[3] = { 0.f, zUp ? 0.f : 1.f, zUp ? 1.f : 0.f };
    float normal[3];
    transform.matrix.applyVector(up, normal);
    const float norm = sqrtf(normal[0] * normal[0] + normal[1] * normal[1] + normal[2] * normal[2]);
    for (int c = 0; c < 3; c++) { normal[c] = normal[c] / norm; }

    for (int i = 0; i < 6; i++) {
        float p[3];
        transform.matrix.applyPoint(zUp ? zUpPoints[i] : yUpPoints[i], p);
        mesh.positions.insert(mesh.positions.end(), p, p + 3);
        mesh.uvs.insert(mesh.uvs.end(), quadUVs[i], quadUVs[i] + 2);
        mesh.normals.insert(mesh.normals.end(), normal, normal + 3);
        mesh.indices.push_back((uint32_t)i);
    }
    mesh.faceMaterial.assign(2, material);
    return mesh;
}

int requireMaterial(int material, const char *what)
{
    if (material < 0) { throw SceneLoadError(std::string("scene: ") + what + " needs a bsdf"); }
    return material;
}

void parseModels(const Json &models, LoaderContext &context)
{
    FlatScene &scene = *context.scene;
    for (const Json &model : models.elements()) {
        if (parseBool(model["skip"], false)) { continue; }
        const std::string type = model["type"].isString() ? model["type"].asString() : "";

        // "internal_medium": scene_parser.cpp:324-337, 370-379, 503-514; an unknown name is a null medium there
        int medium = -1;
        if (model["internal_medium"].isString()) {
            auto found = context.mediumLookup.find(model["internal_medium"].asString());
            if (found != context.mediumLookup.end()) { medium = found->second; }
        }

        if (type == "obj") {
            const Transform transform = parseTransform(model["transform"]);
            const int material = parseMaterial(model["bsdf"], context);
            const std::string prefix = model["materialPrefix"].isString() ? model["materialPrefix"].asString() : "";
            ObjReader reader(context, transform, prefix, material);
            appendMesh(scene, reader.parse(context.resolve(model["filename"].asString())), medium);
        } else if (type == "ply") {
            const Transform transform = parseTransform(model["transform"]);
            int material = parseMaterial(model["bsdf"], context);
            if (material < 0) {
                // ply_parser.cpp:118-120: green Lambertian placeholder
                const float green[3] = { 0.f, 1.f, 0.f };
                material = context.addMaterial(makeLambertian(green, kBlack));
            }
            appendMesh(scene, parsePly(context.resolve(model["filename"].asString()), transform, material), medium);
        } else if (type == "quad") {
            const Transform transform = parseTransform(model["transform"]);
            const int material = requireMaterial(parseMaterial(model["bsdf"], context), "quad");
            bool zUp = false;
            if (model["upAxis"].isString()) {
                const std::string axis = model["upAxis"].asString();
                if (axis == "z") { zUp = true; }
                else if (axis == "y") { zUp = false; }
                else { throw SceneLoadError("Unsupported axis: " + axis); }
            }
            appendMesh(scene, makeQuad(transform, material, zUp), -1);
        } else if (type == "sphere") {
            PathedSphere sphere;
            sphere.material = requireMaterial(parseMaterial(model["bsdf"], context), "sphere");
            parseTriple(model["center"], sphere.center_sample, "sphere center");
            sphere.radius = parseFloat(model["radius"], "sphere radius");
            const Transform transform = parseTransform(model["transform"]);
            transform.matrix.applyPoint(sphere.center_sample, sphere.center_world);

            PathedGeom geom;
            geom.type = PATHED_GEOM_SPHERE;
            geom.first = (int32_t)scene.spheres.size();
            geom.count = 1;
            geom.medium = medium;
            scene.geoms.push_back(geom);
            scene.spheres.push_back(sphere);
        } else if (type == "instance" || type == "instanced" || type == "pbrt-curve" || type == "b-spline") {
            throw SceneLoadError("Unsupported model type (outside hot-path scope, SURVEY.md §2 #3/#16): " + type);
        }
        // unknown types are ignored, as the reference's if-chain does
    }
}

}  // namespace

PathedMaterial makeLambertian(const float diffuse[3], const float emit[3])
{
    PathedMaterial material = blankMaterial(PATHED_MAT_LAMBERTIAN);
    for (int i = 0; i < 3; i++) {
        material.diffuse[i] = diffuse[i];
        material.emit[i] = emit[i];
    }
    return material;
}

PathedSceneDesc FlatScene::desc() const
{
    PathedSceneDesc d;
    std::memset(&d, 0, sizeof d);
    d.abi_version = PATHED_ABI_VERSION;
    d.camera = camera;
    d.n_vertices = (uint32_t)(positions.size() / 3);
    d.positions = positions.data();
    d.normals = normals.data();
    d.uvs = uvs.data();
    d.n_triangles = (uint32_t)(indices.size() / 3);
    d.indices = indices.data();
    d.tri_material = triMaterial.data();
    d.n_spheres = (uint32_t)spheres.size();
    d.spheres = spheres.data();
    d.n_geoms = (uint32_t)geoms.size();
    d.geoms = geoms.data();
    d.n_materials = (uint32_t)materials.size();
    d.materials = materials.data();
    if (hasEnv) {
        // keep the pixel pointer valid after the FlatScene has been copied or moved
        const_cast<FlatScene *>(this)->env.rgba = envData.data();
    }
    d.env = hasEnv ? &env : nullptr;
    for (size_t t = 0; t < textures.size(); t++) { const_cast<FlatScene *>(this)->textures[t].rgb = textureData[t].data(); }
    d.n_textures = (uint32_t)textures.size();
    d.textures = textures.data();
    d.n_media = (uint32_t)media.size();
    d.media = media.data();
    return d;
}

FlatScene loadScene(
    const std::string &scenePath,
    int width, int height,
    const std::string &assetRoot
) {
    FlatScene scene;
    LoaderContext context;
    context.scene = &scene;
    context.assetRoot = assetRoot;

    Json json;
    try {
        json = Json::parseFile(context.resolve(scenePath));
    } catch (const JsonError &error) {
        throw SceneLoadError(error.what());
    }

    // camera: scene_parser.cpp:146-158
    const Json &sensor = json["sensor"];
    const float fovDegrees = parseFloat(sensor["fov"], "sensor.fov");
    std::memset(&scene.camera, 0, sizeof scene.camera);
    parseTriple(sensor["lookAt"]["origin"], scene.camera.origin, "lookAt.origin");
    parseTriple(sensor["lookAt"]["target"], scene.camera.target, "lookAt.target");
    parseTriple(sensor["lookAt"]["up"], scene.camera.up, "lookAt.up");
    // `fov / 180.f * M_PI`: float division, double multiply, narrowed by the Camera ctor
    scene.camera.vertical_fov = (float)((double)(fovDegrees / 180.f) * M_PI);
    scene.camera.width = width;
    scene.camera.height = height;
    scene.camera.flip_handedness = parseBool(sensor["flipHandedness"], false) ? 1 : 0;

    // named materials: scene_parser.cpp:555-572
    const Json &materials = json["materials"];
    if (materials.isArray()) {
        for (const Json &materialJson : materials.elements()) {
            const std::string name = materialJson["name"].asString();
            context.materialLookup[name] = parseMaterial(materialJson, context);
        }
    }

    // media: scene_parser.cpp:202-229 (homogeneous only; voxel grids are outside the scope)
    const Json &media = json["media"];
    if (media.isArray()) {
        for (const Json &mediumJson : media.elements()) {
            const std::string kind = mediumJson["type"].isString() ? mediumJson["type"].asString() : "";
            if (kind == "heterogeneous") { throw SceneLoadError("Unsupported: heterogeneous (voxel) media are outside the hot-path scope (SURVEY.md §2)"); }
            if (kind != "homogeneous") { continue; }   // the reference ignores unknown kinds
            PathedMedium medium;
            parseColor(mediumJson["sigma_t"], kBlack, medium.sigma_t);
            parseColor(mediumJson["sigma_s"], kBlack, medium.sigma_s);
            if (!mediumJson["name"].isString()) { throw SceneLoadError("media: a medium needs a string \"name\""); }
            context.mediumLookup[mediumJson["name"].asString()] = (int)scene.media.size();
            scene.media.push_back(medium);
        }
    }

    parseModels(json["models"], context);

    // environment light: scene_parser.cpp:540-553
    const Json &envJson = json["environmentLight"];
    if (envJson.isObject()) {
        int envWidth = 0, envHeight = 0;
        std::string error;
        const std::string path = context.resolve(envJson["filename"].asString());
        if (!readExrRGBA(path, &envWidth, &envHeight, &scene.envData, &error)) {
            throw SceneLoadError(error);
        }
        const Transform transform = parseTransform(envJson["transform"]);
        scene.hasEnv = true;
        scene.env.width = envWidth;
        scene.env.height = envHeight;
        scene.env.rgba = scene.envData.data();
        scene.env.scale = parseFloatDefault(envJson["scale"], 1.f);
        transform.matrix.toArray(scene.env.map_to_world);
        transform.inverse.toArray(scene.env.world_to_map);
    }

    return scene;
}

std::vector<std::string> tokenizeLine(const std::string &line) { return tokenize(line); }
std::string leftTrim(const std::string &token) { return lTrim(token); }
std::vector<MtlMaterial> parseMtlFile(const std::string &path) { return readMtlFile(path); }

}  // namespace pathed
