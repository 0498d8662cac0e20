// C entry points of libpathed_host.so for the Python test / bench harness (ctypes).
// These are plumbing around the C++ host classes; the product boundary is
// include/pathed_hip.h.
#include "scene_loader.h"

#include <cstring>
#include <string>

namespace {
thread_local std::string g_hostError;

struct LoadedScene {
    pathed::FlatScene scene;
    PathedSceneDesc desc;
};
}  // namespace

extern "C" {

const char *pathed_host_last_error(void)
{
    return g_hostError.c_str();
}

void *pathed_host_load_scene(const char *scenePath, int width, int height, const char *assetRoot)
{
    try {
        LoadedScene *loaded = new LoadedScene();
        loaded->scene = pathed::loadScene(scenePath, width, height, assetRoot ? assetRoot : "");
        loaded->desc = loaded->scene.desc();
        return loaded;
    } catch (const std::exception &error) {
        g_hostError = error.what();
        return nullptr;
    }
}

const PathedSceneDesc *pathed_host_scene_desc(void *handle)
{
    if (!handle) { return nullptr; }
    return &((LoadedScene *)handle)->desc;
}

void pathed_host_free_scene(void *handle)
{
    delete (LoadedScene *)handle;
}

}  // extern "C"
