// Loader.h -- Mitsuba-XML + OBJ scene loader (kept entry point of the reference,
// S/engine/Loader.h:29: Scene loadScene(Engine&, Renderer&, const std::string& path)).
// Engine/Renderer only supplied the asset directory and GPU mesh uploads; here the
// asset directory is a plain argument and meshes stay on the host.
#pragma once
#include <string>

#include "Scene.h"

namespace GPUSpectral {

// dormantFeatures = false (default): the reference's behaviour -- a textured reflectance falls back to the colour
// default and a top-level emitter is ignored, each with a warning (Loader.cpp:122-143,338-346 are commented out there).
// true: the branches the reference left dormant are live -- `bitmap` textures (PNG / JPEG, Image.h) and `checkerboard`
// textures on diffuse / roughplastic / roughconductor, and an `envmap` emitter (PFM / .hdr) -- see
// include/gpuspectral_pt.h for what the renderer does with them.
struct LoadOptions {
  bool dormantFeatures = false;
  bool srgbTextures = true;
};

// assetDir: where rect.obj / box.obj / disk.obj live (Engine::assetPath); "" = the
// directory shipped with this library (gpuspectral_amd/assets).
Scene loadScene(const std::string& path, const std::string& assetDir = "", const LoadOptions& options = LoadOptions());

// Throws std::runtime_error like the reference on unreadable files.
MeshPtr loadMesh(const std::string& objPath, uint32_t id);

void setDefaultAssetDir(const std::string& dir);

}  // namespace GPUSpectral
