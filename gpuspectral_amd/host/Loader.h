// Loader.h -- Mitsuba-XML + OBJ scene loader (kept entry point of the reference,
// S/engine/Loader.h:29: Scene loadScene(Engine&, Renderer&, const std::string& path)).
// Engine/Renderer only supplied the asset directory and GPU mesh uploads; here the
// asset directory is a plain argument and meshes stay on the host.
#pragma once
#include <string>

#include "Scene.h"

namespace GPUSpectral {

// assetDir: where rect.obj / box.obj / disk.obj live (Engine::assetPath); "" = the
// directory shipped with this library (gpuspectral_amd/assets).
Scene loadScene(const std::string& path, const std::string& assetDir = "");

// Throws std::runtime_error like the reference on unreadable files.
MeshPtr loadMesh(const std::string& objPath, uint32_t id);

void setDefaultAssetDir(const std::string& dir);

}  // namespace GPUSpectral
