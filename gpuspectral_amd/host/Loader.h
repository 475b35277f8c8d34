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
  // `<shape type="disk">` maps to assets/disk.obj in the reference (Loader.cpp:276), a file its checkout does
  // not hold, and `sphere` falls through to an empty file name (:278): both shapes are skipped with a warning by default,
  // as they are there -- which leaves `living-room` without its only emitters and `staircase2` without five of its lights.
  // builtinShapes = true builds them instead: Mitsuba's unit disk (z = 0, radius 1, normal +z; 64 fan triangles) and unit
  // sphere (octahedron subdivided three times, 512 triangles, smooth normals), tessellated with float32 + * / sqrt only
  // (no libm: oracle/mitsuba_loader.py builds the same floats).  `to_world` and `center` as for every shape
  // (Loader.cpp:284-293); `radius` -- which the reference never reads -- scales the sphere.
  bool builtinShapes = false;
};

// assetDir: where rect.obj / box.obj / disk.obj live (Engine::assetPath); "" = the
// directory shipped with this library (gpuspectral_amd/assets).
Scene loadScene(const std::string& path, const std::string& assetDir = "", const LoadOptions& options = LoadOptions());

// Throws std::runtime_error like the reference on unreadable files.
MeshPtr loadMesh(const std::string& objPath, uint32_t id);

void setDefaultAssetDir(const std::string& dir);

}  // namespace GPUSpectral
