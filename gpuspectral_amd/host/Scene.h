// Scene.h -- host-side scene data model of the path tracer.
//
// Mirrors the reference's S/renderer/Scene.h:1-186, Camera.h/.cpp and Mesh.h
// (same type and member names, same add<X>BSDF / addMaterial / addRenderObject
// API) minus everything that was a Vulkan handle: a Mesh is the CPU copy of the
// de-indexed vertices the reference keeps beside its GPU buffers
// (S/renderer/Mesh.h:59-60).  The BSDF records ARE the C-ABI PODs, whose layouts
// equal the reference structs (Scene.h:29-81), so flattening copies no fields.
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../../include/gpuspectral_pt.h"

namespace GPUSpectral {

struct vec2 {
  float x = 0, y = 0;
};
struct vec3 {
  float x = 0, y = 0, z = 0;
};
struct vec4 {
  float x = 0, y = 0, z = 0, w = 0;
};
// glm::mat4 stand-in: column-major, m[c][r]
struct mat4 {
  std::array<std::array<float, 4>, 4> m{};
  static mat4 identity() {
    mat4 r;
    for (int i = 0; i < 4; ++i) r.m[i][i] = 1.0f;
    return r;
  }
  std::array<float, 4>& operator[](int c) { return m[c]; }
  const std::array<float, 4>& operator[](int c) const { return m[c]; }
  const float* data() const { return &m[0][0]; }
};
// glm::make_mat4(ptr): 16 floats read in memory (column-major) order
inline mat4 make_mat4(const float* p) {
  mat4 r;
  for (int c = 0; c < 4; ++c)
    for (int k = 0; k < 4; ++k) r.m[c][k] = p[4 * c + k];
  return r;
}
inline mat4 transpose(const mat4& a) {
  mat4 r;
  for (int c = 0; c < 4; ++c)
    for (int k = 0; k < 4; ++k) r.m[c][k] = a.m[k][c];
  return r;
}
// glm::mat4 * glm::vec4 in glm's association: (m0*x + m1*y) + (m2*z + m3*w)
inline vec4 operator*(const mat4& m, const vec4& v) {
  vec4 r;
  float* o = &r.x;
  for (int k = 0; k < 4; ++k) o[k] = (m[0][k] * v.x + m[1][k] * v.y) + (m[2][k] * v.z + m[3][k] * v.w);
  return r;
}

class Mesh {
 public:
  struct Vertex {
    vec3 pos;
    vec3 normal;
    vec2 uv;
  };
  Mesh(uint32_t id, std::vector<Vertex> vertices) : id(id), vertices(std::move(vertices)) {}
  uint32_t getID() const noexcept { return id; }
  const std::vector<Vertex>& getVertices() const noexcept { return vertices; }

 private:
  uint32_t id;
  std::vector<Vertex> vertices;
};
using MeshPtr = std::shared_ptr<Mesh>;
using MaterialHandle = int;

struct RenderObject {
  mat4 transform = mat4::identity();
  MeshPtr mesh;
  MaterialHandle material = 0;
};

enum BSDFType : uint16_t {
  BSDF_DIFFUSE = GSP_BSDF_DIFFUSE,
  BSDF_SMOOTH_DIELECTRIC = GSP_BSDF_SMOOTH_DIELECTRIC,
  BSDF_SMOOTH_CONDUCTOR = GSP_BSDF_SMOOTH_CONDUCTOR,
  BSDF_SMOOTH_PLASTIC = GSP_BSDF_SMOOTH_PLASTIC,
  BSDF_ROUGH_CONDUCTOR = GSP_BSDF_ROUGH_CONDUCTOR,
  BSDF_SMOOTH_FLOOR = GSP_BSDF_SMOOTH_FLOOR,
  BSDF_ROUGH_FLOOR = GSP_BSDF_ROUGH_FLOOR,
  BSDF_ROUGH_PLASTIC = GSP_BSDF_ROUGH_PLASTIC,
};

using DiffuseBSDF = gsp_diffuse_bsdf;
using SmoothDielectricBSDF = gsp_smooth_dielectric_bsdf;
using SmoothConductorBSDF = gsp_smooth_conductor_bsdf;
using SmoothPlasticBSDF = gsp_smooth_plastic_bsdf;
using RoughConductorBSDF = gsp_rough_conductor_bsdf;
using SmoothFloorBSDF = gsp_smooth_floor_bsdf;
using RoughFloorBSDF = gsp_rough_floor_bsdf;
using RoughPlasticBSDF = gsp_rough_plastic_bsdf;
using TriangleLight = gsp_triangle_light;

struct BSDFHandle {
  BSDFHandle() = default;
  BSDFHandle(BSDFType type, uint32_t index) : handle(GSP_BSDF_HANDLE(type, index)) {}
  BSDFType type() const { return static_cast<BSDFType>((handle >> 16) & 0xffff); }
  uint32_t index() const { return handle & 0xFFFF; }
  uint32_t handle = 0;
};

struct Material {
  vec3 emission{};
  bool twofaced = false;
  bool facenormals = false;
  BSDFHandle bsdf;
};

// ---- dormant features of the reference (include/gpuspectral_pt.h "Dormant features") ----
// The reference keeps a Handle<HwTexture> where this keeps the decoded image (loadTexture, Loader.cpp:66-86), and its
// Envmap is {texture, transform} (Scene.h:116-119).
struct Texture {
  uint32_t width = 0, height = 0;
  std::vector<uint32_t> texels;  // RGBA8, row 0 = bottom image row
};
struct Envmap {
  uint32_t width = 0, height = 0;
  std::vector<float> texels;  // RGBA32F, row 0 = bottom image row
  mat4 transform = mat4::identity();  // to_world of the emitter (envMapTransform, Loader.cpp:343-345)
};

class Camera {
 public:
  void setToWorld(const mat4& m) { toWorld = m; }
  void setFov(float fovY) { fov = fovY; }
  vec3 getPosition() const noexcept { return vec3{toWorld[3][0], toWorld[3][1], toWorld[3][2]}; }
  const mat4& getToWorld() const noexcept { return toWorld; }
  float getFov() const noexcept { return fov; }

 private:
  mat4 toWorld = mat4::identity();
  float fov = 0.5f;
};

struct Scene {
  MaterialHandle addMaterial(const Material& material) {
    materials.push_back(material);
    return (MaterialHandle)materials.size() - 1;
  }
  Material& getMaterial(MaterialHandle h) { return materials[h]; }
  const Material& getMaterial(MaterialHandle h) const { return materials[h]; }
  void addRenderObject(const RenderObject& object) { renderObjects.push_back(object); }
  void addTriangleLight(const TriangleLight& light) { triangleLights.push_back(light); }

#define GSP_BSDF_ADDER(NAME, FIELD, TYPE)                       \
  BSDFHandle add##NAME(const NAME& bsdf) {                      \
    BSDFHandle h{BSDF_##TYPE, (uint32_t)FIELD##s.size()};       \
    FIELD##s.push_back(bsdf);                                   \
    return h;                                                   \
  }
  GSP_BSDF_ADDER(DiffuseBSDF, diffuseBSDF, DIFFUSE)
  GSP_BSDF_ADDER(SmoothDielectricBSDF, smoothDielectricBSDF, SMOOTH_DIELECTRIC)
  GSP_BSDF_ADDER(SmoothConductorBSDF, smoothConductorBSDF, SMOOTH_CONDUCTOR)
  GSP_BSDF_ADDER(SmoothPlasticBSDF, smoothPlasticBSDF, SMOOTH_PLASTIC)
  GSP_BSDF_ADDER(RoughConductorBSDF, roughConductorBSDF, ROUGH_CONDUCTOR)
  GSP_BSDF_ADDER(SmoothFloorBSDF, smoothFloorBSDF, SMOOTH_FLOOR)
  GSP_BSDF_ADDER(RoughFloorBSDF, roughFloorBSDF, ROUGH_FLOOR)
  GSP_BSDF_ADDER(RoughPlasticBSDF, roughPlasticBSDF, ROUGH_PLASTIC)
#undef GSP_BSDF_ADDER

  Camera camera;
  std::vector<RenderObject> renderObjects;
  std::vector<Material> materials;
  std::vector<TriangleLight> triangleLights;
  std::vector<DiffuseBSDF> diffuseBSDFs;
  std::vector<SmoothDielectricBSDF> smoothDielectricBSDFs;
  std::vector<SmoothConductorBSDF> smoothConductorBSDFs;
  std::vector<SmoothPlasticBSDF> smoothPlasticBSDFs;
  std::vector<RoughConductorBSDF> roughConductorBSDFs;
  std::vector<SmoothFloorBSDF> smoothFloorBSDFs;
  std::vector<RoughFloorBSDF> roughFloorBSDFs;
  std::vector<RoughPlasticBSDF> roughPlasticBSDFs;
  std::vector<std::string> warnings;  // what the loader skipped (the reference printed these to stdout)

  // dormant features: filled only by loadScene(..., LoadOptions{.dormantFeatures = true}); a BSDF record's hasTexture is
  // 1 + the index into `textures`
  uint32_t addTexture(Texture t) {
    textures.push_back(std::move(t));
    return (uint32_t)textures.size();
  }
  std::vector<Texture> textures;
  bool srgbTextures = true;  // 8-bit bitmaps hold sRGB-encoded values (what Mitsuba / Tungsten assume); false = value / 255
  bool hasEnvMap = false;    // std::optional<Envmap> envMap (Scene.h:182)
  Envmap envMap;
};

// Scene flattened to the arrays gsp_scene_desc points at (owns the storage).
struct FlatScene {
  std::vector<gsp_instance> instances;
  std::vector<float> positions, normals;
  std::vector<float> uvs, texelDecode, envTexels;  // dormant features
  std::vector<gsp_texture> textures;
  std::vector<uint32_t> texels;
  gsp_scene_desc desc{};
};
// 4x4 inverse (cofactors, float); used for Envmap::transform -> gsp_envmap.to_local
mat4 inverse(const mat4& m);
// Host half of PathTracer::prepareScene (S/renderer/PathTracer.cpp:58-93): instance table in
// renderObjects order, shared meshes stored once.
void flattenScene(const Scene& scene, FlatScene& out);
// The instance table alone (no vertex copies): what flattenScene would put in FlatScene::instances.  This is the part of
// the scene the reference rebuilds on every createRenderPass (PathTracer.cpp:10-19,58-70).
void flattenInstances(const Scene& scene, std::vector<gsp_instance>& out);
// The table part of a gsp_scene_desc (eight BSDF arrays + lights, pointing INTO `scene`) and its camera.
void describeTables(const Scene& scene, gsp_scene_desc& d);

}  // namespace GPUSpectral
