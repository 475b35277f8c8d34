// PathTracer.cpp -- see PathTracer.h.  Reference: S/renderer/PathTracer.cpp:5-93.
#include "PathTracer.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>

namespace GPUSpectral {

// ---- SceneTracker -----------------------------------------------------------------------------------------------------
namespace {
template <class T>
void appendBytes(std::vector<unsigned char>& out, const std::vector<T>& v) {
  const uint64_t n = v.size();
  const unsigned char* c = reinterpret_cast<const unsigned char*>(&n);
  out.insert(out.end(), c, c + sizeof(n));
  const unsigned char* p = reinterpret_cast<const unsigned char*>(v.data());
  out.insert(out.end(), p, p + n * sizeof(T));
}
// the bytes PathTracer.cpp:74-87 uploads every frame: BSDF.inc's eight arrays, then the lights
void tableBytes(const Scene& s, std::vector<unsigned char>& out) {
  out.clear();
  appendBytes(out, s.diffuseBSDFs);
  appendBytes(out, s.smoothDielectricBSDFs);
  appendBytes(out, s.smoothConductorBSDFs);
  appendBytes(out, s.smoothPlasticBSDFs);
  appendBytes(out, s.roughConductorBSDFs);
  appendBytes(out, s.smoothFloorBSDFs);
  appendBytes(out, s.roughFloorBSDFs);
  appendBytes(out, s.roughPlasticBSDFs);
  appendBytes(out, s.triangleLights);
}
void assetKey(const Scene& s, std::vector<uint64_t>& out) {
  out.clear();
  out.push_back(s.textures.size());
  for (const Texture& t : s.textures) out.push_back(((uint64_t)t.width << 32) | t.height), out.push_back(t.texels.size());
  out.push_back(s.srgbTextures ? 1 : 0);
  out.push_back(s.hasEnvMap ? 1 : 0);
  if (s.hasEnvMap) {
    out.push_back(((uint64_t)s.envMap.width << 32) | s.envMap.height);
    for (int c = 0; c < 4; ++c)
      for (int r = 0; r < 4; ++r) {
        uint32_t bits;
        std::memcpy(&bits, &s.envMap.transform[c][r], 4);
        out.push_back(bits);
      }
  }
}
gsp_camera cameraOf(const Scene& s) {
  gsp_camera c{};
  std::memcpy(c.to_world, s.camera.getToWorld().data(), sizeof(c.to_world));
  c.fov = s.camera.getFov();
  return c;
}
}  // namespace

unsigned SceneTracker::diff(const Scene& scene, std::vector<gsp_instance>& inst) const {
  flattenInstances(scene, inst);
  if (!valid || scene.renderObjects.size() != meshes.size()) return Everything;
  for (size_t i = 0; i < meshes.size(); ++i)
    if (scene.renderObjects[i].mesh != meshes[i]) return Everything;  // another mesh = another BLAS (Renderer.cpp:122-131)
  std::vector<uint64_t> a;
  assetKey(scene, a);
  if (a != assets) return Everything;
  unsigned change = None;
  // (same meshes in the same order => same vertex ranges; what may differ is transform / emission / bsdf / twofaced)
  if (inst.size() != instances.size() ||
      (!inst.empty() && std::memcmp(inst.data(), instances.data(), inst.size() * sizeof(gsp_instance)) != 0))
    change |= InstancesChanged;
  std::vector<unsigned char> t;
  tableBytes(scene, t);
  if (t != tables) change |= TablesChanged;
  const gsp_camera cam = cameraOf(scene);
  if (std::memcmp(&cam, &camera, sizeof(cam)) != 0) change |= CameraChanged;
  return change;
}

void SceneTracker::remember(const Scene& scene, const std::vector<gsp_instance>& inst) {
  meshes.clear();
  for (const RenderObject& o : scene.renderObjects) meshes.push_back(o.mesh);
  instances = inst;
  tableBytes(scene, tables);
  assetKey(scene, assets);
  camera = cameraOf(scene);
  valid = true;
}

PathTracer::PathTracer(uint32_t width, uint32_t height, int device, const std::vector<uint32_t>& pixelIds,
                       const gsp_ctx_options* opt)
    : width(width), height(height), device(device), pixelIds(pixelIds) {
  gsp_default_render_params(&params);
  gsp_default_ctx_options(&options);
  if (opt) options = *opt;
  setup();
}

PathTracer::~PathTracer() { gsp_ctx_destroy(ctx); }

void PathTracer::check(int rc, const char* what) {
  if (rc != GSP_OK) throw std::runtime_error(std::string(what) + ": " + gsp_last_error(ctx));
}

// PathTracer.cpp:5-7
void PathTracer::setup() {
  // gsp_stats carries no struct_size (the library fills it at ITS size): the header this file was compiled against must be the library's
  if (gsp_abi_version() != GSP_ABI_VERSION)
    throw std::runtime_error("libgpuspectral_pt.so has ABI " + std::to_string(gsp_abi_version()) + ", this host was built against ABI " +
                             std::to_string(GSP_ABI_VERSION));
  int rc = gsp_ctx_create_ex(device, &options, &ctx);
  if (rc != GSP_OK) throw std::runtime_error(std::string("gsp_ctx_create_ex: ") + gsp_last_error(nullptr));
  check(gsp_frame_begin(ctx, width, height, pixelIds.empty() ? nullptr : pixelIds.data(), pixelIds.size()),
        "gsp_frame_begin");
}

void PathTracer::reset() {
  timestamp = 0;
  check(gsp_frame_begin(ctx, width, height, pixelIds.empty() ? nullptr : pixelIds.data(), pixelIds.size()),
        "gsp_frame_begin");
}

// PathTracer.cpp:10-19,58-93.  The reference rebuilds the TLAS and re-uploads every table, the instance records and the
// camera each frame; here the scene is compared BY VALUE with what the device holds (SceneTracker) and only what changed
// is sent: nothing in the common case, the camera alone for a moved camera, the tables for an edited BSDF or light, a
// re-bake from the resident meshes + refit of the tree (a rebuild when it has degraded) for an edited transform or material, everything for another object list.
// Tables go before instances: an instance may name a BSDF the new tables add.
void PathTracer::prepareScene(const Scene& scene) {
  std::vector<gsp_instance> inst;
  const unsigned change = tracker.diff(scene, inst);
  if (change == SceneTracker::None) return;
  tracker.forget();  // (a failed call below leaves nothing remembered: the next pass uploads everything)
  if (change & SceneTracker::Everything) {
    FlatScene flat;
    flattenScene(scene, flat);
    check(gsp_upload_scene(ctx, &flat.desc), "gsp_upload_scene");
  } else {
    gsp_scene_desc d{};
    describeTables(scene, d);
    // an instance edit that names a BSDF the OLD tables lack needs the new tables first, and tables that drop a record an
    // OLD instance names need the new instances first: send the pair in the order that keeps every handle in range, or
    // (both directions at once) fall back to the full upload
    const bool both = (change & SceneTracker::TablesChanged) && (change & SceneTracker::InstancesChanged);
    int rcT = GSP_OK;
    if (change & SceneTracker::TablesChanged) rcT = gsp_update_tables(ctx, &d);
    if (rcT != GSP_OK && !both) check(rcT, "gsp_update_tables");
    if (change & SceneTracker::InstancesChanged) {
      int rcI = gsp_update_instances(ctx, inst.data(), (uint32_t)inst.size());
      if (rcI != GSP_OK && rcT == GSP_OK) check(rcI, "gsp_update_instances");
      if (rcI == GSP_OK && rcT != GSP_OK) check(gsp_update_tables(ctx, &d), "gsp_update_tables");
      if (rcI != GSP_OK && rcT != GSP_OK) {
        FlatScene flat;
        flattenScene(scene, flat);
        check(gsp_upload_scene(ctx, &flat.desc), "gsp_upload_scene");
      }
    }
    if (change & SceneTracker::CameraChanged) check(gsp_update_camera(ctx, &d.camera), "gsp_update_camera");
  }
  tracker.remember(scene, inst);
}

void PathTracer::render(const Scene& scene, uint32_t spp) {
  prepareScene(scene);
  gsp_render_params p = params;
  p.spp = spp;
  p.first_timestamp = (uint32_t)timestamp;  // renderState.params.timestamp, PathTracer.cpp:91
  check(gsp_render(ctx, &p), "gsp_render");
  timestamp += (int)spp;                    // PathTracer.cpp:92
}

// PathTracer.cpp:9-56: one traceRays(W, H) = one sample per pixel
void PathTracer::createRenderPass(const Scene& scene) { render(scene, 1); }

std::vector<float> PathTracer::download() {
  std::vector<float> out((size_t)width * height * 4);
  check(gsp_download(ctx, out.data()), "gsp_download");
  return out;
}

std::vector<float> PathTracer::peek(uint32_t* samplesFolded) {
  std::vector<float> out((pixelIds.empty() ? (size_t)width * height : pixelIds.size()) * 4);
  check(gsp_peek(ctx, out.data(), samplesFolded), "gsp_peek");
  return out;
}

void PathTracer::peekToDevice(void* deviceDst, uint64_t bytes, uint32_t* samplesFolded) {
  check(gsp_peek_to_device(ctx, deviceDst, bytes, samplesFolded), "gsp_peek_to_device");
}

gsp_stats PathTracer::stats() {
  gsp_stats s;
  check(gsp_get_stats(ctx, &s), "gsp_get_stats");
  return s;
}

// ---- several GPUs ---------------------------------------------------------------------------------------------
MultiGpuPathTracer::MultiGpuPathTracer(uint32_t width, uint32_t height, const std::vector<int>& devices, const gsp_ctx_options* opt)
    : width(width), height(height), devices(devices) {
  gsp_default_render_params(&params);
  int rc = gsp_multi_create_ex(devices.data(), (int)devices.size(), opt, &multi);
  if (rc != GSP_OK) throw std::runtime_error(std::string("gsp_multi_create_ex: ") + gsp_multi_last_error(nullptr));
  check(gsp_multi_frame_begin(multi, width, height), "gsp_multi_frame_begin");
}

MultiGpuPathTracer::~MultiGpuPathTracer() { gsp_multi_destroy(multi); }

void MultiGpuPathTracer::check(int rc, const char* what) {
  if (rc != GSP_OK) throw std::runtime_error(std::string(what) + ": " + gsp_multi_last_error(multi));
}

void MultiGpuPathTracer::reset() {
  timestamp = 0;
  check(gsp_multi_frame_begin(multi, width, height), "gsp_multi_frame_begin");
}

void MultiGpuPathTracer::prepareScene(const Scene& scene) {
  std::vector<gsp_instance> inst;
  const unsigned change = tracker.diff(scene, inst);
  if (change == SceneTracker::None) return;
  tracker.forget();
  // (same order as PathTracer::prepareScene; the handle-range corner cases take the full upload here)
  bool full = (change & SceneTracker::Everything) != 0;
  if (!full) {
    gsp_scene_desc d{};
    describeTables(scene, d);
    if ((change & SceneTracker::TablesChanged) && gsp_multi_update_tables(multi, &d) != GSP_OK) full = true;
    if (!full && (change & SceneTracker::InstancesChanged) &&
        gsp_multi_update_instances(multi, inst.data(), (uint32_t)inst.size()) != GSP_OK)
      full = true;
    if (!full && (change & SceneTracker::CameraChanged)) check(gsp_multi_update_camera(multi, &d.camera), "gsp_multi_update_camera");
  }
  if (full) {
    FlatScene flat;
    flattenScene(scene, flat);
    check(gsp_multi_upload_scene(multi, &flat.desc), "gsp_multi_upload_scene");
  }
  tracker.remember(scene, inst);
}

void MultiGpuPathTracer::render(const Scene& scene, uint32_t spp) {
  prepareScene(scene);
  gsp_render_params p = params;
  p.spp = spp;
  p.first_timestamp = (uint32_t)timestamp;
  check(gsp_multi_render(multi, &p), "gsp_multi_render");
  timestamp += (int)spp;
}

void MultiGpuPathTracer::createRenderPass(const Scene& scene) { render(scene, 1); }

std::vector<float> MultiGpuPathTracer::download() {
  std::vector<float> out((size_t)width * height * 4);
  check(gsp_multi_download(multi, out.data()), "gsp_multi_download");
  return out;
}

gsp_stats MultiGpuPathTracer::stats(std::vector<gsp_stats>* perShare) {
  gsp_stats s;
  if (perShare) perShare->resize(devices.size());
  check(gsp_multi_get_stats(multi, &s, perShare ? perShare->data() : nullptr), "gsp_multi_get_stats");
  return s;
}

void writePfm(const std::string& path, const float* rgba, uint32_t width, uint32_t height) {
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) throw std::runtime_error("cannot write " + path);
  std::fprintf(f, "PF\n%u %u\n-1.0\n", width, height);
  std::vector<float> row(3ull * width);
  for (uint32_t y = 0; y < height; ++y) {
    const float* src = rgba + 4ull * (height - 1 - y) * width;
    for (uint32_t x = 0; x < width; ++x) {
      row[3 * x] = src[4 * x];
      row[3 * x + 1] = src[4 * x + 1];
      row[3 * x + 2] = src[4 * x + 2];
    }
    std::fwrite(row.data(), sizeof(float), row.size(), f);
  }
  std::fclose(f);
}

// ACESFilm, S/assets/shaders/common.glsl:74-82
static inline float acesFilm(float x) {
  const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
  float y = (x * (a * x + b)) / (x * (c * x + d) + e);
  return y < 0.0f ? 0.0f : (y > 1.0f ? 1.0f : y);
}

void toneMapToRgb8(const float* rgba, uint32_t width, uint32_t height, bool toneMap, std::vector<uint8_t>& rgb8) {
  rgb8.resize(3ull * width * height);
  for (size_t i = 0; i < (size_t)width * height; ++i)
    for (int c = 0; c < 3; ++c) {
      float v = rgba[4 * i + c];
      if (!(v == v)) v = 0.0f;
      v = toneMap ? acesFilm(v) : (v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v));
      v = std::pow(v, 1.0f / 2.2f);
      rgb8[3 * i + c] = (uint8_t)(v * 255.0f + 0.5f);
    }
}

void writePpm(const std::string& path, const float* rgba, uint32_t width, uint32_t height, bool toneMap) {
  std::vector<uint8_t> rgb;
  toneMapToRgb8(rgba, width, height, toneMap, rgb);
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) throw std::runtime_error("cannot write " + path);
  std::fprintf(f, "P6\n%u %u\n255\n", width, height);
  std::fwrite(rgb.data(), 1, rgb.size(), f);
  std::fclose(f);
}

}  // namespace GPUSpectral
