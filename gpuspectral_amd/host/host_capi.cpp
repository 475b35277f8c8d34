// host_capi.cpp -- C wrappers over the C++ host layer (Scene / loadScene / PathTracer) so
// that the Python tests can drive exactly the classes a C++ caller would use.
#include <cstring>
#include <exception>
#include <stdexcept>
#include <string>

#include "Image.h"
#include "Loader.h"
#include "PathTracer.h"

using namespace GPUSpectral;

namespace {
thread_local std::string g_err;
struct SceneBox {
  Scene scene;
  FlatScene flat;
};
template <class F>
int guard(F&& f) {
  try {
    f();
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return 1;
  }
}
}  // namespace

extern "C" {

const char* gsph_last_error(void) { return g_err.c_str(); }

void* gsph_load_scene(const char* path, const char* asset_dir) {
  SceneBox* b = nullptr;
  int rc = guard([&] {
    b = new SceneBox();
    b->scene = loadScene(path, asset_dir ? asset_dir : "");
    flattenScene(b->scene, b->flat);
  });
  if (rc) {
    delete b;
    return nullptr;
  }
  return b;
}
// dormant != 0: loadScene with LoadOptions{dormantFeatures = true, srgbTextures = srgb != 0}
void* gsph_load_scene_ex(const char* path, const char* asset_dir, int dormant, int srgb) {
  SceneBox* b = nullptr;
  int rc = guard([&] {
    b = new SceneBox();
    LoadOptions opt;
    opt.dormantFeatures = dormant != 0;
    opt.srgbTextures = srgb != 0;
    b->scene = loadScene(path, asset_dir ? asset_dir : "", opt);
    flattenScene(b->scene, b->flat);
  });
  if (rc) {
    delete b;
    return nullptr;
  }
  return b;
}
// flags: 1 = LoadOptions::dormantFeatures, 2 = srgbTextures, 4 = builtinShapes (disk / sphere)
void* gsph_load_scene_opts(const char* path, const char* asset_dir, unsigned flags) {
  SceneBox* b = nullptr;
  int rc = guard([&] {
    b = new SceneBox();
    LoadOptions opt;
    opt.dormantFeatures = (flags & 1u) != 0;
    opt.srgbTextures = (flags & 2u) != 0;
    opt.builtinShapes = (flags & 4u) != 0;
    b->scene = loadScene(path, asset_dir ? asset_dir : "", opt);
    flattenScene(b->scene, b->flat);
  });
  if (rc) {
    delete b;
    return nullptr;
  }
  return b;
}
void gsph_scene_free(void* s) { delete (SceneBox*)s; }

// Image.h readers.  Two calls: texels == NULL returns the size, the second call fills width * height RGBA8 words
// (floats x 4 for the HDR reader); rows bottom-up.
int gsph_load_bitmap(const char* path, uint32_t* width, uint32_t* height, uint32_t* texels) {
  return guard([&] {
    Image8 img = loadBitmap(path);
    *width = img.width;
    *height = img.height;
    if (texels) std::memcpy(texels, img.texels.data(), img.texels.size() * sizeof(uint32_t));
  });
}
int gsph_load_hdr_bitmap(const char* path, uint32_t* width, uint32_t* height, float* texels) {
  return guard([&] {
    ImageF img = loadHdrBitmap(path);
    *width = img.width;
    *height = img.height;
    if (texels) std::memcpy(texels, img.texels.data(), img.texels.size() * sizeof(float));
  });
}
// The flattened scene (pointers stay valid until gsph_scene_free).
const gsp_scene_desc* gsph_scene_desc(void* s) { return &((SceneBox*)s)->flat.desc; }
uint32_t gsph_scene_num_warnings(void* s) { return (uint32_t)((SceneBox*)s)->scene.warnings.size(); }
const char* gsph_scene_warning(void* s, uint32_t i) { return ((SceneBox*)s)->scene.warnings[i].c_str(); }
uint32_t gsph_scene_num_materials(void* s) { return (uint32_t)((SceneBox*)s)->scene.materials.size(); }

// ---- scene edits between frames (what a host application does to its Scene; the tests drive PathTracer::prepareScene's
// value comparison with them) ----
uint32_t gsph_scene_num_objects(void* s) { return (uint32_t)((SceneBox*)s)->scene.renderObjects.size(); }
int gsph_scene_set_camera(void* s, const float* to_world16, float fov) {
  return guard([&] {
    Scene& sc = ((SceneBox*)s)->scene;
    sc.camera.setToWorld(make_mat4(to_world16));
    sc.camera.setFov(fov);
  });
}
int gsph_scene_set_transform(void* s, uint32_t object, const float* m16) {
  return guard([&] {
    Scene& sc = ((SceneBox*)s)->scene;
    if (object >= sc.renderObjects.size()) throw std::runtime_error("object index out of range");
    sc.renderObjects[object].transform = make_mat4(m16);
  });
}
// the material of render object `object`: emission (rgb or NULL = keep), twofaced (-1 = keep)
int gsph_scene_set_object_material(void* s, uint32_t object, const float* emission3, int twofaced) {
  return guard([&] {
    Scene& sc = ((SceneBox*)s)->scene;
    if (object >= sc.renderObjects.size()) throw std::runtime_error("object index out of range");
    Material& m = sc.getMaterial(sc.renderObjects[object].material);
    if (emission3) m.emission = vec3{emission3[0], emission3[1], emission3[2]};
    if (twofaced >= 0) m.twofaced = twofaced != 0;
  });
}
int gsph_scene_set_diffuse_reflectance(void* s, uint32_t index, const float* rgb) {
  return guard([&] {
    Scene& sc = ((SceneBox*)s)->scene;
    if (index >= sc.diffuseBSDFs.size()) throw std::runtime_error("diffuse BSDF index out of range");
    for (int k = 0; k < 3; ++k) sc.diffuseBSDFs[index].reflectance[k] = rgb[k];
  });
}
// gives render object `object` a NEW rough-conductor BSDF (appended to the table): tables and instances change together
int gsph_scene_make_object_rough_conductor(void* s, uint32_t object, const float* eta3, const float* k3, float alpha) {
  return guard([&] {
    Scene& sc = ((SceneBox*)s)->scene;
    if (object >= sc.renderObjects.size()) throw std::runtime_error("object index out of range");
    RoughConductorBSDF b{};
    for (int c = 0; c < 3; ++c) b.eta[c] = eta3[c], b.k[c] = k3[c], b.reflectance[c] = 1.0f;
    b.alpha = alpha;
    sc.getMaterial(sc.renderObjects[object].material).bsdf = sc.addRoughConductorBSDF(b);
  });
}
int gsph_scene_reflatten(void* s) {
  return guard([&] { flattenScene(((SceneBox*)s)->scene, ((SceneBox*)s)->flat); });
}

// SceneTracker (host/PathTracer.h) without a device: what would PathTracer::prepareScene send?
static SceneTracker g_tracker;
int gsph_tracker_remember(void* s) {
  return guard([&] {
    std::vector<gsp_instance> inst;
    (void)g_tracker.diff(((SceneBox*)s)->scene, inst);
    g_tracker.remember(((SceneBox*)s)->scene, inst);
  });
}
int gsph_tracker_diff(void* s) {
  std::vector<gsp_instance> inst;
  return (int)g_tracker.diff(((SceneBox*)s)->scene, inst);
}
int gsph_tracker_probe(void* uploaded, void* now) {
  SceneTracker t;
  std::vector<gsp_instance> inst;
  (void)t.diff(((SceneBox*)uploaded)->scene, inst);
  t.remember(((SceneBox*)uploaded)->scene, inst);
  return (int)t.diff(((SceneBox*)now)->scene, inst);
}

void* gsph_pathtracer_create(uint32_t width, uint32_t height, int device, const uint32_t* pixel_ids, uint64_t n) {
  PathTracer* pt = nullptr;
  int rc = guard([&] {
    std::vector<uint32_t> ids;
    if (pixel_ids) ids.assign(pixel_ids, pixel_ids + n);
    pt = new PathTracer(width, height, device, ids);
  });
  return rc ? nullptr : pt;
}
void gsph_pathtracer_free(void* pt) { delete (PathTracer*)pt; }
int gsph_pathtracer_create_render_pass(void* pt, void* scene) {
  return guard([&] { ((PathTracer*)pt)->createRenderPass(((SceneBox*)scene)->scene); });
}
int gsph_pathtracer_render(void* pt, void* scene, uint32_t spp) {
  return guard([&] { ((PathTracer*)pt)->render(((SceneBox*)scene)->scene, spp); });
}
int gsph_pathtracer_set_params(void* pt, const gsp_render_params* p) {
  return guard([&] { ((PathTracer*)pt)->params = *p; });
}
int gsph_pathtracer_timestamp(void* pt) { return ((PathTracer*)pt)->getTimestamp(); }
int gsph_pathtracer_reset(void* pt) {
  return guard([&] { ((PathTracer*)pt)->reset(); });
}
int gsph_pathtracer_download(void* pt, float* out, uint64_t count) {
  return guard([&] {
    auto img = ((PathTracer*)pt)->download();
    if (count < img.size()) throw std::runtime_error("output buffer too small");
    std::memcpy(out, img.data(), img.size() * sizeof(float));
  });
}
int gsph_pathtracer_stats(void* pt, gsp_stats* out) {
  return guard([&] { *out = ((PathTracer*)pt)->stats(); });
}
// TEST of the stale-scene hazard: every file is loaded into a `Scene` that lives in the SAME stack slot of this frame (the
// loop body's local), rendered with `spp` samples from timestamp 0 and downloaded; addresses[i] receives &scene of
// iteration i so that the caller can assert they really coincided.  out = n images of width*height*4 floats.
int gsph_pathtracer_render_files_same_slot(void* pt, const char* const* paths, const char* asset_dir, uint32_t n, uint32_t spp,
                                           float* out, uint64_t* addresses) {
  return guard([&] {
    PathTracer* p = (PathTracer*)pt;
    for (uint32_t i = 0; i < n; ++i) {
      Scene scene = loadScene(paths[i], asset_dir ? asset_dir : "");
      addresses[i] = (uint64_t)(uintptr_t)&scene;
      p->reset();
      p->render(scene, spp);
      const std::vector<float> img = p->download();
      std::memcpy(out + (size_t)i * img.size(), img.data(), img.size() * sizeof(float));
    }
  });
}
int gsph_write_pfm(const char* path, const float* rgba, uint32_t width, uint32_t height) {
  return guard([&] { writePfm(path, rgba, width, height); });
}

int gsph_write_ppm(const char* path, const float* rgba, uint32_t width, uint32_t height, int tone_map) {
  return guard([&] { writePpm(path, rgba, width, height, tone_map != 0); });
}
int gsph_tone_map(const float* rgba, uint32_t width, uint32_t height, int tone_map, uint8_t* rgb8) {
  return guard([&] {
    std::vector<uint8_t> v;
    toneMapToRgb8(rgba, width, height, tone_map != 0, v);
    std::memcpy(rgb8, v.data(), v.size());
  });
}

}  // extern "C"
