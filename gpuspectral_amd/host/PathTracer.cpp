// PathTracer.cpp -- see PathTracer.h.  Reference: S/renderer/PathTracer.cpp:5-93.
#include "PathTracer.h"

#include <cmath>
#include <cstdio>
#include <stdexcept>

namespace GPUSpectral {

PathTracer::PathTracer(uint32_t width, uint32_t height, int device, const std::vector<uint32_t>& pixelIds)
    : width(width), height(height), device(device), pixelIds(pixelIds) {
  gsp_default_render_params(&params);
  setup();
}

PathTracer::~PathTracer() { gsp_ctx_destroy(ctx); }

void PathTracer::check(int rc, const char* what) {
  if (rc != GSP_OK) throw std::runtime_error(std::string(what) + ": " + gsp_last_error(ctx));
}

// PathTracer.cpp:5-7
void PathTracer::setup() {
  int rc = gsp_ctx_create(device, &ctx);
  if (rc != GSP_OK) throw std::runtime_error(std::string("gsp_ctx_create: ") + gsp_last_error(nullptr));
  check(gsp_frame_begin(ctx, width, height, pixelIds.empty() ? nullptr : pixelIds.data(), pixelIds.size()),
        "gsp_frame_begin");
}

void PathTracer::reset() {
  timestamp = 0;
  check(gsp_frame_begin(ctx, width, height, pixelIds.empty() ? nullptr : pixelIds.data(), pixelIds.size()),
        "gsp_frame_begin");
}

// PathTracer.cpp:58-93.  The reference rebuilds the TLAS and re-uploads every table each
// frame; the scene is immutable between frames in every caller, so this uploads once per
// Scene object (identity + object count) and keeps the BVH resident.
void PathTracer::prepareScene(const Scene& scene) {
  if (uploaded == &scene && uploadedObjects == scene.renderObjects.size()) return;
  FlatScene flat;
  flattenScene(scene, flat);
  check(gsp_upload_scene(ctx, &flat.desc), "gsp_upload_scene");
  uploaded = &scene;
  uploadedObjects = scene.renderObjects.size();
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

gsp_stats PathTracer::stats() {
  gsp_stats s;
  check(gsp_get_stats(ctx, &s), "gsp_get_stats");
  return s;
}

// ---- several GPUs ---------------------------------------------------------------------------------------------
MultiGpuPathTracer::MultiGpuPathTracer(uint32_t width, uint32_t height, const std::vector<int>& devices)
    : width(width), height(height), devices(devices) {
  gsp_default_render_params(&params);
  int rc = gsp_multi_create(devices.data(), (int)devices.size(), &multi);
  if (rc != GSP_OK) throw std::runtime_error(std::string("gsp_multi_create: ") + gsp_multi_last_error(nullptr));
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
  if (uploaded == &scene && uploadedObjects == scene.renderObjects.size()) return;
  FlatScene flat;
  flattenScene(scene, flat);
  check(gsp_multi_upload_scene(multi, &flat.desc), "gsp_multi_upload_scene");
  uploaded = &scene;
  uploadedObjects = scene.renderObjects.size();
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
