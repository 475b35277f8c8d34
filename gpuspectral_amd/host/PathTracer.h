// PathTracer.h -- host half of the path-tracing pass, the drop-in for the reference's
// `class PathTracer : public RenderPassCreator` (S/renderer/PathTracer.h:48-68,
// S/renderer/Renderer.h:22-25).
//
// Same two-phase shape as the reference:
//   construction  = setup(): allocate the frame-sized RGBA32F accumulate buffer
//                   (PathTracer.cpp:5-7)
//   createRenderPass(scene) = one call per presented frame: prepareScene + traceRays
//                   -> ONE more sample per pixel folded into the running mean, with the
//                   internal `timestamp` counter advanced (PathTracer.cpp:9-56,91-92)
// plus `render(scene, spp)` for offline use.  Instead of recording Vulkan passes into a
// FrameGraph it drives the HIP wavefront tracer through the C ABI
// (include/gpuspectral_pt.h).  Errors are C++ exceptions (std::runtime_error), as in the
// reference.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "Scene.h"

namespace GPUSpectral {

// What a pass last handed to the device, BY VALUE.  The reference's createRenderPass(fg, scene) re-reads the camera, every
// object's transform and material, the eight BSDF tables and the lights on every call and keeps only the BLAS of a mesh,
// cached by mesh id (S/renderer/PathTracer.cpp:10-19,58-93; Renderer.cpp:122-131).  Its Scene is a plain struct with
// public vectors, so no mutator could keep a revision counter honest; the tracker therefore compares VALUES each frame
// (~100 B per object + the tables: microseconds) and answers with the cheapest C-ABI call that brings the device up to
// date.  Mesh identity is the MeshPtr itself, HELD here: while the tracker holds it no other Mesh can be constructed at
// that address, so a different scene built in the same stack slot can never be mistaken for the uploaded one.
struct SceneTracker {
  enum Change : unsigned { None = 0, CameraChanged = 1, TablesChanged = 2, InstancesChanged = 4, Everything = 8 };
  // what differs between `scene` and the snapshot (Everything: other meshes / object list / textures: a full upload)
  unsigned diff(const Scene& scene, std::vector<gsp_instance>& instances) const;
  void remember(const Scene& scene, const std::vector<gsp_instance>& instances);
  void forget() { valid = false; meshes.clear(); }

 private:
  bool valid = false;
  std::vector<MeshPtr> meshes;          // per render object
  std::vector<gsp_instance> instances;  // transform, emission, bsdf, twofaced, vertex range per object
  std::vector<unsigned char> tables;    // counts + the eight BSDF arrays + the lights, back to back
  gsp_camera camera{};
  std::vector<uint64_t> assets;         // dormant features: texture / environment-map sizes, flags, envmap transform
};

// The plugin interface of the reference (Renderer.h:22-25) without the Vulkan FrameGraph.
class RenderPassCreator {
 public:
  virtual ~RenderPassCreator() = default;
  virtual void createRenderPass(const Scene& scene) = 0;
};

class PathTracer : public RenderPassCreator {
 public:
  // `pixelIds` optionally restricts this tracer to a subset of the frame (multi-GPU tiles).
  // `options` (optional): per-context resources, gsp_ctx_options of include/gpuspectral_pt.h (path-pool size, memory share, ...)
  PathTracer(uint32_t width, uint32_t height, int device = 0, const std::vector<uint32_t>& pixelIds = {},
             const gsp_ctx_options* options = nullptr);
  ~PathTracer() override;
  PathTracer(const PathTracer&) = delete;
  PathTracer& operator=(const PathTracer&) = delete;

  void setup();
  void createRenderPass(const Scene& scene) override;  // +1 spp
  void render(const Scene& scene, uint32_t spp);       // +spp samples
  // Brings the device up to date with `scene` as it is NOW (the reference re-reads it every frame): nothing when nothing
  // changed, gsp_update_camera / _tables / _instances for edits, a full flatten + upload + BVH build for another object list.
  void prepareScene(const Scene& scene);
  // Forget what was uploaded: the next pass uploads everything.  Needed only after texel CONTENTS of a texture or
  // environment map were rewritten in place (dormant features; sizes and flags are tracked, contents are not).
  void invalidateScene() { tracker.forget(); }

  // RGBA32F, row-major, width*height*4 floats (running mean, alpha 1)
  std::vector<float> download();
  // The frame as it stands without waiting for paths still in flight (what the reference's blit pass shows every
  // frame, PathTracer.cpp:41-55): compact RGBA32F over the owned pixels; *samplesFolded = timestamps in every pixel.
  std::vector<float> peek(uint32_t* samplesFolded = nullptr);
  // ... into caller-owned DEVICE memory (>= owned pixels * 16 bytes): the blit's source without a trip through the host
  void peekToDevice(void* deviceDst, uint64_t bytes, uint32_t* samplesFolded = nullptr);
  void reset();  // timestamp = 0, accumulate buffer cleared
  int getTimestamp() const { return timestamp; }
  gsp_stats stats();
  gsp_render_params params;  // reference literals by default (MAX_DEPTH 50, RR > 10, clamp 20)

 private:
  void check(int rc, const char* what);
  gsp_context* ctx = nullptr;
  uint32_t width, height;
  int device;
  std::vector<uint32_t> pixelIds;
  gsp_ctx_options options{};
  SceneTracker tracker;
  int timestamp{0};
};

// The same pass over several GPUs of one node: the frame is cut into interleaved 32x32 tiles (one share per entry of
// `devices`; an index may repeat), the scene is replicated, every GPU renders all samples of its tiles and download()
// gathers the HDR tiles into the first GPU (device-to-device over xGMI) before the one copy to the host
// (gsp_multi_*, include/gpuspectral_pt.h).  Same two-phase shape and the same image, bit for bit, as
// PathTracer on one GPU.
class MultiGpuPathTracer : public RenderPassCreator {
 public:
  MultiGpuPathTracer(uint32_t width, uint32_t height, const std::vector<int>& devices, const gsp_ctx_options* options = nullptr);
  ~MultiGpuPathTracer() override;
  MultiGpuPathTracer(const MultiGpuPathTracer&) = delete;
  MultiGpuPathTracer& operator=(const MultiGpuPathTracer&) = delete;

  void createRenderPass(const Scene& scene) override;  // +1 spp on every GPU's share
  void render(const Scene& scene, uint32_t spp);
  void prepareScene(const Scene& scene);  // as PathTracer::prepareScene, on every share
  void invalidateScene() { tracker.forget(); }
  std::vector<float> download();  // RGBA32F, row-major, width*height*4 floats
  void reset();
  int getTimestamp() const { return timestamp; }
  int numShares() const { return (int)devices.size(); }
  gsp_stats stats(std::vector<gsp_stats>* perShare = nullptr);  // totals over the shares
  gsp_render_params params;

 private:
  void check(int rc, const char* what);
  gsp_multi* multi = nullptr;
  uint32_t width, height;
  std::vector<int> devices;
  SceneTracker tracker;
  int timestamp{0};
};

// Headless output step: little-endian PFM ("PF", bottom-to-top rows) of the RGB channels.
void writePfm(const std::string& path, const float* rgba, uint32_t width, uint32_t height);

// Display transform of the presentation pass.  The reference blits the accumulate image unchanged
// (S/assets/shaders/DrawTexture.frag:9-13; `RenderParams.toneMap` is never read, S/renderer/
// PathTracer.h:36-41) and keeps the ACES fit as a dormant helper (S/assets/shaders/common.glsl:74-82).
// toneMap = false: clamp + gamma 2.2 (the ldrfilm setting of the shipped scenes); true: ACESFilm, then gamma.
void toneMapToRgb8(const float* rgba, uint32_t width, uint32_t height, bool toneMap, std::vector<uint8_t>& rgb8);
// Binary PPM ("P6") of the tone-mapped image (top-to-bottom rows).
void writePpm(const std::string& path, const float* rgba, uint32_t width, uint32_t height, bool toneMap);

}  // namespace GPUSpectral
