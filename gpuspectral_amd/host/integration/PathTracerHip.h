// PathTracerHip.h -- the binding a GPUSpectral maintainer adds to the REFERENCE tree in place of
// src/GPUSpectral/renderer/PathTracer.{h,cpp} (path B): a RenderPassCreator that keeps the reference's own
// Scene / Renderer and drives libgpuspectral_pt.so through the C ABI only.
//
// It is written against the reference's interfaces --
//   class RenderPassCreator { virtual void createRenderPass(FrameGraph& fg, const Scene& scene) = 0; }   S/renderer/Renderer.h:22-25
//   Scene::renderObjects / getMaterial / <x>BSDFs / triangleLights / camera                                S/renderer/Scene.h:140-186
//   Mesh::getVertices(), Material{emission, twofaced, bsdf}, BSDFHandle::handle                             S/renderer/Mesh.h:59-60, Scene.h:83-104
// -- and includes nothing of this repository but the C header, so the including translation unit decides which `Scene` it
// sees: the reference's (with glm) in the reference tree, or this repository's mirror classes (host/Scene.h), which is how
// tests/test_integration_stub.py compiles it (CPU) and runs it on the GPU against the oracle.  The reference's glm::mat4 and
// the mirror's mat4 are both 16 floats in column-major order reachable through &m[0][0].
#pragma once
#include <gpuspectral_pt.h>

#include <cstring>
#include <stdexcept>
#include <unordered_map>
#include <utility>
#include <vector>

namespace GPUSpectral {

class PathTracerHip : public RenderPassCreator {
 public:
  PathTracerHip(uint32_t w, uint32_t h) : width(w), height(h) {
    if (gsp_ctx_create(0, &ctx)) throw std::runtime_error(gsp_last_error(nullptr));
    check(gsp_frame_begin(ctx, w, h, nullptr, 0));  // = setup(): accumulateBuffer, PathTracer.cpp:5-7
  }
  ~PathTracerHip() override { gsp_ctx_destroy(ctx); }

  void createRenderPass(FrameGraph&, const Scene& scene) override {  // PathTracer.cpp:9-56
    if (!sameMeshes(scene)) upload(scene);  // meshes -> device once: the reference's BLAS cache (Renderer.cpp:122-131)
    else refresh(scene);                    // what the reference re-reads every frame (PathTracer.cpp:10-19,58-93)
    gsp_render_params p;
    gsp_default_render_params(&p);          // MAX_DEPTH 50, RR > 10, clamp 20, NEE on
    p.spp = 1;
    p.first_timestamp = (uint32_t)timestamp++;  // renderState.params.timestamp, PathTracer.cpp:91-92
    check(gsp_render(ctx, &p));
  }
  void download(float* rgba) { check(gsp_download(ctx, rgba)); }  // what the DrawTexture blit sampled, once every path has ended
  // ... and as it stands this frame, into the device memory the blit reads (PathTracer.cpp:41-55): no wait, no trip through the host
  void blitSource(void* deviceRgba, uint64_t bytes, uint32_t* samples = nullptr) { check(gsp_peek_to_device(ctx, deviceRgba, bytes, samples)); }
  void restart() {                                                // a viewer that wants a fresh running mean after an edit
    timestamp = 0;
    check(gsp_frame_begin(ctx, width, height, nullptr, 0));
  }

 private:
  // PathTracer::prepareScene, PathTracer.cpp:58-70: one record per render object; shared meshes stored once
  void instancesOf(const Scene& s, std::vector<gsp_instance>& inst, std::vector<float>* pos, std::vector<float>* nrm) {
    std::unordered_map<const Mesh*, std::pair<uint32_t, uint32_t>> placed;
    uint32_t next = 0;
    for (auto& obj : s.renderObjects) {
      auto it = placed.find(obj.mesh.get());
      if (it == placed.end()) {
        const auto& vs = obj.mesh->getVertices();  // Mesh.cpp:109-111 (CPU copy of the de-indexed vertices)
        const uint32_t count = (uint32_t)(vs.size() / 3 * 3);
        if (pos)
          for (uint32_t i = 0; i < count; ++i) {
            pos->insert(pos->end(), {vs[i].pos.x, vs[i].pos.y, vs[i].pos.z});
            nrm->insert(nrm->end(), {vs[i].normal.x, vs[i].normal.y, vs[i].normal.z});
          }
        it = placed.emplace(obj.mesh.get(), std::make_pair(next, count)).first;
        next += count;
      }
      const Material& m = s.getMaterial(obj.material);
      gsp_instance gi{};
      std::memcpy(gi.transform, &obj.transform[0][0], 64);  // glm::mat4 memory order
      gi.emission[0] = m.emission.x, gi.emission[1] = m.emission.y, gi.emission[2] = m.emission.z;
      gi.bsdf = m.bsdf.handle;
      gi.twofaced = m.twofaced ? 1u : 0u;
      gi.first_vertex = it->second.first;
      gi.vertex_count = it->second.second;
      inst.push_back(gi);
    }
  }
  // PathTracer.cpp:74-90: the eight BSDF arrays (BSDF.inc order), the lights, the camera -- by pointer, no repacking
  static void fillTables(const Scene& s, gsp_scene_desc& d) {
    d.diffuse_bsdfs = (const gsp_diffuse_bsdf*)s.diffuseBSDFs.data(), d.num_bsdfs[0] = (uint32_t)s.diffuseBSDFs.size();
    d.smooth_dielectric_bsdfs = (const gsp_smooth_dielectric_bsdf*)s.smoothDielectricBSDFs.data(), d.num_bsdfs[1] = (uint32_t)s.smoothDielectricBSDFs.size();
    d.smooth_conductor_bsdfs = (const gsp_smooth_conductor_bsdf*)s.smoothConductorBSDFs.data(), d.num_bsdfs[2] = (uint32_t)s.smoothConductorBSDFs.size();
    d.smooth_plastic_bsdfs = (const gsp_smooth_plastic_bsdf*)s.smoothPlasticBSDFs.data(), d.num_bsdfs[3] = (uint32_t)s.smoothPlasticBSDFs.size();
    d.rough_conductor_bsdfs = (const gsp_rough_conductor_bsdf*)s.roughConductorBSDFs.data(), d.num_bsdfs[4] = (uint32_t)s.roughConductorBSDFs.size();
    d.smooth_floor_bsdfs = (const gsp_smooth_floor_bsdf*)s.smoothFloorBSDFs.data(), d.num_bsdfs[5] = (uint32_t)s.smoothFloorBSDFs.size();
    d.rough_floor_bsdfs = (const gsp_rough_floor_bsdf*)s.roughFloorBSDFs.data(), d.num_bsdfs[6] = (uint32_t)s.roughFloorBSDFs.size();
    d.rough_plastic_bsdfs = (const gsp_rough_plastic_bsdf*)s.roughPlasticBSDFs.data(), d.num_bsdfs[7] = (uint32_t)s.roughPlasticBSDFs.size();
    d.lights = (const gsp_triangle_light*)s.triangleLights.data(), d.num_lights = (uint32_t)s.triangleLights.size();
    const auto tw = s.camera.getToWorld();
    std::memcpy(d.camera.to_world, &tw[0][0], 64);
    d.camera.fov = s.camera.getFov();
  }
  bool sameMeshes(const Scene& s) const {  // the MeshPtrs are HELD in `meshes`: no other Mesh can reuse their addresses
    if (s.renderObjects.size() != meshes.size()) return false;
    for (size_t i = 0; i < meshes.size(); ++i)
      if (s.renderObjects[i].mesh != meshes[i]) return false;
    return !meshes.empty() || uploaded;
  }
  void upload(const Scene& s) {  // prepareScene + BLAS / TLAS build
    std::vector<gsp_instance> inst;
    std::vector<float> pos, nrm;
    instancesOf(s, inst, &pos, &nrm);
    gsp_scene_desc d{};
    d.instances = inst.data(), d.num_instances = (uint32_t)inst.size();
    d.positions = pos.data(), d.normals = nrm.data(), d.num_vertices = pos.size() / 3;
    fillTables(s, d);
    check(gsp_upload_scene(ctx, &d));  // copies; device bake + BVH build
    meshes.clear();
    for (auto& obj : s.renderObjects) meshes.push_back(obj.mesh);
    uploaded = true;
  }
  // Each call returns before any wait when its input equals what the device holds (the library compares), so an unchanged
  // scene costs three comparisons.  Tables go first (an edited instance may name a BSDF the new tables add); when the new
  // tables DROP a record a resident instance still names, the library refuses them, the instances go first and the tables
  // are sent again; an edit that needs both orders at once is a full upload (as host/PathTracer.cpp::prepareScene does).
  void refresh(const Scene& s) {
    gsp_scene_desc d{};
    fillTables(s, d);
    std::vector<gsp_instance> inst;
    instancesOf(s, inst, nullptr, nullptr);
    const int rcT = gsp_update_tables(ctx, &d);                                          // PathTracer.cpp:74-87
    const int rcI = gsp_update_instances(ctx, inst.data(), (uint32_t)inst.size());      // PathTracer.cpp:10-19,60-70 (TLAS + Instance records)
    if (rcT != GSP_OK && rcI == GSP_OK) check(gsp_update_tables(ctx, &d));
    else if (rcT != GSP_OK || rcI != GSP_OK) {
      uploaded = false;
      upload(s);
      return;
    }
    check(gsp_update_camera(ctx, &d.camera));                                            // PathTracer.cpp:88-90
  }
  void check(int rc) {
    if (rc) throw std::runtime_error(gsp_last_error(ctx));  // the reference's convention: exceptions
  }
  gsp_context* ctx = nullptr;
  uint32_t width, height;
  int timestamp = 0;
  bool uploaded = false;
  std::vector<MeshPtr> meshes;
};

}  // namespace GPUSpectral
