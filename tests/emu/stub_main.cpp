// tests/emu/stub_main.cpp -- TEST HARNESS: compiles INTEGRATION.md's path-B binding (host/integration/PathTracerHip.h) in a
// translation unit that supplies what the reference tree would -- a FrameGraph type, the reference-shaped RenderPassCreator
// (S/renderer/Renderer.h:22-25) and `Scene` (here: this repository's mirror classes, host/Scene.h) -- and drives it the way
// S/main.cpp does: createRenderPass(fg, scene) once per frame.   stub_main scene.xml out.pfm W H frames [dx]
// dx != 0: after half of the frames the camera moves by dx and an object is shifted (the scene is re-read every frame).
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <string>
#include <vector>

#include "../../gpuspectral_amd/host/Loader.h"
#include "../../gpuspectral_amd/host/Scene.h"

namespace GPUSpectral {
struct FrameGraph {};  // S/backend/.../FrameGraph.h: untouched by this pass
class RenderPassCreator {  // S/renderer/Renderer.h:22-25
 public:
  virtual ~RenderPassCreator() = default;
  virtual void createRenderPass(FrameGraph& fg, const Scene& scene) = 0;
};
void writePfm(const std::string& path, const float* rgba, uint32_t width, uint32_t height);  // libgpuspectral_host
}  // namespace GPUSpectral

#include "../../gpuspectral_amd/host/integration/PathTracerHip.h"

using namespace GPUSpectral;

int main(int argc, char** argv) {
  if (argc < 6) return 2;
  const uint32_t W = (uint32_t)std::atoi(argv[3]), H = (uint32_t)std::atoi(argv[4]);
  const int frames = std::atoi(argv[5]);
  const float dx = argc > 6 ? (float)std::atof(argv[6]) : 0.0f;
  try {
    Scene scene = loadScene(argv[1], "");
    FrameGraph fg;
    PathTracerHip pass(W, H);
    for (int f = 0; f < frames; ++f) {
      if (dx != 0.0f && f == frames / 2) {  // a viewer edits its scene between two frames
        mat4 m = scene.camera.getToWorld();
        m[3][0] += dx;
        scene.camera.setToWorld(m);
        scene.renderObjects[5].transform[3][1] += dx;  // the short box rises
        scene.diffuseBSDFs[0].reflectance[2] = 0.9f;
      }
      pass.createRenderPass(fg, scene);
    }
    std::vector<float> img((size_t)W * H * 4);
    pass.download(img.data());
    writePfm(argv[2], img.data(), W, H);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
