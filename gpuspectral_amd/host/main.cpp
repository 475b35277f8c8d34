// main.cpp -- headless counterpart of the reference's S/main.cpp:15-30:
//   Engine + PathTracer + loadScene + Window::run(frame loop)
// becomes: load the Mitsuba XML, render N samples per pixel, write the HDR framebuffer.
//   gsp_render [--dormant-features] [--builtin-shapes] [--no-nee] [--memory-share F] [--pool-paths N] <scene.xml> <out.pfm> [width height spp [devices]]
//   devices: "0" (default) or a list "0,1,2,3": the frame is then tiled over those GPUs (MultiGpuPathTracer); an index
//   may repeat.  --dormant-features: LoadOptions::dormantFeatures (bitmap / checkerboard textures, envmap emitter);
//   --builtin-shapes: LoadOptions::builtinShapes (`disk` / `sphere` shapes are tessellated instead of skipped);
//   --no-nee: gsp_render_params.disable_nee = 1 (RenderParams.nee = false); --memory-share / --pool-paths: gsp_ctx_options (how much device memory the path pool takes)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <string>
#include <vector>

#include "Loader.h"
#include "PathTracer.h"

using namespace GPUSpectral;

int main(int argc, char** argv) {
  LoadOptions options;
  gsp_ctx_options ctxOptions;
  gsp_default_ctx_options(&ctxOptions);
  bool nee = true;
  while (argc > 1 && argv[1][0] == '-' && argv[1][1] == '-') {
    const std::string flag = argv[1];
    int used = 1;
    if (flag == "--dormant-features") options.dormantFeatures = true;
    else if (flag == "--builtin-shapes") options.builtinShapes = true;
    else if (flag == "--no-nee") nee = false;
    else if (flag == "--memory-share" && argc > 2) ctxOptions.memory_share = std::atof(argv[2]), used = 2;
    else if (flag == "--pool-paths" && argc > 2) ctxOptions.pool_paths = std::strtoull(argv[2], nullptr, 10), used = 2;
    else {
      std::fprintf(stderr, "gsp_render: unknown option '%s'\n", argv[1]);
      return 2;
    }
    argc -= used;
    argv += used;
  }
  if (argc < 3) {
    std::fprintf(stderr, "usage: gsp_render [--dormant-features] [--builtin-shapes] [--no-nee] [--memory-share F] [--pool-paths N] scene.xml out.pfm [width height spp [device | d0,d1,...]]\n");
    return 2;
  }
  const uint32_t width = argc > 3 ? (uint32_t)std::atoi(argv[3]) : 500;  // S/main.cpp:17: 500x500 window
  const uint32_t height = argc > 4 ? (uint32_t)std::atoi(argv[4]) : 500;
  const uint32_t spp = argc > 5 ? (uint32_t)std::atoi(argv[5]) : 64;
  // device list: decimal indices separated by single commas ("0", "0,1,2,3"; a repeated index = several shares on one
  // GPU); anything else is a usage error, not "device 0"
  std::vector<int> devices;
  {
    const char* p = argc > 6 ? argv[6] : "0";
    bool ok = *p != 0;
    while (ok) {
      if (*p < '0' || *p > '9') {
        ok = false;
        break;
      }
      char* e;
      const long v = std::strtol(p, &e, 10);
      if (e == p || v < 0 || v > 1023) {
        ok = false;
        break;
      }
      devices.push_back((int)v);
      if (*e == 0) break;
      if (*e != ',') ok = false;
      p = e + 1;
    }
    if (!ok || devices.empty()) {
      std::fprintf(stderr, "gsp_render: bad device list '%s' (expected e.g. 0 or 0,1,2,3)\n", argc > 6 ? argv[6] : "");
      return 2;
    }
  }
  try {
    Scene scene = loadScene(argv[1], "", options);
    for (auto& w : scene.warnings) std::fprintf(stderr, "WARN: %s\n", w.c_str());
    std::vector<float> img;
    gsp_stats st;
    double s;
    if (devices.size() == 1) {
      PathTracer pt(width, height, devices[0], {}, &ctxOptions);
      pt.params.disable_nee = nee ? 0u : 1u;
      auto t0 = std::chrono::steady_clock::now();
      pt.render(scene, spp);
      img = pt.download();
      s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      st = pt.stats();
    } else {
      MultiGpuPathTracer pt(width, height, devices, &ctxOptions);
      pt.params.disable_nee = nee ? 0u : 1u;
      auto t0 = std::chrono::steady_clock::now();
      pt.render(scene, spp);
      img = pt.download();
      s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      st = pt.stats();
      std::printf("%zu shares (32x32 tiles), gathered on device %d\n", devices.size(), devices[0]);
    }
    writePfm(argv[2], img.data(), width, height);
    writePpm(std::string(argv[2]) + ".ppm", img.data(), width, height, false);  // LDR preview, gamma 2.2
    std::printf("%llu triangles, %ux%u x %u spp in %.3f s: %.1f Mrays/s, %.2f Msamples/s (BVH build %.1f ms)\n",
                (unsigned long long)st.num_triangles, width, height, spp, s,
                (st.extension_rays + st.shadow_rays) / s / 1e6, st.samples / s / 1e6, st.bvh_build_ms);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
