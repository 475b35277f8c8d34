// Sanitizer driver of the C++ scene loader (CPU only; built with -fsanitize=address,undefined by
// tests/test_sanitizers.py): real scenes must load, malformed XML / OBJ must fail cleanly or load what is valid.
#include "Loader.h"
#include "Scene.h"
#include <cstdio>
#include <fstream>
using namespace GPUSpectral;
int main(int argc, char** argv) {
  // argv: asset dir, then scene files that must load
  const std::string R = argc > 1 ? argv[1] : ".";
  for (int a = 2; a < argc; ++a) {
    Scene s = loadScene(argv[a], R);
    printf("%s: %zu objects\n", argv[a], s.renderObjects.size());
  }
  const char* bad[] = {"", "<scene>", "<scene version='0.5.0'><shape type='obj'><string name='filename' value='nope.obj'/></shape></scene>",
                       "<scene><bsdf type='diffuse' id='a'><rgb name='reflectance' value='1,2'/></bsdf><shape type='rectangle'><ref id='zzz'/></shape></scene>",
                       "<scene><sensor><transform name='toWorld'><matrix value='1 2 3'/></transform></sensor></scene>",
                       "<scene><shape type='rectangle'><emitter type='area'><rgb name='radiance' value='x'/></emitter></shape></scene>",
                       "<<<>>>", "<scene><bsdf type='twosided'><bsdf type='twosided'></bsdf></bsdf></scene>"};
  int i = 0;
  for (const char* txt : bad) {
    std::string p = "/tmp/bad_" + std::to_string(i++) + ".xml";
    std::ofstream(p) << txt;
    try {
      Scene s = loadScene(p, R);
      printf("malformed %d loaded: %zu objects\n", i, s.renderObjects.size());
    } catch (const std::exception& e) {
      printf("malformed %d: error ok: %.70s\n", i, e.what());
    }
  }
  // malformed OBJ
  std::ofstream("/tmp/bad.obj") << "v 1 2\nv a b c\nf 1 2 3\nf 1/2/3 9999 -7\nvn\nf\n";
  std::ofstream("/tmp/bad_obj.xml") << "<scene><shape type='obj'><string name='filename' value='/tmp/bad.obj'/></shape></scene>";
  try { Scene s = loadScene("/tmp/bad_obj.xml", R); printf("bad obj loaded: %zu objects\n", s.renderObjects.size()); }
  catch (const std::exception& e) { printf("bad obj: error ok: %.70s\n", e.what()); }
  puts("loader asan ok");
}
