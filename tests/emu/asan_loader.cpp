// Sanitizer driver of the C++ scene loader (CPU only; built with -fsanitize=address,undefined by
// tests/test_sanitizers.py): real scenes must load, malformed XML / OBJ must fail cleanly or load what is valid.
#include "Image.h"
#include "Loader.h"
#include "Scene.h"
#include <cstdio>
#include <fstream>
#include <iterator>
#include <vector>
using namespace GPUSpectral;
int main(int argc, char** argv) {
  // argv: asset dir, then scene files that must load; after "--images": image files that must decode, and whose
  // truncated / bit-flipped copies must decode or throw (never touch memory they do not own)
  const std::string R = argc > 1 ? argv[1] : ".";
  int firstImage = argc;
  for (int a = 2; a < argc; ++a)
    if (std::string(argv[a]) == "--images") {
      firstImage = a + 1;
      argc = a;
    }
  {
    char** av = argv;
    int total = 0;
    while (av[total]) ++total;
    unsigned rng = 12345u;
    auto next = [&]() { return rng = rng * 1664525u + 1013904223u; };
    int ok = 0, thrown = 0;
    for (int a = firstImage; a < total; ++a) {
      std::ifstream f(av[a], std::ios::binary);
      std::vector<uint8_t> data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
      const bool hdr = data.size() > 1 && (data[0] == '#' || data[0] == 'P');
      auto decode = [&](const std::vector<uint8_t>& d) {
        if (hdr) {
          if (d.size() > 1 && d[0] == 'P') decodePfm(d.data(), d.size());
          else decodeRgbe(d.data(), d.size());
        } else if (d.size() > 1 && d[0] == 0xff) {
          decodeJpeg(d.data(), d.size());
        } else {
          decodePng(d.data(), d.size());
        }
      };
      decode(data);  // the intact file must decode
      for (int k = 0; k < 400; ++k) {
        std::vector<uint8_t> m = data;
        if (k % 4 == 0) m.resize(next() % (m.size() + 1));
        for (unsigned j = 0; j < 1 + next() % 4 && !m.empty(); ++j) m[next() % m.size()] ^= (uint8_t)(1u << (next() % 8));
        try {
          decode(m);
          ++ok;
        } catch (const std::exception&) {
          ++thrown;
        }
      }
    }
    if (firstImage < total) printf("image fuzz: %d decoded, %d rejected\n", ok, thrown);
  }
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
