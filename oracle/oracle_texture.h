// oracle/oracle_texture.h -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// The dormant features of the reference: bitmap textures and the environment map.  The reference
// DECLARES them (Scene.h:31,66,75 hasTexture; Scene.h:116-119,182 Envmap; Loader.cpp:66-116 loadTexture /
// loadHdrTexture; rayhit.rchit:341-654 every BSDF function takes `vec2 uv`) but never runs them: the loader branches are
// commented out (Loader.cpp:122-143,338-346) and the closest-hit shader passes uv = vec2(0) (rayhit.rchit:716,729).
// There is therefore no reference behaviour to restate; this file is the DEFINITION of the extension that
// include/gpuspectral_pt.h documents, written as the sampler a Vulkan SAMPLER2D with linear filtering and repeat
// addressing performs (unnormalised coordinate u*w - 0.5, the four neighbours, weights frac).  PARITY UNPINNED by
// construction; scenes that do not set the fields never reach this code.
#pragma once
#include "oracle_bsdf.h"

namespace orc {

static inline int modWrap(int i, int n) {
  int r = i % n;
  return r < 0 ? r + n : r;
}
static inline int clampi(int i, int lo, int hi) { return i < lo ? lo : (i > hi ? hi : i); }
// texel-space coordinate of a normalised one; NaN / huge -> 0
static inline float unnormalised(float t, uint32_t n) {
  float c = t * (float)n - 0.5f;
  return fabsf(c) < 1.0e9f ? c : 0.0f;
}
static inline vec3 bilerp(vec3 a, vec3 b, vec3 c, vec3 d, float s, float t) {
  vec3 bottom = a * (1.0f - s) + b * s;
  vec3 top = c * (1.0f - s) + d * s;
  return bottom * (1.0f - t) + top * t;
}

struct TexelDecode {
  float table[256];
  explicit TexelDecode(const float* user) {
    for (int b = 0; b < 256; ++b) table[b] = user ? user[b] : (float)b / 255.0f;
  }
};

static inline vec3 rgba8(const TexelDecode& D, uint32_t px) {
  return V(D.table[px & 255u], D.table[(px >> 8) & 255u], D.table[(px >> 16) & 255u]);
}

// texture(sampler2D, uv).rgb: linear filter, repeat addressing
static inline vec3 textureLookup(const gsp_scene_desc& sc, const TexelDecode& D, uint32_t index, vec2 uv) {
  const gsp_texture& tex = sc.textures[index];
  const int w = (int)tex.width, h = (int)tex.height;
  const uint32_t* px = sc.texels + tex.first_texel;
  float cx = unnormalised(uv.x, tex.width), cy = unnormalised(uv.y, tex.height);
  float ix = floorf(cx), iy = floorf(cy);
  int i0 = modWrap((int)ix, w), j0 = modWrap((int)iy, h);
  int i1 = modWrap(i0 + 1, w), j1 = modWrap(j0 + 1, h);
  return bilerp(rgba8(D, px[(size_t)j0 * w + i0]), rgba8(D, px[(size_t)j0 * w + i1]), rgba8(D, px[(size_t)j1 * w + i0]),
                rgba8(D, px[(size_t)j1 * w + i1]), cx - ix, cy - iy);
}

// which texture (1-based) a BSDF handle asks for, 0 = none
static inline int textureOf(const gsp_scene_desc& sc, uint32_t handle) {
  uint32_t i = handle & 0xffffu;
  int k = 0;
  switch (handle >> 16) {
    case GSP_BSDF_DIFFUSE: k = sc.diffuse_bsdfs[i].has_texture; break;
    case GSP_BSDF_ROUGH_CONDUCTOR: k = sc.rough_conductor_bsdfs[i].has_texture; break;
    case GSP_BSDF_ROUGH_PLASTIC: k = sc.rough_plastic_bsdfs[i].has_texture; break;
    default: break;
  }
  if (k <= 0 || (uint32_t)k > sc.num_textures || !sc.uvs || !sc.textures || !sc.texels) return 0;
  return k;
}

// lat-long environment lookup for a world-space direction
static inline vec3 envmapLookup(const gsp_scene_desc& sc, vec3 dir) {
  const gsp_envmap& env = sc.envmap;
  vec3 e = xform_dir(env.to_local, dir);
  float azimuth = det_atan2f(e.x, -e.z);
  float polar = det_atan2f(sqrtf(e.x * e.x + e.z * e.z), e.y);
  float u = azimuth * 0.15915494309189533577f + 0.5f;
  float v = 1.0f - polar * 0.31830988618379067154f;
  const int w = (int)env.width, h = (int)env.height;
  float cx = unnormalised(u, env.width), cy = unnormalised(v, env.height);
  float ix = floorf(cx), iy = floorf(cy);
  int i0 = modWrap((int)ix, w), i1 = modWrap(i0 + 1, w);
  int j0 = clampi((int)iy, 0, h - 1), j1 = clampi((int)iy + 1, 0, h - 1);
  auto at = [&](int i, int j) { return ld3(env.texels + 4 * ((size_t)j * w + i)); };
  return bilerp(at(i0, j0), at(i1, j0), at(i0, j1), at(i1, j1), cx - ix, cy - iy);
}

}  // namespace orc
