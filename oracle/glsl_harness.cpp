// oracle/glsl_harness.cpp -- TEST INFRASTRUCTURE ONLY.  Built only where /root/reference is mounted, by
// tests/golden/make_glsl_vectors.py, into oracle/_ref/libglsl_ref.so (git-ignored).
//
// Runs the reference's own shader TEXT.  The four GLSL_PART_* macros name temporary files that hold line ranges of
// S/assets/shaders/pt_common.glsl and rayhit.rchit after the three token-level rewrites described in glsl_shim.h; this file
// supplies what the shader gets from its environment (the RenderState block and the buffer-reference types,
// pt_common.glsl:16-66, rayhit.rchit:71-87) and C entry points that call the extracted functions with given inputs.
// No function of the reference is restated here.
#include <cstring>

#include "../include/gpuspectral_pt.h"
#include "glsl_shim.h"

#if !defined(GLSL_PART_HANDLES) || !defined(GLSL_PART_RNG_ONB) || !defined(GLSL_PART_STRUCTS) || !defined(GLSL_PART_FUNCTIONS)
#error "built by tests/golden/make_glsl_vectors.py (needs /root/reference)"
#endif

namespace glsl {

#include GLSL_PART_HANDLES   // pt_common.glsl:28-42    BSDFHandle, bsdfHandle / bsdfType / bsdfIndex
#include GLSL_PART_RNG_ONB   // pt_common.glsl:86-151   rngState, randPcg, pcgHash, randUniform, tea, Onb
#include GLSL_PART_STRUCTS   // rayhit.rchit:17-69      TriangleLight and the eight BSDF records

// the shader's environment: one device pointer per BSDF table (pt_common.glsl:54-66 through the reference's own BSDF.inc),
// the light table, the light count
#define BSDFDefinition(BSDFNAME, BSDFFIELD, BSDFTYPE) typedef BufferRef<BSDFNAME> BSDFNAME##Buffer;
#include "BSDF.inc"
#undef BSDFDefinition
typedef BufferRef<TriangleLight> TriangleLightBuffer;
struct Scene {
#define BSDFDefinition(BSDFNAME, BSDFFIELD, BSDFTYPE) const BSDFNAME* BSDFFIELD##s = nullptr;
#include "BSDF.inc"
#undef BSDFDefinition
  const TriangleLight* triangleLights = nullptr;
  int numLights = 0;
};
struct RenderState {
  Scene scene;
};
static RenderState renderState;

#include GLSL_PART_FUNCTIONS  // rayhit.rchit:89-654    samplers, Fresnel terms, 8 BSDF pairs, dispatch, light sampling

// the records cross the C ABI as the byte-compatible PODs of include/gpuspectral_pt.h (S/renderer/Scene.h:29-109)
static_assert(sizeof(DiffuseBSDF) == sizeof(gsp_diffuse_bsdf) && sizeof(SmoothDielectricBSDF) == sizeof(gsp_smooth_dielectric_bsdf) &&
                  sizeof(SmoothConductorBSDF) == sizeof(gsp_smooth_conductor_bsdf) &&
                  sizeof(SmoothPlasticBSDF) == sizeof(gsp_smooth_plastic_bsdf) &&
                  sizeof(RoughConductorBSDF) == sizeof(gsp_rough_conductor_bsdf) && sizeof(SmoothFloorBSDF) == sizeof(gsp_smooth_floor_bsdf) &&
                  sizeof(RoughFloorBSDF) == sizeof(gsp_rough_floor_bsdf) && sizeof(RoughPlasticBSDF) == sizeof(gsp_rough_plastic_bsdf),
              "the shader's scalar-layout records and the ABI PODs are the same bytes");

template <class T, class S>
static const T* keep(const S* src, uint32_t n) {
  T* p = new T[n ? n : 1];
  if (n) memcpy((void*)p, src, sizeof(T) * (size_t)n);
  return p;
}

static inline uint32_t bits(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}
static inline float fbits(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}

}  // namespace glsl

using namespace glsl;

extern "C" {

// the tables of `sc` become the shader's RenderState.scene (leaks the previous ones: a generator, run once)
void glsl_set_tables(const gsp_scene_desc* sc) {
  Scene& s = renderState.scene;
  s.diffuseBSDFs = keep<DiffuseBSDF>(sc->diffuse_bsdfs, sc->num_bsdfs[0]);
  s.smoothDielectricBSDFs = keep<SmoothDielectricBSDF>(sc->smooth_dielectric_bsdfs, sc->num_bsdfs[1]);
  s.smoothConductorBSDFs = keep<SmoothConductorBSDF>(sc->smooth_conductor_bsdfs, sc->num_bsdfs[2]);
  s.smoothPlasticBSDFs = keep<SmoothPlasticBSDF>(sc->smooth_plastic_bsdfs, sc->num_bsdfs[3]);
  s.roughConductorBSDFs = keep<RoughConductorBSDF>(sc->rough_conductor_bsdfs, sc->num_bsdfs[4]);
  s.smoothFloorBSDFs = keep<SmoothFloorBSDF>(sc->smooth_floor_bsdfs, sc->num_bsdfs[5]);
  s.roughFloorBSDFs = keep<RoughFloorBSDF>(sc->rough_floor_bsdfs, sc->num_bsdfs[6]);
  s.roughPlasticBSDFs = keep<RoughPlasticBSDF>(sc->rough_plastic_bsdfs, sc->num_bsdfs[7]);
  TriangleLight* L = new TriangleLight[sc->num_lights ? sc->num_lights : 1];
  for (uint32_t i = 0; i < sc->num_lights; ++i) {
    for (int k = 0; k < 3; ++k) memcpy(&L[i].positions[k], sc->lights[i].positions[k], 16);
    L[i].emission.x = sc->lights[i].radiance[0];
    L[i].emission.y = sc->lights[i].radiance[1];
    L[i].emission.z = sc->lights[i].radiance[2];
  }
  s.triangleLights = L;
  s.numLights = (int)sc->num_lights;
}

// n x {sampleBSDF(vec2(0), handle, wo, wi, res)} with rngState = seed (rayhit.rchit:668,716):
// out = n x 9 words {wi.xyz, res.bsdf.rgb, res.pdf, isDelta ? 1.0f : 0.0f, rngState afterwards}
void glsl_bsdf_sample(uint64_t n, const uint32_t* handles, const float* wo, const uint32_t* seeds, uint32_t* out) {
  for (uint64_t i = 0; i < n; ++i) {
    rngState = seeds[i];
    vec3 wi;
    BSDFOutput res;
    memset((void*)&wi, 0, sizeof wi);
    memset((void*)&res, 0, sizeof res);
    sampleBSDF(vec2(0), handles[i], vec3(wo[3 * i], wo[3 * i + 1], wo[3 * i + 2]), wi, res);
    const float r[8] = {wi.x, wi.y, wi.z, res.bsdf.x, res.bsdf.y, res.bsdf.z, res.pdf, res.isDelta ? 1.0f : 0.0f};
    for (int k = 0; k < 8; ++k) out[9 * i + k] = bits(r[k]);
    out[9 * i + 8] = rngState;
  }
}

// n x {evalBSDF(handle, vec2(0), wo, wi, res)} (rayhit.rchit:729): out = n x 5 words {res.bsdf.rgb, res.pdf, isDelta}
void glsl_bsdf_eval(uint64_t n, const uint32_t* handles, const float* wo, const float* wi, uint32_t* out) {
  for (uint64_t i = 0; i < n; ++i) {
    BSDFOutput res;
    memset((void*)&res, 0, sizeof res);
    evalBSDF(handles[i], vec2(0), vec3(wo[3 * i], wo[3 * i + 1], wo[3 * i + 2]), vec3(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]), res);
    const float r[5] = {res.bsdf.x, res.bsdf.y, res.bsdf.z, res.pdf, res.isDelta ? 1.0f : 0.0f};
    for (int k = 0; k < 5; ++k) out[5 * i + k] = bits(r[k]);
  }
}

// n x {sampleLight(pos)} with rngState = seed (rayhit.rchit:720): out = n x 8 words {position, emission, pdf, rngState}
void glsl_sample_light(uint64_t n, const float* pos, const uint32_t* seeds, uint32_t* out) {
  for (uint64_t i = 0; i < n; ++i) {
    rngState = seeds[i];
    LightOutput r = sampleLight(vec3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]));
    const float f[7] = {r.position.x, r.position.y, r.position.z, r.emission.x, r.emission.y, r.emission.z, r.pdf};
    for (int k = 0; k < 7; ++k) out[8 * i + k] = bits(f[k]);
    out[8 * i + 7] = rngState;
  }
}

// n x {Onb onb = onbCreate(nrm); onbTransform(onb, v); onbUntransform(onb, v)} (pt_common.glsl:122-151):
// out = n x 15 words {tangent, binormal, normal, transformed, untransformed}
void glsl_onb(uint64_t n, const float* nrm, const float* v, uint32_t* out) {
  for (uint64_t i = 0; i < n; ++i) {
    Onb o = onbCreate(vec3(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]));
    vec3 a = onbTransform(o, vec3(v[3 * i], v[3 * i + 1], v[3 * i + 2]));
    vec3 b = onbUntransform(o, vec3(v[3 * i], v[3 * i + 1], v[3 * i + 2]));
    const float f[15] = {o.tangent.x, o.tangent.y, o.tangent.z, o.binormal.x, o.binormal.y, o.binormal.z, o.normal.x, o.normal.y,
                         o.normal.z,  a.x,         a.y,         a.z,          b.x,          b.y,          b.z};
    for (int k = 0; k < 15; ++k) out[15 * i + k] = bits(f[k]);
  }
}

// RNG (pt_common.glsl:86-120): out = n x 6 words {tea(a, b), pcgHash(a), then with rngState = a: randPcg(), randPcg(),
// randUniform() as bits, rngState afterwards}
void glsl_rng(uint64_t n, const uint32_t* a, const uint32_t* b, uint32_t* out) {
  for (uint64_t i = 0; i < n; ++i) {
    out[6 * i + 0] = tea(a[i], b[i]);
    out[6 * i + 1] = pcgHash(a[i]);
    rngState = a[i];
    out[6 * i + 2] = randPcg();
    out[6 * i + 3] = randPcg();
    out[6 * i + 4] = bits(randUniform());
    out[6 * i + 5] = rngState;
  }
}

// scalar helpers on their own (rayhit.rchit:212-216,107-115): out = n x 3 words {powerHeuristic(1, f, 1, g),
// cosineHemispherePdf(vec3(0, 0, f)), isTransimissionBSDF(handle type) }
void glsl_helpers(uint64_t n, const float* f, const float* g, const uint32_t* handles, uint32_t* out) {
  for (uint64_t i = 0; i < n; ++i) {
    out[3 * i + 0] = bits(powerHeuristic(1, f[i], 1, g[i]));
    out[3 * i + 1] = bits(cosineHemispherePdf(vec3(0.0f, 0.0f, f[i])));
    out[3 * i + 2] = isTransimissionBSDF(bsdfType(handles[i])) ? 1u : 0u;
  }
}

}  // extern "C"
