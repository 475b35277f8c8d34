// oracle/glsl_harness.cpp -- TEST INFRASTRUCTURE ONLY.  Built only where /root/reference is mounted, by
// tests/golden/make_glsl_vectors.py, into oracle/_ref/libglsl_ref.so (git-ignored).
//
// Runs the reference's own shader TEXT.  The GLSL_PART_* macros name temporary files that hold line ranges of
// S/assets/shaders/{pt_common.glsl, rayhit.rchit, raygen.rgen, miss.rmiss, shadowmiss.rmiss} after the token-level rewrites described
// in glsl_shim.h; this file supplies what a shader gets from its ENVIRONMENT -- the RenderState block and the buffer-reference
// types (pt_common.glsl:53-84, rayhit.rchit:71-87), the GL_EXT_ray_tracing built-in variables, the payloads, the storage image,
// traceRayEXT (whose traversal is the driver's: the oracle's stands in) -- and C entry points that call the extracted functions.
// No function of the reference is restated here.
#include <cstdint>
#include <cstring>

#include "../include/gpuspectral_pt.h"
#include "glsl_shim.h"

#if !defined(GLSL_PART_HANDLES) || !defined(GLSL_PART_RNG_ONB) || !defined(GLSL_PART_STRUCTS) || !defined(GLSL_PART_FUNCTIONS) || !defined(GLSL_PART_RGEN)
#error "built by tests/golden/make_glsl_vectors.py (needs /root/reference)"
#endif

namespace glsl {

#include GLSL_PART_PAYLOAD   // pt_common.glsl:4-28     HitPayload, Camera, RenderParams
#include GLSL_PART_HANDLES   // pt_common.glsl:28-42    BSDFHandle, bsdfHandle / bsdfType / bsdfIndex
#include GLSL_PART_INSTANCE  // pt_common.glsl:44-51    Instance
#include GLSL_PART_RNG_ONB   // pt_common.glsl:86-151   rngState, randPcg, pcgHash, randUniform, tea, Onb
#include GLSL_PART_STRUCTS   // rayhit.rchit:17-69      TriangleLight and the eight BSDF records

// the shader's environment: one device pointer per BSDF table (pt_common.glsl:54-66 through the reference's own BSDF.inc),
// the light table, the light count
#define BSDFDefinition(BSDFNAME, BSDFFIELD, BSDFTYPE) typedef BufferRef<BSDFNAME> BSDFNAME##Buffer;
#include "BSDF.inc"
#undef BSDFDefinition
typedef BufferRef<TriangleLight> TriangleLightBuffer;
typedef BufferRef<Instance> InstanceBuffer;
// `PositionBuffer(instance.positionBuffer).positions[i]` (pt_common.glsl:74-82): a typed view of the 64-bit address in a uvec2
static inline uvec2 pack_address(const void* p) {
  const unsigned long long a = (unsigned long long)(uintptr_t)p;
  return uvec2{(uint)a, (uint)(a >> 32)};
}
static inline const void* unpack_address(uvec2 u) { return (const void*)(uintptr_t)(((unsigned long long)u.y << 32) | u.x); }
struct PositionBuffer {
  const vec3* positions;
  explicit PositionBuffer(uvec2 p) : positions((const vec3*)unpack_address(p)) {}
};
struct NormalBuffer {
  const vec3* normals;
  explicit NormalBuffer(uvec2 p) : normals((const vec3*)unpack_address(p)) {}
};
struct Scene {
  const Instance* instances = nullptr;
#define BSDFDefinition(BSDFNAME, BSDFFIELD, BSDFTYPE) const BSDFNAME* BSDFFIELD##s = nullptr;
#include "BSDF.inc"
#undef BSDFDefinition
  const TriangleLight* triangleLights = nullptr;
  int numLights = 0;
};
struct RenderState {  // pt_common.glsl:68-72 with this file's Scene
  Camera camera;
  Scene scene;
  RenderParams params;
};
static RenderState renderState;

#include GLSL_PART_FUNCTIONS  // rayhit.rchit:89-654    samplers, Fresnel terms, 8 BSDF pairs, dispatch, light sampling

// ---- the shaders' main() functions (r06, second step): closest hit, the two miss shaders, ray generation --------------------------
// What the ray-tracing pipeline hands a shader: the payloads (rayhit.rchit:13-15, raygen.rgen:18), the built-in variables of
// GL_EXT_ray_tracing, the storage image, and traceRayEXT -- which is the DRIVER's traversal (vendor-opaque: the oracle's BVH and
// triangle test stand in for it, through a callback into liboracle_pt.so) followed by the shader the hit / miss selects.
static HitPayload prd;
static bool shadowed;
static vec3 attribs;
static int gl_PrimitiveID, gl_InstanceID;
static mat4x3 gl_ObjectToWorldEXT;
static vec3 gl_WorldRayOriginEXT, gl_WorldRayDirectionEXT;
static float gl_HitTEXT;
static uvec3 gl_LaunchIDEXT, gl_LaunchSizeEXT;
static const uint gl_RayFlagsOpaqueEXT = 1u, gl_RayFlagsTerminateOnFirstHitEXT = 4u, gl_RayFlagsSkipClosestHitShaderEXT = 8u;
static const int topLevelAS = 0;
struct Image2D {
  float* rgba = nullptr;
  uint width = 0;
};
static Image2D image;
static inline vec4 imageLoad(const Image2D& im, ivec2 p) {
  const float* q = im.rgba + 4ull * ((size_t)p.y * im.width + (size_t)p.x);
  return vec4(q[0], q[1], q[2], q[3]);
}
static inline void imageStore(const Image2D& im, ivec2 p, vec4 v) {
  float* q = im.rgba + 4ull * ((size_t)p.y * im.width + (size_t)p.x);
  q[0] = v.x, q[1] = v.y, q[2] = v.z, q[3] = v.w;
}
static void traceRayEXT(int, uint rayFlags, uint cullMask, uint sbtOffset, uint sbtStride, uint missIndex, vec3 origin, float tMin, vec3 direction,
                        float tMax, int payload);

#define main rchit_main
#include GLSL_PART_RCHIT_MAIN  // rayhit.rchit:656-797   NEE, isvalid, main()
#undef main
#define main miss_main
#include GLSL_PART_MISS        // miss.rmiss:15-18
#undef main
#define main shadowmiss_main
#include GLSL_PART_SHADOWMISS  // shadowmiss.rmiss:6-9
#undef main
#define main rgen_main
#include GLSL_PART_RGEN        // raygen.rgen:20-108     rayDir, MAX_DEPTH, main()
#undef main

// the driver's part of traceRayEXT
struct HitRecord {
  float t, u, v;
  int32_t prim;
};
typedef int (*trace_fn)(void* oracle, const float* ray8, uint64_t n, int any_hit, void* hits);
static trace_fn g_trace = nullptr;
static void* g_oracle = nullptr;
static const gsp_instance* g_instances = nullptr;
static uint32_t g_num_instances = 0;
static const uint32_t* g_first_tri = nullptr;  // first global triangle id of each instance (+ the total at the end)
static unsigned long long g_ext_rays = 0, g_shadow_rays = 0;

static void traceRayEXT(int, uint rayFlags, uint, uint, uint, uint missIndex, vec3 origin, float tMin, vec3 direction, float tMax, int payload) {
  const float ray[8] = {origin.x, origin.y, origin.z, tMin, direction.x, direction.y, direction.z, tMax};
  HitRecord hr;
  if (payload == 2) {  // the shadow ray (rayhit.rchit:737-748): TerminateOnFirstHit | SkipClosestHitShader, miss shader 1
    ++g_shadow_rays;
    g_trace(g_oracle, ray, 1, 1, &hr);
    if (hr.prim < 0) shadowmiss_main();
    (void)rayFlags, (void)missIndex;
    return;
  }
  ++g_ext_rays;
  g_trace(g_oracle, ray, 1, 0, &hr);
  if (hr.prim < 0) {
    miss_main();
    return;
  }
  uint32_t inst = 0;
  while (inst + 1 < g_num_instances && (uint32_t)hr.prim >= g_first_tri[inst + 1]) ++inst;
  gl_InstanceID = (int)inst;
  gl_PrimitiveID = (int)((uint32_t)hr.prim - g_first_tri[inst]);
  memcpy(gl_ObjectToWorldEXT.a, g_instances[inst].transform, 64);
  gl_WorldRayOriginEXT = origin;
  gl_WorldRayDirectionEXT = direction;
  gl_HitTEXT = hr.t;
  attribs = vec3(hr.u, hr.v, 0.0f);
  rchit_main();
}

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

// ---- whole frames through the shaders' main() functions -----------------------------------------------------------------------------
// `sc`: the scene as it crosses the C ABI; transform_inv_t: 16 floats per instance = glm::inverse(glm::transpose(M)) as the host
// computes it (S/renderer/PathTracer.cpp:62: host code, taken from the oracle's restatement of it); trace / oracle: the stand-in for
// the driver's acceleration structure (oracle_trace of liboracle_pt.so and its handle).
static Instance* g_inst_records = nullptr;
static uint32_t* g_first = nullptr;
void glsl_set_scene(const gsp_scene_desc* sc, const float* transform_inv_t, void* trace, void* oracle) {
  glsl_set_tables(sc);
  g_trace = (trace_fn)trace;
  g_oracle = oracle;
  g_instances = sc->instances;
  g_num_instances = sc->num_instances;
  g_inst_records = new Instance[sc->num_instances ? sc->num_instances : 1];
  g_first = new uint32_t[sc->num_instances + 1];
  uint32_t first = 0;
  for (uint32_t i = 0; i < sc->num_instances; ++i) {
    const gsp_instance& in = sc->instances[i];
    Instance& r = g_inst_records[i];
    memcpy(r.transformInvT.a, transform_inv_t + 16ull * i, 64);
    r.positionBuffer = pack_address(sc->positions + 3ull * in.first_vertex);  // PathTracer.cpp:63-64: the mesh's own buffers
    r.normalBuffer = pack_address(sc->normals + 3ull * in.first_vertex);
    r.emission = vec4(in.emission[0], in.emission[1], in.emission[2], 0.0f);  // PathTracer.cpp:65: vec4(emission, 0)
    r.bsdf = in.bsdf;
    r.twofaced = in.twofaced;
    g_first[i] = first;
    first += in.vertex_count / 3;
  }
  g_first[sc->num_instances] = first;
  g_first_tri = g_first;
  renderState.scene.instances = g_inst_records;
  memcpy(renderState.camera.view.a, sc->camera.to_world, 64);                  // PathTracer.cpp:88-90, Camera.cpp:41-45
  renderState.camera.eye = vec4(sc->camera.to_world[12], sc->camera.to_world[13], sc->camera.to_world[14], sc->camera.to_world[15]);
  renderState.camera.fov = sc->camera.fov;
}

// spp launches of width x height invocations of raygen.rgen's main(), timestamps first_timestamp ..: accum = the storage image
void glsl_render(uint32_t width, uint32_t height, uint32_t spp, uint32_t first_timestamp, float* accum, unsigned long long* rays2) {
  image.rgba = accum;
  image.width = width;
  g_ext_rays = g_shadow_rays = 0;
  gl_LaunchSizeEXT = uvec3{width, height, 1u};
  for (uint32_t s = 0; s < spp; ++s) {
    renderState.params.timestamp = first_timestamp + s;
    for (uint32_t y = 0; y < height; ++y)
      for (uint32_t x = 0; x < width; ++x) {
        gl_LaunchIDEXT = uvec3{x, y, 0u};
        rgen_main();
      }
  }
  if (rays2) rays2[0] = g_ext_rays, rays2[1] = g_shadow_rays;
}

}  // extern "C"
