// pt_shading.h -- per-vertex shading arithmetic of the wavefront tracer:
// RNG, shading frame, BSDF sample/eval for the eight material types, triangle
// light sampling.  Device code (gfx950); GSP_HD also lets tests/emu compile it
// for the host.
//
// Behaviour follows the reference shaders (S/assets/shaders/):
//   RNG                 pt_common.glsl:86-120   (bit exact, uint32)
//   Onb                 pt_common.glsl:122-151
//   sampling / Fresnel  rayhit.rchit:89-330
//   BSDFs               rayhit.rchit:341-654
//   lights              rayhit.rchit:117-153
// including its quirks (Beckmann-sampled / GGX-valued rough conductor,
// randUniform() in [0,1] inclusive, eval-side 0.01 clamp of rough plastic).
#pragma once
#include "../../include/gpuspectral_pt.h"
#include "pt_math.h"

namespace gsp {

// ---- lane profile of k_shade (measurement build only: -DGSP_SHADE_PROFILE, scripts/shade_lane_profile.py) -----------------
// A REGION is a stretch of code entered and left by the same lanes of a wave; at its end the first active lane adds
// {1, lanes enabled, shader cycles, cycles x lanes} to the block's LDS table (flushed to g_shade_profile by k_shade).
// Regions nest; cycles are the wave's wall time in the region (its own issue + whatever it waited for).
enum ShadeRegion {
  PR_TILE = 0, PR_LOADSORT, PR_FETCH, PR_VERTEX, PR_PACKET, PR_SAMPLE, PR_LIGHT, PR_EVAL, PR_TAIL, PR_COMPACT, PR_WRITE, PR_MISS,
  PR_SAMPLE_T0 = 16,  // + BSDF type (8)
  PR_EVAL_T0 = 24,    // + BSDF type (8)
  PR_TYPES_IN_WAVE = 32,  // + distinct sort keys in the wave (1..10): entries only
  PR_COUNT = 48
};
#if defined(GSP_SHADE_PROFILE) && defined(__HIPCC__)
__device__ __forceinline__ unsigned long long* gsp_prof_table() {
  __shared__ unsigned long long s_prof[PR_COUNT * 4];
  return s_prof;
}
__device__ __forceinline__ void gsp_prof_end(int id, unsigned long long t0) {
  const unsigned long long dt = __builtin_readcyclecounter() - t0;
  const unsigned long long m = __ballot(1);
  const unsigned long long n = (unsigned long long)__popcll(m);
  if ((int)(threadIdx.x & 63) == __ffsll(m) - 1) {
    unsigned long long* p = gsp_prof_table() + 4 * id;
    atomicAdd(p, 1ull);
    atomicAdd(p + 1, n);
    atomicAdd(p + 2, dt);
    atomicAdd(p + 3, dt * n);
  }
}
#define GSP_PROF_BEGIN(id) const unsigned long long prof_t0_##id = __builtin_readcyclecounter()
#define GSP_PROF_END(id) gsp_prof_end(id, prof_t0_##id)
#define GSP_PROF_END_T(id, base, type) gsp_prof_end((base) + (int)(type), prof_t0_##id)
#else
#define GSP_PROF_BEGIN(id) ((void)0)
#define GSP_PROF_END(id) ((void)0)
#define GSP_PROF_END_T(id, base, type) ((void)0)
#endif

// ---- RNG -------------------------------------------------------------------
GSP_HD uint32_t pcg_output(uint32_t state) {
  uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
  return (word >> 22u) ^ word;
}
GSP_HD uint32_t pcg_next(uint32_t& state) {  // randPcg, pt_common.glsl:87-93
  uint32_t s = state;
  state = s * 747796405u + 2891336453u;
  return pcg_output(s);
}
GSP_HD uint32_t pcg_hash(uint32_t v) { return pcg_output(v * 747796405u + 2891336453u); }  // :95-100
GSP_HD float u01(uint32_t r) { return (float)r * 2.3283064365386962890625e-10f; }             // r * 2^-32, :102-104
GSP_HD float rand_uniform(uint32_t& state) { return u01(pcg_next(state)); }
GSP_HD uint32_t tea(uint32_t v0, uint32_t v1) {  // :106-120
  uint32_t s0 = 0;
#pragma unroll
  for (int n = 0; n < 4; n++) {
    s0 += 0x9e3779b9u;
    v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
    v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
  }
  return v0;
}

// ---- shading frame (pt_common.glsl:122-151) --------------------------------
struct Frame {
  f3 t, b, n;
};
GSP_HD Frame make_frame(f3 nrm) {
  Frame f;
  f.n = normalize(nrm);
  f.b = gabs(f.n.x) > gabs(f.n.z) ? mk3(-f.n.y, f.n.x, 0.0f) : mk3(0.0f, -f.n.z, f.n.y);
  f.b = normalize(f.b);
  f.t = cross(f.b, f.n);
  return f;
}
GSP_HD f3 to_local(const Frame& f, f3 v) { return mk3(dot(v, f.t), dot(v, f.b), dot(v, f.n)); }
GSP_HD f3 to_world(const Frame& f, f3 v) { return (f.t * v.x + f.b * v.y) + f.n * v.z; }

// ---- sampling helpers ----------------------------------------------------------
// cosine-weighted hemisphere through Shirley's concentric map, rayhit.rchit:89-111
GSP_HD f3 sample_cosine_hemisphere(uint32_t& rng) {
  float sx = rand_uniform(rng);
  float sy = rand_uniform(rng);
  float ux = 2.0f * sx - 1.0f;
  float uy = 2.0f * sy - 1.0f;
  float dx = 0.0f, dy = 0.0f;
  if (!(ux == 0.0f && uy == 0.0f)) {
    float r, th;
    if (gabs(ux) > gabs(uy)) {
      r = ux;
      th = (kPi / 4.0f) * (uy / ux);
    } else {
      r = uy;
      th = kPi / 2.0f - (kPi / 4.0f) * (ux / uy);
    }
    float s, c;
    det_sincosf(th, s, c);
    dx = r * c;
    dy = r * s;
  }
  float z = gsqrt(gmax(0.0f, (1.0f - dx * dx) - dy * dy));
  return mk3(dx, dy, z);
}
GSP_HD float cosine_pdf(f3 w) { return gmax(gabs(w.z) / kPi, 0.000001f); }  // :113-115

// Beckmann half vector, rayhit.rchit:155-166
GSP_HD f3 sample_half_beckmann(uint32_t& rng, float alpha) {
  float ux = rand_uniform(rng);
  float uy = rand_uniform(rng);
  float phi = (2.0f * kPi) * ux;
  float lg = det_logf(1.0f - uy);
  if (gisinf(lg)) lg = 0.0f;
  float tan2 = (-alpha * alpha) * lg;
  float cost = 1.0f / gsqrt(1.0f + tan2);
  float sint = gsqrt(gmax(0.0f, 1.0f - cost * cost));
  float sp, cp;
  det_sincosf(phi, sp, cp);
  f3 wh = mk3(cp * sint, sp * sint, cost);
  if (wh.z <= 0.0f) wh = wh * -1.0f;  // :511-513
  return wh;
}
GSP_HD float tan2_theta(f3 w, float& cos2) {
  cos2 = w.z * w.z;
  return (w.x * w.x + w.y * w.y) / cos2;
}
GSP_HD float beckmann_d(f3 wh, float alpha) {  // :177-183
  float cos2;
  float tan2 = tan2_theta(wh, cos2);
  float a = det_expf(-tan2 / (alpha * alpha));
  float b = ((kPi * alpha) * alpha) * cos2 * cos2;
  return a / b;
}
GSP_HD float ggx_d(f3 wh, float alpha) {  // :185-192
  float cos2;
  float tan2 = tan2_theta(wh, cos2);
  if (gisinf(tan2)) return 0.0f;
  float b = 1.0f + tan2 / (alpha * alpha);
  float a = ((((kPi * alpha) * alpha) * cos2) * cos2) * b * b;
  return 1.0f / a;
}
GSP_HD float ggx_lambda(f3 w, float alpha) {  // :194-200
  float cos2;
  float tan2 = tan2_theta(w, cos2);
  if (gisinf(tan2)) return 0.0f;
  return 0.5f * (-1.0f + gsqrt(1.0f + (alpha * alpha) * tan2));
}
GSP_HD float ggx_g(f3 wo, f3 wi, float alpha) {  // :202-204
  return 1.0f / ((1.0f + ggx_lambda(wo, alpha)) + ggx_lambda(wi, alpha));
}
GSP_HD float power_heuristic(float fPdf, float gPdf) {  // :206-210 with nf = ng = 1
  float f = 1.0f * fPdf;
  float g = 1.0f * gPdf;
  return (f * f) / (f * f + g * g);
}

struct q4s {  // 16-byte quad (layout of pt_trace.h's q4; this header comes first)
  float x, y, z, w;
};

// ---- Fresnel terms ------------------------------------------------------------
GSP_HD float fresnel_polarized(float no, float cosTho, float nt, float cosTht) {  // :218-226
  float a = nt * cosTho - no * cosTht;
  float ad = nt * cosTho + no * cosTht;
  float b = no * cosTho - nt * cosTht;
  float bd = no * cosTho + nt * cosTht;
  float A = (a * a) / (ad * ad);
  float B = (b * b) / (bd * bd);
  return 0.5f * (A + B);
}
// common tail of the two dielectric overloads (:228-247): sinTho given
GSP_HD float fresnel_from_sin(float sinTho, float cosTho, float no, float nt) {
  float sqrtTerm = 1.0f - ((no * no) / (nt * nt)) * (sinTho * sinTho);
  if (sqrtTerm <= 0.0f) return 1.0f;
  return fresnel_polarized(no, cosTho, nt, gsqrt(sqrtTerm));
}
// ... with r2 = (no * no) / (nt * nt) taken from the record's derived quad (bake_bsdf): the same bits, no division
GSP_HD float fresnel_from_sin_r2(float sinTho, float cosTho, float no, float nt, float r2) {
  float sqrtTerm = 1.0f - r2 * (sinTho * sinTho);
  if (sqrtTerm <= 0.0f) return 1.0f;
  return fresnel_polarized(no, cosTho, nt, gsqrt(sqrtTerm));
}
GSP_HD float fresnel_wo_r2(f3 wo, float no, float nt, float r2) {
  return fresnel_from_sin_r2(gsqrt(gmax(wo.x * wo.x + wo.y * wo.y, 0.0f)), gabs(wo.z), no, nt, r2);
}
GSP_HD float fresnel_cos_r2(float cosTho, float no, float nt, float r2) {
  return fresnel_from_sin_r2(gsqrt(gmax(1.0f - cosTho * cosTho, 0.0f)), cosTho, no, nt, r2);
}
GSP_HD float fresnel_wo(f3 wo, float no, float nt) {  // :228-237
  return fresnel_from_sin(gsqrt(gmax(wo.x * wo.x + wo.y * wo.y, 0.0f)), gabs(wo.z), no, nt);
}
GSP_HD float fresnel_cos(float cosTho, float no, float nt) {  // :239-247
  return fresnel_from_sin(gsqrt(gmax(1.0f - cosTho * cosTho, 0.0f)), cosTho, no, nt);
}
GSP_HD f3 fresnel_conductor(f3 eta, f3 k, float c) {  // FresnelDieletricConductor :269-288
  float c2 = c * c;
  float s2 = 1.0f - c2;
  f3 e2 = eta * eta;
  f3 k2 = k * k;
  f3 t0 = (e2 - k2) - s2;
  f3 a2b2 = sqrt3(t0 * t0 + (4.0f * e2) * k2);
  f3 t1 = a2b2 + c2;
  f3 a = sqrt3(0.5f * (a2b2 + t0));
  f3 t2 = (2.0f * a) * c;
  f3 Rs = (t1 - t2) / (t1 + t2);
  f3 t3 = c2 * a2b2 + s2 * s2;
  f3 t4 = t2 * s2;
  f3 Rp = (Rs * (t3 - t4)) / (t3 + t4);
  return 0.5f * (Rp + Rs);
}
GSP_HD float coupled_diffuse_k(float R0) { return 21.0f / ((20.0f * kPi) * (1.0f - R0)); }  // :302, record-only: bake_bsdf
GSP_HD float coupled_diffuse_with(float k, float cosTho, float cosThi) {
  float a = 1.0f - cosTho;
  float b = 1.0f - cosThi;
  float a5 = a * a * a * a * a;
  float b5 = b * b * b * b * b;
  return (k * (1.0f - a5)) * (1.0f - b5);
}
GSP_HD float coupled_diffuse(float R0, float cosTho, float cosThi) {  // :301-308
  float k = 21.0f / ((20.0f * kPi) * (1.0f - R0));
  float a = 1.0f - cosTho;
  float b = 1.0f - cosThi;
  float a5 = a * a * a * a * a;
  float b5 = b * b * b * b * b;
  return (k * (1.0f - a5)) * (1.0f - b5);
}
GSP_HD float fresnel_blend_diffuse(float R0, float cosTho, float cosThi) {  // :310-317
  float k = 28.0f / (23.0f * kPi);
  float a = 1.0f - 0.5f * cosTho;
  float b = 1.0f - 0.5f * cosThi;
  float a5 = a * a * a * a * a;
  float b5 = b * b * b * b * b;
  return ((k * (1.0f - R0)) * (1.0f - a5)) * (1.0f - b5);
}
GSP_HD float escape_fraction(float R0, float no, float nt) {  // internalScatterEscapeFraction :320-324
  float Re = (((kPi * 20.0f) * R0) + 1.0f) / 21.0f;
  float eta = no / nt;
  return 1.0f - (eta * eta) * (1.0f - Re);
}
GSP_HD float schlick(float R0, float cosTho) {  // :326-330
  float a = 1.0f - cosTho;
  float a5 = a * a * a * a * a;
  return R0 + a5 * (1.0f - R0);
}

// ---- BSDF records ----------------------------------------------------------------
// r05: every resident BSDF table is PRECEDED by one derived quad per record (bake_bsdf below), in reverse order -- the quad of
// record i of a table sits 16 (i + 1) bytes in front of the table's first record -- so that no kernel argument is spent on them
GSP_HD q4s derived_of(const void* table, uint32_t i) { return ((const q4s*)table)[-1 - (int32_t)i]; }
struct BsdfTables {
  const gsp_diffuse_bsdf* diffuse;
  const gsp_smooth_dielectric_bsdf* smooth_dielectric;
  const gsp_smooth_conductor_bsdf* smooth_conductor;
  const gsp_smooth_plastic_bsdf* smooth_plastic;
  const gsp_rough_conductor_bsdf* rough_conductor;
  const gsp_smooth_floor_bsdf* smooth_floor;
  const gsp_rough_floor_bsdf* rough_floor;
  const gsp_rough_plastic_bsdf* rough_plastic;
};

struct BsdfResult {
  f3 f;  // BSDF value
  float pdf;
  bool delta;
};
// What sampleBSDF has computed from (record, wo) alone and evalBSDF of the SAME vertex would compute again, bit for bit:
// the rough conductor's Fresnel term (rayhit.rchit:509 = :523, ~200 instructions with its six square roots and six
// divisions), the smooth plastic's Fresnel term at wo (:463 = :495).  bsdf_sample fills it, bsdf_eval reads it (r05).
struct BsdfCarry {
  f3 v;
};

GSP_HD f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }

// ---- dormant-feature extension: textures and the environment map (include/gpuspectral_pt.h) ----------------------------
struct TextureView {
  const float* tri_uv = nullptr;           // 8 floats per triangle slot: u0 v0 u1 v1 | u2 v2 0 0
  const gsp_texture* textures = nullptr;
  const uint32_t* texels = nullptr;        // RGBA8
  const float* decode = nullptr;           // 256 floats
  uint32_t num_textures = 0;
  const float* env_texels = nullptr;       // RGBA32F
  uint32_t env_width = 0, env_height = 0;
  float env_to_local[16] = {};
};

GSP_HD int32_t wrap_index(int32_t i, int32_t n) {
  i = i % n;
  return i < 0 ? i + n : i;
}
GSP_HD float texel_coord(float t, uint32_t n) {  // continuous texel coordinate, texel centres at integers
  float x = t * (float)n - 0.5f;
  if (!(gabs(x) < 1.0e9f)) x = 0.0f;  // huge or NaN uv: texel 0
  return x;
}
GSP_HD f3 lerp4(f3 c00, f3 c10, f3 c01, f3 c11, float tx, float ty) {
  const f3 a = c00 * (1.0f - tx) + c10 * tx;
  const f3 b = c01 * (1.0f - tx) + c11 * tx;
  return a * (1.0f - ty) + b * ty;
}
GSP_HD f3 decode_texel(const float* decode, uint32_t t) {
  return mk3(decode[t & 0xffu], decode[(t >> 8) & 0xffu], decode[(t >> 16) & 0xffu]);
}
// bilinear, repeat in both directions
GSP_HD f3 sample_texture(const TextureView& T, uint32_t id, float u, float v) {
  const gsp_texture tx = T.textures[id];
  const int32_t w = (int32_t)tx.width, h = (int32_t)tx.height;
  const float x = texel_coord(u, tx.width), y = texel_coord(v, tx.height);
  const float fx = __builtin_floorf(x), fy = __builtin_floorf(y);
  const int32_t x0 = wrap_index((int32_t)fx, w), y0 = wrap_index((int32_t)fy, h);
  const int32_t x1 = x0 + 1 == w ? 0 : x0 + 1, y1 = y0 + 1 == h ? 0 : y0 + 1;
  const uint32_t* base = T.texels + tx.first_texel;
  const f3 c00 = decode_texel(T.decode, base[(int64_t)y0 * w + x0]), c10 = decode_texel(T.decode, base[(int64_t)y0 * w + x1]);
  const f3 c01 = decode_texel(T.decode, base[(int64_t)y1 * w + x0]), c11 = decode_texel(T.decode, base[(int64_t)y1 * w + x1]);
  return lerp4(c00, c10, c01, c11, x - fx, y - fy);
}
// lat-long environment radiance for world direction d: repeat in u, clamp in v
GSP_HD f3 sample_envmap(const TextureView& T, f3 d) {
  const f3 e = xform_dir(T.env_to_local, d);
  const float kInv2Pi = 0.15915494309189533577f, kInvPi = 0.31830988618379067154f;
  const float u = det_atan2f(e.x, -e.z) * kInv2Pi + 0.5f;
  const float v = 1.0f - det_atan2f(gsqrt(e.x * e.x + e.z * e.z), e.y) * kInvPi;
  const int32_t w = (int32_t)T.env_width, h = (int32_t)T.env_height;
  const float x = texel_coord(u, T.env_width), y = texel_coord(v, T.env_height);
  const float fx = __builtin_floorf(x), fy = __builtin_floorf(y);
  const int32_t x0 = wrap_index((int32_t)fx, w);
  const int32_t x1 = x0 + 1 == w ? 0 : x0 + 1;
  int32_t y0 = (int32_t)fy, y1 = y0 + 1;
  y0 = y0 < 0 ? 0 : (y0 > h - 1 ? h - 1 : y0);
  y1 = y1 < 0 ? 0 : (y1 > h - 1 ? h - 1 : y1);
  const float* p = T.env_texels;
  const f3 c00 = ld3(p + 4 * ((int64_t)y0 * w + x0)), c10 = ld3(p + 4 * ((int64_t)y0 * w + x1));
  const f3 c01 = ld3(p + 4 * ((int64_t)y1 * w + x0)), c11 = ld3(p + 4 * ((int64_t)y1 * w + x1));
  return lerp4(c00, c10, c01, c11, x - fx, y - fy);
}
GSP_HD f3 mirror(f3 wo) { return mk3(-wo.x, -wo.y, wo.z); }
GSP_HD f3 reflect_about(f3 wo, f3 wh) { return normalize(-wo + (2.0f * dot(wh, wo)) * wh); }
GSP_HD float microfacet_pdf_half(f3 wo, f3 wh, float alpha) {  // 0.5 * D_beckmann * |wh.z| / (4 |wo.wh|)
  return ((0.5f * beckmann_d(wh, alpha)) * gabs(wh.z)) / (4.0f * gabs(dot(wo, wh)));
}

// plastic substrate term shared by SmoothPlastic eval and both RoughPlastic paths
// (kD (1-Fri) (1-Fro) eta^2 / (pi (1 - kD Ri)), :500,:556,:576)
GSP_HD f3 plastic_diffuse(f3 kD, float Fri, float Fro, float eta, float Ri) {
  return ((((kD * (1.0f - Fri)) * (1.0f - Fro)) * eta) * eta) / (kPi * (1.0f - kD * Ri));
}

// dv = the record's derived quad {r2, eta, Ri} (bake_bsdf)
GSP_HD void rough_plastic_value(const gsp_rough_plastic_bsdf& b, const q4s& dv, f3 wo, f3 wi, f3& wh, f3& f, bool kd_on = false, f3 kd = f3{}) {
  float no = b.ior_out, nt = b.ior_in;
  float eta = dv.y;  // no / nt
  wh = normalize(wi + wo);
  float Fri = fresnel_cos_r2(gabs(dot(wh, wo)), no, nt, dv.x);
  float Fro = fresnel_cos_r2(gabs(dot(wh, wi)), no, nt, dv.x);
  float Ri = dv.z;   // escape_fraction(b.r0, no, nt)
  f3 spec = ((splat(Fri) * ggx_d(wh, b.alpha)) * ggx_g(wo, wi, b.alpha)) / ((4.0f * gabs(wo.z)) * gabs(wi.z));
  f = plastic_diffuse(kd_on ? kd : ld3(b.diffuse), Fri, Fro, eta, Ri) + spec;
}
GSP_HD void rough_floor_value(const gsp_rough_floor_bsdf& b, f3 wo, f3 wi, BsdfResult& r) {  // :595-601,:607-614
  f3 wh = normalize(wi + wo);
  float Fr = schlick(b.r0, gabs(dot(wo, wh)));
  f3 d = ld3(b.diffuse) * fresnel_blend_diffuse(b.r0, gabs(wo.z), gabs(wi.z));
  f3 spec = (splat(Fr) * ggx_d(wh, b.alpha)) / ((4.0f * gabs(dot(wo, wh))) * gmax(gabs(wo.z), gabs(wi.z)));
  r.pdf = microfacet_pdf_half(wo, wh, b.alpha) + 0.5f * cosine_pdf(wi);
  r.f = d + spec;
  r.delta = false;
}
// 50/50 lobe choice of the rough plastic / rough floor samplers (:533-547, :584-594)
GSP_HD f3 sample_half_or_cosine(uint32_t& rng, f3 wo, float alpha) {
  float u = rand_uniform(rng);
  if (u < 0.5f) return reflect_about(wo, sample_half_beckmann(rng, alpha));
  return sample_cosine_hemisphere(rng);
}

// sampleBSDF, rayhit.rchit:630-641.  wo, wi in the shading frame.
// index + 1 of the texture that replaces the record's kD, 0 = none (dormant-feature extension, gpuspectral_pt.h)
GSP_HD uint32_t bsdf_texture(const BsdfTables& T, uint32_t handle) {
  const uint32_t i = handle & 0xffffu;
  int32_t k = 0;
  switch (handle >> 16) {
    case GSP_BSDF_DIFFUSE: k = T.diffuse[i].has_texture; break;
    case GSP_BSDF_ROUGH_CONDUCTOR: k = T.rough_conductor[i].has_texture; break;
    case GSP_BSDF_ROUGH_PLASTIC: k = T.rough_plastic[i].has_texture; break;
    default: break;
  }
  return k > 0 ? (uint32_t)k : 0u;
}

// kd_on: `kd` (the texel at the hit) stands in for the record's kD
GSP_HD void bsdf_sample(const BsdfTables& T, uint32_t handle, uint32_t& rng, f3 wo, f3& wi, BsdfResult& r, BsdfCarry& cy,
                        bool kd_on = false, f3 kd = f3{}) {
  const uint32_t i = handle & 0xffffu;
  cy.v = splat(0.0f);
  r.f = splat(0.0f);
  r.pdf = 0.0f;
  r.delta = false;
  wi = mk3(0.0f, 0.0f, 1.0f);
  switch (handle >> 16) {
    case GSP_BSDF_DIFFUSE: {  // :341-349
      GSP_PROF_BEGIN(PR_SAMPLE_T0);
      wi = sample_cosine_hemisphere(rng);
      r.f = kd_on ? kd / kPi : ld3(T.diffuse[i].reflectance);  // (the resident record holds reflectance / pi: bake_diffuse)
      r.pdf = cosine_pdf(wi);
      GSP_PROF_END_T(PR_SAMPLE_T0, PR_SAMPLE_T0, GSP_BSDF_DIFFUSE);
    } break;
    case GSP_BSDF_SMOOTH_DIELECTRIC: {  // :362-398
      GSP_PROF_BEGIN(PR_SAMPLE_T0);
      const gsp_smooth_dielectric_bsdf b = T.smooth_dielectric[i];
      const q4s dv = derived_of(T.smooth_dielectric, i);
      bool entering = wo.z > 0.0f;
      float no = entering ? b.ior_out : b.ior_in;
      float nt = entering ? b.ior_in : b.ior_out;
      const float r2 = entering ? dv.x : dv.y;   // (no * no) / (nt * nt)
      const float eta = entering ? dv.z : dv.w;  // no / nt
      float cosTho = wo.z;
      r.delta = true;
      // refractRay (:290-299) against n = faceforward(+z, -wo, +z)
      // (GLSL's -N negates every component: the flipped normal is (-0, -0, -1), which decides
      // the sign of zero components of wt)
      const bool keep = -wo.z < 0.0f;
      float sinTho = gsqrt(gmax(wo.x * wo.x + wo.y * wo.y, 0.0f));
      float sqrtTerm = 1.0f - r2 * (sinTho * sinTho);
      if (sqrtTerm <= 0.0f) {  // total internal reflection
        wi = mirror(wo);
        r.f = 1.0f * splat(1.0f / gabs(cosTho));
        r.pdf = 1.0f;
        GSP_PROF_END_T(PR_SAMPLE_T0, PR_SAMPLE_T0, GSP_BSDF_SMOOTH_DIELECTRIC);
        break;
      }
      float cosTht = gsqrt(sqrtTerm);
      f3 n = keep ? mk3(0.0f, 0.0f, 1.0f) : mk3(-0.0f, -0.0f, -1.0f);
      f3 wt = eta * (-wo) + (eta * dot(wo, n) - cosTht) * n;
      float Fr = fresnel_polarized(no, gabs(cosTho), nt, gabs(wt.z));
      float u = rand_uniform(rng);
      if (u < Fr) {
        wi = mirror(wo);
        r.f = Fr * splat(1.0f / gabs(cosTho));
        r.pdf = Fr;
      } else {
        wi = wt;
        r.f = splat((r2 * (1.0f - Fr)) / gabs(wt.z));
        r.pdf = 1.0f - Fr;
      }
      GSP_PROF_END_T(PR_SAMPLE_T0, PR_SAMPLE_T0, GSP_BSDF_SMOOTH_DIELECTRIC);
    } break;
    case GSP_BSDF_SMOOTH_CONDUCTOR: {  // :406-418
      GSP_PROF_BEGIN(PR_SAMPLE_T0);
      const gsp_smooth_conductor_bsdf b = T.smooth_conductor[i];
      float Fr = b.ior_in == 0.0f ? 1.0f : fresnel_wo_r2(wo, b.ior_out, b.ior_in, derived_of(T.smooth_conductor, i).x);
      wi = mirror(wo);
      r.f = Fr * splat(1.0f / gabs(wo.z));
      r.delta = true;
      r.pdf = 1.0f;
      GSP_PROF_END_T(PR_SAMPLE_T0, PR_SAMPLE_T0, GSP_BSDF_SMOOTH_CONDUCTOR);
    } break;
    case GSP_BSDF_SMOOTH_PLASTIC: {  // :461-491
      GSP_PROF_BEGIN(PR_SAMPLE_T0);
      const gsp_smooth_plastic_bsdf b = T.smooth_plastic[i];
      float u = rand_uniform(rng);
      const q4s dv = derived_of(T.smooth_plastic, i);
      float no = b.ior_out, nt = b.ior_in;
      float Fri = fresnel_cos_r2(gabs(wo.z), no, nt, dv.x);
      cy.v.x = Fri;
      if (u < Fri) {
        wi = mirror(wo);
        r.f = Fri * splat(1.0f / gabs(wo.z));
        r.pdf = Fri;
        r.delta = true;
      } else {
        wi = sample_cosine_hemisphere(rng);
        float Fro = fresnel_cos_r2(gabs(wi.z), no, nt, dv.x);
        float Ri = dv.z;   // escape_fraction(b.r0, no, nt)
        float eta = dv.y;  // no / nt
        f3 kD = ld3(b.diffuse);
        // sample-side association differs from eval (:484 vs :500)
        r.f = ((((kD * eta) * eta) * (1.0f - Fri)) * (1.0f - Fro)) / (kPi * (1.0f - kD * Ri));
        r.pdf = (1.0f - Fri) * cosine_pdf(wi);
      }
      GSP_PROF_END_T(PR_SAMPLE_T0, PR_SAMPLE_T0, GSP_BSDF_SMOOTH_PLASTIC);
    } break;
    case GSP_BSDF_ROUGH_CONDUCTOR: {  // :508-520
      GSP_PROF_BEGIN(PR_SAMPLE_T0);
      const gsp_rough_conductor_bsdf b = T.rough_conductor[i];
      f3 Fr = fresnel_conductor(ld3(b.eta), ld3(b.k), gabs(wo.z));
      cy.v = Fr;
      f3 wh = sample_half_beckmann(rng, b.alpha);
      wi = reflect_about(wo, wh);
      r.f = ((((kd_on ? kd : ld3(b.reflectance)) * Fr) * ggx_d(wh, b.alpha)) * ggx_g(wo, wi, b.alpha)) /
            ((4.0f * gabs(wi.z)) * gabs(wo.z));
      r.pdf = (beckmann_d(wh, b.alpha) * gabs(wh.z)) / (4.0f * gabs(dot(wo, wh)));
      GSP_PROF_END_T(PR_SAMPLE_T0, PR_SAMPLE_T0, GSP_BSDF_ROUGH_CONDUCTOR);
    } break;
    case GSP_BSDF_SMOOTH_FLOOR: {  // :428-449
      GSP_PROF_BEGIN(PR_SAMPLE_T0);
      const gsp_smooth_floor_bsdf b = T.smooth_floor[i];
      float Fr = schlick(b.r0, gabs(wo.z));
      float u = rand_uniform(rng);
      if (u < Fr) {
        wi = mirror(wo);
        r.f = ld3(b.diffuse) * coupled_diffuse_with(derived_of(T.smooth_floor, i).x, gabs(wo.z), gabs(wi.z)) + Fr * splat(1.0f / gabs(wo.z));
        r.pdf = Fr;
        r.delta = true;
      } else {
        wi = sample_cosine_hemisphere(rng);
        r.f = ld3(b.diffuse) * coupled_diffuse_with(derived_of(T.smooth_floor, i).x, gabs(wo.z), gabs(wi.z));
        r.pdf = (1.0f - Fr) * cosine_pdf(wi);
      }
      GSP_PROF_END_T(PR_SAMPLE_T0, PR_SAMPLE_T0, GSP_BSDF_SMOOTH_FLOOR);
    } break;
    case GSP_BSDF_ROUGH_FLOOR: {  // :583-604
      GSP_PROF_BEGIN(PR_SAMPLE_T0);
      const gsp_rough_floor_bsdf b = T.rough_floor[i];
      wi = sample_half_or_cosine(rng, wo, b.alpha);
      rough_floor_value(b, wo, wi, r);
      GSP_PROF_END_T(PR_SAMPLE_T0, PR_SAMPLE_T0, GSP_BSDF_ROUGH_FLOOR);
    } break;
    case GSP_BSDF_ROUGH_PLASTIC: {  // :532-563
      GSP_PROF_BEGIN(PR_SAMPLE_T0);
      const gsp_rough_plastic_bsdf b = T.rough_plastic[i];
      wi = sample_half_or_cosine(rng, wo, b.alpha);
      f3 wh;
      rough_plastic_value(b, derived_of(T.rough_plastic, i), wo, wi, wh, r.f, kd_on, kd);
      r.pdf = microfacet_pdf_half(wo, wh, b.alpha) + 0.5f * cosine_pdf(wi);
      GSP_PROF_END_T(PR_SAMPLE_T0, PR_SAMPLE_T0, GSP_BSDF_ROUGH_PLASTIC);
    } break;
    default: break;
  }
}

// evalBSDF, rayhit.rchit:643-654
// cy: what bsdf_sample left for THIS vertex (same handle, same wo)
GSP_HD void bsdf_eval(const BsdfTables& T, uint32_t handle, f3 wo, f3 wi, BsdfResult& r, const BsdfCarry& cy, bool kd_on = false,
                      f3 kd = f3{}) {
  const uint32_t i = handle & 0xffffu;
  r.f = splat(0.0f);
  r.pdf = 0.0f;
  r.delta = false;
  switch (handle >> 16) {
    case GSP_BSDF_DIFFUSE: {  // :351-358
      GSP_PROF_BEGIN(PR_EVAL_T0);
      r.f = kd_on ? kd / kPi : ld3(T.diffuse[i].reflectance);  // (reflectance / pi: bake_diffuse)
      r.pdf = cosine_pdf(wi);
      GSP_PROF_END_T(PR_EVAL_T0, PR_EVAL_T0, GSP_BSDF_DIFFUSE);
    } break;
    case GSP_BSDF_SMOOTH_DIELECTRIC:  // :400-404
    case GSP_BSDF_SMOOTH_CONDUCTOR:   // :420-426
      r.pdf = 1.0f;
      r.delta = true;
      break;
    case GSP_BSDF_SMOOTH_PLASTIC: {  // :493-506
      GSP_PROF_BEGIN(PR_EVAL_T0);
      const gsp_smooth_plastic_bsdf b = T.smooth_plastic[i];
      float no = b.ior_out, nt = b.ior_in;
      const q4s dv = derived_of(T.smooth_plastic, i);
      float Fri = cy.v.x;  // = fresnel_cos(gabs(wo.z), no, nt), :495
      float Fro = fresnel_cos_r2(gabs(wi.z), no, nt, dv.x);
      float Ri = dv.z;     // escape_fraction(b.r0, no, nt)
      r.f = plastic_diffuse(ld3(b.diffuse), Fri, Fro, dv.y, Ri);
      r.pdf = (1.0f - Fri) * cosine_pdf(wi);
      GSP_PROF_END_T(PR_EVAL_T0, PR_EVAL_T0, GSP_BSDF_SMOOTH_PLASTIC);
    } break;
    case GSP_BSDF_ROUGH_CONDUCTOR: {  // :522-530
      GSP_PROF_BEGIN(PR_EVAL_T0);
      const gsp_rough_conductor_bsdf b = T.rough_conductor[i];
      f3 Fr = cy.v;  // = fresnel_conductor(eta, k, |wo.z|), :523
      f3 wh = normalize(wo + wi);
      r.f = (((Fr * (kd_on ? kd : ld3(b.reflectance))) * ggx_d(wh, b.alpha)) * ggx_g(wo, wi, b.alpha)) /
            ((4.0f * gabs(wi.z)) * gabs(wo.z));
      r.pdf = (beckmann_d(wh, b.alpha) * gabs(wh.z)) / (4.0f * gabs(dot(wo, wh)));
      GSP_PROF_END_T(PR_EVAL_T0, PR_EVAL_T0, GSP_BSDF_ROUGH_CONDUCTOR);
    } break;
    case GSP_BSDF_SMOOTH_FLOOR: {  // :451-458
      GSP_PROF_BEGIN(PR_EVAL_T0);
      const gsp_smooth_floor_bsdf b = T.smooth_floor[i];
      float Fr = schlick(b.r0, gabs(wo.z));
      r.f = ld3(b.diffuse) * coupled_diffuse_with(derived_of(T.smooth_floor, i).x, gabs(wo.z), gabs(wi.z));
      r.pdf = (1.0f - Fr) * cosine_pdf(wi);
      GSP_PROF_END_T(PR_EVAL_T0, PR_EVAL_T0, GSP_BSDF_SMOOTH_FLOOR);
    } break;
    case GSP_BSDF_ROUGH_FLOOR: {  // :606-617
      GSP_PROF_BEGIN(PR_EVAL_T0);
      rough_floor_value(T.rough_floor[i], wo, wi, r);
      GSP_PROF_END_T(PR_EVAL_T0, PR_EVAL_T0, GSP_BSDF_ROUGH_FLOOR);
    } break;
    case GSP_BSDF_ROUGH_PLASTIC: {  // :565-582
      GSP_PROF_BEGIN(PR_EVAL_T0);
      const gsp_rough_plastic_bsdf b = T.rough_plastic[i];
      f3 wh;
      rough_plastic_value(b, derived_of(T.rough_plastic, i), wo, wi, wh, r.f, kd_on, kd);
      r.pdf = (0.5f * gmax(beckmann_d(wh, b.alpha) * gabs(wh.z), 0.01f)) / (4.0f * gabs(dot(wo, wh))) +
              0.5f * cosine_pdf(wi);
      GSP_PROF_END_T(PR_EVAL_T0, PR_EVAL_T0, GSP_BSDF_ROUGH_PLASTIC);
    } break;
    default: break;
  }
}

GSP_HD bool bsdf_transmits(uint32_t handle) { return (handle >> 16) == GSP_BSDF_SMOOTH_DIELECTRIC; }  // :620-627

// ---- resident ("baked") table records (r05) -----------------------------------------------------------------------------------
// Some of what a shaded vertex computes depends on its table record ALONE: sampleLight's triangle normal and area
// (rayhit.rchit:131-134: two cross products, two square roots and a division per vertex), the diffuse BSDF's reflectance / pi
// (:345,:354: three IEEE divisions in sampleBSDF and three more in evalBSDF).  The resident copies of those two tables carry
// these values in place -- the light's normal in the unused w components of its positions and its area in radiance.w, the
// diffuse record reflectance / pi instead of reflectance -- computed ONCE per upload by k_bake_tables with the very
// expressions the shader states, so every vertex reads the bits it would have computed.  (The host keeps the records as
// the caller gave them: gsp_update_tables compares those.)
GSP_HD void bake_light(gsp_triangle_light& L) {
  const f3 v0 = ld3(L.positions[0]), v1 = ld3(L.positions[1]), v2 = ld3(L.positions[2]);
  const float A = 0.5f * gabs(length(cross(v2 - v0, v1 - v0)));  // :132
  const f3 normal = normalize(cross(v1 - v0, v2 - v0));          // :133
  L.positions[0][3] = normal.x;
  L.positions[1][3] = normal.y;
  L.positions[2][3] = normal.z;
  L.radiance[3] = A;
}
GSP_HD void bake_diffuse(gsp_diffuse_bsdf& b) {
  const f3 f = ld3(b.reflectance) / kPi;  // :345 = :354
  b.reflectance[0] = f.x;
  b.reflectance[1] = f.y;
  b.reflectance[2] = f.z;
}

// One derived quad per BSDF record: what sampleBSDF / evalBSDF compute from the record alone, with the shader's expressions
//   smooth dielectric  {r2 entering, r2 leaving, eta entering, eta leaving}: r2 = (no no) / (nt nt), eta = no / nt  (:372,:290-299,:390)
//   smooth conductor   {r2}                                                  with no = ior_out, nt = ior_in          (:228-237)
//   smooth plastic, rough plastic  {r2, eta = no / nt, Ri = internalScatterEscapeFraction(r0, no, nt)}              (:239-247,:320-324)
//   smooth floor       {k of coupledDiffuse = 21 / (20 pi (1 - R0))}                                                 (:302)
// (escape_fraction and the Fresnel helpers are defined above; rough conductor / rough floor / diffuse: nothing here)
GSP_HD q4s bake_bsdf(uint32_t type, const void* rec) {
  q4s d{0.0f, 0.0f, 0.0f, 0.0f};
  switch (type) {
    case GSP_BSDF_SMOOTH_DIELECTRIC: {
      const gsp_smooth_dielectric_bsdf& b = *(const gsp_smooth_dielectric_bsdf*)rec;
      d.x = (b.ior_out * b.ior_out) / (b.ior_in * b.ior_in);  // entering: no = ior_out, nt = ior_in
      d.y = (b.ior_in * b.ior_in) / (b.ior_out * b.ior_out);
      d.z = b.ior_out / b.ior_in;
      d.w = b.ior_in / b.ior_out;
    } break;
    case GSP_BSDF_SMOOTH_CONDUCTOR: {
      const gsp_smooth_conductor_bsdf& b = *(const gsp_smooth_conductor_bsdf*)rec;
      d.x = (b.ior_out * b.ior_out) / (b.ior_in * b.ior_in);
    } break;
    case GSP_BSDF_SMOOTH_PLASTIC: {
      const gsp_smooth_plastic_bsdf& b = *(const gsp_smooth_plastic_bsdf*)rec;
      d.x = (b.ior_out * b.ior_out) / (b.ior_in * b.ior_in);
      d.y = b.ior_out / b.ior_in;
      d.z = escape_fraction(b.r0, b.ior_out, b.ior_in);
    } break;
    case GSP_BSDF_ROUGH_PLASTIC: {
      const gsp_rough_plastic_bsdf& b = *(const gsp_rough_plastic_bsdf*)rec;
      d.x = (b.ior_out * b.ior_out) / (b.ior_in * b.ior_in);
      d.y = b.ior_out / b.ior_in;
      d.z = escape_fraction(b.r0, b.ior_out, b.ior_in);
    } break;
    case GSP_BSDF_SMOOTH_FLOOR: {
      const gsp_smooth_floor_bsdf& b = *(const gsp_smooth_floor_bsdf*)rec;
      d.x = coupled_diffuse_k(b.r0);
    } break;
    default: break;
  }
  return d;
}

// ---- light sampling (rayhit.rchit:123-153) ---------------------------------------
struct LightSample {
  f3 position;
  f3 emission;
  float pdf;
};
// Draws R (index), U (e1), U (e2) -- always three draws.  `lights` = BAKED records (bake_light);
// inv_num_lights = 1.0f / (float)num_lights (:151), formed once on the host.
GSP_HD LightSample sample_light(const gsp_triangle_light* lights, uint32_t num_lights, float inv_num_lights, uint32_t& rng, f3 pos) {
  LightSample ls;
  uint32_t r = pcg_next(rng);
  float e1 = rand_uniform(rng);
  float e2 = rand_uniform(rng);
  if (num_lights == 0) {  // modulo by zero in the reference (:148): no light, zero pdf
    ls.position = pos;
    ls.emission = splat(0.0f);
    ls.pdf = 0.0f;
    return ls;
  }
  const gsp_triangle_light* L = lights + (r % num_lights);
  float se1 = gsqrt(e1);
  float u = 1.0f - se1;
  float v = e2 * se1;
  float w = (1.0f - u) - v;
  f3 v0 = ld3(L->positions[0]);
  f3 v1 = ld3(L->positions[1]);
  f3 v2 = ld3(L->positions[2]);
  float A = L->radiance[3];                                                         // baked: 0.5 |cross(v2 - v0, v1 - v0)|
  f3 normal = mk3(L->positions[0][3], L->positions[1][3], L->positions[2][3]);     // baked: normalize(cross(v1 - v0, v2 - v0))
  f3 lightPos = (u * v0 + v * v1) + w * v2;
  f3 toL = lightPos - pos;
  float ldist = length(toL);
  f3 l = normalize(toL);
  float c = dot(-l, normal);
  ls.position = lightPos;
  ls.emission = ld3(L->radiance) * (c > 0.0f ? 1.0f : 0.0f);
  ls.pdf = (ldist * ldist) / (gabs(c) * A);
  ls.pdf = ls.pdf * inv_num_lights;
  return ls;
}

}  // namespace gsp
