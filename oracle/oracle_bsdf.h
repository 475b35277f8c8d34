// oracle/oracle_bsdf.h -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of the reference's shading arithmetic:
//   RNG                      S/assets/shaders/pt_common.glsl:86-120
//   Onb                      pt_common.glsl:122-151
//   sampling helpers         S/assets/shaders/rayhit.rchit:89-210
//   Fresnel / helper terms   rayhit.rchit:218-330
//   8 BSDF sample/eval pairs rayhit.rchit:341-617
//   dispatch                 rayhit.rchit:620-654
// (S/ = /root/reference/src/GPUSpectral/).  Every function keeps the GLSL
// name and evaluation order; the cited lines are the ones it follows.
#pragma once
#include "../include/gpuspectral_pt.h"
#include "oracle_math.h"

namespace orc {

// ---- bias attribution (tests/test_oracle_pins.py, tests/tools/quirk_probe.py) --------------------------------------
// The reference's estimator departs from an unbiased MIS path tracer in four documented places.  A bit set in
// g_quirks_off replaces that ONE departure by the textbook form, so that a render can be compared with the unbiased
// third-party images the reference ships (TungstenRender.png) and each departure's share of the difference measured.
// 0 (the default, and the only value oracle_render uses) is the reference as shipped.  Never set by a parity test.
enum : uint32_t {
  kQuirkNeeSampledPdf = 1u,        // rayhit.rchit:751  NEE weight takes the pdf of the SAMPLED direction wi, not of L
  kQuirkDirectWeight = 2u,         // :763-765,785-790  emitter hit weighted with the pdf of the previous vertex's light
                                   //                   SAMPLE (1 when that sample was shadowed), not of reaching this point
  kQuirkFireflyClamp = 4u,         // raygen.rgen:60-63 vertex contributions with a channel >= cutoff are dropped
  kQuirkRoughPlasticPdfFloor = 8u, // rayhit.rchit:577  max(D * cos, 0.01) in the eval pdf only
};
inline uint32_t g_quirks_off = 0;

// ---- RNG: pt_common.glsl:86-120 ------------------------------------------
struct Rng {
  uint32_t state;
};
// pt_common.glsl:87-93
static inline uint32_t randPcg(Rng& g) {
  uint32_t state = g.state;
  g.state = g.state * 747796405u + 2891336453u;
  uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
  return (word >> 22u) ^ word;
}
// pt_common.glsl:95-100
static inline uint32_t pcgHash(uint32_t v) {
  uint32_t state = v * 747796405u + 2891336453u;
  uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
  return (word >> 22u) ^ word;
}
// pt_common.glsl:102-104.  float(0xffffffffu) rounds to 2^32, so the scale is
// exactly 2^-32 and the result lies in [0, 1] INCLUSIVE (uint->float is RNE).
static inline float randUniform(Rng& g) {
  return (float)randPcg(g) * (1.0f / 4294967296.0f);
}
// pt_common.glsl:106-120
static inline uint32_t tea(uint32_t val0, uint32_t val1) {
  uint32_t v0 = val0, v1 = val1, s0 = 0;
  for (uint32_t n = 0; n < 4; n++) {
    s0 += 0x9e3779b9u;
    v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
    v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
  }
  return v0;
}

// ---- Onb: pt_common.glsl:122-151 -------------------------------------------
struct Onb {
  vec3 tangent, binormal, normal;
};
static inline Onb onbCreate(vec3 n) {
  Onb onb;
  onb.normal = normalize(n);
  if (gabs(onb.normal.x) > gabs(onb.normal.z)) {
    onb.binormal = V(-onb.normal.y, onb.normal.x, 0.0f);
  } else {
    onb.binormal = V(0.0f, -onb.normal.z, onb.normal.y);
  }
  onb.binormal = normalize(onb.binormal);
  onb.tangent = cross(onb.binormal, onb.normal);
  return onb;
}
static inline vec3 onbUntransform(const Onb& o, vec3 v) {
  return (o.tangent * v.x + o.binormal * v.y) + o.normal * v.z;
}
static inline vec3 onbTransform(const Onb& o, vec3 v) {
  return V(dot(v, o.tangent), dot(v, o.binormal), dot(v, o.normal));
}

// ---- sampling helpers: rayhit.rchit:89-115 ---------------------------------
struct vec2 {
  float x, y;
};
// rayhit.rchit:89-105 (s.x is drawn before s.y: SURVEY Appendix B)
static inline vec2 sampleConcentric(Rng& g) {
  float sx = randUniform(g);
  float sy = randUniform(g);
  vec2 u = {2.0f * sx - 1.0f, 2.0f * sy - 1.0f};
  if (u.x == 0.0f && u.y == 0.0f) return vec2{0.0f, 0.0f};
  float r, th;
  if (gabs(u.x) > gabs(u.y)) {
    r = u.x;
    th = (kPi / 4.0f) * (u.y / u.x);
  } else {
    r = u.y;
    th = kPi / 2.0f - (kPi / 4.0f) * (u.x / u.y);
  }
  float s, c;
  det_sincosf(th, &s, &c);
  return vec2{r * c, r * s};
}
// rayhit.rchit:107-111
static inline vec3 randCosineHemisphere(Rng& g) {
  vec2 u = sampleConcentric(g);
  float z = sqrtf(gmax(0.0f, (1.0f - u.x * u.x) - u.y * u.y));
  return V(u.x, u.y, z);
}
// rayhit.rchit:113-115
static inline float cosineHemispherePdf(vec3 wo) { return gmax(gabs(wo.z) / kPi, 0.000001f); }

// rayhit.rchit:155-166
static inline vec3 sampleHalf(Rng& g, float alpha) {
  float ux = randUniform(g);
  float uy = randUniform(g);
  float phi = (2.0f * kPi) * ux;
  float logSample = det_logf(1.0f - uy);
  if (gisinf(logSample)) logSample = 0.0f;
  float tan2 = (-alpha * alpha) * logSample;
  float cost = 1.0f / sqrtf(1.0f + tan2);
  float sint = sqrtf(gmax(0.0f, 1.0f - cost * cost));
  float sp, cp;
  det_sincosf(phi, &sp, &cp);
  return V(cp * sint, sp * sint, cost);
}
// rayhit.rchit:177-183
static inline float beckmannD(vec3 wh, float alpha) {
  float cos2 = wh.z * wh.z;
  float tan2 = (wh.x * wh.x + wh.y * wh.y) / cos2;
  float a = det_expf(-tan2 / (alpha * alpha));
  float b = ((kPi * alpha) * alpha) * cos2 * cos2;
  return a / b;
}
// rayhit.rchit:185-192
static inline float ggxD(vec3 wh, float alpha) {
  float cos2 = wh.z * wh.z;
  float tan2 = (wh.x * wh.x + wh.y * wh.y) / cos2;
  if (gisinf(tan2)) return 0.0f;
  float b = 1.0f + tan2 / (alpha * alpha);
  float a = ((((kPi * alpha) * alpha) * cos2) * cos2) * b * b;
  return 1.0f / a;
}
// rayhit.rchit:194-200
static inline float ggxLambda(vec3 wh, float alpha) {
  float cos2 = wh.z * wh.z;
  float tan2 = (wh.x * wh.x + wh.y * wh.y) / cos2;
  if (gisinf(tan2)) return 0.0f;
  float a = -1.0f + sqrtf(1.0f + (alpha * alpha) * tan2);
  return 0.5f * a;
}
// rayhit.rchit:202-204
static inline float ggxMask(vec3 wo, vec3 wi, float alpha) {
  return 1.0f / ((1.0f + ggxLambda(wo, alpha)) + ggxLambda(wi, alpha));
}
// rayhit.rchit:206-210
static inline float powerHeuristic(int nf, float fPdf, int ng, float gPdf) {
  float f = (float)nf * fPdf;
  float g = (float)ng * gPdf;
  return (f * f) / (f * f + g * g);
}

struct BSDFOutput {
  vec3 bsdf;
  float pdf;
  bool isDelta;
};

// ---- Fresnel & helper terms: rayhit.rchit:218-330 -------------------------
// rayhit.rchit:218-226
static inline float fresnel4(float no, float cosTho, float nt, float cosTht) {
  float a = nt * cosTho - no * cosTht;
  float ad = nt * cosTho + no * cosTht;
  float b = no * cosTho - nt * cosTht;
  float bd = no * cosTho + nt * cosTht;
  float A = (a * a) / (ad * ad);
  float B = (b * b) / (bd * bd);
  return 0.5f * (A + B);
}
// rayhit.rchit:228-237  fresnel(vec3 wo, float no, float nt)
static inline float fresnelWo(vec3 wo, float no, float nt) {
  float sinTho = sqrtf(gmax(wo.x * wo.x + wo.y * wo.y, 0.0f));
  float sqrtTerm = 1.0f - ((no * no) / (nt * nt)) * (sinTho * sinTho);
  if (sqrtTerm <= 0.0f) return 1.0f;
  float cosTht = sqrtf(sqrtTerm);
  float cosTho = gabs(wo.z);
  return fresnel4(no, cosTho, nt, cosTht);
}
// rayhit.rchit:239-247  fresnel(float cosTho, float no, float nt)
static inline float fresnelCos(float cosTho, float no, float nt) {
  float sinTho = sqrtf(gmax(1.0f - cosTho * cosTho, 0.0f));
  float sqrtTerm = 1.0f - ((no * no) / (nt * nt)) * (sinTho * sinTho);
  if (sqrtTerm <= 0.0f) return 1.0f;
  float cosTht = sqrtf(sqrtTerm);
  return fresnel4(no, cosTho, nt, cosTht);
}
// rayhit.rchit:269-288
static inline vec3 FresnelDieletricConductor(vec3 Eta, vec3 Etak, float CosTheta) {
  float CosTheta2 = CosTheta * CosTheta;
  float SinTheta2 = 1.0f - CosTheta2;
  vec3 Eta2 = Eta * Eta;
  vec3 Etak2 = Etak * Etak;
  vec3 t0 = (Eta2 - Etak2) - SinTheta2;
  vec3 a2plusb2 = vsqrt(t0 * t0 + (4.0f * Eta2) * Etak2);
  vec3 t1 = a2plusb2 + CosTheta2;
  vec3 a = vsqrt(0.5f * (a2plusb2 + t0));
  vec3 t2 = (2.0f * a) * CosTheta;
  vec3 Rs = (t1 - t2) / (t1 + t2);
  vec3 t3 = CosTheta2 * a2plusb2 + SinTheta2 * SinTheta2;
  vec3 t4 = t2 * SinTheta2;
  vec3 Rp = (Rs * (t3 - t4)) / (t3 + t4);
  return 0.5f * (Rp + Rs);
}
// rayhit.rchit:290-299
static inline bool refractRay(vec3 wo, vec3 n, float no, float nt, vec3& wt) {
  float sinTho = sqrtf(gmax(wo.x * wo.x + wo.y * wo.y, 0.0f));
  float sqrtTerm = 1.0f - ((no * no) / (nt * nt)) * (sinTho * sinTho);
  if (sqrtTerm <= 0.0f) return false;
  float cosTht = sqrtf(sqrtTerm);
  wt = (no / nt) * (-wo) + ((no / nt) * dot(wo, n) - cosTht) * n;
  return true;
}
// rayhit.rchit:301-308
static inline float coupledDiffuseTerm(float R0, float cosTho, float cosThi) {
  float k = 21.0f / ((20.0f * kPi) * (1.0f - R0));
  float a = 1.0f - cosTho;
  float b = 1.0f - cosThi;
  float a5 = a * a * a * a * a;
  float b5 = b * b * b * b * b;
  return (k * (1.0f - a5)) * (1.0f - b5);
}
// rayhit.rchit:310-317
static inline float fresnelBlendDiffuseTerm(float R0, float cosTho, float cosThi) {
  float k = 28.0f / (23.0f * kPi);
  float a = 1.0f - 0.5f * cosTho;
  float b = 1.0f - 0.5f * cosThi;
  float a5 = a * a * a * a * a;
  float b5 = b * b * b * b * b;
  return ((k * (1.0f - R0)) * (1.0f - a5)) * (1.0f - b5);
}
// rayhit.rchit:320-324
static inline float internalScatterEscapeFraction(float R0, float no, float nt) {
  float Re = (((kPi * 20.0f) * R0) + 1.0f) / 21.0f;
  float eta = no / nt;
  return 1.0f - (eta * eta) * (1.0f - Re);
}
// rayhit.rchit:326-330
static inline float schlickFresnel(float R0, float cosTho) {
  float a = 1.0f - cosTho;
  float a5 = a * a * a * a * a;
  return R0 + a5 * (1.0f - R0);
}

static inline vec3 ld3(const float* p) { return V(p[0], p[1], p[2]); }
static inline vec3 mirrorZ(vec3 wo) { return V(-wo.x, -wo.y, wo.z); }

// ---- Diffuse: rayhit.rchit:341-358 -----------------------------------------
static inline void diffuseBSDFSample(const gsp_diffuse_bsdf& b, Rng& g, vec3 wo, vec3& wi, BSDFOutput& res) {
  vec3 kD = ld3(b.reflectance);
  wi = randCosineHemisphere(g);
  res.bsdf = kD / kPi;
  res.pdf = cosineHemispherePdf(wi);
  res.isDelta = false;
}
static inline void diffuseBSDFEval(const gsp_diffuse_bsdf& b, vec3 wo, vec3 wi, BSDFOutput& res) {
  vec3 kD = ld3(b.reflectance);
  res.bsdf = kD / kPi;
  res.pdf = cosineHemispherePdf(wi);
  res.isDelta = false;
}

// ---- SmoothDielectric: rayhit.rchit:362-404 ---------------------------------
static inline void smoothDielectricBSDFSample(const gsp_smooth_dielectric_bsdf& b, Rng& g, vec3 wo, vec3& wi,
                                              BSDFOutput& res) {
  bool entering = wo.z > 0.0f;
  float no = entering ? b.ior_out : b.ior_in;
  float nt = entering ? b.ior_in : b.ior_out;
  float cosTho = wo.z;
  vec3 wt = V(0.0f);
  vec3 n = faceforward(V(0.0f, 0.0f, 1.0f), -wo, V(0.0f, 0.0f, 1.0f));
  if (!refractRay(wo, n, no, nt, wt)) {
    wi = mirrorZ(wo);
    res.bsdf = 1.0f * V(1.0f / gabs(cosTho));
    res.isDelta = true;
    res.pdf = 1.0f;
    return;
  }
  float Fr = fresnel4(no, gabs(cosTho), nt, gabs(wt.z));
  float u = randUniform(g);
  if (u < Fr) {
    wi = mirrorZ(wo);
    res.bsdf = Fr * V(1.0f / gabs(cosTho));
    res.isDelta = true;
    res.pdf = Fr;
  } else {
    wi = wt;
    res.bsdf = V((((no * no) / (nt * nt)) * (1.0f - Fr)) / gabs(wt.z));
    res.isDelta = true;
    res.pdf = 1.0f - Fr;
  }
}
static inline void smoothDielectricBSDFEval(const gsp_smooth_dielectric_bsdf&, vec3, vec3, BSDFOutput& res) {
  res.bsdf = V(0.0f);
  res.pdf = 1.0f;
  res.isDelta = true;
}

// ---- SmoothConductor: rayhit.rchit:406-426 ----------------------------------
static inline void smoothConductorBSDFSample(const gsp_smooth_conductor_bsdf& b, Rng&, vec3 wo, vec3& wi,
                                             BSDFOutput& res) {
  float no = b.ior_out;
  float nt = b.ior_in;
  float Fr = nt == 0.0f ? 1.0f : fresnelWo(wo, no, nt);
  wi = mirrorZ(wo);
  res.bsdf = Fr * V(1.0f / gabs(wo.z));
  res.isDelta = true;
  res.pdf = 1.0f;
}
static inline void smoothConductorBSDFEval(const gsp_smooth_conductor_bsdf&, vec3, vec3, BSDFOutput& res) {
  res.bsdf = V(0.0f);
  res.pdf = 1.0f;
  res.isDelta = true;
}

// ---- SmoothFloor: rayhit.rchit:428-458 --------------------------------------
static inline void smoothFloorBSDFSample(const gsp_smooth_floor_bsdf& b, Rng& g, vec3 wo, vec3& wi, BSDFOutput& res) {
  float Fr = schlickFresnel(b.r0, gabs(wo.z));
  float u = randUniform(g);
  vec3 diffuse = ld3(b.diffuse);
  if (u < Fr) {
    wi = mirrorZ(wo);
    res.bsdf = diffuse * coupledDiffuseTerm(b.r0, gabs(wo.z), gabs(wi.z)) + Fr * V(1.0f / gabs(wo.z));
    res.pdf = Fr;
    res.isDelta = true;
  } else {
    wi = randCosineHemisphere(g);
    res.bsdf = diffuse * coupledDiffuseTerm(b.r0, gabs(wo.z), gabs(wi.z));
    res.pdf = (1.0f - Fr) * cosineHemispherePdf(wi);
    res.isDelta = false;
  }
}
static inline void smoothFloorBSDFEval(const gsp_smooth_floor_bsdf& b, vec3 wo, vec3 wi, BSDFOutput& res) {
  float Fr = schlickFresnel(b.r0, gabs(wo.z));
  res.bsdf = ld3(b.diffuse) * coupledDiffuseTerm(b.r0, gabs(wo.z), gabs(wi.z));
  res.pdf = (1.0f - Fr) * cosineHemispherePdf(wi);
  res.isDelta = false;
}

// ---- SmoothPlastic: rayhit.rchit:461-506 ------------------------------------
static inline void smoothPlasticBSDFSample(const gsp_smooth_plastic_bsdf& b, Rng& g, vec3 wo, vec3& wi,
                                           BSDFOutput& res) {
  float u = randUniform(g);
  float no = b.ior_out;
  float nt = b.ior_in;
  float Fri = fresnelCos(gabs(wo.z), no, nt);
  if (u < Fri) {
    wi = mirrorZ(wo);
    res.bsdf = Fri * V(1.0f / gabs(wo.z));
    res.pdf = Fri;
    res.isDelta = true;
  } else {
    wi = randCosineHemisphere(g);
    float Fro = fresnelCos(gabs(wi.z), no, nt);
    float Ri = internalScatterEscapeFraction(b.r0, no, nt);
    float eta = no / nt;
    vec3 diffuse = ld3(b.diffuse);
    vec3 d = ((((diffuse * eta) * eta) * (1.0f - Fri)) * (1.0f - Fro)) / (kPi * (1.0f - diffuse * Ri));
    res.bsdf = d;
    res.pdf = (1.0f - Fri) * cosineHemispherePdf(wi);
    res.isDelta = false;
  }
}
static inline void smoothPlasticBSDFEval(const gsp_smooth_plastic_bsdf& b, vec3 wo, vec3 wi, BSDFOutput& res) {
  float no = b.ior_out;
  float nt = b.ior_in;
  float Fri = fresnelCos(gabs(wo.z), no, nt);
  float Fro = fresnelCos(gabs(wi.z), no, nt);
  float Ri = internalScatterEscapeFraction(b.r0, no, nt);
  float eta = no / nt;
  vec3 diffuse = ld3(b.diffuse);
  vec3 d = ((((diffuse * (1.0f - Fri)) * (1.0f - Fro)) * eta) * eta) / (kPi * (1.0f - diffuse * Ri));
  res.bsdf = d;
  res.pdf = (1.0f - Fri) * cosineHemispherePdf(wi);
  res.isDelta = false;
}

// ---- RoughConductor: rayhit.rchit:508-530 -----------------------------------
static inline void roughConductorBSDFSample(const gsp_rough_conductor_bsdf& b, Rng& g, vec3 wo, vec3& wi,
                                            BSDFOutput& res) {
  vec3 Fr = FresnelDieletricConductor(ld3(b.eta), ld3(b.k), gabs(wo.z));
  vec3 wh = sampleHalf(g, b.alpha);
  if (wh.z <= 0.0f) wh = wh * -1.0f;
  wi = normalize(-wo + (2.0f * dot(wh, wo)) * wh);
  res.bsdf = (((ld3(b.reflectance) * Fr) * ggxD(wh, b.alpha)) * ggxMask(wo, wi, b.alpha)) /
             ((4.0f * gabs(wi.z)) * gabs(wo.z));
  res.pdf = (beckmannD(wh, b.alpha) * gabs(wh.z)) / (4.0f * gabs(dot(wo, wh)));
  res.isDelta = false;
}
static inline void roughConductorBSDFEval(const gsp_rough_conductor_bsdf& b, vec3 wo, vec3 wi, BSDFOutput& res) {
  vec3 Fr = FresnelDieletricConductor(ld3(b.eta), ld3(b.k), gabs(wo.z));
  vec3 wh = normalize(wo + wi);
  res.bsdf = (((Fr * ld3(b.reflectance)) * ggxD(wh, b.alpha)) * ggxMask(wo, wi, b.alpha)) /
             ((4.0f * gabs(wi.z)) * gabs(wo.z));
  res.pdf = (beckmannD(wh, b.alpha) * gabs(wh.z)) / (4.0f * gabs(dot(wo, wh)));
  res.isDelta = false;
}

// ---- RoughPlastic: rayhit.rchit:532-582 -------------------------------------
static inline void roughPlasticTerms(const gsp_rough_plastic_bsdf& b, vec3 wo, vec3 wi, vec3& wh, vec3& bsdf) {
  float no = b.ior_out;
  float nt = b.ior_in;
  float eta = no / nt;
  wh = normalize(wi + wo);
  float Fri = fresnelCos(gabs(dot(wh, wo)), no, nt);
  float Fro = fresnelCos(gabs(dot(wh, wi)), no, nt);
  float Ri = internalScatterEscapeFraction(b.r0, no, nt);
  vec3 kD = ld3(b.diffuse);
  vec3 specular = ((V(Fri) * ggxD(wh, b.alpha)) * ggxMask(wo, wi, b.alpha)) / ((4.0f * gabs(wo.z)) * gabs(wi.z));
  vec3 d = ((((kD * (1.0f - Fri)) * (1.0f - Fro)) * eta) * eta) / (kPi * (1.0f - kD * Ri));
  bsdf = d + specular;
}
static inline void roughPlasticBSDFSample(const gsp_rough_plastic_bsdf& b, Rng& g, vec3 wo, vec3& wi,
                                          BSDFOutput& res) {
  float u = randUniform(g);
  if (u < 0.5f) {
    vec3 wh = sampleHalf(g, b.alpha);
    if (wh.z <= 0.0f) wh = wh * -1.0f;
    wi = normalize(-wo + (2.0f * dot(wh, wo)) * wh);
  } else {
    wi = randCosineHemisphere(g);
  }
  vec3 wh;
  roughPlasticTerms(b, wo, wi, wh, res.bsdf);
  res.pdf = ((0.5f * beckmannD(wh, b.alpha)) * gabs(wh.z)) / (4.0f * gabs(dot(wo, wh))) +
            0.5f * cosineHemispherePdf(wi);
  res.isDelta = false;
}
static inline void roughPlasticBSDFEval(const gsp_rough_plastic_bsdf& b, vec3 wo, vec3 wi, BSDFOutput& res) {
  vec3 wh;
  roughPlasticTerms(b, wo, wi, wh, res.bsdf);
  if (g_quirks_off & kQuirkRoughPlasticPdfFloor) {  // bias attribution only: the pdf roughPlasticBSDFSample reports (:541)
    res.pdf = ((0.5f * beckmannD(wh, b.alpha)) * gabs(wh.z)) / (4.0f * gabs(dot(wo, wh))) + 0.5f * cosineHemispherePdf(wi);
    res.isDelta = false;
    return;
  }
  res.pdf = (0.5f * gmax(beckmannD(wh, b.alpha) * gabs(wh.z), 0.01f)) / (4.0f * gabs(dot(wo, wh))) +
            0.5f * cosineHemispherePdf(wi);
  res.isDelta = false;
}

// ---- RoughFloor: rayhit.rchit:583-617 ---------------------------------------
static inline void roughFloorTerms(const gsp_rough_floor_bsdf& b, vec3 wo, vec3 wi, BSDFOutput& res) {
  vec3 wh = normalize(wi + wo);
  float Fr = schlickFresnel(b.r0, gabs(dot(wo, wh)));
  vec3 d = ld3(b.diffuse) * fresnelBlendDiffuseTerm(b.r0, gabs(wo.z), gabs(wi.z));
  vec3 specular = (V(Fr) * ggxD(wh, b.alpha)) / ((4.0f * gabs(dot(wo, wh))) * gmax(gabs(wo.z), gabs(wi.z)));
  res.pdf = ((0.5f * beckmannD(wh, b.alpha)) * gabs(wh.z)) / (4.0f * gabs(dot(wo, wh))) +
            0.5f * cosineHemispherePdf(wi);
  res.bsdf = d + specular;
  res.isDelta = false;
}
static inline void roughFloorBSDFSample(const gsp_rough_floor_bsdf& b, Rng& g, vec3 wo, vec3& wi, BSDFOutput& res) {
  float u = randUniform(g);
  if (u < 0.5f) {
    vec3 wh = sampleHalf(g, b.alpha);
    if (wh.z <= 0.0f) wh = wh * -1.0f;
    wi = normalize(-wo + (2.0f * dot(wh, wo)) * wh);
  } else {
    wi = randCosineHemisphere(g);
  }
  roughFloorTerms(b, wo, wi, res);
}
static inline void roughFloorBSDFEval(const gsp_rough_floor_bsdf& b, vec3 wo, vec3 wi, BSDFOutput& res) {
  roughFloorTerms(b, wo, wi, res);
}

// rayhit.rchit:620-627
static inline bool isTransimissionBSDF(uint32_t type) { return type == GSP_BSDF_SMOOTH_DIELECTRIC; }

// rayhit.rchit:630-654.  An out-of-range type/index is a scene error that the
// reference leaves undefined; the oracle returns a zero, non-delta response.
// `tex`: non-null when the record's hasTexture selects a texture -- the texel at the hit's uv, which stands in for kD
// (the GLSL signature passes `vec2 uv`; the lookup itself is oracle_texture.h, the dormant-feature extension)
static inline void sampleBSDF(const gsp_scene_desc& sc, uint32_t handle, Rng& g, vec3 wo, vec3& wi, BSDFOutput& res,
                              const vec3* tex = nullptr) {
  uint32_t i = handle & 0xffffu;
  res.bsdf = V(0.0f);
  res.pdf = 0.0f;
  res.isDelta = false;
  wi = V(0.0f, 0.0f, 1.0f);
  switch (handle >> 16) {
    case GSP_BSDF_DIFFUSE: {
      auto rec = sc.diffuse_bsdfs[i];
      if (tex) rec.reflectance[0] = tex->x, rec.reflectance[1] = tex->y, rec.reflectance[2] = tex->z;
      diffuseBSDFSample(rec, g, wo, wi, res);
    } break;
    case GSP_BSDF_SMOOTH_DIELECTRIC: smoothDielectricBSDFSample(sc.smooth_dielectric_bsdfs[i], g, wo, wi, res); break;
    case GSP_BSDF_SMOOTH_CONDUCTOR: smoothConductorBSDFSample(sc.smooth_conductor_bsdfs[i], g, wo, wi, res); break;
    case GSP_BSDF_SMOOTH_PLASTIC: smoothPlasticBSDFSample(sc.smooth_plastic_bsdfs[i], g, wo, wi, res); break;
    case GSP_BSDF_ROUGH_CONDUCTOR: {
      auto rec = sc.rough_conductor_bsdfs[i];
      if (tex) rec.reflectance[0] = tex->x, rec.reflectance[1] = tex->y, rec.reflectance[2] = tex->z;
      roughConductorBSDFSample(rec, g, wo, wi, res);
    } break;
    case GSP_BSDF_SMOOTH_FLOOR: smoothFloorBSDFSample(sc.smooth_floor_bsdfs[i], g, wo, wi, res); break;
    case GSP_BSDF_ROUGH_FLOOR: roughFloorBSDFSample(sc.rough_floor_bsdfs[i], g, wo, wi, res); break;
    case GSP_BSDF_ROUGH_PLASTIC: {
      auto rec = sc.rough_plastic_bsdfs[i];
      if (tex) rec.diffuse[0] = tex->x, rec.diffuse[1] = tex->y, rec.diffuse[2] = tex->z;
      roughPlasticBSDFSample(rec, g, wo, wi, res);
    } break;
    default: break;
  }
}
static inline void evalBSDF(const gsp_scene_desc& sc, uint32_t handle, vec3 wo, vec3 wi, BSDFOutput& res, const vec3* tex = nullptr) {
  uint32_t i = handle & 0xffffu;
  res.bsdf = V(0.0f);
  res.pdf = 0.0f;
  res.isDelta = false;
  switch (handle >> 16) {
    case GSP_BSDF_DIFFUSE: {
      auto rec = sc.diffuse_bsdfs[i];
      if (tex) rec.reflectance[0] = tex->x, rec.reflectance[1] = tex->y, rec.reflectance[2] = tex->z;
      diffuseBSDFEval(rec, wo, wi, res);
    } break;
    case GSP_BSDF_SMOOTH_DIELECTRIC: smoothDielectricBSDFEval(sc.smooth_dielectric_bsdfs[i], wo, wi, res); break;
    case GSP_BSDF_SMOOTH_CONDUCTOR: smoothConductorBSDFEval(sc.smooth_conductor_bsdfs[i], wo, wi, res); break;
    case GSP_BSDF_SMOOTH_PLASTIC: smoothPlasticBSDFEval(sc.smooth_plastic_bsdfs[i], wo, wi, res); break;
    case GSP_BSDF_ROUGH_CONDUCTOR: {
      auto rec = sc.rough_conductor_bsdfs[i];
      if (tex) rec.reflectance[0] = tex->x, rec.reflectance[1] = tex->y, rec.reflectance[2] = tex->z;
      roughConductorBSDFEval(rec, wo, wi, res);
    } break;
    case GSP_BSDF_SMOOTH_FLOOR: smoothFloorBSDFEval(sc.smooth_floor_bsdfs[i], wo, wi, res); break;
    case GSP_BSDF_ROUGH_FLOOR: roughFloorBSDFEval(sc.rough_floor_bsdfs[i], wo, wi, res); break;
    case GSP_BSDF_ROUGH_PLASTIC: {
      auto rec = sc.rough_plastic_bsdfs[i];
      if (tex) rec.diffuse[0] = tex->x, rec.diffuse[1] = tex->y, rec.diffuse[2] = tex->z;
      roughPlasticBSDFEval(rec, wo, wi, res);
    } break;
    default: break;
  }
}

// ---- light sampling: rayhit.rchit:117-153 ----------------------------------
struct LightOutput {
  vec3 position;
  vec3 emission;
  float pdf;
};
// rayhit.rchit:123-145
static inline LightOutput sampleTrangleLight(const gsp_triangle_light& light, Rng& g, vec3 pos) {
  float e1 = randUniform(g);
  float e2 = randUniform(g);
  float u = 1.0f - sqrtf(e1);
  float v = e2 * sqrtf(e1);
  float w = (1.0f - u) - v;
  vec3 v0 = ld3(light.positions[0]);
  vec3 v1 = ld3(light.positions[1]);
  vec3 v2 = ld3(light.positions[2]);
  float A = 0.5f * gabs(length(cross(v2 - v0, v1 - v0)));
  vec3 normal = normalize(cross(v1 - v0, v2 - v0));
  vec3 lightPos = (u * v0 + v * v1) + w * v2;
  float ldist = length(lightPos - pos);
  vec3 l = normalize(lightPos - pos);
  LightOutput res;
  res.position = lightPos;
  res.emission = ld3(light.radiance) * (dot(-l, normal) > 0.0f ? 1.0f : 0.0f);
  res.pdf = (ldist * ldist) / (gabs(dot(-l, normal)) * A);
  return res;
}
// rayhit.rchit:147-153.  numLights == 0 is a modulo by zero in the reference
// (undefined); the oracle returns a zero-pdf sample without drawing the index
// modulo, but still advances the stream by the same three draws.
static inline LightOutput sampleLight(const gsp_scene_desc& sc, Rng& g, vec3 pos) {
  uint32_t r = randPcg(g);
  if (sc.num_lights == 0) {
    (void)randUniform(g);
    (void)randUniform(g);
    LightOutput z;
    z.position = pos;
    z.emission = V(0.0f);
    z.pdf = 0.0f;
    return z;
  }
  uint32_t lightIdx = r % sc.num_lights;
  LightOutput res = sampleTrangleLight(sc.lights[lightIdx], g, pos);
  res.pdf = res.pdf * (1.0f / (float)sc.num_lights);
  return res;
}

}  // namespace orc
