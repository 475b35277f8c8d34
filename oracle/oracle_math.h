// oracle/oracle_math.h -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// float32 vector algebra with GLSL semantics, written out operation by
// operation so that evaluation order is explicit (GLSL evaluates binary
// operators left to right; this file must be compiled with -ffp-contract=off
// so that no a*b+c is fused unless fmaf() is written).
//
// Transcendentals: the reference calls GLSL built-ins (sin, cos, log, exp,
// tan) whose precision is vendor-defined (GLSL 4.60 spec 4.7.1 / Vulkan
// "Precision and Operation of SPIR-V Instructions": sin/cos absolute error
// 2^-11, log/exp 3 ULP).  The oracle fixes ONE concrete implementation for
// them (Cephes-style single precision minimax kernels, public domain, S. Moshier)
// so that a second implementation can be compared with it to the last bit.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

constexpr float kPi = 3.14159265358979323846f;  // pt_common.glsl:1 (float literal in GLSL)

struct vec3 {
  float x, y, z;
};

static inline vec3 V(float x, float y, float z) { return vec3{x, y, z}; }
static inline vec3 V(float s) { return vec3{s, s, s}; }
static inline vec3 operator+(vec3 a, vec3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline vec3 operator-(vec3 a, vec3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline vec3 operator*(vec3 a, vec3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline vec3 operator/(vec3 a, vec3 b) { return V(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline vec3 operator*(vec3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline vec3 operator*(float s, vec3 a) { return V(s * a.x, s * a.y, s * a.z); }
static inline vec3 operator/(vec3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
static inline vec3 operator+(vec3 a, float s) { return V(a.x + s, a.y + s, a.z + s); }
static inline vec3 operator-(vec3 a, float s) { return V(a.x - s, a.y - s, a.z - s); }
static inline vec3 operator-(float s, vec3 a) { return V(s - a.x, s - a.y, s - a.z); }
static inline vec3 operator-(vec3 a) { return V(-a.x, -a.y, -a.z); }

// GLSL dot / cross, left-to-right sums.
static inline float dot(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline vec3 cross(vec3 a, vec3 b) {
  return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float length(vec3 a) { return sqrtf(dot(a, a)); }
// GLSL normalize(): v * (1/length(v)) (one IEEE divide, three multiplies).
static inline vec3 normalize(vec3 a) {
  float inv = 1.0f / sqrtf(dot(a, a));
  return V(a.x * inv, a.y * inv, a.z * inv);
}
static inline vec3 vsqrt(vec3 a) { return V(sqrtf(a.x), sqrtf(a.y), sqrtf(a.z)); }
// GLSL min/max/abs/clamp (min(x,y) = y<x ? y : x ; max(x,y) = x<y ? y : x)
static inline float gmin(float x, float y) { return y < x ? y : x; }
static inline float gmax(float x, float y) { return x < y ? y : x; }
static inline float gabs(float x) { return fabsf(x); }
static inline float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }
// GLSL faceforward(N, I, Nref): dot(Nref, I) < 0 ? N : -N
static inline vec3 faceforward(vec3 N, vec3 I, vec3 Nref) { return dot(Nref, I) < 0.0f ? N : -N; }
static inline bool gisinf(float x) { return std::isinf(x); }
static inline bool gisnan(float x) { return std::isnan(x); }

// glm::mat4 in glm memory order: m[c][r] = a[4*c + r].
struct mat4 {
  float a[16];
};
// GLSL mat4 * vec4(p, 1): columns scaled and summed left to right
// (gl_ObjectToWorldEXT * vec4(pos, 1.0), rayhit.rchit:679-681).
static inline vec3 xform_point(const float* m, vec3 p) {
  return V(((m[0] * p.x + m[4] * p.y) + m[8] * p.z) + m[12],
           ((m[1] * p.x + m[5] * p.y) + m[9] * p.z) + m[13],
           ((m[2] * p.x + m[6] * p.y) + m[10] * p.z) + m[14]);
}
// GLSL mat4 * vec4(n, 0) (instance.transformInvT * vec4(normal, 0.0), rayhit.rchit:686-688).
static inline vec3 xform_dir(const float* m, vec3 n) {
  return V((m[0] * n.x + m[4] * n.y) + m[8] * n.z, (m[1] * n.x + m[5] * n.y) + m[9] * n.z,
           (m[2] * n.x + m[6] * n.y) + m[10] * n.z);
}

// glm::inverse(glm::transpose(M)) -- PathTracer.cpp:62.  Cofactor expansion
// in the order glm/detail/func_matrix.inl (compute_inverse<4,4,...>) uses.
static inline void mat4_transpose(const float* m, float* o) {
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) o[4 * c + r] = m[4 * r + c];
}
static inline void mat4_inverse(const float* a, float* o) {
#define M_(c, r) a[4 * (c) + (r)]
  float Coef00 = M_(2, 2) * M_(3, 3) - M_(3, 2) * M_(2, 3);
  float Coef02 = M_(1, 2) * M_(3, 3) - M_(3, 2) * M_(1, 3);
  float Coef03 = M_(1, 2) * M_(2, 3) - M_(2, 2) * M_(1, 3);
  float Coef04 = M_(2, 1) * M_(3, 3) - M_(3, 1) * M_(2, 3);
  float Coef06 = M_(1, 1) * M_(3, 3) - M_(3, 1) * M_(1, 3);
  float Coef07 = M_(1, 1) * M_(2, 3) - M_(2, 1) * M_(1, 3);
  float Coef08 = M_(2, 1) * M_(3, 2) - M_(3, 1) * M_(2, 2);
  float Coef10 = M_(1, 1) * M_(3, 2) - M_(3, 1) * M_(1, 2);
  float Coef11 = M_(1, 1) * M_(2, 2) - M_(2, 1) * M_(1, 2);
  float Coef12 = M_(2, 0) * M_(3, 3) - M_(3, 0) * M_(2, 3);
  float Coef14 = M_(1, 0) * M_(3, 3) - M_(3, 0) * M_(1, 3);
  float Coef15 = M_(1, 0) * M_(2, 3) - M_(2, 0) * M_(1, 3);
  float Coef16 = M_(2, 0) * M_(3, 2) - M_(3, 0) * M_(2, 2);
  float Coef18 = M_(1, 0) * M_(3, 2) - M_(3, 0) * M_(1, 2);
  float Coef19 = M_(1, 0) * M_(2, 2) - M_(2, 0) * M_(1, 2);
  float Coef20 = M_(2, 0) * M_(3, 1) - M_(3, 0) * M_(2, 1);
  float Coef22 = M_(1, 0) * M_(3, 1) - M_(3, 0) * M_(1, 1);
  float Coef23 = M_(1, 0) * M_(2, 1) - M_(2, 0) * M_(1, 1);
  float Fac0[4] = {Coef00, Coef00, Coef02, Coef03};
  float Fac1[4] = {Coef04, Coef04, Coef06, Coef07};
  float Fac2[4] = {Coef08, Coef08, Coef10, Coef11};
  float Fac3[4] = {Coef12, Coef12, Coef14, Coef15};
  float Fac4[4] = {Coef16, Coef16, Coef18, Coef19};
  float Fac5[4] = {Coef20, Coef20, Coef22, Coef23};
  float Vec0[4] = {M_(1, 0), M_(0, 0), M_(0, 0), M_(0, 0)};
  float Vec1[4] = {M_(1, 1), M_(0, 1), M_(0, 1), M_(0, 1)};
  float Vec2[4] = {M_(1, 2), M_(0, 2), M_(0, 2), M_(0, 2)};
  float Vec3[4] = {M_(1, 3), M_(0, 3), M_(0, 3), M_(0, 3)};
  const float SignA[4] = {+1, -1, +1, -1};
  const float SignB[4] = {-1, +1, -1, +1};
  float Inv[4][4];
  for (int i = 0; i < 4; ++i) {
    float Inv0 = (Vec1[i] * Fac0[i] - Vec2[i] * Fac1[i]) + Vec3[i] * Fac2[i];
    float Inv1 = (Vec0[i] * Fac0[i] - Vec2[i] * Fac3[i]) + Vec3[i] * Fac4[i];
    float Inv2 = (Vec0[i] * Fac1[i] - Vec1[i] * Fac3[i]) + Vec3[i] * Fac5[i];
    float Inv3 = (Vec0[i] * Fac2[i] - Vec1[i] * Fac4[i]) + Vec2[i] * Fac5[i];
    Inv[0][i] = Inv0 * SignA[i];
    Inv[1][i] = Inv1 * SignB[i];
    Inv[2][i] = Inv2 * SignA[i];
    Inv[3][i] = Inv3 * SignB[i];
  }
  float Row0[4] = {Inv[0][0], Inv[1][0], Inv[2][0], Inv[3][0]};
  float Dot0[4] = {M_(0, 0) * Row0[0], M_(0, 1) * Row0[1], M_(0, 2) * Row0[2], M_(0, 3) * Row0[3]};
  float Dot1 = (Dot0[0] + Dot0[1]) + (Dot0[2] + Dot0[3]);
  float OneOverDeterminant = 1.0f / Dot1;
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) o[4 * c + r] = Inv[c][r] * OneOverDeterminant;
#undef M_
}

// ---------------------------------------------------------------------------
// Deterministic single-precision transcendentals (see header comment).
// Only +,-,*,fmaf, integer ops and rintf: identical on any IEEE-754 machine.
// ---------------------------------------------------------------------------
static inline uint32_t f2u(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  return u;
}
static inline float u2f(uint32_t u) {
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

// sin and cos of x, |x| <~ 1e4 (used for |x| <= 2*pi).  Cody-Waite reduction by
// pi/2, Cephes sinf/cosf kernels on [-pi/4, pi/4].
static inline void det_sincosf(float x, float* s_out, float* c_out) {
  const float kTwoOverPi = 0.63661977236758134308f;
  const float kPio2Hi = 1.57079637050628662109375f;       // float(pi/2)
  const float kPio2Lo = -4.37113900018624283e-8f;         // pi/2 - float(pi/2)
  float k = rintf(x * kTwoOverPi);
  float r = fmaf(-k, kPio2Hi, x);
  r = fmaf(-k, kPio2Lo, r);
  float z = r * r;
  float ps = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
  ps = fmaf(ps, z, -1.6666654611e-1f);
  float s = fmaf(ps * z, r, r);
  float pc = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
  pc = fmaf(pc, z, 4.166664568298827e-2f);
  float c = fmaf(pc * z, z, fmaf(-0.5f, z, 1.0f));
  int q = ((int)k) & 3;
  float ss = (q & 1) ? c : s;
  float cc = (q & 1) ? s : c;
  if (q == 1 || q == 2) cc = -cc;
  if (q >= 2) ss = -ss;
  *s_out = ss;
  *c_out = cc;
}
static inline float det_sinf(float x) {
  float s, c;
  det_sincosf(x, &s, &c);
  return s;
}
static inline float det_cosf(float x) {
  float s, c;
  det_sincosf(x, &s, &c);
  return c;
}

// natural log.  log(0) = -inf, log(x<0) = NaN, log(inf) = inf.  Cephes logf.
static inline float det_logf(float x) {
  if (x != x) return x;
  if (x < 0.0f) return u2f(0x7fc00000u);
  if (x == 0.0f) return u2f(0xff800000u);
  uint32_t bits = f2u(x);
  if (bits == 0x7f800000u) return x;
  int e = 0;
  if (bits < 0x00800000u) {  // subnormal: scale into the normal range
    x = x * 8388608.0f;
    bits = f2u(x);
    e = -23;
  }
  e += (int)((bits >> 23) & 0xffu) - 126;
  float m = u2f((bits & 0x007fffffu) | 0x3f000000u);  // [0.5, 1)
  float f;
  if (m < 0.70710678118654752440f) {
    e -= 1;
    f = (m + m) - 1.0f;
  } else {
    f = m - 1.0f;
  }
  float z = f * f;
  float p = fmaf(7.0376836292e-2f, f, -1.1514610310e-1f);
  p = fmaf(p, f, 1.1676998740e-1f);
  p = fmaf(p, f, -1.2420140846e-1f);
  p = fmaf(p, f, 1.4249322787e-1f);
  p = fmaf(p, f, -1.6668057665e-1f);
  p = fmaf(p, f, 2.0000714765e-1f);
  p = fmaf(p, f, -2.4999993993e-1f);
  p = fmaf(p, f, 3.3333331174e-1f);
  float fe = (float)e;
  float y = (p * f) * z;
  y = fmaf(-2.12194440e-4f, fe, y);
  y = fmaf(-0.5f, z, y);
  float r = f + y;
  r = fmaf(0.693359375f, fe, r);
  return r;
}

// e^x.  Cephes expf kernel; results below the normal range flush to 0.
static inline float det_expf(float x) {
  if (x != x) return x;
  if (x > 88.72283905206835f) return u2f(0x7f800000u);
  if (x < -87.33654475055310898657f) return 0.0f;
  const float kLog2e = 1.44269504088896341f;
  const float kLn2Hi = 0.693359375f;
  const float kLn2Lo = -2.12194440e-4f;
  float n = rintf(x * kLog2e);
  float r = fmaf(-n, kLn2Hi, x);
  r = fmaf(-n, kLn2Lo, r);
  float z = r * r;
  float p = fmaf(1.9875691500e-4f, r, 1.3981999507e-3f);
  p = fmaf(p, r, 8.3334519073e-3f);
  p = fmaf(p, r, 4.1665795894e-2f);
  p = fmaf(p, r, 1.6666665459e-1f);
  p = fmaf(p, r, 5.0000001201e-1f);
  float y = fmaf(p, z, r) + 1.0f;
  int ni = (int)n;
  int n1 = ni / 2;
  int n2 = ni - n1;
  float s1 = u2f((uint32_t)(n1 + 127) << 23);
  float s2 = u2f((uint32_t)(n2 + 127) << 23);
  return (y * s1) * s2;
}

// atan2 for the environment-map lookup of the dormant-feature extension (include/gpuspectral_pt.h).  No reference code
// runs it; the oracle fixes one implementation: Cephes atanf (reduction at tan(pi/8), tan(3pi/8); degree-4 minimax in
// x^2), the first-quadrant angle of (|y|, |x|) reflected by the signs.  atan2(0, 0) = 0.
static inline float det_atanf_q1(float t) {  // t >= 0
  float base = 0.0f;
  if (t > 2.414213562373095f) {
    base = 1.5707963267948966f;
    t = -1.0f / t;
  } else if (t > 0.4142135623730950f) {
    base = 0.7853981633974483f;
    t = (t - 1.0f) / (t + 1.0f);
  }
  float z = t * t;
  float p = fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
  p = fmaf(p, z, 1.99777106478e-1f);
  p = fmaf(p, z, -3.33329491539e-1f);
  return base + fmaf(p * z, t, t);
}
static inline float det_atan2f(float y, float x) {
  float ax = fabsf(x), ay = fabsf(y);
  if (!(ax > 0.0f) && !(ay > 0.0f)) return 0.0f;
  float a = ax == 0.0f ? 1.5707963267948966f : det_atanf_q1(ay / ax);
  if (x < 0.0f) a = 3.14159265358979323846f - a;
  return y < 0.0f ? -a : a;
}

}  // namespace orc
