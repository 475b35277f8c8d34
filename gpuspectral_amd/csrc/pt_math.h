// pt_math.h -- float32 vector algebra and transcendentals of the HIP path tracer.
//
// Every expression is written with explicit evaluation order and the file is
// compiled with -ffp-contract=off, so the only fused multiply-adds are the
// fmaf() calls spelled out here.  Results therefore do not depend on the
// compiler's contraction choices and can be compared bit for bit with the CPU
// oracle (oracle/oracle_math.h).
//
// GSP_HD functions compile for gfx950 (hipcc) and, for the logic tests under
// tests/emu, for the host (g++); the host build is never part of the product.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GSP_HD __host__ __device__ __forceinline__
#else
#include <cmath>
#include <cstring>
#define GSP_HD inline
#endif

namespace gsp {

constexpr float kPi = 3.14159265358979323846f;  // pt_common.glsl:1

struct f3 {
  float x, y, z;
};

GSP_HD f3 mk3(float x, float y, float z) {
  f3 r;
  r.x = x;
  r.y = y;
  r.z = z;
  return r;
}
GSP_HD f3 splat(float s) { return mk3(s, s, s); }
GSP_HD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
GSP_HD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
GSP_HD f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
GSP_HD f3 operator/(f3 a, f3 b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }
GSP_HD f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
GSP_HD f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
GSP_HD f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
GSP_HD f3 operator+(f3 a, float s) { return mk3(a.x + s, a.y + s, a.z + s); }
GSP_HD f3 operator-(f3 a, float s) { return mk3(a.x - s, a.y - s, a.z - s); }
GSP_HD f3 operator-(float s, f3 a) { return mk3(s - a.x, s - a.y, s - a.z); }
GSP_HD f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }

GSP_HD float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
GSP_HD f3 cross(f3 a, f3 b) {
  return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
GSP_HD float gsqrt(float x) { return __builtin_sqrtf(x); }  // correctly rounded on both targets
GSP_HD float length(f3 a) { return gsqrt(dot(a, a)); }
GSP_HD f3 normalize(f3 a) {
  float inv = 1.0f / gsqrt(dot(a, a));
  return mk3(a.x * inv, a.y * inv, a.z * inv);
}
GSP_HD f3 sqrt3(f3 a) { return mk3(gsqrt(a.x), gsqrt(a.y), gsqrt(a.z)); }
GSP_HD float gmin(float x, float y) { return y < x ? y : x; }  // GLSL min
GSP_HD float gmax(float x, float y) { return x < y ? y : x; }  // GLSL max
GSP_HD float gabs(float x) { return __builtin_fabsf(x); }
GSP_HD float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }
GSP_HD f3 faceforward(f3 N, f3 I, f3 Nref) { return dot(Nref, I) < 0.0f ? N : -N; }

GSP_HD uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
GSP_HD float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
GSP_HD bool gisinf(float x) { return (f2u(x) & 0x7fffffffu) == 0x7f800000u; }
GSP_HD bool gisnan(float x) { return (f2u(x) & 0x7fffffffu) > 0x7f800000u; }
GSP_HD bool gisvalid(float x) { return (f2u(x) & 0x7f800000u) != 0x7f800000u; }  // rayhit.rchit:658-660

// mat4 (glm memory order, m[4*c + r]) times (p,1) / (n,0), columns summed left to right
GSP_HD f3 xform_point(const float* m, f3 p) {
  return mk3(((m[0] * p.x + m[4] * p.y) + m[8] * p.z) + m[12], ((m[1] * p.x + m[5] * p.y) + m[9] * p.z) + m[13],
             ((m[2] * p.x + m[6] * p.y) + m[10] * p.z) + m[14]);
}
GSP_HD f3 xform_dir(const float* m, f3 n) {
  return mk3((m[0] * n.x + m[4] * n.y) + m[8] * n.z, (m[1] * n.x + m[5] * n.y) + m[9] * n.z,
             (m[2] * n.x + m[6] * n.y) + m[10] * n.z);
}

// ---------------------------------------------------------------------------
// sin/cos/log/exp.  GLSL leaves the built-ins' precision to the vendor; this
// tracer fixes them to Cody-Waite range reduction + Cephes single-precision
// minimax kernels (S. Moshier, public domain) evaluated with explicit fmaf, so
// the values are the same on any IEEE-754 machine.  ~1 ulp on the ranges used
// (|x| <= 2*pi for sin/cos, (0,1] for log, <= 0 for exp).
// ---------------------------------------------------------------------------
GSP_HD float gfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
GSP_HD float grint(float x) { return __builtin_rintf(x); }

GSP_HD void det_sincosf(float x, float& s_out, float& c_out) {
  const float kTwoOverPi = 0.63661977236758134308f;
  const float kPio2Hi = 1.57079637050628662109375f;
  const float kPio2Lo = -4.37113900018624283e-8f;
  float k = grint(x * kTwoOverPi);
  float r = gfma(-k, kPio2Hi, x);
  r = gfma(-k, kPio2Lo, r);
  float z = r * r;
  float ps = gfma(-1.9515295891e-4f, z, 8.3321608736e-3f);
  ps = gfma(ps, z, -1.6666654611e-1f);
  float s = gfma(ps * z, r, r);
  float pc = gfma(2.443315711809948e-5f, z, -1.388731625493765e-3f);
  pc = gfma(pc, z, 4.166664568298827e-2f);
  float c = gfma(pc * z, z, gfma(-0.5f, z, 1.0f));
  int q = ((int)k) & 3;
  float ss = (q & 1) ? c : s;
  float cc = (q & 1) ? s : c;
  if (q == 1 || q == 2) cc = -cc;
  if (q >= 2) ss = -ss;
  s_out = ss;
  c_out = cc;
}

GSP_HD float det_logf(float x) {
  if (x != x) return x;
  if (x < 0.0f) return u2f(0x7fc00000u);
  if (x == 0.0f) return u2f(0xff800000u);
  uint32_t bits = f2u(x);
  if (bits == 0x7f800000u) return x;
  int e = 0;
  if (bits < 0x00800000u) {
    x = x * 8388608.0f;
    bits = f2u(x);
    e = -23;
  }
  e += (int)((bits >> 23) & 0xffu) - 126;
  float m = u2f((bits & 0x007fffffu) | 0x3f000000u);
  float f;
  if (m < 0.70710678118654752440f) {
    e -= 1;
    f = (m + m) - 1.0f;
  } else {
    f = m - 1.0f;
  }
  float z = f * f;
  float p = gfma(7.0376836292e-2f, f, -1.1514610310e-1f);
  p = gfma(p, f, 1.1676998740e-1f);
  p = gfma(p, f, -1.2420140846e-1f);
  p = gfma(p, f, 1.4249322787e-1f);
  p = gfma(p, f, -1.6668057665e-1f);
  p = gfma(p, f, 2.0000714765e-1f);
  p = gfma(p, f, -2.4999993993e-1f);
  p = gfma(p, f, 3.3333331174e-1f);
  float fe = (float)e;
  float y = (p * f) * z;
  y = gfma(-2.12194440e-4f, fe, y);
  y = gfma(-0.5f, z, y);
  float r = f + y;
  r = gfma(0.693359375f, fe, r);
  return r;
}

GSP_HD float det_expf(float x) {
  if (x != x) return x;
  if (x > 88.72283905206835f) return u2f(0x7f800000u);
  if (x < -87.33654475055310898657f) return 0.0f;
  const float kLog2e = 1.44269504088896341f;
  const float kLn2Hi = 0.693359375f;
  const float kLn2Lo = -2.12194440e-4f;
  float n = grint(x * kLog2e);
  float r = gfma(-n, kLn2Hi, x);
  r = gfma(-n, kLn2Lo, r);
  float z = r * r;
  float p = gfma(1.9875691500e-4f, r, 1.3981999507e-3f);
  p = gfma(p, r, 8.3334519073e-3f);
  p = gfma(p, r, 4.1665795894e-2f);
  p = gfma(p, r, 1.6666665459e-1f);
  p = gfma(p, r, 5.0000001201e-1f);
  float y = gfma(p, z, r) + 1.0f;
  int ni = (int)n;
  int n1 = ni / 2;
  int n2 = ni - n1;
  float s1 = u2f((uint32_t)(n1 + 127) << 23);
  float s2 = u2f((uint32_t)(n2 + 127) << 23);
  return (y * s1) * s2;
}

// atan2 for the environment-map lookup (dormant-feature extension, include/gpuspectral_pt.h): Cephes atanf kernel
// (reduction at tan(pi/8) and tan(3pi/8), degree-4 polynomial in z = x^2) with explicit fmaf, quadrant by sign bits;
// atan2(0, 0) = 0.  ~2 ulp; what matters here is that it is the same function everywhere.
GSP_HD float det_atanf_pos(float x) {  // x >= 0
  float y0 = 0.0f;
  if (x > 2.414213562373095f) {
    y0 = 1.5707963267948966f;
    x = -1.0f / x;
  } else if (x > 0.4142135623730950f) {
    y0 = 0.7853981633974483f;
    x = (x - 1.0f) / (x + 1.0f);
  }
  const float z = x * x;
  float p = gfma(8.05374449538e-2f, z, -1.38776856032e-1f);
  p = gfma(p, z, 1.99777106478e-1f);
  p = gfma(p, z, -3.33329491539e-1f);
  return y0 + gfma(p * z, x, x);
}
GSP_HD float det_atan2f(float y, float x) {
  const float ax = gabs(x), ay = gabs(y);
  if (!(ax > 0.0f) && !(ay > 0.0f)) return 0.0f;  // (0, 0) and NaN
  float r;
  if (ax == 0.0f) r = 1.5707963267948966f;
  else r = det_atanf_pos(ay / ax);  // [0, pi/2]
  if (x < 0.0f) r = 3.14159265358979323846f - r;
  return y < 0.0f ? -r : r;
}

}  // namespace gsp
