// oracle/glsl_shim.h -- TEST INFRASTRUCTURE ONLY (see oracle/README.md, "The reference's own text, executed").
//
// The smallest C++ vocabulary under which the PURE functions of the reference's shaders
// (S/assets/shaders/pt_common.glsl:28-42,86-151, rayhit.rchit:17-69,89-654) compile as they are written.  Nothing here
// restates the reference: the shader text is read from /root/reference by tests/golden/make_glsl_vectors.py, which applies
// three token-level rewrites (`out T x` -> `T& x`; an unsuffixed real literal gets the `f` it has in GLSL, where it is a
// 32-bit float; `.xyz` -> `.xyz()`), compiles it against this header and records what the functions return.
//
// What this header DOES decide, because GLSL leaves it to the implementation, is the arithmetic of the built-ins; each is
// mapped to the definition oracle_math.h already fixes (and the product's pt_math.h states again):
//   dot / cross / length / normalize / faceforward, min / max / abs / clamp        -> orc::  (left-to-right sums, one divide)
//   sin, cos, log, exp                                                             -> orc::det_* (Cody-Waite + Cephes kernels)
//   sqrt                                                                           -> sqrtf (IEEE, exact)
//   atan(y, x), acos  (sphericalPhi / sphericalTheta, rayhit.rchit:168-175: defined, never called)  -> det_atan2f / libm
//   matrix x vector, mix, tan (the shaders' main() functions, r06 second step)     -> see mat4 below; x (1 - a) + y a; tanf
// GLSL evaluates function and constructor arguments left to right (GLSL 4.60 spec 6.1.1: "in order, from left to right"); C++
// leaves the order open for parenthesised calls, and it matters wherever two arguments draw random numbers
// (`vec2(randUniform(), randUniform())`, rayhit.rchit:90,156).  The constructor macros at the end turn every `vecN(...)`
// into the braced form `vecN{...}`, whose evaluation order C++ fixes left to right.  No call in the extracted text passes two
// side-effecting arguments to an ordinary function.
#pragma once
#include <cmath>
#include <cstdint>

#include "oracle_math.h"

namespace glsl {

typedef uint32_t uint;

struct uvec2 {
  uint x, y;
};
struct uvec3 {
  uint x, y, z;
  uvec2 xy() const { return uvec2{x, y}; }
};
struct ivec2 {
  int x, y;
  ivec2() = default;
  ivec2(int a, int b) : x(a), y(b) {}
  explicit ivec2(uvec2 v) : x((int)v.x), y((int)v.y) {}
};
struct ivec3 {
  int x, y, z;
  ivec3() = default;
  ivec3(int a, int b, int c) : x(a), y(b), z(c) {}
};
struct vec2 {
  float x, y;
  vec2() = default;
  explicit vec2(float s) : x(s), y(s) {}
  vec2(float a, float b) : x(a), y(b) {}
  explicit vec2(uvec2 v) : x((float)v.x), y((float)v.y) {}
};
struct vec3 {
  float x, y, z;
  vec3() = default;
  explicit vec3(float s) : x(s), y(s), z(s) {}
  vec3(float a, float b, float c) : x(a), y(b), z(c) {}
  vec3(orc::vec3 v) : x(v.x), y(v.y), z(v.z) {}
  explicit vec3(const struct vec4& v);
  operator orc::vec3() const { return orc::vec3{x, y, z}; }
  vec3 xyz() const { return *this; }
  vec3& operator*=(float s) { return *this = vec3(x * s, y * s, z * s); }
  vec3& operator/=(float s) { return *this = vec3(x / s, y / s, z / s); }
  vec3& operator*=(vec3 b) { return *this = vec3(x * b.x, y * b.y, z * b.z); }
  vec3& operator+=(vec3 b) { return *this = vec3(x + b.x, y + b.y, z + b.z); }
};
struct vec4 {
  float x, y, z, w;
  vec4() = default;
  vec4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
  vec4(vec3 v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
  vec3 xyz() const { return vec3(x, y, z); }
};
inline vec3::vec3(const vec4& v) : x(v.x), y(v.y), z(v.z) {}
// glm / GLSL column-major matrices: m[c][r] = a[4 * c + r].  matrix x vector = the columns scaled and summed left to right, the
// order oracle_math.h's xform_point / xform_dir state -- with ALL FOUR terms here: the oracle leaves the w term out where w is a
// literal 0 or folds it where w is 1 (x + m12 * 1 == x + m12; x + m12 * 0 == x except for the sign of an all-zero sum), and the
// image tests (tests/test_glsl_vectors.py::test_oracle_image_equals_the_executed_shaders) show that this reading changes no bit.
struct mat4 {
  float a[16];
};
struct mat4x3 {  // gl_ObjectToWorldEXT: 4 columns of 3 rows, kept in the glm mat4 layout it was made from
  float a[16];
};
static inline vec4 operator*(const mat4& m, vec4 v) {
  return vec4(((m.a[0] * v.x + m.a[4] * v.y) + m.a[8] * v.z) + m.a[12] * v.w, ((m.a[1] * v.x + m.a[5] * v.y) + m.a[9] * v.z) + m.a[13] * v.w,
              ((m.a[2] * v.x + m.a[6] * v.y) + m.a[10] * v.z) + m.a[14] * v.w, ((m.a[3] * v.x + m.a[7] * v.y) + m.a[11] * v.z) + m.a[15] * v.w);
}
static inline vec3 operator*(const mat4x3& m, vec4 v) {
  return vec3(((m.a[0] * v.x + m.a[4] * v.y) + m.a[8] * v.z) + m.a[12] * v.w, ((m.a[1] * v.x + m.a[5] * v.y) + m.a[9] * v.z) + m.a[13] * v.w,
              ((m.a[2] * v.x + m.a[6] * v.y) + m.a[10] * v.z) + m.a[14] * v.w);
}
struct bvec3 {
  bool x, y, z;
};
static_assert(sizeof(vec3) == 12 && alignof(vec3) == 4 && sizeof(vec4) == 16, "scalar block layout");

// component-wise operators, the scalar broadcast on either side (GLSL 4.60 spec 5.9)
static inline vec2 operator*(float s, vec2 a) { return vec2(s * a.x, s * a.y); }
static inline vec2 operator-(vec2 a, float s) { return vec2(a.x - s, a.y - s); }
static inline vec2 operator-(vec2 a, vec2 b) { return vec2(a.x - b.x, a.y - b.y); }
static inline vec2 operator/(vec2 a, float s) { return vec2(a.x / s, a.y / s); }
static inline bool operator==(vec3 a, vec3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }  // GLSL == on vectors: all components
static inline vec3 operator+(vec3 a, vec3 b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline vec3 operator-(vec3 a, vec3 b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline vec3 operator*(vec3 a, vec3 b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline vec3 operator/(vec3 a, vec3 b) { return vec3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline vec3 operator*(vec3 a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
static inline vec3 operator*(float s, vec3 a) { return vec3(s * a.x, s * a.y, s * a.z); }
static inline vec3 operator/(vec3 a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
static inline vec3 operator/(float s, vec3 a) { return vec3(s / a.x, s / a.y, s / a.z); }
static inline vec3 operator+(vec3 a, float s) { return vec3(a.x + s, a.y + s, a.z + s); }
static inline vec3 operator+(float s, vec3 a) { return vec3(s + a.x, s + a.y, s + a.z); }
static inline vec3 operator-(vec3 a, float s) { return vec3(a.x - s, a.y - s, a.z - s); }
static inline vec3 operator-(float s, vec3 a) { return vec3(s - a.x, s - a.y, s - a.z); }
static inline vec3 operator-(vec3 a) { return vec3(-a.x, -a.y, -a.z); }

// built-ins (see the header comment)
static inline float dot(vec3 a, vec3 b) { return orc::dot(a, b); }
static inline vec3 cross(vec3 a, vec3 b) { return orc::cross(a, b); }
static inline float length(vec3 a) { return orc::length(a); }
static inline vec3 normalize(vec3 a) { return orc::normalize(a); }
static inline vec3 faceforward(vec3 N, vec3 I, vec3 Nref) { return orc::faceforward(N, I, Nref); }
static inline float abs(float x) { return orc::gabs(x); }
static inline float min(float x, float y) { return orc::gmin(x, y); }
static inline float max(float x, float y) { return orc::gmax(x, y); }
static inline float clamp(float x, float lo, float hi) { return orc::gclamp(x, lo, hi); }
static inline float sqrt(float x) { return sqrtf(x); }
static inline vec3 sqrt(vec3 a) { return vec3(sqrtf(a.x), sqrtf(a.y), sqrtf(a.z)); }
static inline float sin(float x) { return orc::det_sinf(x); }
static inline float cos(float x) { return orc::det_cosf(x); }
static inline float log(float x) { return orc::det_logf(x); }
static inline float exp(float x) { return orc::det_expf(x); }
static inline float atan(float y, float x) { return orc::det_atan2f(y, x); }
static inline float acos(float x) { return ::acosf(x); }
static inline bool isinf(float x) { return orc::gisinf(x); }
static inline bool isnan(float x) { return orc::gisnan(x); }
static inline bvec3 isnan(vec3 a) { return bvec3{orc::gisnan(a.x), orc::gisnan(a.y), orc::gisnan(a.z)}; }
static inline bool any(bvec3 b) { return b.x || b.y || b.z; }
static inline float tan(float x) { return ::tanf(x); }  // (raygen.rgen:22; the oracle, too, takes the host's tanf: one value per frame)
static inline vec3 mix(vec3 x, vec3 y, float a) { return x * (1.0f - a) + y * a; }  // GLSL 4.60 8.3: x (1 - a) + y a

// `Name##Buffer(devicePointer).values[i]` (rayhit.rchit:71-82, GL_EXT_buffer_reference): a typed view of an address
template <class T>
struct BufferRef {
  const T* values;
  explicit BufferRef(const T* p) : values(p) {}
};

}  // namespace glsl

// pt_common.glsl:1.  A GLSL real literal without suffix is a 32-bit float (doubles need `lf`), so M_PI is float(pi).
#undef M_PI
#define M_PI 3.14159265358979323846f

// constructor calls -> braced initialisation: arguments evaluated left to right, as GLSL prescribes (header comment)
#define vec2(...) vec2{__VA_ARGS__}
#define vec3(...) vec3{__VA_ARGS__}
#define vec4(...) vec4{__VA_ARGS__}
