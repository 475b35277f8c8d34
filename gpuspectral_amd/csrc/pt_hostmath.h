// pt_hostmath.h -- host-side float32 linear algebra of gsp_upload_scene.
#pragma once

namespace gsp {

// ---- host-side linear algebra for PathTracer::prepareScene (PathTracer.cpp:62) ------
// glm::inverse(glm::transpose(M)); cofactor expansion in glm's order so the
// float32 result matches a glm build of the reference.
inline void transpose4(const float* m, float* o) {
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) o[4 * c + r] = m[4 * r + c];
}
inline void inverse4(const float* a, float* o) {
  auto M = [&](int c, int r) { return a[4 * c + r]; };
  const float c00 = M(2, 2) * M(3, 3) - M(3, 2) * M(2, 3), c02 = M(1, 2) * M(3, 3) - M(3, 2) * M(1, 3),
              c03 = M(1, 2) * M(2, 3) - M(2, 2) * M(1, 3), c04 = M(2, 1) * M(3, 3) - M(3, 1) * M(2, 3),
              c06 = M(1, 1) * M(3, 3) - M(3, 1) * M(1, 3), c07 = M(1, 1) * M(2, 3) - M(2, 1) * M(1, 3),
              c08 = M(2, 1) * M(3, 2) - M(3, 1) * M(2, 2), c10 = M(1, 1) * M(3, 2) - M(3, 1) * M(1, 2),
              c11 = M(1, 1) * M(2, 2) - M(2, 1) * M(1, 2), c12 = M(2, 0) * M(3, 3) - M(3, 0) * M(2, 3),
              c14 = M(1, 0) * M(3, 3) - M(3, 0) * M(1, 3), c15 = M(1, 0) * M(2, 3) - M(2, 0) * M(1, 3),
              c16 = M(2, 0) * M(3, 2) - M(3, 0) * M(2, 2), c18 = M(1, 0) * M(3, 2) - M(3, 0) * M(1, 2),
              c19 = M(1, 0) * M(2, 2) - M(2, 0) * M(1, 2), c20 = M(2, 0) * M(3, 1) - M(3, 0) * M(2, 1),
              c22 = M(1, 0) * M(3, 1) - M(3, 0) * M(1, 1), c23 = M(1, 0) * M(2, 1) - M(2, 0) * M(1, 1);
  const float f0[4] = {c00, c00, c02, c03}, f1[4] = {c04, c04, c06, c07}, f2[4] = {c08, c08, c10, c11},
              f3_[4] = {c12, c12, c14, c15}, f4[4] = {c16, c16, c18, c19}, f5[4] = {c20, c20, c22, c23};
  const float v0[4] = {M(1, 0), M(0, 0), M(0, 0), M(0, 0)}, v1[4] = {M(1, 1), M(0, 1), M(0, 1), M(0, 1)},
              v2[4] = {M(1, 2), M(0, 2), M(0, 2), M(0, 2)}, v3[4] = {M(1, 3), M(0, 3), M(0, 3), M(0, 3)};
  const float sa[4] = {1.0f, -1.0f, 1.0f, -1.0f}, sb[4] = {-1.0f, 1.0f, -1.0f, 1.0f};
  float inv[4][4];
  for (int i = 0; i < 4; ++i) {
    inv[0][i] = ((v1[i] * f0[i] - v2[i] * f1[i]) + v3[i] * f2[i]) * sa[i];
    inv[1][i] = ((v0[i] * f0[i] - v2[i] * f3_[i]) + v3[i] * f4[i]) * sb[i];
    inv[2][i] = ((v0[i] * f1[i] - v1[i] * f3_[i]) + v3[i] * f5[i]) * sa[i];
    inv[3][i] = ((v0[i] * f2[i] - v1[i] * f4[i]) + v2[i] * f5[i]) * sb[i];
  }
  const float d0 = M(0, 0) * inv[0][0], d1 = M(0, 1) * inv[1][0], d2 = M(0, 2) * inv[2][0], d3 = M(0, 3) * inv[3][0];
  const float ood = 1.0f / ((d0 + d1) + (d2 + d3));
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) o[4 * c + r] = inv[c][r] * ood;
}


}  // namespace gsp
