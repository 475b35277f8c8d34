// pt_trace.h -- BVH node / triangle packet layouts in HBM, the node encoder, the wide-node step and the triangle
// test every traversal shares (wave kernel pt_wavetrace.h, k_finish, host test harness tests/emu).
//
// Replaces the driver's acceleration-structure traversal behind traceRayEXT
// (S/assets/shaders/raygen.rgen:58, rayhit.rchit:738-748; build call sites
// S/backend/vulkan/VulkanRays.cpp:6-181).
//
// r03: the tree is a wide BVH with CONTIGUOUS CHILDREN traversed in a STATIC per-octant order, so that a node step
// has no sort network and pushes at most ONE stack entry (the rest of the node's hit children: a "node group"):
//   * the inner children of a node are consecutive nodes, its leaf children are consecutive triangle slots (one
//     triangle per leaf child), so a group of children is {base, which ones, in which order} in 32 bits instead of one
//     pointer per child;
//   * the visiting order of a node's children is fixed at build time for each of the 8 sign octants of the ray
//     direction (children sorted by the projection of their box centre, in units of the node's extent, on the
//     octant's diagonal).  Measured on the CPU (scripts/experiments/wide_bvh_probe.cpp, 200 k-triangle interior,
//     quantised boxes): 10.46 node visits per ray against 10.46 with an exact distance sort per node.
// The node width is 4.  (An 8-wide variant after Ylitie, Karras, Laine 2017 was built and measured in r03 -- slower on
// every scene, LAB_NOTES.md -- and lives on in scripts/experiments/ and in the history, not here.)
//
// HBM layout (all records are whole float4s so one lane moves 16 B per load):
//   4-wide node, "W4T" (64 B):
//                  q0 = {origin.x, origin.y, origin.z, scale.x}
//                       plane = origin + q * scale, q in 0..255, scale = a power of two per axis
//                  q1 = {qlo.x[child0..3], qlo.y[0..3], qlo.z[0..3], qhi.x[0..3]}   one byte per child
//                  q2 = {qhi.y[0..3], qhi.z[0..3], bits(child_base), bits(tri_base - ni)}
//                  q3 = {bits(order_lo), bits(order_hi), scale.y, scale.z}
//           children: the ni inner children first (positions 0..ni-1: nodes child_base + position), then the leaf
//           children (position p >= ni: triangle slot (tri_base - ni) + p); unused positions carry an inverted box
//           (qlo = 255, qhi = 0: never hit), so the traversal has no empty-slot test.
//           order_lo / order_hi: 7 bits per ray octant (0..3 / 4..7) = 2 * the code of {ni, visiting order of the
//           inner children}; the sequence table (SeqTable4, staged in LDS by the kernels) maps {miss mask, code} to
//           the hit inner children in visiting order + the hit leaf children.  order_lo bits 28..31 = mask of the
//           inner positions: any-hit rays, for which the order hardly matters (scripts/experiments/
//           wide_bvh_probe.cpp: 7.64 vs 7.74 node visits) but the latency of a step does, take the hit children in
//           position order straight from the miss mask instead of through the table.
//   child boxes are quantised outward (floor / ceil, verified against the decode), so they contain the exact
//   boxes: culling stays conservative and results do not change.
//   triangle packet (48 B, in the order the collapse emits leaf children):
//                  p0 = {v0.x, v0.y, v0.z, bits(global triangle id << 3 | BSDF type of the owning instance)}
//                  p1 = {v1.x, v1.y, v1.z, -}
//                  p2 = {v2.x, v2.y, v2.z, -}
//   Closest hit = smallest t, ties broken by the smaller global triangle id (the p0.w words order like the ids), so
//   the result does not depend on the BVH topology or traversal order.
#pragma once
#include "pt_math.h"

namespace gsp {

struct q4 {  // 16-byte quad, layout-compatible with float4
  float x, y, z, w;
};

struct HitRec {
  float t, u, v;
  int32_t slot;  // triangle slot in BVH leaf order; -1 = miss
};

constexpr int kWide = 4;
constexpr uint32_t kNodeQuads = 4u;  // 16-B quads per node
constexpr uint32_t kNodeBytes = 16u * kNodeQuads;
// The first kTopNodes records of the node array (the tree is emitted level by level: the root and the levels under it)
// are staged into LDS by every block of k_trace; the node buffer is allocated at least that long.
#ifndef GSP_TOP_NODES
#define GSP_TOP_NODES 64
#endif
constexpr uint32_t kTopNodes = GSP_TOP_NODES;
// (r04: staging the first 64 triangle SLOTS as well -- the leaf children of the top of the tree, i.e. the scene's largest
// triangles, which take 2.14 of the 3.93 triangle tests of a closest-hit ray on the bench scene -- was built, is bit-exact and
// changes nothing: 90.4-91.2 vs 89.9-90.9 ms; neither does 32 / 96 / 128 instead of 64 nodes.  The kernel is bound by VALU
// issue, not by the request path.  profiles/r04_lds_budget_probe.txt, r04_ab_lds_triangles.txt, r04_ab_top_nodes.txt;
// scripts/experiments/r04_lds_triangle_packets.patch)
constexpr size_t kNodeAllocMin = (size_t)(kTopNodes > 0 ? kTopNodes : 1) * kNodeBytes;
// A stack entry is (child_base << kGroupBits) | group: the group that is PUSHED has already lost the child being
// descended into (group_next runs first), so it names at most three children -- a marker bit over 3 x 2 bits for a
// closest-hit ray, a 4-bit mask for an any-hit ray -- and 7 bits hold it; child_base gets the other 25.  The wide tree of
// n triangles has fewer than n nodes, so every scene gsp_upload_scene accepts (n < kMaxNodes) fits: no late rejection.
constexpr uint32_t kGroupBits = 7;
constexpr uint32_t kGroupMask = (1u << kGroupBits) - 1u;
constexpr uint32_t kMaxNodes = 1u << (32 - kGroupBits);
GSP_HD uint32_t pack_group(uint32_t gb, uint32_t gs) { return (gb << kGroupBits) | gs; }

// Watertight ray/triangle test (Woop, Benthin, Wald: "Watertight Ray/Triangle
// Intersection", JCGT 2013), no back-face culling: the ray is sheared so that it
// runs along +z of a permuted frame; the three 2D edge functions of a shared edge
// are exact negations of each other in the two triangles that share it, so a ray
// can never slip between them (hardware traversal behind traceRayEXT is watertight
// too).  Zero edge values are re-evaluated in double.  The operation order is part
// of the parity contract with the oracle (oracle/oracle_pt.cpp intersectTri).
struct RayShear {
  int kx, ky, kz;
  float Sx, Sy, Sz;
};
GSP_HD float comp(f3 v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); }
GSP_HD RayShear make_shear(f3 d) {
  RayShear r;
  const float ax = gabs(d.x), ay = gabs(d.y), az = gabs(d.z);
  r.kz = (ax > ay) ? ((ax > az) ? 0 : 2) : ((ay > az) ? 1 : 2);
  r.kx = r.kz == 2 ? 0 : r.kz + 1;
  r.ky = r.kx == 2 ? 0 : r.kx + 1;
  const float dz = comp(d, r.kz);
  if (dz < 0.0f) {
    const int t = r.kx;
    r.kx = r.ky;
    r.ky = t;
  }
  r.Sx = comp(d, r.kx) / dz;
  r.Sy = comp(d, r.ky) / dz;
  r.Sz = 1.0f / dz;
  return r;
}
// u, v = barycentric weights of v1 and v2 (hitAttributeEXT attribs.xy, rayhit.rchit:690)
GSP_HD bool intersect_tri(f3 v0, f3 v1, f3 v2, f3 o, const RayShear& rs, float tmin, float tmax, float& t, float& u,
                          float& v) {
  const f3 A = v0 - o, B = v1 - o, C = v2 - o;
  const float Akz = comp(A, rs.kz), Bkz = comp(B, rs.kz), Ckz = comp(C, rs.kz);
  const float Ax = comp(A, rs.kx) - rs.Sx * Akz, Ay = comp(A, rs.ky) - rs.Sy * Akz;
  const float Bx = comp(B, rs.kx) - rs.Sx * Bkz, By = comp(B, rs.ky) - rs.Sy * Bkz;
  const float Cx = comp(C, rs.kx) - rs.Sx * Ckz, Cy = comp(C, rs.ky) - rs.Sy * Ckz;
  float U = Cx * By - Cy * Bx;
  float V = Ax * Cy - Ay * Cx;
  float W = Bx * Ay - By * Ax;
  if (U == 0.0f || V == 0.0f || W == 0.0f) {
    U = (float)((double)Cx * (double)By - (double)Cy * (double)Bx);
    V = (float)((double)Ax * (double)Cy - (double)Ay * (double)Cx);
    W = (float)((double)Bx * (double)Ay - (double)By * (double)Ax);
  }
  if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return false;
  const float det = (U + V) + W;
  if (det == 0.0f) return false;
  const float Az = rs.Sz * Akz, Bz = rs.Sz * Bkz, Cz = rs.Sz * Ckz;
  const float T = (U * Az + V * Bz) + W * Cz;
  const float rcp = 1.0f / det;
  t = T * rcp;
  u = V * rcp;
  v = W * rcp;
  return t > tmin && t < tmax;
}

// The same test for the wave kernel's leaf step, restated for its instruction stream (same operations in the same
// order on every accepted triangle, so the same t, u, v bit for bit):
//  * no early exits: a wave runs the division whenever one of its lanes gets that far, so the branches buy nothing
//    there, and every value a divergent block defines costs register copies at the joins (a rejected triangle may
//    compute inf / NaN, which no comparison accepts);
//  * the ray's axis permutation -- make_shear's (kx, ky, kz), the exchange of kx and ky for dz < 0 included, so that even the
//    sign of a zero u or v is the oracle's -- as three lane masks and full-rate bit selects (v_bitop3_b32) instead of 24
//    half-rate compares / conditional moves per triangle.
struct RayShearRot {
  uint32_t m0, m1, ms;  // all ones where kz == 0 / kz == 1 / d[kz] < 0
  float Sx, Sy, Sz;
};
GSP_HD float bit_select(uint32_t m, float a, float b) {  // m ? a : b, bitwise
#if defined(__HIP_DEVICE_COMPILE__)
  // (m & a) | (~m & b) as ONE v_bitop3_b32 (truth table 0xCA), which issues at full rate; the compiler's own choice for this
  // pattern is v_bfi_b32 -- half rate on gfx950 (4.24 vs 2.69 cycles, profiles/r05_valu_rate.txt), 27 of them per leaf step
  // (r05: closest-hit kernel -0.5 %, profiles/r05_ab_bitop3.txt)
  return __uint_as_float(__builtin_amdgcn_bitop3_b32(m, __float_as_uint(a), __float_as_uint(b), 0xCA));
#else
  uint32_t ua, ub;
  __builtin_memcpy(&ua, &a, 4);
  __builtin_memcpy(&ub, &b, 4);
  const uint32_t r = (ua & m) | (ub & ~m);
  float f;
  __builtin_memcpy(&f, &r, 4);
  return f;
#endif
}
// (x, y, z) -> (component kx, ky, kz): kz by the masks, kx = kz + 1, ky = kz + 2 (mod 3), exchanged where ms is set
GSP_HD f3 permute_axes(const RayShearRot& rs, f3 p) {
  const float a = bit_select(rs.m0, p.y, bit_select(rs.m1, p.z, p.x)), b = bit_select(rs.m0, p.z, bit_select(rs.m1, p.x, p.y));
  return mk3(bit_select(rs.ms, b, a), bit_select(rs.ms, a, b), bit_select(rs.m0, p.x, bit_select(rs.m1, p.y, p.z)));
}
// inv_dz = 1 / d[kz], correctly rounded (the traversal's RayBox holds it)
GSP_HD RayShearRot make_shear_rot(f3 d) {
  RayShearRot r;
  const float ax = gabs(d.x), ay = gabs(d.y), az = gabs(d.z);
  const int kz = (ax > ay) ? ((ax > az) ? 0 : 2) : ((ay > az) ? 1 : 2);  // as make_shear
  r.m0 = kz == 0 ? 0xffffffffu : 0u;
  r.m1 = kz == 1 ? 0xffffffffu : 0u;
  r.ms = comp(d, kz) < 0.0f ? 0xffffffffu : 0u;
  const f3 q = permute_axes(r, d);
  r.Sx = q.x / q.z;
  r.Sy = q.y / q.z;
  r.Sz = 1.0f / q.z;
  return r;
}
GSP_HD bool intersect_tri_rot(f3 v0, f3 v1, f3 v2, f3 o, const RayShearRot& rs, float tmin, float tmax, float& t, float& u,
                              float& v) {
  const f3 A = permute_axes(rs, v0 - o), B = permute_axes(rs, v1 - o), C = permute_axes(rs, v2 - o);
  const float Ax = A.x - rs.Sx * A.z, Ay = A.y - rs.Sy * A.z;
  const float Bx = B.x - rs.Sx * B.z, By = B.y - rs.Sy * B.z;
  const float Cx = C.x - rs.Sx * C.z, Cy = C.y - rs.Sy * C.z;
  float U = Cx * By - Cy * Bx;
  float V = Ax * Cy - Ay * Cx;
  float W = Bx * Ay - By * Ax;
  if (U == 0.0f || V == 0.0f || W == 0.0f) {
    U = (float)((double)Cx * (double)By - (double)Cy * (double)Bx);
    V = (float)((double)Ax * (double)Cy - (double)Ay * (double)Cx);
    W = (float)((double)Bx * (double)Ay - (double)By * (double)Ax);
  }
  const bool neg = (U < 0.0f) | (V < 0.0f) | (W < 0.0f), pos = (U > 0.0f) | (V > 0.0f) | (W > 0.0f);
  const float det = (U + V) + W;
  const float Az = rs.Sz * A.z, Bz = rs.Sz * B.z, Cz = rs.Sz * C.z;
  const float T = (U * Az + V * Bz) + W * Cz;
  const float rcp = 1.0f / det;
  t = T * rcp;
  u = V * rcp;
  v = W * rcp;
  return !(neg & pos) & (det != 0.0f) & (t > tmin) & (t < tmax);
}

// min/max here are the NaN-dropping IEEE forms, so a NaN from 0*inf (origin on a slab plane, zero direction
// component) leaves that slab unconstrained: conservative.
GSP_HD float fmin_(float a, float b) { return __builtin_fminf(a, b); }
GSP_HD float fmax_(float a, float b) { return __builtin_fmaxf(a, b); }

GSP_HD q4 make_q4(float x, float y, float z, float w) {
  q4 r;
  r.x = x;
  r.y = y;
  r.z = z;
  r.w = w;
  return r;
}

// ---- static tables of the traversal -----------------------------------------------------------------------------
// 4-wide: the order code c of a node for one ray octant names {ni, visiting order of the inner children 0..ni-1}:
//   c = 0: ni 0 | 1: ni 1 | 2..3: ni 2 | 4..9: ni 3 | 10..33: ni 4, permutations in lexicographic order.
// SeqTable4[miss][c] (16 rows of 64 entries, 2 KB) = seq | leaf_hits << 9, where seq lists the positions of the hit
// inner children in visiting order, two bits each, first one lowest, under a marker bit (seq == 1: none), and
// leaf_hits bit p = the leaf child at position p (>= ni) is hit.
constexpr uint32_t kSeqEmpty = 1u;
constexpr int kOrderCodes = 34;
struct OrderCode {
  int ni;
  int perm[4];
};
constexpr OrderCode decode_order(int c) {
  OrderCode r{0, {0, 0, 0, 0}};
  int idx = 0;
  if (c == 0) return r;
  if (c == 1) r.ni = 1, idx = 0;
  else if (c < 4) r.ni = 2, idx = c - 2;
  else if (c < 10) r.ni = 3, idx = c - 4;
  else r.ni = 4, idx = c - 10;
  int avail[4] = {0, 1, 2, 3};
  int left = r.ni;
  for (int i = 0; i < r.ni; ++i) {
    int f = 1;
    for (int k = 2; k < left; ++k) f *= k;  // (left - 1)!
    const int k = idx / f;
    idx %= f;
    r.perm[i] = avail[k];
    for (int j = k; j + 1 < left; ++j) avail[j] = avail[j + 1];
    --left;
  }
  return r;
}
// inverse of decode_order: the code of {ni, perm[0..ni)}
constexpr int encode_order(int ni, const int* perm) {
  if (ni == 0) return 0;
  const int base = ni == 1 ? 1 : (ni == 2 ? 2 : (ni == 3 ? 4 : 10));
  int avail[4] = {0, 1, 2, 3};
  int left = ni, idx = 0;
  for (int i = 0; i < ni; ++i) {
    int f = 1;
    for (int k = 2; k < left; ++k) f *= k;
    int k = 0;
    while (avail[k] != perm[i]) ++k;
    idx += k * f;
    for (int j = k; j + 1 < left; ++j) avail[j] = avail[j + 1];
    --left;
  }
  return base + idx;
}
struct SeqTable4 {
  uint16_t v[16 * 64];
  constexpr SeqTable4() : v() {
    for (int m = 0; m < 16; ++m)
      for (int c = 0; c < 64; ++c) {
        uint32_t e = kSeqEmpty;
        if (c < kOrderCodes) {
          const OrderCode oc = decode_order(c);
          const uint32_t hit = ~(uint32_t)m & 15u;
          uint32_t seq = kSeqEmpty;
          for (int i = oc.ni - 1; i >= 0; --i)
            if ((hit >> oc.perm[i]) & 1u) seq = (seq << 2) | (uint32_t)oc.perm[i];
          e = seq | ((hit & ~((1u << oc.ni) - 1u)) << 9);
        }
        v[m * 64 + c] = (uint16_t)e;
      }
  }
};
constexpr uint32_t kStepTableBytes = 2048;
constexpr SeqTable4 kStepTable{};

// ---- node encoder (device BVH build, pt_bvh.hip; host test harness, tests/emu) --------------------------------------
struct WideChild {
  q4 lo, hi;  // box (xyz)
};
// common part: origin, per-axis power-of-two scale, and the quantised planes q[plane][child] (plane: lo.x lo.y lo.z hi.x hi.y hi.z)
GSP_HD void quantise_children(const WideChild* e, int cnt, float* lo, float* scale, uint8_t q[6][8]) {
  float hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  lo[0] = lo[1] = lo[2] = 3.0e38f;
  for (int k = 0; k < cnt; ++k) {
    lo[0] = fmin_(lo[0], e[k].lo.x); lo[1] = fmin_(lo[1], e[k].lo.y); lo[2] = fmin_(lo[2], e[k].lo.z);
    hi[0] = fmax_(hi[0], e[k].hi.x); hi[1] = fmax_(hi[1], e[k].hi.y); hi[2] = fmax_(hi[2], e[k].hi.z);
  }
  if (cnt == 0) lo[0] = lo[1] = lo[2] = hi[0] = hi[1] = hi[2] = 0.0f;
  for (int a = 0; a < 3; ++a) {
    const float ext = hi[a] - lo[a];
    int ex = -100;
    if (ext > 0.0f) (void)frexpf(ext / 255.0f, &ex);  // 2^ex >= ext / 255
    int eb = ex + 127;
    eb = eb < 1 ? 1 : (eb > 254 ? 254 : eb);
    // the largest code must reach the far side of the node
    while (eb < 254 && __builtin_fmaf(255.0f, u2f((uint32_t)eb << 23), lo[a]) < hi[a]) ++eb;
    scale[a] = u2f((uint32_t)eb << 23);
  }
  for (int k = 0; k < 8; ++k)  // unused: inverted box (near plane beyond the far plane: misses)
    for (int a = 0; a < 3; ++a) q[a][k] = 255, q[3 + a][k] = 0;
  for (int k = 0; k < cnt; ++k) {
    const float cl[3] = {e[k].lo.x, e[k].lo.y, e[k].lo.z}, ch[3] = {e[k].hi.x, e[k].hi.y, e[k].hi.z};
    for (int a = 0; a < 3; ++a) {
      float ql = fmin_(fmax_(__builtin_floorf((cl[a] - lo[a]) / scale[a]), 0.0f), 255.0f);
      while (ql > 0.0f && __builtin_fmaf(ql, scale[a], lo[a]) > cl[a]) ql -= 1.0f;  // decoded plane must not exceed the box
      float qh = fmin_(fmax_(__builtin_ceilf((ch[a] - lo[a]) / scale[a]), 0.0f), 255.0f);
      while (qh < 255.0f && __builtin_fmaf(qh, scale[a], lo[a]) < ch[a]) qh += 1.0f;
      q[a][k] = (uint8_t)ql;
      q[3 + a][k] = (uint8_t)qh;
    }
  }
}
// the key that orders the children of a node for the rays of one sign octant (bit a of `oct` set = direction negative
// on axis a): centre of the child box in units of the node's extent, projected on the octant's diagonal
GSP_HD float order_key(const WideChild& c, const float* ext, int oct) {
  const float cx = ext[0] > 0.0f ? (c.lo.x + c.hi.x) / ext[0] : 0.0f;
  const float cy = ext[1] > 0.0f ? (c.lo.y + c.hi.y) / ext[1] : 0.0f;
  const float cz = ext[2] > 0.0f ? (c.lo.z + c.hi.z) / ext[2] : 0.0f;
  return ((oct & 1 ? -cx : cx) + (oct & 2 ? -cy : cy)) + (oct & 4 ? -cz : cz);
}
GSP_HD uint32_t pack4(const uint8_t* b) { return (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24); }

// 4-wide node: e[0..ni) inner children (nodes child_base + k), e[ni..ni+nl) leaf children (triangle slots tri_base + j)
// The encoder as first written -- loops over the children, arrays indexed by loop variables -- kept as the DEFINITION the one
// below is checked against (tests/test_emu_parity.py::test_node_encoder_forms_agree): on the device its arrays live in scratch
// memory and a node takes ~5 k dependent instructions, which is what a refit of the tree waited for, level by level
// (21 launches of 9-90 us each for a million triangles: profiles/r05_refit_encoder.txt).
GSP_HD void encode_node_w4_ref(q4* __restrict__ o, const WideChild* e, int ni, int nl, uint32_t child_base, uint32_t tri_base) {
  float lo[3], scale[3];
  uint8_t q[6][8];
  quantise_children(e, ni + nl, lo, scale, q);
  float ext[3] = {0.0f, 0.0f, 0.0f};
  for (int k = 0; k < ni + nl; ++k) {
    ext[0] = fmax_(ext[0], e[k].hi.x - lo[0]);
    ext[1] = fmax_(ext[1], e[k].hi.y - lo[1]);
    ext[2] = fmax_(ext[2], e[k].hi.z - lo[2]);
  }
  uint32_t order[2] = {0u, 0u};
  for (int oct = 0; oct < 8; ++oct) {
    float key[4];
    int perm[4] = {0, 1, 2, 3};
    for (int k = 0; k < ni; ++k) key[k] = order_key(e[k], ext, oct);
    for (int i = 1; i < ni; ++i)  // insertion sort, ties keep the position order
      for (int j = i; j > 0 && key[perm[j]] < key[perm[j - 1]]; --j) {
        const int t = perm[j];
        perm[j] = perm[j - 1];
        perm[j - 1] = t;
      }
    const uint32_t c2 = 2u * (uint32_t)encode_order(ni, perm);
    order[oct >> 2] |= c2 << (7 * (oct & 3));
  }
  order[0] |= ((1u << ni) - 1u) << 28;
  o[0] = make_q4(lo[0], lo[1], lo[2], scale[0]);
  o[1] = make_q4(u2f(pack4(q[0])), u2f(pack4(q[1])), u2f(pack4(q[2])), u2f(pack4(q[3])));
  o[2] = make_q4(u2f(pack4(q[4])), u2f(pack4(q[5])), u2f(child_base), u2f(tri_base - (uint32_t)ni));
  o[3] = make_q4(u2f(order[0]), u2f(order[1]), scale[1], scale[2]);
}

// The same record, written for the device: the four child positions in named variables, every loop unrolled over them with the
// position's validity as a predicate, the stable insertion sort of the per-octant keys as its six compare-exchanges (each one
// conditional on the one before, as the loop's early exit is), the order code from the Lehmer digits in closed form, power-of-two
// divisions as exact multiplications.  Same operations in the same order on the same values: the same bits.
GSP_HD void encode_node_w4(q4* __restrict__ o, const WideChild* e, int ni, int nl, uint32_t child_base, uint32_t tri_base) {
  const int cnt = ni + nl;
  const bool on[4] = {cnt > 0, cnt > 1, cnt > 2, cnt > 3};
  WideChild c[4];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int k = 0; k < 4; ++k) {
    c[k].lo = make_q4(0.0f, 0.0f, 0.0f, 0.0f);
    c[k].hi = make_q4(0.0f, 0.0f, 0.0f, 0.0f);
    if (on[k]) c[k] = e[k];
  }
  // ---- quantise_children: origin, power-of-two scales ----
  float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int k = 0; k < 4; ++k)
    if (on[k]) {
      lo[0] = fmin_(lo[0], c[k].lo.x); lo[1] = fmin_(lo[1], c[k].lo.y); lo[2] = fmin_(lo[2], c[k].lo.z);
      hi[0] = fmax_(hi[0], c[k].hi.x); hi[1] = fmax_(hi[1], c[k].hi.y); hi[2] = fmax_(hi[2], c[k].hi.z);
    }
  if (cnt == 0) lo[0] = lo[1] = lo[2] = hi[0] = hi[1] = hi[2] = 0.0f;
  float scale[3], inv_scale[3];
  bool exact_inv[3];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int a = 0; a < 3; ++a) {
    const float ext = hi[a] - lo[a];
    int ex = -100;
    if (ext > 0.0f) (void)frexpf(ext / 255.0f, &ex);  // 2^ex >= ext / 255
    int eb = ex + 127;
    eb = eb < 1 ? 1 : (eb > 254 ? 254 : eb);
    while (eb < 254 && __builtin_fmaf(255.0f, u2f((uint32_t)eb << 23), lo[a]) < hi[a]) ++eb;  // the largest code must reach the far side
    scale[a] = u2f((uint32_t)eb << 23);
    // x / 2^k == x * 2^-k, correctly rounded both, whenever 2^-k is a normal number
    exact_inv[a] = eb <= 253;
    inv_scale[a] = u2f((uint32_t)(254 - eb) << 23);
  }
  // ---- quantised planes, packed: word[plane] byte k = child k; unused positions carry the inverted box (255 / 0) ----
  uint32_t qw[6] = {0u, 0u, 0u, 0u, 0u, 0u};
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int k = 0; k < 4; ++k) {
    const float cl[3] = {c[k].lo.x, c[k].lo.y, c[k].lo.z}, ch[3] = {c[k].hi.x, c[k].hi.y, c[k].hi.z};
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int a = 0; a < 3; ++a) {
      uint32_t bl = 255u, bh = 0u;
      if (on[k]) {
        const float dl = cl[a] - lo[a], dh = ch[a] - lo[a];
        float ql = fmin_(fmax_(__builtin_floorf(exact_inv[a] ? dl * inv_scale[a] : dl / scale[a]), 0.0f), 255.0f);
        while (ql > 0.0f && __builtin_fmaf(ql, scale[a], lo[a]) > cl[a]) ql -= 1.0f;  // decoded plane must not exceed the box
        float qh = fmin_(fmax_(__builtin_ceilf(exact_inv[a] ? dh * inv_scale[a] : dh / scale[a]), 0.0f), 255.0f);
        while (qh < 255.0f && __builtin_fmaf(qh, scale[a], lo[a]) < ch[a]) qh += 1.0f;
        bl = (uint32_t)(uint8_t)ql;
        bh = (uint32_t)(uint8_t)qh;
      }
      qw[a] |= bl << (8 * k);
      qw[3 + a] |= bh << (8 * k);
    }
  }
  // ---- per-octant visiting orders of the inner children ----
  float ext[3] = {0.0f, 0.0f, 0.0f};
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int k = 0; k < 4; ++k)
    if (on[k]) {
      ext[0] = fmax_(ext[0], c[k].hi.x - lo[0]);
      ext[1] = fmax_(ext[1], c[k].hi.y - lo[1]);
      ext[2] = fmax_(ext[2], c[k].hi.z - lo[2]);
    }
  float cx[4], cy[4], cz[4];  // order_key's centre of child k in units of the node's extent (the same for all eight octants)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int k = 0; k < 4; ++k) {
    cx[k] = ext[0] > 0.0f ? (c[k].lo.x + c[k].hi.x) / ext[0] : 0.0f;
    cy[k] = ext[1] > 0.0f ? (c[k].lo.y + c[k].hi.y) / ext[1] : 0.0f;
    cz[k] = ext[2] > 0.0f ? (c[k].lo.z + c[k].hi.z) / ext[2] : 0.0f;
  }
  uint32_t order[2] = {0u, 0u};
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int oct = 0; oct < 8; ++oct) {
    // slot j holds perm[j] and its key
    float k0 = ((oct & 1 ? -cx[0] : cx[0]) + (oct & 2 ? -cy[0] : cy[0])) + (oct & 4 ? -cz[0] : cz[0]);
    float k1 = ((oct & 1 ? -cx[1] : cx[1]) + (oct & 2 ? -cy[1] : cy[1])) + (oct & 4 ? -cz[1] : cz[1]);
    float k2 = ((oct & 1 ? -cx[2] : cx[2]) + (oct & 2 ? -cy[2] : cy[2])) + (oct & 4 ? -cz[2] : cz[2]);
    float k3 = ((oct & 1 ? -cx[3] : cx[3]) + (oct & 2 ? -cy[3] : cy[3])) + (oct & 4 ? -cz[3] : cz[3]);
    int p0 = 0, p1 = 1, p2 = 2, p3 = 3;
#define GSP_CSWAP(KA, KB, PA, PB, COND)        \
  {                                            \
    const bool s_ = (COND);                    \
    const float ta_ = KA, tb_ = KB;            \
    const int qa_ = PA, qb_ = PB;              \
    KA = s_ ? tb_ : ta_;                       \
    KB = s_ ? ta_ : tb_;                       \
    PA = s_ ? qb_ : qa_;                       \
    PB = s_ ? qa_ : qb_;                       \
  }
    bool m;  // "the element being inserted is still moving left"
    m = ni > 1 && k1 < k0;
    GSP_CSWAP(k0, k1, p0, p1, m)
    m = ni > 2 && k2 < k1;
    GSP_CSWAP(k1, k2, p1, p2, m)
    m = m && k1 < k0;
    GSP_CSWAP(k0, k1, p0, p1, m)
    m = ni > 3 && k3 < k2;
    GSP_CSWAP(k2, k3, p2, p3, m)
    m = m && k2 < k1;
    GSP_CSWAP(k1, k2, p1, p2, m)
    m = m && k1 < k0;
    GSP_CSWAP(k0, k1, p0, p1, m)
#undef GSP_CSWAP
    // encode_order: Lehmer digits d_i = perm[i] - #{j < i : perm[j] < perm[i]}, weights (ni - 1 - i)!
    const int d0 = p0;
    const int d1 = p1 - (p0 < p1 ? 1 : 0);
    const int d2 = p2 - (p0 < p2 ? 1 : 0) - (p1 < p2 ? 1 : 0);
    int code = 0;
    if (ni == 1) code = 1;
    else if (ni == 2) code = 2 + d0;
    else if (ni == 3) code = 4 + 2 * d0 + d1;
    else if (ni == 4) code = 10 + 6 * d0 + 2 * d1 + d2;
    order[oct >> 2] |= (2u * (uint32_t)code) << (7 * (oct & 3));
  }
  order[0] |= ((1u << ni) - 1u) << 28;
  o[0] = make_q4(lo[0], lo[1], lo[2], scale[0]);
  o[1] = make_q4(u2f(qw[0]), u2f(qw[1]), u2f(qw[2]), u2f(qw[3]));
  o[2] = make_q4(u2f(qw[4]), u2f(qw[5]), u2f(child_base), u2f(tri_base - (uint32_t)ni));
  o[3] = make_q4(u2f(order[0]), u2f(order[1]), scale[1], scale[2]);
}

// ---- the node step ------------------------------------------------------------------------------------------------
// Per-ray constants of the step.
struct RayBox {
  f3 o, inv;               // origin, 1 / direction
  f3 invc;                 // inv clamped to +-2^64 (box tests only; the triangle test uses the exact ray)
  bool negx, negy, negz;   // sign of 1/d per axis: which plane of a slab is the near one
  uint32_t mx, my, mz;     // ... as lane masks (all ones / zero): the node step picks near / far planes with full-rate bit selects (r05)
  bool octhi;              // sign octant >= 4: the node's order_hi word
  uint32_t octshift;       // 7 * (octant & 3): position of the octant's order code in that word
};
GSP_HD RayBox make_raybox(f3 o, f3 d) {
  RayBox r;
  r.o = o;
  r.inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  r.negx = r.inv.x < 0.0f;
  r.negy = r.inv.y < 0.0f;
  r.negz = r.inv.z < 0.0f;
  r.mx = r.negx ? 0xffffffffu : 0u;
  r.my = r.negy ? 0xffffffffu : 0u;
  r.mz = r.negz ? 0xffffffffu : 0u;
  const float big = 18446744073709551616.0f;  // 2^64
  r.invc = mk3(fmin_(fmax_(r.inv.x, -big), big), fmin_(fmax_(r.inv.y, -big), big), fmin_(fmax_(r.inv.z, -big), big));
  r.octhi = r.negz;
  r.octshift = (r.negx ? 7u : 0u) + (r.negy ? 14u : 0u);
  return r;
}

// m = (m << 1) | (x < 0): one v_alignbit_b32 on the device (the sign bit of x is shifted into the mask)
GSP_HD uint32_t shift_in_sign(uint32_t m, float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbit(m, __float_as_uint(x), 31);
#else
  return (m << 1) | (f2u(x) >> 31);
#endif
}

// (float)((q >> 8k) & 0xff) compiles to v_cvt_f32_ubyteK
#define GSP_UB0(q) ((float)((q) & 0xffu))
#define GSP_UB1(q) ((float)(((q) >> 8) & 0xffu))
#define GSP_UB2(q) ((float)(((q) >> 16) & 0xffu))
#define GSP_UB3(q) ((float)((q) >> 24))
// One child box against the ray segment [tmin, tfar]; a miss shifts a 1 into M.
//   decode: plane = origin + q * scale (per-axis power of two); its ray parameter is
//   t = fma(q, scale / d, (origin - o) / d): one fma per plane after six multiplies per node (scale / d is exact up to
//   the rounding of 1/d: scale is a power of two).  1/d is CLAMPED to +-2^64 for this: with an infinite 1/d (a zero
//   direction component) every plane of that axis would be fma(q, inf, +-inf) = NaN, the slab would stop
//   constraining and axis-parallel rays would lose their culling (measured in round 1 without the clamp: +13 % / +47 %
//   kernel time).  With the clamp such a ray sees t = (plane - o) * 2^64: both planes of a slab it is outside of land
//   far beyond tmax <= 1e10 on the same side (culled, correctly: the ray never enters), a slab it is inside of gives
//   -huge / +huge (unconstrained), and no product can overflow (|scale|, |origin - o| < 2^40).  The rounding error of
//   t is of the same order as with an uncompressed (b - o) * (1/d) test -- 2^-24 |origin - o| / |d| -- and is covered
//   by the outward quantisation, the padded leaf boxes and the 8-ulp slack on the far bound.
//   Near / far planes are picked by the sign of 1/d instead of min / max per child: for inv > 0
//   (lo - o) * inv <= (hi - o) * inv by monotonic rounding, so the values are the ones min / max would return; a
//   NaN (0 * inf) is dropped by max / min and leaves that side unconstrained.
//   hit  <=>  lo <= hi * (1 + 8 ulp)  <=>  fma(hi, 1.000001, -lo) >= 0 (an fma and the sign bit instead of a multiply,
//   a compare and a select; lo and hi are never NaN: tmin and tfar are not).
#define GSP_CHILD(M, CVT, QNX, QFX, QNY, QFY, QNZ, QFZ)                                            \
  {                                                                                                 \
    const float tnx = __builtin_fmaf(CVT(QNX), sx, dx), tfx = __builtin_fmaf(CVT(QFX), sx, dx);     \
    const float tny = __builtin_fmaf(CVT(QNY), sy, dy), tfy = __builtin_fmaf(CVT(QFY), sy, dy);     \
    const float tnz = __builtin_fmaf(CVT(QNZ), sz, dz), tfz = __builtin_fmaf(CVT(QFZ), sz, dz);     \
    const float lo = fmax_(fmax_(tnx, tny), fmax_(tnz, tmin));                                      \
    const float hi = fmin_(fmin_(tfx, tfy), fmin_(tfz, tfar));                                      \
    M = shift_in_sign(M, __builtin_fmaf(hi, 1.000001f, -lo));                                       \
  }

// bit k of the result = child k is MISSED
GSP_HD uint32_t node_test(const q4* n, const RayBox& rb, float tmin, float tfar) {
  // per-node constants of the decode: scale / d and (origin - o) / d per axis (6 multiplies), then ONE fma per plane
  const float sx = n[0].w * rb.invc.x, sy = n[3].z * rb.invc.y, sz = n[3].w * rb.invc.z;
  const float dx = (n[0].x - rb.o.x) * rb.invc.x, dy = (n[0].y - rb.o.y) * rb.invc.y, dz = (n[0].z - rb.o.z) * rb.invc.z;
  const uint32_t qlx = f2u(n[1].x), qly = f2u(n[1].y), qlz = f2u(n[1].z), qhx = f2u(n[1].w), qhy = f2u(n[2].x), qhz = f2u(n[2].y);
  // near / far plane words by the sign of 1/d: twelve selects per node.  As `neg ? hi : lo` they are v_cndmask_b32 (half rate on
  // gfx950, 4.24 cycles); as (m & hi) | (~m & lo) on lane masks one v_bitop3_b32 each (full rate, 2.69): closest-hit kernel
  // -1.1 %, any-hit -1.9 % for three more VGPRs (71 of 72; profiles/r05_ab_bitop3.txt)
#if defined(__HIP_DEVICE_COMPILE__)
#define GSP_BSEL(m, a, b) __builtin_amdgcn_bitop3_b32(m, a, b, 0xCA)
  const uint32_t qnx = GSP_BSEL(rb.mx, qhx, qlx), qfx = GSP_BSEL(rb.mx, qlx, qhx);
  const uint32_t qny = GSP_BSEL(rb.my, qhy, qly), qfy = GSP_BSEL(rb.my, qly, qhy);
  const uint32_t qnz = GSP_BSEL(rb.mz, qhz, qlz), qfz = GSP_BSEL(rb.mz, qlz, qhz);
#undef GSP_BSEL
#else
  const uint32_t qnx = rb.negx ? qhx : qlx, qfx = rb.negx ? qlx : qhx;
  const uint32_t qny = rb.negy ? qhy : qly, qfy = rb.negy ? qly : qhy;
  const uint32_t qnz = rb.negz ? qhz : qlz, qfz = rb.negz ? qlz : qhz;
#endif
  uint32_t m = 0;  // child 3 first, so that child k ends at bit k
  GSP_CHILD(m, GSP_UB3, qnx, qfx, qny, qfy, qnz, qfz)
  GSP_CHILD(m, GSP_UB2, qnx, qfx, qny, qfy, qnz, qfz)
  GSP_CHILD(m, GSP_UB1, qnx, qfx, qny, qfy, qnz, qfz)
  GSP_CHILD(m, GSP_UB0, qnx, qfx, qny, qfy, qnz, qfz)
  return m;
}
#undef GSP_CHILD
#undef GSP_UB0
#undef GSP_UB1
#undef GSP_UB2
#undef GSP_UB3

// A node group = the children of ONE node that a ray still has to visit: {gb, gs}.
//   closest hit: gb = child_base, gs = their positions in visiting order (2 bits each under a marker bit;
//                kSeqEmpty: none)
//   any hit:     gb = child_base, gs = mask of their positions (taken in position order)
// A triangle group = the hit leaf children of one node: {tb, tm}: tb = tri_base - ni, tm = hit leaf positions
// (bit p = slot tb + p).
// `TAB(byte offset)` reads the step table (LDS copy on the device, kStepTable on the host).
template <bool ANY>
GSP_HD bool group_empty(uint32_t gs) { return ANY ? gs == 0u : gs <= kSeqEmpty; }
GSP_HD bool tris_empty(uint32_t tm) { return tm == 0u; }
template <bool ANY>
constexpr uint32_t root_group() { return ANY ? 1u : 4u; }  // {base 0, one child at position 0}
template <bool ANY>
constexpr uint32_t no_group() { return ANY ? 0u : kSeqEmpty; }
template <bool ANY, class TAB>
GSP_HD void node_step(const q4* n, const RayBox& rb, float tmin, float tfar, const TAB& tab, uint32_t& gb, uint32_t& gs, uint32_t& tb,
                      uint32_t& tm) {
  const uint32_t miss = node_test(n, rb, tmin, tfar);
  if (ANY) {  // position order, no table (and no LDS round trip on the dependent chain of the step)
    const uint32_t hit = ~miss & 15u;
    gs = hit & (f2u(n[3].x) >> 28);
    tm = hit ^ gs;
  } else {
    const uint32_t pw = rb.octhi ? f2u(n[3].y) : f2u(n[3].x);
    const uint32_t c2 = (pw >> rb.octshift) & 127u;
    const uint32_t e = tab((miss << 7) | c2);
    gs = e & 511u;
    tm = e >> 9;
  }
  gb = f2u(n[2].z);
  tb = f2u(n[2].w);
}
template <bool ANY, class TAB>
GSP_HD uint32_t group_next(uint32_t gb, uint32_t& gs, const RayBox&, const TAB&) {
  uint32_t r;
  if (ANY) {
    r = (uint32_t)__builtin_ctz(gs);
    gs &= gs - 1u;
  } else {
    r = gs & 3u;
    gs >>= 2;
  }
  return (gb + r) << 6;
}
GSP_HD uint32_t tris_next(uint32_t tb, uint32_t& tm) {
  const uint32_t p = (uint32_t)__builtin_ctz(tm);
  tm &= tm - 1u;
  return tb + p;
}

// host / k_finish view of the step table
struct StepTableRef {
  const void* base;
  GSP_HD uint32_t operator()(uint32_t byte_off) const { return ((const uint16_t*)base)[byte_off >> 1]; }
};

// One ray against the wide BVH on one thread, start to finish: the traversal of k_finish (which runs the last few
// thousand paths of a drain to completion without the wavefront queues) and of the host test harness.  Same node
// step, same triangle test and the same min-t / min-id rule as the wave kernel (pt_wavetrace.h), so the hit is the
// same; the visiting order is not (and need not be).  ANY = true: return at the first accepted triangle
// (TerminateOnFirstHit | SkipClosestHitShader).  aux = BSDF type of the accepted triangle (low 3 bits of p0.w).
// STACK: push(uint32_t) / pop() of 32-bit words; at most one group (one word) per tree level.
template <bool ANY, class STACK, class TAB>
GSP_HD bool trace_ray(const q4* __restrict__ nodes, const q4* __restrict__ tris, f3 o, f3 d, float tmin, float tmax, HitRec& h,
                      uint32_t& aux, STACK& stk, const TAB& tab, uint32_t* key_out = nullptr) {
  const RayBox rb = make_raybox(o, d);
  RayShear rs = make_shear(d);
  rs.Sz = comp(rb.inv, rs.kz);  // = 1 / d[kz], the same correctly rounded quotient make_shear computes
  h.t = tmax;
  h.u = h.v = 0.0f;
  h.slot = -1;
  aux = 0;
  uint32_t best_id = 0xffffffffu;
  uint32_t gb = 0, gs = root_group<ANY>();
  int depth = 0;
  for (;;) {
    if (group_empty<ANY>(gs)) {
      if (depth == 0) break;
      --depth;
      const uint32_t e = stk.pop();
      gb = e >> kGroupBits;
      gs = e & kGroupMask;
      continue;
    }
    const q4* nd = (const q4*)((const char*)nodes + group_next<ANY>(gb, gs, rb, tab));
    q4 n[kNodeQuads];
    for (uint32_t k = 0; k < kNodeQuads; ++k) n[k] = nd[k];
    uint32_t ngb, ngs, tb, tm;
    node_step<ANY>(n, rb, tmin, h.t, tab, ngb, ngs, tb, tm);
    while (!tris_empty(tm)) {
      const uint32_t slot = tris_next(tb, tm);
      const q4* p = tris + 3ll * slot;
      const q4 p0 = p[0], p1 = p[1], p2 = p[2];
      float t, u, v;
      if (intersect_tri(mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), o, rs, tmin, tmax, t, u, v)) {
        if (ANY) {
          h.t = t;
          h.slot = (int32_t)slot;
          return true;
        }
        const uint32_t id = f2u(p0.w);
        if (t < h.t || (t == h.t && id < best_id)) {
          h.t = t;
          h.u = u;
          h.v = v;
          h.slot = (int32_t)slot;
          best_id = id;
          aux = id & 7u;
        }
      }
    }
    if (!group_empty<ANY>(ngs)) {
      if (!group_empty<ANY>(gs)) {
        stk.push(pack_group(gb, gs));
        ++depth;
      }
      gb = ngb;
      gs = ngs;
    }
  }
  if (key_out) *key_out = best_id;  // the tie-break key of the hit: a caller that walks two trees picks min (t, key) of the two
  return h.slot >= 0;
}

}  // namespace gsp
