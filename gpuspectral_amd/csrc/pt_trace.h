// pt_trace.h -- BVH node / triangle packet layouts in HBM, the node encoder, the 4-wide node step and the
// triangle test every traversal shares (wave kernel pt_wavetrace.h, k_finish, host test harness tests/emu).
//
// Replaces the driver's acceleration-structure traversal behind traceRayEXT
// (S/assets/shaders/raygen.rgen:58, rayhit.rchit:738-748; build call sites
// S/backend/vulkan/VulkanRays.cpp:6-181).
//
// HBM layout (all records are whole float4s so one lane moves 16 B per load):
//   4-wide node (64 B, compressed; the kernels are bound by the number of divergent 16-B load
//   instructions per node step, so a node is 4 loads instead of the 7 an uncompressed one needs):
//                  q0 = {origin.x, origin.y, origin.z, scale.x}
//                       plane = origin + q * scale, q in 0..255, scale = a power of two per axis
//                  q1 = {qlo.x[child0..3], qlo.y[0..3], qlo.z[0..3], qhi.x[0..3]}   one byte per child
//                  q2 = {qhi.y[0..3], qhi.z[0..3], bits(child0), bits(child1)}
//                  q3 = {bits(child2), bits(child3), scale.y, scale.z}
//           child boxes are quantised outward (floor / ceil, verified against the decode), so they
//           contain the exact boxes: culling stays conservative and results do not change
//           child >= 0 : inner node, BYTE offset of the node (index * 64) from the node array
//           unused slot: inverted box (qlo = 255, qhi = 0) + the leaf of the degenerate triangle kept
//                        in slot num_tris, so the traversal has no empty-slot test
//           child <  0 : leaf, c = ~child, first slot = c >> 2, count = (c & 3) + 1
//   triangle packet (48 B, in BVH leaf order):
//                  p0 = {v0.x, v0.y, v0.z, bits(global triangle id)}
//                  p1 = {v1.x, v1.y, v1.z, bits(BSDF type of the owning instance)}
//                  p2 = {v2.x, v2.y, v2.z, -}
//   Closest hit = smallest t, ties broken by the smaller global triangle id, so
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

constexpr int kLeafMaxTris = 4;
// A/B variant ("BVH top levels through LDS"): the first GSP_LDS_TOP slots of the node array hold a
// breadth-first copy of the top of the tree (root = slot 0; 21 = three levels, 85 = four), which k_trace stages into
// LDS once per block and reads with ds_read_b128 instead of global loads.  0 = off.
#ifndef GSP_LDS_TOP
#define GSP_LDS_TOP 0
#endif
constexpr uint32_t kTopNodes = GSP_LDS_TOP;
constexpr int32_t kEmptyChild = 0x7ffffffe;  // unused slot of a 4-wide node
GSP_HD int32_t make_leaf(uint32_t first_slot, uint32_t count) { return ~(int32_t)((first_slot << 2) | (count - 1u)); }

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

// min/max here are the NaN-dropping IEEE forms, so a NaN from 0*inf (origin on a slab plane, zero direction
// component) leaves that slab unconstrained: conservative.
GSP_HD float fmin_(float a, float b) { return __builtin_fminf(a, b); }
GSP_HD float fmax_(float a, float b) { return __builtin_fmaxf(a, b); }

// ---- compressed 4-wide node: encoder (device BVH build, pt_bvh.hip; host test harness, tests/emu) --------------
struct Entry4 {
  q4 lo, hi;
  int32_t code;
};
GSP_HD q4 make_q4(float x, float y, float z, float w) {
  q4 r;
  r.x = x;
  r.y = y;
  r.z = z;
  r.w = w;
  return r;
}
// Writes one compressed 4-wide node (64 B, layout at the top of this file) from up to four child entries.
// Child boxes are quantised outward and checked against the decode (plane = fma(q, scale, origin)).
GSP_HD void encode_node4(q4* __restrict__ o, const Entry4* e, int cnt, uint32_t dummy_slot) {
  float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  for (int k = 0; k < cnt; ++k) {
    lo[0] = fmin_(lo[0], e[k].lo.x); lo[1] = fmin_(lo[1], e[k].lo.y); lo[2] = fmin_(lo[2], e[k].lo.z);
    hi[0] = fmax_(hi[0], e[k].hi.x); hi[1] = fmax_(hi[1], e[k].hi.y); hi[2] = fmax_(hi[2], e[k].hi.z);
  }
  float scale[3];
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
  uint32_t q[6] = {0, 0, 0, 0, 0, 0};  // qlo.x, qlo.y, qlo.z, qhi.x, qhi.y, qhi.z : one byte per child
  for (int k = 0; k < cnt; ++k) {
    const float cl[3] = {e[k].lo.x, e[k].lo.y, e[k].lo.z}, ch[3] = {e[k].hi.x, e[k].hi.y, e[k].hi.z};
    for (int a = 0; a < 3; ++a) {
      float ql = fmin_(fmax_(__builtin_floorf((cl[a] - lo[a]) / scale[a]), 0.0f), 255.0f);
      while (ql > 0.0f && __builtin_fmaf(ql, scale[a], lo[a]) > cl[a]) ql -= 1.0f;  // decoded plane must not exceed the box
      float qh = fmin_(fmax_(__builtin_ceilf((ch[a] - lo[a]) / scale[a]), 0.0f), 255.0f);
      while (qh < 255.0f && __builtin_fmaf(qh, scale[a], lo[a]) < ch[a]) qh += 1.0f;
      q[a] |= (uint32_t)ql << (8 * k);
      q[3 + a] |= (uint32_t)qh << (8 * k);
    }
  }
  // unused slots: inverted box (near plane beyond the far plane: misses) that leads to the degenerate
  // triangle, so the traversal needs no empty-slot test and a rounding fluke costs one triangle test
  for (int k = cnt; k < 4; ++k)
    for (int a = 0; a < 3; ++a) q[a] |= 255u << (8 * k);
  int32_t code[4];
  for (int k = 0; k < 4; ++k) code[k] = k < cnt ? e[k].code : make_leaf(dummy_slot, 1);
  o[0] = make_q4(lo[0], lo[1], lo[2], scale[0]);
  o[1] = make_q4(u2f(q[0]), u2f(q[1]), u2f(q[2]), u2f(q[3]));
  o[2] = make_q4(u2f(q[4]), u2f(q[5]), u2f((uint32_t)code[0]), u2f((uint32_t)code[1]));
  o[3] = make_q4(u2f((uint32_t)code[2]), u2f((uint32_t)code[3]), scale[1], scale[2]);
}

// ---- compressed 4-wide node: one traversal step (the ONE decode + slab test + ordering every traversal uses:
// k_trace and k_finish on the device, tests/emu on the host) ------------------------------------------------------
// Per-ray constants of the step.
struct RayBox {
  f3 o, inv;               // origin, 1 / direction
  f3 invc;                 // inv clamped to +-2^64 (box tests only; the triangle test uses the exact ray)
  bool negx, negy, negz;   // sign of 1/d per axis: which plane of a slab is the near one
};
GSP_HD RayBox make_raybox(f3 o, f3 d) {
  RayBox r;
  r.o = o;
  r.inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  r.negx = r.inv.x < 0.0f;
  r.negy = r.inv.y < 0.0f;
  r.negz = r.inv.z < 0.0f;
  const float big = 18446744073709551616.0f;  // 2^64
  r.invc = mk3(fmin_(fmax_(r.inv.x, -big), big), fmin_(fmax_(r.inv.y, -big), big), fmin_(fmax_(r.inv.z, -big), big));
  return r;
}
// Tests the four child boxes of the node {n0..n3} against the ray segment [tmin, tfar] and returns the children
// ordered by entry distance: e0 nearest .. e3, of which the first `hits` are hit (the rest are unspecified).
// The return value is hits * UNIT (the wave kernel keeps the count in stack-offset units).
//   decode: plane = origin + q * scale (per-axis power of two, stored as a float); its ray parameter is
//   t = fma(q, scale / d, (origin - o) / d): one fma per plane after six multiplies per node (scale / d is exact up to
//   the rounding of 1/d: scale is a power of two).  1/d is CLAMPED to +-2^64 for this: with an infinite 1/d (a zero
//   direction component) every plane of that axis would be fma(q, inf, +-inf) = NaN, the slab would stop
//   constraining and axis-parallel rays would lose their culling (measured in round 1 without the clamp: +13 % / +47 %
//   kernel time).  With the clamp such a ray sees t = (plane - o) * 2^64: both planes of a slab it is outside of land
//   far beyond tmax <= 1e10 on the same side (culled, correctly: the ray never enters), a slab it is inside of gives
//   -huge / +huge (unconstrained), and no product can overflow (|scale|, |origin - o| < 2^40).  The rounding error of
//   t is of the same order as with an uncompressed (b - o) * (1/d) test -- 2^-24 |origin - o| / |d| -- and is covered
//   by the outward quantisation, the padded leaf boxes and the 8-ulp slack on the far bound.  r02 A/B against
//   (fma(q, scale, origin - o)) * (1/d): closest-hit kernel -3.4 %, any-hit -1.4 %, bit-identical images
//   (profiles/r02_ab_fold_ldstop.txt).
//   Near / far planes are picked by the sign of 1/d instead of min / max per child: for inv > 0
//   (lo - o) * inv <= (hi - o) * inv by monotonic rounding, so the values are the ones min / max would return; a
//   NaN (0 * inf) is dropped by max / min and leaves that side unconstrained.  The far bound is relaxed by 8 ulp on
//   top of the padded boxes.  Unused slots carry an inverted box and the degenerate triangle's leaf: no test needed.
template <uint32_t UNIT>
GSP_HD uint32_t node4_step(const q4& n0, const q4& n1, const q4& n2, const q4& n3, const RayBox& rb, float tmin, float tfar,
                           int32_t& e0, int32_t& e1, int32_t& e2, int32_t& e3) {
  // per-node constants of the decode: scale / d and (origin - o) / d per axis (6 multiplies), then ONE fma per plane
  const float sx = n0.w * rb.invc.x, sy = n3.z * rb.invc.y, sz = n3.w * rb.invc.z;
  const float dx = (n0.x - rb.o.x) * rb.invc.x, dy = (n0.y - rb.o.y) * rb.invc.y, dz = (n0.z - rb.o.z) * rb.invc.z;
  const uint32_t qlx = f2u(n1.x), qly = f2u(n1.y), qlz = f2u(n1.z), qhx = f2u(n1.w), qhy = f2u(n2.x), qhz = f2u(n2.y);
  const uint32_t qnx = rb.negx ? qhx : qlx, qfx = rb.negx ? qlx : qhx;
  const uint32_t qny = rb.negy ? qhy : qly, qfy = rb.negy ? qly : qhy;
  const uint32_t qnz = rb.negz ? qhz : qlz, qfz = rb.negz ? qlz : qhz;
  float lo4[4];
  bool hit4[4];
// (float)((q >> 8k) & 0xff) compiles to v_cvt_f32_ubyteK
#define GSP_UB0(q) ((float)((q) & 0xffu))
#define GSP_UB1(q) ((float)(((q) >> 8) & 0xffu))
#define GSP_UB2(q) ((float)(((q) >> 16) & 0xffu))
#define GSP_UB3(q) ((float)((q) >> 24))
#define GSP_CHILD(K, CVT)                                                                                         \
  {                                                                                                               \
    const float tnx = __builtin_fmaf(CVT(qnx), sx, dx), tfx = __builtin_fmaf(CVT(qfx), sx, dx);                   \
    const float tny = __builtin_fmaf(CVT(qny), sy, dy), tfy = __builtin_fmaf(CVT(qfy), sy, dy);                   \
    const float tnz = __builtin_fmaf(CVT(qnz), sz, dz), tfz = __builtin_fmaf(CVT(qfz), sz, dz);                   \
    const float lo = fmax_(fmax_(tnx, tny), fmax_(tnz, tmin));                                                    \
    const float hi = fmin_(fmin_(tfx, tfy), fmin_(tfz, tfar));                                                    \
    lo4[K] = lo;                                                                                                  \
    hit4[K] = lo <= hi * 1.000001f;                                                                               \
  }
  GSP_CHILD(0, GSP_UB0)
  GSP_CHILD(1, GSP_UB1)
  GSP_CHILD(2, GSP_UB2)
  GSP_CHILD(3, GSP_UB3)
#undef GSP_CHILD
#undef GSP_UB0
#undef GSP_UB1
#undef GSP_UB2
#undef GSP_UB3
  const bool h0 = hit4[0], h1 = hit4[1], h2 = hit4[2], h3 = hit4[3];
  // order the hit children by entry distance: 5-comparator network on {distance bits, child}
  // (entry distances are >= tmin >= 0, so their bit patterns order like unsigned integers; a miss sorts last)
  uint32_t k0 = h0 ? f2u(lo4[0]) : 0xffffffffu, k1 = h1 ? f2u(lo4[1]) : 0xffffffffu;
  uint32_t k2 = h2 ? f2u(lo4[2]) : 0xffffffffu, k3 = h3 ? f2u(lo4[3]) : 0xffffffffu;
  e0 = (int32_t)f2u(n2.z);
  e1 = (int32_t)f2u(n2.w);
  e2 = (int32_t)f2u(n3.x);
  e3 = (int32_t)f2u(n3.y);
#define GSP_CSWAP(ka, kb, ea, eb)   \
  {                                 \
    const bool sw = kb < ka;        \
    const uint32_t tk = sw ? kb : ka; \
    kb = sw ? ka : kb;              \
    ka = tk;                        \
    const int32_t te = sw ? eb : ea; \
    eb = sw ? ea : eb;              \
    ea = te;                        \
  }
  GSP_CSWAP(k0, k1, e0, e1)
  GSP_CSWAP(k2, k3, e2, e3)
  GSP_CSWAP(k0, k2, e0, e2)
  GSP_CSWAP(k1, k3, e1, e3)
  GSP_CSWAP(k1, k2, e1, e2)
#undef GSP_CSWAP
  return (h0 ? UNIT : 0u) + (h1 ? UNIT : 0u) + (h2 ? UNIT : 0u) + (h3 ? UNIT : 0u);
}

// One ray against the 4-wide BVH on one thread, start to finish: the traversal of k_finish (which runs the last few
// thousand paths of a drain to completion without the wavefront queues) and of the host test harness.  Same node
// step, same triangle test and the same min-t / min-id rule as the wave kernel (pt_wavetrace.h), so the hit is the
// same; the visiting order is not (and need not be).  ANY = true: return at the first accepted triangle
// (TerminateOnFirstHit | SkipClosestHitShader).  aux = p1.w of the accepted triangle (BSDF type).
constexpr int kLaneStackDepth = 96;
template <bool ANY>
GSP_HD bool trace_ray4(const q4* __restrict__ nodes, const q4* __restrict__ tris, int32_t root, f3 o, f3 d, float tmin,
                       float tmax, HitRec& h, uint32_t& aux) {
  int32_t stack[kLaneStackDepth];
  int sp = 0;
  const RayBox rb = make_raybox(o, d);
  RayShear rs = make_shear(d);
  rs.Sz = comp(rb.inv, rs.kz);  // = 1 / d[kz], the same correctly rounded quotient make_shear computes
  h.t = tmax;
  h.u = h.v = 0.0f;
  h.slot = -1;
  aux = 0;
  uint32_t best_id = 0xffffffffu;
  int32_t cur = root;
  for (;;) {
    if (cur >= 0) {
      const q4* nd = (const q4*)((const char*)nodes + (uint32_t)cur);
      int32_t e[4];
      const uint32_t hits = node4_step<1u>(nd[0], nd[1], nd[2], nd[3], rb, tmin, h.t, e[0], e[1], e[2], e[3]);
      for (uint32_t k = hits; k-- > 1u;) stack[sp++] = e[k];  // farthest first
      if (hits != 0u) {
        cur = e[0];
        continue;
      }
    } else {
      const uint32_t cc = (uint32_t)~cur;
      const uint32_t first = cc >> 2, count = (cc & 3u) + 1u;
      for (uint32_t k = 0; k < count; ++k) {
        const q4* p = tris + 3ll * (first + k);
        const q4 p0 = p[0], p1 = p[1], p2 = p[2];
        float t, u, v;
        if (intersect_tri(mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), o, rs, tmin, tmax, t, u, v)) {
          if (ANY) {
            h.t = t;
            h.slot = (int32_t)(first + k);
            return true;
          }
          const uint32_t id = f2u(p0.w);
          if (t < h.t || (t == h.t && id < best_id)) {
            h.t = t;
            h.u = u;
            h.v = v;
            h.slot = (int32_t)(first + k);
            best_id = id;
            aux = f2u(p1.w);
          }
        }
      }
    }
    if (sp == 0) break;
    cur = stack[--sp];
  }
  return h.slot >= 0;
}

}  // namespace gsp
