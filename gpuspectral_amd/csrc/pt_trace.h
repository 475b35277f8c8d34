// pt_trace.h -- BVH node / triangle packet layouts in HBM and the per-ray
// traversal loop of the extend (closest hit) and connect (any hit) kernels.
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
//   (the host-side test harness tests/emu keeps an uncompressed binary layout of its own:
//    q0 = {lmin.xyz, lmax.x} q1 = {lmax.yz, rmin.xy} q2 = {rmin.z, rmax.xyz} q3 = {left, right, -, -})
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

// Slab test.  min/max here are the NaN-dropping IEEE forms, so a NaN from
// 0*inf (origin on a slab plane, zero direction component) leaves that slab
// unconstrained: conservative.  The far bound is relaxed by 8 ulp.
GSP_HD float fmin_(float a, float b) { return __builtin_fminf(a, b); }
GSP_HD float fmax_(float a, float b) { return __builtin_fmaxf(a, b); }
GSP_HD bool slab(float bx0, float by0, float bz0, float bx1, float by1, float bz1, f3 o, f3 inv, float tmin,
                 float tmax, float& tnear) {
  float t0x = (bx0 - o.x) * inv.x, t1x = (bx1 - o.x) * inv.x;
  float t0y = (by0 - o.y) * inv.y, t1y = (by1 - o.y) * inv.y;
  float t0z = (bz0 - o.z) * inv.z, t1z = (bz1 - o.z) * inv.z;
  float lo = fmax_(fmax_(fmin_(t0x, t1x), fmin_(t0y, t1y)), fmax_(fmin_(t0z, t1z), tmin));
  float hi = fmin_(fmin_(fmax_(t0x, t1x), fmax_(t0y, t1y)), fmin_(fmax_(t0z, t1z), tmax));
  tnear = lo;
  return lo <= hi * 1.000001f;  // 8 ulp of slack on top of the padded boxes
}

struct TraceCounters {
  uint32_t nodes, tris;
};

// One ray against the BVH.  ANY = true: return at the first accepted triangle
// (TerminateOnFirstHit | SkipClosestHitShader).  `Stack` supplies push/pop/empty.
template <bool ANY, bool STATS, class Stack>
GSP_HD bool traverse(const q4* __restrict__ nodes, const q4* __restrict__ tris, int32_t root, f3 o, f3 d, float tmin,
                     float tmax, Stack& stk, HitRec& hit, TraceCounters& cnt) {
  hit.t = tmax;
  hit.u = 0.0f;
  hit.v = 0.0f;
  hit.slot = -1;
  uint32_t best_id = 0xffffffffu;
  const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  const RayShear rs = make_shear(d);
  int32_t cur = root;
  bool found = false;
  for (;;) {
    if (cur >= 0) {
      const q4* n = nodes + 4ll * cur;
      const q4 q0 = n[0], q1 = n[1], q2 = n[2], q3 = n[3];
      if (STATS) cnt.nodes++;
      float tl, tr;
      const bool hl = slab(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, o, inv, tmin, hit.t, tl);
      const bool hr = slab(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, o, inv, tmin, hit.t, tr);
      const int32_t cl = (int32_t)f2u(q3.x), cr = (int32_t)f2u(q3.y);
      if (hl && hr) {
        const bool left_first = tl <= tr;
        stk.push(left_first ? cr : cl);
        cur = left_first ? cl : cr;
        continue;
      } else if (hl) {
        cur = cl;
        continue;
      } else if (hr) {
        cur = cr;
        continue;
      }
    } else {
      const uint32_t c = (uint32_t)~cur;
      const uint32_t first = c >> 2, count = (c & 3u) + 1u;
      for (uint32_t k = 0; k < count; ++k) {
        const q4* p = tris + 3ll * (first + k);
        const q4 p0 = p[0], p1 = p[1], p2 = p[2];
        if (STATS) cnt.tris++;
        float t, u, v;
        if (intersect_tri(mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), o, rs, tmin, tmax, t, u,
                          v)) {
          if (ANY) {
            hit.t = t;
            hit.slot = (int32_t)(first + k);
            found = true;
            break;
          }
          const uint32_t id = f2u(p0.w);
          if (t < hit.t || (t == hit.t && id < best_id)) {
            hit.t = t;
            hit.u = u;
            hit.v = v;
            hit.slot = (int32_t)(first + k);
            best_id = id;
          }
        }
      }
      if (ANY && found) break;
    }
    if (stk.empty()) break;
    cur = stk.pop();
  }
  return hit.slot >= 0;
}

}  // namespace gsp
