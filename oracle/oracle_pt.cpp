// oracle/oracle_pt.cpp -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// Scalar C++ restatement of the reference's PathTracer pass:
//   raygen.rgen:20-108      camera ray, bounce loop, firefly clamp, Russian
//                           roulette, running-mean accumulate
//   rayhit.rchit:666-797    closest-hit shading vertex
//   miss.rmiss:15-18, shadowmiss.rmiss:6-9
//   PathTracer.cpp:58-93    host marshalling (transformInvT, camera block)
// over a CPU BVH that stands in for the driver's acceleration structure
// (VulkanRays.cpp:6-181; arithmetic vendor-opaque -> PARITY UNPINNED for hit
// t / barycentrics / tie-breaks, see oracle/README.md).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// load this library.  The product (gpuspectral_amd/) never does.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "oracle_bsdf.h"
#include "oracle_texture.h"

namespace orc {

// ---------------------------------------------------------------------------
// Triangle soup in world space + binned-SAH BVH2 (stand-in for BLAS/TLAS).
// ---------------------------------------------------------------------------
struct Tri {
  vec3 v0, v1, v2;
};

struct Node {
  float bmin[3], bmax[3];
  uint32_t left;   // internal: index of left child (right = left + 1); leaf: first tri slot
  uint32_t count;  // 0 = internal, else leaf triangle count
};

struct Hit {
  float t, u, v;
  int32_t prim;  // global triangle index (instance order, then triangle order); -1 = miss
};

struct TravStats {
  uint64_t nodes = 0, tris = 0, rays = 0;
};

struct Accel {
  std::vector<Tri> tris;          // indexed by global triangle id
  std::vector<uint32_t> order;    // BVH leaf slots -> global triangle id
  std::vector<Node> nodes;
  std::vector<uint32_t> tri_instance;  // global triangle id -> instance
};

// Ray/triangle test: watertight algorithm of Woop, Benthin, Wald (JCGT 2013), no culling.
// The Vulkan driver's intersector is opaque (PARITY UNPINNED, README.md); hardware
// traversal is watertight, so the oracle uses the published watertight test.  The
// arithmetic below IS the oracle's definition; the product kernel states the same sequence.
struct Shear {
  int kx, ky, kz;
  float Sx, Sy, Sz;
};
static inline float comp(vec3 v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); }
static inline Shear makeShear(vec3 d) {
  Shear r;
  const float ax = gabs(d.x), ay = gabs(d.y), az = gabs(d.z);
  r.kz = (ax > ay) ? ((ax > az) ? 0 : 2) : ((ay > az) ? 1 : 2);
  r.kx = r.kz == 2 ? 0 : r.kz + 1;
  r.ky = r.kx == 2 ? 0 : r.kx + 1;
  const float dz = comp(d, r.kz);
  if (dz < 0.0f) std::swap(r.kx, r.ky);
  r.Sx = comp(d, r.kx) / dz;
  r.Sy = comp(d, r.ky) / dz;
  r.Sz = 1.0f / dz;
  return r;
}
static inline bool intersectTri(const Tri& tr, vec3 o, const Shear& rs, float tmin, float tmax, float& t, float& u,
                                float& v) {
  const vec3 A = tr.v0 - o, B = tr.v1 - o, C = tr.v2 - o;
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
  u = V * rcp;  // weight of v1
  v = W * rcp;  // weight of v2
  return t > tmin && t < tmax;
}

static void buildAccel(const gsp_scene_desc& sc, Accel& A) {
  // world-space triangles: gl_ObjectToWorldEXT * vec4(pos, 1)   rayhit.rchit:676-681
  for (uint32_t i = 0; i < sc.num_instances; ++i) {
    const gsp_instance& in = sc.instances[i];
    for (uint32_t k = 0; k + 3 <= in.vertex_count; k += 3) {
      const float* p = sc.positions + 3ull * (in.first_vertex + k);
      vec3 p0 = xform_point(in.transform, V(p[0], p[1], p[2]));
      vec3 p1 = xform_point(in.transform, V(p[3], p[4], p[5]));
      vec3 p2 = xform_point(in.transform, V(p[6], p[7], p[8]));
      A.tris.push_back(Tri{p0, p1, p2});
      A.tri_instance.push_back(i);
    }
  }
  const uint32_t n = (uint32_t)A.tris.size();
  A.order.resize(n);
  std::vector<float> cmin(3ull * n), cmax(3ull * n), cen(3ull * n);
  for (uint32_t i = 0; i < n; ++i) {
    A.order[i] = i;
    const Tri& t = A.tris[i];
    vec3 a = t.v0, b = t.v1, c = t.v2;
    float lo[3] = {std::min(a.x, std::min(b.x, c.x)), std::min(a.y, std::min(b.y, c.y)), std::min(a.z, std::min(b.z, c.z))};
    float hi[3] = {std::max(a.x, std::max(b.x, c.x)), std::max(a.y, std::max(b.y, c.y)), std::max(a.z, std::max(b.z, c.z))};
    // conservative padding so that box culling never rejects a triangle the triangle test would
    // accept; it scales with the triangle's extent as well as with its coordinates (flat boxes at
    // coordinate 0 are padded too)
    const float diag = std::max(hi[0] - lo[0], std::max(hi[1] - lo[1], hi[2] - lo[2]));
    // slivers: the float32 triangle test's t error grows like (longest edge)^2 / area, and the box must still hold what
    // the test reports or the winner of two nearly coincident hits depends on the traversal order; the pad grows with
    // that aspect ratio (the product pads the same way, gpuspectral_amd/csrc/pt_bvh.hip k_bake)
    const vec3 e1 = b - a, e2 = c - a, e3 = c - b, cr = cross(e1, e2);
    const float l2 = std::max(std::max(dot(e1, e1), dot(e2, e2)), dot(e3, e3));
    // (a triangle thinner than 1e-6 of its length is a line at the precision of its own coordinates: no extra pad)
    const float aspect = l2 / std::max(std::sqrt(dot(cr, cr)), 1e-30f);
    const float sliver = aspect < 1.0e6f ? std::min(std::max(aspect * (1.0f / 32.0f), 1.0f), 1024.0f) : 1.0f;
    for (int k = 0; k < 3; ++k) {
      float pad = 1e-5f * std::max(std::max(std::fabs(lo[k]), std::fabs(hi[k])), std::max(diag, 1e-3f)) * sliver;
      cmin[3ull * i + k] = lo[k] - pad;
      cmax[3ull * i + k] = hi[k] + pad;
      cen[3ull * i + k] = 0.5f * (lo[k] + hi[k]);
    }
  }
  A.nodes.clear();
  A.nodes.reserve(2ull * n + 1);
  if (n == 0) {
    Node r{};
    r.count = 0;
    r.left = 0;
    for (int k = 0; k < 3; ++k) {
      r.bmin[k] = 1.0f;
      r.bmax[k] = -1.0f;
    }
    A.nodes.push_back(r);
    return;
  }
  struct Work {
    uint32_t node, first, count;
  };
  A.nodes.push_back(Node{});
  std::vector<Work> stack{{0u, 0u, n}};
  constexpr int kBins = 16;
  while (!stack.empty()) {
    Work w = stack.back();
    stack.pop_back();
    float bmin[3] = {1e30f, 1e30f, 1e30f}, bmax[3] = {-1e30f, -1e30f, -1e30f};
    float kmin[3] = {1e30f, 1e30f, 1e30f}, kmax[3] = {-1e30f, -1e30f, -1e30f};
    for (uint32_t i = w.first; i < w.first + w.count; ++i) {
      uint32_t id = A.order[i];
      for (int k = 0; k < 3; ++k) {
        bmin[k] = std::min(bmin[k], cmin[3ull * id + k]);
        bmax[k] = std::max(bmax[k], cmax[3ull * id + k]);
        kmin[k] = std::min(kmin[k], cen[3ull * id + k]);
        kmax[k] = std::max(kmax[k], cen[3ull * id + k]);
      }
    }
    Node nd;
    for (int k = 0; k < 3; ++k) {
      nd.bmin[k] = bmin[k];
      nd.bmax[k] = bmax[k];
    }
    auto makeLeaf = [&]() {
      nd.left = w.first;
      nd.count = w.count;
      A.nodes[w.node] = nd;
    };
    if (w.count <= 2) {
      makeLeaf();
      continue;
    }
    // binned SAH over the three axes
    int bestAxis = -1, bestSplit = -1;
    float bestCost = 1e30f;
    auto area = [](const float* lo, const float* hi) {
      float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
      return 2.0f * (dx * dy + dy * dz + dz * dx);
    };
    for (int ax = 0; ax < 3; ++ax) {
      float ext = kmax[ax] - kmin[ax];
      if (!(ext > 0.0f)) continue;
      uint32_t cnt[kBins] = {};
      float blo[kBins][3], bhi[kBins][3];
      for (int b = 0; b < kBins; ++b)
        for (int k = 0; k < 3; ++k) {
          blo[b][k] = 1e30f;
          bhi[b][k] = -1e30f;
        }
      float scale = (float)kBins / ext;
      for (uint32_t i = w.first; i < w.first + w.count; ++i) {
        uint32_t id = A.order[i];
        int b = std::min(kBins - 1, std::max(0, (int)((cen[3ull * id + ax] - kmin[ax]) * scale)));
        cnt[b]++;
        for (int k = 0; k < 3; ++k) {
          blo[b][k] = std::min(blo[b][k], cmin[3ull * id + k]);
          bhi[b][k] = std::max(bhi[b][k], cmax[3ull * id + k]);
        }
      }
      float rArea[kBins];
      uint32_t rCnt[kBins];
      float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
      uint32_t c = 0;
      for (int b = kBins - 1; b > 0; --b) {
        c += cnt[b];
        for (int k = 0; k < 3; ++k) {
          lo[k] = std::min(lo[k], blo[b][k]);
          hi[k] = std::max(hi[k], bhi[b][k]);
        }
        rCnt[b] = c;
        rArea[b] = c ? area(lo, hi) : 0.0f;
      }
      for (int k = 0; k < 3; ++k) {
        lo[k] = 1e30f;
        hi[k] = -1e30f;
      }
      c = 0;
      for (int b = 0; b < kBins - 1; ++b) {
        c += cnt[b];
        for (int k = 0; k < 3; ++k) {
          lo[k] = std::min(lo[k], blo[b][k]);
          hi[k] = std::max(hi[k], bhi[b][k]);
        }
        if (c == 0 || rCnt[b + 1] == 0) continue;
        float cost = area(lo, hi) * (float)c + rArea[b + 1] * (float)rCnt[b + 1];
        if (cost < bestCost) {
          bestCost = cost;
          bestAxis = ax;
          bestSplit = b;
        }
      }
    }
    uint32_t mid;
    if (bestAxis < 0) {
      if (w.count <= 8) {
        makeLeaf();
        continue;
      }
      mid = w.first + w.count / 2;  // all centroids coincide: split by index
    } else {
      float leafCost = area(bmin, bmax) * (float)w.count;
      if (w.count <= 4 && leafCost <= bestCost + area(bmin, bmax) * 0.5f) {
        makeLeaf();
        continue;
      }
      float scale = (float)kBins / (kmax[bestAxis] - kmin[bestAxis]);
      auto it = std::partition(A.order.begin() + w.first, A.order.begin() + w.first + w.count, [&](uint32_t id) {
        int b = std::min(kBins - 1, std::max(0, (int)((cen[3ull * id + bestAxis] - kmin[bestAxis]) * scale)));
        return b <= bestSplit;
      });
      mid = (uint32_t)(it - A.order.begin());
      if (mid == w.first || mid == w.first + w.count) mid = w.first + w.count / 2;
    }
    nd.count = 0;
    nd.left = (uint32_t)A.nodes.size();
    A.nodes[w.node] = nd;
    A.nodes.push_back(Node{});
    A.nodes.push_back(Node{});
    stack.push_back({nd.left + 1, mid, w.first + w.count - mid});
    stack.push_back({nd.left, w.first, mid - w.first});
  }
}

static inline bool slab(const Node& n, vec3 o, vec3 inv, float tmin, float tmax, float& tnear) {
  float t0x = (n.bmin[0] - o.x) * inv.x, t1x = (n.bmax[0] - o.x) * inv.x;
  float t0y = (n.bmin[1] - o.y) * inv.y, t1y = (n.bmax[1] - o.y) * inv.y;
  float t0z = (n.bmin[2] - o.z) * inv.z, t1z = (n.bmax[2] - o.z) * inv.z;
  // fminf/fmaxf drop NaNs (0 * inf when the origin lies on a slab plane)
  float lo = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), tmin));
  float hi = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), tmax));
  tnear = lo;
  return lo <= hi * 1.000001f;
}

// closest hit: smallest t, ties -> smallest global triangle id (a rule that is
// independent of the BVH topology, so any conforming BVH gives the same hit).
static Hit closestHit(const Accel& A, vec3 o, vec3 d, float tmin, float tmax, TravStats* st) {
  Hit best{tmax, 0.0f, 0.0f, -1};
  if (A.tris.empty()) {  // empty scene: the placeholder root has an inverted box, which a slab test with an infinite 1/d
    if (st) st->rays += 1;  // component does not reject -- and it has no children to descend into
    return best;
  }
  vec3 inv = V(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  const Shear rs = makeShear(d);
  uint32_t stack[128];
  int sp = 0;
  stack[sp++] = 0;
  uint64_t nn = 0, nt = 0;
  while (sp) {
    const Node& n = A.nodes[stack[--sp]];
    float tn;
    ++nn;
    if (!slab(n, o, inv, tmin, best.t, tn)) continue;
    if (n.count) {
      for (uint32_t i = 0; i < n.count; ++i) {
        uint32_t id = A.order[n.left + i];
        float t, u, v;
        ++nt;
        if (intersectTri(A.tris[id], o, rs, tmin, tmax, t, u, v)) {
          if (t < best.t || (t == best.t && (int32_t)id < best.prim)) best = Hit{t, u, v, (int32_t)id};
        }
      }
    } else {
      float ta, tb;
      bool ha = slab(A.nodes[n.left], o, inv, tmin, best.t, ta);
      bool hb = slab(A.nodes[n.left + 1], o, inv, tmin, best.t, tb);
      if (ha && hb) {
        if (ta <= tb) {
          stack[sp++] = n.left + 1;
          stack[sp++] = n.left;
        } else {
          stack[sp++] = n.left;
          stack[sp++] = n.left + 1;
        }
      } else if (ha) {
        stack[sp++] = n.left;
      } else if (hb) {
        stack[sp++] = n.left + 1;
      }
    }
  }
  if (st) {
    st->nodes += nn;
    st->tris += nt;
    st->rays += 1;
  }
  return best;
}

// any hit in (tmin, tmax): TerminateOnFirstHit | SkipClosestHitShader   rayhit.rchit:738-748
static bool anyHit(const Accel& A, vec3 o, vec3 d, float tmin, float tmax, TravStats* st) {
  if (A.tris.empty()) {
    if (st) st->rays += 1;
    return false;
  }
  vec3 inv = V(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  const Shear rs = makeShear(d);
  uint32_t stack[128];
  int sp = 0;
  stack[sp++] = 0;
  uint64_t nn = 0, nt = 0;
  bool found = false;
  while (sp && !found) {
    const Node& n = A.nodes[stack[--sp]];
    float tn;
    ++nn;
    if (!slab(n, o, inv, tmin, tmax, tn)) continue;
    if (n.count) {
      for (uint32_t i = 0; i < n.count; ++i) {
        float t, u, v;
        ++nt;
        if (intersectTri(A.tris[A.order[n.left + i]], o, rs, tmin, tmax, t, u, v)) {
          found = true;
          break;
        }
      }
    } else {
      stack[sp++] = n.left + 1;
      stack[sp++] = n.left;
    }
  }
  if (st) {
    st->nodes += nn;
    st->tris += nt;
    st->rays += 1;
  }
  return found;
}

// ---------------------------------------------------------------------------
// Integrator
// ---------------------------------------------------------------------------
// pt_common.glsl:4-14
struct HitPayload {
  vec3 emitted, weight, origin, direction;
  float directWeight;
  uint32_t seed;
  int wasDelta, countEmitted, done;
  float lastPdf = 0.0f;  // bias attribution only (kQuirkDirectWeight): pdf of the direction that produced this ray
};

struct Counters {
  uint64_t extension_rays = 0, shadow_rays = 0, shaded_vertices = 0;
  TravStats trav;
};

struct SceneCtx {
  gsp_scene_desc sc;
  Accel accel;
  std::vector<float> transformInvT;  // 16 floats per instance, PathTracer.cpp:62
  std::vector<uint32_t> triFirst;    // first global triangle id of each instance
  float fov;
  float toWorld[16];
  TexelDecode decode{nullptr};  // dormant-feature extension (oracle_texture.h)
};

static inline bool isvalid(float x) { return !gisnan(x) && !gisinf(x); }
static inline bool isvalid3(vec3 x) { return isvalid(x.x) && isvalid(x.y) && isvalid(x.z); }

// rayhit.rchit:666-797
// NEE: RenderParams.nee (PathTracer.h:36-41), which the shipped shader replaces by `#define NEE true` (:656); false = the
// other side of its `if (NEE)` / `!NEE ||` branches (:733, :763, :766)
static void closestHitShader(const SceneCtx& S, HitPayload& prd, const Hit& hit, vec3 rayOrigin, Counters& C,
                             bool collect, bool NEE = true) {
  Rng rng{prd.seed};                                                      // :668
  const uint32_t instId = S.accel.tri_instance[hit.prim];
  const gsp_instance& instance = S.sc.instances[instId];                  // :672
  // gl_PrimitiveID is the triangle index inside the instance's mesh       :670
  const float* tInvT = &S.transformInvT[16ull * instId];
  // gl_PrimitiveID: global triangle ids are assigned instance by instance
  const uint32_t firstTri = S.triFirst[instId];
  uint32_t prim = (uint32_t)hit.prim - firstTri;
  uint32_t ix = instance.first_vertex + 3u * prim;                         // :670
  const float* P = S.sc.positions + 3ull * ix;
  const float* Nn = S.sc.normals + 3ull * ix;
  vec3 pos0 = xform_point(instance.transform, V(P[0], P[1], P[2]));        // :676-681
  vec3 pos1 = xform_point(instance.transform, V(P[3], P[4], P[5]));
  vec3 pos2 = xform_point(instance.transform, V(P[6], P[7], P[8]));
  vec3 normal0 = xform_dir(tInvT, V(Nn[0], Nn[1], Nn[2]));                 // :683-688
  vec3 normal1 = xform_dir(tInvT, V(Nn[3], Nn[4], Nn[5]));
  vec3 normal2 = xform_dir(tInvT, V(Nn[6], Nn[7], Nn[8]));

  vec3 bary = V((1.0f - hit.u) - hit.v, hit.u, hit.v);                     // :690
  vec3 position = rayOrigin + prd.direction * hit.t;                       // :692
  vec3 SN = normalize((bary.x * normal0 + bary.y * normal1) + bary.z * normal2);  // :693
  vec3 N = normalize(cross(pos1 - pos0, pos2 - pos0));                     // :694

  vec3 rayDir = prd.direction;                                             // :696
  vec3 emission = V(instance.emission[0], instance.emission[1], instance.emission[2]);
  if (dot(N, -rayDir) < 0.0f) {                                            // :698-707
    if (instance.twofaced == 1 && emission.x == 0.0f && emission.y == 0.0f && emission.z == 0.0f) {
      N = N * -1.0f;
      SN = SN * -1.0f;
    }
  }
  Onb onb = onbCreate(SN);                                                 // :712
  vec3 wo = normalize(onbTransform(onb, -rayDir));                         // :713
  // the reference passes uv = vec2(0) here and never samples (:716,729); with the dormant-feature extension the hit's
  // interpolated uv selects the texel that stands in for kD
  vec3 texel = V(0.0f);
  const vec3* tex = nullptr;
  if (int k = textureOf(S.sc, instance.bsdf)) {
    const float* T = S.sc.uvs + 2ull * ix;
    vec2 uv = {(bary.x * T[0] + bary.y * T[2]) + bary.z * T[4], (bary.x * T[1] + bary.y * T[3]) + bary.z * T[5]};
    texel = textureLookup(S.sc, S.decode, (uint32_t)k - 1u, uv);
    tex = &texel;
  }
  BSDFOutput bsdfRes;
  vec3 wi;
  sampleBSDF(S.sc, instance.bsdf, rng, wo, wi, bsdfRes, tex);              // :716
  float NoW = gabs(wi.z);                                                  // :717
  wi = onbUntransform(onb, wi);                                            // :718

  LightOutput lightRes = sampleLight(S.sc, rng, position);                 // :720
  vec3 L = normalize(lightRes.position - position);                        // :722
  vec3 wL = onbTransform(onb, L);                                          // :723
  float Ldist = length(lightRes.position - position);                      // :724
  const float NoL = gabs(dot(SN, L));                                      // :725
  float lightPdf = lightRes.pdf;                                           // :726

  BSDFOutput lightBsdfRes;
  evalBSDF(S.sc, instance.bsdf, wo, wL, lightBsdfRes, tex);                // :729

  bool neeDone = false;
  const uint32_t btype = instance.bsdf >> 16;
  if (NEE)                                                                 // :733
  if (!bsdfRes.isDelta) {                                                  // :735
    if ((dot(N, -rayDir) > 0.0f && dot(N, L) > 0.0f) || isTransimissionBSDF(btype)) {  // :736
      C.shadow_rays++;
      bool shadowed = anyHit(S.accel, position, L, 0.01f, Ldist - 0.01f, collect ? &C.trav : nullptr);  // :737-748
      if (!shadowed && lightPdf != 0.0f) {                                 // :750
        float w = powerHeuristic(1, lightPdf, 1, bsdfRes.pdf);             // :751
        if (g_quirks_off & kQuirkNeeSampledPdf) w = powerHeuristic(1, lightPdf, 1, lightBsdfRes.pdf);  // pdf of L itself
        prd.emitted = prd.emitted + ((((w * NoL) * lightBsdfRes.bsdf) * prd.weight) * lightRes.emission) / lightPdf;  // :752
        neeDone = true;
      }
    }
  }
  prd.seed = rng.state;                                                    // :759
  float lightFlag = dot(N, -rayDir) > 0.0f ? 1.0f : 0.0f;                  // :760
  if (NEE && prd.countEmitted == 0 && prd.wasDelta == 0) {                 // :763-765
    float directWeight = prd.directWeight;
    if (g_quirks_off & kQuirkDirectWeight) {
      // the textbook weight: light sampling's solid-angle pdf of reaching THIS point of THIS emitter from the previous
      // vertex (sampleTrangleLight's own formula, :141, with the hit triangle's area) against the pdf that sent the ray
      const float A = 0.5f * gabs(length(cross(pos2 - pos0, pos1 - pos0)));
      const float hitPdf = ((hit.t * hit.t) / (gabs(dot(-rayDir, N)) * A)) * (1.0f / (float)S.sc.num_lights);
      const bool isLight = emission.x != 0.0f || emission.y != 0.0f || emission.z != 0.0f;
      directWeight = isLight ? powerHeuristic(1, prd.lastPdf, 1, hitPdf) : 1.0f;
    }
    prd.emitted = prd.emitted + ((directWeight * emission) * lightFlag) * prd.weight;
  }
  if (!NEE || prd.countEmitted == 1 || prd.wasDelta == 1) {                // :766-768
    prd.emitted = prd.emitted + (emission * lightFlag) * prd.weight;
  }
  if (dot(wi, N) <= 0.0f && !isTransimissionBSDF(btype)) {                 // :770-773
    prd.done = 1;
    return;
  }
  if (dot(N, -rayDir) <= 0.0f && !isTransimissionBSDF(btype)) {            // :776-779
    prd.done = 1;
    return;
  }
  if (!isvalid(bsdfRes.pdf) || !isvalid3(bsdfRes.bsdf) || bsdfRes.pdf == 0.0f) {  // :781-784
    prd.done = 1;
    return;
  }
  if (neeDone) {                                                           // :785-790
    prd.directWeight = powerHeuristic(1, bsdfRes.pdf, 1, lightPdf);
  } else {
    prd.directWeight = 1.0f;
  }
  prd.countEmitted = 0;                                                    // :792
  prd.lastPdf = bsdfRes.pdf;
  prd.origin = position + 0.0001f * faceforward(N, -wi, N);                // :793
  prd.direction = wi;                                                      // :794
  prd.weight = prd.weight * ((bsdfRes.bsdf * NoW) / bsdfRes.pdf);          // :795
  prd.wasDelta = bsdfRes.isDelta ? 1 : 0;                                  // :796
}

struct RenderCfg {
  uint32_t width, height;
  uint32_t max_depth, rr_start_depth;
  float clamp;
  bool nee = true;
};

// raygen.rgen:20-25 with z hoisted (it does not depend on the pixel)
static inline vec3 rayDirFn(float w, float h, float px, float py, float z) {
  float x = px - w / 2.0f;
  float y = py - h / 2.0f;
  return normalize(V(-x, y, z));
}

// Debug aid of the test infrastructure (oracle_ray_log): the extension rays of a render, in trace order, so that a GPU /
// oracle disagreement on a path can be pinned to the ray that caused it (single-threaded renders only).
static std::vector<float>* g_ray_log = nullptr;

// raygen.rgen:29-82: one sample of one pixel; returns `result`
static vec3 samplePixel(const SceneCtx& S, const RenderCfg& cfg, uint32_t px, uint32_t py, uint32_t timestamp,
                        float zplane, Counters& C, bool collect) {
  vec3 origin = V(S.toWorld[12], S.toWorld[13], S.toWorld[14]);            // camera.eye = toWorld[3]  Camera.cpp:41-45
  vec3 dl = rayDirFn((float)cfg.width, (float)cfg.height, (float)px, (float)py, zplane);
  vec3 direction = xform_dir(S.toWorld, dl);                               // :33
  direction.y = direction.y * -1.0f;                                       // :35
  Rng rng{pcgHash(tea(cfg.width * py + px, timestamp))};                   // :37
  vec3 result = V(0.0f);
  HitPayload prd;
  prd.weight = V(1.0f);
  prd.directWeight = 1.0f;
  prd.countEmitted = 1;
  prd.wasDelta = 0;
  prd.done = 0;
  prd.seed = rng.state;
  prd.origin = origin;
  prd.direction = direction;
  prd.emitted = V(0.0f);
  uint32_t depth = 0;
  while (true) {                                                           // :51
    prd.emitted = V(0.0f);
    C.extension_rays++;
    if (g_ray_log) {
      const float r[8] = {prd.origin.x, prd.origin.y, prd.origin.z, 0.0f, prd.direction.x, prd.direction.y, prd.direction.z, 1e10f};
      g_ray_log->insert(g_ray_log->end(), r, r + 8);
    }
    Hit h = closestHit(S.accel, prd.origin, prd.direction, 0.0f, 1e10f, collect ? &C.trav : nullptr);  // :53-58
    if (h.prim >= 0) {
      C.shaded_vertices++;
      closestHitShader(S, prd, h, prd.origin, C, collect, cfg.nee);
    } else {
      prd.done = 1;                                                        // miss.rmiss:15-18
      // dormant-feature extension: the escaping path sees the environment map
      if (S.sc.envmap.texels) prd.emitted = envmapLookup(S.sc, prd.direction) * prd.weight;
    }
    rng.state = prd.seed;                                                  // :59
    if ((prd.emitted.x < cfg.clamp && prd.emitted.y < cfg.clamp && prd.emitted.z < cfg.clamp) ||  // :60-63
        (g_quirks_off & kQuirkFireflyClamp)) {
      result = result + prd.emitted;
    }
    if (depth > cfg.rr_start_depth) {                                      // :66-71
      float q = gclamp(gmax(gmax(prd.weight.x, prd.weight.y), prd.weight.z), 0.05f, 1.0f);
      if (randUniform(rng) > q) break;
      prd.weight = prd.weight / q;
    }
    if (depth > cfg.max_depth) break;                                      // :73-75
    if (prd.done == 1) break;                                              // :77-78
    ++depth;
  }
  return result;
}

struct Oracle {
  SceneCtx S;
  std::vector<gsp_instance> instances;
  std::vector<float> positions, normals;
  std::vector<gsp_diffuse_bsdf> b0;
  std::vector<gsp_smooth_dielectric_bsdf> b1;
  std::vector<gsp_smooth_conductor_bsdf> b2;
  std::vector<gsp_smooth_plastic_bsdf> b3;
  std::vector<gsp_rough_conductor_bsdf> b4;
  std::vector<gsp_smooth_floor_bsdf> b5;
  std::vector<gsp_rough_floor_bsdf> b6;
  std::vector<gsp_rough_plastic_bsdf> b7;
  std::vector<gsp_triangle_light> lights;
  std::vector<float> uvs, envTexels;  // dormant-feature extension
  std::vector<gsp_texture> textures;
  std::vector<uint32_t> texels;
  double build_seconds = 0.0;
};

template <class T>
static void copyv(std::vector<T>& dst, const T* src, size_t n) {
  dst.assign(src, src + (src ? n : 0));
}

}  // namespace orc

using namespace orc;

extern "C" {

struct oracle_stats {
  uint64_t extension_rays, shadow_rays, shaded_vertices, samples;
  uint64_t nodes_visited, tris_tested, stat_rays;
  double seconds;
  uint64_t num_triangles, num_bvh_nodes;
};

void* oracle_create(const gsp_scene_desc* sc) {
  if (!sc) return nullptr;
  auto t0 = std::chrono::steady_clock::now();
  Oracle* o = new Oracle();
  copyv(o->instances, sc->instances, sc->num_instances);
  copyv(o->positions, sc->positions, 3 * (size_t)sc->num_vertices);
  copyv(o->normals, sc->normals, 3 * (size_t)sc->num_vertices);
  copyv(o->b0, sc->diffuse_bsdfs, sc->num_bsdfs[0]);
  copyv(o->b1, sc->smooth_dielectric_bsdfs, sc->num_bsdfs[1]);
  copyv(o->b2, sc->smooth_conductor_bsdfs, sc->num_bsdfs[2]);
  copyv(o->b3, sc->smooth_plastic_bsdfs, sc->num_bsdfs[3]);
  copyv(o->b4, sc->rough_conductor_bsdfs, sc->num_bsdfs[4]);
  copyv(o->b5, sc->smooth_floor_bsdfs, sc->num_bsdfs[5]);
  copyv(o->b6, sc->rough_floor_bsdfs, sc->num_bsdfs[6]);
  copyv(o->b7, sc->rough_plastic_bsdfs, sc->num_bsdfs[7]);
  copyv(o->lights, sc->lights, sc->num_lights);
  gsp_scene_desc& d = o->S.sc;
  d = *sc;
  d.instances = o->instances.data();
  d.positions = o->positions.data();
  d.normals = o->normals.data();
  d.diffuse_bsdfs = o->b0.data();
  d.smooth_dielectric_bsdfs = o->b1.data();
  d.smooth_conductor_bsdfs = o->b2.data();
  d.smooth_plastic_bsdfs = o->b3.data();
  d.rough_conductor_bsdfs = o->b4.data();
  d.smooth_floor_bsdfs = o->b5.data();
  d.rough_floor_bsdfs = o->b6.data();
  d.rough_plastic_bsdfs = o->b7.data();
  d.lights = o->lights.data();
  // dormant-feature extension
  const bool textured = sc->num_textures && sc->textures && sc->texels && sc->uvs;
  if (textured) {
    copyv(o->uvs, sc->uvs, 2 * (size_t)sc->num_vertices);
    copyv(o->textures, sc->textures, sc->num_textures);
    copyv(o->texels, sc->texels, (size_t)sc->num_texels);
  }
  d.uvs = textured ? o->uvs.data() : nullptr;
  d.textures = textured ? o->textures.data() : nullptr;
  d.texels = textured ? o->texels.data() : nullptr;
  d.num_textures = textured ? sc->num_textures : 0;
  o->S.decode = TexelDecode(sc->texel_decode);
  d.texel_decode = nullptr;  // (copied into S.decode)
  if (sc->envmap.texels && sc->envmap.width && sc->envmap.height) {
    copyv(o->envTexels, sc->envmap.texels, 4 * (size_t)sc->envmap.width * sc->envmap.height);
    d.envmap.texels = o->envTexels.data();
  } else {
    d.envmap.texels = nullptr;
  }
  // PathTracer.cpp:62  transformInvT = inverse(transpose(M))
  o->S.transformInvT.resize(16ull * sc->num_instances);
  uint32_t first = 0;
  for (uint32_t i = 0; i < sc->num_instances; ++i) {
    float tr[16];
    mat4_transpose(o->instances[i].transform, tr);
    mat4_inverse(tr, &o->S.transformInvT[16ull * i]);
    o->S.triFirst.push_back(first);
    first += o->instances[i].vertex_count / 3;
  }
  for (int k = 0; k < 16; ++k) o->S.toWorld[k] = sc->camera.to_world[k];
  o->S.fov = sc->camera.fov;
  buildAccel(d, o->S.accel);
  o->build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return o;
}

void oracle_destroy(void* h) { delete (Oracle*)h; }

double oracle_build_seconds(void* h) { return ((Oracle*)h)->build_seconds; }

// Adds `spp` samples (timestamps first_timestamp ..) to `accum` (RGBA32F,
// compact over pixel_ids when given, else width*height) with the reference's
// running mean (raygen.rgen:84-108).  threads <= 0 -> hardware concurrency.
int oracle_render(void* h, uint32_t width, uint32_t height, const uint32_t* pixel_ids, uint64_t num_pixels,
                  const gsp_render_params* rp, float* accum, int threads, int collect_trav, oracle_stats* out) {
  Oracle* o = (Oracle*)h;
  if (!o || !rp || !accum) return 1;
  RenderCfg cfg{width, height, rp->max_depth, rp->rr_start_depth, rp->clamp, rp->disable_nee == 0};
  const uint64_t npix = pixel_ids ? num_pixels : (uint64_t)width * height;
  // raygen.rgen:22  z = (max(size.x,size.y)/2) / tan(fov/2); tan() evaluated on the host
  const float zplane = (gmax((float)width, (float)height) / 2.0f) / tanf(o->S.fov / 2.0f);
  int nt = threads > 0 ? threads : (int)std::max(1u, std::thread::hardware_concurrency());
  std::vector<Counters> counters(nt);
  std::atomic<uint64_t> next{0};
  const uint64_t chunk = 256;
  auto t0 = std::chrono::steady_clock::now();
  auto worker = [&](int tid) {
    Counters C;  // thread-local: neighbouring elements of `counters` share cache lines
    struct Flush {
      Counters& dst;
      Counters& src;
      ~Flush() { dst = src; }
    } flush{counters[tid], C};
    for (;;) {
      uint64_t b = next.fetch_add(chunk);
      if (b >= npix) break;
      uint64_t e = std::min(npix, b + chunk);
      for (uint64_t i = b; i < e; ++i) {
        uint32_t gid = pixel_ids ? pixel_ids[i] : (uint32_t)i;
        uint32_t px = gid % width, py = gid / width;
        float* dst = accum + 4 * i;
        for (uint32_t s = 0; s < rp->spp; ++s) {
          uint32_t ts = rp->first_timestamp + s;
          vec3 accumColor = samplePixel(o->S, cfg, px, py, ts, zplane, C, collect_trav != 0);
          if (ts > 0) {                                                    // raygen.rgen:86-91
            const float a = 1.0f / (float)(ts + 1u);
            vec3 prev = V(dst[0], dst[1], dst[2]);
            accumColor = prev * (1.0f - a) + accumColor * a;               // mix(prev, new, a)
          }
          if (!(gisnan(accumColor.x) || gisnan(accumColor.y) || gisnan(accumColor.z))) {  // :106-108
            dst[0] = accumColor.x;
            dst[1] = accumColor.y;
            dst[2] = accumColor.z;
            dst[3] = 1.0f;
          }
        }
      }
    }
  };
  if (nt == 1) {
    worker(0);
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back(worker, t);
    for (auto& t : th) t.join();
  }
  double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (out) {
    *out = oracle_stats{};
    for (auto& c : counters) {
      out->extension_rays += c.extension_rays;
      out->shadow_rays += c.shadow_rays;
      out->shaded_vertices += c.shaded_vertices;
      out->nodes_visited += c.trav.nodes;
      out->tris_tested += c.trav.tris;
      out->stat_rays += c.trav.rays;
    }
    out->samples = npix * rp->spp;
    out->seconds = sec;
    out->num_triangles = o->S.accel.tris.size();
    out->num_bvh_nodes = o->S.accel.nodes.size();
  }
  return 0;
}

// Bias attribution (oracle_bsdf.h, kQuirk*): the mask stays set until the next call; 0 = the reference as shipped.
// Process-wide and not synchronised with a running oracle_render: set it between renders.
void oracle_set_quirks_off(uint32_t mask) { g_quirks_off = mask; }
uint32_t oracle_get_quirks_off() { return g_quirks_off; }

// enable != 0: start logging the extension rays of single-threaded oracle_render calls; 0: stop and drop the log.
// oracle_ray_log_read copies up to cap rays (8 floats each) and returns how many were logged.
void oracle_ray_log(int enable) {
  delete g_ray_log;
  g_ray_log = enable ? new std::vector<float>() : nullptr;
}
uint64_t oracle_ray_log_read(float* out, uint64_t cap) {
  if (!g_ray_log) return 0;
  const uint64_t n = g_ray_log->size() / 8;
  if (out) memcpy(out, g_ray_log->data(), sizeof(float) * 8 * std::min(n, cap));
  return n;
}

// rays = n * {ox,oy,oz,tmin, dx,dy,dz,tmax}; hits = n * {t,u,v,prim}
int oracle_trace(void* h, const float* rays, uint64_t n, int any_hit, void* hits_out) {
  Oracle* o = (Oracle*)h;
  if (!o) return 1;
  struct HitRec {
    float t, u, v;
    int32_t prim;
  };
  HitRec* out = (HitRec*)hits_out;
  for (uint64_t i = 0; i < n; ++i) {
    const float* r = rays + 8 * i;
    vec3 org = V(r[0], r[1], r[2]), dir = V(r[4], r[5], r[6]);
    if (any_hit) {
      bool occ = anyHit(o->S.accel, org, dir, r[3], r[7], nullptr);
      out[i] = HitRec{0.0f, 0.0f, 0.0f, occ ? 0 : -1};
    } else {
      Hit hh = closestHit(o->S.accel, org, dir, r[3], r[7], nullptr);
      out[i] = HitRec{hh.prim >= 0 ? hh.t : 0.0f, hh.u, hh.v, hh.prim};
    }
  }
  return 0;
}

// primary ray of pixel (px,py): out = {ox,oy,oz, dx,dy,dz}   raygen.rgen:20-35
void oracle_primary_ray(void* h, uint32_t width, uint32_t height, uint32_t px, uint32_t py, float* out6) {
  Oracle* o = (Oracle*)h;
  const float zplane = (gmax((float)width, (float)height) / 2.0f) / tanf(o->S.fov / 2.0f);
  vec3 dl = rayDirFn((float)width, (float)height, (float)px, (float)py, zplane);
  vec3 d = xform_dir(o->S.toWorld, dl);
  d.y = d.y * -1.0f;
  out6[0] = o->S.toWorld[12];
  out6[1] = o->S.toWorld[13];
  out6[2] = o->S.toWorld[14];
  out6[3] = d.x;
  out6[4] = d.y;
  out6[5] = d.z;
}

// ---- known-answer hooks -----------------------------------------------------
uint32_t oracle_tea(uint32_t a, uint32_t b) { return tea(a, b); }
uint32_t oracle_pcg_hash(uint32_t v) { return pcgHash(v); }
// out[0..n) = successive randPcg() outputs from state `seed`; returns final state
uint32_t oracle_rand_pcg(uint32_t seed, uint32_t n, uint32_t* out) {
  Rng g{seed};
  for (uint32_t i = 0; i < n; ++i) out[i] = randPcg(g);
  return g.state;
}
float oracle_rand_uniform(uint32_t seed) {
  Rng g{seed};
  return randUniform(g);
}

// BSDF sample: out = {wi.xyz, bsdf.rgb, pdf, isDelta, final rng state (as float bits)}
void oracle_bsdf_sample(void* h, uint32_t handle, const float* wo, uint32_t seed, float* out9) {
  Oracle* o = (Oracle*)h;
  Rng g{seed};
  vec3 wi;
  BSDFOutput r;
  sampleBSDF(o->S.sc, handle, g, V(wo[0], wo[1], wo[2]), wi, r);
  out9[0] = wi.x;
  out9[1] = wi.y;
  out9[2] = wi.z;
  out9[3] = r.bsdf.x;
  out9[4] = r.bsdf.y;
  out9[5] = r.bsdf.z;
  out9[6] = r.pdf;
  out9[7] = r.isDelta ? 1.0f : 0.0f;
  out9[8] = u2f(g.state);
}
// BSDF eval: out = {bsdf.rgb, pdf, isDelta}
void oracle_bsdf_eval(void* h, uint32_t handle, const float* wo, const float* wi, float* out5) {
  Oracle* o = (Oracle*)h;
  BSDFOutput r;
  evalBSDF(o->S.sc, handle, V(wo[0], wo[1], wo[2]), V(wi[0], wi[1], wi[2]), r);
  out5[0] = r.bsdf.x;
  out5[1] = r.bsdf.y;
  out5[2] = r.bsdf.z;
  out5[3] = r.pdf;
  out5[4] = r.isDelta ? 1.0f : 0.0f;
}
// light sample: out = {pos.xyz, emission.rgb, pdf, final rng state bits}
void oracle_sample_light(void* h, const float* pos, uint32_t seed, float* out8) {
  Oracle* o = (Oracle*)h;
  Rng g{seed};
  LightOutput r = sampleLight(o->S.sc, g, V(pos[0], pos[1], pos[2]));
  out8[0] = r.position.x;
  out8[1] = r.position.y;
  out8[2] = r.position.z;
  out8[3] = r.emission.x;
  out8[4] = r.emission.y;
  out8[5] = r.emission.z;
  out8[6] = r.pdf;
  out8[7] = u2f(g.state);
}
// Onb (pt_common.glsl:122-151): out15 = {tangent, binormal, normal, onbTransform(v), onbUntransform(v)}
void oracle_onb(const float* nrm, const float* v, float* out15) {
  Onb o = onbCreate(V(nrm[0], nrm[1], nrm[2]));
  vec3 a = onbTransform(o, V(v[0], v[1], v[2])), b = onbUntransform(o, V(v[0], v[1], v[2]));
  const vec3 r[5] = {o.tangent, o.binormal, o.normal, a, b};
  for (int k = 0; k < 5; ++k) {
    out15[3 * k] = r[k].x;
    out15[3 * k + 1] = r[k].y;
    out15[3 * k + 2] = r[k].z;
  }
}
float oracle_power_heuristic(float f, float g) { return powerHeuristic(1, f, 1, g); }
float oracle_cosine_pdf(float z) { return cosineHemispherePdf(V(0.0f, 0.0f, z)); }
int oracle_is_transmission(uint32_t handle) { return isTransimissionBSDF(handle >> 16) ? 1 : 0; }
// deterministic transcendentals (for the accuracy test against libm)
void oracle_det_math(const float* x, uint64_t n, float* s, float* c, float* lg, float* ex) {
  for (uint64_t i = 0; i < n; ++i) {
    det_sincosf(x[i], &s[i], &c[i]);
    lg[i] = det_logf(x[i]);
    ex[i] = det_expf(x[i]);
  }
}
// glm::inverse(glm::transpose(M)) as the oracle computes it (16 floats in/out)
void oracle_transform_inv_t(const float* m, float* out) {
  float tr[16];
  mat4_transpose(m, tr);
  mat4_inverse(tr, out);
}

}  // extern "C"
