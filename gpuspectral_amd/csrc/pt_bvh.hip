// pt_bvh.hip -- device-side scene bake + LBVH build for gfx950.
//
// Replaces the driver-side acceleration-structure build of the reference
// (Renderer::getOrCreateBLAS S/renderer/Renderer.cpp:122-131, createTLAS
// S/renderer/PathTracer.cpp:10-19, vkCmdBuildAccelerationStructuresKHR
// S/backend/vulkan/VulkanRays.cpp:81-85,176-180).  The two-level BLAS/TLAS is
// flattened: every instance's triangles are transformed to world space
// (gl_ObjectToWorldEXT * vec4(pos,1), rayhit.rchit:679-681) and one BVH is built
// over all of them.
//
// Passes (all on one stream, one thread per triangle / node):
//   bake      world-space triangle packets, shading packets (geometric normal,
//             transformInvT * vertex normals), padded boxes, scene bounds
//   morton    63-bit Morton code of the box centre
//   sort      rocPRIM radix sort of (code, triangle) pairs
//   scatter   packets into leaf (Morton) order -> coherent rays touch adjacent HBM lines
//   hierarchy binary tree over the Morton-sorted leaves: PLOC (parallel locally-ordered clustering,
//             Meister & Bittner 2018: repeatedly merge mutual nearest neighbours, by merged surface
//             area, within a window of the cluster array) -- near-SAH quality; the Karras 2012 radix
//             tree (LBVH) is kept behind GSP_BVH=lbvh for comparison
//   fit       bottom-up boxes of the binary tree (each binary node holds both child boxes)
//   collapse  binary tree -> 4-wide BVH (compressed 64-B nodes, pt_trace.h): every binary node at even depth becomes one
//             node whose children are its grandchildren (leaf children stay), so a ray makes half
//             as many dependent node fetches; node index = exclusive scan of the even-depth flags
//             (rocPRIM), which keeps the locality of the binary numbering.  GSP_COLLAPSE=greedy
//             selects a breadth-first greedy surface-area collapse (always 4 children where
//             possible): measured within +-3 % of the parity collapse, so it is not the default.
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "pt_internal.h"

namespace gsp {

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ uint32_t float_to_ordered(float f) {
  uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ __forceinline__ float ordered_to_float(uint32_t k) {
  uint32_t b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
#if defined(__HIP_DEVICE_COMPILE__)
  return __uint_as_float(b);
#else
  float f;
  memcpy(&f, &b, 4);
  return f;
#endif
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

__device__ __forceinline__ q4 mkq(float x, float y, float z, float w) {
  q4 r;
  r.x = x;
  r.y = y;
  r.z = z;
  r.w = w;
  return r;
}

// scene_bounds[0..2] = ordered min, [3..5] = ordered max
__global__ __launch_bounds__(kBlock) void k_bake(BuildInput in, q4* __restrict__ isect, q4* __restrict__ shade,
                                                  q4* __restrict__ box_lo, q4* __restrict__ box_hi,
                                                  uint32_t* __restrict__ scene_bounds) {
  const uint32_t g = blockIdx.x * kBlock + threadIdx.x;
  float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  if (g < in.num_tris) {
    // instance that owns global triangle g: last i with tri_first[i] <= g
    uint32_t a = 0, b = in.num_instances;
    while (b - a > 1) {
      uint32_t m = (a + b) >> 1;
      if (in.tri_first[m] <= g) a = m; else b = m;
    }
    const gsp_instance& I = in.instances[a];
    const uint32_t v = I.first_vertex + 3u * (g - in.tri_first[a]);
    const float* P = in.positions + 3ull * v;
    const float* Nn = in.normals + 3ull * v;
    const f3 p0 = xform_point(I.transform, mk3(P[0], P[1], P[2]));
    const f3 p1 = xform_point(I.transform, mk3(P[3], P[4], P[5]));
    const f3 p2 = xform_point(I.transform, mk3(P[6], P[7], P[8]));
    const float* T = in.inv_t + 16ull * a;
    const f3 n0 = xform_dir(T, mk3(Nn[0], Nn[1], Nn[2]));
    const f3 n1 = xform_dir(T, mk3(Nn[3], Nn[4], Nn[5]));
    const f3 n2 = xform_dir(T, mk3(Nn[6], Nn[7], Nn[8]));
    const f3 e1 = p1 - p0, e2 = p2 - p0;
    const f3 N = normalize(cross(e1, e2));  // rayhit.rchit:694
    isect[3ull * g + 0] = mkq(p0.x, p0.y, p0.z, __uint_as_float(g));
    isect[3ull * g + 1] = mkq(p1.x, p1.y, p1.z, __uint_as_float(I.bsdf >> 16));  // BSDF type of the hit, for the shade sort
    isect[3ull * g + 2] = mkq(p2.x, p2.y, p2.z, 0.0f);
    shade[4ull * g + 0] = mkq(N.x, N.y, N.z, __uint_as_float(pack_material(I.bsdf, I.twofaced)));
    shade[4ull * g + 1] = mkq(n0.x, n0.y, n0.z, I.emission[0]);
    shade[4ull * g + 2] = mkq(n1.x, n1.y, n1.z, I.emission[1]);
    shade[4ull * g + 3] = mkq(n2.x, n2.y, n2.z, I.emission[2]);
    const float px[3] = {p0.x, p0.y, p0.z}, qx[3] = {p1.x, p1.y, p1.z}, rx[3] = {p2.x, p2.y, p2.z};
    float l[3], h[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      l[k] = fminf(px[k], fminf(qx[k], rx[k]));
      h[k] = fmaxf(px[k], fmaxf(qx[k], rx[k]));
    }
    // conservative padding: box culling must never reject a triangle the triangle test would
    // accept.  It scales with the triangle's own extent as well as with its coordinates, so an
    // axis-aligned (flat) box at coordinate 0 is still padded.
    const float diag = fmaxf(h[0] - l[0], fmaxf(h[1] - l[1], h[2] - l[2]));
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float pad = 1e-5f * fmaxf(fmaxf(fabsf(l[k]), fabsf(h[k])), fmaxf(diag, 1e-3f));
      lo[k] = l[k] - pad;
      hi[k] = h[k] + pad;
    }
    box_lo[g] = mkq(lo[0], lo[1], lo[2], 0.0f);
    box_hi[g] = mkq(hi[0], hi[1], hi[2], 0.0f);
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float l = wave_min(lo[k]);
    float h = wave_max(hi[k]);
    if ((threadIdx.x & 63) == 0 && l <= h) {
      atomicMin(&scene_bounds[k], float_to_ordered(l));
      atomicMax(&scene_bounds[3 + k], float_to_ordered(h));
    }
  }
}

__device__ __forceinline__ uint64_t expand21(uint64_t v) {
  v &= 0x1fffffull;
  v = (v | (v << 32)) & 0x001f00000000ffffull;
  v = (v | (v << 16)) & 0x001f0000ff0000ffull;
  v = (v | (v << 8)) & 0x100f00f00f00f00full;
  v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}

__global__ __launch_bounds__(kBlock) void k_morton(uint32_t n, const q4* __restrict__ box_lo,
                                                   const q4* __restrict__ box_hi,
                                                   const uint32_t* __restrict__ scene_bounds,
                                                   uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const uint32_t g = blockIdx.x * kBlock + threadIdx.x;
  if (g >= n) return;
  const float smin[3] = {ordered_to_float(scene_bounds[0]), ordered_to_float(scene_bounds[1]),
                         ordered_to_float(scene_bounds[2])};
  const float smax[3] = {ordered_to_float(scene_bounds[3]), ordered_to_float(scene_bounds[4]),
                         ordered_to_float(scene_bounds[5])};
  const q4 l = box_lo[g], h = box_hi[g];
  const float c[3] = {0.5f * (l.x + h.x), 0.5f * (l.y + h.y), 0.5f * (l.z + h.z)};
  uint64_t code = 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float ext = smax[k] - smin[k];
    float t = ext > 0.0f ? (c[k] - smin[k]) / ext : 0.0f;
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    uint64_t q = (uint64_t)fminf(t * 2097152.0f, 2097151.0f);
    code |= expand21(q) << (2 - k);
  }
  keys[g] = code;
  vals[g] = g;
}

__global__ __launch_bounds__(kBlock) void k_scatter(uint32_t n, const uint32_t* __restrict__ sorted_vals,
                                                    const q4* __restrict__ isect_in, const q4* __restrict__ shade_in,
                                                    const q4* __restrict__ lo_in, const q4* __restrict__ hi_in,
                                                    q4* __restrict__ isect, q4* __restrict__ shade,
                                                    q4* __restrict__ leaf_lo, q4* __restrict__ leaf_hi,
                                                    uint32_t* __restrict__ slot_to_global) {
  const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
  if (s >= n) return;
  const uint32_t g = sorted_vals[s];
#pragma unroll
  for (int k = 0; k < 3; ++k) isect[3ull * s + k] = isect_in[3ull * g + k];
#pragma unroll
  for (int k = 0; k < 4; ++k) shade[4ull * s + k] = shade_in[4ull * g + k];
  leaf_lo[s] = lo_in[g];
  leaf_hi[s] = hi_in[g];
  slot_to_global[s] = g;
}

// common-prefix length of sorted keys i and j (index breaks ties), -1 outside the array
__device__ __forceinline__ int delta(const uint64_t* __restrict__ keys, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  const uint64_t a = keys[i], b = keys[j];
  if (a == b) return 64 + __clz((unsigned)(i ^ j));
  return __clzll((long long)(a ^ b));
}

// Karras 2012, "Maximizing parallelism in the construction of BVHs, octrees and k-d trees"
__global__ __launch_bounds__(kBlock) void k_hierarchy(int n, const uint64_t* __restrict__ keys,
                                                      int32_t* __restrict__ child_l, int32_t* __restrict__ child_r,
                                                      int32_t* __restrict__ parent_int,
                                                      int32_t* __restrict__ parent_leaf) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n - 1) return;
  const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
  const int dmin = delta(keys, n, i, i - d);
  int lmax = 2;
  while (delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
  int l = 0;
  for (int t = lmax >> 1; t >= 1; t >>= 1)
    if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = delta(keys, n, i, j);
  int s = 0;
  int t = l;
  do {
    t = (t + 1) >> 1;
    if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
  } while (t > 1);
  const int gamma = i + s * d + min(d, 0);
  const int lo = min(i, j), hi = max(i, j);
  if (lo == gamma) {
    child_l[i] = make_leaf((uint32_t)gamma, 1);
    parent_leaf[gamma] = i;
  } else {
    child_l[i] = gamma;
    parent_int[gamma] = i;
  }
  if (hi == gamma + 1) {
    child_r[i] = make_leaf((uint32_t)(gamma + 1), 1);
    parent_leaf[gamma + 1] = i;
  } else {
    child_r[i] = gamma + 1;
    parent_int[gamma + 1] = i;
  }
  if (i == 0) parent_int[0] = -1;
}

__device__ __forceinline__ void child_box(int32_t code, const q4* __restrict__ leaf_lo, const q4* __restrict__ leaf_hi,
                                          const q4* int_lo, const q4* int_hi, q4& lo, q4& hi) {
  if (code < 0) {
    const uint32_t slot = ((uint32_t)~code) >> 2;
    lo = leaf_lo[slot];
    hi = leaf_hi[slot];
  } else {
    lo = int_lo[code];
    hi = int_hi[code];
  }
}

// Bottom-up fit: the second thread to reach a node (arrival counter) owns it.
// Boxes written by the sibling subtree come from another CU / XCD, hence the
// agent-scope fences on both sides of the counter (MI355X: per-CU L1 and
// per-XCD L2 are not coherent).
__global__ __launch_bounds__(kBlock) void k_fit(int n, const int32_t* __restrict__ child_l,
                                                const int32_t* __restrict__ child_r,
                                                const int32_t* __restrict__ parent_int,
                                                const int32_t* __restrict__ parent_leaf,
                                                const q4* __restrict__ leaf_lo, const q4* __restrict__ leaf_hi,
                                                q4* int_lo, q4* int_hi, uint32_t* arrive, q4* __restrict__ nodes,
                                                uint32_t* __restrict__ max_depth) {
  const int s = blockIdx.x * kBlock + threadIdx.x;
  if (s >= n) return;
  int node = parent_leaf[s];
  uint32_t depth = 0;
  while (node >= 0) {
    ++depth;
    __threadfence();
    const uint32_t old = atomicAdd(&arrive[node], 1u);
    if (old == 0) break;  // sibling subtree not finished: it will continue from here
    __threadfence();
    const int32_t cl = child_l[node], cr = child_r[node];
    q4 llo, lhi, rlo, rhi;
    child_box(cl, leaf_lo, leaf_hi, int_lo, int_hi, llo, lhi);
    child_box(cr, leaf_lo, leaf_hi, int_lo, int_hi, rlo, rhi);
    q4* N = nodes + 4ll * node;
    N[0] = mkq(llo.x, llo.y, llo.z, lhi.x);
    N[1] = mkq(lhi.y, lhi.z, rlo.x, rlo.y);
    N[2] = mkq(rlo.z, rhi.x, rhi.y, rhi.z);
    N[3] = mkq(__uint_as_float((uint32_t)cl), __uint_as_float((uint32_t)cr), 0.0f, 0.0f);
    int_lo[node] = mkq(fminf(llo.x, rlo.x), fminf(llo.y, rlo.y), fminf(llo.z, rlo.z), 0.0f);
    int_hi[node] = mkq(fmaxf(lhi.x, rhi.x), fmaxf(lhi.y, rhi.y), fmaxf(lhi.z, rhi.z), 0.0f);
    node = parent_int[node];
  }
  // a thread that walked to the root saw every level of its own path
  if (node < 0) atomicMax(max_depth, depth);
}

// depth of the deepest leaf (number of internal nodes above it)
__global__ __launch_bounds__(kBlock) void k_depth(int n, const int32_t* __restrict__ parent_int,
                                                  const int32_t* __restrict__ parent_leaf,
                                                  uint32_t* __restrict__ max_depth) {
  const int s = blockIdx.x * kBlock + threadIdx.x;
  uint32_t depth = 0;
  if (s < n) {
    int node = parent_leaf[s];
    while (node >= 0) {
      ++depth;
      node = parent_int[node];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) depth = max(depth, (uint32_t)__shfl_xor((int)depth, o));
  if ((threadIdx.x & 63) == 0 && depth) atomicMax(max_depth, depth);
}

// ---- PLOC ------------------------------------------------------------------------------------
#ifndef GSP_PLOC_RADIUS
#define GSP_PLOC_RADIUS 32
#endif
constexpr int kPlocRadius = GSP_PLOC_RADIUS;

__device__ __forceinline__ float half_area(const q4& lo, const q4& hi) {
  const float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
  return dx * dy + dy * dz + dz * dx;
}

__global__ __launch_bounds__(kBlock) void k_ploc_init(int n, int32_t* __restrict__ code) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) code[i] = make_leaf((uint32_t)i, 1);
}

// nearest neighbour of every cluster inside the window [i - R, i + R]: smallest merged half-area
__global__ __launch_bounds__(kBlock) void k_ploc_nn(int n, const q4* __restrict__ lo, const q4* __restrict__ hi,
                                                    int32_t* __restrict__ nn) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const q4 a_lo = lo[i], a_hi = hi[i];
  float best = 3.0e38f;
  int best_j = i > 0 ? i - 1 : i + 1;
  const int j0 = max(0, i - kPlocRadius), j1 = min(n - 1, i + kPlocRadius);
  for (int j = j0; j <= j1; ++j) {
    if (j == i) continue;
    const q4 b_lo = lo[j], b_hi = hi[j];
    const q4 m_lo = mkq(fminf(a_lo.x, b_lo.x), fminf(a_lo.y, b_lo.y), fminf(a_lo.z, b_lo.z), 0.0f);
    const q4 m_hi = mkq(fmaxf(a_hi.x, b_hi.x), fmaxf(a_hi.y, b_hi.y), fmaxf(a_hi.z, b_hi.z), 0.0f);
    const float a = half_area(m_lo, m_hi);
    if (a < best) {
      best = a;
      best_j = j;
    }
  }
  nn[i] = best_j;
}

// keep[i] = 0 for the right partner of a mutual pair (it disappears), isnew[i] = 1 for the left one
__global__ __launch_bounds__(kBlock) void k_ploc_flags(int n, const int32_t* __restrict__ nn,
                                                       uint32_t* __restrict__ keep, uint32_t* __restrict__ isnew) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int j = nn[i];
  const bool mutual = nn[j] == i;
  keep[i] = (mutual && i > j) ? 0u : 1u;
  isnew[i] = (mutual && i < j) ? 1u : 0u;
}

__global__ __launch_bounds__(kBlock) void k_ploc_apply(int n, const int32_t* __restrict__ nn,
                                                       const uint32_t* __restrict__ keep,
                                                       const uint32_t* __restrict__ isnew,
                                                       const uint32_t* __restrict__ kpos,
                                                       const uint32_t* __restrict__ npos, uint32_t node_base,
                                                       const int32_t* __restrict__ code_in, const q4* __restrict__ lo_in,
                                                       const q4* __restrict__ hi_in, int32_t* __restrict__ code_out,
                                                       q4* __restrict__ lo_out, q4* __restrict__ hi_out,
                                                       q4* __restrict__ nodes2, int32_t* __restrict__ parent_int,
                                                       int32_t* __restrict__ parent_leaf) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n || !keep[i]) return;
  const uint32_t p = kpos[i];
  if (!isnew[i]) {
    code_out[p] = code_in[i];
    lo_out[p] = lo_in[i];
    hi_out[p] = hi_in[i];
    return;
  }
  const int j = nn[i];
  const int32_t id = (int32_t)(node_base + npos[i]);
  const int32_t cl = code_in[i], cr = code_in[j];
  const q4 llo = lo_in[i], lhi = hi_in[i], rlo = lo_in[j], rhi = hi_in[j];
  q4* N = nodes2 + 4ll * id;
  N[0] = mkq(llo.x, llo.y, llo.z, lhi.x);
  N[1] = mkq(lhi.y, lhi.z, rlo.x, rlo.y);
  N[2] = mkq(rlo.z, rhi.x, rhi.y, rhi.z);
  N[3] = mkq(__uint_as_float((uint32_t)cl), __uint_as_float((uint32_t)cr), 0.0f, 0.0f);
  if (cl < 0) parent_leaf[((uint32_t)~cl) >> 2] = id; else parent_int[cl] = id;
  if (cr < 0) parent_leaf[((uint32_t)~cr) >> 2] = id; else parent_int[cr] = id;
  code_out[p] = id;
  lo_out[p] = mkq(fminf(llo.x, rlo.x), fminf(llo.y, rlo.y), fminf(llo.z, rlo.z), 0.0f);
  hi_out[p] = mkq(fmaxf(lhi.x, rhi.x), fmaxf(lhi.y, rhi.y), fmaxf(lhi.z, rhi.z), 0.0f);
}

// flag[i] = 1 when binary node i sits at even depth (root = depth 0)
__global__ __launch_bounds__(kBlock) void k_flag_even(int n_int, const int32_t* __restrict__ parent_int,
                                                      uint32_t* __restrict__ flag) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n_int) return;
  uint32_t depth = 0;
  for (int node = parent_int[i]; node >= 0; node = parent_int[node]) ++depth;
  flag[i] = (depth & 1u) ? 0u : 1u;
}

// parity collapse: one thread per even-depth binary node
__global__ __launch_bounds__(kBlock) void k_emit4(int n_int, const q4* __restrict__ nodes2,
                                                  const uint32_t* __restrict__ flag,
                                                  const uint32_t* __restrict__ idx4, q4* __restrict__ nodes4,
                                                  uint32_t dummy_slot) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n_int || !flag[i]) return;
  Entry4 e[4];
  int cnt = 0;
  auto conv = [&](int32_t g) { return g < 0 ? g : (int32_t)((idx4[g] + kTopNodes) * 64u); };  // inner child = byte offset of its node
  auto expand = [&](int32_t c, q4 lo, q4 hi) {
    if (c < 0) {
      e[cnt].lo = lo;
      e[cnt].hi = hi;
      e[cnt].code = c;
      ++cnt;
    } else {  // odd-depth inner node: absorb it, adopt its two children
      const q4* m = nodes2 + 4ll * c;
      const q4 a = m[0], b = m[1], d = m[2], k = m[3];
      e[cnt].lo = mkq(a.x, a.y, a.z, 0.0f);
      e[cnt].hi = mkq(a.w, b.x, b.y, 0.0f);
      e[cnt].code = conv((int32_t)__float_as_uint(k.x));
      ++cnt;
      e[cnt].lo = mkq(b.z, b.w, d.x, 0.0f);
      e[cnt].hi = mkq(d.y, d.z, d.w, 0.0f);
      e[cnt].code = conv((int32_t)__float_as_uint(k.y));
      ++cnt;
    }
  };
  const q4* me = nodes2 + 4ll * i;
  const q4 a = me[0], b = me[1], d = me[2], k = me[3];
  expand((int32_t)__float_as_uint(k.x), mkq(a.x, a.y, a.z, 0.0f), mkq(a.w, b.x, b.y, 0.0f));
  expand((int32_t)__float_as_uint(k.y), mkq(b.z, b.w, d.x, 0.0f), mkq(d.y, d.z, d.w, 0.0f));
  encode_node4(nodes4 + 4ll * (idx4[i] + kTopNodes), e, cnt, dummy_slot);
}

// Greedy 4-wide collapse, one thread per output node of the current level.
// work item = {binary node id, 4-wide node id}
__global__ __launch_bounds__(kBlock) void k_collapse4(int count, const int2* __restrict__ qin,
                                                      const q4* __restrict__ nodes2, q4* __restrict__ nodes4,
                                                      uint32_t* __restrict__ next_id, int2* __restrict__ qout,
                                                      uint32_t* __restrict__ qout_count, uint32_t dummy_slot) {
  const int t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= count) return;
  const int2 w = qin[t];
  Entry4 e[4];
  int cnt = 0;
  auto load2 = [&](int32_t b, Entry4& x, Entry4& y) {
    const q4* m = nodes2 + 4ll * b;
    const q4 a = m[0], bq = m[1], d = m[2], k = m[3];
    x.lo = mkq(a.x, a.y, a.z, 0.0f);
    x.hi = mkq(a.w, bq.x, bq.y, 0.0f);
    x.code = (int32_t)__float_as_uint(k.x);
    y.lo = mkq(bq.z, bq.w, d.x, 0.0f);
    y.hi = mkq(d.y, d.z, d.w, 0.0f);
    y.code = (int32_t)__float_as_uint(k.y);
  };
  load2(w.x, e[0], e[1]);
  cnt = 2;
  while (cnt < 4) {
    int best = -1;
    float best_area = -1.0f;
    for (int k = 0; k < cnt; ++k) {
      if (e[k].code < 0) continue;
      const float ar = half_area(e[k].lo, e[k].hi);
      if (ar > best_area) {
        best_area = ar;
        best = k;
      }
    }
    if (best < 0) break;
    Entry4 x, y;
    load2(e[best].code, x, y);
    e[best] = x;
    e[cnt++] = y;
  }
  // inner children become work items of the next level; their 4-wide ids are consecutive
  int inner = 0;
  for (int k = 0; k < cnt; ++k) inner += e[k].code >= 0 ? 1 : 0;
  uint32_t id0 = 0, q0 = 0;
  if (inner) {
    id0 = atomicAdd(next_id, (uint32_t)inner);
    q0 = atomicAdd(qout_count, (uint32_t)inner);
  }
  for (int k = 0; k < cnt; ++k) {
    if (e[k].code >= 0) {
      qout[q0++] = make_int2(e[k].code, (int)id0);
      e[k].code = (int32_t)(id0++ * 64u);
    }
  }
  encode_node4(nodes4 + 4ll * w.y, e, cnt, dummy_slot);
}

struct Scratch {
  std::vector<void*> ptrs;
  size_t bytes = 0;
  ~Scratch() {
    for (void* p : ptrs) (void)hipFree(p);
  }
  template <class T>
  hipError_t alloc(T** p, size_t count) {
    size_t b = std::max<size_t>(count, 1) * sizeof(T);
    hipError_t e = hipMalloc((void**)p, b);
    if (e == hipSuccess) {
      ptrs.push_back(*p);
      bytes += b;
    }
    return e;
  }
};

inline uint32_t blocks_for(uint64_t n) { return (uint32_t)((n + kBlock - 1) / kBlock); }

}  // namespace

void free_bvh(DeviceBvh& b) {
  (void)hipFree(b.nodes);
  (void)hipFree(b.tri_isect);
  (void)hipFree(b.tri_shade);
  (void)hipFree(b.slot_to_global);
  b = DeviceBvh{};
}

int build_bvh(hipStream_t stream, const BuildInput& in, DeviceBvh& out, std::string& err) {
  free_bvh(out);
  const uint32_t n = in.num_tris;
  const uint32_t slots = n + 1;  // slot n is a degenerate all-zero triangle (det == 0: never hit): the target of
                                  // empty child slots and the leaf of an empty scene
  out.num_tris = n;
  out.num_nodes = 0;
  size_t b_is = (size_t)slots * 48, b_sh = (size_t)slots * 64, b_map = (size_t)slots * 4;
  GSP_HIP_TRY(hipMalloc((void**)&out.tri_isect, b_is));
  GSP_HIP_TRY(hipMalloc((void**)&out.tri_shade, b_sh));
  GSP_HIP_TRY(hipMalloc((void**)&out.slot_to_global, b_map));
  out.bytes = b_is + b_sh + b_map;
  GSP_HIP_TRY(hipMemsetAsync(out.tri_isect, 0, b_is, stream));
  GSP_HIP_TRY(hipMemsetAsync(out.tri_shade, 0, b_sh, stream));
  GSP_HIP_TRY(hipMemsetAsync(out.slot_to_global, 0, b_map, stream));
  out.root = make_leaf(0, 1);  // n == 0: slot 0 is the degenerate triangle
  out.depth = 0;
  if (n < 2) {  // no inner node: the root is the single (or dummy) triangle's leaf
    GSP_HIP_TRY(hipMalloc((void**)&out.nodes, 128));
    GSP_HIP_TRY(hipMemsetAsync(out.nodes, 0, 128, stream));
    out.bytes += 128;
  }
  if (n == 0) {
    GSP_HIP_TRY(hipStreamSynchronize(stream));
    return GSP_OK;
  }

  Scratch S;
  q4 *isect_g, *shade_g, *lo_g, *hi_g, *leaf_lo, *leaf_hi, *int_lo, *int_hi, *nodes2;
  uint32_t *bounds, *vals_in, *vals_out, *arrive, *d_depth, *flag, *idx4;
  uint64_t *keys_in, *keys_out;
  int32_t *child_l, *child_r, *parent_int, *parent_leaf;
  GSP_HIP_TRY(S.alloc(&isect_g, 3ull * n));
  GSP_HIP_TRY(S.alloc(&shade_g, 4ull * n));
  GSP_HIP_TRY(S.alloc(&lo_g, n));
  GSP_HIP_TRY(S.alloc(&hi_g, n));
  GSP_HIP_TRY(S.alloc(&leaf_lo, n));
  GSP_HIP_TRY(S.alloc(&leaf_hi, n));
  GSP_HIP_TRY(S.alloc(&int_lo, n));
  GSP_HIP_TRY(S.alloc(&int_hi, n));
  GSP_HIP_TRY(S.alloc(&bounds, 8));
  GSP_HIP_TRY(S.alloc(&keys_in, n));
  GSP_HIP_TRY(S.alloc(&keys_out, n));
  GSP_HIP_TRY(S.alloc(&vals_in, n));
  GSP_HIP_TRY(S.alloc(&vals_out, n));
  GSP_HIP_TRY(S.alloc(&arrive, n));
  GSP_HIP_TRY(S.alloc(&d_depth, 1));
  GSP_HIP_TRY(S.alloc(&child_l, n));
  GSP_HIP_TRY(S.alloc(&child_r, n));
  GSP_HIP_TRY(S.alloc(&parent_int, n));
  GSP_HIP_TRY(S.alloc(&parent_leaf, n));
  GSP_HIP_TRY(S.alloc(&nodes2, 4ull * n));
  GSP_HIP_TRY(S.alloc(&flag, n + 1ull));
  GSP_HIP_TRY(S.alloc(&idx4, n + 1ull));

  uint32_t collapse_levels = 0;
  const uint32_t init_bounds[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u};
  GSP_HIP_TRY(hipMemcpyAsync(bounds, init_bounds, sizeof(init_bounds), hipMemcpyHostToDevice, stream));
  GSP_HIP_TRY(hipMemsetAsync(arrive, 0, sizeof(uint32_t) * n, stream));
  GSP_HIP_TRY(hipMemsetAsync(d_depth, 0, sizeof(uint32_t), stream));

  hipLaunchKernelGGL(k_bake, dim3(blocks_for(n)), dim3(kBlock), 0, stream, in, isect_g, shade_g, lo_g, hi_g, bounds);
  hipLaunchKernelGGL(k_morton, dim3(blocks_for(n)), dim3(kBlock), 0, stream, n, lo_g, hi_g, bounds, keys_in, vals_in);
  GSP_HIP_TRY(hipGetLastError());

  size_t temp_bytes = 0;
  GSP_HIP_TRY(rocprim::radix_sort_pairs(nullptr, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0, 63, stream));
  void* temp = nullptr;
  GSP_HIP_TRY(S.alloc((char**)&temp, temp_bytes));
  GSP_HIP_TRY(rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0, 63, stream));

  hipLaunchKernelGGL(k_scatter, dim3(blocks_for(n)), dim3(kBlock), 0, stream, n, vals_out, isect_g, shade_g, lo_g, hi_g,
                     out.tri_isect, out.tri_shade, leaf_lo, leaf_hi, out.slot_to_global);
  if (n >= 2) {
    int32_t root2 = 0;  // binary root
    const char* mode = getenv("GSP_BVH");
    if (mode && std::string(mode) == "lbvh") {
      hipLaunchKernelGGL(k_hierarchy, dim3(blocks_for(n - 1)), dim3(kBlock), 0, stream, (int)n, keys_out, child_l, child_r,
                         parent_int, parent_leaf);
      hipLaunchKernelGGL(k_fit, dim3(blocks_for(n)), dim3(kBlock), 0, stream, (int)n, child_l, child_r, parent_int,
                         parent_leaf, leaf_lo, leaf_hi, int_lo, int_hi, arrive, nodes2, d_depth);
    } else {
      // PLOC: cluster arrays ping-pong between (code_a, leaf_lo/hi) and (code_b, int_lo/hi)
      int32_t *code_a = child_l, *code_b = child_r, *nn = (int32_t*)arrive;
      uint32_t *keep = flag, *isnew = idx4, *kpos, *npos;
      GSP_HIP_TRY(S.alloc(&kpos, n + 1ull));
      GSP_HIP_TRY(S.alloc(&npos, n + 1ull));
      size_t sb = 0;
      GSP_HIP_TRY(rocprim::exclusive_scan(nullptr, sb, keep, kpos, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), stream));
      void* stmp = nullptr;
      GSP_HIP_TRY(S.alloc((char**)&stmp, sb));
      GSP_HIP_TRY(hipMemsetAsync(parent_int, 0xff, sizeof(int32_t) * n, stream));
      GSP_HIP_TRY(hipMemsetAsync(parent_leaf, 0xff, sizeof(int32_t) * n, stream));
      hipLaunchKernelGGL(k_ploc_init, dim3(blocks_for(n)), dim3(kBlock), 0, stream, (int)n, code_a);
      q4 *lo_a = leaf_lo, *hi_a = leaf_hi, *lo_b = int_lo, *hi_b = int_hi;
      uint32_t m = n, node_base = 0;
      while (m > 1) {
        hipLaunchKernelGGL(k_ploc_nn, dim3(blocks_for(m)), dim3(kBlock), 0, stream, (int)m, lo_a, hi_a, nn);
        hipLaunchKernelGGL(k_ploc_flags, dim3(blocks_for(m)), dim3(kBlock), 0, stream, (int)m, nn, keep, isnew);
        GSP_HIP_TRY(hipMemsetAsync(keep + m, 0, sizeof(uint32_t), stream));
        GSP_HIP_TRY(hipMemsetAsync(isnew + m, 0, sizeof(uint32_t), stream));
        GSP_HIP_TRY(rocprim::exclusive_scan(stmp, sb, keep, kpos, 0u, (size_t)m + 1, rocprim::plus<uint32_t>(), stream));
        GSP_HIP_TRY(rocprim::exclusive_scan(stmp, sb, isnew, npos, 0u, (size_t)m + 1, rocprim::plus<uint32_t>(), stream));
        hipLaunchKernelGGL(k_ploc_apply, dim3(blocks_for(m)), dim3(kBlock), 0, stream, (int)m, nn, keep, isnew, kpos, npos,
                           node_base, code_a, lo_a, hi_a, code_b, lo_b, hi_b, nodes2, parent_int, parent_leaf);
        uint32_t counts[2] = {0, 0};
        GSP_HIP_TRY(hipMemcpyAsync(&counts[0], kpos + m, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        GSP_HIP_TRY(hipMemcpyAsync(&counts[1], npos + m, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        GSP_HIP_TRY(hipStreamSynchronize(stream));
        if (counts[1] == 0 || counts[0] != m - counts[1]) {
          err = "PLOC made no progress (internal error)";
          return GSP_ERR_DEVICE;
        }
        node_base += counts[1];
        m = counts[0];
        std::swap(code_a, code_b);
        std::swap(lo_a, lo_b);
        std::swap(hi_a, hi_b);
      }
      GSP_HIP_TRY(hipMemcpyAsync(&root2, code_a, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
      GSP_HIP_TRY(hipStreamSynchronize(stream));
      // leaf_lo/leaf_hi may have been overwritten by the ping-pong: nothing below reads them again
    }
    hipLaunchKernelGGL(k_depth, dim3(blocks_for(n)), dim3(kBlock), 0, stream, (int)n, parent_int, parent_leaf, d_depth);
    // ---- collapse to the 4-wide tree ----
    const int n_int = (int)n - 1;
    const char* cmode = getenv("GSP_COLLAPSE");
    if (!(cmode && std::string(cmode) == "greedy")) {
      GSP_HIP_TRY(hipMemsetAsync(flag, 0, sizeof(uint32_t) * (n + 1ull), stream));
      hipLaunchKernelGGL(k_flag_even, dim3(blocks_for(n_int)), dim3(kBlock), 0, stream, n_int, parent_int, flag);
      size_t scan_bytes = 0;
      GSP_HIP_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, flag, idx4, 0u, (size_t)n_int + 1, rocprim::plus<uint32_t>(), stream));
      void* scan_tmp = nullptr;
      GSP_HIP_TRY(S.alloc((char**)&scan_tmp, scan_bytes));
      GSP_HIP_TRY(rocprim::exclusive_scan(scan_tmp, scan_bytes, flag, idx4, 0u, (size_t)n_int + 1, rocprim::plus<uint32_t>(), stream));
      uint32_t n4 = 0;
      GSP_HIP_TRY(hipMemcpyAsync(&n4, idx4 + n_int, sizeof(n4), hipMemcpyDeviceToHost, stream));
      GSP_HIP_TRY(hipStreamSynchronize(stream));
      out.num_nodes = n4;
      if (((uint64_t)n4 + kTopNodes) * 64ull >= 0x7fffff00ull) {
        err = "scene too large: BVH node offsets exceed 31 bits";
        return GSP_ERR_INVALID;
      }
      const size_t b_nodes = (size_t)(std::max<uint32_t>(n4, 1) + kTopNodes) * 64;
      GSP_HIP_TRY(hipMalloc((void**)&out.nodes, b_nodes));
      if (kTopNodes) GSP_HIP_TRY(hipMemsetAsync(out.nodes, 0, (size_t)kTopNodes * 64, stream));
      out.bytes += b_nodes;
      hipLaunchKernelGGL(k_emit4, dim3(blocks_for(n_int)), dim3(kBlock), 0, stream, n_int, nodes2, flag, idx4, out.nodes, n);
      uint32_t root4 = 0;  // the binary root has depth 0, so it owns a 4-wide node
      GSP_HIP_TRY(hipMemcpyAsync(&root4, idx4 + root2, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
      GSP_HIP_TRY(hipStreamSynchronize(stream));
      out.root = (int32_t)((root4 + kTopNodes) * 64u);
      collapse_levels = 0;
      if (kTopNodes) {
        // breadth-first copy of the top of the tree into slots [0, kTopNodes): child links that stay inside the copy
        // are rewritten, the others keep pointing at the original nodes (which stay in place)
        std::vector<q4> top(4ull * kTopNodes);
        std::vector<int32_t> orig(kTopNodes);
        orig[0] = out.root;
        uint32_t count = 1;
        for (uint32_t i = 0; i < count; ++i) {
          q4* nd = &top[4ull * i];
          GSP_HIP_TRY(hipMemcpyAsync(nd, (const char*)out.nodes + (uint32_t)orig[i], 64, hipMemcpyDeviceToHost, stream));
          GSP_HIP_TRY(hipStreamSynchronize(stream));
          float* link[4] = {&nd[2].z, &nd[2].w, &nd[3].x, &nd[3].y};
          for (int k = 0; k < 4; ++k) {
            const int32_t c = (int32_t)f2u(*link[k]);
            if (c >= 0 && count < kTopNodes) {
              orig[count] = c;
              *link[k] = u2f(count * 64u);
              ++count;
            }
          }
        }
        GSP_HIP_TRY(hipMemcpyAsync(out.nodes, top.data(), (size_t)count * 64, hipMemcpyHostToDevice, stream));
        GSP_HIP_TRY(hipStreamSynchronize(stream));
        out.root = 0;
      }
    } else {
      // greedy SAH collapse, breadth-first; at most n - 1 output nodes
      q4* all4 = nullptr;
      GSP_HIP_TRY(hipMalloc((void**)&all4, (size_t)n_int * 64));
      int2 *qa, *qb;
      uint32_t* ctr;  // [0] next node id, [1] next-level queue size
      GSP_HIP_TRY(S.alloc(&qa, (size_t)n_int));
      GSP_HIP_TRY(S.alloc(&qb, (size_t)n_int));
      GSP_HIP_TRY(S.alloc(&ctr, 2));
      const int2 first = make_int2(root2, 0);
      const uint32_t init[2] = {1u, 0u};
      GSP_HIP_TRY(hipMemcpyAsync(qa, &first, sizeof(first), hipMemcpyHostToDevice, stream));
      GSP_HIP_TRY(hipMemcpyAsync(ctr, init, sizeof(init), hipMemcpyHostToDevice, stream));
      uint32_t count = 1, total = 1;
      collapse_levels = 0;
      while (count > 0) {
        hipLaunchKernelGGL(k_collapse4, dim3(blocks_for(count)), dim3(kBlock), 0, stream, (int)count, qa, nodes2, all4,
                           ctr, qb, ctr + 1, n);
        uint32_t h[2];
        GSP_HIP_TRY(hipMemcpyAsync(h, ctr, sizeof(h), hipMemcpyDeviceToHost, stream));
        GSP_HIP_TRY(hipStreamSynchronize(stream));
        GSP_HIP_TRY(hipMemsetAsync(ctr + 1, 0, sizeof(uint32_t), stream));
        total = h[0];
        count = h[1];
        std::swap(qa, qb);
        ++collapse_levels;
      }
      out.nodes = all4;
      out.num_nodes = total;
      out.bytes += (size_t)n_int * 64;
      out.root = 0;
    }
  }
  GSP_HIP_TRY(hipGetLastError());
  uint32_t depth = 0;
  GSP_HIP_TRY(hipMemcpyAsync(&depth, d_depth, sizeof(depth), hipMemcpyDeviceToHost, stream));
  GSP_HIP_TRY(hipStreamSynchronize(stream));
  out.depth = collapse_levels ? collapse_levels : depth / 2 + 1;  // levels of the 4-wide tree
  return GSP_OK;
}

}  // namespace gsp
