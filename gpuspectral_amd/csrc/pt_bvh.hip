// pt_bvh.hip -- device-side scene bake + BVH build (PLOC, reinsertion, wide collapse) for gfx950.
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
//             area, within a window of the cluster array) -- near-SAH quality; a binary node's record holds both child
//             boxes + codes.  (The Karras 2012 radix tree it replaced in r01 is in the history: interior +7.5 %,
//             Cornell-materials +34 %, caustics +60 % Mrays/s for PLOC, LAB_NOTES.md)
//   reinsert  6 rounds of parallel reinsertion over the PLOC tree (k_ri_*, below): -8 % / -6 % node visits per ray;
//             the boxes of the result are VALIDATED on the device (k_ri_validate) before the tree is used
//   collapse  binary tree -> wide BVH with contiguous children (pt_trace.h), level by level: the children of a 4-wide
//             node are the grandchildren of its binary node ("parity" collapse; the greedy surface-area choice loses on
//             PLOC trees, 15.1 vs 14.7 nodes per extension ray, profiles/r03_collapse.txt, and is in the history);
//             inner children of a node = consecutive nodes, leaf children = consecutive triangle slots: the collapse
//             defines the final triangle order
//   refit     (refit_bvh, gsp_update_instances) the reference rebuilds only its TLAS when an object moves
//             (PathTracer.cpp:10-19); the flattened tree's counterpart: re-bake every packet into its slot, recompute all
//             node boxes + child orders bottom-up in the existing topology, one launch per level
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_reduce.hpp>
#include <rocprim/device/device_scan.hpp>

#include "pt_internal.h"

namespace gsp {

namespace {

constexpr int kBlock = 256;
constexpr uint32_t kFirstSlot = 4;  // first triangle slot in use (>= the largest ni of a 4-wide node)

// leaf of the BINARY tree (the build's intermediate form): ~((first slot << 2) | (count - 1)), count always 1
__host__ __device__ __forceinline__ int32_t make_leaf(uint32_t first_slot, uint32_t count) {
  return ~(int32_t)((first_slot << 2) | (count - 1u));
}

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
// remap == nullptr (build): thread t bakes global triangle t into record t.  Refit: thread t bakes global triangle
// remap[out_first + t] into record out_first + t -- the triangle slot it already has in the tree.
__global__ __launch_bounds__(kBlock) void k_bake(BuildInput in, q4* __restrict__ isect, q4* __restrict__ shade,
                                                  q4* __restrict__ box_lo, q4* __restrict__ box_hi,
                                                  uint32_t* __restrict__ scene_bounds, const uint32_t* __restrict__ remap,
                                                  uint32_t out_first) {
  const uint32_t t_ = blockIdx.x * kBlock + threadIdx.x;
  float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  if (t_ < in.num_tris) {
    const uint32_t g = remap ? remap[out_first + t_] : t_;
    const uint32_t o_ = remap ? out_first + t_ : t_;
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
    // p0.w: tie-break key of the closest-hit rule (orders like g) + the BSDF type of the hit for the shade sort
    const uint32_t gid = in.tri_id_first ? in.tri_id_first[a] + (g - in.tri_first[a]) : g;
    isect[3ull * o_ + 0] = mkq(p0.x, p0.y, p0.z, __uint_as_float((gid << 3) | ((I.bsdf >> 16) & 7u)));
    isect[3ull * o_ + 1] = mkq(p1.x, p1.y, p1.z, 0.0f);
    isect[3ull * o_ + 2] = mkq(p2.x, p2.y, p2.z, 0.0f);
    shade[4ull * o_ + 0] = mkq(N.x, N.y, N.z, __uint_as_float(pack_material(I.bsdf, I.twofaced)));
    shade[4ull * o_ + 1] = mkq(n0.x, n0.y, n0.z, I.emission[0]);
    shade[4ull * o_ + 2] = mkq(n1.x, n1.y, n1.z, I.emission[1]);
    shade[4ull * o_ + 3] = mkq(n2.x, n2.y, n2.z, I.emission[2]);
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
    // Slivers (r03): the float32 triangle test loses accuracy in its t in proportion to (longest edge)^2 / area -- a
    // 6.3 m x 2 mm bevel of the reference's living-room scene reports t 5e-4 short of where the ray meets it, 8 x the
    // plain pad, so whether its box was still entered after a nearer hit, and with it which of two triangles won,
    // depended on the visiting order (1 ray in 918 k).  The pad grows with that aspect ratio (x 1 up to aspect 32, x 1024
    // at most).
    const f3 e3 = p2 - p1;
    const float l2 = fmaxf(fmaxf(dot(e1, e1), dot(e2, e2)), dot(e3, e3));
    const f3 cr = cross(e1, e2);
    // (a triangle thinner than 1e-6 of its length is a line at the precision of its own coordinates: nothing to protect,
    // and 5 % of the reference's staircase2 scene is such triangles -- padding them cost it 9 % of its speed)
    const float aspect = l2 / fmaxf(gsqrt(dot(cr, cr)), 1e-30f);
    const float sliver = aspect < 1.0e6f ? fminf(fmaxf(aspect * (1.0f / 32.0f), 1.0f), 1024.0f) : 1.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float pad = 1e-5f * fmaxf(fmaxf(fabsf(l[k]), fabsf(h[k])), fmaxf(diag, 1e-3f)) * sliver;
      lo[k] = l[k] - pad;
      hi[k] = h[k] + pad;
    }
    box_lo[o_] = mkq(lo[0], lo[1], lo[2], 0.0f);
    box_hi[o_] = mkq(hi[0], hi[1], hi[2], 0.0f);
  }
  // scene bounds (the Morton grid of a build; a refit passes nullptr): reduced over the block first -- six atomics per WAVE on one
  // cache line are 94 k requests for a million triangles, and a line takes 88 M/s (profiles/r04_atomic_rate.txt): they were
  // 1.0 of this kernel's 1.06 ms
  if (!scene_bounds) return;  // (block-uniform)
  __shared__ float s_lo[kBlock / 64][3], s_hi[kBlock / 64][3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float l = wave_min(lo[k]);
    const float h = wave_max(hi[k]);
    if ((threadIdx.x & 63) == 0) {
      s_lo[threadIdx.x >> 6][k] = l;
      s_hi[threadIdx.x >> 6][k] = h;
    }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int k = threadIdx.x;
    float l = s_lo[0][k], h = s_hi[0][k];
    for (int w = 1; w < kBlock / 64; ++w) {
      l = fminf(l, s_lo[w][k]);
      h = fmaxf(h, s_hi[w][k]);
    }
    if (l <= h) {
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

// nearest neighbour of cluster i inside the window [i - R, i + R]: smallest merged half-area
__device__ __forceinline__ int ploc_nearest(int i, int n, const q4* lo, const q4* hi) {
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
  return best_j;
}
__global__ __launch_bounds__(kBlock) void k_ploc_nn(int n, const q4* __restrict__ lo, const q4* __restrict__ hi,
                                                    int32_t* __restrict__ nn) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  nn[i] = ploc_nearest(i, n, lo, hi);
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

// cluster i survives this round at position p of the next array: unchanged, or (isnew) merged with its partner j into
// binary node `id`
__device__ __forceinline__ void ploc_emit(int i, int j, bool isnew, uint32_t p, int32_t id, const int32_t* code_in, const q4* lo_in,
                                          const q4* hi_in, int32_t* code_out, q4* lo_out, q4* hi_out, q4* nodes2, int32_t* parent_int,
                                          int32_t* parent_leaf) {
  if (!isnew) {
    code_out[p] = code_in[i];
    lo_out[p] = lo_in[i];
    hi_out[p] = hi_in[i];
    return;
  }
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
  ploc_emit(i, nn[i], isnew[i] != 0u, kpos[i], (int32_t)(node_base + npos[i]), code_in, lo_in, hi_in, code_out, lo_out, hi_out, nodes2,
            parent_int, parent_leaf);
}

// The last rounds of PLOC: once few clusters are left a round is five launches, two scans and a read-back for microseconds of
// work, and it takes as many rounds to get from 2048 clusters to one as from a million to 2048.  ONE block runs them all:
// the same search, the same mutual-pair rule, the same positions as the kernels above (so the same tree), the scans in LDS.
constexpr int kPlocTail = 2048, kPlocTailThreads = 1024;
__global__ __launch_bounds__(kPlocTailThreads) void k_ploc_tail(int m, uint32_t node_base, int32_t* code_a, q4* lo_a, q4* hi_a,
                                                                int32_t* code_b, q4* lo_b, q4* hi_b, q4* nodes2, int32_t* parent_int,
                                                                int32_t* parent_leaf, int32_t* result) {
  __shared__ int32_t s_nn[kPlocTail];
  __shared__ uint16_t s_kpos[kPlocTail], s_npos[kPlocTail];  // position inside the thread's own run of clusters
  __shared__ uint32_t s_part[2][kPlocTailThreads];           // per-thread totals, then their exclusive scan
  __shared__ uint32_t s_tot[2];
  const int tid = threadIdx.x;
  while (m > 1) {
    for (int i = tid; i < m; i += kPlocTailThreads) s_nn[i] = ploc_nearest(i, m, lo_a, hi_a);
    __syncthreads();
    const int per = (m + kPlocTailThreads - 1) / kPlocTailThreads, b = min(m, tid * per), e = min(m, b + per);
    uint32_t ck = 0, cn = 0;
    for (int i = b; i < e; ++i) {
      const int j = s_nn[i];
      const bool mutual = s_nn[j] == i;
      s_kpos[i] = (uint16_t)ck;
      s_npos[i] = (uint16_t)cn;
      ck += (mutual && i > j) ? 0u : 1u;
      cn += (mutual && i < j) ? 1u : 0u;
    }
    s_part[0][tid] = ck;
    s_part[1][tid] = cn;
    __syncthreads();
    if (tid < 2) {
      uint32_t acc = 0;
      for (int t = 0; t < kPlocTailThreads; ++t) {
        const uint32_t c = s_part[tid][t];
        s_part[tid][t] = acc;
        acc += c;
      }
      s_tot[tid] = acc;
    }
    __syncthreads();
    for (int i = b; i < e; ++i) {
      const int j = s_nn[i];
      const bool mutual = s_nn[j] == i;
      if (mutual && i > j) continue;  // the right partner of a pair disappears
      ploc_emit(i, j, mutual && i < j, s_part[0][tid] + s_kpos[i], (int32_t)(node_base + s_part[1][tid] + s_npos[i]), code_a, lo_a, hi_a,
                code_b, lo_b, hi_b, nodes2, parent_int, parent_leaf);
    }
    __syncthreads();  // (the next round reads what this one wrote: same block, workgroup-scope fence)
    const uint32_t kept = s_tot[0], made = s_tot[1];
    __syncthreads();
    if (made == 0 || kept != (uint32_t)m - made) {  // cannot happen (the closest pair is mutual); do not spin on it
      if (tid == 0) result[0] = -1, result[1] = 0;
      return;
    }
    node_base += made;
    m = (int)kept;
    int32_t* tc = code_a; code_a = code_b; code_b = tc;
    q4* tl = lo_a; lo_a = lo_b; lo_b = tl;
    q4* th = hi_a; hi_a = hi_b; hi_b = th;
  }
  if (tid == 0) result[0] = code_a[0], result[1] = 1;
}

// a child of a binary node: box + code
struct Pick {
  q4 lo, hi;
  int32_t code;  // >= 0 binary inner node, < 0 leaf (make_leaf)
};
__device__ __forceinline__ void load2(const q4* __restrict__ nodes2, int32_t b, Pick& x, Pick& y) {
  const q4* m = nodes2 + 4ll * b;
  const q4 a = m[0], bq = m[1], d = m[2], k = m[3];
  x.lo = mkq(a.x, a.y, a.z, 0.0f);
  x.hi = mkq(a.w, bq.x, bq.y, 0.0f);
  x.code = (int32_t)__float_as_uint(k.x);
  y.lo = mkq(bq.z, bq.w, d.x, 0.0f);
  y.hi = mkq(d.y, d.z, d.w, 0.0f);
  y.code = (int32_t)__float_as_uint(k.y);
}

// ---- parallel reinsertion (r03) --------------------------------------------------------------------------------------
// Insertion-based optimisation of the PLOC tree, after Meister & Bittner, "Parallel Reinsertion for Bounding Volume
// Hierarchy Optimization" (2018).  One round: every node N (inner or leaf, neither the root nor one of its children)
// searches the FROZEN tree for the node X beside which it would sit best -- the search climbs from N's parent P, descends
// into the sibling subtrees of the path with branch and bound, and credits the shrinking of the path once N is gone; P is
// recycled as the new parent of {X, N}.  Candidates with a positive gain take locks (atomicMax of {gain, id}) on the nodes
// whose links they would rewrite -- N, P, the sibling S, the grandparent G, X and X's parent -- and the ones that hold all of
// theirs are applied; then every box is refitted bottom-up.  The CPU probe (scripts/experiments/wide_bvh_probe.cpp,
// PREINSERT=rounds: the same algorithm) says 6 rounds cut the node visits of the bench scene's rays by 10 % / 8 %
// (closest hit / any hit); searches are short (tens to a few hundred node visits), so a round costs about one trace launch.
// Node index space of the per-node arrays: inner node i -> i, leaf (Morton slot s) -> (n - 1) + s.
#ifndef GSP_REINSERT_ROUNDS
#define GSP_REINSERT_ROUNDS 6
#endif
constexpr int kReinsertRounds = GSP_REINSERT_ROUNDS;  // gsp_ctx_options.reinsert_rounds overrides (1 = none: the plain PLOC tree)
#ifndef GSP_RI_STACK
#define GSP_RI_STACK 48
#endif
constexpr uint32_t kRiStack = GSP_RI_STACK;      // entries of a search's stack (a deeper path is not followed further: still correct)
#ifndef GSP_RI_BUDGET
#define GSP_RI_BUDGET 768
#endif
constexpr uint32_t kRiBudget = GSP_RI_BUDGET;    // node visits a search may spend
constexpr int32_t kRiNone = 0x7fffffff;

__device__ __forceinline__ uint32_t ri_index(int32_t code, uint32_t n) {
  return code >= 0 ? (uint32_t)code : (n - 1u) + (((uint32_t)~code) >> 2);
}
__device__ __forceinline__ int32_t ri_parent(int32_t code, const int32_t* __restrict__ parent_int,
                                             const int32_t* __restrict__ parent_leaf) {
  return code >= 0 ? parent_int[code] : parent_leaf[((uint32_t)~code) >> 2];
}
struct RiRec {  // a binary node's record, unpacked
  q4 lo[2], hi[2];
  int32_t code[2];
};
__device__ __forceinline__ RiRec ri_load(const q4* nodes2, int32_t b) {  // (no __restrict__: k_ri_apply reads what it wrote)
  const q4* m = nodes2 + 4ll * b;
  const q4 a = m[0], bq = m[1], d = m[2], k = m[3];
  RiRec r;
  r.lo[0] = mkq(a.x, a.y, a.z, 0.0f), r.hi[0] = mkq(a.w, bq.x, bq.y, 0.0f), r.code[0] = (int32_t)__float_as_uint(k.x);
  r.lo[1] = mkq(bq.z, bq.w, d.x, 0.0f), r.hi[1] = mkq(d.y, d.z, d.w, 0.0f), r.code[1] = (int32_t)__float_as_uint(k.y);
  return r;
}
__device__ __forceinline__ void ri_store(q4* nodes2, int32_t b, const RiRec& r) {
  q4* N = nodes2 + 4ll * b;
  N[0] = mkq(r.lo[0].x, r.lo[0].y, r.lo[0].z, r.hi[0].x);
  N[1] = mkq(r.hi[0].y, r.hi[0].z, r.lo[1].x, r.lo[1].y);
  N[2] = mkq(r.lo[1].z, r.hi[1].x, r.hi[1].y, r.hi[1].z);
  N[3] = mkq(__uint_as_float((uint32_t)r.code[0]), __uint_as_float((uint32_t)r.code[1]), 0.0f, 0.0f);
}
__device__ __forceinline__ float ri_union_area(const q4& alo, const q4& ahi, const q4& blo, const q4& bhi) {
  return half_area(mkq(fminf(alo.x, blo.x), fminf(alo.y, blo.y), fminf(alo.z, blo.z), 0.0f),
                   mkq(fmaxf(ahi.x, bhi.x), fmaxf(ahi.y, bhi.y), fmaxf(ahi.z, bhi.z), 0.0f));
}

__global__ __launch_bounds__(kBlock) void k_ri_search(uint32_t n, int32_t root, const q4* __restrict__ nodes2,
                                                      const int32_t* __restrict__ parent_int,
                                                      const int32_t* __restrict__ parent_leaf, int32_t* __restrict__ mv_x,
                                                      int32_t* __restrict__ mv_xp, float* __restrict__ mv_gain) {
  const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= 2u * n - 1u) return;
  const int32_t in = t < n - 1u ? (int32_t)t : make_leaf(t - (n - 1u), 1);
  int32_t best_x = kRiNone, best_xp = -1;
  float best = 0.0f;
  const int32_t P = in == root ? -1 : ri_parent(in, parent_int, parent_leaf);
  if (P >= 0 && P != root) {
    const RiRec rp = ri_load(nodes2, P);
    const int side = rp.code[0] == in ? 0 : 1;
    const q4 in_lo = rp.lo[side], in_hi = rp.hi[side];
    const float a_parent = ri_union_area(rp.lo[0], rp.hi[0], rp.lo[1], rp.hi[1]);
    float d_bound = 0.0f;
    int32_t pivot = P;
    uint32_t sib = ((uint32_t)P << 1) | (uint32_t)(side ^ 1);  // {node whose record holds it, side}
    q4 pv_lo = mkq(3.0e38f, 3.0e38f, 3.0e38f, 0.0f), pv_hi = mkq(-3.0e38f, -3.0e38f, -3.0e38f, 0.0f);
    uint32_t st_n[kRiStack];
    float st_d[kRiStack];
    uint32_t visits = 0;
    for (uint32_t guard = 0; guard < 4096u; ++guard) {  // (the depth of a tree the build accepts)
      uint32_t sp = 0;
      st_n[0] = sib, st_d[0] = d_bound, sp = 1;
      q4 sib_lo = pv_lo, sib_hi = pv_hi;
      bool first = true;
      while (sp > 0 && visits < kRiBudget) {
        --sp;
        const uint32_t e = st_n[sp];
        const float d_par = st_d[sp];
        const RiRec r = ri_load(nodes2, (int32_t)(e >> 1));
        const int s2 = (int)(e & 1u);
        if (first) sib_lo = r.lo[s2], sib_hi = r.hi[s2], first = false;
        if (d_par + a_parent <= best) continue;  // not even a free insertion below here beats the best
        ++visits;
        const float a_merged = ri_union_area(r.lo[s2], r.hi[s2], in_lo, in_hi);
        const float d_direct = a_parent - a_merged;  // P's old box goes, the merged one comes
        if (d_par + d_direct > best) {
          best = d_par + d_direct;
          best_x = r.code[s2];
          best_xp = (int32_t)(e >> 1);
        }
        if (r.code[s2] >= 0) {
          const float d = d_par + half_area(r.lo[s2], r.hi[s2]) - a_merged;  // this node grows to the merged box
          if (d + a_parent > best && sp + 2 <= kRiStack) {
            st_n[sp] = ((uint32_t)r.code[s2] << 1), st_d[sp] = d, ++sp;
            st_n[sp] = ((uint32_t)r.code[s2] << 1) | 1u, st_d[sp] = d, ++sp;
          }
        }
      }
      if (first) {  // budget spent before this sibling was read
        const RiRec r = ri_load(nodes2, (int32_t)(sib >> 1));
        sib_lo = r.lo[sib & 1u], sib_hi = r.hi[sib & 1u];
      }
      // climb: the pivot loses `in`
      pv_lo = mkq(fminf(pv_lo.x, sib_lo.x), fminf(pv_lo.y, sib_lo.y), fminf(pv_lo.z, sib_lo.z), 0.0f);
      pv_hi = mkq(fmaxf(pv_hi.x, sib_hi.x), fmaxf(pv_hi.y, sib_hi.y), fmaxf(pv_hi.z, sib_hi.z), 0.0f);
      if (pivot != P) {
        const RiRec r = ri_load(nodes2, pivot);
        d_bound += ri_union_area(r.lo[0], r.hi[0], r.lo[1], r.hi[1]) - half_area(pv_lo, pv_hi);
      }
      if (pivot == root || visits >= kRiBudget) break;
      const int32_t pp = parent_int[pivot];
      if (pp < 0) break;
      const RiRec r = ri_load(nodes2, pp);
      sib = ((uint32_t)pp << 1) | (uint32_t)(r.code[0] == pivot ? 1 : 0);
      pivot = pp;
    }
    // (the first candidate examined is N's own sibling with a gain of exactly 0: never chosen)
  }
  mv_x[t] = best_x;
  mv_xp[t] = best_xp;
  mv_gain[t] = best;
}

struct RiMove {
  int32_t N, P, S, G, X, XP;
  unsigned long long key;
  bool valid;
};
__device__ __forceinline__ RiMove ri_move(uint32_t t, uint32_t n, const q4* nodes2, const int32_t* parent_int,
                                          const int32_t* parent_leaf, const int32_t* mv_x, const int32_t* mv_xp,
                                          const float* mv_gain) {
  RiMove m;
  m.valid = false;
  const float g = mv_gain[t];
  m.X = mv_x[t];
  if (m.X == kRiNone || !(g > 0.0f)) return m;
  m.N = t < n - 1u ? (int32_t)t : make_leaf(t - (n - 1u), 1);
  m.P = ri_parent(m.N, parent_int, parent_leaf);
  const RiRec rp = ri_load(nodes2, m.P);
  m.S = rp.code[0] == m.N ? rp.code[1] : rp.code[0];
  m.G = parent_int[m.P];
  m.XP = mv_xp[t];
  m.key = ((unsigned long long)__float_as_uint(g) << 32) | (unsigned long long)t;
  m.valid = m.G >= 0 && m.XP >= 0 && m.X != m.S && m.X != m.P && m.XP != m.P;
  return m;
}
__global__ __launch_bounds__(kBlock) void k_ri_lock(uint32_t n, const q4* __restrict__ nodes2,
                                                    const int32_t* __restrict__ parent_int,
                                                    const int32_t* __restrict__ parent_leaf, const int32_t* __restrict__ mv_x,
                                                    const int32_t* __restrict__ mv_xp, const float* __restrict__ mv_gain,
                                                    unsigned long long* __restrict__ lock) {
  const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= 2u * n - 1u) return;
  const RiMove m = ri_move(t, n, nodes2, parent_int, parent_leaf, mv_x, mv_xp, mv_gain);
  if (!m.valid) return;
  atomicMax(&lock[ri_index(m.N, n)], m.key);
  atomicMax(&lock[ri_index(m.P, n)], m.key);
  atomicMax(&lock[ri_index(m.S, n)], m.key);
  atomicMax(&lock[ri_index(m.G, n)], m.key);
  atomicMax(&lock[ri_index(m.X, n)], m.key);
  atomicMax(&lock[ri_index(m.XP, n)], m.key);
}
__device__ __forceinline__ void ri_set_parent(int32_t code, int32_t p, int32_t* parent_int, int32_t* parent_leaf) {
  if (code >= 0) parent_int[code] = p; else parent_leaf[((uint32_t)~code) >> 2] = p;
}
__global__ __launch_bounds__(kBlock) void k_ri_apply(uint32_t n, q4* nodes2, int32_t* parent_int, int32_t* parent_leaf,
                                                     const int32_t* __restrict__ mv_x, const int32_t* __restrict__ mv_xp,
                                                     const float* __restrict__ mv_gain,
                                                     const unsigned long long* __restrict__ lock, uint32_t* __restrict__ applied) {
  const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= 2u * n - 1u) return;
  const RiMove m = ri_move(t, n, nodes2, parent_int, parent_leaf, mv_x, mv_xp, mv_gain);
  if (!m.valid) return;
  if (lock[ri_index(m.N, n)] != m.key || lock[ri_index(m.P, n)] != m.key || lock[ri_index(m.S, n)] != m.key ||
      lock[ri_index(m.G, n)] != m.key || lock[ri_index(m.X, n)] != m.key || lock[ri_index(m.XP, n)] != m.key)
    return;
  // every record and parent entry written below belongs to a node this move holds the lock of
  RiRec rp = ri_load(nodes2, m.P);
  const int sn = rp.code[0] == m.N ? 0 : 1;
  const q4 n_lo = rp.lo[sn], n_hi = rp.hi[sn], s_lo = rp.lo[sn ^ 1], s_hi = rp.hi[sn ^ 1];
  {  // G: the slot that held P takes S
    RiRec rg = ri_load(nodes2, m.G);
    const int sg = rg.code[0] == m.P ? 0 : 1;
    rg.code[sg] = m.S, rg.lo[sg] = s_lo, rg.hi[sg] = s_hi;
    ri_store(nodes2, m.G, rg);
  }
  q4 x_lo, x_hi;
  {  // XP (may be G, as rewritten above): the slot that held X takes P
    RiRec rx = ri_load(nodes2, m.XP);
    const int sx = rx.code[0] == m.X ? 0 : 1;
    x_lo = rx.lo[sx], x_hi = rx.hi[sx];
    rx.code[sx] = m.P;
    rx.lo[sx] = mkq(fminf(x_lo.x, n_lo.x), fminf(x_lo.y, n_lo.y), fminf(x_lo.z, n_lo.z), 0.0f);
    rx.hi[sx] = mkq(fmaxf(x_hi.x, n_hi.x), fmaxf(x_hi.y, n_hi.y), fmaxf(x_hi.z, n_hi.z), 0.0f);
    ri_store(nodes2, m.XP, rx);
  }
  rp.code[0] = m.X, rp.lo[0] = x_lo, rp.hi[0] = x_hi;
  rp.code[1] = m.N, rp.lo[1] = n_lo, rp.hi[1] = n_hi;
  ri_store(nodes2, m.P, rp);
  ri_set_parent(m.S, m.G, parent_int, parent_leaf);
  parent_int[m.P] = m.XP;
  ri_set_parent(m.X, m.P, parent_int, parent_leaf);
  ri_set_parent(m.N, m.P, parent_int, parent_leaf);
  atomicAdd(applied, 1u);
}
// every inner node's box, bottom-up, into the record of its parent (the leaves' boxes sit in their parents' records already).
// The hand-over between the finisher of a child and the thread that continues from the parent goes through memory the
// XCDs agree on: write-through (sc1) stores, drained before the arrive counter is bumped, and sc1 loads behind the counter --
// NOT __threadfence(), which at agent scope is an L2 write-back + invalidate per call (the first version spent 5 ms a round there).
// ARCHITECTURE ASSUMPTION (gfx942 / gfx950): an agent-scope atomic store / load is emitted with sc1 and goes through to
// the memory the eight XCDs share instead of staying in the issuing XCD's L2, and `s_waitcnt vmcnt(0)` retires this
// thread's stores before the counter is bumped.  A box that arrived late would make a parent too SMALL -- wrong hits, not
// a crash -- so the result is checked: k_ri_validate below recomputes every inner node's box from its children's after
// the last round (one pass, ~0.1 ms for a million triangles) and the build fails loudly on any difference.
__device__ __forceinline__ void ri_store_word(float* p, float v) {
  __hip_atomic_store((uint32_t*)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ri_load_word(const float* p) {
  return __uint_as_float(__hip_atomic_load((const uint32_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__global__ __launch_bounds__(kBlock) void k_ri_refit(uint32_t n, q4* nodes2, const int32_t* __restrict__ parent_int,
                                                     const int32_t* __restrict__ parent_leaf, uint32_t* arrive) {
  const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
  if (s >= n) return;
  int32_t node = parent_leaf[s];
  for (uint32_t guard = 0; node >= 0 && guard < 65536u; ++guard) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's box stores have left before it announces itself
    const uint32_t old = __hip_atomic_fetch_add(&arrive[node], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == 0) break;  // the other subtree is not done: whoever finishes it continues from here
    const int32_t pp = parent_int[node];
    if (pp < 0) break;
    const float* N = (const float*)(nodes2 + 4ll * node);  // record = {left lo, left hi, right lo, right hi, codes}: 6 words a side
    float w[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) w[k] = ri_load_word(N + k);
    float* Pn = (float*)(nodes2 + 4ll * pp);
    const int off = (int32_t)__float_as_uint(Pn[12]) == node ? 0 : 6;  // (the codes do not change in this kernel)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      ri_store_word(Pn + off + k, fminf(w[k], w[6 + k]));
      ri_store_word(Pn + off + 3 + k, fmaxf(w[3 + k], w[9 + k]));
    }
    node = pp;
  }
}

// every binary inner node: the box its parent's record holds for it == the union of the two boxes its own record holds,
// bit for bit (the refit computes exactly that union).  Runs in a launch of its own after the last round, so it reads
// what the memory holds, not what a cache of the refit kernel saw.
__global__ __launch_bounds__(kBlock) void k_ri_validate(uint32_t n, int32_t root, const q4* __restrict__ nodes2,
                                                        const int32_t* __restrict__ parent_int, uint32_t* __restrict__ bad) {
  const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= n - 1u || (int32_t)t == root) return;
  const int32_t pp = parent_int[t];
  if (pp < 0) {
    atomicAdd(bad, 1u);
    return;
  }
  const RiRec r = ri_load(nodes2, (int32_t)t), rp = ri_load(nodes2, pp);
  const int s = rp.code[0] == (int32_t)t ? 0 : 1;
  // (bit patterns, not float ==: a box of NaN coordinates -- garbage in -- still equals itself)
  auto same = [](float a, float b) { return __float_as_uint(a) == __float_as_uint(b); };
  const bool ok = rp.code[s] == (int32_t)t && same(rp.lo[s].x, fminf(r.lo[0].x, r.lo[1].x)) && same(rp.lo[s].y, fminf(r.lo[0].y, r.lo[1].y)) &&
                  same(rp.lo[s].z, fminf(r.lo[0].z, r.lo[1].z)) && same(rp.hi[s].x, fmaxf(r.hi[0].x, r.hi[1].x)) &&
                  same(rp.hi[s].y, fmaxf(r.hi[0].y, r.hi[1].y)) && same(rp.hi[s].z, fmaxf(r.hi[0].z, r.hi[1].z));
  if (!ok) atomicAdd(bad, 1u);
}

// ---- collapse of the binary tree into the wide tree (pt_trace.h), one level per launch pair ---------------------------
// A work item of a level = the binary node that becomes a wide node; the items of a level are consecutive wide nodes,
// and so are the inner children of every node (exclusive scan of the per-item inner-child counts), which is what lets
// a traversal name "the children of node X still to visit" as {base, which ones} instead of one pointer per child.
// Leaf children likewise take consecutive triangle slots: the collapse defines the final triangle order.
// The children of a wide node are the grandchildren of its binary node b ("parity" collapse): 2 to 4 of them.
__device__ __forceinline__ int gather_children(const q4* __restrict__ nodes2, int32_t b, Pick* e) {
  Pick l, r;
  load2(nodes2, b, l, r);
  int cnt = 0;
  if (l.code >= 0) {
    load2(nodes2, l.code, e[0], e[1]);
    cnt = 2;
  } else {
    e[cnt++] = l;
  }
  if (r.code >= 0) {
    load2(nodes2, r.code, e[cnt], e[cnt + 1]);
    cnt += 2;
  } else {
    e[cnt++] = r;
  }
  return cnt;
}

__global__ __launch_bounds__(kBlock) void k_wide_count(int count, const int32_t* __restrict__ items,
                                                       const q4* __restrict__ nodes2, uint32_t* __restrict__ n_inner,
                                                       uint32_t* __restrict__ n_leaf) {
  const int t = blockIdx.x * kBlock + threadIdx.x;
  if (t > count) return;
  if (t == count) {  // the scans run over count + 1 elements: the last one yields the totals
    n_inner[t] = 0;
    n_leaf[t] = 0;
    return;
  }
  Pick e[kWide];
  const int cnt = gather_children(nodes2, items[t], e);
  uint32_t ni = 0;
  for (int k = 0; k < cnt; ++k) ni += e[k].code >= 0 ? 1u : 0u;
  n_inner[t] = ni;
  n_leaf[t] = (uint32_t)cnt - ni;
}

// node_first: wide-node index of this level's item 0; child_first / tri_first: index of the first node of the NEXT
// level / the first triangle slot this level hands out
__global__ __launch_bounds__(kBlock) void k_wide_emit(int count, const int32_t* __restrict__ items,
                                                      const q4* __restrict__ nodes2, const uint32_t* __restrict__ inner_off,
                                                      const uint32_t* __restrict__ leaf_off, uint32_t node_first,
                                                      uint32_t child_first, uint32_t tri_first, q4* __restrict__ nodes_out,
                                                      int32_t* __restrict__ next_items, uint32_t* __restrict__ tri_src) {
  const int t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= count) return;
  Pick e[kWide];
  const int cnt = gather_children(nodes2, items[t], e);
  const uint32_t child_base = child_first + inner_off[t], tri_base = tri_first + leaf_off[t];
  q4* out = nodes_out + (size_t)kNodeQuads * (node_first + (uint32_t)t);
  WideChild wc[4];
  int ni = 0, nl = 0;
  for (int k = 0; k < cnt; ++k)  // inner children first, in the order the collapse found them
    if (e[k].code >= 0) {
      wc[ni].lo = e[k].lo, wc[ni].hi = e[k].hi;
      next_items[child_base - child_first + ni] = e[k].code;
      ++ni;
    }
  for (int k = 0; k < cnt; ++k)
    if (e[k].code < 0) {
      wc[ni + nl].lo = e[k].lo, wc[ni + nl].hi = e[k].hi;
      tri_src[tri_base + nl] = ((uint32_t)~e[k].code) >> 2;
      ++nl;
    }
  encode_node_w4(out, wc, ni, nl, child_base, tri_base);
}

// triangle packets into their final slots: slot s takes what Morton slot tri_src[s] held
__global__ __launch_bounds__(kBlock) void k_permute_tris(uint32_t n, const uint32_t* __restrict__ tri_src,
                                                         const q4* __restrict__ isect_in, const q4* __restrict__ shade_in,
                                                         const uint32_t* __restrict__ s2g_in, q4* __restrict__ isect,
                                                         q4* __restrict__ shade, uint32_t* __restrict__ s2g) {
  const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
  if (s >= n) return;
  const uint32_t m = tri_src[s];
#pragma unroll
  for (int k = 0; k < 3; ++k) isect[3ull * s + k] = isect_in[3ull * m + k];
#pragma unroll
  for (int k = 0; k < 4; ++k) shade[4ull * s + k] = shade_in[4ull * m + k];
  s2g[s] = s2g_in[m];
}

// ---- refit --------------------------------------------------------------------------------------------------------
// what a node record says about its children (pt_trace.h): ni inner ones at nodes child_base + k, then nl leaf ones at
// triangle slots tri_base + j; an unused position carries the inverted box (qlo 255, qhi 0)
__device__ __forceinline__ void node_children(const q4* n, int& ni, int& nl, uint32_t& child_base, uint32_t& tri_base) {
  ni = __popc(__float_as_uint(n[3].x) >> 28);
  child_base = __float_as_uint(n[2].z);
  tri_base = __float_as_uint(n[2].w) + (uint32_t)ni;
  const uint32_t qlx = __float_as_uint(n[1].x), qhx = __float_as_uint(n[1].w);
  nl = 0;
  for (int p = ni; p < kWide; ++p)
    if (!(((qlx >> (8 * p)) & 255u) == 255u && ((qhx >> (8 * p)) & 255u) == 0u)) ++nl;
}

// One level of the wide tree, deepest level first: a node's child boxes are the node boxes of the level below (box_lo /
// box_hi, written by the previous launch) and the padded boxes of its triangles (leaf_lo / leaf_hi by slot, k_bake); the record
// is encoded again in place -- origin, scales, quantised planes, the per-octant child orders -- with the same children in
// the same positions.
__global__ __launch_bounds__(kBlock) void k_refit_level(uint32_t first, uint32_t count, q4* __restrict__ nodes,
                                                        const q4* __restrict__ leaf_lo, const q4* __restrict__ leaf_hi,
                                                        q4* __restrict__ box_lo, q4* __restrict__ box_hi) {
  const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= count) return;
  const uint32_t node = first + t;
  q4* rec = nodes + (size_t)kNodeQuads * node;
  const q4 n[4] = {rec[0], rec[1], rec[2], rec[3]};
  int ni, nl;
  uint32_t child_base, tri_base;
  node_children(n, ni, nl, child_base, tri_base);
  WideChild wc[kWide];
  float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
  for (int k = 0; k < kWide; ++k) {  // (positions by constant index: the four boxes stay in registers)
    wc[k].lo = wc[k].hi = mkq(0.0f, 0.0f, 0.0f, 0.0f);
    if (k < ni + nl) {
      const q4* pl = k < ni ? box_lo + (child_base + (uint32_t)k) : leaf_lo + (tri_base + (uint32_t)(k - ni));
      const q4* ph = k < ni ? box_hi + (child_base + (uint32_t)k) : leaf_hi + (tri_base + (uint32_t)(k - ni));
      wc[k].lo = *pl;
      wc[k].hi = *ph;
      lo[0] = fminf(lo[0], wc[k].lo.x), lo[1] = fminf(lo[1], wc[k].lo.y), lo[2] = fminf(lo[2], wc[k].lo.z);
      hi[0] = fmaxf(hi[0], wc[k].hi.x), hi[1] = fmaxf(hi[1], wc[k].hi.y), hi[2] = fmaxf(hi[2], wc[k].hi.z);
    }
  }
  encode_node_w4(rec, wc, ni, nl, child_base, tri_base);
  box_lo[node] = mkq(lo[0], lo[1], lo[2], 0.0f);
  box_hi[node] = mkq(hi[0], hi[1], hi[2], 0.0f);
}

// The tree's quality as the traversal sees it: half the surface area of every child box, decoded from the quantised planes.
// A ray's chance of entering a box grows with that area, so the sum over the tree tracks the node visits per ray.
__global__ __launch_bounds__(kBlock) void k_wide_area(uint32_t num_nodes, const q4* __restrict__ nodes, double* __restrict__ area) {
  const uint32_t node = blockIdx.x * kBlock + threadIdx.x;
  if (node >= num_nodes) return;
  const q4* n = nodes + (size_t)kNodeQuads * node;
  const q4 a = n[0], b = n[1], c = n[2], d = n[3];
  const float sc[3] = {a.w, d.z, d.w};
  const uint32_t ql[3] = {__float_as_uint(b.x), __float_as_uint(b.y), __float_as_uint(b.z)};
  const uint32_t qh[3] = {__float_as_uint(b.w), __float_as_uint(c.x), __float_as_uint(c.y)};
  double sum = 0.0;
  for (int p = 0; p < kWide; ++p) {
    float e[3];
    bool used = true;
    for (int k = 0; k < 3; ++k) {
      const uint32_t l = (ql[k] >> (8 * p)) & 255u, h = (qh[k] >> (8 * p)) & 255u;
      if (l > h) used = false;
      e[k] = (float)(h - l) * sc[k];
    }
    if (used) sum += ((double)e[0] * e[1] + (double)e[1] * e[2]) + (double)e[2] * e[0];
  }
  area[node] = sum;
}

// Build scratch: ~30 arrays that live for one build.  One arena (reserve) instead of thirty hipMalloc / hipFree pairs -- each of
// those is a driver call of 0.1-0.2 ms, a third of a million-triangle build's wall time; a request the arena cannot hold falls
// back to its own allocation.
struct Scratch {
  std::vector<void*> ptrs;
  size_t bytes = 0;
  char* arena = nullptr;
  size_t arena_bytes = 0, arena_used = 0;
  ~Scratch() {
    for (void* p : ptrs) (void)hipFree(p);
    (void)hipFree(arena);
  }
  void reserve(size_t b) {  // best effort: without the arena every request allocates for itself
    if (arena || b == 0) return;
    if (hipMalloc((void**)&arena, b) == hipSuccess) arena_bytes = b, bytes += b;
    else arena = nullptr, (void)hipGetLastError();
  }
  template <class T>
  hipError_t alloc(T** p, size_t count) {
    size_t b = std::max<size_t>(count, 1) * sizeof(T);
    const size_t at = (arena_used + 255) & ~(size_t)255;
    if (arena && at + b <= arena_bytes) {
      *p = (T*)(arena + at);
      arena_used = at + b;
      return hipSuccess;
    }
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
  if (!b.arrays_external) {
    (void)hipFree(b.nodes);
    (void)hipFree(b.tri_isect);
    (void)hipFree(b.tri_shade);
  }
  (void)hipFree(b.slot_to_global);
  (void)hipFree(b.rf_leaf_lo);
  (void)hipFree(b.rf_leaf_hi);
  (void)hipFree(b.rf_box_lo);
  (void)hipFree(b.rf_box_hi);
  (void)hipFree(b.rf_area);
  (void)hipFree(b.rf_total);
  (void)hipFree(b.rf_tmp);
  (void)hipFree(b.rf_bounds);
  b = DeviceBvh{};
}

int build_bvh(hipStream_t stream, const BuildInput& in, DeviceBvh& out, std::string& err) {
  free_bvh(out);
  const uint32_t n = in.num_tris;
  // triangle slots are handed out from kFirstSlot on: a node's tri_base - ni (pt_trace.h) is then never negative and
  // fits the 28 bits the traversal packs it into; the leading slots stay all-zero triangles (det == 0: never hit)
  // ... and kWide - 1 all-zero slots FOLLOW the last triangle: an unused child position of a node is marked only by an inverted
  // quantised box, and when a node's extent is below ~8 ulp of the ray's distance to it the 8-ulp slack of the box test lets
  // such a position through as "leaf slot tri_base + p" -- a neighbouring triangle for most nodes (harmless: tested, and
  // accepted only if really hit), the slots behind the array for the last one
  const uint32_t slots = n + kFirstSlot + (kWide - 1);
  out.num_tris = n;
  out.first_slot = kFirstSlot;
  out.num_nodes = 0;
  size_t b_is = (size_t)slots * 48, b_sh = (size_t)slots * 64, b_map = (size_t)slots * 4;
  GSP_HIP_TRY(hipMalloc((void**)&out.tri_isect, b_is));
  GSP_HIP_TRY(hipMalloc((void**)&out.tri_shade, b_sh));
  GSP_HIP_TRY(hipMalloc((void**)&out.slot_to_global, b_map));
  out.bytes = b_is + b_sh + b_map;
  GSP_HIP_TRY(hipMemsetAsync(out.tri_isect, 0, b_is, stream));
  GSP_HIP_TRY(hipMemsetAsync(out.tri_shade, 0, b_sh, stream));
  GSP_HIP_TRY(hipMemsetAsync(out.slot_to_global, 0, b_map, stream));
  out.root = 0;  // the root is always node 0 (a node exists even for an empty scene)
  out.depth = 1;
  if (n == 0) {  // one node without children: every ray misses
    q4 node[kNodeQuads];
    encode_node_w4(node, nullptr, 0, 0, 0u, kFirstSlot);
    GSP_HIP_TRY(hipMalloc((void**)&out.nodes, kNodeAllocMin));
    GSP_HIP_TRY(hipMemcpyAsync(out.nodes, node, kNodeBytes, hipMemcpyHostToDevice, stream));
    out.bytes += kNodeBytes;
    out.num_nodes = 1;
    out.level_first = {0u, 1u};
    GSP_HIP_TRY(hipStreamSynchronize(stream));
    return GSP_OK;
  }

  Scratch S;
  S.reserve(640ull * n + (1ull << 20));  // (the requests below sum to ~600 B per triangle + the sort's and scans' temporaries)
  q4 *isect_g, *shade_g, *lo_g, *hi_g, *leaf_lo, *leaf_hi, *int_lo, *int_hi, *nodes2, *isect_m, *shade_m;
  uint32_t *bounds, *vals_in, *vals_out, *arrive, *flag, *idx4, *s2g_m;
  uint64_t *keys_in, *keys_out;
  int32_t *child_l, *child_r, *parent_int, *parent_leaf;
  GSP_HIP_TRY(S.alloc(&isect_g, 3ull * n));
  GSP_HIP_TRY(S.alloc(&shade_g, 4ull * n));
  GSP_HIP_TRY(S.alloc(&isect_m, 3ull * n));  // packets in Morton order (the collapse defines the final order)
  GSP_HIP_TRY(S.alloc(&shade_m, 4ull * n));
  GSP_HIP_TRY(S.alloc(&s2g_m, n));
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
  GSP_HIP_TRY(S.alloc(&child_l, n));
  GSP_HIP_TRY(S.alloc(&child_r, n));
  GSP_HIP_TRY(S.alloc(&parent_int, n));
  GSP_HIP_TRY(S.alloc(&parent_leaf, n));
  GSP_HIP_TRY(S.alloc(&nodes2, 4ull * n));
  GSP_HIP_TRY(S.alloc(&flag, n + 1ull));
  GSP_HIP_TRY(S.alloc(&idx4, n + 1ull));

  const uint32_t init_bounds[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u};
  GSP_HIP_TRY(hipMemcpyAsync(bounds, init_bounds, sizeof(init_bounds), hipMemcpyHostToDevice, stream));
  GSP_HIP_TRY(hipMemsetAsync(arrive, 0, sizeof(uint32_t) * n, stream));

  hipLaunchKernelGGL(k_bake, dim3(blocks_for(n)), dim3(kBlock), 0, stream, in, isect_g, shade_g, lo_g, hi_g, bounds, (const uint32_t*)nullptr, 0u);
  hipLaunchKernelGGL(k_morton, dim3(blocks_for(n)), dim3(kBlock), 0, stream, n, lo_g, hi_g, bounds, keys_in, vals_in);
  GSP_HIP_TRY(hipGetLastError());

  size_t temp_bytes = 0;
  GSP_HIP_TRY(rocprim::radix_sort_pairs(nullptr, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0, 63, stream));
  void* temp = nullptr;
  GSP_HIP_TRY(S.alloc((char**)&temp, temp_bytes));
  GSP_HIP_TRY(rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0, 63, stream));

  hipLaunchKernelGGL(k_scatter, dim3(blocks_for(n)), dim3(kBlock), 0, stream, n, vals_out, isect_g, shade_g, lo_g, hi_g,
                     isect_m, shade_m, leaf_lo, leaf_hi, s2g_m);
  GSP_HIP_TRY(hipGetLastError());
  if (n == 1) {  // no binary inner node: one wide node with the triangle as its only child
    q4 box[2];
    GSP_HIP_TRY(hipMemcpyAsync(&box[0], leaf_lo, sizeof(q4), hipMemcpyDeviceToHost, stream));
    GSP_HIP_TRY(hipMemcpyAsync(&box[1], leaf_hi, sizeof(q4), hipMemcpyDeviceToHost, stream));
    GSP_HIP_TRY(hipStreamSynchronize(stream));
    q4 node[kNodeQuads];
    WideChild c[8];
    c[0].lo = box[0];
    c[0].hi = box[1];
    encode_node_w4(node, c, 0, 1, 0u, kFirstSlot);
    GSP_HIP_TRY(hipMalloc((void**)&out.nodes, kNodeAllocMin));
    GSP_HIP_TRY(hipMemcpyAsync(out.nodes, node, kNodeBytes, hipMemcpyHostToDevice, stream));
    GSP_HIP_TRY(hipMemcpyAsync(out.tri_isect + 3ull * kFirstSlot, isect_m, 48, hipMemcpyDeviceToDevice, stream));
    GSP_HIP_TRY(hipMemcpyAsync(out.tri_shade + 4ull * kFirstSlot, shade_m, 64, hipMemcpyDeviceToDevice, stream));
    GSP_HIP_TRY(hipMemcpyAsync(out.slot_to_global + kFirstSlot, s2g_m, 4, hipMemcpyDeviceToDevice, stream));
    out.bytes += kNodeBytes;
    out.num_nodes = 1;
    out.level_first = {0u, 1u};
    GSP_HIP_TRY(hipStreamSynchronize(stream));
    return GSP_OK;
  }

  int32_t root2 = 0;  // binary root
  uint32_t* ri_bad = nullptr;  // violations k_ri_validate counted (read back with the first collapse level's totals)
  {
    {
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
      while (m > (uint32_t)kPlocTail) {
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
      if (m > 1) {  // the remaining rounds in one launch
        int32_t* tail_result = (int32_t*)kpos;  // (two words of a scan array nobody reads any more)
        hipLaunchKernelGGL(k_ploc_tail, dim3(1), dim3(kPlocTailThreads), 0, stream, (int)m, node_base, code_a, lo_a, hi_a, code_b, lo_b, hi_b,
                           nodes2, parent_int, parent_leaf, tail_result);
        GSP_HIP_TRY(hipGetLastError());
        int32_t res[2] = {0, 0};
        GSP_HIP_TRY(hipMemcpyAsync(res, tail_result, sizeof(res), hipMemcpyDeviceToHost, stream));
        GSP_HIP_TRY(hipStreamSynchronize(stream));
        if (res[1] != 1) {
          err = "PLOC made no progress (internal error)";
          return GSP_ERR_DEVICE;
        }
        root2 = res[0];
      } else {
        GSP_HIP_TRY(hipMemcpyAsync(&root2, code_a, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        GSP_HIP_TRY(hipStreamSynchronize(stream));
      }
      // leaf_lo/leaf_hi may have been overwritten by the ping-pong: nothing below reads them again
      // ---- parallel reinsertion rounds (above): no host round trip inside
      const int rounds = in.reinsert_rounds < 0 ? kReinsertRounds : std::min(in.reinsert_rounds, 64);
      if (rounds > 0 && n >= 16) {
        const size_t idx = 2ull * n;
        int32_t *mv_x, *mv_xp;
        float* mv_gain;
        unsigned long long* lock;
        uint32_t* applied;
        GSP_HIP_TRY(S.alloc(&mv_x, idx));
        GSP_HIP_TRY(S.alloc(&mv_xp, idx));
        GSP_HIP_TRY(S.alloc(&mv_gain, idx));
        GSP_HIP_TRY(S.alloc(&lock, idx));
        GSP_HIP_TRY(S.alloc(&applied, 64));
        GSP_HIP_TRY(hipMemsetAsync(applied, 0, 64 * sizeof(uint32_t), stream));
        const uint32_t nb = blocks_for(2ull * n - 1);
        for (int r = 0; r < rounds; ++r) {
          GSP_HIP_TRY(hipMemsetAsync(lock, 0, idx * sizeof(unsigned long long), stream));
          hipLaunchKernelGGL(k_ri_search, dim3(nb), dim3(kBlock), 0, stream, n, root2, nodes2, parent_int, parent_leaf, mv_x, mv_xp, mv_gain);
          hipLaunchKernelGGL(k_ri_lock, dim3(nb), dim3(kBlock), 0, stream, n, nodes2, parent_int, parent_leaf, mv_x, mv_xp, mv_gain, lock);
          hipLaunchKernelGGL(k_ri_apply, dim3(nb), dim3(kBlock), 0, stream, n, nodes2, parent_int, parent_leaf, mv_x, mv_xp, mv_gain, lock,
                             applied + r);
          GSP_HIP_TRY(hipMemsetAsync(arrive, 0, sizeof(uint32_t) * n, stream));
          hipLaunchKernelGGL(k_ri_refit, dim3(blocks_for(n)), dim3(kBlock), 0, stream, n, nodes2, parent_int, parent_leaf, arrive);
          GSP_HIP_TRY(hipGetLastError());
        }
        GSP_HIP_TRY(S.alloc(&ri_bad, 1));
        GSP_HIP_TRY(hipMemsetAsync(ri_bad, 0, sizeof(uint32_t), stream));
        hipLaunchKernelGGL(k_ri_validate, dim3(blocks_for(n - 1)), dim3(kBlock), 0, stream, n, root2, nodes2, parent_int, ri_bad);
        GSP_HIP_TRY(hipGetLastError());
      }
    }
  }
  // ---- collapse to the wide tree, level by level ----
  const uint32_t n_int = n - 1;  // binary inner nodes: an upper bound of the wide nodes
  q4* wide = nullptr;
  GSP_HIP_TRY(S.alloc(&wide, (size_t)kNodeQuads * n_int));
  int32_t *items_a = child_l, *items_b = child_r;  // (free again after the hierarchy pass)
  uint32_t *n_inner = flag, *n_leaf = idx4, *inner_off, *leaf_off, *tri_src;
  GSP_HIP_TRY(S.alloc(&tri_src, (size_t)n + kFirstSlot));
  GSP_HIP_TRY(S.alloc(&inner_off, n + 1ull));
  GSP_HIP_TRY(S.alloc(&leaf_off, n + 1ull));
  size_t scan_bytes = 0;
  GSP_HIP_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, n_inner, inner_off, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), stream));
  void* scan_tmp = nullptr;
  GSP_HIP_TRY(S.alloc((char**)&scan_tmp, scan_bytes));
  GSP_HIP_TRY(hipMemcpyAsync(items_a, &root2, sizeof(int32_t), hipMemcpyHostToDevice, stream));
  uint32_t count = 1, node_first = 0, tri_done = kFirstSlot, levels = 0;
  out.level_first.clear();
  // 4-wide default: the parity collapse (children = grandchildren).  On PLOC trees it beats the greedy surface-area
  // collapse -- bench scene 14.7 vs 15.1 nodes per extension ray, 7.5 vs 9.2 per shadow ray (profiles/r03_collapse.txt;
  // the CPU probe agrees: 10.2 vs 10.4, 7.7 vs 8.3) -- while on top-down SAH trees it is the other way round.
  while (count > 0) {
    out.level_first.push_back(node_first);
    hipLaunchKernelGGL(k_wide_count, dim3(blocks_for(count + 1ull)), dim3(kBlock), 0, stream, (int)count, items_a, nodes2, n_inner, n_leaf);
    GSP_HIP_TRY(rocprim::exclusive_scan(scan_tmp, scan_bytes, n_inner, inner_off, 0u, (size_t)count + 1, rocprim::plus<uint32_t>(), stream));
    GSP_HIP_TRY(rocprim::exclusive_scan(scan_tmp, scan_bytes, n_leaf, leaf_off, 0u, (size_t)count + 1, rocprim::plus<uint32_t>(), stream));
    const uint32_t child_first = node_first + count;
    hipLaunchKernelGGL(k_wide_emit, dim3(blocks_for(count)), dim3(kBlock), 0, stream, (int)count, items_a, nodes2, inner_off, leaf_off,
                       node_first, child_first, tri_done, wide, items_b, tri_src);
    uint32_t totals[2] = {0, 0}, bad = 0;
    GSP_HIP_TRY(hipMemcpyAsync(&totals[0], inner_off + count, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    GSP_HIP_TRY(hipMemcpyAsync(&totals[1], leaf_off + count, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    if (levels == 0 && ri_bad) GSP_HIP_TRY(hipMemcpyAsync(&bad, ri_bad, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    GSP_HIP_TRY(hipStreamSynchronize(stream));
    GSP_HIP_TRY(hipGetLastError());
    if (bad) {
      err = "BVH build: " + std::to_string(bad) + " node boxes of the reinserted tree do not bound their children (internal error: "
            "the cross-XCD hand-over of k_ri_refit failed)";
      return GSP_ERR_DEVICE;
    }
    node_first = child_first;
    tri_done += totals[1];
    count = totals[0];
    std::swap(items_a, items_b);
    ++levels;
    if ((uint64_t)node_first + count > n_int || levels > 4096) {
      err = "wide collapse overran its node bound (internal error)";
      return GSP_ERR_DEVICE;
    }
  }
  if (tri_done != n + kFirstSlot) {
    err = "wide collapse lost triangles (internal error)";
    return GSP_ERR_DEVICE;
  }
  const uint32_t num_nodes = node_first;
  out.level_first.push_back(num_nodes);
  if (num_nodes >= kMaxNodes) {
    err = "scene too large: more BVH nodes than the traversal's node index holds (internal error: gsp_upload_scene bounds the triangle count)";
    return GSP_ERR_INVALID;
  }
  out.num_nodes = num_nodes;
  const size_t b_nodes = (size_t)num_nodes * kNodeBytes;
  GSP_HIP_TRY(hipMalloc((void**)&out.nodes, std::max<size_t>(b_nodes, kNodeAllocMin)));  // k_trace stages the first kTopNodes records
  out.bytes += b_nodes;
  GSP_HIP_TRY(hipMemcpyAsync(out.nodes, wide, b_nodes, hipMemcpyDeviceToDevice, stream));
  hipLaunchKernelGGL(k_permute_tris, dim3(blocks_for(n)), dim3(kBlock), 0, stream, n, tri_src + kFirstSlot, isect_m, shade_m, s2g_m,
                     out.tri_isect + 3ull * kFirstSlot, out.tri_shade + 4ull * kFirstSlot, out.slot_to_global + kFirstSlot);
  GSP_HIP_TRY(hipGetLastError());
  GSP_HIP_TRY(hipStreamSynchronize(stream));
  out.depth = levels;  // levels of the wide tree: a traversal stacks at most one node group per level
  return GSP_OK;
}

namespace {
int tree_area(hipStream_t stream, const DeviceBvh& bvh, double* sum, std::string& err) {
  hipLaunchKernelGGL(k_wide_area, dim3(blocks_for(bvh.num_nodes)), dim3(kBlock), 0, stream, bvh.num_nodes, bvh.nodes, bvh.rf_area);
  GSP_HIP_TRY(hipGetLastError());
  size_t tb = bvh.rf_tmp_bytes;  // (no atomics: the same sum every run)
  GSP_HIP_TRY(rocprim::reduce(bvh.rf_tmp, tb, bvh.rf_area, bvh.rf_total, 0.0, (size_t)bvh.num_nodes, rocprim::plus<double>(), stream));
  GSP_HIP_TRY(hipMemcpyAsync(sum, bvh.rf_total, sizeof(double), hipMemcpyDeviceToHost, stream));
  GSP_HIP_TRY(hipStreamSynchronize(stream));
  return GSP_OK;
}
}  // namespace

int refit_bvh(hipStream_t stream, const BuildInput& in, DeviceBvh& bvh, double* growth, std::string& err) {
  *growth = 1.0;
  const uint32_t n = bvh.num_tris;
  if (in.num_tris != n || bvh.level_first.size() < 2 || bvh.level_first.back() != bvh.num_nodes || !bvh.nodes) {
    err = "refit_bvh: the tree does not belong to this triangle list (internal error)";
    return GSP_ERR_DEVICE;
  }
  if (n == 0) return GSP_OK;
  const uint32_t slots = n + kFirstSlot + (kWide - 1);
  if (!bvh.rf_bounds) {  // first refit of this tree (or an earlier one ran out of memory half-way: start over)
    for (void** q : {(void**)&bvh.rf_leaf_lo, (void**)&bvh.rf_leaf_hi, (void**)&bvh.rf_box_lo, (void**)&bvh.rf_box_hi, (void**)&bvh.rf_area,
                     (void**)&bvh.rf_total, &bvh.rf_tmp}) {
      (void)hipFree(*q);
      *q = nullptr;
    }
    size_t tb = 0;
    GSP_HIP_TRY(rocprim::reduce(nullptr, tb, bvh.rf_area, bvh.rf_total, 0.0, (size_t)bvh.num_nodes, rocprim::plus<double>(), stream));
    tb = std::max<size_t>(tb, 16);
    GSP_HIP_TRY(hipMalloc((void**)&bvh.rf_leaf_lo, sizeof(q4) * slots));
    GSP_HIP_TRY(hipMalloc((void**)&bvh.rf_leaf_hi, sizeof(q4) * slots));
    GSP_HIP_TRY(hipMalloc((void**)&bvh.rf_box_lo, sizeof(q4) * bvh.num_nodes));
    GSP_HIP_TRY(hipMalloc((void**)&bvh.rf_box_hi, sizeof(q4) * bvh.num_nodes));
    GSP_HIP_TRY(hipMalloc((void**)&bvh.rf_area, sizeof(double) * bvh.num_nodes));
    GSP_HIP_TRY(hipMalloc((void**)&bvh.rf_total, sizeof(double)));
    GSP_HIP_TRY(hipMalloc((void**)&bvh.rf_tmp, tb));
    bvh.rf_tmp_bytes = tb;
    GSP_HIP_TRY(hipMalloc((void**)&bvh.rf_bounds, 8 * sizeof(uint32_t)));  // (last: marks the set complete)
    bvh.bytes += 2 * sizeof(q4) * ((size_t)slots + bvh.num_nodes) + sizeof(double) * (bvh.num_nodes + 1ull) + tb + 32;
  }
  int rc;
  if (bvh.area_built <= 0.0) {  // the tree is as built: its own measure first
    rc = tree_area(stream, bvh, &bvh.area_built, err);
    if (rc != GSP_OK) return rc;
  }
  hipLaunchKernelGGL(k_bake, dim3(blocks_for(n)), dim3(kBlock), 0, stream, in, bvh.tri_isect, bvh.tri_shade, bvh.rf_leaf_lo, bvh.rf_leaf_hi,
                     (uint32_t*)nullptr, (const uint32_t*)bvh.slot_to_global, kFirstSlot);
  GSP_HIP_TRY(hipGetLastError());
  for (size_t l = bvh.level_first.size() - 1; l-- > 0;) {
    const uint32_t first = bvh.level_first[l], count = bvh.level_first[l + 1] - first;
    if (count == 0) continue;
    hipLaunchKernelGGL(k_refit_level, dim3(blocks_for(count)), dim3(kBlock), 0, stream, first, count, bvh.nodes, bvh.rf_leaf_lo, bvh.rf_leaf_hi,
                       bvh.rf_box_lo, bvh.rf_box_hi);
  }
  GSP_HIP_TRY(hipGetLastError());
  double now = 0.0;
  rc = tree_area(stream, bvh, &now, err);  // (synchronises: the caller's instance tables may go out of scope)
  if (rc != GSP_OK) return rc;
  ++bvh.refits;
  *growth = bvh.area_built > 0.0 ? now / bvh.area_built : (now > 0.0 ? 1.0e30 : 1.0);
  return GSP_OK;
}

}  // namespace gsp
