// pt_wavetrace.h -- persistent wave64 BVH traversal kernel (device only).
//
// The extend (closest hit), connect (any hit) and test-hook kernels are all this
// one loop with a different ray source / result sink (`IO`).
//
// Why it looks like this (rocprofv3 PMC, profiles/r01_a_pmc_summary.txt and r01_b_*): the
// first version (one ray per lane per grid-stride iteration) kept the VALU pipes ~85 % busy
// with only ~17 % of the lanes enabled -- rays of very different length share a wave, and
// lanes in a leaf wait for lanes in inner nodes.  The kernel is VALU-issue bound, so the
// lever is lanes doing useful work per issued instruction.  Hence a flat, wave-uniform
// state machine (every decision is a ballot + scalar branch, no divergent loops):
//   * each lane owns a ray state {cur node, postponed leaf, stack}; per loop iteration the
//     wave runs a few node steps for all lanes that sit on an inner node, and a leaf step
//     (ONE triangle test per lane; longer leaves stay pending) only when kLeafBatch lanes have
//     a leaf pending or nothing else can run -- triangle tests execute with many lanes enabled
//     (leaf postponing after Aila & Laine 2009, re-tuned for 64 lanes);
//   * lanes whose ray finished commit their result and are refilled from a wave-local pool
//     as soon as kRefillLanes of them are idle; the pool takes 256-ray chunks from 8
//     hand-out counters (one per blockIdx % 8 label = per XCD under round-robin placement;
//     speed only), i.e. one atomic per 256 rays on a line no other XCD touches;
//   * per-lane stack in LDS ([level][lane]: conflict-free) with an HBM spill region behind
//     it; a sentinel at the bottom removes the empty-stack test.
#pragma once
#include "pt_trace.h"

namespace gsp {

constexpr int kTraceBlock = 256;
#ifndef GSP_LDS_LEVELS
#define GSP_LDS_LEVELS 22
#endif
#ifndef GSP_TRACE_WAVES
#define GSP_TRACE_WAVES 7  // waves per SIMD the register allocator must allow (<= 72 VGPRs; 8 would spill)
#endif
constexpr int kLdsStackDepth = GSP_LDS_LEVELS;  // LDS levels per lane (1 KB per level per block)
#ifndef GSP_REFILL_LANES
#define GSP_REFILL_LANES 16
#endif
#ifndef GSP_LEAF_BATCH
#define GSP_LEAF_BATCH 20
#endif
#ifndef GSP_STALL_BATCH
#define GSP_STALL_BATCH 8
#endif
#ifndef GSP_LEAF_BATCH_CLOSEST
#define GSP_LEAF_BATCH_CLOSEST 32
#endif
#ifndef GSP_BATCH_COMMIT
#define GSP_BATCH_COMMIT (ANY ? 32 : 24)
#endif
constexpr int kRefillLanes = GSP_REFILL_LANES;  // idle lanes that trigger a refill
constexpr int kLeafBatch = GSP_LEAF_BATCH;      // pending leaves that trigger a leaf step
#ifndef GSP_CHUNK_LARGE
#define GSP_CHUNK_LARGE 256
#endif
constexpr uint32_t kChunkLarge = GSP_CHUNK_LARGE;  // rays per hand-out (big queues)
constexpr uint32_t kChunkSmall = 64;        // ... when the queue is small: one ray per lane, all waves busy
constexpr int kWorkShards = 8;
constexpr int kWorkStride = 32;             // counters sit on separate 128-B lines
constexpr int32_t kSentinel = 0x7fffffff;

// Explicit address spaces: a generic pointer that may be LDS or HBM compiles to flat_load /
// flat_store on every push and pop; with typed pointers the LDS levels are ds_read/ds_write.
typedef __attribute__((address_space(3))) int32_t lds_i32;
typedef __attribute__((address_space(1))) int32_t glb_i32;

struct WaveStack {
  lds_i32* lds;    // &lds_stack[threadIdx.x]; level L of this lane lives at lds + L * kTraceBlock
  glb_i32* spill;  // &spill[global thread], stride spill_stride
  uint32_t spill_stride;
  // BYTE offset of the next free level from `lds` (level * kLevelBytes): a push or pop is one add and a
  // ds access with an immediate offset -- no per-access shifts (v_lshl_or_b32 issues at half the rate
  // of v_add_u32 on this chip, profiles/r01_h_microbench/valu_rate.txt)
  uint32_t sp;
  static constexpr uint32_t kLevelBytes = 4u * kTraceBlock;
  static constexpr uint32_t kLdsBytes = (uint32_t)kLdsStackDepth * kLevelBytes;
  __device__ __forceinline__ lds_i32* at(uint32_t off) const {
    return (lds_i32*)((__attribute__((address_space(3))) char*)lds + off);
  }
  // Slow, per-lane form: LDS level or HBM spill level.
  __device__ __forceinline__ void store_at(uint32_t off, int32_t v) {
    if (off < kLdsBytes) *at(off) = v;
    else spill[(size_t)((off - kLdsBytes) / kLevelBytes) * spill_stride] = v;
  }
  __device__ __forceinline__ int32_t load_at(uint32_t off) {
    int32_t v;
    if (off < kLdsBytes) v = *at(off);
    else v = spill[(size_t)((off - kLdsBytes) / kLevelBytes) * spill_stride];
    return v;
  }
  __device__ __forceinline__ void push(int32_t v) {
    store_at(sp, v);
    sp += kLevelBytes;
  }
  // Hot path: the LDS-or-spill decision is taken once per wave (a ballot and a scalar branch);
  // almost always every lane is inside the LDS levels and the access is a bare ds_read/ds_write.
  __device__ __forceinline__ int32_t pop() {
    sp -= kLevelBytes;
    if (__builtin_expect(__ballot(sp >= kLdsBytes) == 0, 1)) return *at(sp);
    return load_at(sp);
  }
  // Pushes the m = mb / kLevelBytes (0..3) entries e1 (nearest of the three) .. e3 (farthest), farthest
  // first.  Hot path: three stores at FIXED offsets from the old top -- which value goes where depends on m,
  // and whatever lands above the new top is never read.
  __device__ __forceinline__ void push_sorted(uint32_t mb, int32_t e1, int32_t e2, int32_t e3) {
    if (__builtin_expect(__ballot(sp + 3u * kLevelBytes > kLdsBytes) == 0, 1)) {
      const int32_t w0 = mb == 3u * kLevelBytes ? e3 : (mb == 2u * kLevelBytes ? e2 : e1);
      const int32_t w1 = mb == 3u * kLevelBytes ? e2 : e1;
      lds_i32* p = at(sp);
      p[0] = w0;
      p[kTraceBlock] = w1;
      p[2 * kTraceBlock] = e1;
    } else {
      if (mb > 2u * kLevelBytes) store_at(sp + mb - 3u * kLevelBytes, e3);
      if (mb > kLevelBytes) store_at(sp + mb - 2u * kLevelBytes, e2);
      if (mb > 0u) store_at(sp + mb - kLevelBytes, e1);
    }
    sp += mb;
  }
};

#ifdef GSP_WAVE_PROFILE
// [0] node steps (per wave) [1] lanes enabled in them [2] leaf steps [3] lanes enabled [4] loop passes
// [5] refill passes [6] lanes refilled [7] lanes idle (no ray) summed over node steps [8] lanes stalled (leaf pending, no node) over node steps
// [9] node steps after the hand-out ran dry [10] lanes enabled in them
__device__ unsigned long long g_wave_profile[16];
#endif
struct TraceStatsOut {
  unsigned long long* nodes;
  unsigned long long* tris;
  unsigned long long* rays;
};

// lanes set in a ballot, as a 32-bit scalar: comparing the 64-bit result of __popcll with a constant is compiled
// to a VALU v_cmp_*_u64 on broadcast values (five of them per loop pass)
__device__ __forceinline__ int wave_count(uint64_t m) {
  return __builtin_popcount((uint32_t)m) + __builtin_popcount((uint32_t)(m >> 32));
}

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// IO contract:
//   __device__ void load(uint32_t i, f3& o, f3& d, float& tmin, float& tmax) const;
//   __device__ void store(uint32_t i, const HitRec& h, uint32_t aux) const;   // h.slot < 0: miss / unoccluded;
//                                                  aux = p1.w of the accepted triangle (BSDF type)
template <bool ANY, bool STATS, class IO>
__global__ __launch_bounds__(kTraceBlock, GSP_TRACE_WAVES) void k_trace(const q4* __restrict__ nodes, const q4* __restrict__ tris,
                                                        int32_t root, const uint32_t* __restrict__ n_ptr,
                                                        uint32_t n_imm, uint32_t chunk, IO io,
                                                        uint32_t* __restrict__ work,
                                                        int32_t* __restrict__ spill, uint32_t spill_stride,
                                                        TraceStatsOut so) {
  __shared__ int32_t lds_stack[kLdsStackDepth * kTraceBlock];
#if GSP_LDS_TOP && !defined(GSP_TOP_GLOBAL_ONLY)
  __shared__ q4 lds_top[4 * kTopNodes];
  for (uint32_t i = threadIdx.x; i < 4 * kTopNodes; i += kTraceBlock) lds_top[i] = nodes[i];
  __syncthreads();
#endif
  const uint32_t n = n_ptr ? *n_ptr : n_imm;
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  const uint32_t shard = blockIdx.x % kWorkShards;
  uint32_t* my_work = work + shard * kWorkStride;

  WaveStack stk;
  stk.lds = (lds_i32*)lds_stack + threadIdx.x;
  stk.spill = (glb_i32*)spill + (size_t)blockIdx.x * kTraceBlock + threadIdx.x;
  stk.spill_stride = spill_stride;
  stk.sp = 0;

  // wave-uniform hand-out state
  uint32_t pool_next = 0, pool_end = 0;
  bool exhausted = false;

  // per-lane ray state
  int32_t cur = kSentinel, leaf = 0;
  uint32_t ri = 0xffffffffu, best_id = 0xffffffffu, best_aux = 0;
  RayBox rb = make_raybox(mk3(0, 0, 0), mk3(1, 1, 1));
  RayShear rs;
  rs.kx = rs.ky = rs.kz = 0;
  rs.Sx = rs.Sy = rs.Sz = 0.0f;
  float tmin = 0.0f, tmax = 0.0f;
  HitRec h;
  h.t = 0.0f;
  h.u = h.v = 0.0f;
  h.slot = -1;
  uint32_t c_nodes = 0, c_tris = 0, c_rays = 0;

#ifdef GSP_WAVE_PROFILE
  unsigned long long wp[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  for (;;) {
#ifdef GSP_WAVE_PROFILE
    ++wp[4];
#endif
    // ---- commit finished rays ------------------------------------------------------------
    // batched like the refill: the commit path (for shadow rays: three loads, the firefly test, two stores) is
    // issued for the whole wave, so it waits until enough lanes are out of work (A/B: any-hit kernel -13 % at 32,
    // closest-hit -1.6 % at 24; 40+ starves the wave)
    {
      const bool pending = ri != 0xffffffffu && cur == kSentinel && leaf == 0;
      const uint64_t pend_m = __ballot(pending);
      if (pend_m) {
        const uint64_t out_m = pend_m | __ballot(ri == 0xffffffffu);
        if (wave_count(out_m) >= GSP_BATCH_COMMIT || out_m == ~0ull) {
          if (pending) {
            io.store(ri, h, best_aux);
            ri = 0xffffffffu;
          }
        }
      }
    }
    // ---- refill idle lanes from the wave-local pool -----------------------------------------
    uint64_t idle_m = __ballot(ri == 0xffffffffu);
    if (!exhausted && wave_count(idle_m) >= kRefillLanes) {
#ifdef GSP_WAVE_PROFILE
      ++wp[5];
      wp[6] += __popcll(idle_m);
#endif
      while (idle_m) {  // wave-uniform
        if (pool_next >= pool_end) {
          uint32_t k = 0;
          if (lane == 0) k = atomicAdd(my_work, 1u);
          k = __shfl(k, 0);
          const uint64_t start = ((uint64_t)k * kWorkShards + shard) * chunk;
          if (start >= n) {
            exhausted = true;
            break;
          }
          pool_next = (uint32_t)start;
          pool_end = (uint32_t)(start + chunk < n ? start + chunk : n);
        }
        const uint32_t rank = (uint32_t)__popcll(idle_m & lt_mask);
        const uint32_t avail = pool_end - pool_next;
        if (((idle_m >> lane) & 1ull) && rank < avail) {
          ri = pool_next + rank;
          f3 d;
          f3 o;
          io.load(ri, o, d, tmin, tmax);
          rb = make_raybox(o, d);
          rs = make_shear(d);
          rs.Sz = comp(rb.inv, rs.kz);  // = 1 / d[kz], the same correctly rounded quotient make_shear computes
          h.t = tmax;
          h.u = h.v = 0.0f;
          h.slot = -1;
          best_id = 0xffffffffu;
          stk.sp = 0;
          stk.push(kSentinel);
          cur = root;
          leaf = 0;
          if (cur < 0) {  // the root itself is a leaf (single-triangle or empty scene)
            leaf = cur;
            cur = kSentinel;
            stk.sp = 0;
          }
          if (STATS) ++c_rays;
        }
        const uint32_t want = (uint32_t)__popcll(idle_m);
        pool_next += want < avail ? want : avail;
        idle_m = __ballot(ri == 0xffffffffu);
      }
    }
    // ---- what can run? -------------------------------------------------------------------------
    const bool on_node = (uint32_t)cur < (uint32_t)kSentinel;
    const uint64_t node_m = __ballot(on_node);
    const uint64_t leaf_m = __ballot(leaf < 0);
    if ((node_m | leaf_m) == 0) {
      if (exhausted || idle_m == 0) break;  // nothing in flight and nothing left to hand out
      continue;                             // (all lanes idle: the refill above runs next)
    }
    // ---- one inner-node step for every lane that sits on an inner node ---------------------------
    // closest hit: lanes that cannot advance without a leaf step (leaf pending, no inner node to work on)
    // trigger it early; lanes that still descend can wait for a fuller batch
    const uint64_t stall_m = leaf_m & ~node_m;
    const bool leaf_step = (!ANY && wave_count(stall_m) >= GSP_STALL_BATCH) ||
                           wave_count(leaf_m) >= (ANY ? kLeafBatch : GSP_LEAF_BATCH_CLOSEST) || node_m == 0;
    if (node_m != 0 && !leaf_step) {
      // measured on the 1M-triangle bench scene (scripts/ab_variants.sh): closest-hit 3 steps while >= 32 lanes
      // are on inner nodes, any-hit 4 steps while >= 24 are (+3.6 % Mrays/s over 2 steps / 40 lanes)
#ifndef GSP_NODE_REPS
#define GSP_NODE_REPS (ANY ? 4 : 3)
#endif
#ifndef GSP_REP_LANES
#define GSP_REP_LANES (ANY ? 24 : 32)
#endif
      // up to GSP_NODE_REPS node steps per pass through the bookkeeping above, as long as most
      // lanes are still on inner nodes
      for (int rep = 0; rep < GSP_NODE_REPS; ++rep) {
        const bool on = (uint32_t)cur < (uint32_t)kSentinel;
        if (rep > 0 && wave_count(__ballot(on)) < GSP_REP_LANES) break;
#ifdef GSP_WAVE_PROFILE
        ++wp[0];
        wp[1] += __popcll(__ballot(on));
        wp[7] += __popcll(__ballot(ri == 0xffffffffu));
        wp[8] += __popcll(__ballot(ri != 0xffffffffu && !on));
        if (exhausted) {  // end game: the hand-out is empty, idle lanes stay idle
          ++wp[9];
          wp[10] += __popcll(__ballot(on));
        }
#endif
      if (on) {
          // compressed 4-wide node: 4 quads (pt_trace.h, built by pt_bvh.hip through encode_node4)
          // (`cur` is the node's byte offset: 32-bit offset + uniform base, no 64-bit address arithmetic)
#if GSP_LDS_TOP && !defined(GSP_TOP_GLOBAL_ONLY)
          q4 n0, n1, n2, n3;
          if ((uint32_t)cur < kTopNodes * 64u) {  // top of the tree: the block's LDS copy
            const q4* nd = (const q4*)((const char*)lds_top + (uint32_t)cur);
            n0 = nd[0], n1 = nd[1], n2 = nd[2], n3 = nd[3];
          } else {
            const q4* nd = (const q4*)((const char*)nodes + (uint32_t)cur);
            n0 = nd[0], n1 = nd[1], n2 = nd[2], n3 = nd[3];
          }
#else
          const q4* nd = (const q4*)((const char*)nodes + (uint32_t)cur);
          const q4 n0 = nd[0], n1 = nd[1], n2 = nd[2], n3 = nd[3];
#endif
          if (STATS) ++c_nodes;
          constexpr uint32_t L = WaveStack::kLevelBytes;  // hit count kept in stack-offset units
          int32_t e0, e1, e2, e3;
          const uint32_t nb = node4_step<L>(n0, n1, n2, n3, rb, tmin, h.t, e0, e1, e2, e3);
          stk.push_sorted(nb > 0u ? nb - L : 0u, e1, e2, e3);
          if (nb > 0u) cur = e0;
          else cur = stk.pop();
          if (cur < 0 && leaf == 0) {  // first leaf: postpone it and keep descending
            leaf = cur;
            cur = stk.pop();
          }
        }
      }
      continue;
    }
    // ---- leaf step: triangle tests for every lane with a postponed leaf -----------------------------
#ifdef GSP_WAVE_PROFILE
    ++wp[2];
    wp[3] += __popcll(__ballot(leaf < 0));
#endif
    // one triangle per lane per step: a leaf with more triangles stays pending (first + 1, count - 1), so
    // short leaves do not idle while long ones finish and leaves that arrive in between join the next step
    if (leaf < 0) {
      const uint32_t c = (uint32_t)~leaf;
      const uint32_t first = c >> 2;
      bool stop = false;
      {
        const q4* p = tris + 3ll * first;
        const q4 p0 = p[0], p1 = p[1], p2 = p[2];
        if (STATS) ++c_tris;
        float t, u, v;
        if (intersect_tri(mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), rb.o, rs, tmin, tmax, t, u,
                          v)) {
          if (ANY) {
            h.t = t;
            h.slot = (int32_t)first;
            stop = true;
          } else {
            const uint32_t id = __float_as_uint(p0.w);
            if (t < h.t || (t == h.t && id < best_id)) {
              h.t = t;
              h.u = u;
              h.v = v;
              h.slot = (int32_t)first;
              best_id = id;
              best_aux = __float_as_uint(p1.w);
            }
          }
        }
      }
      if ((c & 3u) != 0u && !(ANY && stop)) {
        leaf = ~(int32_t)(c + 3u);  // first + 1 (bits 2..), count - 1 (bits 0..1): +4 - 1
      } else {
        leaf = 0;
        if (ANY && stop) {
          cur = kSentinel;
        } else if (cur < 0) {  // a second leaf was waiting in `cur`
          leaf = cur;
          cur = stk.pop();
        }
      }
    }
  }
#ifdef GSP_WAVE_PROFILE
  if (lane == 0 && !ANY)
    for (int k = 0; k < 12; ++k) atomicAdd(&g_wave_profile[k], wp[k]);
#endif
  if (STATS) {
    const unsigned long long a = wave_sum_u64(c_nodes), b = wave_sum_u64(c_tris), c = wave_sum_u64(c_rays);
    if (lane == 0) {
      atomicAdd(so.nodes, a);
      atomicAdd(so.tris, b);
      atomicAdd(so.rays, c);
    }
  }
}

}  // namespace gsp
