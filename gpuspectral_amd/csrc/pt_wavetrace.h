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
//   * each lane owns a ray state {node group, up to two postponed triangle groups, stack}; per loop iteration the
//     wave runs a few node steps for all lanes that have a child node to visit, and a leaf step
//     (ONE triangle test per lane; longer groups stay pending) only when kLeafBatch lanes have
//     triangles pending or nothing else can run -- triangle tests execute with many lanes enabled
//     (leaf postponing after Aila & Laine 2009, re-tuned for 64 lanes);
//   * lanes whose ray finished commit their result and are refilled from a wave-local pool
//     as soon as kRefillLanes of them are idle; the pool takes 256-ray chunks from 8
//     hand-out counters (one per blockIdx % 8 label = per XCD under round-robin placement;
//     speed only), i.e. one atomic per 256 rays on a line no other XCD touches;
//   * per-lane stack of node groups in LDS ([level][lane]: conflict-free) with an HBM spill region behind
//     it; a node step pushes at most ONE entry (the rest of the group it descends from: r03, pt_trace.h), so the
//     stack is as deep as the tree, not three times that; a sentinel at the bottom removes the empty-stack test;
//   * the first kTopNodes records of the node array (the top levels of the tree: half of a ray's node visits) are
//     staged into LDS by every block and read with ds_read_b128 (r03);
//   * the leaf step is straight-line code: every update a select on the variable's own register, the triangle test
//     without early exits and with its axis permutation applied through lane masks (pt_trace.h intersect_tri_rot).
// What it runs against (r03, profiles/r03_trace_bound.txt): no single resource -- 70-85 % of VALU issue (by class), of the
// L1 request path and of what 7 waves per SIMD cover in latency; work per ray (node visits) is what moves it.
#pragma once
#include "pt_trace.h"

namespace gsp {

constexpr int kTraceBlock = 256;
#ifndef GSP_LDS_LEVELS
#define GSP_LDS_LEVELS 16
#endif
#ifndef GSP_TRACE_WAVES
#define GSP_TRACE_WAVES 7  // waves per SIMD the register allocator must allow (<= 72 VGPRs; 8 would spill)
#endif
#ifndef GSP_TRACE_WAVES_ANY
#define GSP_TRACE_WAVES_ANY GSP_TRACE_WAVES  // ... of the any-hit instantiations (they keep no u, v, tie-break id)
#endif
constexpr int kStackWords = 1;                       // 32-bit words per stack entry (one node group, pt_trace.h pack_group)
constexpr int kLdsStackDepth = GSP_LDS_LEVELS;       // LDS levels (entries) per lane
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
#ifndef GSP_BATCH_COMMIT_ANY
#define GSP_BATCH_COMMIT_ANY 32
#endif
#ifndef GSP_BATCH_COMMIT
#define GSP_BATCH_COMMIT (ANY ? GSP_BATCH_COMMIT_ANY : 24)
#endif
// triangle groups a lane may hold while it keeps taking node steps (0: it waits for the leaf step as soon as it has
// triangles to test; 1: it descends on until a second group arrives)
#ifndef GSP_POSTPONE_ANY
#define GSP_POSTPONE_ANY 1
#endif
#ifndef GSP_POSTPONE_CLOSEST
#define GSP_POSTPONE_CLOSEST 1
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

// Explicit address spaces: a generic pointer that may be LDS or HBM compiles to flat_load /
// flat_store on every push and pop; with typed pointers the LDS levels are ds_read/ds_write.
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) uint16_t lds_u16;
typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(1))) uint32_t glb_u32;
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4f_t lds_v4f;

// the step table (pt_trace.h) in LDS
struct LdsStepTable {
  const __attribute__((address_space(3))) char* base;
  __device__ __forceinline__ uint32_t operator()(uint32_t byte_off) const { return *(const lds_u16*)(base + byte_off); }
};
__device__ __forceinline__ void stage_step_table(uint32_t* lds_words, uint32_t tid, uint32_t threads) {
  const uint32_t* src = (const uint32_t*)&kStepTable;
  for (uint32_t i = tid; i < kStepTableBytes / 4; i += threads) lds_words[i] = src[i];
}

struct WaveStack {
  lds_u32* lds;    // &lds_stack[threadIdx.x]; word L of this lane lives at lds + L * kTraceBlock
  glb_u32* spill;  // &spill[global thread], stride spill_stride
  uint32_t spill_stride;
  // BYTE offset of the next free word from `lds` (word * kWordBytes): a push or pop is one add and a
  // ds access with an immediate offset -- no per-access shifts (v_lshl_or_b32 issues at half the rate
  // of v_add_u32 on this chip, profiles/r01_h_microbench/valu_rate.txt)
  uint32_t sp;
  static constexpr uint32_t kWordBytes = 4u * kTraceBlock;
  static constexpr uint32_t kEntryBytes = kWordBytes * kStackWords;
  static constexpr uint32_t kLdsBytes = (uint32_t)kLdsStackDepth * kEntryBytes;
  __device__ __forceinline__ lds_u32* at(uint32_t off) const {
    return (lds_u32*)((__attribute__((address_space(3))) char*)lds + off);
  }
  // Slow, per-lane form: LDS word or HBM spill word.
  __device__ __forceinline__ void store_at(uint32_t off, uint32_t v) {
    if (off < kLdsBytes) *at(off) = v;
    else spill[(size_t)((off - kLdsBytes) / kWordBytes) * spill_stride] = v;
  }
  __device__ __forceinline__ uint32_t load_at(uint32_t off) {
    uint32_t v;
    if (off < kLdsBytes) v = *at(off);
    else v = spill[(size_t)((off - kLdsBytes) / kWordBytes) * spill_stride];
    return v;
  }
  // Hot path: the LDS-or-spill decision is taken once per wave (a ballot and a scalar branch);
  // almost always every lane is inside the LDS levels and the access is a bare ds_read / ds_write.
  __device__ __forceinline__ void push_group(uint32_t gb, uint32_t gs) {
    if (__builtin_expect(__ballot(sp + kEntryBytes > kLdsBytes) == 0, 1)) at(sp)[0] = pack_group(gb, gs);
    else store_at(sp, pack_group(gb, gs));
    sp += kEntryBytes;
  }
  __device__ __forceinline__ void pop_group(uint32_t& gb, uint32_t& gs) {
    sp -= kEntryBytes;
    uint32_t e;
    if (__builtin_expect(__ballot(sp >= kLdsBytes) == 0, 1)) e = at(sp)[0];
    else e = load_at(sp);
    gb = e >> kGroupBits;
    gs = e & kGroupMask;
  }
};

#ifdef GSP_WAVE_PROFILE
// [0] node steps (per wave) [1] lanes enabled in them [2] leaf steps [3] lanes enabled [4] loop passes
// [5] refill passes [6] lanes refilled [7] lanes idle (no ray) summed over node steps [8] lanes stalled (triangles pending, no node step possible) over node steps
// [9] node steps after the hand-out ran dry [10] lanes enabled in them
// r06 (scripts/trace_phase_budget.py): [11] commit blocks executed [12] lanes committing in them [13] passes of the refill's inner
// loop [14] node steps in which some lane fetched its record from HBM / L2 [15] lanes that did [16] node steps in which some lane
// read the LDS copy [17] lanes that did [18] rays refilled.  One set per instantiation kind: [0][..] closest hit, [1][..] any hit
__device__ unsigned long long g_wave_profile[2][24];
#endif
struct TraceStatsOut {
  unsigned long long* nodes;
  unsigned long long* tris;
  unsigned long long* rays;
  unsigned long long* hits = nullptr;       // rays that ended with a hit (any-hit: occluded shadow rays)
  unsigned long long* hit_nodes = nullptr;  // node records those rays read
  unsigned long long* lds_nodes = nullptr;  // node records served by the block's LDS copy of the top of the tree
  unsigned long long* no_tri = nullptr;     // rays that ended without one triangle test
  uint32_t* node_hist = nullptr;            // visits per node index / tests per triangle slot (collect_traversal_stats = 2):
  uint32_t* tri_hist = nullptr;             // which records would an LDS copy have to hold?
};

// r06 measurement builds (-DGSP_PAD_NODE=N / -DGSP_PAD_LEAF=N / -DGSP_PAD_BOOK=N): N more full-rate VALU instructions in that phase
// of k_trace, under the phase's own exec mask, on ONE private register (no memory, no dependence on the traversal; 72 VGPRs).  The extra
// kernel time per added instruction is what an issue slot of that phase is WORTH -- the bound on what removing slots can pay
// (scripts/trace_phase_budget.py, profiles/r06_trace_phase_budget.txt).  Results are unchanged.
template <int N>
__device__ __forceinline__ void pad_valu(uint32_t& a) {
#pragma unroll
  for (int i = 0; i < N / 2; ++i) asm volatile("v_add_u32 %0, %0, %0\n\tv_xor_b32 %0, 0x5bd1e995, %0" : "+v"(a));
}
#ifndef GSP_PAD_NODE
#define GSP_PAD_NODE 0
#endif
#ifndef GSP_PAD_LEAF
#define GSP_PAD_LEAF 0
#endif
#ifndef GSP_PAD_BOOK
#define GSP_PAD_BOOK 0
#endif

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
//   __device__ void load(uint32_t i, f3& o, f3& d, float& tmin, float& tmax, uint32_t& pay) const;
//   __device__ void store(uint32_t i, const HitRec& h, uint32_t aux, uint32_t pay) const;   // h.slot < 0: miss / unoccluded;
//                                                  aux = BSDF type of the accepted triangle; pay = a word of the ray's
//                                                  record that load() hands to store() through the traversal
//   static constexpr float kTmin, kTmax: >= 0 = every ray of this source has that bound (load() returns the same
//                                                  value): the kernel keeps it out of the registers
//   static constexpr bool kVersioned: the rays of this source belong to several versions of the geometry (pt_stages.h, geometry
//                                                  ring; gsp_update_instances without a drain).  Then also
//   __device__ void geometry(uint32_t i, uint32_t pay, uint32_t& node_off, uint32_t& tri_base) const;  // byte offset of the ray's
//                                                  node records from `nodes`, slot offset of its triangles
//   __device__ uint32_t top_offset() const;        // node_off of the version whose top of the tree the blocks stage into LDS
//                                                  kVersioned == false: the kernel is what it was before versions existed
//   static constexpr bool kSplit (versioned sources): the scene is split -- a STATIC tree at offset 0 that every version shares
//                                                  and, behind it, the ring with the tree of the EDITED instances: geometry()
//                                                  names the ray's version of that one (node_off != 0), the ray walks it first
//                                                  and the static tree second, with the same hit record (the closest-hit rule does
//                                                  not care which tree a triangle lives in; an any-hit ray that was stopped skips
//                                                  the second walk).  Also  __device__ uint32_t static_slots() const;  -- hit slots
//                                                  are counted through both trees: static ones first
// Rays [first, n) of the queue are traced (first > 0: the leading entries carry memoised results, pt_render.hip).
template <bool ANY, bool STATS, class IO>
__global__ __launch_bounds__(kTraceBlock, ANY ? GSP_TRACE_WAVES_ANY : GSP_TRACE_WAVES) void k_trace(const q4* __restrict__ nodes, const q4* __restrict__ tris,
                                                        const uint32_t* __restrict__ n_ptr,
                                                        uint32_t n_imm, uint32_t first, uint32_t chunk, IO io,
                                                        uint32_t* __restrict__ work,
                                                        uint32_t* __restrict__ spill, uint32_t spill_stride,
                                                        TraceStatsOut so) {
  __shared__ uint32_t lds_stack[kLdsStackDepth * kStackWords * kTraceBlock];
  static_assert((kLdsStackDepth * kStackWords * kTraceBlock * 4 + kStepTableBytes + kTopNodes * kNodeBytes) * (ANY ? GSP_TRACE_WAVES_ANY : GSP_TRACE_WAVES) <= 160 * 1024,
                "LDS per block x resident blocks per CU exceeds 160 KB");
  __shared__ uint32_t lds_table[kStepTableBytes / 4];
  stage_step_table(lds_table, threadIdx.x, kTraceBlock);
  // the top of the tree: every ray starts there, and a fetch from LDS is not one of the divergent 16-B requests of which
  // the vector-memory path takes one per cycle and CU (scripts/microbench/lane_fetch.hip) -- the rate both traversal
  // kernels run at
  // The copy is RECORD-major (node n = lds_top[4 n .. 4 n + 3], read with four ds_read_b128 at one address + offsets).  A
  // quad-major copy (quad k of node n at lds_top[k * kTopNodes + n], so that lanes reading the same quad of different nodes
  // spread over all 32 banks instead of 2 of the 8 four-bank groups) was tried in r04 and NOT kept: it halves
  // SQ_LDS_BANK_CONFLICT (1.89e8 -> 0.90e8 per closest-hit launch, 6.6e7 -> 1.2e7 any-hit), changes the closest-hit kernel by
  // nothing and makes the any-hit kernel 2.5 % slower (one more address op per LDS node step): the replays are hidden behind
  // the other six waves.  profiles/r04_ab_lds_layout.txt
  __shared__ q4 lds_top[kTopNodes > 0 ? kTopNodes * kNodeQuads : 1];
  uint32_t top_off = 0;  // (versioned sources: the newest version's top of the tree is the one in LDS)
  if constexpr (IO::kVersioned) top_off = io.top_offset();
  {
    const q4* top = IO::kVersioned ? (const q4*)((const char*)nodes + top_off) : nodes;
    for (uint32_t i = threadIdx.x; i < kTopNodes * kNodeQuads; i += kTraceBlock) lds_top[i] = top[i];
  }
  __syncthreads();
  const LdsStepTable tab{(const __attribute__((address_space(3))) char*)lds_table};
  const uint32_t n = n_ptr ? *n_ptr : n_imm;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t shard = blockIdx.x % kWorkShards;
  uint32_t* my_work = work + shard * kWorkStride;

  WaveStack stk;
  stk.lds = (lds_u32*)lds_stack + threadIdx.x;
  stk.spill = (glb_u32*)spill + (size_t)blockIdx.x * kTraceBlock + threadIdx.x;
  stk.spill_stride = spill_stride;
  stk.sp = 0;

  // wave-uniform hand-out state
  uint32_t pool_next = 0, pool_end = 0;
  bool exhausted = false;

  // per-lane ray state: the node group being worked on {gb, gs}, the triangle group(s) waiting for a leaf step
  // {tb, tm} (+ {tb2, tm2}: the lane stalls when both are taken)
  uint32_t gb = 0, gs = no_group<ANY>(), tb = 0, tm = 0, tb2 = 0, tm2 = 0;
  uint32_t ri = 0xffffffffu, best_id = 0xffffffffu;  // best_id: p0.w of the closest hit so far (id << 3 | BSDF type)
  uint32_t pay = 0;                                   // (sources that do not use it leave no register behind)
  uint32_t node_off = 0, tri_base = 0;                // versioned sources: where this ray's version of the geometry starts
  RayBox rb = make_raybox(mk3(0, 0, 0), mk3(1, 1, 1));
  RayShearRot rs;
  rs.m0 = rs.m1 = rs.ms = 0u;
  rs.Sx = rs.Sy = rs.Sz = 0.0f;
  float tmin_v = 0.0f, tmax_v = 0.0f;
#define tmin (IO::kTmin >= 0.0f ? IO::kTmin : tmin_v)
#define tmax (IO::kTmax >= 0.0f ? IO::kTmax : tmax_v)
  HitRec h;
  h.t = 0.0f;
  h.u = h.v = 0.0f;
  h.slot = -1;
  uint32_t c_nodes = 0, c_tris = 0, c_rays = 0, c_hits = 0, c_hit_nodes = 0, ray_nodes = 0, c_lds = 0, c_notri = 0, ray_tris = 0;  // (STATS instantiations only)

#ifdef GSP_WAVE_PROFILE
  unsigned long long wp[24] = {};
#endif
  uint32_t pad_a = lane;  // (GSP_PAD_* measurement builds only: unused and removed otherwise)
  for (;;) {
#ifdef GSP_WAVE_PROFILE
    ++wp[4];
#endif
    // ---- commit finished rays ------------------------------------------------------------
    // batched like the refill: the commit path (for shadow rays: three loads, the firefly test, two stores) is
    // issued for the whole wave, so it waits until enough lanes are out of work (A/B: any-hit kernel -13 % at 32,
    // closest-hit -1.6 % at 24; 40+ starves the wave)
    if constexpr (IO::kVersioned && IO::kSplit) {
      // a ray that has finished the tree of the edited instances (node_off != 0) goes on in the static one
      const bool second = ri != 0xffffffffu && group_empty<ANY>(gs) && tris_empty(tm) && node_off != 0u && !(ANY && h.slot >= 0);
      if (second) {
        node_off = 0u;
        tri_base = 0u;
        stk.sp = 0;
        stk.push_group(0u, no_group<ANY>());
        gb = 0u;
        gs = root_group<ANY>();
      } else if (ri != 0xffffffffu && group_empty<ANY>(gs) && tris_empty(tm)) {
        node_off = 0u;  // (done: an any-hit ray stopped in the first tree must not look unfinished)
      }
    }
    {
      const bool pending = ri != 0xffffffffu && group_empty<ANY>(gs) && tris_empty(tm);
      const uint64_t pend_m = __ballot(pending);
      if (pend_m) {
        const uint64_t out_m = pend_m | __ballot(ri == 0xffffffffu);
        if (wave_count(out_m) >= GSP_BATCH_COMMIT || out_m == ~0ull) {
#ifdef GSP_WAVE_PROFILE
          ++wp[11];
          wp[12] += __popcll(pend_m);
#endif
          if (pending) {
            io.store(ri, h, best_id & 7u, pay);
            ri = 0xffffffffu;
            if (STATS && h.slot >= 0) {
              ++c_hits;
              c_hit_nodes += ray_nodes;
            }
            if (STATS && ray_tris == 0) ++c_notri;
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
          const uint64_t start = ((uint64_t)k * kWorkShards + shard) * chunk + first;
          if (start >= n) {
            exhausted = true;
            break;
          }
          pool_next = (uint32_t)start;
          pool_end = (uint32_t)(start + chunk < n ? start + chunk : n);
        }
#ifdef GSP_WAVE_PROFILE
        ++wp[13];
#endif
        // set bits of idle_m below this lane (v_mbcnt: no per-lane mask register)
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle_m, 0u));
        const uint32_t avail = pool_end - pool_next;
        if (((idle_m >> lane) & 1ull) && rank < avail) {
          ri = pool_next + rank;
          f3 d;
          f3 o;
          io.load(ri, o, d, tmin_v, tmax_v, pay);
          if constexpr (IO::kVersioned) io.geometry(ri, pay, node_off, tri_base);
          rb = make_raybox(o, d);
          rs = make_shear_rot(d);
          rs.Sz = permute_axes(rs, rb.inv).z;  // = 1 / d[kz], the same correctly rounded quotient make_shear_rot computes
          h.t = tmax;
          h.u = h.v = 0.0f;
          h.slot = -1;
          best_id = 0xffffffffu;
          stk.sp = 0;
          stk.push_group(0u, no_group<ANY>());  // sentinel: popping it leaves the lane without node work
          gb = 0u;
          gs = root_group<ANY>();
          tm = tm2 = 0u;
          if (STATS) {
            ++c_rays;
            ray_nodes = 0;
            ray_tris = 0;
          }
        }
        const uint32_t want = (uint32_t)__popcll(idle_m);
#ifdef GSP_WAVE_PROFILE
        wp[18] += want < avail ? want : avail;
#endif
        pool_next += want < avail ? want : avail;
        idle_m = __ballot(ri == 0xffffffffu);
      }
    }
    // ---- what can run? -------------------------------------------------------------------------
    // a lane takes a node step when it has a child node to visit and room for the triangle group the step may produce
    constexpr bool kPostpone = (ANY ? GSP_POSTPONE_ANY : GSP_POSTPONE_CLOSEST) != 0;
    if constexpr (GSP_PAD_BOOK > 0) pad_valu<GSP_PAD_BOOK>(pad_a);
    const bool on_node = !group_empty<ANY>(gs) && tris_empty(kPostpone ? tm2 : tm);
    const uint64_t node_m = __ballot(on_node);
    const uint64_t leaf_m = __ballot(!tris_empty(tm));
    if ((node_m | leaf_m) == 0) {
      if (exhausted || idle_m == 0) break;  // nothing in flight and nothing left to hand out
      continue;                             // (all lanes idle: the refill above runs next)
    }
    // ---- one node step for every lane that can take one ------------------------------------------
    // closest hit: lanes that cannot advance without a leaf step (triangles pending, no node step possible)
    // trigger it early; lanes that still descend can wait for a fuller batch
    const uint64_t stall_m = leaf_m & ~node_m;
    const bool leaf_step = (!ANY && wave_count(stall_m) >= GSP_STALL_BATCH) ||
                           wave_count(leaf_m) >= (ANY ? kLeafBatch : GSP_LEAF_BATCH_CLOSEST) || node_m == 0;
    if (node_m != 0 && !leaf_step) {
      // measured on the 1M-triangle bench scene: closest-hit 5 steps while >= 32 lanes can take one (r03: the node step
      // lost its sort network, so the bookkeeping around it weighs more: 5 steps -2 % against 3,
      // profiles/r03_ab_trace_thresholds.txt), any-hit 6 steps while >= 24 can (r04, re-swept after the any-hit commit became a
      // no-op for unoccluded rays: 3 / 4 / 6 / 8 steps 45.0 / 43.7 / 42.5 / 41.2-41.3 ms ... 6 and 8 equal, profiles/r04_ab_anyhit_thresholds.txt)
#ifndef GSP_NODE_REPS
#define GSP_NODE_REPS (ANY ? 6 : 5)
#endif
#ifndef GSP_REP_LANES
#define GSP_REP_LANES (ANY ? 24 : 32)
#endif
      // up to GSP_NODE_REPS node steps per pass through the bookkeeping above, as long as most
      // lanes can still take one
#ifdef GSP_REPS_NOUNROLL
#pragma nounroll
#endif
      for (int rep = 0; rep < GSP_NODE_REPS; ++rep) {
        const bool on = !group_empty<ANY>(gs) && tris_empty(kPostpone ? tm2 : tm);
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
#ifdef GSP_WAVE_PROFILE
        bool wp_from_lds = false;  // (counted behind the block, by all lanes: lane 0 reports and may not be `on`)
#endif
        if (on) {
          // the nearest child of the current group (32-bit offset + uniform base, no 64-bit address arithmetic)
          const uint32_t noff_v = group_next<ANY>(gb, gs, rb, tab);
          const uint32_t noff = IO::kVersioned ? noff_v + node_off : noff_v;        // offset from `nodes`
          const uint32_t loff = IO::kVersioned ? noff - top_off : noff;             // offset inside the LDS copy, if below its size
          q4 nq[kNodeQuads];
#ifdef GSP_WAVE_PROFILE
          wp_from_lds = kTopNodes > 0 && loff < kTopNodes * kNodeBytes;
#endif
          if (kTopNodes > 0 && loff < kTopNodes * kNodeBytes) {
            const lds_v4f* nd = (const lds_v4f*)((const __attribute__((address_space(3))) char*)lds_top + loff);
#pragma unroll
            for (uint32_t k = 0; k < kNodeQuads; ++k) {
              const v4f_t q = nd[k];  // ds_read_b128
              nq[k] = make_q4(q.x, q.y, q.z, q.w);
            }
          } else {
            const q4* nd = (const q4*)((const char*)nodes + noff);
#pragma unroll
            for (uint32_t k = 0; k < kNodeQuads; ++k) nq[k] = nd[k];
          }
          if (STATS) {
            ++c_nodes;
            ++ray_nodes;
            if (kTopNodes > 0 && loff < kTopNodes * kNodeBytes) ++c_lds;
            if (so.node_hist) atomicAdd(so.node_hist + noff / kNodeBytes, 1u);
          }
          uint32_t ngb, ngs, ntb, ntm;
          node_step<ANY>(nq, rb, tmin, h.t, tab, ngb, ngs, ntb, ntm);
          if constexpr (GSP_PAD_NODE > 0) pad_valu<GSP_PAD_NODE>(pad_a);
          if (!group_empty<ANY>(ngs)) {  // descend: the rest of the current group waits on the stack
            if (!group_empty<ANY>(gs)) stk.push_group(gb, gs);
            gb = ngb;
            gs = ngs;
          } else if (group_empty<ANY>(gs)) {
            stk.pop_group(gb, gs);
          }
          if (!tris_empty(ntm)) {  // hit leaf children: postponed to the next leaf step
            if (tris_empty(tm)) {
              tb = ntb;
              tm = ntm;
            } else {
              tb2 = ntb;
              tm2 = ntm;
            }
          }
        }
#ifdef GSP_WAVE_PROFILE
        {
          const uint64_t lm = __ballot(on && wp_from_lds), hm = __ballot(on && !wp_from_lds);
          wp[14] += hm != 0;
          wp[15] += __popcll(hm);
          wp[16] += lm != 0;
          wp[17] += __popcll(lm);
        }
#endif
      }
      continue;
    }
    // ---- leaf step: one triangle test for every lane with a pending triangle group ---------------------
#ifdef GSP_WAVE_PROFILE
    ++wp[2];
    wp[3] += __popcll(__ballot(!tris_empty(tm)));
#endif
    // one triangle per lane per step: a group with more triangles stays pending, so short groups do not idle
    // while long ones finish and groups that arrive in between join the next step
    // Written without conditional blocks around the state: every update is a select on the variable's own register.
    // (A divergent block that assigns loop-carried variables leaves the compiler with two copies of each -- the
    // value before and after -- and the loop then moves them back and forth: 67 v_mov per leaf step, a third of
    // its instructions.)
    {
      const bool act = !tris_empty(tm);
      uint32_t tm_n = tm;
      const uint32_t slot = tris_next(tb, tm_n);  // (tm == 0: some slot number, unused)
      tm = act ? tm_n : tm;
      // the test itself runs under the lane's condition (lanes without a triangle stay switched off: their arithmetic
      // would be thrown away, and on a chip that regulates its clock by power it is not free); what leaves the block are
      // temporaries -- defined by an empty asm in the other lanes, so no copy is needed at the join -- and `hit`
      float t, u, v, aw;
      asm("" : "=v"(t), "=v"(u), "=v"(v), "=v"(aw));  // any value
      bool hit = false;
      if (act) {
        const q4* p = tris + 3ll * (IO::kVersioned ? slot + tri_base : slot);
        const q4 p0 = p[0], p1 = p[1], p2 = p[2];
        aw = p0.w;
        hit = intersect_tri_rot(mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), rb.o, rs, tmin, tmax, t, u, v);
        if constexpr (GSP_PAD_LEAF > 0) pad_valu<GSP_PAD_LEAF>(pad_a);
      }
      if (STATS) {
        c_tris += act ? 1u : 0u;
        ray_tris += act ? 1u : 0u;
        if (act && so.tri_hist) atomicAdd(so.tri_hist + slot, 1u);
      }
      uint32_t gslot = slot;  // hit slots of a split scene are counted through both trees
      if constexpr (IO::kVersioned && IO::kSplit) gslot = slot + (node_off != 0u ? io.static_slots() : 0u);
      if (ANY) {  // the first accepted triangle ends the ray
        h.t = hit ? t : h.t;
        h.slot = hit ? (int32_t)gslot : h.slot;
        gs = hit ? no_group<ANY>() : gs;
        tm = hit ? 0u : tm;
        tm2 = hit ? 0u : tm2;
      } else {
        const uint32_t id = __float_as_uint(aw);
        const bool better = hit & ((t < h.t) | ((t == h.t) & (id < best_id)));
        h.t = better ? t : h.t;
        h.u = better ? u : h.u;
        h.v = better ? v : h.v;
        h.slot = better ? (int32_t)gslot : h.slot;
        best_id = better ? id : best_id;
      }
      const bool up = act & tris_empty(tm);  // the second group, if any, moves up
      tb = up ? tb2 : tb;
      tm = up ? tm2 : tm;
      tm2 = up ? 0u : tm2;
    }
  }
#undef tmin
#undef tmax
  if constexpr (GSP_PAD_NODE + GSP_PAD_LEAF + GSP_PAD_BOOK > 0) asm volatile("" ::"v"(pad_a));
#ifdef GSP_WAVE_PROFILE
  if (lane == 0)
    for (int k = 0; k < 24; ++k) atomicAdd(&g_wave_profile[ANY ? 1 : 0][k], wp[k]);
#endif
  if (STATS) {
    const unsigned long long a = wave_sum_u64(c_nodes), b = wave_sum_u64(c_tris), c = wave_sum_u64(c_rays);
    const unsigned long long d = wave_sum_u64(c_hits), e = wave_sum_u64(c_hit_nodes), f = wave_sum_u64(c_lds), g = wave_sum_u64(c_notri);
    if (lane == 0) {
      atomicAdd(so.nodes, a);
      atomicAdd(so.tris, b);
      atomicAdd(so.rays, c);
      if (so.lds_nodes) atomicAdd(so.lds_nodes, f);
      if (so.hits) atomicAdd(so.hits, d);
      if (so.hit_nodes) atomicAdd(so.hit_nodes, e);
      if (so.no_tri) atomicAdd(so.no_tri, g);
    }
  }
}

}  // namespace gsp

