// scripts/experiments/r06_pt_wavetrace_lq.h -- EXPERIMENT (r06, -DGSP_LEAFQ=1|2|3 with -DGSP_LDS_LEVELS=13): the persistent traversal
// kernel of pt_wavetrace.h with a WAVE-LEVEL LEAF QUEUE (r05 review, item 4; first tried in r02 on the binary tree,
// scripts/experiments/r02_pt_wavetrace_q.h, and lost there).  Included by pt_wavetrace.h when GSP_LEAFQ is set; bit 1 = the closest-hit
// launches of the plain ray source (ExtendIO), bit 2 = the any-hit ones (ConnectIO).
//
// What changes against k_trace.  A lane no longer keeps the triangle groups its node steps produce ({tb, tm}, {tb2, tm2}) and no
// longer waits for a leaf step when both are taken: a node step PUSHES the group {owner lane, tb, tm} onto a 128-entry ring in LDS
// that the wave shares, and goes on.  A leaf step runs when 64 groups are queued (or nothing else can run): lane i takes entry i,
// tests the group's first triangle FOR ITS OWNER -- the owner's ray (origin, shear, axis masks: 9 words) comes through
// ds_bpermute -- and re-queues the rest of the group.  Results: closest hit -> ds_min_u64 on the owner's LDS record {t bits, id}
// (the oracle's rule: smaller t, ties to the smaller id = one unsigned 64-bit compare), the lane whose key the record then holds
// writes its number into the owner's byte and the owner pulls u, v, slot from it (3 x ds_bpermute); any hit -> a store into the
// owner's record.  Every test adds 1 to the owner's `done` word; a ray is complete when its traversal is over and done == pushed.
// Same hits as k_trace (a stale, larger t only admits more candidates; the record keeps the minimum).
//
// Budget: 6 waves per SIMD (the leaf step holds the lane's own ray AND the owner's: ~80 VGPRs; at 72 it spills, r06_lq_proxy.h),
// LDS per block 13 stack levels (13 KB) + step table 2 KB + top of the tree 4 KB + 4 rings x 1 KB + records 2 KB + done 1 KB +
// winner bytes 256 B = 26 880 B: six blocks per CU.
// HOOKS (removed from the product after the measurement; live in commit 62b6528): pt_wavetrace.h includes this file behind k_trace when
// GSP_LEAFQ is set, and pt_render_pipeline.inc launches k_trace_lq<false, ExtendIO> / k_trace_lq<true, ConnectIO> (grid capped at
// num_cus x GSP_LQ_WAVES) in place of the plain k_trace launches for bits 1 / 2.  RESULT: bit-exact, +23 % / +28 % kernel time:
// profiles/r06_ab_leaf_queue.txt.
#pragma once

namespace gsp {

constexpr uint32_t kLqEntries = 128;  // queued groups per wave (a node step may add 64)
#ifndef GSP_LQ_WAVES
#define GSP_LQ_WAVES 6
#endif
#ifndef GSP_LQ_MIN_NODE_LANES
#define GSP_LQ_MIN_NODE_LANES 16  // fewer lanes than this able to take a node step + something queued: run the leaf step first
#endif
#ifndef GSP_LQ_BATCH
#define GSP_LQ_BATCH 64  // queued groups that trigger a leaf step (a step takes at most 64)
#endif
#ifndef GSP_LQ_WAIT_LANES
#define GSP_LQ_WAIT_LANES 65  // this many lanes waiting for their queued tests + something queued: leaf step (65 = off)
#endif
#ifndef GSP_LQ_COMMIT
#define GSP_LQ_COMMIT (ANY ? 32 : 24)
#endif

template <bool ANY, class IO>
__global__ __launch_bounds__(kTraceBlock, GSP_LQ_WAVES) void k_trace_lq(const q4* __restrict__ nodes, const q4* __restrict__ tris,
                                                                         const uint32_t* __restrict__ n_ptr, uint32_t n_imm, uint32_t first, uint32_t chunk,
                                                                         IO io, uint32_t* __restrict__ work, uint32_t* __restrict__ spill,
                                                                         uint32_t spill_stride) {
  static_assert(!IO::kVersioned, "the leaf-queue experiment covers the plain ray sources");
  __shared__ uint32_t lds_stack[kLdsStackDepth * kStackWords * kTraceBlock];
  __shared__ uint32_t lds_table[kStepTableBytes / 4];
  __shared__ q4 lds_top[kTopNodes > 0 ? kTopNodes * kNodeQuads : 1];
  __shared__ unsigned long long lq_q[kTraceBlock / 64][kLqEntries];
  __shared__ unsigned long long lq_rec[kTraceBlock];
  __shared__ uint32_t lq_done[kTraceBlock];
  __shared__ uint8_t lq_win[kTraceBlock];
  static_assert((sizeof(lds_stack) + sizeof(lds_table) + sizeof(lds_top) + sizeof(lq_q) + sizeof(lq_rec) + sizeof(lq_done) + sizeof(lq_win)) * GSP_LQ_WAVES <= 160 * 1024,
                "LDS per block x resident blocks per CU exceeds 160 KB");
  stage_step_table(lds_table, threadIdx.x, kTraceBlock);
  for (uint32_t i = threadIdx.x; i < kTopNodes * kNodeQuads; i += kTraceBlock) lds_top[i] = nodes[i];
  lq_done[threadIdx.x] = 0u;
  lq_win[threadIdx.x] = 0xffu;
  __syncthreads();
  const LdsStepTable tab{(const __attribute__((address_space(3))) char*)lds_table};
  const uint32_t n = n_ptr ? *n_ptr : n_imm;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wbase = threadIdx.x & ~63u;
  unsigned long long* const myq = lq_q[threadIdx.x >> 6];
  const uint32_t shard = blockIdx.x % kWorkShards;
  uint32_t* my_work = work + shard * kWorkStride;

  WaveStack stk;
  stk.lds = (lds_u32*)lds_stack + threadIdx.x;
  stk.spill = (glb_u32*)spill + (size_t)blockIdx.x * kTraceBlock + threadIdx.x;
  stk.spill_stride = spill_stride;
  stk.sp = 0;

  uint32_t pool_next = 0, pool_end = 0;
  bool exhausted = false;
  uint32_t q_head = 0, q_tail = 0;  // wave-uniform, monotonic

  uint32_t gb = 0, gs = no_group<ANY>();
  uint32_t ri = 0xffffffffu, pay = 0, pushed = 0;
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
  // the empty record of a ray: no hit yet (closest: t = tmax, id = all ones; any: the same word means "not occluded")
  const auto empty_rec = [&](float tm_) { return ((unsigned long long)__float_as_uint(tm_) << 32) | 0xffffffffull; };

#ifdef GSP_WAVE_PROFILE
  unsigned long long wp[24] = {};  // [0] node steps [1] lanes on [2] leaf steps [3] lanes [4] loop passes [7] lanes without a ray [8] lanes waiting for queued tests
#endif
  for (;;) {
#ifdef GSP_WAVE_PROFILE
    ++wp[4];
#endif
    // ---- commit: traversal over AND every queued triangle of the ray tested -----------------------------------------
    uint32_t waiting = 0;  // lanes whose traversal is over and whose queued triangles are not all tested yet
    {
      const bool fin = ri != 0xffffffffu && group_empty<ANY>(gs);
      bool complete = false;
      if (__ballot(fin)) complete = fin && *(volatile uint32_t*)&lq_done[threadIdx.x] == pushed;
      const uint64_t pend_m = __ballot(complete);
      waiting = (uint32_t)wave_count(__ballot(fin && !complete));
      if (pend_m) {
        const uint64_t out_m = pend_m | __ballot(ri == 0xffffffffu);
        if (wave_count(out_m) >= GSP_LQ_COMMIT || out_m == ~0ull) {
          if (complete) {
            const unsigned long long rec = *(volatile unsigned long long*)&lq_rec[threadIdx.x];
            uint32_t aux = 0;
            if (ANY) {
              h.slot = rec == empty_rec(tmax) ? -1 : 0;
            } else {
              h.t = __uint_as_float((uint32_t)(rec >> 32));
              aux = (uint32_t)rec & 7u;  // (h.u, h.v, h.slot: pulled from the winners as they came)
            }
            io.store(ri, h, aux, pay);
            ri = 0xffffffffu;
          }
        }
      }
    }
    // ---- refill idle lanes from the wave-local pool (as k_trace) ----------------------------------------------------------
    uint64_t idle_m = __ballot(ri == 0xffffffffu);
    if (!exhausted && wave_count(idle_m) >= kRefillLanes) {
      while (idle_m) {
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
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle_m, 0u));
        const uint32_t avail = pool_end - pool_next;
        if (((idle_m >> lane) & 1ull) && rank < avail) {
          ri = pool_next + rank;
          f3 d;
          f3 o;
          io.load(ri, o, d, tmin_v, tmax_v, pay);
          rb = make_raybox(o, d);
          rs = make_shear_rot(d);
          rs.Sz = permute_axes(rs, rb.inv).z;
          h.t = tmax;
          h.u = h.v = 0.0f;
          h.slot = -1;
          pushed = 0u;
          lq_done[threadIdx.x] = 0u;
          lq_rec[threadIdx.x] = empty_rec(tmax);
          stk.sp = 0;
          stk.push_group(0u, no_group<ANY>());
          gb = 0u;
          gs = root_group<ANY>();
        }
        const uint32_t want = (uint32_t)__popcll(idle_m);
        pool_next += want < avail ? want : avail;
        idle_m = __ballot(ri == 0xffffffffu);
      }
    }
    // ---- what can run? ------------------------------------------------------------------------------------------------------
    const uint32_t q_count = q_tail - q_head;
    const uint64_t node_m = __ballot(ri != 0xffffffffu && !group_empty<ANY>(gs));
    if (node_m == 0 && q_count == 0) {
      if (exhausted || idle_m == 0) break;
      continue;
    }
    const bool leaf_step = q_count >= (uint32_t)GSP_LQ_BATCH || node_m == 0 ||
                           (q_count != 0u && (wave_count(node_m) < GSP_LQ_MIN_NODE_LANES || waiting >= (uint32_t)GSP_LQ_WAIT_LANES));
    if (!leaf_step) {
      for (int rep = 0; rep < GSP_NODE_REPS; ++rep) {
        const bool on = ri != 0xffffffffu && !group_empty<ANY>(gs);
        if (rep > 0 && (wave_count(__ballot(on)) < GSP_REP_LANES || q_tail - q_head > kLqEntries - 64u)) break;
#ifdef GSP_WAVE_PROFILE
        ++wp[0];
        wp[1] += __popcll(__ballot(on));
        wp[7] += __popcll(__ballot(ri == 0xffffffffu));
        wp[8] += __popcll(__ballot(ri != 0xffffffffu && !on));
#endif
        uint32_t ntb = 0, ntm = 0;
        if (on) {
          const uint32_t noff = group_next<ANY>(gb, gs, rb, tab);
          q4 nq[kNodeQuads];
          if (kTopNodes > 0 && noff < kTopNodes * kNodeBytes) {
            const lds_v4f* nd = (const lds_v4f*)((const __attribute__((address_space(3))) char*)lds_top + noff);
#pragma unroll
            for (uint32_t k = 0; k < kNodeQuads; ++k) {
              const v4f_t q = nd[k];
              nq[k] = make_q4(q.x, q.y, q.z, q.w);
            }
          } else {
            const q4* nd = (const q4*)((const char*)nodes + noff);
#pragma unroll
            for (uint32_t k = 0; k < kNodeQuads; ++k) nq[k] = nd[k];
          }
          uint32_t ngb, ngs;
          node_step<ANY>(nq, rb, tmin, h.t, tab, ngb, ngs, ntb, ntm);
          if (!group_empty<ANY>(ngs)) {
            if (!group_empty<ANY>(gs)) stk.push_group(gb, gs);
            gb = ngb;
            gs = ngs;
          } else if (group_empty<ANY>(gs)) {
            stk.pop_group(gb, gs);
          }
        }
        // the step's hit leaf children go to the wave's queue
        const uint64_t pm = __ballot(!tris_empty(ntm));
        if (pm) {
          const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
          if (!tris_empty(ntm)) {
            myq[(q_tail + rk) & (kLqEntries - 1u)] = ((unsigned long long)ntb << 32) | ((lane << 4) | ntm);
            pushed += (uint32_t)__builtin_popcount(ntm);
          }
          q_tail += (uint32_t)wave_count(pm);
        }
      }
      continue;
    }
    // ---- leaf step: lane i tests the first triangle of queued group i for its owner -----------------------------------------
    {
      const uint32_t nb = q_count < 64u ? q_count : 64u;
      const bool act = lane < nb;
#ifdef GSP_WAVE_PROFILE
      ++wp[2];
      wp[3] += nb;
#endif
      unsigned long long e = 0;
      if (act) e = myq[(q_head + lane) & (kLqEntries - 1u)];
      q_head += nb;
      const uint32_t elo = (uint32_t)e, tb = (uint32_t)(e >> 32);
      const int own = (int)((elo >> 4) & 63u);
      const uint32_t tm_ = elo & 15u;
      const uint32_t slot = tb + (uint32_t)__builtin_ctz(tm_ | 16u);
      const uint32_t rem = tm_ & (tm_ - 1u);
      {  // the rest of the group waits for a later step
        const uint64_t rm = __ballot(act && rem != 0u);
        if (rm) {
          const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(rm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)rm, 0u));
          if (act && rem != 0u) myq[(q_tail + rk) & (kLqEntries - 1u)] = ((unsigned long long)tb << 32) | (((uint32_t)own << 4) | rem);
          q_tail += (uint32_t)wave_count(rm);
        }
      }
      // the owner's ray
      const f3 oo = mk3(__shfl(rb.o.x, own), __shfl(rb.o.y, own), __shfl(rb.o.z, own));
      RayShearRot os;
      os.m0 = __shfl(rs.m0, own), os.m1 = __shfl(rs.m1, own), os.ms = __shfl(rs.ms, own);
      os.Sx = __shfl(rs.Sx, own), os.Sy = __shfl(rs.Sy, own), os.Sz = __shfl(rs.Sz, own);
      float otmax = IO::kTmax >= 0.0f ? IO::kTmax : __shfl(tmax_v, own);
      float t, u, v, aw;
      asm("" : "=v"(t), "=v"(u), "=v"(v), "=v"(aw));
      bool hit = false;
      if (act) {
        const q4* p = tris + 3ll * slot;
        const q4 p0 = p[0], p1 = p[1], p2 = p[2];
        aw = p0.w;
        hit = intersect_tri_rot(mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), oo, os, tmin, otmax, t, u, v);
      }
      const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | __float_as_uint(aw);
      unsigned long long* const orec = &lq_rec[wbase + (uint32_t)own];
      if (ANY) {
        if (hit) *(volatile unsigned long long*)orec = 0ull;  // any accepted triangle: the ray is occluded
      } else {
        if (hit) atomicMin(orec, key);
      }
      if (act) atomicAdd(&lq_done[wbase + (uint32_t)own], 1u);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      if (__ballot(hit) == 0) {
        // nothing accepted in this batch: no record changed
      } else if (ANY) {
        if (ri != 0xffffffffu && *(volatile unsigned long long*)&lq_rec[threadIdx.x] != empty_rec(tmax)) gs = no_group<ANY>();  // occluded: the traversal ends
      } else {
        if (hit && *(volatile unsigned long long*)orec == key) ((volatile uint8_t*)lq_win)[wbase + (uint32_t)own] = (uint8_t)lane;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        const uint32_t w = ((volatile uint8_t*)lq_win)[threadIdx.x];
        const int wl = (int)(w & 63u);
        const float nu = __shfl(u, wl), nv = __shfl(v, wl);
        const int ns = __shfl((int)slot, wl);
        const bool got = w != 0xffu;
        h.u = got ? nu : h.u;
        h.v = got ? nv : h.v;
        h.slot = got ? ns : h.slot;
        if (got) ((volatile uint8_t*)lq_win)[threadIdx.x] = 0xffu;
        h.t = __uint_as_float((uint32_t)(*(volatile unsigned long long*)&lq_rec[threadIdx.x] >> 32));  // the far bound of the node tests
      }
    }
  }
#undef tmin
#undef tmax
#ifdef GSP_WAVE_PROFILE
  if (lane == 0)
    for (int k = 0; k < 24; ++k) atomicAdd(&g_wave_profile[ANY ? 1 : 0][k], wp[k]);
#endif
}

}  // namespace gsp
