// (r02 experiment, rejected: DESIGN 4 "wave-level leaf queue".  Written against the r02 traversal (4-wide nodes with a
// sort network); kept as a record, not built.)
// pt_wavetrace_q.h -- EXPERIMENT, not part of the product build: the persistent traversal kernel of pt_wavetrace.h with a
// wave-level leaf queue.  Built only as a variant (scripts/build_variant.sh q5 "-include pt_wavetrace_q.h -DGSP_LEAF_QUEUE
// -DGSP_LDS_LEVELS=22 -DGSP_BLOCKS_PER_CU=5 -DGSP_TRACE_WAVES=5"): force-included ahead of pt_render.hip, it renames the
// kernel the launches name.  Bit-identical results (15 GPU parity tests), 15 % fewer VALU instructions, 18 % SLOWER:
// DESIGN.md section 4, profiles/r02_ab_leaf_queue.txt.
//
// pt_wavetrace.h postpones ONE leaf per lane and runs a leaf step when enough lanes hold one; a lane that meets a second
// leaf stalls, and the leaf step itself runs with ~25 of 64 lanes (profiles/r02_wave_profile.txt: 16.5 lanes per node step
// stalled on a leaf).  Here a lane that reaches a leaf appends {leaf, owner lane} to a queue of its wave in LDS and keeps
// traversing; when 64 items are queued (or nothing else can run) every lane takes ONE item -- not necessarily of its own
// ray: the owner's ray constants come through ds_bpermute -- and the hits are merged into per-lane result records in LDS
// with a 64-bit atomic min on {t bits, triangle id}, which is exactly the closest-hit rule (smallest t, ties to the smaller
// id), so the result does not depend on who tested what when.  A ray is finished when its stack is empty AND the queue
// head has passed its last item.
#pragma once
#include "pt_wavetrace.h"

namespace gsp {

#ifndef GSP_Q_CAP
#define GSP_Q_CAP 128
#endif
#ifndef GSP_Q_WAIT_LANES
#define GSP_Q_WAIT_LANES 24  // lanes that are only waiting for their queued leaves: run a (partial) batch
#endif
constexpr uint32_t kQCap = GSP_Q_CAP;

template <bool ANY, bool STATS, class IO>
__global__ __launch_bounds__(kTraceBlock, GSP_TRACE_WAVES) void k_trace_q(const q4* __restrict__ nodes, const q4* __restrict__ tris,
                                                          int32_t root, const uint32_t* __restrict__ n_ptr,
                                                          uint32_t n_imm, uint32_t chunk, IO io,
                                                          uint32_t* __restrict__ work,
                                                          int32_t* __restrict__ spill, uint32_t spill_stride,
                                                          TraceStatsOut so) {
  __shared__ int32_t lds_stack[kLdsStackDepth * kTraceBlock];
  __shared__ uint32_t s_qcode[kTraceBlock / 64][kQCap];   // leaf code (~cur)
  __shared__ uint32_t s_qowner[kTraceBlock / 64][kQCap];  // lane that owns the ray
  __shared__ unsigned long long s_key[kTraceBlock];       // running min of {t bits << 32 | triangle id} per lane
  __shared__ q4 s_hit[kTraceBlock];                       // {u, v, bits(slot), bits(aux)} of the current minimum
  const uint32_t n = n_ptr ? *n_ptr : n_imm;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = threadIdx.x >> 6;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  const uint32_t shard = blockIdx.x % kWorkShards;
  uint32_t* my_work = work + shard * kWorkStride;
  uint32_t* qcode = s_qcode[wave];
  uint32_t* qowner = s_qowner[wave];
  unsigned long long* keys = s_key + wave * 64;  // this wave's records, indexed by lane
  q4* hits = s_hit + wave * 64;

  WaveStack stk;
  stk.lds = (lds_i32*)lds_stack + threadIdx.x;
  stk.spill = (glb_i32*)spill + (size_t)blockIdx.x * kTraceBlock + threadIdx.x;
  stk.spill_stride = spill_stride;
  stk.sp = 0;

  uint32_t pool_next = 0, pool_end = 0;
  bool exhausted = false;
  uint32_t q_head = 0, q_tail = 0;  // wave-uniform sequence numbers: items [q_head, q_tail) are queued

  int32_t cur = kSentinel;
  uint32_t ri = 0xffffffffu;
  uint32_t last_seq = 0;  // q_tail right after this lane's last enqueue: outstanding while last_seq > q_head
  RayBox rb = make_raybox(mk3(0, 0, 0), mk3(1, 1, 1));
  RayShear rs;
  rs.kx = rs.ky = rs.kz = 0;
  rs.Sx = rs.Sy = rs.Sz = 0.0f;
  float tmin = 0.0f, tmax = 0.0f, tfar = 0.0f;
  uint32_t c_nodes = 0, c_tris = 0, c_rays = 0;

#ifdef GSP_Q_GUARD
  uint32_t guard = 0;
#endif
#ifdef GSP_WAVE_PROFILE
  unsigned long long wp[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  for (;;) {
#ifdef GSP_WAVE_PROFILE
    ++wp[4];
#endif
#ifdef GSP_Q_GUARD
    if (++guard > (1u << 24)) break;  // first runs of the experiment: never hang the box
#endif
    // ---- commit finished rays: stack empty and every queued leaf of the ray tested --------------------------------
    {
      const bool pending = ri != 0xffffffffu && cur == kSentinel && last_seq <= q_head;
      const uint64_t pend_m = __ballot(pending);
      if (pend_m) {
        const uint64_t out_m = pend_m | __ballot(ri == 0xffffffffu);
        if (wave_count(out_m) >= GSP_BATCH_COMMIT || out_m == ~0ull) {
          if (pending) {
            const unsigned long long k = keys[lane];
            const q4 r = hits[lane];
            HitRec h;
            h.t = __uint_as_float((uint32_t)(k >> 32));
            h.u = r.x;
            h.v = r.y;
            h.slot = (int32_t)__float_as_uint(r.z);
            io.store(ri, h, __float_as_uint(r.w));
            ri = 0xffffffffu;
          }
        }
      }
    }
    // ---- refill idle lanes from the wave-local pool ----------------------------------------------------------------
    uint64_t idle_m = __ballot(ri == 0xffffffffu);
    if (!exhausted && wave_count(idle_m) >= kRefillLanes) {
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
          rs.Sz = comp(rb.inv, rs.kz);
          tfar = tmax;
          keys[lane] = ((unsigned long long)__float_as_uint(tmax) << 32) | 0xffffffffull;
          hits[lane] = make_q4(0.0f, 0.0f, __uint_as_float(0xffffffffu), 0.0f);
          last_seq = q_head;  // nothing outstanding
          stk.sp = 0;
          stk.push(kSentinel);
          cur = root;
          if (STATS) ++c_rays;
        }
        const uint32_t want = (uint32_t)__popcll(idle_m);
        pool_next += want < avail ? want : avail;
        idle_m = __ballot(ri == 0xffffffffu);
      }
    }
    // ---- queue the leaves the lanes stand on, let those lanes move on -----------------------------------------------
    uint32_t qn = q_tail - q_head;
    {
      const bool at_leaf = cur < 0;
      const uint64_t leaf_m = __ballot(at_leaf);
      if (leaf_m != 0 && qn <= kQCap - 64u) {
        if (at_leaf) {
          const uint32_t pos = (q_tail + (uint32_t)__popcll(leaf_m & lt_mask)) % kQCap;
          qcode[pos] = (uint32_t)~cur;
          qowner[pos] = lane;
        }
        q_tail += (uint32_t)wave_count(leaf_m);
        if (at_leaf) {
          last_seq = q_tail;
          cur = stk.pop();
        }
        qn = q_tail - q_head;
      }
    }
    const bool on_node = (uint32_t)cur < (uint32_t)kSentinel;
    const uint64_t node_m = __ballot(on_node);
    const uint64_t atleaf_m = __ballot(cur < 0);
    if ((node_m | atleaf_m) == 0 && qn == 0) {
      const uint64_t live_m = __ballot(ri != 0xffffffffu);
      if (live_m != 0) continue;  // finished rays wait for the commit threshold (out_m == ~0 commits them all)
      if (exhausted || idle_m == 0) break;
      continue;
    }
    // ---- leaf batch: every lane tests one queued item ----------------------------------------------------------------
    const uint64_t wait_m = __ballot(ri != 0xffffffffu && cur == kSentinel && last_seq > q_head);
    const bool batch = qn >= 64u || (qn != 0 && (node_m == 0 || wave_count(wait_m) >= GSP_Q_WAIT_LANES || qn > kQCap - 64u));
    if (batch) {
      const uint32_t m = qn < 64u ? qn : 64u;
#ifdef GSP_WAVE_PROFILE
      ++wp[2];
      wp[3] += m;
#endif
      const bool mine = lane < m;
      const uint32_t pos = (q_head + lane) % kQCap;
      const uint32_t code = mine ? qcode[pos] : 0u;
      const uint32_t owner = mine ? qowner[pos] : lane;
      // the owner's ray, lane to lane (all 64 lanes execute the permutes)
      const float ox = __shfl(rb.o.x, (int)owner), oy = __shfl(rb.o.y, (int)owner), oz = __shfl(rb.o.z, (int)owner);
      RayShear os;
      os.Sx = __shfl(rs.Sx, (int)owner);
      os.Sy = __shfl(rs.Sy, (int)owner);
      os.Sz = __shfl(rs.Sz, (int)owner);
      const int kpack = __shfl((int)(rs.kx | (rs.ky << 2) | (rs.kz << 4)), (int)owner);
      os.kx = kpack & 3;
      os.ky = (kpack >> 2) & 3;
      os.kz = (kpack >> 4) & 3;
      const float otmin = __shfl(tmin, (int)owner), otmax = __shfl(tmax, (int)owner);
      if (mine) {
        const uint32_t first = code >> 2, count = (code & 3u) + 1u;
        for (uint32_t k = 0; k < count; ++k) {
          const q4* p = tris + 3ll * (first + k);
          const q4 p0 = p[0], p1 = p[1], p2 = p[2];
          if (STATS) ++c_tris;
          float t, u, v;
          if (intersect_tri(mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), mk3(ox, oy, oz), os, otmin, otmax, t, u,
                            v)) {
            const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | (ANY ? 0ull : (unsigned long long)__float_as_uint(p0.w));
            const unsigned long long old = atomicMin(&keys[owner], key);
            if (key < old) {
              // (two items of one owner in the same batch: the later atomic with the smaller key also writes later)
              if (keys[owner] == key) hits[owner] = make_q4(u, v, __uint_as_float(first + k), p1.w);
            }
          }
        }
      }
      q_head += m;
      // every ray's pruning distance follows its record
      if (ri != 0xffffffffu) {
        const unsigned long long k = keys[lane];
        tfar = __uint_as_float((uint32_t)(k >> 32));
        if (ANY && (uint32_t)k != 0xffffffffu) {  // occluded: stop traversing (queued items of the ray still drain)
          cur = kSentinel;
          stk.sp = WaveStack::kLevelBytes;  // only the sentinel below
        }
      }
      continue;
    }
    // ---- node steps ----------------------------------------------------------------------------------------------
    if (node_m != 0) {
      for (int rep = 0; rep < GSP_NODE_REPS; ++rep) {
        const bool on = (uint32_t)cur < (uint32_t)kSentinel;
        if (rep > 0 && wave_count(__ballot(on)) < GSP_REP_LANES) break;
#ifdef GSP_WAVE_PROFILE
        ++wp[0];
        wp[1] += __popcll(__ballot(on));
        wp[7] += __popcll(__ballot(ri == 0xffffffffu));
        wp[8] += __popcll(__ballot(ri != 0xffffffffu && !on));
#endif
        if (on) {
          const q4* nd = (const q4*)((const char*)nodes + (uint32_t)cur);
          const q4 n0 = nd[0], n1 = nd[1], n2 = nd[2], n3 = nd[3];
          if (STATS) ++c_nodes;
          constexpr uint32_t L = WaveStack::kLevelBytes;
          int32_t e0, e1, e2, e3;
          const uint32_t nb = node4_step<L>(n0, n1, n2, n3, rb, tmin, tfar, e0, e1, e2, e3);
          stk.push_sorted(nb > 0u ? nb - L : 0u, e1, e2, e3);
          if (nb > 0u) cur = e0;
          else cur = stk.pop();
        }
        // leaves reached in this step go to the queue at once (if it has room), so that the lane keeps descending
        const bool at_leaf = cur < 0;
        const uint64_t lm = __ballot(at_leaf);
        if (lm != 0) {
          if (q_tail - q_head > kQCap - 64u) break;
          if (at_leaf) {
            const uint32_t pos = (q_tail + (uint32_t)__popcll(lm & lt_mask)) % kQCap;
            qcode[pos] = (uint32_t)~cur;
            qowner[pos] = lane;
          }
          q_tail += (uint32_t)wave_count(lm);
          if (at_leaf) {
            last_seq = q_tail;
            cur = stk.pop();
          }
          if (q_tail - q_head >= 64u) break;  // a full batch is ready
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

#ifdef GSP_LEAF_QUEUE
#define k_trace k_trace_q  // pt_render.hip's launches (this header is force-included before it)
#endif
