// scripts/experiments/r06_lq_proxy.h -- MEASUREMENT BUILD ONLY (-DGSP_LQ_PROXY=1|2|3, with -DGSP_LDS_LEVELS=12 so that the block
// still fits seven to a CU).  Not a working leaf queue: the COST side of one.
//
// A wave-level leaf queue (r02, scripts/experiments/r02_pt_wavetrace_q.h; r05 review, item 4) lets every lane test one queued
// {owner, triangle} item, so leaf steps run with ~63 lanes instead of ~28 and lanes never wait for a leaf step.  What it saves is
// bounded by profiles/r06_trace_phase_budget.txt (instruction floors x the measured worth of an issue slot).  What it COSTS is
// the machinery below, which this header bolts onto the product kernel without changing a result:
//   per node step  (bit 1): the hand-over of the step's triangle group into the wave's LDS queue: ballot, rank (2 x mbcnt), ds_write_b64
//   per leaf step  (bit 2): the item's read (ds_read_b64), the owner's ray through ds_bpermute (origin, shear, masks: 9 words) and its
//                           current t from the LDS record, the merge of an accepted hit with ds_min_u64 on {t bits, id}, the re-read that
//                           tells the winner, its lane number into the owner's byte, and the owner's pull of u, v, slot (3 x ds_bpermute)
// The gathered values FEED the triangle test (or-ed in under an opaque zero), so the LDS round trips sit on the step's dependent
// chain as they would in the real thing.  In the real queue a leaf step runs 2.3 x less often than here (4.1 instead of 9.4 per 64
// rays) and a node step 0.8 x as often: scale the measured cost accordingly (scripts/trace_phase_budget.py does).
// HOOKS (removed from the product after the measurement; live in commit 62b6528): pt_wavetrace.h includes this file when GSP_LQ_PROXY is
// set and calls LQ_PROXY_SHARED behind the loop's locals, LQ_PROXY_NODE behind node_step and LQ_PROXY_LEAF_GATHER / _MERGE around the
// triangle test.  RESULT: profiles/r06_ab_leaf_queue_proxy.txt.
#pragma once

#define LQ_PROXY_SHARED                                                            \
  __shared__ unsigned long long lq_rec[kTraceBlock];                               \
  __shared__ unsigned long long lq_queue[kTraceBlock / 64][64];                    \
  __shared__ uint8_t lq_win[kTraceBlock];                                          \
  lq_rec[threadIdx.x] = ~0ull;                                                     \
  lq_win[threadIdx.x] = 0xffu;                                                     \
  uint32_t lq_tail = 0, lq_zero;                                                   \
  asm volatile("v_mov_b32 %0, 0" : "=v"(lq_zero));

// node step: this lane's new triangle group {ntb, ntm} goes to the queue's tail
#define LQ_PROXY_NODE(ntb, ntm)                                                                                                     \
  if constexpr ((GSP_LQ_PROXY & 1) != 0) {                                                                                          \
    const uint64_t pm = __ballot(!tris_empty(ntm));                                                                                 \
    const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));              \
    if (!tris_empty(ntm)) lq_queue[threadIdx.x >> 6][(lq_tail + rk) & 63u] = ((unsigned long long)(ntb) << 32) | ((lane << 4) | (ntm)); \
    lq_tail += (uint32_t)__popcll(pm);                                                                                              \
  }

// leaf step, in front of the test: the item, the owner's ray and current t; the test then runs on o_ / rs_ / tmax_ (= the lane's own
// values: everything gathered is and-ed with the opaque zero)
#define LQ_PROXY_LEAF_GATHER(o_, rs_, tmax_)                                                                                        \
  if constexpr ((GSP_LQ_PROXY & 2) != 0) {                                                                                          \
    const unsigned long long it = lq_queue[threadIdx.x >> 6][lane];                                                                 \
    const int own = (int)(((uint32_t)it >> 4) & 63u);                                                                               \
    const uint32_t g = (__float_as_uint(__shfl(rb.o.x, own)) ^ __float_as_uint(__shfl(rb.o.y, own)) ^ __float_as_uint(__shfl(rb.o.z, own)) ^ \
                        __float_as_uint(__shfl(rs.Sx, own)) ^ __float_as_uint(__shfl(rs.Sy, own)) ^ __float_as_uint(__shfl(rs.Sz, own)) ^   \
                        __shfl(rs.m0, own) ^ __shfl(rs.m1, own) ^ __shfl(rs.ms, own) ^ (uint32_t)(lq_rec[(threadIdx.x & ~63u) + own] >> 32) ^ (uint32_t)(it >> 32)) & lq_zero; \
    o_.x = __uint_as_float(__float_as_uint(o_.x) | g);                                                                              \
    rs_.Sx = __uint_as_float(__float_as_uint(rs_.Sx) | g);                                                                          \
    tmax_ = __uint_as_float(__float_as_uint(tmax_) | g);                                                                            \
    lq_own = own;                                                                                                                   \
  }

// leaf step, behind the test: merge, find the winner, the owner's pull.  `sink` receives the pulled words under the opaque zero.
#define LQ_PROXY_LEAF_MERGE(hit, t, u, v, id, slot, sink)                                                                           \
  if constexpr ((GSP_LQ_PROXY & 2) != 0) {                                                                                          \
    const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | (id);                                           \
    unsigned long long* rec = &lq_rec[(threadIdx.x & ~63u) + lq_own];                                                               \
    if (hit) atomicMin(rec, key | ~(unsigned long long)lq_zero);  /* (never smaller than the record: nothing changes) */            \
    if (hit && *(volatile unsigned long long*)rec == key) lq_win[(threadIdx.x & ~63u) + lq_own] = (uint8_t)lane;                    \
    const uint32_t w = ((volatile uint8_t*)lq_win)[threadIdx.x];                                                                    \
    const int wl = (int)(w & 63u);                                                                                                  \
    const uint32_t pull = (__float_as_uint(__shfl(u, wl)) ^ __float_as_uint(__shfl(v, wl)) ^ (uint32_t)__shfl((int)(slot), wl)) & lq_zero; \
    sink = __uint_as_float(__float_as_uint(sink) | (w == 0xffu ? 0u : pull));                                                        \
  }
