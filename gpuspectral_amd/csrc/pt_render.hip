// pt_render.hip -- wavefront path tracer for gfx950 (MI355X) behind the C ABI
// of include/gpuspectral_pt.h.
//
// One render pass handles K consecutive timestamps of every owned pixel
// (K * num_pixels paths).  Per bounce, three kernels run over dense queues in HBM:
//
//   extend  : persistent wave64 state machine over the compressed wide BVH (pt_wavetrace.h; per-lane
//             stack of node groups in LDS), writes a 16-B hit record  (traceRayEXT, raygen.rgen:53-58)
//   shade   : one shading vertex per lane (rayhit.rchit:666-797 + the raygen
//             bookkeeping of raygen.rgen:59-80); each 256-path tile is counting-sorted by BSDF type in
//             LDS, survivors are compacted into the next queue with a wave64 ballot + an LDS scan +
//             one atomic pair per tile; paths that need next-event estimation emit a 48-B
//             shadow-queue record and their unoccluded outcome at once (optimistic commit)
//   connect : any-hit traversal of the shadow queue, adds the bounce's emitted
//             radiance to the sample and sets the continuing path's MIS weight
//                                                (rayhit.rchit:737-757)
//
// and one `resolve` kernel per pass folds the K samples of each pixel into the
// RGBA32F accumulate buffer in timestamp order (raygen.rgen:84-108).
//
// Queue records (SoA of 16-B quads, coalesced 1 KiB per wave-load):
//   P0 = {o.x, o.y, o.z, d.x}   P1 = {d.y, d.z, bits(seed), bits(sid)}
//   P2 = {w.r, w.g, w.b, directWeight}   P3 = {sum.r, sum.g, sum.b, bits(flags)}
//   HIT = {t, u, v, bits(slot)}
//   S0 = {o.xyz, tmax}  S1 = {d.xyz, bits(next)}
//   S3 = {sum if occluded .rgb, bits(flags of the continuing path | sample id of a path that ended)}
// `sum` = the radiance the sample has collected so far (raygen.rgen:60-63 `result`).  It TRAVELS WITH THE PATH (r03): a
// bounce adds to the copy it read with its path record and hands the new value to the continuing path's record, and
// only the bounce that ends the path stores it in the sample-result ring.  Until r02 every bounce read-modify-wrote the
// ring entry of its sample instead -- a scattered 16-B load on the dependent chain of k_trace<ConnectIO>'s commit
// (three loads -> load -> add -> store), 26 % of that kernel's time (profiles/r03_ab_connect_ablation.txt).  The
// additions per sample and their order are the same, so the images are.  k_shade forms BOTH outcomes of a bounce that
// traces a shadow ray with connect_vertex itself.  r04: the UNOCCLUDED outcome -- what happens to 80-93 % of the shadow rays
// -- is written at once where it belongs (the continuing path's P3 and MIS weight, or the sample-result ring), and only the
// occluded outcome travels in the shadow record: the commit of an unoccluded ray is nothing, that of an occluded one a 16-B
// load and the stores that overwrite the optimistic values (any-hit kernel -5 ... -21 %, profiles/r04_ab_optimistic_nee.txt).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <vector>

#include "pt_hostmath.h"
#include "pt_internal.h"
#include "pt_versions.h"
#include "pt_wavetrace.h"

namespace gsp {

namespace {

constexpr int kBlock = 256;

struct PathQueue {
  q4* P0;
  q4* P1;
  q4* P2;
  q4* P3;
};
struct ShadowQueue {
  q4* S0;
  q4* S1;
  q4* S3;  // (S2, the unoccluded outcome, left the record in r04: k_shade writes it where it belongs)
};

// Device counter words of a pipeline lane.  Two TAIL SETS of 32 words (one 128-B line each), used alternately by
// successive iterations: iteration i appends to the queues whose tails live in set i & 1 and reads its own input size from
// set (i - 1) & 1, so the host can queue iteration i + 1 before it has seen iteration i's counters.  Then the live-path
// counters of the sample slots, then the ray hand-out counters of the extend and connect launches (kWorkShards words each,
// on separate 128-B lines; one set: launches of one stream run in order).  Words [0, C_READBACK) travel to the host once
// per iteration.
// kMaxSlots: a slot of the sample-result ring is held from the injection of its batch until the batch's LAST path has
// ended (up to 52 bounces later) although ~95 % of its paths end within a few bounces, so the number of slots -- not
// the path pool -- bounds the paths in flight on scenes with short paths: with 64 slots the reference's coffee scene
// (4.2 rays per sample) ran 3 M-path launches in a 32 M-path pool (profiles/r02_scene_probe.txt).
constexpr int kMaxSlots = 1024;
constexpr int kTailSet = 32;  // words per tail set
constexpr uint32_t kFillerSid = 0xffffffffu;  // sample-id word of a queue record that is no path (k_shade's chunked reservation)
enum { T_NEXT = 0, T_SHADOW = 1, T_FIN_EXT = 2, T_FIN_SH = 4,  // within a tail set (the two 64-bit k_finish totals are 8-byte aligned)
       T_HOLES_NEXT = 6, T_HOLES_SHADOW = 7 };                   // filler records behind the blocks' last chunks (chunked reservation)
enum { C_LIVE = 2 * kTailSet, C_READBACK = C_LIVE + kMaxSlots,
       C_WORK_EXT = ((C_READBACK + 31) / 32) * 32, C_WORK_SH = C_WORK_EXT + kWorkShards * kWorkStride,
       C_COUNT = C_WORK_SH + kWorkShards * kWorkStride };
#ifdef GSP_SHADE_PROFILE
__device__ unsigned long long g_shade_profile[PR_COUNT * 4];
#endif
struct DevStats {
  unsigned long long shaded, nodes, tris, stat_rays, sh_nodes, sh_tris, sh_rays, sh_occluded, sh_occluded_nodes, lds_nodes, sh_lds_nodes, sh_no_tri;
};

__device__ __forceinline__ q4 mkq(float x, float y, float z, float w) {
  q4 r;
  r.x = x;
  r.y = y;
  r.z = z;
  r.w = w;
  return r;
}
// Queue records are written once and read once per iteration, ~7 GB per iteration through a 256-MiB Infinity Cache that
// would otherwise hold the BVH, the triangle packets and the shading packets (140 MB for a million triangles): the
// streaming accesses of k_generate and k_shade carry the non-temporal hint (A/B: k_shade -3.7 %, bench +1.1 %; on the
// ray loads / hit stores of k_trace the hint costs 1 %, so those stay plain; profiles/r02_ab_nt_queues.txt).
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ q4 qld(const q4* p) {
  const v4f v = __builtin_nontemporal_load((const v4f*)p);
  q4 r;
  r.x = v.x;
  r.y = v.y;
  r.z = v.z;
  r.w = v.w;
  return r;
}
__device__ __forceinline__ void qst(q4* p, q4 a) {
  const v4f v = {a.x, a.y, a.z, a.w};
  __builtin_nontemporal_store(v, (v4f*)p);
}
__device__ __forceinline__ uint32_t qld(const uint32_t* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void qst(uint32_t* p, uint32_t a) { __builtin_nontemporal_store(a, p); }
__device__ __forceinline__ float ub(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t fb(float f) { return __float_as_uint(f); }

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---- generate ------------------------------------------------------------------
// Appends K timestamps x num_pixels new paths to the queue at `offset`; their sample slots start
// at `sid_base` in the result ring.
__global__ __launch_bounds__(kBlock) void k_generate(RenderConsts rc, uint32_t num_pixels, uint32_t K,
                                                      uint32_t first_timestamp,
                                                      const uint32_t* __restrict__ pixel_ids, PathQueue q,
                                                      uint32_t offset, uint32_t sid_base, const q4* __restrict__ memo,
                                                      q4* __restrict__ hits, uint32_t lane, uint32_t lanes, uint32_t ver_bits) {
  // num_pixels = pixels of this pipeline lane: owned pixel lp * lanes + lane for lp in [0, num_pixels)
  const uint64_t total = (uint64_t)num_pixels * K;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (uint64_t)gridDim.x * kBlock) {
    const uint32_t k = (uint32_t)(i / num_pixels), lp = (uint32_t)(i % num_pixels) * lanes + lane;
    const uint32_t gid = pixel_ids ? pixel_ids[lp] : lp;
    const uint32_t sid = sid_base + (uint32_t)i;
    PathState p;
    generate_path(rc, gid, first_timestamp + k, sid, p);
    p.flags |= ver_bits;  // the version of the BSDF / light tables this sample belongs to (pt_stages.h kVerMask)
    const uint64_t j = offset + i;
    qst(&q.P0[j], mkq(p.o.x, p.o.y, p.o.z, p.d.x));
    qst(&q.P1[j], mkq(p.d.y, p.d.z, ub(p.seed), ub(p.sid)));
    qst(&q.P2[j], mkq(p.weight.x, p.weight.y, p.weight.z, p.directWeight));
    qst(&q.P3[j], mkq(0.0f, 0.0f, 0.0f, ub(p.flags)));  // (the ring entry is written once, by the bounce that ends the path)
    // primary-hit memo: the camera ray of a pixel is the same for every sample (no jitter, raygen.rgen:31-38), so its
    // hit record is copied instead of traced again; the extend launch skips these leading entries of the queue
    if (memo) qst(&hits[j], memo[i % num_pixels]);
  }
}

// ---- extend / connect / test hook: ray sources and result sinks of k_trace -----------------
struct ExtendIO {  // raygen.rgen:53-58: tmin 0, tmax 1e10, closest hit
  static constexpr float kTmin = 0.0f, kTmax = 1e10f;
  static constexpr bool kVersioned = false, kSplit = false;
  PathQueue q;
  q4* hits;
  __device__ __forceinline__ void load(uint32_t i, f3& o, f3& d, float& tmin, float& tmax, uint32_t& pay) const {
    const q4 p0 = q.P0[i], p1 = q.P1[i];
    o = mk3(p0.x, p0.y, p0.z);
    d = mk3(p0.w, p1.x, p1.y);
    tmin = 0.0f;
    tmax = 1e10f;
  }
  // hit word: slot in bits 0..27, BSDF type of the hit triangle in bits 28..30; miss = all ones
  __device__ __forceinline__ void store(uint32_t i, const HitRec& h, uint32_t aux, uint32_t) const {
    hits[i] = mkq(h.t, h.u, h.v, ub(h.slot < 0 ? 0xffffffffu : ((uint32_t)h.slot | ((aux & 7u) << 28))));
  }
};

// Primary-hit memo (r03): the reference shoots the SAME camera ray for every sample of a pixel -- the sub-pixel jitter is
// commented out, raygen.rgen:38, only the RNG seed depends on the timestamp -- so the depth-0 hit of each owned pixel is
// traced once per frame (this source: ray i = camera ray of owned pixel i, hit record in ExtendIO's format) and copied
// into the hit queue by k_generate for every later sample.  Bit-exact by construction: the closest-hit rule does not
// depend on the traversal.  gsp_stats.memoised_rays counts the path segments answered from the memo: they are not
// traced rays (bench.py's Mrays/s leaves them out).
struct MemoIO {
  static constexpr float kTmin = 0.0f, kTmax = 1e10f;
  static constexpr bool kVersioned = false, kSplit = false;
  RenderConsts rc;
  const uint32_t* pixel_ids;
  uint32_t lane, lanes;
  q4* memo;
  __device__ __forceinline__ void load(uint32_t i, f3& o, f3& d, float& tmin, float& tmax, uint32_t& pay) const {
    const uint32_t lp = i * lanes + lane;
    PathState p;
    generate_path(rc, pixel_ids ? pixel_ids[lp] : lp, 0u, 0u, p);
    o = p.o;
    d = p.d;
    tmin = 0.0f;
    tmax = 1e10f;
  }
  __device__ __forceinline__ void store(uint32_t i, const HitRec& h, uint32_t aux, uint32_t) const {
    memo[i] = mkq(h.t, h.u, h.v, ub(h.slot < 0 ? 0xffffffffu : ((uint32_t)h.slot | ((aux & 7u) << 28))));
  }
};

struct ConnectIO {  // rayhit.rchit:737-757: tmin 0.01, tmax Ldist - 0.01, any hit
  static constexpr float kTmin = 0.01f, kTmax = -1.0f;
  static constexpr bool kVersioned = false, kSplit = false;
  ShadowQueue sq;
  q4* next_P2;
  q4* next_P3;
  q4* result;
  float clampv;
  __device__ __forceinline__ void load(uint32_t i, f3& o, f3& d, float& tmin, float& tmax, uint32_t& pay) const {
    const q4 s0 = sq.S0[i], s1 = sq.S1[i];
    o = mk3(s0.x, s0.y, s0.z);
    d = mk3(s1.x, s1.y, s1.z);
    tmin = 0.01f;
    tmax = s0.w;
    pay = fb(s1.w);  // index of the continuing path in the next queue, or none: the commit needs it
  }
  // k_shade has already written the outcome of the UNOCCLUDED verdict where it belongs -- the continuing path's P3 (sum | flags)
  // and P2.w (the MIS weight of rayhit.rchit:785-787), or the sample-result ring for a path that ended at this vertex -- because
  // that is what happens to 80-93 % of the shadow rays (profiles/r04_scene_probe_occlusion.txt).  Only an OCCLUDED ray has work
  // left: one 16-B load of the other outcome and the stores that overwrite the optimistic ones (same stream, later kernel).
  // The values are the ones connect_vertex computed in k_shade either way: nothing is recomputed here.
  __device__ __forceinline__ void store(uint32_t i, const HitRec& h, uint32_t, uint32_t nx) const {
    if (h.slot < 0) return;
    q4 res = sq.S3[i];  // {sum if occluded .rgb, flags of the continuing path | sample id of a path that ended}
    if (nx != 0xffffffffu) {
      next_P3[nx] = res;
      next_P2[nx].w = 1.0f;  // no NEE happened: directWeight stays 1 (rayhit.rchit:788-790)
    } else {
      const uint32_t sid = fb(res.w);
      res.w = 0.0f;
      result[sid] = res;
    }
  }
};

// The same two sources while samples of several GEOMETRY versions are in flight (gsp_update_instances without a drain,
// pt_stages.h): a path's stamp sits in its flags word; a shadow ray finds it in the flags of the continuing path (S3.w) or, when
// the path ended at the vertex, in the word that would have named it.  `nodes` / `tris` of the launch are slot 0 of the ring.
struct GeoRing {
  uint32_t geo;           // pt_stages.h pack_geo
  uint32_t top_off;       // node-record byte offset of the tree whose top the blocks stage into LDS
  uint32_t static_slots;  // split scene: slots of the static tree in front of the ring (0: the ring holds whole trees)
  __device__ __forceinline__ void offsets(uint32_t stamp, uint32_t& node_off, uint32_t& tri_base) const {
    tri_base = static_slots + geo_slot_offset(geo, stamp);
    node_off = tri_base * kNodeBytes;
  }
};
constexpr uint32_t kNoNextBit = 0x80000000u;  // <VER> shadow records: index of the continuing path, or this bit | the geometry stamp
// SPLIT: the ring holds the tree of the edited instances only, the static tree at offset 0 is walked second (pt_wavetrace.h)
template <bool SPLIT>
struct ExtendVerIOT : ExtendIO {
  static constexpr bool kVersioned = true, kSplit = SPLIT;
  GeoRing g;
  __device__ __forceinline__ uint32_t top_offset() const { return g.top_off; }
  __device__ __forceinline__ uint32_t static_slots() const { return g.static_slots; }
  __device__ __forceinline__ void geometry(uint32_t i, uint32_t, uint32_t& node_off, uint32_t& tri_base) const {
    g.offsets(geo_stamp(((const uint32_t*)&q.P3[i])[3]), node_off, tri_base);
  }
};
template <bool SPLIT>
struct ConnectVerIOT : ConnectIO {
  static constexpr bool kVersioned = true, kSplit = SPLIT;
  GeoRing g;
  __device__ __forceinline__ uint32_t top_offset() const { return g.top_off; }
  __device__ __forceinline__ uint32_t static_slots() const { return g.static_slots; }
  __device__ __forceinline__ void geometry(uint32_t i, uint32_t pay, uint32_t& node_off, uint32_t& tri_base) const {
    const uint32_t stamp = (pay & kNoNextBit) ? (pay & (kGeoVersions - 1u)) : geo_stamp(((const uint32_t*)&sq.S3[i])[3]);
    g.offsets(stamp, node_off, tri_base);
  }
  __device__ __forceinline__ void store(uint32_t i, const HitRec& h, uint32_t aux, uint32_t pay) const {
    ConnectIO::store(i, h, aux, (pay & kNoNextBit) ? 0xffffffffu : pay);
  }
};
typedef ExtendVerIOT<false> ExtendVerIO;
typedef ConnectVerIOT<false> ConnectVerIO;
typedef ExtendVerIOT<true> ExtendSplitIO;
typedef ConnectVerIOT<true> ConnectSplitIO;
// the camera rays of the memo and the rays of gsp_trace on a split scene: every ray belongs to the newest version (`stamp`)
struct MemoSplitIO : MemoIO {
  static constexpr bool kVersioned = true, kSplit = true;
  GeoRing g;
  uint32_t stamp;
  __device__ __forceinline__ uint32_t top_offset() const { return g.top_off; }
  __device__ __forceinline__ uint32_t static_slots() const { return g.static_slots; }
  __device__ __forceinline__ void geometry(uint32_t, uint32_t, uint32_t& node_off, uint32_t& tri_base) const { g.offsets(stamp, node_off, tri_base); }
};

struct TestIO {  // gsp_trace
  static constexpr float kTmin = -1.0f, kTmax = -1.0f;
  static constexpr bool kVersioned = false, kSplit = false;
  const float* rays;
  q4* hits;
  const uint32_t* slot_to_global;
  int any_hit;
  uint32_t num_tris;
  __device__ __forceinline__ void load(uint32_t i, f3& o, f3& d, float& tmin, float& tmax, uint32_t& pay) const {
    const float* r = rays + 8ull * i;
    o = mk3(r[0], r[1], r[2]);
    d = mk3(r[4], r[5], r[6]);
    tmin = r[3];
    tmax = r[7];
  }
  __device__ __forceinline__ void store(uint32_t i, const HitRec& h, uint32_t, uint32_t) const {
    const bool hit = h.slot >= 0 && num_tris != 0;
    if (any_hit) hits[i] = mkq(0.0f, 0.0f, 0.0f, ub(hit ? 0u : 0xffffffffu));
    else hits[i] = hit ? mkq(h.t, h.u, h.v, ub(slot_to_global[h.slot])) : mkq(0.0f, 0.0f, 0.0f, ub(0xffffffffu));
  }
};

struct TestSplitIO : TestIO {  // gsp_trace on a split scene (slot_to_global is indexed by the slots counted through both trees)
  static constexpr bool kVersioned = true, kSplit = true;
  GeoRing g;
  uint32_t stamp;
  __device__ __forceinline__ uint32_t top_offset() const { return g.top_off; }
  __device__ __forceinline__ uint32_t static_slots() const { return g.static_slots; }
  __device__ __forceinline__ void geometry(uint32_t, uint32_t, uint32_t& node_off, uint32_t& tri_base) const { g.offsets(stamp, node_off, tri_base); }
};

// ---- shade ----------------------------------------------------------------------
// Four blocks of 256 threads per CU (4 waves per SIMD, <= 128 VGPRs).  The two queue tails (next queue,
// shadow queue) are single words: with one atomic pair per WAVE the ~130k same-address atomics of
// an 8M-path launch serialise at the ~88/us a single address sustains (MI355X_MICROARCH.md,
// "dequeue" row) and cost half the kernel.  So survivors are counted per block through LDS and the
// block reserves its ranges with ONE atomic pair per tile.  Tile = block = 256 paths (r01_j A/B with the
// register budget pinned at 128: 1024 threads 77.7 ms, 512 76.2, 256 72.2, 128 104 per 48 spp): the waves of a
// block cost differently after the BSDF-type sort and meet at five barriers per tile, so four small blocks per CU
// overlap better than one large one, until the atomics per tile take over.
#ifndef GSP_SHADE_BLOCK
#define GSP_SHADE_BLOCK 256
#endif
constexpr int kShadeBlock = GSP_SHADE_BLOCK;
#ifndef GSP_SHADE_TABLE_BYTES
#define GSP_SHADE_TABLE_BYTES 8192
#endif
constexpr int kShadeTableBytes = GSP_SHADE_TABLE_BYTES;  // BSDF + light tables up to this size are staged into LDS by k_shade
constexpr int kShadeWaves = kShadeBlock / 64;
#ifndef GSP_SHADE_GRID_MULT
#define GSP_SHADE_GRID_MULT 1  // grid = exactly the resident blocks, each loops over tiles
#endif
#ifndef GSP_SHADE_MINWAVES
#define GSP_SHADE_MINWAVES 4  // 128 VGPRs: 4 blocks of 256 threads per CU
#endif
// r04: a software-pipelined tile loop (the hit / P0 / P1 records of tile i + 1 brought into a second set of LDS buffers by
// LDS-DMA, global_load_lds_dwordx4, while tile i is shaded; P2 / P3 gathered at the sorted index) was built, is bit-exact
// and measured flat to 2 % SLOWER on five scenes (profiles/r04_ab_shade_pipeline_*.txt; the patch:
// scripts/experiments/r04_shade_pipeline_lds_dma.patch): what the prefetch hides, the two extra gathers cost.
// TEX: scene with textures / an environment map (dormant-feature extension): a second instantiation, so that the code
// of the reference's path (TEX = false) is what it was
// VER: tables of several versions are live (gsp_update_tables while samples were in flight): every vertex reads the version its
// path carries, from HBM / L2 (no LDS copy: it would have to hold every live version) -- unless the versions in flight differ only
// in their GEOMETRY (gsp_update_instances): then the one live version of the tables is staged as usual (SceneView::ver_stride == 0)
template <bool TEX, bool VER>
__global__ __launch_bounds__(kShadeBlock, GSP_SHADE_MINWAVES) void k_shade(SceneView S, RenderConsts rc, const uint32_t* __restrict__ n_ptr, PathQueue cur,
                                                        const q4* __restrict__ hits, PathQueue nxt, ShadowQueue sq,
                                                        q4* __restrict__ result, uint32_t* __restrict__ tails,
                                                        uint32_t* __restrict__ live,
                                                        uint32_t slot_paths, uint32_t chunk, DevStats* __restrict__ stats) {
  const uint32_t n = *n_ptr;  // written by the previous iteration's k_shade / the host's memset (stream order)
#ifndef GSP_NO_LDS_TABLES
  // The BSDF and light tables of a scene are a few hundred bytes to a few KB, and every vertex makes two DEPENDENT
  // fetches into them (material record after the shading packet, light record after the RNG draw): staged into LDS
  // once per block those become ~64-cycle reads instead of L2 round trips in a kernel whose 4 waves per SIMD cannot
  // hide them.
  // (VER: only while ONE version of the tables is live -- ver_stride == 0: the versions in flight differ in their geometry)
  __shared__ uint4 s_tables[kShadeTableBytes / 16];
  if ((!VER || S.ver_stride == 0u) && S.tables_bytes <= (uint32_t)kShadeTableBytes) {
    const uint4* src = (const uint4*)S.tables;
    for (uint32_t k = threadIdx.x; k < S.tables_bytes / 16; k += kShadeBlock) s_tables[k] = src[k];
    const uint8_t* lb = (const uint8_t*)s_tables;
    const uint8_t* gb = S.tables;
    S.bsdf.diffuse = (const gsp_diffuse_bsdf*)(lb + ((const uint8_t*)S.bsdf.diffuse - gb));
    S.bsdf.smooth_dielectric = (const gsp_smooth_dielectric_bsdf*)(lb + ((const uint8_t*)S.bsdf.smooth_dielectric - gb));
    S.bsdf.smooth_conductor = (const gsp_smooth_conductor_bsdf*)(lb + ((const uint8_t*)S.bsdf.smooth_conductor - gb));
    S.bsdf.smooth_plastic = (const gsp_smooth_plastic_bsdf*)(lb + ((const uint8_t*)S.bsdf.smooth_plastic - gb));
    S.bsdf.rough_conductor = (const gsp_rough_conductor_bsdf*)(lb + ((const uint8_t*)S.bsdf.rough_conductor - gb));
    S.bsdf.smooth_floor = (const gsp_smooth_floor_bsdf*)(lb + ((const uint8_t*)S.bsdf.smooth_floor - gb));
    S.bsdf.rough_floor = (const gsp_rough_floor_bsdf*)(lb + ((const uint8_t*)S.bsdf.rough_floor - gb));
    S.bsdf.rough_plastic = (const gsp_rough_plastic_bsdf*)(lb + ((const uint8_t*)S.bsdf.rough_plastic - gb));
    S.lights = (const gsp_triangle_light*)(lb + ((const uint8_t*)S.lights - gb));
  }
  // (the barrier behind the s_dead initialisation below also publishes the staged tables)
#endif
  // textured scenes: the byte -> value table and the texture headers sit on the dependent chain packet -> BSDF record ->
  // header -> texels -> decode; from LDS the last and the third hop cost no memory round trip
  constexpr uint32_t kLdsTextures = 64;
  __shared__ float s_decode[TEX ? 256 : 1];
  __shared__ gsp_texture s_textures[TEX ? kLdsTextures : 1];
  if (TEX && S.tex.num_textures != 0) {
    for (uint32_t k = threadIdx.x; k < 256u; k += kShadeBlock) s_decode[k] = S.tex.decode[k];
    S.tex.decode = s_decode;
    if (S.tex.num_textures <= kLdsTextures) {
      for (uint32_t k = threadIdx.x; k < S.tex.num_textures; k += kShadeBlock) s_textures[k] = S.tex.textures[k];
      S.tex.textures = s_textures;
    }
  }
  __shared__ uint32_t s_dead[kMaxSlots];
  __shared__ uint32_t s_bin[12];               // counting sort of the tile by BSDF type: counts, then starts
  __shared__ uint16_t s_order[kShadeBlock];    // sorted position -> thread offset inside the tile
  __shared__ uint32_t s_cnt[2][kShadeWaves];   // per-wave survivor / shadow counts of this iteration
  __shared__ uint32_t s_base[2][kShadeWaves];  // per-wave first rank among the tile's survivors / shadow rays
  __shared__ uint32_t s_alloc[2][3];           // where those ranks go: rank r < R ? A + r : B + (r - R)   {A, R, B}
  __shared__ uint32_t s_room[2][2];            // the block's current chunk of each queue: {next free entry, end}
  if (threadIdx.x < 4) s_room[threadIdx.x >> 1][threadIdx.x & 1] = 0;
  __shared__ q4 s_hq[kShadeBlock], s_p0[kShadeBlock], s_p1[kShadeBlock], s_p2[kShadeBlock], s_p3[kShadeBlock];
  for (uint32_t k = threadIdx.x; k < (uint32_t)kMaxSlots; k += kShadeBlock) s_dead[k] = 0;
#ifdef GSP_SHADE_PROFILE
  for (uint32_t k = threadIdx.x; k < (uint32_t)PR_COUNT * 4; k += kShadeBlock) gsp_prof_table()[k] = 0;
#endif
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = threadIdx.x >> 6;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  // every thread of the block runs the same number of iterations (barriers inside the loop)
  const uint32_t stride = gridDim.x * kShadeBlock;
  const uint32_t iters = (n + stride - 1) / stride;
  unsigned long long shaded = 0;
  for (uint32_t it = 0; it < iters; ++it) {
    // ---- order the tile by the BSDF type of the hit (8 = miss, 9 = beyond the queue) so that the
    // lanes of a wave run the same branch of the 8-way BSDF switch (rayhit.rchit:630-654): LDS
    // counting sort of the tile's keys, laid out back to back.  (r05: on the bench scene that leaves three pure diffuse waves
    // and one wave with every other type of the tile, for which the others wait at the compaction barrier -- the lane profile,
    // profiles/r05_shade_lane_profile.txt; dealing the keys over the waves halves that wait and does not make the kernel
    // faster, profiles/r05_ab_shade_placement.txt, scripts/experiments/r05_shade_balanced_placement.patch)
    const uint32_t tile = it * stride + blockIdx.x * kShadeBlock;
    GSP_PROF_BEGIN(PR_TILE);
    GSP_PROF_BEGIN(PR_LOADSORT);
    {
      if (threadIdx.x < 12) s_bin[threadIdx.x] = 0;
      __syncthreads();
      const uint32_t i0 = tile + threadIdx.x;
      uint32_t key = 9;
      // the tile is read in queue order (fully coalesced) and handed to its sorted position through LDS
      if (i0 < n) {
        const q4 hq0 = qld(&hits[i0]);
        s_hq[threadIdx.x] = hq0;
        s_p0[threadIdx.x] = qld(&cur.P0[i0]);
        const q4 p1q = qld(&cur.P1[i0]);
        s_p1[threadIdx.x] = p1q;
        s_p2[threadIdx.x] = qld(&cur.P2[i0]);
        s_p3[threadIdx.x] = qld(&cur.P3[i0]);
        const uint32_t w = fb(hq0.w);
        key = (w == 0xffffffffu) ? 8u : ((w >> 28) & 7u);
        if (fb(p1q.w) == kFillerSid) key = 9u;  // a filler record behind some block's last chunk: not a path
      }
      const uint32_t rank = atomicAdd(&s_bin[key], 1u);
      __syncthreads();
      if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (int k = 0; k < 10; ++k) {
          const uint32_t c = s_bin[k];
          s_bin[k] = acc;
          acc += c;
        }
      }
      __syncthreads();
      s_order[s_bin[key] + rank] = (uint16_t)threadIdx.x;
      __syncthreads();
    }
    GSP_PROF_END(PR_LOADSORT);
    const uint32_t src = s_order[threadIdx.x];
    const uint32_t i = tile + src;
    bool alive = false, has_shadow = false;
    uint32_t my_sid = 0;
    ShadeOut out;
    q4 sum = mkq(0.0f, 0.0f, 0.0f, 0.0f);
#ifdef GSP_SHADE_PROFILE
    {  // how many sort keys (BSDF types, miss, beyond the queue) share this wave?
      const q4 hk = s_hq[src];
      const uint32_t wk = fb(hk.w), key2 = (i < n) ? ((wk == 0xffffffffu) ? 8u : ((wk >> 28) & 7u)) : 9u;
      int kinds = 0;
      for (uint32_t k = 0; k < 10; ++k) kinds += __ballot(key2 == k) != 0ull;
      if (lane == 0) atomicAdd(gsp_prof_table() + 4 * (PR_TYPES_IN_WAVE + kinds), 1ull);
    }
#endif
    const bool valid = i < n && fb(s_p1[src].w) != kFillerSid;
    if (valid) {
      GSP_PROF_BEGIN(PR_FETCH);
      const q4 hq = s_hq[src];
      const q4 p0 = s_p0[src], p1 = s_p1[src], p2 = s_p2[src];
      sum = s_p3[src];  // the sample's sum so far + the flags word
      const uint32_t fl = fb(sum.w);
      my_sid = fb(p1.w);
      HitRec h;
      h.t = hq.x;
      h.u = hq.y;
      h.v = hq.z;
      {
        const uint32_t w = fb(hq.w);
        h.slot = (w == 0xffffffffu) ? -1 : (int32_t)(w & 0x0fffffffu);
      }
      GSP_PROF_END(PR_FETCH);
      if (h.slot >= 0) {  // miss: miss.rmiss:15-18, the path ends and adds nothing
        GSP_PROF_BEGIN(PR_VERTEX);
        PathState in;
        in.o = mk3(p0.x, p0.y, p0.z);
        in.d = mk3(p0.w, p1.x, p1.y);
        in.seed = fb(p1.z);
        in.sid = my_sid;
        in.weight = mk3(p2.x, p2.y, p2.z);
        in.directWeight = p2.w;
        in.flags = fl;
        shade_vertex<TEX, VER>(S, rc, in, h, out);
        alive = out.alive;
        has_shadow = out.has_shadow;
        ++shaded;
        if (!has_shadow) add_emitted(rc.clamp, out.emitted, sum);  // (else k_trace<ConnectIO> adds the bounce's terms)
        GSP_PROF_END(PR_VERTEX);
      } else if (TEX && S.tex.env_texels != nullptr) {  // escaped: environment radiance, then the path ends
        PathState in;
        in.d = mk3(p0.w, p1.x, p1.y);
        in.weight = mk3(p2.x, p2.y, p2.z);
        add_emitted(rc.clamp, miss_emitted(S, in), sum);
      }
      // the path ends here and no shadow ray is pending: its sum is the sample
      if (!alive && !has_shadow) result[my_sid] = mkq(sum.x, sum.y, sum.z, 0.0f);
    }
    // paths that ended here leave their sample slot's live count (a slot is resolved when it
    // reaches 0): summed per block in LDS, flushed once at the end of the kernel
    GSP_PROF_BEGIN(PR_COMPACT);
    {
      const bool died = valid && !alive;
      uint64_t dm = __ballot(died);
      const uint32_t slot = my_sid / slot_paths;
      while (dm) {  // wave-uniform; lanes of a wave almost always share a slot
        const int first = __ffsll((unsigned long long)dm) - 1;
        const uint32_t s0 = (uint32_t)__shfl((int)slot, first);
        const uint64_t same = __ballot(died && slot == s0) & dm;
        if ((int)lane == first) atomicAdd(&s_dead[s0], (uint32_t)__popcll(same));
        dm &= ~same;
      }
    }
    // compaction: wave64 ballot + prefix popcount inside the wave, LDS scan over the 16 waves,
    // one atomic per queue per block
    const uint64_t am = __ballot(alive);
    const uint64_t sm = __ballot(has_shadow);
    if (lane == 0) {
      s_cnt[0][wave] = (uint32_t)__popcll(am);
      s_cnt[1][wave] = (uint32_t)__popcll(sm);
    }
    __syncthreads();
    // Room in the two output queues.  Every resident block asks once per tile and waits for the answer at the next barrier,
    // and one cache line takes 88 M atomic requests per second whatever they carry (scripts/microbench/atomic_rate.hip:
    // u32, u64, two lanes of one instruction alike; sixteen lines take sixteen times that).  r04: at 70 M tiles per second the
    // tail line was 79 % busy; r05: 80 M tiles per second, 91 %.  So a block of a large launch takes its room in CHUNKS of
    // `chunk` entries per queue -- one request per ~chunk / 150 tiles, a 64-bit add when both queues run out at once -- and
    // fills its chunks densely: a tile that does not fit uses up the old chunk and continues in the new one.  What a block has
    // left when the kernel ends is filled with records that are no paths (sid / next-path word all ones; rays that start 1e30
    // away and miss the root) and counted in T_HOLES_*: under 1 % of a queue.  chunk == 0 (small launches, statistics runs):
    // the tile's exact room, as before, with ONE 64-bit add for both queues.
    if (threadIdx.x == 0) {
      static_assert(T_NEXT == 0 && T_SHADOW == 1, "the two tails share one 64-bit word");
      uint32_t tot[2] = {0, 0};
      for (int w = 0; w < kShadeWaves; ++w) {
        s_base[0][w] = tot[0];
        s_base[1][w] = tot[1];
        tot[0] += s_cnt[0][w];
        tot[1] += s_cnt[1][w];
      }
      uint32_t rem[2], want[2];
      for (int q = 0; q < 2; ++q) {
        rem[q] = s_room[q][1] - s_room[q][0];
        want[q] = tot[q] > rem[q] ? (chunk ? chunk : tot[q]) : 0u;
      }
      unsigned long long got = 0;
      if (want[0] | want[1]) got = atomicAdd((unsigned long long*)tails, ((unsigned long long)want[1] << 32) | want[0]);
      for (int q = 0; q < 2; ++q) {
        const uint32_t b = q == 0 ? (uint32_t)got : (uint32_t)(got >> 32);
        s_alloc[q][0] = s_room[q][0];
        s_alloc[q][1] = want[q] ? rem[q] : 0xffffffffu;
        s_alloc[q][2] = b;
        if (want[q]) {
          s_room[q][0] = b + (tot[q] - rem[q]);
          s_room[q][1] = b + want[q];
        } else {
          s_room[q][0] += tot[q];
        }
      }
    }
    __syncthreads();
    uint32_t j = s_base[0][wave] + (uint32_t)__popcll(am & lt_mask);
    j = j < s_alloc[0][1] ? s_alloc[0][0] + j : s_alloc[0][2] + (j - s_alloc[0][1]);
    GSP_PROF_END(PR_COMPACT);
    GSP_PROF_BEGIN(PR_WRITE);
    // both outcomes of a bounce with a shadow ray, by the function k_finish and the host harness apply once the verdict is
    // known; the UNOCCLUDED one is written where it belongs right here, the occluded one travels with the shadow ray
    q4 clear = sum, occ = sum;
    if (has_shadow) {
      bool nee_done;
      connect_vertex(rc.clamp, out.shadow, false, clear, nee_done);
      connect_vertex(rc.clamp, out.shadow, true, occ, nee_done);
    }
    if (alive) {
      const PathState& p = out.next;
      qst(&nxt.P0[j], mkq(p.o.x, p.o.y, p.o.z, p.d.x));
      qst(&nxt.P1[j], mkq(p.d.y, p.d.z, ub(p.seed), ub(p.sid)));
      qst(&nxt.P2[j], mkq(p.weight.x, p.weight.y, p.weight.z, has_shadow ? out.shadow.dw_nee : p.directWeight));  // rayhit.rchit:785-787
      qst(&nxt.P3[j], mkq(clear.x, clear.y, clear.z, ub(p.flags)));  // (clear == sum without a shadow ray)
    }
    if (has_shadow) {
      uint32_t s = s_base[1][wave] + (uint32_t)__popcll(sm & lt_mask);
      s = s < s_alloc[1][1] ? s_alloc[1][0] + s : s_alloc[1][2] + (s - s_alloc[1][1]);
      const ShadowRay& r = out.shadow;
      if (!alive) result[my_sid] = mkq(clear.x, clear.y, clear.z, 0.0f);  // the path ended here: its sample, unless occluded
      qst(&sq.S0[s], mkq(r.o.x, r.o.y, r.o.z, r.tmax));
      if (VER) qst(&sq.S1[s], mkq(r.d.x, r.d.y, r.d.z, ub(alive ? j : (kNoNextBit | geo_stamp(fb(sum.w))))));  // (sum.w: the flags the vertex came with)
      else qst(&sq.S1[s], mkq(r.d.x, r.d.y, r.d.z, ub(alive ? j : 0xffffffffu)));
      qst(&sq.S3[s], mkq(occ.x, occ.y, occ.z, ub(alive ? out.next.flags : r.sid)));
    }
    GSP_PROF_END(PR_WRITE);
    GSP_PROF_END(PR_TILE);
  }
  __syncthreads();
  // what is left of the block's last chunks: records that are no paths (see the reservation above)
  {
    const uint32_t a0 = s_room[0][0], e0 = s_room[0][1], a1 = s_room[1][0], e1 = s_room[1][1];
    for (uint32_t k = a0 + threadIdx.x; k < e0; k += kShadeBlock) {
      qst(&nxt.P0[k], mkq(1e30f, 1e30f, 1e30f, 1.0f));
      qst(&nxt.P1[k], mkq(1.0f, 1.0f, 0.0f, ub(kFillerSid)));
      // the versioned ray sources read a record's geometry stamp from P3.w (ExtendVerIOT::geometry): a filler names stamp 0, not
      // whatever the slot held before (r05 ADVICE: it worked only because geo_slot_offset masks to the ring and the ring is zeroed)
      if (VER) qst(&nxt.P3[k], mkq(0.0f, 0.0f, 0.0f, ub(0u)));
    }
    for (uint32_t k = a1 + threadIdx.x; k < e1; k += kShadeBlock) {
      qst(&sq.S0[k], mkq(1e30f, 1e30f, 1e30f, -1.0f));  // tmax < tmin: the ray is over before the root
      qst(&sq.S1[k], mkq(1.0f, 1.0f, 1.0f, ub(VER ? kNoNextBit : 0xffffffffu)));  // (VER: "no next path", stamp 0 -- ConnectVerIOT::geometry)
    }
    if (threadIdx.x == 0 && e0 != a0) atomicAdd(&tails[T_HOLES_NEXT], e0 - a0);
    if (threadIdx.x == 0 && e1 != a1) atomicAdd(&tails[T_HOLES_SHADOW], e1 - a1);
  }
#ifdef GSP_SHADE_PROFILE
  for (uint32_t k = threadIdx.x; k < (uint32_t)PR_COUNT * 4; k += kShadeBlock)
    if (gsp_prof_table()[k]) atomicAdd(&g_shade_profile[k], gsp_prof_table()[k]);
#endif
  for (uint32_t k = threadIdx.x; k < (uint32_t)kMaxSlots; k += kShadeBlock)
    if (s_dead[k]) atomicSub(&live[k], s_dead[k]);
  shaded = wave_sum(shaded);
  if (lane == 0 && shaded) atomicAdd(&stats->shaded, shaded);
}

// ---- finish ----------------------------------------------------------------------
// The last few thousand paths of a drain, one path per lane from its current vertex to its end: extend, shade,
// shadow ray, next bounce, with no queues and no launches in between.  A wavefront iteration over a few hundred
// paths costs ~0.3 ms of launch, memset and read-back latency and a drain has ~45 of them; here every path pays
// only its own chain of dependent loads.  Same stage functions (shade_vertex, connect_vertex, add_emitted) and the
// same per-sample order of additions as k_shade / ConnectIO, so the arithmetic per path is unchanged.
constexpr uint32_t kFinishPaths = 262144;  // scan 0 / 64 k / 256 k / 1 M: 8-spp call 62 / 56 / 54 / 56 ms, 500x500 1-spp frames 88 / 138 / 182 / 184 per s
// the traversal stack of a k_finish lane: one node group per tree level, in LDS ([word][thread]; r02 kept 96 entries per
// lane in scratch memory, 400 B).  Trees deeper than this leave the tail of a drain to the wavefront kernels.
constexpr uint32_t kFinishLevels = 36;
struct FinishStack {
  lds_u32* col;  // this thread's column
  uint32_t top;  // words in use
  __device__ __forceinline__ void push(uint32_t v) { col[(top++) * kBlock] = v; }
  __device__ __forceinline__ uint32_t pop() { return col[(--top) * kBlock]; }
};

template <bool TEX, bool VER>
__global__ __launch_bounds__(kBlock) void k_finish(SceneView S, RenderConsts rc, uint32_t n, PathQueue q,
                                                    q4* __restrict__ result, uint32_t* __restrict__ tails,
                                                    uint32_t* __restrict__ live,
                                                    uint32_t slot_paths, DevStats* __restrict__ stats) {
  __shared__ uint32_t s_stack[kFinishLevels * kStackWords * kBlock];
  __shared__ uint32_t s_table[kStepTableBytes / 4];
  stage_step_table(s_table, threadIdx.x, kBlock);
  __syncthreads();
  const LdsStepTable tab{(const __attribute__((address_space(3))) char*)s_table};
  FinishStack stk{(lds_u32*)s_stack + threadIdx.x, 0u};
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  unsigned long long ext = 0, sh = 0, shaded = 0;
  uint32_t done_slot = 0xffffffffu;  // sample slot of the path this lane has run to its end (none: beyond the queue)
  if (i < n && fb(q.P1[i].w) != kFillerSid) {  // (not a filler record of k_shade's chunked reservation)
    const q4 p0 = q.P0[i], p1 = q.P1[i], p2 = q.P2[i];
    PathState in;
    in.o = mk3(p0.x, p0.y, p0.z);
    in.d = mk3(p0.w, p1.x, p1.y);
    in.seed = fb(p1.z);
    in.sid = fb(p1.w);
    in.weight = mk3(p2.x, p2.y, p2.z);
    in.directWeight = p2.w;
    const q4 p3 = q.P3[i];
    in.flags = fb(p3.w);
    const uint32_t sid = in.sid;
    q4 res = mkq(p3.x, p3.y, p3.z, 0.0f);  // the sample's sum so far
    // VER: this path's version of the geometry (constant along the path).  Split scene (S.static_slots != 0): the versioned tree
    // holds the edited instances only and sits behind the static tree, which every version shares: a ray walks both, and the hit
    // is min (t, tie-break key) of the two -- the closest-hit rule does not care which tree a triangle lives in.
    const uint32_t goff = VER ? S.static_slots + geo_slot_offset(S.geo, geo_stamp(in.flags)) : 0u;
    const q4* nodes_v = VER ? (const q4*)((const char*)S.nodes + (size_t)goff * kNodeBytes) : S.nodes;
    const q4* isect_v = VER ? S.tri_isect + 3ull * goff : S.tri_isect;
    const bool split = VER && S.static_slots != 0u;
    for (;;) {
      HitRec h;
      uint32_t aux;
      ++ext;
      stk.top = 0;
      uint32_t key = 0xffffffffu;
      bool hit = trace_ray<false>(nodes_v, isect_v, in.o, in.d, 0.0f, 1e10f, h, aux, stk, tab, &key);
      if (split) {
        if (hit) h.slot += (int32_t)S.static_slots;  // (slots are counted through both trees, version-free)
        HitRec h2;
        uint32_t aux2, key2 = 0xffffffffu;
        stk.top = 0;
        if (trace_ray<false>(S.nodes, S.tri_isect, in.o, in.d, 0.0f, 1e10f, h2, aux2, stk, tab, &key2) &&
            (!hit || h2.t < h.t || (h2.t == h.t && key2 < key))) {
          h = h2;
          aux = aux2;
          hit = true;
        }
      }
      if (!hit) {  // miss.rmiss:15-18
        if (TEX && S.tex.env_texels != nullptr) add_emitted(rc.clamp, miss_emitted(S, in), res);
        break;
      }
      ShadeOut out;
      shade_vertex<TEX, VER>(S, rc, in, h, out);
      ++shaded;
      if (!out.has_shadow) {
        add_emitted(rc.clamp, out.emitted, res);
      } else {
        HitRec hs;
        uint32_t aux2;
        ++sh;
        stk.top = 0;
        bool occluded = trace_ray<true>(nodes_v, isect_v, out.shadow.o, out.shadow.d, 0.01f, out.shadow.tmax, hs, aux2, stk, tab);
        if (split && !occluded) {
          stk.top = 0;
          occluded = trace_ray<true>(S.nodes, S.tri_isect, out.shadow.o, out.shadow.d, 0.01f, out.shadow.tmax, hs, aux2, stk, tab);
        }
        bool nee_done;
        connect_vertex(rc.clamp, out.shadow, occluded, res, nee_done);
        if (nee_done && out.alive) out.next.directWeight = out.shadow.dw_nee;  // rayhit.rchit:785-787
      }
      if (!out.alive) break;
      in = out.next;
    }
    result[sid] = res;
    done_slot = sid / slot_paths;
  }
  // every path of this launch leaves its sample slot's live count: one atomic per wave and slot (the lanes of a wave almost
  // always share a slot), not one per path -- 262 144 of them on one word are 3 ms at the 88 M requests/s a cache line takes
  // (profiles/r04_atomic_rate.txt), most of what this kernel took at the end of a drain
  {
    uint64_t dm = __ballot(done_slot != 0xffffffffu);
    while (dm) {  // wave-uniform
      const int first = __ffsll((unsigned long long)dm) - 1;
      const uint32_t s0 = (uint32_t)__shfl((int)done_slot, first);
      const uint64_t same = __ballot(done_slot == s0) & dm;
      if ((int)(threadIdx.x & 63) == first) atomicSub(&live[s0], (uint32_t)__popcll(same));
      dm &= ~same;
    }
  }
  ext = wave_sum(ext);
  sh = wave_sum(sh);
  shaded = wave_sum(shaded);
  if ((threadIdx.x & 63) == 0) {
    if (ext) atomicAdd((unsigned long long*)(tails + T_FIN_EXT), ext);
    if (sh) atomicAdd((unsigned long long*)(tails + T_FIN_SH), sh);
    if (shaded) atomicAdd(&stats->shaded, shaded);
  }
}

// ---- resolve ----------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_resolve(uint32_t num_pixels, uint32_t K, uint32_t first_timestamp,
                                                     const q4* __restrict__ result, q4* __restrict__ accum,
                                                     uint32_t lane, uint32_t lanes) {
  for (uint32_t l = blockIdx.x * kBlock + threadIdx.x; l < num_pixels; l += gridDim.x * kBlock) {
    const uint64_t lp = (uint64_t)l * lanes + lane;
    q4 a = accum[lp];
    for (uint32_t k = 0; k < K; ++k) resolve_sample(first_timestamp + k, result[(uint64_t)k * num_pixels + l], a);
    accum[lp] = a;
  }
}

// ---- resident table records ---------------------------------------------------------------------------------------------
// once per gsp_upload_scene / gsp_update_tables: what a vertex would compute from its light / diffuse record alone (pt_shading.h)
struct BakeTables {  // the resident BSDF tables: records of type t at rec[t] (stride rec_bytes[t]), their derived quads in front (derived_of)
  uint8_t* rec[GSP_BSDF_TYPE_COUNT];
  uint32_t rec_bytes[GSP_BSDF_TYPE_COUNT], num[GSP_BSDF_TYPE_COUNT];
};
__global__ __launch_bounds__(kBlock) void k_bake_tables(gsp_triangle_light* __restrict__ lights, uint32_t num_lights,
                                                        gsp_diffuse_bsdf* __restrict__ diffuse, uint32_t num_diffuse, BakeTables bt) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  for (uint32_t t = 0; t < GSP_BSDF_TYPE_COUNT; ++t)  // (before the diffuse record is rewritten: bake_bsdf reads records as uploaded)
    if (i < bt.num[t]) ((q4s*)bt.rec[t])[-1 - (int32_t)i] = bake_bsdf(t, bt.rec[t] + (size_t)i * bt.rec_bytes[t]);
  if (i < num_lights) {
    gsp_triangle_light L = lights[i];
    bake_light(L);
    lights[i] = L;
  }
  if (i < num_diffuse) {
    gsp_diffuse_bsdf b = diffuse[i];
    bake_diffuse(b);
    diffuse[i] = b;
  }
}

// ---- per-slot texture coordinates (dormant-feature extension) -----------------------------------------------------
// slot -> global triangle (BVH order) -> instance (tri_first is ascending) -> the instance's vertices in the uv array
__global__ __launch_bounds__(kBlock) void k_gather_uv(uint32_t num_tris, uint32_t first_slot, const uint32_t* __restrict__ slot_to_global,
                                                      const uint32_t* __restrict__ tri_first, uint32_t num_instances,
                                                      const gsp_instance* __restrict__ instances, const float* __restrict__ uvs,
                                                      float* __restrict__ tri_uv) {
  uint32_t s = blockIdx.x * kBlock + threadIdx.x;
  if (s >= num_tris) return;
  s += first_slot;
  const uint32_t g = slot_to_global[s];
  uint32_t lo = 0, hi = num_instances;  // last instance with tri_first <= g
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) / 2;
    if (tri_first[mid] <= g) lo = mid;
    else hi = mid;
  }
  const float* src = uvs + 2ull * (instances[lo].first_vertex + 3ull * (g - tri_first[lo]));
  float* dst = tri_uv + 8ull * s;
  for (int k = 0; k < 6; ++k) dst[k] = src[k];
  dst[6] = dst[7] = 0.0f;
}

// split scene: the triangles of the instances that have moved into the edited instances' tree leave the static one -- their
// slots become all-zero triangles (det == 0: never hit, like the padding slots); the boxes above them stay as they were
__global__ __launch_bounds__(kBlock) void k_retire_triangles(uint32_t num_tris, uint32_t first_slot, const uint32_t* __restrict__ slot_to_global,
                                                             const uint32_t* __restrict__ tri_first, uint32_t num_instances,
                                                             const uint8_t* __restrict__ retired, q4* __restrict__ isect) {
  uint32_t s = blockIdx.x * kBlock + threadIdx.x;
  if (s >= num_tris) return;
  s += first_slot;
  const uint32_t g = slot_to_global[s];
  uint32_t lo = 0, hi = num_instances;  // last instance with tri_first <= g
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) / 2;
    if (tri_first[mid] <= g) lo = mid;
    else hi = mid;
  }
  if (retired[lo]) {
    const q4 z = mkq(0.0f, 0.0f, 0.0f, 0.0f);
    isect[3ull * s + 0] = z;
    isect[3ull * s + 1] = z;
    isect[3ull * s + 2] = z;
  }
}

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t count = 0;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    count = 0;
  }
  hipError_t ensure(size_t n, size_t* tally) {
    if (n <= count && p) return hipSuccess;
    if (tally && p) *tally -= count * sizeof(T);
    release();
    n = n ? n : 1;
    hipError_t e = hipMalloc((void**)&p, n * sizeof(T));
    if (e == hipSuccess) {
      count = n;
      if (tally) *tally += n * sizeof(T);
    }
    return e;
  }
  hipError_t upload(const T* src, size_t n, hipStream_t s, size_t* tally) {
    hipError_t e = ensure(n, tally);
    if (e != hipSuccess || n == 0 || !src) return e;
    return hipMemcpyAsync(p, src, n * sizeof(T), hipMemcpyHostToDevice, s);
  }
};

std::mutex g_err_mutex;
std::string g_create_error = "";

}  // namespace

}  // namespace gsp

using namespace gsp;

struct gsp_context {
  int device = 0;
  int num_cus = 256;
  hipStream_t stream = nullptr;
  std::string err;
  size_t bytes = 0;

  // scene
  bool have_scene = false;
  DeviceBvh bvh;
  // the eight BSDF tables and the light table live back to back in ONE allocation (16-B aligned each), so that a kernel
  // can stage all of them into LDS with one cooperative copy when they are small (k_shade)
  // r05: `tables` is a RING of tab.slots slots (1 until an edit arrives with samples in flight, then up to kTableVersions: r06)
  // of tab.slot_bytes each; version v of the tables sits in slot (v - tab.rot) % tab.slots.  A sample carries its slot in its path flags, so gsp_update_tables need not wait for the
  // samples in flight (they finish on the version they started with) as long as the new tables have the layout of the old ones
  // and a slot is free.  While only ONE version is live it sits in slot 0 (tab.rot == tab.ver) and the flags field is 0.
  DevBuf<uint8_t> tables;
  size_t table_off[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // byte offsets inside a slot: BSDF types 0..7, then the lights
  size_t tables_bytes = 0;
  TableRing tab;  // slots, slot size, current version, rotation (pt_versions.h: pure bookkeeping, model-tested on the CPU)
  uint32_t num_lights = 0;
  // r05: the GEOMETRY ring (pt_stages.h): 2^geo.log2 <= kGeoVersions slots of geo.stride triangle slots each -- node records, intersection
  // triangles, shading packets -- so that gsp_update_instances need not wait for the samples in flight either: the refit goes
  // into the next slot, new samples are stamped with it, the old ones finish in theirs.  Made by the first gsp_update_instances
  // of a tree (which drains, as every one did until r04); `bvh.nodes / tri_isect / tri_shade` then point at the NEWEST slot.
  DevBuf<q4> ring_nodes, ring_isect, ring_shade;
  GeoVersions geo;  // stride (0 = no ring), log2 of the slots, current version, base slot (pt_versions.h)
  bool geo_ring_failed = false;  // no memory for it: edits drain, as before
  // r05: SPLIT scene (gsp_update_instances): the instances the host has edited since the last full build live in a tree of their
  // own, `dyn`, and only THAT tree goes through the ring; `bvh` holds the instances that never changed, once, in front of the ring
  // (static_slots triangle slots / 64-B node records): the versions in flight share it, so the working set of a dozen versions is one
  // large tree + a dozen small ones instead of a dozen large ones (profiles/r05_edit_frame_breakdown.txt).  A ray walks both
  // (pt_wavetrace.h kSplit); the <VER> kernels run all the time then.
  bool split = false;
  bool split_declined = false;  // make_split has said no for this scene: edits go through the ring of whole trees
  DeviceBvh dyn;
  uint32_t static_slots = 0;
  std::vector<uint8_t> inst_dynamic;   // per instance: edited since the last full build
  std::vector<uint32_t> sub_index[2];  // [0] static, [1] edited: scene index of the subset's instances
  DevBuf<gsp_instance> d_inst_sub[2];
  DevBuf<float> d_invt_sub[2];
  DevBuf<uint32_t> d_first_sub[2], d_idfirst_sub[2];
  DevBuf<uint8_t> d_retired;           // per instance: its triangles have left the static tree
  DevBuf<uint32_t> s2g_all;            // slot counted through both trees -> scene triangle index (gsp_trace)
  bool s2g_all_valid = false;
  gsp_camera camera{};
  double bvh_build_ms = 0.0;
  // what gsp_upload_scene leaves resident for the per-frame edits (gsp_update_instances re-bakes from it, as the reference
  // keeps the BLAS of a mesh and rebuilds the TLAS, Renderer.cpp:122-131 / PathTracer.cpp:10-19): the object-space
  // vertex arrays (72 B per triangle), the instance table and its host copy, the table sizes
  DevBuf<gsp_instance> d_inst;
  DevBuf<float> d_invt, d_pos, d_nrm, d_uv;
  DevBuf<uint32_t> d_first;
  std::vector<gsp_instance> h_inst;
  std::vector<uint8_t> h_tables;  // host copy of the table image (counts + tables): gsp_update_tables compares before it drains
  uint32_t num_bsdfs[GSP_BSDF_TYPE_COUNT] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t num_vertices = 0, total_tris = 0;
  gsp_ctx_options opt{};  // resolved at creation (gsp_internal_resolve_options)

  // frame
  bool have_frame = false;
  uint32_t width = 0, height = 0;
  uint64_t num_pixels = 0;
  bool subset = false;
  DevBuf<uint32_t> pixel_ids;
  std::vector<uint32_t> pixel_ids_host;
  DevBuf<q4> accum;

  DevBuf<DevStats> dstats;
  // dormant-feature extension (textures / environment map, include/gpuspectral_pt.h); `textured` selects the <TEX> kernels
  DevBuf<float> tri_uv, texel_decode, env_texels;
  DevBuf<gsp_texture> textures;
  DevBuf<uint32_t> texels;
  uint32_t num_textures = 0, env_width = 0, env_height = 0;
  float env_to_local[16] = {};
  bool textured = false;
  // frame read-back (gsp_download / gsp_peek): two pinned staging buffers; chunk k + 1 crosses PCIe while chunk k is copied
  // into the caller's (pageable) framebuffer -- a plain hipMemcpy into pageable memory stages through ONE bounce buffer and
  // took 10 ms for the 33-MB frame of 1080p (r06: inside bench.py's timed region)
  static constexpr size_t kStageBytes = 4u << 20;
  uint8_t* h_stage[2] = {nullptr, nullptr};
  hipEvent_t stage_ev[2] = {nullptr, nullptr};
  DevBuf<float> trace_rays;  // gsp_trace: grow-only staging, kept across calls
  DevBuf<q4> trace_hits;
  DevBuf<uint32_t> trace_work;
  DevBuf<uint32_t> node_hist, tri_hist;  // collect_traversal_stats = 2 (gsp_debug_visit_histograms)
  uint32_t spill_stride = 0;
  gsp_stats stats{};

  // Streaming pipeline state: survives across gsp_render calls, drained by gsp_sync & friends.
  struct Batch {
    uint32_t t0, kb, slot;
    uint32_t ver;   // table version its samples were generated under
    uint32_t gver;  // ... and geometry version
  };
  struct Pipeline {
    bool active = false;
    uint64_t Kb = 1, batch_paths = 0, pool_target = 0, cap = 0;
    uint32_t num_slots = 0;
    std::deque<Batch> inflight;
    std::vector<char> slot_used;
    uint64_t n = 0;   // paths the next iteration to be queued will trace: exact when no iteration is in flight, else an upper bound
    double n_est = 0.0;    // ... and the expected number (survival ratio of the last iterations seen)
    double survive = 0.9;  // share of an iteration's paths that continue (running estimate)
    int cur = 0;      // queue buffer that holds them
    uint32_t iteration = 0;
    uint32_t next_ts = 0, remaining = 0;
    uint32_t folded_end = 0;  // one past the last timestamp folded into the accumulate buffer
    uint64_t front = 0;       // leading paths of the current queue whose hit records came from the primary-hit memo
  };
  // The owned pixels are dealt to kLanes independent pipelines (pixel lp belongs to lane lp % lanes), each
  // with its own pool, counters and stream.  While the host reads one lane's counters back and queues its
  // next iteration, the other lane's kernels keep the GPU busy, and a latency-bound k_shade of one lane
  // overlaps a VALU-bound k_trace of the other.  Same arithmetic per pixel, so the image does not depend on
  // the lane count.  Default 1 lane: with the large pool the second lane adds nothing, and concurrent kernels
  // make per-kernel durations (the roofline measurement) meaningless.
  struct Lane {
    uint32_t index = 0;
    hipStream_t stream = nullptr;
    uint64_t num_pixels = 0;
    uint64_t pool_cap = 0, result_cap = 0;
    DevBuf<q4> P0[2], P1[2], P2[2], P3[2], hits[2], result, S0, S1, S3;
    DevBuf<q4> memo;          // primary-hit memo: one hit record per owned pixel of this lane
    bool memo_valid = false;  // ... traced for the current scene / camera / frame
    DevBuf<uint32_t> counters;
    DevBuf<uint32_t> spill;
    uint32_t* h_counters = nullptr;  // pinned: one read-back buffer of C_READBACK words per iteration parity
    hipEvent_t done[2] = {nullptr, nullptr};  // that read-back has landed
    std::vector<hipEvent_t> ev;      // 2 x 4 kernel-timing events (collect_kernel_times)
    Pipeline pipe;
    // Iterations in flight: queued on `stream`, counters not yet read by the host.  Up to kPipeDepth of them, so the GPU
    // starts iteration i + 1 the moment iteration i ends instead of waiting for the host to wake up, read 4 KB and launch.
    struct Iter {
      bool traced = false, timing = false, finish = false;
      uint64_t slack = 0;     // filler records this iteration's k_shade may leave in each output queue (chunked reservation)
      uint64_t injected = 0;  // paths generated into the queue that the following iteration traces
      uint64_t front = 0;     // leading paths of this iteration's queue that were not traced (primary-hit memo)
    } it[2];
    uint32_t queued = 0;       // iterations in flight (0 .. kPipeDepth)
    uint32_t enq = 0, col = 0; // running index of the next iteration to queue / to collect (parity picks the tail set)
    uint64_t n_in = 0;         // EXACT input size of iteration `col` (queue entries, filler records included)
    uint64_t holes_in = 0;     // ... of which filler records of k_shade's chunked reservation (no paths, no rays)
    std::vector<uint32_t> h_live;      // host mirror of the slots' live counters
    std::vector<uint32_t> live_since;  // per slot: iteration whose read-back is the first to show the slot's current batch
  };
  static constexpr int kMaxLanes = 2;
  Lane lanes[kMaxLanes];
  uint32_t num_lanes = 1;  // gsp_ctx_options.lanes = 2: +11 % with 8 M-path pools, +-0 with the 32 M-path pool and k_finish
  gsp_render_params pipe_params{};  // integrator constants the lanes are running with
  uint32_t folded_idle = 0;         // timestamps folded when no pipeline is running (gsp_peek)
  uint32_t finish_paths = 0;        // k_finish takes over below this many live paths (gsp_ctx_options.finish_paths; 0 = never)
  bool primary_memo = true;         // gsp_ctx_options.primary_memo = 2: every sample traces its camera ray
  double memory_share = 0.4;        // of the free device memory, for the path pool + result ring (gsp_ctx_options.memory_share)
  bool pipe_active = false;

  // oldest table version a sample in flight may carry (= tab.ver when none is)
  uint32_t oldest_live_version() const {
    uint32_t o = tab.ver;
    for (uint32_t l = 0; l < num_lanes; ++l)
      if (lanes[l].pipe.active && !lanes[l].pipe.inflight.empty()) o = std::min(o, lanes[l].pipe.inflight.front().ver);
    return o;
  }
  uint32_t oldest_live_geo() const {
    uint32_t o = geo.ver;
    for (uint32_t l = 0; l < num_lanes; ++l)
      if (lanes[l].pipe.active && !lanes[l].pipe.inflight.empty()) o = std::min(o, lanes[l].pipe.inflight.front().gver);
    return o;
  }
  // the <VER> kernels' shadow records keep a flag in the top bit of a queue index (kNoNextBit)
  bool caps_allow_versions() const {
    for (uint32_t l = 0; l < num_lanes; ++l)
      if (lanes[l].pipe.active && lanes[l].pipe.cap >= (uint64_t)kNoNextBit) return false;
    return true;
  }
  // versioned == false: the tables and the geometry of the current version (all samples in flight belong to it); true: slot 0
  // of the rings + the strides, for the <VER> instantiations
  SceneView view(bool versioned = false) const {
    SceneView v;
    if (split) versioned = true;  // (a split scene has no array a plain kernel could walk)
    const bool ring = versioned && geo.stride != 0;
    const bool tab_versions = versioned && oldest_live_version() != tab.ver;  // (else: one version of the tables, wherever it sits)
    v.nodes = ring ? ring_nodes.p : bvh.nodes;
    v.tri_isect = ring ? ring_isect.p : bvh.tri_isect;
    v.tri_shade = ring ? ring_shade.p : bvh.tri_shade;
    v.geo = ring ? pack_geo(geo.base, geo.stride, geo.log2) : 0u;
    v.static_slots = split ? static_slots : 0u;
    // slot 0 = the base the <VER> kernels add their offset to, or the slot of the one live version (slot 0 again whenever the
    // plain kernels run: lane_enqueue moves it there)
    const uint8_t* tb = tables.p + (versioned && !tab_versions ? (size_t)tab.slot_of(tab.ver) * tab.slot_bytes : 0);
    v.ver_stride = tab_versions ? (uint32_t)tab.slot_bytes : 0u;
    v.bsdf.diffuse = (const gsp_diffuse_bsdf*)(tb + table_off[0]);
    v.bsdf.smooth_dielectric = (const gsp_smooth_dielectric_bsdf*)(tb + table_off[1]);
    v.bsdf.smooth_conductor = (const gsp_smooth_conductor_bsdf*)(tb + table_off[2]);
    v.bsdf.smooth_plastic = (const gsp_smooth_plastic_bsdf*)(tb + table_off[3]);
    v.bsdf.rough_conductor = (const gsp_rough_conductor_bsdf*)(tb + table_off[4]);
    v.bsdf.smooth_floor = (const gsp_smooth_floor_bsdf*)(tb + table_off[5]);
    v.bsdf.rough_floor = (const gsp_rough_floor_bsdf*)(tb + table_off[6]);
    v.bsdf.rough_plastic = (const gsp_rough_plastic_bsdf*)(tb + table_off[7]);
    v.lights = (const gsp_triangle_light*)(tb + table_off[8]);
    v.tables = tb;
    v.tables_bytes = (uint32_t)tables_bytes;
    v.num_lights = num_lights;
    v.inv_num_lights = num_lights ? 1.0f / (float)num_lights : 0.0f;
    if (textured) {
      v.tex.tri_uv = num_textures ? tri_uv.p : nullptr;
      v.tex.textures = textures.p;
      v.tex.texels = texels.p;
      v.tex.decode = texel_decode.p;
      v.tex.num_textures = num_textures;
      v.tex.env_texels = env_width ? env_texels.p : nullptr;
      v.tex.env_width = env_width;
      v.tex.env_height = env_height;
      for (int k = 0; k < 16; ++k) v.tex.env_to_local[k] = env_to_local[k];
    }
    return v;
  }
#ifndef GSP_BLOCKS_PER_CU
#define GSP_BLOCKS_PER_CU 7  // 7 x 22 KB LDS (stack + step table), 7 waves per SIMD (A/B: 5 -> -4 %, 6 -> -1 %, 8 spills)
#endif
#ifndef GSP_BLOCKS_PER_CU_ANY
#define GSP_BLOCKS_PER_CU_ANY GSP_BLOCKS_PER_CU  // ... of the any-hit launches
#endif
  uint32_t max_blocks(bool any = false) const {  // resident 256-thread blocks of a k_trace launch
    return (uint32_t)num_cus * (any ? GSP_BLOCKS_PER_CU_ANY : GSP_BLOCKS_PER_CU);
  }
  uint32_t grid_for(uint64_t n) const {
    uint64_t b = (n + kBlock - 1) / kBlock;
    return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(b, max_blocks()));
  }
  // Rays are handed out in `chunk`-sized pieces from kWorkShards counters; chunk c belongs to shard
  // c % kWorkShards and only blocks with blockIdx % kWorkShards == shard serve it, so the grid must
  // hold a block for every shard that owns a chunk.
  uint32_t trace_grid(uint64_t n, uint32_t chunk, bool any = false) const {
    const uint64_t chunks = (n + chunk - 1) / chunk;
    const uint64_t by_threads = (n + kTraceBlock - 1) / kTraceBlock;
    const uint64_t g = std::max<uint64_t>(by_threads, std::min<uint64_t>(chunks, kWorkShards));
    return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(g, max_blocks(any)));
  }
  int ensure_spill() {
    // a descent pushes at most one node group per level of the wide tree (+ sentinel, slack)
    const uint32_t bound = std::max(bvh.depth, split ? dyn.depth : 0u) + 2;
    const uint32_t need = (bound > (uint32_t)kLdsStackDepth ? bound - kLdsStackDepth : 1) * kStackWords;
    spill_stride = std::max(max_blocks(false), max_blocks(true)) * kBlock;
    for (uint32_t l = 0; l < num_lanes; ++l) GSP_HIP_TRY(lanes[l].spill.ensure((size_t)need * spill_stride, &bytes));
    return GSP_OK;
  }
};

#define CTX_TRY(ctx, expr)                                                                        \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" + std::to_string(__LINE__) + ")"; \
      return e_ == hipErrorOutOfMemory ? GSP_ERR_NOMEM : GSP_ERR_DEVICE;                          \
    }                                                                                             \
  } while (0)

extern "C" {

extern const char gsp_build_info_string[];  // build/pt_buildinfo.cpp, written by the Makefile

void gsp_default_render_params(gsp_render_params* p) {
  if (!p) return;
  std::memset(p, 0, sizeof(*p));
  p->spp = 1;
  p->first_timestamp = 0;
  p->max_depth = 50;       // raygen.rgen:27
  p->rr_start_depth = 10;  // raygen.rgen:66
  p->clamp = 20.0f;        // raygen.rgen:60
  p->disable_nee = 0;      // `#define NEE true`, rayhit.rchit:656
}

int gsp_abi_version(void) { return GSP_ABI_VERSION; }

const char* gsp_build_info(void) { return gsp_build_info_string; }

int gsp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* gsp_last_error(const gsp_context* ctx) {
  if (ctx) return ctx->err.c_str();
  std::lock_guard<std::mutex> lk(g_err_mutex);
  static thread_local std::string copy;
  copy = g_create_error;
  return copy.c_str();
}

static void set_create_error(const std::string& s) {
  std::lock_guard<std::mutex> lk(g_err_mutex);
  g_create_error = s;
}

static int pipeline_drain(gsp_context* ctx);
static int ensure_read_back_stage(gsp_context* ctx);
void gsp_ctx_destroy(gsp_context* ctx);
int gsp_ctx_create_ex(int device, const gsp_ctx_options* options, gsp_context** out);

void gsp_default_ctx_options(gsp_ctx_options* o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->struct_size = (uint32_t)sizeof(*o);
  o->lanes = 1;
  o->pool_paths = 96ull << 20;   // r03 scan: 32 M 7.84 / 48 M 7.98 / 64 M 8.09 / 96 M 8.24 / 128 M 8.21 Grays/s (profiles/r03_ab_pool_size.txt)
  o->ring_bytes = 16ull << 30;
  o->memory_share = 0.4;
  o->primary_memo = 1;
  o->finish_paths = kFinishPaths;
  o->reinsert_rounds = 7;        // 6 rounds
  o->gather_route = GSP_GATHER_AUTO;
  o->refit_growth = 1.25;
  o->geometry_versions = kGeoVersions;
}

}  // extern "C"

// defaults for the fields the caller left 0 or whose header is older than this library's
void gsp::gsp_internal_resolve_options(const gsp_ctx_options* in, gsp_ctx_options* out) {
  gsp_ctx_options d;
  gsp_default_ctx_options(&d);
  gsp_ctx_options c;
  std::memset(&c, 0, sizeof(c));
  if (in) std::memcpy(&c, in, std::min<size_t>(in->struct_size, sizeof(c)));
  *out = d;
  if (c.lanes) out->lanes = std::min<uint32_t>(c.lanes, (uint32_t)gsp_context::kMaxLanes);
  if (c.pool_paths) out->pool_paths = std::max<uint64_t>(1ull << 16, c.pool_paths);
  if (c.ring_bytes) out->ring_bytes = std::max<uint64_t>(1ull << 24, c.ring_bytes);
  if (c.memory_share > 0.0) out->memory_share = std::min(0.9, std::max(0.01, c.memory_share));
  if (c.primary_memo) out->primary_memo = c.primary_memo == 2 ? 2u : 1u;
  if (c.finish_paths) out->finish_paths = c.finish_paths;
  if (c.reinsert_rounds) out->reinsert_rounds = std::min<uint32_t>(c.reinsert_rounds, 65u);
  if (c.gather_route <= GSP_GATHER_COPY) out->gather_route = c.gather_route;
  if (c.refit_growth > 0.0) out->refit_growth = c.refit_growth;  // (<= 1: no refit can stay below it -> always rebuild)
  if (c.geometry_versions) out->geometry_versions = std::min<uint32_t>(c.geometry_versions, kGeoVersions);
}

extern "C" {

int gsp_ctx_create(int device, gsp_context** out) { return gsp_ctx_create_ex(device, nullptr, out); }

int gsp_ctx_create_ex(int device, const gsp_ctx_options* options, gsp_context** out) {
  if (!out) return GSP_ERR_INVALID;
  *out = nullptr;
  if (options && options->struct_size < 8) {
    set_create_error("gsp_ctx_options.struct_size is not set");
    return GSP_ERR_INVALID;
  }
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    set_create_error(std::string("no HIP device available (") + (e != hipSuccess ? hipGetErrorString(e) : "count 0") +
                     "); this library has no CPU fallback");
    return GSP_ERR_DEVICE;
  }
  if (device < 0 || device >= n) {
    set_create_error("device index out of range");
    return GSP_ERR_INVALID;
  }
  e = hipSetDevice(device);
  if (e != hipSuccess) {
    set_create_error(std::string("hipSetDevice: ") + hipGetErrorString(e));
    return GSP_ERR_DEVICE;
  }
  gsp_context* c = new gsp_context();
  c->device = device;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->num_cus = prop.multiProcessorCount;
  e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  gsp_internal_resolve_options(options, &c->opt);
  c->finish_paths = c->opt.finish_paths == 0xffffffffu ? 0u : c->opt.finish_paths;
  c->primary_memo = c->opt.primary_memo != 2;
  c->num_lanes = c->opt.lanes;
  c->memory_share = c->opt.memory_share;
  for (uint32_t l = 0; l < c->num_lanes && e == hipSuccess; ++l) {
    gsp_context::Lane& L = c->lanes[l];
    L.index = l;
    e = hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipHostMalloc((void**)&L.h_counters, 2 * C_READBACK * sizeof(uint32_t), hipHostMallocDefault);
    if (e == hipSuccess) std::memset(L.h_counters, 0, 2 * C_READBACK * sizeof(uint32_t));
    for (int k = 0; k < 2 && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&L.done[k], hipEventDisableTiming);
    L.h_live.assign(kMaxSlots, 0);
    L.live_since.assign(kMaxSlots, 0);
  }
  if (e != hipSuccess) {
    set_create_error(std::string("context setup: ") + hipGetErrorString(e));
    gsp_ctx_destroy(c);
    return GSP_ERR_DEVICE;
  }
  *out = c;
  return GSP_OK;
}

void gsp_ctx_destroy(gsp_context* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)pipeline_drain(ctx);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  for (gsp_context::Lane& L : ctx->lanes) {
    if (L.stream) (void)hipStreamSynchronize(L.stream);
    for (hipEvent_t e : L.ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : L.done)
      if (e) (void)hipEventDestroy(e);
    if (L.h_counters) (void)hipHostFree(L.h_counters);
    if (L.stream) (void)hipStreamDestroy(L.stream);
  }
  free_bvh(ctx->bvh);
  free_bvh(ctx->dyn);
  for (int k = 0; k < 2; ++k) {
    if (ctx->h_stage[k]) (void)hipHostFree(ctx->h_stage[k]);
    if (ctx->stage_ev[k]) (void)hipEventDestroy(ctx->stage_ev[k]);
  }
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

// ---- scene upload and per-frame edits -----------------------------------------------------------------------------------
// gsp_upload_scene = validate -> tables (BSDF arrays + lights) -> resident geometry + instance table -> bake + BVH build;
// gsp_update_tables / gsp_update_instances / gsp_update_camera redo only their own part (PathTracer.cpp:58-93 re-reads all
// of it every frame; only the BLAS of a mesh is kept there, Renderer.cpp:122-131).

static int check_instances(gsp_context* ctx, const gsp_instance* inst, uint32_t n, const uint32_t* num_bsdfs, uint64_t num_vertices,
                           uint64_t* total_tris) {
  uint64_t total = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const gsp_instance& in = inst[i];
    if (in.vertex_count % 3 != 0 || (uint64_t)in.first_vertex + in.vertex_count > num_vertices) {
      ctx->err = "instance " + std::to_string(i) + ": vertex range outside the position/normal arrays";
      return GSP_ERR_SCENE;
    }
    const uint32_t type = in.bsdf >> 16, idx = in.bsdf & 0xffffu;
    if (type >= GSP_BSDF_TYPE_COUNT || idx >= num_bsdfs[type]) {
      ctx->err = "instance " + std::to_string(i) + ": BSDF handle out of range";
      return GSP_ERR_SCENE;
    }
    total += in.vertex_count / 3;
  }
  if (total_tris) *total_tris = total;
  return GSP_OK;
}

// has_texture words against `num_textures` entries (0 textures: the words are ignored, as the reference's shaders ignore them)
static int check_texture_words(gsp_context* ctx, const gsp_scene_desc* sc, uint32_t num_textures) {
  if (num_textures == 0) return GSP_OK;
  auto bad = [&](int32_t h) { return h < 0 || (uint32_t)h > num_textures; };
  bool oob = false;
  for (uint32_t k = 0; k < sc->num_bsdfs[GSP_BSDF_DIFFUSE] && sc->diffuse_bsdfs; ++k) oob |= bad(sc->diffuse_bsdfs[k].has_texture);
  for (uint32_t k = 0; k < sc->num_bsdfs[GSP_BSDF_ROUGH_CONDUCTOR] && sc->rough_conductor_bsdfs; ++k)
    oob |= bad(sc->rough_conductor_bsdfs[k].has_texture);
  for (uint32_t k = 0; k < sc->num_bsdfs[GSP_BSDF_ROUGH_PLASTIC] && sc->rough_plastic_bsdfs; ++k)
    oob |= bad(sc->rough_plastic_bsdfs[k].has_texture);
  if (oob) {
    ctx->err = "has_texture must be 0 or 1 + the index of an entry of `textures`";
    return GSP_ERR_SCENE;
  }
  return GSP_OK;
}

// The eight BSDF arrays + the lights of `sc`, packed back to back (16-B aligned each) as they sit in the context's one table
// allocation; the counts ride in front of the image so that "equal images" means equal tables.
struct TableImage {
  std::vector<uint8_t> bytes;  // [9 x uint64 count][tables ...]: the device holds bytes.data() + kHead onwards
  size_t off[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, total = 0;
  static constexpr size_t kHead = 9 * sizeof(uint64_t);
};
static int pack_tables(gsp_context* ctx, const gsp_scene_desc* sc, TableImage& img) {
  const void* src[9] = {sc->diffuse_bsdfs, sc->smooth_dielectric_bsdfs, sc->smooth_conductor_bsdfs, sc->smooth_plastic_bsdfs,
                        sc->rough_conductor_bsdfs, sc->smooth_floor_bsdfs, sc->rough_floor_bsdfs, sc->rough_plastic_bsdfs, sc->lights};
  const size_t rec[9] = {sizeof(gsp_diffuse_bsdf), sizeof(gsp_smooth_dielectric_bsdf), sizeof(gsp_smooth_conductor_bsdf),
                         sizeof(gsp_smooth_plastic_bsdf), sizeof(gsp_rough_conductor_bsdf), sizeof(gsp_smooth_floor_bsdf),
                         sizeof(gsp_rough_floor_bsdf), sizeof(gsp_rough_plastic_bsdf), sizeof(gsp_triangle_light)};
  size_t bytes[9];
  uint64_t counts[9];
  img.total = 0;
  for (int k = 0; k < 9; ++k) {
    counts[k] = k < 8 ? sc->num_bsdfs[k] : sc->num_lights;
    bytes[k] = rec[k] * counts[k];
    if (bytes[k] && !src[k]) {
      ctx->err = "null array with non-zero count";
      return GSP_ERR_SCENE;
    }
    // a BSDF table is preceded by one derived quad per record (pt_shading.h derived_of / bake_bsdf; zero in the host image,
    // filled on the device by k_bake_tables): off[k] is the table's FIRST RECORD
    if (k < GSP_BSDF_TYPE_COUNT) img.total += 16 * (size_t)counts[k];
    img.off[k] = img.total;
    img.total += (bytes[k] + 15) & ~(size_t)15;
  }
  img.bytes.assign(TableImage::kHead + std::max<size_t>(img.total, 16), 0);
  std::memcpy(img.bytes.data(), counts, sizeof(counts));
  for (int k = 0; k < 9; ++k)
    if (bytes[k]) std::memcpy(img.bytes.data() + TableImage::kHead + img.off[k], src[k], bytes[k]);
  return GSP_OK;
}
// image -> device, as the NEXT version of the tables (queued on the context's stream; the image becomes the context's host copy,
// so the source stays alive).  next_version == false: the pipeline is drained -- the ring is (re)allocated if the layout has
// changed and the image becomes version tab.ver; true (gsp_update_tables with samples in flight, same layout, a free slot): the
// image goes into the slot of tab.ver + 1, which no sample in flight reads, and becomes current.
static int upload_tables(gsp_context* ctx, const gsp_scene_desc* sc, TableImage& img, bool next_version = false) {
  const size_t slot = (std::max<size_t>(img.total, 16) + 255) & ~(size_t)255;
  uint32_t write_slot = 0;
  if (!next_version) {
    // behind a drain (or the first upload): ONE slot -- the ring is made by the first edit that arrives with samples in flight
    // (grow_table_ring), not at upload: 64 slots of a large light table are gigabytes (r05 review)
    if (ctx->tab.upload_behind_drain(slot) || !ctx->tables.p) {  // (another slot size starts with one slot again)
      if (ctx->tables.p) ctx->bytes -= std::min(ctx->bytes, ctx->tables.count);
      ctx->tables.release();
      CTX_TRY(ctx, ctx->tables.ensure(ctx->tab.allocation_bytes(), &ctx->bytes));
    }
  } else {
    write_slot = ctx->tab.begin_next_version();
  }
  uint8_t* const base = ctx->tables.p + (size_t)write_slot * ctx->tab.slot_bytes;
  ctx->h_tables.swap(img.bytes);
  CTX_TRY(ctx, hipMemcpyAsync(base, ctx->h_tables.data() + TableImage::kHead, std::max<size_t>(img.total, 16), hipMemcpyHostToDevice, ctx->stream));
  for (int k = 0; k < 9; ++k) ctx->table_off[k] = img.off[k];
  ctx->tables_bytes = img.total;
  ctx->num_lights = sc->num_lights;
  for (int k = 0; k < GSP_BSDF_TYPE_COUNT; ++k) ctx->num_bsdfs[k] = sc->num_bsdfs[k];
  // the resident light and diffuse records carry what a vertex would compute from them alone (pt_shading.h bake_light / bake_diffuse)
  uint32_t nb = sc->num_lights;
  BakeTables bt;
  const uint32_t rec_bytes[GSP_BSDF_TYPE_COUNT] = {sizeof(gsp_diffuse_bsdf), sizeof(gsp_smooth_dielectric_bsdf), sizeof(gsp_smooth_conductor_bsdf),
                                                   sizeof(gsp_smooth_plastic_bsdf), sizeof(gsp_rough_conductor_bsdf), sizeof(gsp_smooth_floor_bsdf),
                                                   sizeof(gsp_rough_floor_bsdf), sizeof(gsp_rough_plastic_bsdf)};
  for (uint32_t t = 0; t < GSP_BSDF_TYPE_COUNT; ++t) {
    bt.rec[t] = base + img.off[t];
    bt.rec_bytes[t] = rec_bytes[t];
    bt.num[t] = sc->num_bsdfs[t];
    nb = std::max(nb, sc->num_bsdfs[t]);
  }
  if (nb) {
    hipLaunchKernelGGL(k_bake_tables, dim3((nb + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream,
                       (gsp_triangle_light*)(base + img.off[8]), sc->num_lights,
                       (gsp_diffuse_bsdf*)(base + img.off[GSP_BSDF_DIFFUSE]), sc->num_bsdfs[GSP_BSDF_DIFFUSE], bt);
    CTX_TRY(ctx, hipGetLastError());
  }
  return GSP_OK;
}

// r06: the table ring is made when an edit first arrives with samples in flight.  The live version (the one-slot allocation,
// flags field 0) is copied into slot 0 of the new ring and the old allocation released: between two gsp_render calls no launch
// is queued (pipeline_run leaves every stream idle), so nothing reads it any more -- the samples in flight are records in the
// path pool, and the launches that pick them up next get the new pointer.  GSP_OK with tab.slots still 1 when the memory is not
// to be had (the edit then waits for the queued samples, as every edit did until r04).
static int grow_table_ring(gsp_context* ctx, uint32_t slots) {
  if (slots < 2 || ctx->tab.slots != 1 || !ctx->tables.p) return GSP_OK;
  DevBuf<uint8_t> ring;
  if (ring.ensure(ctx->tab.slot_bytes * slots, nullptr) != hipSuccess) {
    (void)hipGetLastError();
    return GSP_OK;
  }
  CTX_TRY(ctx, hipMemcpyAsync(ring.p, ctx->tables.p, ctx->tab.slot_bytes, hipMemcpyDeviceToDevice, ctx->stream));
  CTX_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->bytes -= std::min(ctx->bytes, ctx->tables.count);
  std::swap(ctx->tables.p, ring.p);  // (`ring` now holds the one-slot allocation and frees it on return)
  std::swap(ctx->tables.count, ring.count);
  ctx->bytes += ctx->tables.count;
  ctx->tab.on_grown(slots);
  return GSP_OK;
}

// ---- geometry ring (gsp_context::ring_*) ----------------------------------------------------------------------------------
// Both need an idle pipeline: nothing queued may name the arrays that go away.
static void drop_geo_ring(gsp_context* ctx) {
  if (ctx->bvh.arrays_external) {  // the tree's arrays were a slot of the ring
    ctx->bvh.nodes = ctx->bvh.tri_isect = ctx->bvh.tri_shade = nullptr;
    ctx->bvh.arrays_external = false;
  }
  if (ctx->split) {  // ... and so were the edited instances' tree's
    ctx->dyn.nodes = ctx->dyn.tri_isect = ctx->dyn.tri_shade = nullptr;
    ctx->dyn.arrays_external = false;
    ctx->bytes -= std::min(ctx->bytes, ctx->dyn.bytes);
    free_bvh(ctx->dyn);
    ctx->split = false;
    ctx->static_slots = 0;
  }
  for (DevBuf<q4>* b : {&ctx->ring_nodes, &ctx->ring_isect, &ctx->ring_shade}) {
    if (b->p) ctx->bytes -= b->count * sizeof(q4);
    b->release();
  }
  ctx->geo.stride = 0;
  ctx->geo.log2 = 0;
  ctx->geo.ver = 0;
  ctx->geo.base = 0;
}
// Moves the tree's three arrays into slot 0 of a new ring of as many versions (a power of two, at most kGeoVersions) as 32-bit node
// offsets and a quarter of the free device memory allow: 176 B per triangle and version, 11 GB for 64 versions of a million
// triangles.  GSP_OK also when there is no ring to be had (fewer than four versions fit): gsp_update_instances then drains every
// time, as before.
static int make_geo_ring(gsp_context* ctx) {
  DeviceBvh& b = ctx->bvh;
  if (ctx->geo.stride != 0 || ctx->geo_ring_failed || !b.nodes || b.num_tris == 0 || b.arrays_external) return GSP_OK;
  const uint64_t slots = (uint64_t)b.num_tris + b.first_slot + (kWide - 1);
  size_t free_b = 0, total_b = 0;
  CTX_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
  // log2 of the slots: at most what the options allow (default kGeoVersions), 32-bit node offsets, a quarter of the free memory
  const GeoRingPlan plan = plan_geo_ring(0, 0, slots, b.num_nodes, ctx->opt.geometry_versions, free_b, kNodeAllocMin / kNodeBytes, kGeoMaxStride, kNodeBytes);
  if (plan.log2 < 2) return GSP_OK;
  const uint64_t stride = plan.stride_ring;
  const uint32_t lg = (uint32_t)plan.log2;
  hipStream_t st = ctx->stream;
  const size_t total = (size_t)stride << lg;
  if (ctx->ring_nodes.ensure(total * kNodeQuads, &ctx->bytes) != hipSuccess || ctx->ring_isect.ensure(total * 3, &ctx->bytes) != hipSuccess ||
      ctx->ring_shade.ensure(total * 4, &ctx->bytes) != hipSuccess) {
    (void)hipGetLastError();
    drop_geo_ring(ctx);
    ctx->geo_ring_failed = true;
    return GSP_OK;
  }
  // all-zero everywhere: the leading / trailing triangle slots of every version (build_bvh), node records nothing refers to
  CTX_TRY(ctx, hipMemsetAsync(ctx->ring_nodes.p, 0, total * kNodeQuads * sizeof(q4), st));
  CTX_TRY(ctx, hipMemsetAsync(ctx->ring_isect.p, 0, total * 3 * sizeof(q4), st));
  CTX_TRY(ctx, hipMemsetAsync(ctx->ring_shade.p, 0, total * 4 * sizeof(q4), st));
  ctx->geo.ver = 0;
  ctx->geo.base = 0;
  ctx->geo.log2 = lg;
  q4* nn = ctx->ring_nodes.p;  // (slot 0)
  q4* ni = ctx->ring_isect.p;
  q4* ns = ctx->ring_shade.p;
  const size_t b_nodes = (size_t)b.num_nodes * kNodeBytes, b_is = slots * 3 * sizeof(q4), b_sh = slots * 4 * sizeof(q4);
  CTX_TRY(ctx, hipMemcpyAsync(nn, b.nodes, b_nodes, hipMemcpyDeviceToDevice, st));
  CTX_TRY(ctx, hipMemcpyAsync(ni, b.tri_isect, b_is, hipMemcpyDeviceToDevice, st));
  CTX_TRY(ctx, hipMemcpyAsync(ns, b.tri_shade, b_sh, hipMemcpyDeviceToDevice, st));
  CTX_TRY(ctx, hipStreamSynchronize(st));
  (void)hipFree(b.nodes);
  (void)hipFree(b.tri_isect);
  (void)hipFree(b.tri_shade);
  const size_t freed = b_nodes + b_is + b_sh;  // (as build_bvh counted them)
  b.bytes -= std::min(b.bytes, freed);
  ctx->bytes -= std::min(ctx->bytes, freed);
  b.nodes = nn;
  b.tri_isect = ni;
  b.tri_shade = ns;
  b.arrays_external = true;
  ctx->geo.stride = (uint32_t)stride;
  return GSP_OK;
}
// the tree's arrays := the slot of version v
static void point_bvh_at(gsp_context* ctx, uint32_t v) {
  const size_t off = (size_t)ctx->geo.phys(v) * ctx->geo.stride;
  ctx->bvh.nodes = ctx->ring_nodes.p + off * kNodeQuads;
  ctx->bvh.tri_isect = ctx->ring_isect.p + off * 3;
  ctx->bvh.tri_shade = ctx->ring_shade.p + off * 4;
}

// ---- split scene -----------------------------------------------------------------------------------------------------------
// device tables of one subset of ctx->h_inst (which = 0 static / 1 edited) for build_bvh / refit_bvh
static int upload_subset(gsp_context* ctx, int which, BuildInput& bi) {
  const std::vector<uint32_t>& idx = ctx->sub_index[which];
  const uint32_t ni = (uint32_t)idx.size();
  std::vector<gsp_instance> inst(ni);
  std::vector<float> inv_t(16ull * ni);
  std::vector<uint32_t> first(ni + 1ull), idfirst(ni + 1ull);
  std::vector<uint32_t> scene_first(ctx->h_inst.size() + 1ull);
  uint32_t acc = 0;
  for (size_t i = 0; i < ctx->h_inst.size(); ++i) scene_first[i] = acc, acc += ctx->h_inst[i].vertex_count / 3;
  acc = 0;
  for (uint32_t k = 0; k < ni; ++k) {
    inst[k] = ctx->h_inst[idx[k]];
    float tr[16];
    transpose4(inst[k].transform, tr);
    inverse4(tr, &inv_t[16ull * k]);
    first[k] = acc;
    idfirst[k] = scene_first[idx[k]];
    acc += inst[k].vertex_count / 3;
  }
  first[ni] = acc;
  idfirst[ni] = 0;
  hipStream_t st = ctx->stream;
  CTX_TRY(ctx, ctx->d_inst_sub[which].upload(inst.data(), ni, st, &ctx->bytes));
  CTX_TRY(ctx, ctx->d_invt_sub[which].upload(inv_t.data(), inv_t.size(), st, &ctx->bytes));
  CTX_TRY(ctx, ctx->d_first_sub[which].upload(first.data(), first.size(), st, &ctx->bytes));
  CTX_TRY(ctx, ctx->d_idfirst_sub[which].upload(idfirst.data(), idfirst.size(), st, &ctx->bytes));
  CTX_TRY(ctx, hipStreamSynchronize(st));  // (the host vectors go out of scope)
  bi = BuildInput{};
  bi.instances = ctx->d_inst_sub[which].p;
  bi.inv_t = ctx->d_invt_sub[which].p;
  bi.tri_first = ctx->d_first_sub[which].p;
  bi.tri_id_first = ctx->d_idfirst_sub[which].p;
  bi.num_instances = ni;
  bi.positions = ctx->d_pos.p;
  bi.normals = ctx->d_nrm.p;
  bi.num_tris = acc;
  bi.reinsert_rounds = (int)ctx->opt.reinsert_rounds - 1;
  return GSP_OK;
}

// Makes (or re-makes, when an edit touches an instance not edited before) the two trees of a split scene and lays them out:
// [static tree | ring of the edited instances' tree].  The static tree IS the tree of the whole scene as it stands, minus the
// triangles of the edited instances (k_retire_triangles: their slots become all-zero triangles; the boxes above them keep what
// they enclosed): nothing is rebuilt but the small tree.  Needs an idle pipeline.  *made = false (and nothing changed) when the
// scene does not lend itself to a split: textures, nothing or too much edited (more than a quarter of the triangles), no room for
// at least four versions.
static int make_split(gsp_context* ctx, bool* made) {
  *made = false;
  if (ctx->num_textures != 0 || ctx->opt.refit_growth <= 1.0 || ctx->opt.geometry_versions < 4 || ctx->geo_ring_failed) return GSP_OK;
  if (ctx->geo.stride != 0 && !ctx->split) return GSP_OK;  // (the scene lives in a ring of whole trees already)
  if (ctx->stats.scene_splits >= kMaxSceneSplits) return GSP_OK;  // a host that keeps touching new objects: a wait per edit is no bargain
  DeviceBvh& S = ctx->bvh;
  if (!S.nodes || S.num_tris != ctx->total_tris || ctx->total_tris == 0) return GSP_OK;  // (the tree of the WHOLE scene: first split or a re-split)
  uint64_t tris[2] = {0, 0};
  ctx->sub_index[0].clear();
  ctx->sub_index[1].clear();
  for (size_t i = 0; i < ctx->h_inst.size(); ++i) {
    const int w = ctx->inst_dynamic[i] ? 1 : 0;
    ctx->sub_index[w].push_back((uint32_t)i);
    tris[w] += ctx->h_inst[i].vertex_count / 3;
  }
  if (!split_worthwhile(tris[0], tris[1])) return GSP_OK;
  hipStream_t st = ctx->stream;
  // ---- the edited instances' tree ----
  BuildInput bi;
  DeviceBvh D;
  int rc = upload_subset(ctx, 1, bi);
  if (rc == GSP_OK) rc = build_bvh(st, bi, D, ctx->err);
  if (rc != GSP_OK) {
    (void)hipStreamSynchronize(st);
    free_bvh(D);
    return rc;
  }
  struct Guard {  // (an early return below gives the small tree back)
    DeviceBvh* d;
    ~Guard() {
      if (d) free_bvh(*d);
    }
  } guard{&D};
  CTX_TRY(ctx, hipStreamSynchronize(st));
  const uint64_t slots_s = (uint64_t)S.num_tris + S.first_slot + (kWide - 1), slots_d = (uint64_t)D.num_tris + D.first_slot + (kWide - 1);
  size_t free_b = 0, total_b = 0;
  CTX_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
  const GeoRingPlan plan = plan_geo_ring(slots_s, S.num_nodes, slots_d, D.num_nodes, ctx->opt.geometry_versions, free_b, kNodeAllocMin / kNodeBytes, kGeoMaxStride, kNodeBytes);
  if (plan.log2 < 2) return GSP_OK;
  const uint64_t stride_s = plan.stride_static, stride_d = plan.stride_ring;
  const uint32_t lg = (uint32_t)plan.log2;
  auto total_slots = [&](uint32_t l) { return stride_s + (stride_d << l); };
  // ---- the new layout: [static | ring]; the static tree's arrays are copied into it, wherever they were ----
  const size_t total = (size_t)total_slots(lg);
  DevBuf<q4> nn, ni, ns;
  if (nn.ensure(total * kNodeQuads, nullptr) != hipSuccess || ni.ensure(total * 3, nullptr) != hipSuccess || ns.ensure(total * 4, nullptr) != hipSuccess) {
    (void)hipGetLastError();
    ctx->geo_ring_failed = true;
    return GSP_OK;  // (nothing has changed: the scene stays one tree)
  }
  CTX_TRY(ctx, hipMemsetAsync(nn.p, 0, total * kNodeQuads * sizeof(q4), st));
  CTX_TRY(ctx, hipMemsetAsync(ni.p, 0, total * 3 * sizeof(q4), st));
  CTX_TRY(ctx, hipMemsetAsync(ns.p, 0, total * 4 * sizeof(q4), st));
  CTX_TRY(ctx, hipMemcpyAsync(nn.p, S.nodes, (size_t)S.num_nodes * kNodeBytes, hipMemcpyDeviceToDevice, st));
  CTX_TRY(ctx, hipMemcpyAsync(ni.p, S.tri_isect, slots_s * 3 * sizeof(q4), hipMemcpyDeviceToDevice, st));
  CTX_TRY(ctx, hipMemcpyAsync(ns.p, S.tri_shade, slots_s * 4 * sizeof(q4), hipMemcpyDeviceToDevice, st));
  CTX_TRY(ctx, hipMemcpyAsync(nn.p + stride_s * kNodeQuads, D.nodes, (size_t)D.num_nodes * kNodeBytes, hipMemcpyDeviceToDevice, st));
  CTX_TRY(ctx, hipMemcpyAsync(ni.p + stride_s * 3, D.tri_isect, slots_d * 3 * sizeof(q4), hipMemcpyDeviceToDevice, st));
  CTX_TRY(ctx, hipMemcpyAsync(ns.p + stride_s * 4, D.tri_shade, slots_d * 4 * sizeof(q4), hipMemcpyDeviceToDevice, st));
  // the edited instances' triangles leave the static tree
  {
    std::vector<uint32_t> first(ctx->h_inst.size() + 1ull);
    uint32_t acc = 0;
    for (size_t i = 0; i < ctx->h_inst.size(); ++i) first[i] = acc, acc += ctx->h_inst[i].vertex_count / 3;
    first[ctx->h_inst.size()] = acc;
    CTX_TRY(ctx, ctx->d_first.upload(first.data(), first.size(), st, &ctx->bytes));
    CTX_TRY(ctx, ctx->d_retired.upload(ctx->inst_dynamic.data(), ctx->inst_dynamic.size(), st, &ctx->bytes));
    hipLaunchKernelGGL(k_retire_triangles, dim3((uint32_t)((S.num_tris + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, S.num_tris, S.first_slot,
                       (const uint32_t*)S.slot_to_global, (const uint32_t*)ctx->d_first.p, (uint32_t)ctx->h_inst.size(),
                       (const uint8_t*)ctx->d_retired.p, ni.p);
    CTX_TRY(ctx, hipGetLastError());
  }
  CTX_TRY(ctx, hipStreamSynchronize(st));
  // ---- swap the new arrays in ----
  if (ctx->split) {  // a re-split: the old edited tree goes
    ctx->dyn.nodes = ctx->dyn.tri_isect = ctx->dyn.tri_shade = nullptr;
    ctx->dyn.arrays_external = false;
    ctx->bytes -= std::min(ctx->bytes, ctx->dyn.bytes);
    free_bvh(ctx->dyn);
  }
  if (!S.arrays_external) {  // the first split: the tree owned its arrays
    const size_t freed = (size_t)S.num_nodes * kNodeBytes + slots_s * 7 * sizeof(q4);
    (void)hipFree(S.nodes);
    (void)hipFree(S.tri_isect);
    (void)hipFree(S.tri_shade);
    S.bytes -= std::min(S.bytes, freed);
    ctx->bytes -= std::min(ctx->bytes, freed);
  }
  for (DevBuf<q4>* old : {&ctx->ring_nodes, &ctx->ring_isect, &ctx->ring_shade})
    if (old->p) ctx->bytes -= std::min(ctx->bytes, old->count * sizeof(q4));
  std::swap(ctx->ring_nodes.p, nn.p), std::swap(ctx->ring_nodes.count, nn.count);  // (nn, ni, ns release the old arrays on return)
  std::swap(ctx->ring_isect.p, ni.p), std::swap(ctx->ring_isect.count, ni.count);
  std::swap(ctx->ring_shade.p, ns.p), std::swap(ctx->ring_shade.count, ns.count);
  ctx->bytes += (ctx->ring_nodes.count + ctx->ring_isect.count + ctx->ring_shade.count) * sizeof(q4);
  S.nodes = ctx->ring_nodes.p;
  S.tri_isect = ctx->ring_isect.p;
  S.tri_shade = ctx->ring_shade.p;
  S.arrays_external = true;
  {
    const size_t freed = (size_t)D.num_nodes * kNodeBytes + slots_d * 7 * sizeof(q4);
    (void)hipFree(D.nodes);
    (void)hipFree(D.tri_isect);
    (void)hipFree(D.tri_shade);
    D.bytes -= std::min(D.bytes, freed);
    D.nodes = ctx->ring_nodes.p + stride_s * kNodeQuads;
    D.tri_isect = ctx->ring_isect.p + stride_s * 3;
    D.tri_shade = ctx->ring_shade.p + stride_s * 4;
    D.arrays_external = true;
    ctx->bytes += D.bytes;
  }
  ctx->dyn = std::move(D);
  guard.d = nullptr;
  ctx->split = true;
  ctx->static_slots = (uint32_t)stride_s;
  ctx->geo.stride = (uint32_t)stride_d;
  ctx->geo.log2 = lg;
  ctx->geo.ver = 0;
  ctx->geo.base = 0;
  ctx->s2g_all_valid = false;  // (gsp_trace makes it when it needs it)
  *made = true;
  return ctx->ensure_spill();
}

// slot counted through both trees -> scene triangle index, for gsp_trace on a split scene (made on first use)
static int make_split_s2g(gsp_context* ctx) {
  hipStream_t st = ctx->stream;
  std::vector<uint32_t> all((size_t)ctx->static_slots + ctx->geo.stride, 0xffffffffu);
  {  // the static tree is the tree of the whole scene: its slots name scene triangles already
    const uint64_t nslots = (uint64_t)ctx->bvh.num_tris + ctx->bvh.first_slot + (kWide - 1);
    CTX_TRY(ctx, hipMemcpyAsync(all.data(), ctx->bvh.slot_to_global, nslots * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    CTX_TRY(ctx, hipStreamSynchronize(st));
  }
  {  // the edited instances' tree counts the triangles of ITS instance list
    const DeviceBvh& D = ctx->dyn;
    const uint64_t nslots = (uint64_t)D.num_tris + D.first_slot + (kWide - 1);
    std::vector<uint32_t> s2g(nslots);
    CTX_TRY(ctx, hipMemcpyAsync(s2g.data(), D.slot_to_global, nslots * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    CTX_TRY(ctx, hipStreamSynchronize(st));
    std::vector<uint32_t> scene_first(ctx->h_inst.size() + 1ull);
    uint32_t acc = 0;
    for (size_t i = 0; i < ctx->h_inst.size(); ++i) scene_first[i] = acc, acc += ctx->h_inst[i].vertex_count / 3;
    const std::vector<uint32_t>& idx = ctx->sub_index[1];
    std::vector<uint32_t> first(idx.size() + 1), idfirst(idx.size());
    acc = 0;
    for (size_t k = 0; k < idx.size(); ++k) {
      first[k] = acc;
      idfirst[k] = scene_first[idx[k]];
      acc += ctx->h_inst[idx[k]].vertex_count / 3;
    }
    first[idx.size()] = acc;
    for (uint64_t sl = D.first_slot; sl < (uint64_t)D.first_slot + D.num_tris; ++sl) {
      const uint32_t lt = s2g[sl];
      const size_t k = (size_t)(std::upper_bound(first.begin(), first.end(), lt) - first.begin()) - 1;
      all[(size_t)ctx->static_slots + sl] = idfirst[k] + (lt - first[k]);
    }
  }
  CTX_TRY(ctx, ctx->s2g_all.upload(all.data(), all.size(), st, &ctx->bytes));
  CTX_TRY(ctx, hipStreamSynchronize(st));
  ctx->s2g_all_valid = true;
  return GSP_OK;
}

// the edited instances' tree := its slot of version v
static void point_dyn_at(gsp_context* ctx, uint32_t v) {
  const size_t off = (size_t)ctx->static_slots + (size_t)ctx->geo.phys(v) * ctx->geo.stride;
  ctx->dyn.nodes = ctx->ring_nodes.p + off * kNodeQuads;
  ctx->dyn.tri_isect = ctx->ring_isect.p + off * 3;
  ctx->dyn.tri_shade = ctx->ring_shade.p + off * 4;
}

// instance table (ctx->h_inst) -> device, transformInvT per instance, world-space bake + BVH build from the RESIDENT vertex
// arrays, per-slot uv gather of a textured scene, traversal spill region
// refit == true (gsp_update_instances): keep the tree's topology if its boxes stay within gsp_ctx_options.refit_growth of what
// they were after the last full build; *refitted says which of the two happened
static int bake_and_build(gsp_context* ctx, bool refit = false, bool* refitted = nullptr) {
  hipStream_t st = ctx->stream;
  const uint32_t ni = (uint32_t)ctx->h_inst.size();
  if (refitted) *refitted = false;
  // ---- PathTracer::prepareScene (PathTracer.cpp:58-93): per-instance table ----
  std::vector<float> inv_t(16ull * ni);
  std::vector<uint32_t> tri_first(ni + 1ull);
  uint32_t acc = 0;
  for (uint32_t i = 0; i < ni; ++i) {
    const gsp_instance& in = ctx->h_inst[i];
    float tr[16];
    transpose4(in.transform, tr);
    inverse4(tr, &inv_t[16ull * i]);
    tri_first[i] = acc;
    acc += in.vertex_count / 3;
  }
  tri_first[ni] = acc;
  CTX_TRY(ctx, ctx->d_inst.upload(ctx->h_inst.data(), ni, st, &ctx->bytes));
  CTX_TRY(ctx, ctx->d_invt.upload(inv_t.data(), inv_t.size(), st, &ctx->bytes));
  CTX_TRY(ctx, ctx->d_first.upload(tri_first.data(), tri_first.size(), st, &ctx->bytes));
  BuildInput bi;
  bi.instances = ctx->d_inst.p;
  bi.inv_t = ctx->d_invt.p;
  bi.tri_first = ctx->d_first.p;
  bi.num_instances = ni;
  bi.positions = ctx->d_pos.p;
  bi.normals = ctx->d_nrm.p;
  bi.num_tris = (uint32_t)ctx->total_tris;
  bi.reinsert_rounds = (int)ctx->opt.reinsert_rounds - 1;
  if (refit && !ctx->split && ctx->opt.refit_growth > 1.0 && ctx->bvh.nodes && ctx->bvh.num_tris == bi.num_tris && bi.num_tris > 0) {  // (a split scene goes back to one tree by a build)
    double growth = 0.0;
    const size_t held = ctx->bvh.bytes;
    int rc = refit_bvh(st, bi, ctx->bvh, &growth, ctx->err);  // (synchronises the stream)
    ctx->bytes += ctx->bvh.bytes - held;                      // (the first refit of a tree allocates its scratch)
    if (rc != GSP_OK) {
      (void)hipStreamSynchronize(st);
      return rc;
    }
    if (growth <= ctx->opt.refit_growth) {  // (per-slot uv of a textured scene: the slot order has not changed)
      if (refitted) *refitted = true;
      return GSP_OK;
    }
  }
  // a new tree: its arrays are its own again, and whatever is in flight ends on the old ones first (the committed version's:
  // a refit into the next slot of the ring that grew too much is abandoned here)
  if (ctx->geo.stride && !ctx->split) point_bvh_at(ctx, ctx->geo.ver);
  if (ctx->pipe_active) {
    ++ctx->stats.scene_drains;
    int rc_ = pipeline_drain(ctx);
    if (rc_ != GSP_OK) return rc_;
  }
  drop_geo_ring(ctx);
  ctx->bytes -= ctx->bvh.bytes;
  if (ctx->node_hist.p) ctx->bytes -= (ctx->node_hist.count + ctx->tri_hist.count) * sizeof(uint32_t);
  ctx->node_hist.release();
  ctx->tri_hist.release();
  int rc = build_bvh(st, bi, ctx->bvh, ctx->err);  // (synchronises the stream: inv_t / tri_first may go out of scope)
  if (rc != GSP_OK) {
    (void)hipStreamSynchronize(st);  // ... also on a failed build: the uploads above read host vectors of this frame
    free_bvh(ctx->bvh);              // whatever the failed build had allocated: nothing of it is counted in ctx->bytes
    return rc;
  }
  ctx->bytes += ctx->bvh.bytes;
  if (ctx->num_textures) {  // the collapse defines the slot order: the per-slot uv follow it
    CTX_TRY(ctx, ctx->tri_uv.ensure(8ull * (ctx->total_tris + ctx->bvh.first_slot + kWide), &ctx->bytes));
    if (ctx->total_tris) {
      hipLaunchKernelGGL(k_gather_uv, dim3((uint32_t)((ctx->total_tris + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, (uint32_t)ctx->total_tris,
                         ctx->bvh.first_slot, ctx->bvh.slot_to_global, ctx->d_first.p, ni, ctx->d_inst.p, ctx->d_uv.p, ctx->tri_uv.p);
      CTX_TRY(ctx, hipGetLastError());
    }
  }
  CTX_TRY(ctx, hipStreamSynchronize(st));
  return ctx->ensure_spill();
}

int gsp_upload_scene(gsp_context* ctx, const gsp_scene_desc* sc) {
  if (!ctx || !sc) return GSP_ERR_INVALID;
  CTX_TRY(ctx, hipSetDevice(ctx->device));
  {
    int rc_ = pipeline_drain(ctx);
    if (rc_ != GSP_OK) return rc_;
  }
  ctx->have_scene = false;
  for (gsp_context::Lane& L : ctx->lanes) L.memo_valid = false;
  ctx->geo_ring_failed = false;
  ctx->inst_dynamic.clear();
  ctx->split_declined = false;
  // ---- validate ----
  if ((sc->num_instances && !sc->instances) || (sc->num_vertices && (!sc->positions || !sc->normals)) ||
      (sc->num_lights && !sc->lights)) {
    ctx->err = "null array with non-zero count";
    return GSP_ERR_SCENE;
  }
  uint64_t total_tris = 0;
  {
    int rc_ = check_instances(ctx, sc->instances, sc->num_instances, sc->num_bsdfs, sc->num_vertices, &total_tris);
    if (rc_ != GSP_OK) return rc_;
  }
  // the wide tree of n triangles has fewer than n nodes and the traversal's stack entry holds node indices below kMaxNodes
  // (pt_trace.h): checked here, before any device work
  if (total_tris >= (uint64_t)kMaxNodes) {
    ctx->err = "too many triangles: " + std::to_string(total_tris) + " (limit 2^25 - 1 = 33 554 431)";
    return GSP_ERR_SCENE;
  }
  // dormant-feature extension: textures / environment map
  const bool want_tex = sc->num_textures != 0;
  const bool want_env = sc->envmap.texels != nullptr;
  if (want_tex) {
    if (!sc->textures || !sc->texels || (sc->num_vertices && !sc->uvs)) {
      ctx->err = "textures need the textures, texels and uvs arrays";
      return GSP_ERR_SCENE;
    }
    for (uint32_t k = 0; k < sc->num_textures; ++k) {
      const gsp_texture& t = sc->textures[k];
      if (t.width == 0 || t.height == 0 || t.width > (1u << 15) || t.height > (1u << 15) ||
          t.first_texel + (uint64_t)t.width * t.height > sc->num_texels) {
        ctx->err = "texture " + std::to_string(k) + ": size 0, above 32768 or outside the texel array";
        return GSP_ERR_SCENE;
      }
    }
    int rc_ = check_texture_words(ctx, sc, sc->num_textures);
    if (rc_ != GSP_OK) return rc_;
  }
  if (want_env && (sc->envmap.width == 0 || sc->envmap.height == 0 || sc->envmap.width > (1u << 15) || sc->envmap.height > (1u << 15))) {
    ctx->err = "environment map: size 0 or above 32768";
    return GSP_ERR_SCENE;
  }
  auto t0 = std::chrono::steady_clock::now();
  hipStream_t st = ctx->stream;
  {
    TableImage img;
    int rc_ = pack_tables(ctx, sc, img);
    if (rc_ == GSP_OK) rc_ = upload_tables(ctx, sc, img);
    if (rc_ != GSP_OK) return rc_;
  }
  ctx->camera = sc->camera;
  // ---- resident geometry (the "BLAS inputs": object-space vertices; Mesh.cpp:7-51) ----
  CTX_TRY(ctx, ctx->d_pos.upload(sc->positions, 3ull * sc->num_vertices, st, &ctx->bytes));
  CTX_TRY(ctx, ctx->d_nrm.upload(sc->normals, 3ull * sc->num_vertices, st, &ctx->bytes));
  ctx->num_vertices = sc->num_vertices;
  ctx->total_tris = total_tris;
  ctx->h_inst.assign(sc->instances, sc->instances + sc->num_instances);
  // ---- dormant-feature extension: texel arrays, uv, environment map ----
  ctx->textured = want_tex || want_env;
  ctx->num_textures = want_tex ? sc->num_textures : 0;
  ctx->env_width = want_env ? sc->envmap.width : 0;
  ctx->env_height = want_env ? sc->envmap.height : 0;
  if (want_tex) {
    float decode[256];
    for (int b = 0; b < 256; ++b) decode[b] = sc->texel_decode ? sc->texel_decode[b] : (float)b / 255.0f;
    CTX_TRY(ctx, ctx->texel_decode.upload(decode, 256, st, &ctx->bytes));
    CTX_TRY(ctx, ctx->textures.upload(sc->textures, sc->num_textures, st, &ctx->bytes));
    CTX_TRY(ctx, ctx->texels.upload(sc->texels, sc->num_texels, st, &ctx->bytes));
    CTX_TRY(ctx, ctx->d_uv.upload(sc->uvs, 2ull * sc->num_vertices, st, &ctx->bytes));
    CTX_TRY(ctx, hipStreamSynchronize(st));  // `decode` is a stack array
  }
  if (want_env) {
    CTX_TRY(ctx, ctx->env_texels.upload(sc->envmap.texels, 4ull * sc->envmap.width * sc->envmap.height, st, &ctx->bytes));
    for (int k = 0; k < 16; ++k) ctx->env_to_local[k] = sc->envmap.to_local[k];
  }
  // ---- bake + device BVH build ----
  int rc = bake_and_build(ctx);
  if (rc != GSP_OK) return rc;
  ctx->bvh_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  ctx->stats.scene_updates = 0;
  ctx->stats.scene_refits = 0;
  ctx->stats.scene_drains = 0;
  ctx->stats.scene_splits = 0;
  ctx->have_scene = true;
  return GSP_OK;
}

// common head of the gsp_update_* calls.  (The drain -- the samples already queued belong to the scene as it was -- comes only
// once the call has found something to change: a host that mirrors the reference calls all three every frame.)
static int begin_update(gsp_context* ctx, const char* what) {
  CTX_TRY(ctx, hipSetDevice(ctx->device));
  if (!ctx->have_scene) {
    ctx->err = std::string(what) + " needs gsp_upload_scene first";
    return GSP_ERR_INVALID;
  }
  return GSP_OK;
}

// No drain: a path in flight left the camera behind when its primary ray was generated (k_generate / the memo are the only
// readers of it, and gsp_render returns only after every sample of the call has been generated), so the samples queued so far
// finish as what they are -- samples through the old camera -- while the next gsp_render generates through the new one; they
// are folded in timestamp order either way.  A viewer that moves its camera every frame keeps the path pool full
// (profiles/r04_update_latency.txt: 500x500, one sample per frame).
int gsp_update_camera(gsp_context* ctx, const gsp_camera* camera) {
  if (!ctx || !camera) return GSP_ERR_INVALID;
  CTX_TRY(ctx, hipSetDevice(ctx->device));
  if (!ctx->have_scene) {
    ctx->err = "gsp_update_camera needs gsp_upload_scene first";
    return GSP_ERR_INVALID;
  }
  if (std::memcmp(&ctx->camera, camera, sizeof(gsp_camera)) == 0) return GSP_OK;
  ctx->camera = *camera;  // (render_consts reads it at the next gsp_render)
  for (gsp_context::Lane& L : ctx->lanes) L.memo_valid = false;  // the memo holds the hits of the OLD camera rays: re-traced
                                                                 // on the lane's stream before the next batch is generated
  ++ctx->stats.scene_updates;
  return GSP_OK;
}

int gsp_update_instances(gsp_context* ctx, const gsp_instance* instances, uint32_t num_instances) {
  if (!ctx || (!instances && num_instances)) return GSP_ERR_INVALID;
  int rc = begin_update(ctx, "gsp_update_instances");
  if (rc != GSP_OK) return rc;
  if (num_instances != ctx->h_inst.size()) {
    ctx->err = "gsp_update_instances: " + std::to_string(num_instances) + " instances, the uploaded scene has " +
               std::to_string(ctx->h_inst.size()) + " (a different object list needs gsp_upload_scene)";
    return GSP_ERR_SCENE;
  }
  for (uint32_t i = 0; i < num_instances; ++i)
    if (instances[i].first_vertex != ctx->h_inst[i].first_vertex || instances[i].vertex_count != ctx->h_inst[i].vertex_count) {
      ctx->err = "gsp_update_instances: instance " + std::to_string(i) + " names another vertex range than the uploaded one (a different mesh needs gsp_upload_scene)";
      return GSP_ERR_SCENE;
    }
  rc = check_instances(ctx, instances, num_instances, ctx->num_bsdfs, ctx->num_vertices, nullptr);
  if (rc != GSP_OK) return rc;
  if (num_instances == 0 || std::memcmp(ctx->h_inst.data(), instances, num_instances * sizeof(gsp_instance)) == 0) return GSP_OK;
  // r05, split scene.  The host that edits while samples are in flight (a viewer) usually moves a few objects of many: the first
  // such edit -- and every later one that touches an instance not edited before -- builds TWO trees, one over the instances that
  // have never changed and one over the edited ones (make_split: behind a drain, two builds); from then on an edit of those
  // instances refits the small tree only, into the next slot of its ring, and the versions in flight share the large one.
  if (ctx->inst_dynamic.size() != num_instances) ctx->inst_dynamic.assign(num_instances, 0);
  bool only_edited = ctx->split;
  for (uint32_t i = 0; i < num_instances; ++i)
    if (std::memcmp(&instances[i], &ctx->h_inst[i], sizeof(gsp_instance)) != 0 && !ctx->inst_dynamic[i]) only_edited = false;
  if (ctx->split && only_edited) {
    const bool in_slot = ctx->pipe_active && ctx->caps_allow_versions() && ctx->geo.ver + 1 - ctx->oldest_live_geo() < ctx->geo.slots();
    if (!in_slot && ctx->pipe_active) {
      ++ctx->stats.scene_drains;
      rc = pipeline_drain(ctx);
      if (rc != GSP_OK) return rc;
    }
    auto t0 = std::chrono::steady_clock::now();
    ctx->have_scene = false;
    for (gsp_context::Lane& L : ctx->lanes) L.memo_valid = false;
    ctx->h_inst.assign(instances, instances + num_instances);
    BuildInput bi;
    rc = upload_subset(ctx, 1, bi);
    if (rc != GSP_OK) return rc;
    if (in_slot) {
      const q4* from = ctx->dyn.nodes;
      point_dyn_at(ctx, ctx->geo.ver + 1);
      CTX_TRY(ctx, hipMemcpyAsync(ctx->dyn.nodes, from, (size_t)ctx->dyn.num_nodes * kNodeBytes, hipMemcpyDeviceToDevice, ctx->stream));
    }
    double growth = 0.0;
    const size_t held = ctx->dyn.bytes;
    rc = refit_bvh(ctx->stream, bi, ctx->dyn, &growth, ctx->err);
    ctx->bytes += ctx->dyn.bytes - held;
    if (rc != GSP_OK) {
      (void)hipStreamSynchronize(ctx->stream);
      point_dyn_at(ctx, ctx->geo.ver);
      return rc;
    }
    if (growth <= ctx->opt.refit_growth) {
      if (in_slot) ++ctx->geo.ver;
      ctx->bvh_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      ctx->have_scene = true;
      ++ctx->stats.scene_updates;
      ++ctx->stats.scene_refits;
      return GSP_OK;
    }
    // the edited tree has degraded: both trees are built again (below), behind a drain
    point_dyn_at(ctx, ctx->geo.ver);
  }
  if (ctx->split || (!ctx->split_declined && ctx->pipe_active && ctx->opt.refit_growth > 1.0 && ctx->opt.geometry_versions >= 4 &&
                     ctx->num_textures == 0)) {
    // The FIRST split needs no wait: the edited instances' tree is built where they WERE (version 0: with the static tree it is,
    // triangle for triangle, the scene the samples in flight were generated under -- they carry stamp 0 -- and the hit records
    // the memo has handed out name slots of the static tree whose shading packets stay), and the edit itself is version 1, a
    // refit into the next slot like every later one.  Nothing is queued between two gsp_render calls, so the arrays can change hands.
    const bool no_wait = !ctx->split && ctx->pipe_active && ctx->geo.ver == 0 && ctx->geo.stride == 0 && ctx->caps_allow_versions();
    if (ctx->pipe_active && !no_wait) {
      ++ctx->stats.scene_drains;
      rc = pipeline_drain(ctx);
      if (rc != GSP_OK) return rc;
    }
    auto t0 = std::chrono::steady_clock::now();
    for (uint32_t i = 0; i < num_instances; ++i)
      if (std::memcmp(&instances[i], &ctx->h_inst[i], sizeof(gsp_instance)) != 0) ctx->inst_dynamic[i] = 1;
    ctx->have_scene = false;
    for (gsp_context::Lane& L : ctx->lanes) L.memo_valid = false;
    if (!no_wait) ctx->h_inst.assign(instances, instances + num_instances);
    bool made = false;
    rc = make_split(ctx, &made);
    if (rc != GSP_OK && rc != GSP_ERR_NOMEM) return rc;
    if (made && no_wait) {
      ctx->h_inst.assign(instances, instances + num_instances);
      BuildInput bi;
      rc = upload_subset(ctx, 1, bi);
      if (rc != GSP_OK) return rc;
      const q4* from = ctx->dyn.nodes;
      point_dyn_at(ctx, 1);
      CTX_TRY(ctx, hipMemcpyAsync(ctx->dyn.nodes, from, (size_t)ctx->dyn.num_nodes * kNodeBytes, hipMemcpyDeviceToDevice, ctx->stream));
      double growth = 0.0;
      const size_t held = ctx->dyn.bytes;
      rc = refit_bvh(ctx->stream, bi, ctx->dyn, &growth, ctx->err);
      ctx->bytes += ctx->dyn.bytes - held;
      if (rc != GSP_OK) {
        (void)hipStreamSynchronize(ctx->stream);
        return rc;
      }
      if (growth <= ctx->opt.refit_growth) {
        ctx->geo.ver = 1;
      } else {  // the object has moved far in one step: the small tree is built where it IS, behind a wait after all
        point_dyn_at(ctx, 0);
        ++ctx->stats.scene_drains;
        rc = pipeline_drain(ctx);
        if (rc != GSP_OK) return rc;
        made = false;
        rc = make_split(ctx, &made);
        if (rc != GSP_OK && rc != GSP_ERR_NOMEM) return rc;
      }
    }
    if (made) {
      ctx->bvh_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      ctx->have_scene = true;
      ++ctx->stats.scene_updates;
      ++ctx->stats.scene_splits;
      return GSP_OK;
    }
    // no split to be had (too much of the scene is being edited, textures, no memory): the scene stays / becomes ONE tree with the
    // ring of whole trees -- below, with the pipeline idle now -- and the question is not asked again for this scene
    ctx->split_declined = true;
    ctx->have_scene = true;
    (void)t0;
  }
  // r05: NO DRAIN when the tree has its geometry ring and a slot of it is free: the refit -- re-bake of every packet, node boxes
  // bottom-up in the existing topology -- goes into the NEXT slot, on the context's stream, while the lanes' streams finish the
  // samples in flight in the slots they were generated under (a path carries its slot in its flags word, pt_stages.h); the next
  // gsp_render stamps its samples with the new one.  Otherwise -- first edit of this tree (the ring is made then), as many edits
  // as the ring has slots within the life of one sample, a tree that degrades and is rebuilt -- the queued samples finish first,
  // as until r04.
  const bool in_ring = ctx->pipe_active && ctx->geo.stride != 0 && ctx->opt.refit_growth > 1.0 && ctx->caps_allow_versions() &&
                       ctx->geo.ver + 1 - ctx->oldest_live_geo() < ctx->geo.slots();
  if (!in_ring) {
    if (ctx->pipe_active) ++ctx->stats.scene_drains;
    rc = pipeline_drain(ctx);
    if (rc != GSP_OK) return rc;
    // (the ring of WHOLE trees only where a split scene is not to be had: an edit that arrives with nothing in flight refits in
    // place, and the first one that arrives with samples in flight asks for the split first, above)
    const bool may_split = !ctx->split_declined && ctx->opt.geometry_versions >= 4 && ctx->num_textures == 0;
    if (ctx->opt.refit_growth > 1.0 && !may_split) {
      rc = make_geo_ring(ctx);
      if (rc != GSP_OK) return rc;
    }
  }
  auto t0 = std::chrono::steady_clock::now();
  ctx->have_scene = false;  // (a failed rebuild leaves no half-built tree in use)
  for (gsp_context::Lane& L : ctx->lanes) L.memo_valid = false;
  ctx->h_inst.assign(instances, instances + num_instances);
  bool refitted = false;
  if (in_ring) {
    // topology (child links, triangle ranges) of the current version -> next slot; refit_bvh rewrites every box and packet there
    const q4* from = ctx->bvh.nodes;
    point_bvh_at(ctx, ctx->geo.ver + 1);
    CTX_TRY(ctx, hipMemcpyAsync(ctx->bvh.nodes, from, (size_t)ctx->bvh.num_nodes * kNodeBytes, hipMemcpyDeviceToDevice, ctx->stream));
  }
  rc = bake_and_build(ctx, true, &refitted);
  if (rc != GSP_OK) {
    if (ctx->geo.stride) point_bvh_at(ctx, ctx->geo.ver);
    return rc;
  }
  if (in_ring && refitted) ++ctx->geo.ver;  // (not refitted: the tree was rebuilt behind a drain and owns its arrays again)
  ctx->bvh_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  ctx->have_scene = true;
  ++ctx->stats.scene_updates;
  if (refitted) ++ctx->stats.scene_refits;
  return GSP_OK;
}

int gsp_update_tables(gsp_context* ctx, const gsp_scene_desc* sc) {
  if (!ctx || !sc) return GSP_ERR_INVALID;
  int rc = begin_update(ctx, "gsp_update_tables");
  if (rc != GSP_OK) return rc;
  if (sc->num_lights && !sc->lights) {
    ctx->err = "null array with non-zero count";
    return GSP_ERR_SCENE;
  }
  rc = check_instances(ctx, ctx->h_inst.data(), (uint32_t)ctx->h_inst.size(), sc->num_bsdfs, ctx->num_vertices, nullptr);
  if (rc != GSP_OK) {
    ctx->err = "gsp_update_tables: the new tables leave a resident " + ctx->err;
    return rc;
  }
  rc = check_texture_words(ctx, sc, ctx->num_textures);
  if (rc != GSP_OK) return rc;
  TableImage img;
  rc = pack_tables(ctx, sc, img);
  if (rc != GSP_OK) return rc;
  if (img.bytes == ctx->h_tables) return GSP_OK;  // same counts, same bytes: nothing to do, nothing to wait for
  // r05: NO DRAIN when the new tables have the layout of the resident ones (same record counts: an edited colour, IOR, radiance
  // -- what a per-frame edit is) and the version ring has a free slot: the samples in flight carry their version in the path
  // flags and finish on the tables they started with, the next gsp_render generates under the new version.  (PathTracer.cpp:74-87
  // re-reads the tables every frame; a host that edits a material every frame keeps the path pool full now,
  // profiles/r05_update_latency.txt.)  Otherwise -- another layout, or kTableVersions edits within the life of one sample --
  // the queued samples finish first, as until r04.
  bool same_layout = ctx->pipe_active && sc->num_lights == ctx->num_lights && img.total == ctx->tables_bytes && ctx->tables.p != nullptr;
  for (int k = 0; k < GSP_BSDF_TYPE_COUNT && same_layout; ++k) same_layout = sc->num_bsdfs[k] == ctx->num_bsdfs[k];
  size_t free_b = 0, total_b = 0;
  if (same_layout && ctx->tab.slots == 1) CTX_TRY(ctx, hipMemGetInfo(&free_b, &total_b));  // (only the decision to grow asks)
  TableRing::Update how = ctx->tab.decide(ctx->pipe_active, same_layout, ctx->caps_allow_versions(), ctx->oldest_live_version(), free_b);
  if (how == TableRing::Update::kGrowThenInPlace) {
    rc = grow_table_ring(ctx, TableRing::slots_for(ctx->tab.slot_bytes, TableRing::budget_for(free_b)));
    if (rc != GSP_OK) return rc;
    how = ctx->tab.slots > 1 ? TableRing::Update::kInPlace : TableRing::Update::kDrain;  // (no memory for it: wait, as until r04)
  }
  const bool in_place = how == TableRing::Update::kInPlace;
  if (!in_place) {
    if (ctx->pipe_active) ++ctx->stats.scene_drains;
    rc = pipeline_drain(ctx);
    if (rc != GSP_OK) return rc;
  }
  rc = upload_tables(ctx, sc, img, in_place);  // (k_shade stages the tables per launch: nothing else holds a copy)
  if (rc != GSP_OK) {
    ctx->have_scene = false;
    return rc;
  }
  CTX_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ++ctx->stats.scene_updates;
  return GSP_OK;
}

int gsp_frame_begin(gsp_context* ctx, uint32_t width, uint32_t height, const uint32_t* pixel_ids,
                    uint64_t num_pixels) {
  if (!ctx || width == 0 || height == 0) return GSP_ERR_INVALID;
  CTX_TRY(ctx, hipSetDevice(ctx->device));
  {
    int rc_ = pipeline_drain(ctx);
    if (rc_ != GSP_OK) return rc_;
  }
  const uint64_t frame = (uint64_t)width * height;
  if (frame >= (1ull << 32)) {
    ctx->err = "frame too large";
    return GSP_ERR_INVALID;
  }
  ctx->have_frame = false;
  for (gsp_context::Lane& L : ctx->lanes) L.memo_valid = false;
  ctx->subset = pixel_ids != nullptr;
  if (pixel_ids) {
    for (uint64_t i = 0; i < num_pixels; ++i) {
      if (pixel_ids[i] >= frame || (i && pixel_ids[i] <= pixel_ids[i - 1])) {
        ctx->err = "pixel_ids must be strictly increasing and inside the frame";
        return GSP_ERR_INVALID;
      }
    }
    ctx->pixel_ids_host.assign(pixel_ids, pixel_ids + num_pixels);
    CTX_TRY(ctx, ctx->pixel_ids.upload(pixel_ids, num_pixels, ctx->stream, &ctx->bytes));
  } else {
    num_pixels = frame;
    ctx->pixel_ids_host.clear();
  }
  ctx->width = width;
  ctx->height = height;
  ctx->num_pixels = num_pixels;
  for (uint32_t l = 0; l < ctx->num_lanes; ++l)
    ctx->lanes[l].num_pixels = (num_pixels + ctx->num_lanes - 1 - l) / ctx->num_lanes;  // owned pixels lp with lp % lanes == l
  CTX_TRY(ctx, ctx->accum.ensure(num_pixels, &ctx->bytes));
  if (num_pixels * sizeof(q4) > gsp_context::kStageBytes / 4) {
    int rc_ = ensure_read_back_stage(ctx);
    if (rc_ != GSP_OK) return rc_;
  }
  CTX_TRY(ctx, hipMemsetAsync(ctx->accum.p, 0, std::max<uint64_t>(num_pixels, 1) * sizeof(q4), ctx->stream));
  CTX_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->folded_idle = 0;
  ctx->have_frame = true;
  return GSP_OK;
}

static int ensure_pool(gsp_context* ctx, gsp_context::Lane& L, uint64_t cap, uint64_t result_entries) {
  if (result_entries > L.result_cap) {
    CTX_TRY(ctx, L.result.ensure(result_entries, &ctx->bytes));
    L.result_cap = result_entries;
  }
  if (!L.counters.p) {
    CTX_TRY(ctx, L.counters.ensure(C_COUNT, &ctx->bytes));
    CTX_TRY(ctx, hipMemsetAsync(L.counters.p, 0, C_COUNT * sizeof(uint32_t), L.stream));
  }
  if (!ctx->dstats.p) {
    CTX_TRY(ctx, ctx->dstats.ensure(1, &ctx->bytes));
    CTX_TRY(ctx, hipMemsetAsync(ctx->dstats.p, 0, sizeof(DevStats), ctx->stream));
    CTX_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  if (cap <= L.pool_cap) return GSP_OK;
  for (int k = 0; k < 2; ++k) {
    CTX_TRY(ctx, L.P0[k].ensure(cap, &ctx->bytes));
    CTX_TRY(ctx, L.P1[k].ensure(cap, &ctx->bytes));
    CTX_TRY(ctx, L.P2[k].ensure(cap, &ctx->bytes));
    CTX_TRY(ctx, L.P3[k].ensure(cap, &ctx->bytes));
  }
  CTX_TRY(ctx, L.hits[0].ensure(cap, &ctx->bytes));
  CTX_TRY(ctx, L.hits[1].ensure(cap, &ctx->bytes));
  CTX_TRY(ctx, L.S0.ensure(cap, &ctx->bytes));
  CTX_TRY(ctx, L.S1.ensure(cap, &ctx->bytes));
  CTX_TRY(ctx, L.S3.ensure(cap, &ctx->bytes));
  L.pool_cap = cap;
  return GSP_OK;
}

static RenderConsts render_consts(const gsp_context* ctx) {
  const gsp_render_params* rp = &ctx->pipe_params;
  RenderConsts rcst;
  rcst.width = ctx->width;
  rcst.height = ctx->height;
  rcst.max_depth = rp->max_depth;
  rcst.rr_start_depth = rp->rr_start_depth;
  rcst.clamp = rp->clamp;
  rcst.nee = rp->disable_nee != 0 ? 0u : 1u;
  // raygen.rgen:22, tan() evaluated once on the host
  rcst.zplane = (std::max((float)ctx->width, (float)ctx->height) / 2.0f) / tanf(ctx->camera.fov / 2.0f);
  for (int i = 0; i < 16; ++i) rcst.cam_to_world[i] = ctx->camera.to_world[i];
  rcst.cam_origin[0] = ctx->camera.to_world[12];  // Camera::getPosition, Camera.cpp:41-45
  rcst.cam_origin[1] = ctx->camera.to_world[13];
  rcst.cam_origin[2] = ctx->camera.to_world[14];
  return rcst;
}

#ifndef GSP_PIPE_DEPTH
#define GSP_PIPE_DEPTH 2  // iterations queued ahead of the host's view of the counters (1 = wait for every read-back)
#endif
constexpr uint32_t kPipeDepth = GSP_PIPE_DEPTH;
static_assert(kPipeDepth >= 1 && kPipeDepth <= 2, "two tail sets / read-back buffers");

// Queues one iteration of a lane on its stream without waiting.  Iteration i:
//   tails[i & 1] := {next-queue size = paths injected now, shadow-queue size = 0}
//   k_generate   new batches (while the pool has room and ring slots are free) at the FRONT of the next queue
//   k_trace<Extend> / k_shade / k_trace<Connect> over the current queue, whose size they read from tails[(i - 1) & 1]
//                   on the device; survivors are appended behind the injected paths
//   copy of the counter words to the host buffer of this parity + an event
// `exact` = no iteration is in flight, so P.n is the true queue size (and 0 means there is nothing to trace).
static int lane_enqueue(gsp_context* ctx, gsp_context::Lane& L, const RenderConsts& rcst, const SceneView& view, bool drain) {
  gsp_context::Pipeline& P = L.pipe;
  hipStream_t st = L.stream;
  const gsp_render_params* rp = &ctx->pipe_params;
  const uint64_t npix = L.num_pixels;
  const uint64_t batch_paths = P.batch_paths;
  const bool stats_mode = rp->collect_traversal_stats != 0;
  const bool timing = rp->collect_kernel_times != 0;  // per-kernel HIP event timing (bench)
  while (timing && L.ev.size() < 8) {
    hipEvent_t e;
    CTX_TRY(ctx, hipEventCreate(&e));
    L.ev.push_back(e);
  }
  const uint32_t t = L.enq & 1u;
  const bool exact = L.queued == 0;
  uint32_t* tails_in = L.counters.p + kTailSet * (t ^ 1u);
  uint32_t* tails_out = L.counters.p + kTailSet * t;
  uint32_t* live = L.counters.p + C_LIVE;
  hipEvent_t* ev = timing ? &L.ev[4 * t] : nullptr;
  PathQueue Q[2];
  for (int k = 0; k < 2; ++k) Q[k] = PathQueue{L.P0[k].p, L.P1[k].p, L.P2[k].p, L.P3[k].p};
  ShadowQueue SQ{L.S0.p, L.S1.p, L.S3.p};
  TraceStatsOut so_ext{&ctx->dstats.p->nodes, &ctx->dstats.p->tris, &ctx->dstats.p->stat_rays, nullptr, nullptr, &ctx->dstats.p->lds_nodes};
  const TraceStatsOut so_sh{&ctx->dstats.p->sh_nodes, &ctx->dstats.p->sh_tris, &ctx->dstats.p->sh_rays, &ctx->dstats.p->sh_occluded,
                            &ctx->dstats.p->sh_occluded_nodes, &ctx->dstats.p->sh_lds_nodes, &ctx->dstats.p->sh_no_tri};
  if (rp->collect_traversal_stats >= 2) {  // per-record visit counts of the closest-hit rays (measurement hook; the two
    so_ext.node_hist = ctx->node_hist.p;   // histograms are allocated and zeroed by gsp_render before any lane runs)
    so_ext.tri_hist = ctx->tri_hist.p;
  }
  const uint64_t n = P.n;  // exact, or an upper bound of what this iteration traces
  const int cur = P.cur;
  gsp_context::Lane::Iter& I = L.it[t];
  I = gsp_context::Lane::Iter{};
  // do the samples in flight belong to more than one version of the BSDF / light tables?  (only after a gsp_update_tables
  // that did not drain: then k_shade / k_finish read every vertex's tables through the version its path carries)
  // ... or to more than one version of the geometry (gsp_update_instances without a drain: then the traversal kernels too take
  // every ray's geometry from the slot its path names)
  const bool multi_version = ctx->oldest_live_version() != ctx->tab.ver || ctx->oldest_live_geo() != ctx->geo.ver ||
                             ctx->split;  // (a split scene is always walked by the <VER> kernels: there is no whole tree in one place)
  if (!multi_version) ctx->geo.base = ctx->geo.phys(ctx->geo.ver);  // stamp 0 = the one live version (no copy: the kernels get its slot's pointers)
  if (!multi_version && ctx->tab.rot != ctx->tab.ver) {
    // the edits are over and the samples of the older versions have ended: the one live version moves into slot 0 and the
    // version field of the paths goes back to 0 (the <VER = false> kernels write 0).  Nothing reads slot 0 any more -- it held a
    // version whose last sample the host has seen end -- and nothing but the launches queued from here on reads the new copy.
    const uint32_t from = ctx->tab.collapse();
    if (from != 0) {
      CTX_TRY(ctx, hipMemcpyAsync(ctx->tables.p, ctx->tables.p + (size_t)from * ctx->tab.slot_bytes, ctx->tab.slot_bytes, hipMemcpyDeviceToDevice, ctx->stream));
      CTX_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
  }
  const SceneView vview = multi_version ? ctx->view(true) : view;
  const uint32_t gen_ver_bits = (ctx->tab.slot_of(ctx->tab.ver) << kVerShift) |  // (0 unless an edit is in flight)
                                (ctx->geo.phys(ctx->geo.ver + ctx->geo.slots() - ctx->geo.base) << kGeoShift);
  // (split scene: the static tree's top goes into LDS -- every ray walks it, whatever its version)
  const GeoRing gring{vview.geo, ctx->split ? 0u : (uint32_t)((size_t)ctx->geo.phys(ctx->geo.ver) * ctx->geo.stride * kNodeBytes),
                      ctx->split ? ctx->static_slots : 0u};
  const uint32_t gen_stamp = ctx->geo.phys(ctx->geo.ver + ctx->geo.slots() - ctx->geo.base);

  CTX_TRY(ctx, hipMemsetAsync(tails_out, 0, kTailSet * sizeof(uint32_t), st));
  const bool use_memo = ctx->primary_memo && !stats_mode;
  if (use_memo && !L.memo_valid && P.remaining > 0) {  // once per scene / camera / frame: trace the camera rays
    CTX_TRY(ctx, L.memo.ensure(npix, &ctx->bytes));
    CTX_TRY(ctx, hipMemsetAsync(L.counters.p + C_WORK_EXT, 0, kWorkShards * kWorkStride * sizeof(uint32_t), st));
    const MemoIO io{rcst, ctx->subset ? ctx->pixel_ids.p : nullptr, L.index, ctx->num_lanes, L.memo.p};
    const uint32_t chunk = npix >= (1u << 20) ? kChunkLarge : kChunkSmall;
    if (ctx->split) {
      MemoSplitIO sio;
      static_cast<MemoIO&>(sio) = io;
      sio.g = gring;
      sio.stamp = gen_stamp;
      hipLaunchKernelGGL((k_trace<false, false, MemoSplitIO>), dim3(ctx->trace_grid(npix, chunk)), dim3(kTraceBlock), 0, st, vview.nodes,
                         vview.tri_isect, (const uint32_t*)nullptr, (uint32_t)npix, 0u, chunk, sio, L.counters.p + C_WORK_EXT, L.spill.p,
                         ctx->spill_stride, so_ext);
    } else
    hipLaunchKernelGGL((k_trace<false, false, MemoIO>), dim3(ctx->trace_grid(npix, chunk)), dim3(kTraceBlock), 0, st, view.nodes,
                       view.tri_isect, (const uint32_t*)nullptr, (uint32_t)npix, 0u, chunk, io, L.counters.p + C_WORK_EXT, L.spill.p,
                       ctx->spill_stride, so_ext);
    CTX_TRY(ctx, hipGetLastError());
    ctx->stats.memo_build_rays += npix;
    L.memo_valid = true;
  }
  const uint64_t front = use_memo ? P.front : 0;  // (a stats pass traces everything: its counters describe all rays)
  I.front = front;
  if (drain && exact && P.remaining == 0 && n > 0 && n <= ctx->finish_paths && !stats_mode &&
      std::max(ctx->bvh.depth, ctx->split ? ctx->dyn.depth : 0u) + 2 <= kFinishLevels) {
    // the caller waits for the image, nothing is left to inject and few paths are alive: every path runs to its end
    // on its own lane (not when gsp_render merely queues work: those paths ride along with the next call's)
    const dim3 fgrid((uint32_t)((n + kBlock - 1) / kBlock));
    if (multi_version) {
      if (ctx->textured)
        hipLaunchKernelGGL((k_finish<true, true>), fgrid, dim3(kBlock), 0, st, vview, rcst, (uint32_t)n, Q[cur], L.result.p, tails_out, live,
                           (uint32_t)batch_paths, ctx->dstats.p);
      else
        hipLaunchKernelGGL((k_finish<false, true>), fgrid, dim3(kBlock), 0, st, vview, rcst, (uint32_t)n, Q[cur], L.result.p, tails_out, live,
                           (uint32_t)batch_paths, ctx->dstats.p);
    } else if (ctx->textured)
      hipLaunchKernelGGL((k_finish<true, false>), fgrid, dim3(kBlock), 0, st, view, rcst, (uint32_t)n, Q[cur], L.result.p, tails_out, live,
                         (uint32_t)batch_paths, ctx->dstats.p);
    else
      hipLaunchKernelGGL((k_finish<false, false>), fgrid, dim3(kBlock), 0, st, view, rcst, (uint32_t)n, Q[cur], L.result.p, tails_out, live,
                         (uint32_t)batch_paths, ctx->dstats.p);
    CTX_TRY(ctx, hipGetLastError());
    I.finish = true;
  } else {
    // ---- inject new batches while there is room: into the next queue, in front of this iteration's survivors ----
    // (the next iteration traces this one's survivors + what is injected now: aim that sum at the pool target with the
    // expected survivors, check the buffers against the upper bound)
    const double n_guess = exact ? (double)n : std::min((double)n, P.n_est);
    const double expect = n_guess * P.survive;
    // k_shade's room in the output queues, taken in chunks when every resident block shades many tiles (see the kernel):
    // the chunk is sized for ~1 % of filler records, and the queues must hold them
#ifndef GSP_ROOM_CHUNK
#define GSP_ROOM_CHUNK 1024
#endif
    const uint64_t shade_blocks = (uint64_t)ctx->num_cus * GSP_SHADE_GRID_MULT * (GSP_SHADE_MINWAVES * 256 / kShadeBlock);
    uint32_t room_chunk = 0;
    if (!stats_mode && GSP_ROOM_CHUNK >= kShadeBlock) {
      room_chunk = GSP_ROOM_CHUNK;
      while (room_chunk >= (uint32_t)kShadeBlock && (double)room_chunk > 0.036 * n_guess / (double)shade_blocks) room_chunk >>= 1;
      if (room_chunk < (uint32_t)kShadeBlock || n + shade_blocks * room_chunk > P.cap) room_chunk = 0;
    }
    const uint64_t slack = shade_blocks * room_chunk;
    I.slack = slack;
    uint64_t inj = 0;
    while (P.remaining > 0 && expect + (double)inj < (double)P.pool_target) {
      const uint32_t kb = (uint32_t)std::min<uint64_t>(P.Kb, P.remaining);
      uint32_t slot = P.num_slots;
      for (uint32_t s2 = 0; s2 < P.num_slots; ++s2)
        if (!P.slot_used[s2]) {
          slot = s2;
          break;
        }
      const uint64_t paths = (uint64_t)kb * npix;
      if (slot == P.num_slots || n + inj + paths + slack > P.cap) break;
      hipLaunchKernelGGL(k_generate, dim3(ctx->grid_for(paths)), dim3(kBlock), 0, st, rcst, (uint32_t)npix, kb, P.next_ts,
                         ctx->subset ? ctx->pixel_ids.p : nullptr, Q[cur ^ 1], (uint32_t)inj, (uint32_t)(slot * batch_paths),
                         use_memo ? L.memo.p : (const q4*)nullptr, L.hits[cur ^ 1].p, L.index, ctx->num_lanes, gen_ver_bits);
      CTX_TRY(ctx, hipGetLastError());
      CTX_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)(live + slot), (int)paths, 1, st));
      L.h_live[slot] = (uint32_t)paths;
      L.live_since[slot] = L.enq;  // read-backs of earlier iterations still show the slot's previous state
      P.slot_used[slot] = 1;
      P.inflight.push_back(gsp_context::Batch{P.next_ts, kb, slot, ctx->tab.ver, ctx->geo.ver});
      inj += paths;
      P.next_ts += kb;
      P.remaining -= kb;
    }
    if (inj) CTX_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)(tails_out + T_NEXT), (int)inj, 1, st));
    I.injected = inj;
    if (!(exact && n == 0)) {
      const uint32_t chunk = n >= (1u << 20) ? kChunkLarge : kChunkSmall;
      const uint32_t grid = ctx->trace_grid(std::max<uint64_t>(n, 1), chunk);
      CTX_TRY(ctx, hipMemsetAsync(L.counters.p + C_WORK_EXT, 0, (C_COUNT - C_WORK_EXT) * sizeof(uint32_t), st));
      if (timing) CTX_TRY(ctx, hipEventRecord(ev[0], st));
      {
        const ExtendIO io{Q[cur], L.hits[cur].p};
        uint32_t* work = L.counters.p + C_WORK_EXT;
        if (ctx->split) {
          const ExtendSplitIO vio{io, gring};
          hipLaunchKernelGGL((k_trace<false, false, ExtendSplitIO>), dim3(grid), dim3(kTraceBlock), 0, st, vview.nodes, vview.tri_isect,
                             (const uint32_t*)(tails_in + T_NEXT), 0u, (uint32_t)front, chunk, vio, work, L.spill.p, ctx->spill_stride,
                             so_ext);
        } else if (multi_version) {
          const ExtendVerIO vio{io, gring};
          hipLaunchKernelGGL((k_trace<false, false, ExtendVerIO>), dim3(grid), dim3(kTraceBlock), 0, st, vview.nodes, vview.tri_isect,
                             (const uint32_t*)(tails_in + T_NEXT), 0u, (uint32_t)front, chunk, vio, work, L.spill.p, ctx->spill_stride,
                             so_ext);
        } else if (stats_mode)
          hipLaunchKernelGGL((k_trace<false, true, ExtendIO>), dim3(grid), dim3(kTraceBlock), 0, st, view.nodes, view.tri_isect,
                             (const uint32_t*)(tails_in + T_NEXT), 0u, (uint32_t)front, chunk, io, work, L.spill.p, ctx->spill_stride,
                             so_ext);
        else
          hipLaunchKernelGGL((k_trace<false, false, ExtendIO>), dim3(grid), dim3(kTraceBlock), 0, st, view.nodes, view.tri_isect,
                             (const uint32_t*)(tails_in + T_NEXT), 0u, (uint32_t)front, chunk, io, work, L.spill.p, ctx->spill_stride,
                             so_ext);
        CTX_TRY(ctx, hipGetLastError());
      }
      if (timing) CTX_TRY(ctx, hipEventRecord(ev[1], st));
      const uint32_t shade_grid = (uint32_t)std::max<uint64_t>(
          1, std::min<uint64_t>((std::max<uint64_t>(n, 1) + kShadeBlock - 1) / kShadeBlock,
                                (uint64_t)ctx->num_cus * GSP_SHADE_GRID_MULT * (GSP_SHADE_MINWAVES * 256 / kShadeBlock)));  // the resident blocks
      if (multi_version) {  // samples of several table versions in flight (gsp_update_tables without a drain)
        if (ctx->textured)
          hipLaunchKernelGGL((k_shade<true, true>), dim3(shade_grid), dim3(kShadeBlock), 0, st, vview, rcst, (const uint32_t*)(tails_in + T_NEXT), Q[cur],
                             L.hits[cur].p, Q[cur ^ 1], SQ, L.result.p, tails_out, live, (uint32_t)batch_paths, room_chunk, ctx->dstats.p);
        else
          hipLaunchKernelGGL((k_shade<false, true>), dim3(shade_grid), dim3(kShadeBlock), 0, st, vview, rcst, (const uint32_t*)(tails_in + T_NEXT), Q[cur],
                             L.hits[cur].p, Q[cur ^ 1], SQ, L.result.p, tails_out, live, (uint32_t)batch_paths, room_chunk, ctx->dstats.p);
      } else if (ctx->textured)
        hipLaunchKernelGGL((k_shade<true, false>), dim3(shade_grid), dim3(kShadeBlock), 0, st, view, rcst, (const uint32_t*)(tails_in + T_NEXT), Q[cur],
                           L.hits[cur].p, Q[cur ^ 1], SQ, L.result.p, tails_out, live, (uint32_t)batch_paths, room_chunk, ctx->dstats.p);
      else
        hipLaunchKernelGGL((k_shade<false, false>), dim3(shade_grid), dim3(kShadeBlock), 0, st, view, rcst, (const uint32_t*)(tails_in + T_NEXT), Q[cur],
                           L.hits[cur].p, Q[cur ^ 1], SQ, L.result.p, tails_out, live, (uint32_t)batch_paths, room_chunk, ctx->dstats.p);
      CTX_TRY(ctx, hipGetLastError());
      if (timing) CTX_TRY(ctx, hipEventRecord(ev[2], st));
      {
        const ConnectIO io{SQ, Q[cur ^ 1].P2, Q[cur ^ 1].P3, L.result.p, rcst.clamp};
        uint32_t* work = L.counters.p + C_WORK_SH;
        const uint32_t grid_any = ctx->trace_grid(std::max<uint64_t>(n, 1), chunk, true);
        if (ctx->split) {
          const ConnectSplitIO vio{io, gring};
          hipLaunchKernelGGL((k_trace<true, false, ConnectSplitIO>), dim3(grid_any), dim3(kTraceBlock), 0, st, vview.nodes, vview.tri_isect,
                             (const uint32_t*)(tails_out + T_SHADOW), 0u, 0u, chunk, vio, work, L.spill.p,
                             ctx->spill_stride, so_sh);
        } else if (multi_version) {
          const ConnectVerIO vio{io, gring};
          hipLaunchKernelGGL((k_trace<true, false, ConnectVerIO>), dim3(grid_any), dim3(kTraceBlock), 0, st, vview.nodes, vview.tri_isect,
                             (const uint32_t*)(tails_out + T_SHADOW), 0u, 0u, chunk, vio, work, L.spill.p,
                             ctx->spill_stride, so_sh);
        } else if (stats_mode)
          hipLaunchKernelGGL((k_trace<true, true, ConnectIO>), dim3(grid_any), dim3(kTraceBlock), 0, st, view.nodes, view.tri_isect,
                             (const uint32_t*)(tails_out + T_SHADOW), 0u, 0u, chunk, io, work, L.spill.p,
                             ctx->spill_stride, so_sh);
        else
          hipLaunchKernelGGL((k_trace<true, false, ConnectIO>), dim3(grid_any), dim3(kTraceBlock), 0, st, view.nodes, view.tri_isect,
                             (const uint32_t*)(tails_out + T_SHADOW), 0u, 0u, chunk, io, work, L.spill.p,
                             ctx->spill_stride, so_sh);
        CTX_TRY(ctx, hipGetLastError());
      }
      if (timing) CTX_TRY(ctx, hipEventRecord(ev[3], st));
      I.traced = true;
      I.timing = timing;
    }
  }
  CTX_TRY(ctx, hipMemcpyAsync(L.h_counters + (size_t)t * C_READBACK, L.counters.p, C_READBACK * sizeof(uint32_t),
                              hipMemcpyDeviceToHost, st));
  CTX_TRY(ctx, hipEventRecord(L.done[t], st));
  ++L.queued;
  ++L.enq;
  P.cur ^= 1;
  P.front = use_memo && !I.finish ? I.injected : 0;
  P.n = (I.finish ? 0 : n + I.slack) + I.injected;  // survivors <= n: an upper bound of the next iteration's input until the read-back says more
  P.n_est = (I.finish ? 0.0 : (exact ? (double)n : std::min((double)n, P.n_est)) * P.survive) + (double)I.injected;
  return GSP_OK;
}

// Waits for the lane's oldest queued iteration, reads its counters and folds finished batches into the
// accumulate buffer, strictly in timestamp order.
static int lane_collect(gsp_context* ctx, gsp_context::Lane& L) {
  gsp_context::Pipeline& P = L.pipe;
  hipStream_t st = L.stream;
  if (L.queued) {
    const uint32_t t = L.col & 1u;
    CTX_TRY(ctx, hipEventSynchronize(L.done[t]));
    CTX_TRY(ctx, hipGetLastError());
    const uint32_t* rb = L.h_counters + (size_t)t * C_READBACK;
    const uint32_t* tails = rb + kTailSet * t;
    const gsp_context::Lane::Iter I = L.it[t];
    const uint64_t n_traced = L.n_in;
    if (I.finish) {
      ctx->stats.extension_rays += (uint64_t)tails[T_FIN_EXT] | ((uint64_t)tails[T_FIN_EXT + 1] << 32);
      ctx->stats.shadow_rays += (uint64_t)tails[T_FIN_SH] | ((uint64_t)tails[T_FIN_SH + 1] << 32);
    } else if (I.traced) {
      ctx->stats.extension_rays += n_traced - L.holes_in;  // path segments; I.front of them were answered from the memo
      ctx->stats.memoised_rays += I.front;
      ctx->stats.shadow_rays += tails[T_SHADOW] - tails[T_HOLES_SHADOW];
    }
    const uint32_t bounce = P.iteration++;
    if (I.timing) {
      float ms = 0.0f, e_ms = 0.0f, s_ms = 0.0f;
      hipEvent_t* ev = &L.ev[4 * t];
      CTX_TRY(ctx, hipEventElapsedTime(&e_ms, ev[0], ev[1]));
      ctx->stats.extend_kernel_ms += e_ms;
      ctx->stats.extend_launches += 1;
      CTX_TRY(ctx, hipEventElapsedTime(&s_ms, ev[1], ev[2]));
      ctx->stats.shade_kernel_ms += s_ms;
      CTX_TRY(ctx, hipEventElapsedTime(&ms, ev[2], ev[3]));
      ctx->stats.connect_kernel_ms += ms;
      if (ctx->pipe_params.collect_kernel_times >= 2)  // (test tools: one line per iteration)
        fprintf(stderr, "lane %u iter %3u: n %9llu shadow %9u injected %9llu inflight %2zu | extend %8.3f ms shade %8.3f ms connect %8.3f ms\n",
                L.index, bounce, (unsigned long long)n_traced, tails[T_SHADOW], (unsigned long long)I.injected, P.inflight.size(), e_ms, s_ms, ms);
    }
    // the input size of the next iteration, exactly: injected paths + survivors
    if (I.traced && n_traced > L.holes_in) {
      const double r = ((double)tails[T_NEXT] - (double)tails[T_HOLES_NEXT] - (double)I.injected) / (double)(n_traced - L.holes_in);
      P.survive = 0.5 * P.survive + 0.5 * std::min(1.0, std::max(0.0, r));
    }
    L.n_in = tails[T_NEXT];
    L.holes_in = I.traced ? tails[T_HOLES_NEXT] : 0;
    for (const gsp_context::Batch& b : P.inflight)
      if (L.live_since[b.slot] <= L.col) L.h_live[b.slot] = rb[C_LIVE + b.slot];
    ++L.col;
    --L.queued;
    // what the next iteration to be QUEUED will trace: exact if nothing is in flight, else bounded through the one that is
    P.n = L.queued ? L.n_in + L.it[L.col & 1u].slack + L.it[L.col & 1u].injected : L.n_in;
    P.n_est = L.queued ? (double)L.n_in * P.survive + (double)L.it[L.col & 1u].injected : (double)L.n_in;
  }
  while (!P.inflight.empty() && L.h_live[P.inflight.front().slot] == 0) {
    const gsp_context::Batch b = P.inflight.front();
    P.inflight.pop_front();
    hipLaunchKernelGGL(k_resolve, dim3(ctx->grid_for(L.num_pixels)), dim3(kBlock), 0, st, (uint32_t)L.num_pixels, b.kb, b.t0,
                       L.result.p + (uint64_t)b.slot * P.batch_paths, ctx->accum.p, L.index, ctx->num_lanes);
    CTX_TRY(ctx, hipGetLastError());
    P.slot_used[b.slot] = 0;
    P.folded_end = b.t0 + b.kb;
    ctx->stats.samples += (uint64_t)b.kb * L.num_pixels;
  }
  if (L.queued == 0 && P.n == 0 && P.remaining == 0 && !P.inflight.empty()) {
    ctx->err = "internal error: paths exhausted with unresolved sample batches";
    return GSP_ERR_DEVICE;
  }
  return GSP_OK;
}

// Runs the streaming pipelines.  drain == false: returns as soon as every queued sample has been
// injected (stragglers of the last batches stay in flight and ride along with the next call's
// launches); drain == true: runs until nothing is in flight and every batch has been folded into the
// accumulate buffer.  While samples remain to be injected each lane keeps kPipeDepth iterations queued, so the GPU
// never waits for the host; the tail of a drain (exact path counts decide k_finish and the end) runs one at a time.
static int pipeline_run(gsp_context* ctx, bool drain) {
  if (!ctx->pipe_active) return GSP_OK;
  const auto t_begin = std::chrono::steady_clock::now();
  const RenderConsts rcst = render_consts(ctx);
  const SceneView view = ctx->view();
  auto has_work = [&](const gsp_context::Lane& L) {
    const gsp_context::Pipeline& P = L.pipe;
    return P.active && (P.remaining > 0 || (drain && (P.n > 0 || !P.inflight.empty())));
  };
  auto depth_for = [&](const gsp_context::Lane& L) { return L.pipe.remaining > 0 && !ctx->pipe_params.collect_traversal_stats ? kPipeDepth : 1u; };
  for (;;) {
    bool any = false;
    for (uint32_t l = 0; l < ctx->num_lanes; ++l) {
      gsp_context::Lane& L = ctx->lanes[l];
      if (!L.pipe.active) continue;
      while (has_work(L) && L.queued < depth_for(L)) {
        if (L.queued == 0 && L.pipe.n == 0 && L.pipe.remaining == 0) break;  // nothing to trace: only batches to fold
        int rc = lane_enqueue(ctx, L, rcst, view, drain);
        if (rc != GSP_OK) return rc;
      }
    }
    for (uint32_t l = 0; l < ctx->num_lanes; ++l) {
      gsp_context::Lane& L = ctx->lanes[l];
      if (!L.pipe.active) continue;
      if (L.queued) {
        int rc = lane_collect(ctx, L);
        if (rc != GSP_OK) return rc;
        any = true;
      } else if (has_work(L)) {  // only batches to fold
        int rc = lane_collect(ctx, L);
        if (rc != GSP_OK) return rc;
        any = any || has_work(L);
      }
    }
    bool more = false;
    for (uint32_t l = 0; l < ctx->num_lanes; ++l) more = more || has_work(ctx->lanes[l]) || (drain && ctx->lanes[l].queued);
    if (!more) break;
    if (!any) break;
  }
  // leave with nothing in flight: the host's view of every lane is exact again
  for (uint32_t l = 0; l < ctx->num_lanes; ++l) {
    gsp_context::Lane& L = ctx->lanes[l];
    while (L.pipe.active && L.queued) {
      int rc = lane_collect(ctx, L);
      if (rc != GSP_OK) return rc;
    }
  }
  for (uint32_t l = 0; l < ctx->num_lanes; ++l) CTX_TRY(ctx, hipStreamSynchronize(ctx->lanes[l].stream));
  if (drain) {
    for (uint32_t l = 0; l < ctx->num_lanes; ++l)
      if (ctx->lanes[l].pipe.active) ctx->folded_idle = ctx->lanes[l].pipe.folded_end;
    ctx->pipe_active = false;
    for (uint32_t l = 0; l < ctx->num_lanes; ++l) ctx->lanes[l].pipe.active = false;
  }
  ctx->stats.render_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
  return GSP_OK;
}

// Completes everything gsp_render has queued.
static int pipeline_drain(gsp_context* ctx) { return pipeline_run(ctx, true); }

int gsp_render(gsp_context* ctx, const gsp_render_params* rp) {
  if (!ctx || !rp) return GSP_ERR_INVALID;
  if (!ctx->have_scene || !ctx->have_frame) {
    ctx->err = "gsp_render needs gsp_upload_scene and gsp_frame_begin first";
    return GSP_ERR_INVALID;
  }
  if (rp->max_depth > 250) {
    ctx->err = "max_depth > 250 unsupported";
    return GSP_ERR_INVALID;
  }
  CTX_TRY(ctx, hipSetDevice(ctx->device));
  if (rp->spp == 0 || ctx->num_pixels == 0) return GSP_OK;
  // A running pipeline is continued when the integrator constants and the batch size are unchanged
  // (paths in flight carry no copy of them); otherwise it is drained first.
  if (ctx->pipe_active &&
      (ctx->pipe_params.max_depth != rp->max_depth || ctx->pipe_params.rr_start_depth != rp->rr_start_depth ||
       ctx->pipe_params.clamp != rp->clamp || ctx->pipe_params.timestamps_in_flight != rp->timestamps_in_flight ||
       (ctx->pipe_params.disable_nee != 0) != (rp->disable_nee != 0))) {
    int rc = pipeline_drain(ctx);
    if (rc != GSP_OK) return rc;
  }
  if (rp->collect_traversal_stats != 0 && ctx->split) {
    // the statistics instantiations of k_trace walk ONE tree: the queued samples finish, the two trees are built as one again
    int rc_ = pipeline_drain(ctx);
    if (rc_ == GSP_OK) {
      ctx->have_scene = false;
      rc_ = bake_and_build(ctx);
      ctx->have_scene = rc_ == GSP_OK;
      ctx->split_declined = true;  // (a host that asks for statistics gets the scene in one piece from here on)
    }
    if (rc_ != GSP_OK) return rc_;
  }
  if (rp->collect_traversal_stats != 0 && ctx->pipe_active &&
      (ctx->oldest_live_version() != ctx->tab.ver || ctx->oldest_live_geo() != ctx->geo.ver)) {
    // the statistics instantiations of k_trace know one version of the scene: samples of older ones finish first
    int rc_ = pipeline_drain(ctx);
    if (rc_ != GSP_OK) return rc_;
  }
  if (!ctx->pipe_active) {
    if (ctx->split) ctx->geo.base = ctx->geo.phys(ctx->geo.ver);  // (nothing in flight: stamp 0 = the current version again)
    // Streaming path pool, per lane.  Samples enter in batches of Kb timestamps (>= ~1M paths); a new batch
    // is injected whenever the pool has room, so every launch works on millions of paths even though 95 %
    // of a batch dies at the Russian-roulette depth and a few stragglers live for 52 bounces.  Each batch
    // owns a slot of the sample-result ring and is folded into the accumulate buffer, in timestamp
    // order, once its live count has dropped to zero.
    // paths in flight over all lanes: per-launch fixed costs (drained wave tails, launch gaps) amortise over the pool
    // size -- r01 bench scan: 8 M 5.35, 12 M 5.76, 24 M 6.11, 48 M 6.16, 96 M 6.25 Grays/s; r02 (final kernels): 32 M 7.48,
    // 48 M 7.63, 64 M 7.64 (profiles/r02_ab_pool_size.txt); r03 (faster kernels, so the fixed cost per launch weighs more): 32 M 7.84,
    // 48 M 7.98, 64 M 8.09, 96 M 8.24, 128 M 8.21 (profiles/r03_ab_pool_size.txt).  96 M paths: 43 GB of queues (capacity 2 x the target)
    const uint64_t total_target = ctx->opt.pool_paths;  // (default 96 Mi)
    for (uint32_t l = 0; l < ctx->num_lanes; ++l) {
      gsp_context::Lane& L = ctx->lanes[l];
      gsp_context::Pipeline& P = L.pipe;
      P = gsp_context::Pipeline{};
      P.folded_end = ctx->folded_idle;
      const uint64_t npix = L.num_pixels;
      if (npix == 0) continue;  // fewer pixels than lanes
      uint64_t Kb = rp->timestamps_in_flight;
      if (Kb == 0) Kb = std::max<uint64_t>(1, (1ull << 20) / npix);
      while (Kb > 1 && Kb * npix >= (1ull << 30)) --Kb;
      P.Kb = Kb;
      P.batch_paths = Kb * npix;
      // (at most 192 samples per pixel in flight for a whole frame: tiny frames do not allocate gigabytes and gsp_peek does not
      // lag far behind; 384 for a pixel SUBSET -- a tile share of a multi-GPU frame: the 1/8 share of a 1080p frame, 259 k
      // pixels, must still fill the 96 Mi-path pool, else every GPU of an 8-GPU job runs launches half the size of the
      // single-GPU run's, profiles/r04_share_probe.txt)
#ifndef GSP_SHARE_SPP
#define GSP_SHARE_SPP 384
#endif
      const uint64_t per_pixel = ctx->subset ? GSP_SHARE_SPP : 192;
      P.pool_target = std::max<uint64_t>(std::min<uint64_t>(total_target / ctx->num_lanes, per_pixel * npix), 2 * P.batch_paths);
      uint64_t ring_bytes = ctx->opt.ring_bytes;  // (default 16 GiB)
      {
        // Several contexts may share one GPU (the shares of gsp_multi on a test box, two viewers, ...): this pipeline
        // takes at most gsp_ctx_options.memory_share (default 40 %) of the device's memory (plus what the lane already holds).  208 B of queues per
        // path of capacity (2 x 64-B path records, 2 x 16-B hit, 48-B shadow record), capacity = 2 x the pool target.
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
          const uint64_t have = L.pool_cap * 208ull + L.result_cap * sizeof(q4);
          // r05: the share is of the device's TOTAL memory (two contexts created one after the other get the same budget --
          // until r04 it was a share of what was FREE at that moment, so the second one sized itself by the first), and never
          // more than 90 % of what is free now
          const uint64_t budget = std::min((uint64_t)((double)total_b * ctx->memory_share), (uint64_t)((double)free_b * 0.9)) + have;
          const uint64_t queues = (2 * P.pool_target + P.batch_paths) * 208ull;
          if (queues > budget / 2) {
            const uint64_t fit = budget / 2 / 208ull;  // paths of capacity that fit
            P.pool_target = std::max<uint64_t>(2 * P.batch_paths, fit > P.batch_paths ? (fit - P.batch_paths) / 2 : 0);
          }
          ring_bytes = std::min<uint64_t>(ring_bytes, std::max<uint64_t>(budget / 2, 4 * P.batch_paths * sizeof(q4)));
        }
      }
      P.cap = 2 * P.pool_target + P.batch_paths;  // survivors (<= the previous queue) + a pool target's worth of new paths
      // Slots of the sample-result ring: a batch holds its slot until its last path has ended, so with paths of
      // ~3 bounces on average and a tail of 52 the alive share of the batches in flight is only a few percent and
      // the ring must hold ~24 x the pool for the pool to fill (coffee: 4.2 rays per sample).  16 B per entry, most
      // of it never touched on scenes with long paths; bounded by gsp_ctx_options.ring_bytes (default 16 GiB of the 288).
      const uint64_t want_slots = 24 * ((P.pool_target + P.batch_paths - 1) / P.batch_paths);
      const uint64_t fit_slots = ring_bytes / ctx->num_lanes / (P.batch_paths * sizeof(q4));
      P.num_slots = (uint32_t)std::min<uint64_t>(kMaxSlots, std::max<uint64_t>(4, std::min(want_slots, fit_slots)));
      if (P.cap >= (1ull << 32) || (uint64_t)P.num_slots * P.batch_paths >= (1ull << 32)) {
        ctx->err = "frame too large for 32-bit path indices";
        return GSP_ERR_INVALID;
      }
      int rc = ensure_pool(ctx, L, P.cap, (uint64_t)P.num_slots * P.batch_paths);
      if (rc != GSP_OK) return rc;
      P.slot_used.assign(P.num_slots, 0);
      P.active = true;
    }
    ctx->pipe_active = true;
  }
  ctx->pipe_params = *rp;  // (stats / timing flags may change from call to call)
  if (rp->collect_traversal_stats >= 2) {
    // the visit histograms belong to the context, every lane's k_trace adds to them on its own stream: allocate and zero them
    // here, on the context's stream, and wait -- not on the stream of whichever lane happens to enqueue first
    const size_t nn = std::max<size_t>(ctx->bvh.num_nodes, 1), ns = (size_t)ctx->bvh.num_tris + ctx->bvh.first_slot + kWide;
    if (ctx->node_hist.count < nn || ctx->tri_hist.count < ns) {
      CTX_TRY(ctx, ctx->node_hist.ensure(nn, &ctx->bytes));
      CTX_TRY(ctx, ctx->tri_hist.ensure(ns, &ctx->bytes));
      CTX_TRY(ctx, hipMemsetAsync(ctx->node_hist.p, 0, ctx->node_hist.count * sizeof(uint32_t), ctx->stream));
      CTX_TRY(ctx, hipMemsetAsync(ctx->tri_hist.p, 0, ctx->tri_hist.count * sizeof(uint32_t), ctx->stream));
      CTX_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
  }
  for (uint32_t l = 0; l < ctx->num_lanes; ++l) {
    gsp_context::Pipeline& P = ctx->lanes[l].pipe;
    if (!P.active) continue;
    P.next_ts = rp->first_timestamp;
    P.remaining += rp->spp;
  }
  return pipeline_run(ctx, false);
}

#ifdef GSP_SHADE_PROFILE
// measurement build only (scripts/shade_lane_profile.py): read and clear the lane profile of k_shade
extern "C" void gsp_debug_shade_profile(unsigned long long* out) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(gsp::g_shade_profile), PR_COUNT * 4 * sizeof(unsigned long long));
  static const unsigned long long zero[PR_COUNT * 4] = {};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(gsp::g_shade_profile), zero, sizeof(zero));
}
#endif
#ifdef GSP_WAVE_PROFILE
// measurement build only (scripts/trace_phase_budget.py): read and clear the 2 x 24 counters of pt_wavetrace.h
extern "C" void gsp_debug_wave_profile(unsigned long long* out) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(gsp::g_wave_profile), 48 * sizeof(unsigned long long));
  static const unsigned long long zero[48] = {};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(gsp::g_wave_profile), zero, sizeof(zero));
}
#endif

int gsp_sync(gsp_context* ctx) {
  if (!ctx) return GSP_ERR_INVALID;
  CTX_TRY(ctx, hipSetDevice(ctx->device));
  {
    int rc = pipeline_drain(ctx);
    if (rc != GSP_OK) return rc;
  }
  CTX_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return GSP_OK;
}

// the two pinned staging buffers of the frame read-back (12 ms the first time: gsp_frame_begin makes them with the frame's other buffers)
static int ensure_read_back_stage(gsp_context* ctx) {
  for (int k = 0; k < 2; ++k) {
    if (!ctx->h_stage[k]) CTX_TRY(ctx, hipHostMalloc((void**)&ctx->h_stage[k], gsp_context::kStageBytes, hipHostMallocDefault));
    if (!ctx->stage_ev[k]) CTX_TRY(ctx, hipEventCreateWithFlags(&ctx->stage_ev[k], hipEventDisableTiming));
  }
  return GSP_OK;
}
// the accumulate buffer -> caller's host memory, through the two pinned staging buffers (gsp_context::h_stage)
static int read_back_frame(gsp_context* ctx, float* out) {
  const size_t total = ctx->num_pixels * sizeof(q4);
  if (total <= gsp_context::kStageBytes / 4) {  // small frames: one plain copy
    CTX_TRY(ctx, hipMemcpyAsync(out, ctx->accum.p, total, hipMemcpyDeviceToHost, ctx->stream));
    CTX_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GSP_OK;
  }
  {
    int rc_ = ensure_read_back_stage(ctx);
    if (rc_ != GSP_OK) return rc_;
  }
  const size_t chunks = (total + gsp_context::kStageBytes - 1) / gsp_context::kStageBytes;
  const uint8_t* src = (const uint8_t*)ctx->accum.p;
  auto bytes_of = [&](size_t c) { return std::min(gsp_context::kStageBytes, total - c * gsp_context::kStageBytes); };
  CTX_TRY(ctx, hipMemcpyAsync(ctx->h_stage[0], src, bytes_of(0), hipMemcpyDeviceToHost, ctx->stream));
  CTX_TRY(ctx, hipEventRecord(ctx->stage_ev[0], ctx->stream));
  for (size_t c = 0; c < chunks; ++c) {
    if (c + 1 < chunks) {  // (buffer (c + 1) & 1 was drained by the memcpy of chunk c - 1, below, before this point)
      CTX_TRY(ctx, hipMemcpyAsync(ctx->h_stage[(c + 1) & 1], src + (c + 1) * gsp_context::kStageBytes, bytes_of(c + 1), hipMemcpyDeviceToHost, ctx->stream));
      CTX_TRY(ctx, hipEventRecord(ctx->stage_ev[(c + 1) & 1], ctx->stream));
    }
    CTX_TRY(ctx, hipEventSynchronize(ctx->stage_ev[c & 1]));
    std::memcpy((uint8_t*)out + c * gsp_context::kStageBytes, ctx->h_stage[c & 1], bytes_of(c));
  }
  return GSP_OK;
}

int gsp_download_compact(gsp_context* ctx, float* out) {
  if (!ctx || !out || !ctx->have_frame) return GSP_ERR_INVALID;
  CTX_TRY(ctx, hipSetDevice(ctx->device));
  {
    int rc_ = pipeline_drain(ctx);
    if (rc_ != GSP_OK) return rc_;
  }
  return read_back_frame(ctx, out);
}

int gsp_peek(gsp_context* ctx, float* out, uint32_t* samples_folded) {
  if (!ctx || !out || !ctx->have_frame) return GSP_ERR_INVALID;
  CTX_TRY(ctx, hipSetDevice(ctx->device));
  // no drain: the folds queued so far finish, the paths in flight keep their state
  uint32_t folded = 0xffffffffu;
  for (uint32_t l = 0; l < ctx->num_lanes; ++l) {
    gsp_context::Lane& L = ctx->lanes[l];
    if (L.num_pixels == 0) continue;
    CTX_TRY(ctx, hipStreamSynchronize(L.stream));
    folded = std::min(folded, ctx->pipe_active && L.pipe.active ? L.pipe.folded_end : ctx->folded_idle);
  }
  {
    int rc_ = read_back_frame(ctx, out);
    if (rc_ != GSP_OK) return rc_;
  }
  if (samples_folded) *samples_folded = folded == 0xffffffffu ? 0u : folded;
  return GSP_OK;
}

int gsp_peek_to_device(gsp_context* ctx, void* dst, uint64_t bytes, uint32_t* samples_folded) {
  if (!ctx || !dst || !ctx->have_frame) return GSP_ERR_INVALID;
  CTX_TRY(ctx, hipSetDevice(ctx->device));
  if (bytes < ctx->num_pixels * sizeof(q4)) {
    ctx->err = "destination too small";
    return GSP_ERR_INVALID;
  }
  uint32_t folded = 0xffffffffu;
  for (uint32_t l = 0; l < ctx->num_lanes; ++l) {  // (as gsp_peek: the folds queued so far finish, the paths in flight stay)
    gsp_context::Lane& L = ctx->lanes[l];
    if (L.num_pixels == 0) continue;
    CTX_TRY(ctx, hipStreamSynchronize(L.stream));
    folded = std::min(folded, ctx->pipe_active && L.pipe.active ? L.pipe.folded_end : ctx->folded_idle);
  }
  CTX_TRY(ctx, hipMemcpyAsync(dst, ctx->accum.p, ctx->num_pixels * sizeof(q4), hipMemcpyDeviceToDevice, ctx->stream));
  CTX_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (samples_folded) *samples_folded = folded == 0xffffffffu ? 0u : folded;
  return GSP_OK;
}

int gsp_download(gsp_context* ctx, float* out) {
  if (!ctx || !out || !ctx->have_frame) return GSP_ERR_INVALID;
  if (!ctx->subset) return gsp_download_compact(ctx, out);
  std::vector<float> tmp(4ull * ctx->num_pixels);
  int rc = gsp_download_compact(ctx, tmp.data());
  if (rc != GSP_OK) return rc;
  std::memset(out, 0, sizeof(float) * 4ull * ctx->width * ctx->height);
  for (uint64_t i = 0; i < ctx->num_pixels; ++i)
    std::memcpy(out + 4ull * ctx->pixel_ids_host[i], tmp.data() + 4ull * i, 4 * sizeof(float));
  return GSP_OK;
}

int gsp_copy_accum_to_device(gsp_context* ctx, void* dst, uint64_t bytes) {
  if (!ctx || !dst || !ctx->have_frame) return GSP_ERR_INVALID;
  CTX_TRY(ctx, hipSetDevice(ctx->device));
  {
    int rc_ = pipeline_drain(ctx);
    if (rc_ != GSP_OK) return rc_;
  }
  if (bytes < ctx->num_pixels * sizeof(q4)) {
    ctx->err = "destination too small";
    return GSP_ERR_INVALID;
  }
  CTX_TRY(ctx, hipMemcpyAsync(dst, ctx->accum.p, ctx->num_pixels * sizeof(q4), hipMemcpyDeviceToDevice, ctx->stream));
  CTX_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return GSP_OK;
}

}  // extern "C"

int gsp::gsp_internal_accum(gsp_context* ctx, void** accum, uint64_t* num_pixels, hipStream_t* stream) {
  if (!ctx || !ctx->have_frame) return GSP_ERR_INVALID;
  CTX_TRY(ctx, hipSetDevice(ctx->device));
  int rc = pipeline_drain(ctx);
  if (rc != GSP_OK) return rc;
  CTX_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (accum) *accum = ctx->accum.p;
  if (num_pixels) *num_pixels = ctx->num_pixels;
  if (stream) *stream = ctx->stream;
  return GSP_OK;
}

extern "C" {

int gsp_upload_accum(gsp_context* ctx, const float* rgba, uint64_t num_pixels) {
  if (!ctx || !rgba || !ctx->have_frame || num_pixels != ctx->num_pixels) return GSP_ERR_INVALID;
  CTX_TRY(ctx, hipSetDevice(ctx->device));
  {
    int rc_ = pipeline_drain(ctx);
    if (rc_ != GSP_OK) return rc_;
  }
  CTX_TRY(ctx, hipMemcpyAsync(ctx->accum.p, rgba, num_pixels * sizeof(q4), hipMemcpyHostToDevice, ctx->stream));
  CTX_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return GSP_OK;
}

int gsp_get_stats(gsp_context* ctx, gsp_stats* out) {
  if (!ctx || !out) return GSP_ERR_INVALID;
  CTX_TRY(ctx, hipSetDevice(ctx->device));
  {
    int rc_ = pipeline_drain(ctx);
    if (rc_ != GSP_OK) return rc_;
  }
  if (ctx->dstats.p) {
    DevStats d;
    CTX_TRY(ctx, hipStreamSynchronize(ctx->stream));
    CTX_TRY(ctx, hipMemcpy(&d, ctx->dstats.p, sizeof(d), hipMemcpyDeviceToHost));
    ctx->stats.shaded_vertices = d.shaded;
    ctx->stats.nodes_visited = d.nodes;
    ctx->stats.tris_tested = d.tris;
    ctx->stats.stat_rays = d.stat_rays;
    ctx->stats.shadow_nodes_visited = d.sh_nodes;
    ctx->stats.shadow_tris_tested = d.sh_tris;
    ctx->stats.shadow_stat_rays = d.sh_rays;
    ctx->stats.nodes_from_lds = d.lds_nodes;
    ctx->stats.shadow_nodes_from_lds = d.sh_lds_nodes;
    ctx->stats.shadow_stat_occluded = d.sh_occluded;
    ctx->stats.shadow_stat_occluded_nodes = d.sh_occluded_nodes;
    ctx->stats.shadow_stat_no_triangle = d.sh_no_tri;
  }
  ctx->stats.bvh_build_ms = ctx->bvh_build_ms;
  ctx->stats.num_triangles = ctx->bvh.num_tris;  // (a split scene: the static tree keeps a slot for every triangle of the scene)
  ctx->stats.num_bvh_nodes = ctx->bvh.num_nodes + (ctx->split ? ctx->dyn.num_nodes : 0u);
  ctx->stats.bvh_depth = std::max(ctx->bvh.depth, ctx->split ? ctx->dyn.depth : 0u);
  ctx->stats.device_bytes = ctx->bytes;
  ctx->stats.algorithmic_bytes = 48ull * ctx->stats.stat_rays + (uint64_t)kNodeBytes * ctx->stats.nodes_visited + 48ull * ctx->stats.tris_tested +
                                 32ull * ctx->stats.shadow_stat_rays + 36ull * ctx->stats.shadow_stat_occluded +
                                 (uint64_t)kNodeBytes * ctx->stats.shadow_nodes_visited + 48ull * ctx->stats.shadow_tris_tested;
  *out = ctx->stats;
  return GSP_OK;
}

int gsp_reset_stats(gsp_context* ctx) {
  if (!ctx) return GSP_ERR_INVALID;
  CTX_TRY(ctx, hipSetDevice(ctx->device));
  {
    int rc_ = pipeline_drain(ctx);
    if (rc_ != GSP_OK) return rc_;
  }
  const uint64_t updates = ctx->stats.scene_updates, refits = ctx->stats.scene_refits, drains = ctx->stats.scene_drains,
                 splits = ctx->stats.scene_splits;  // (counts since the last gsp_upload_scene, not since the last reset)
  ctx->stats = gsp_stats{};
  ctx->stats.scene_updates = updates;
  ctx->stats.scene_refits = refits;
  ctx->stats.scene_drains = drains;
  ctx->stats.scene_splits = splits;
  if (ctx->dstats.p) {
    CTX_TRY(ctx, hipMemsetAsync(ctx->dstats.p, 0, sizeof(DevStats), ctx->stream));
    CTX_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  return GSP_OK;
}

int gsp_debug_visit_histograms(gsp_context* ctx, uint32_t* node_counts, uint64_t num_nodes, uint32_t* slot_counts, uint64_t num_slots) {
  if (!ctx) return GSP_ERR_INVALID;
  CTX_TRY(ctx, hipSetDevice(ctx->device));
  {
    int rc_ = pipeline_drain(ctx);
    if (rc_ != GSP_OK) return rc_;
  }
  if (!ctx->node_hist.p || !ctx->tri_hist.p) {
    ctx->err = "gsp_debug_visit_histograms: no render with collect_traversal_stats = 2 since the last call";
    return GSP_ERR_INVALID;
  }
  if (node_counts) CTX_TRY(ctx, hipMemcpy(node_counts, ctx->node_hist.p, std::min<uint64_t>(num_nodes, ctx->node_hist.count) * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (slot_counts) CTX_TRY(ctx, hipMemcpy(slot_counts, ctx->tri_hist.p, std::min<uint64_t>(num_slots, ctx->tri_hist.count) * sizeof(uint32_t), hipMemcpyDeviceToHost));
  ctx->bytes -= (ctx->node_hist.count + ctx->tri_hist.count) * sizeof(uint32_t);
  ctx->node_hist.release();
  ctx->tri_hist.release();
  return GSP_OK;
}

int gsp_trace(gsp_context* ctx, const float* rays, uint64_t n, int any_hit, void* hits) {
  if (!ctx || (!rays && n) || (!hits && n)) return GSP_ERR_INVALID;
  if (!ctx->have_scene) {
    ctx->err = "gsp_trace needs gsp_upload_scene first";
    return GSP_ERR_INVALID;
  }
  if (n == 0) return GSP_OK;
  if (n >= (1ull << 31)) return GSP_ERR_INVALID;
  CTX_TRY(ctx, hipSetDevice(ctx->device));
  DevBuf<float>& d_rays = ctx->trace_rays;
  DevBuf<q4>& d_hits = ctx->trace_hits;
  DevBuf<uint32_t>& d_work = ctx->trace_work;
  CTX_TRY(ctx, d_rays.upload(rays, 8 * n, ctx->stream, nullptr));
  CTX_TRY(ctx, d_hits.ensure(n, nullptr));
  CTX_TRY(ctx, d_work.ensure(kWorkShards * kWorkStride, nullptr));
  CTX_TRY(ctx, hipMemsetAsync(d_work.p, 0, kWorkShards * kWorkStride * sizeof(uint32_t), ctx->stream));
  if (ctx->split && !ctx->s2g_all_valid) {
    int rc_ = make_split_s2g(ctx);
    if (rc_ != GSP_OK) return rc_;
  }
  const SceneView view = ctx->view();
  const TestIO io{d_rays.p, d_hits.p, ctx->split ? ctx->s2g_all.p : ctx->bvh.slot_to_global, any_hit,
                  ctx->bvh.num_tris};
  const TraceStatsOut none{nullptr, nullptr, nullptr};
  if (ctx->split) {  // both trees, the newest version of the edited one
    TestSplitIO sio;
    static_cast<TestIO&>(sio) = io;
    sio.g = GeoRing{view.geo, 0u, ctx->static_slots};
    sio.stamp = ctx->geo.phys(ctx->geo.ver + ctx->geo.slots() - ctx->geo.base);
    if (any_hit)
      hipLaunchKernelGGL((k_trace<true, false, TestSplitIO>), dim3(ctx->trace_grid(n, kChunkSmall, true)), dim3(kTraceBlock), 0, ctx->stream,
                         view.nodes, view.tri_isect, (const uint32_t*)nullptr, (uint32_t)n, 0u, kChunkSmall, sio, d_work.p,
                         ctx->lanes[0].spill.p, ctx->spill_stride, none);
    else
      hipLaunchKernelGGL((k_trace<false, false, TestSplitIO>), dim3(ctx->trace_grid(n, kChunkSmall)), dim3(kTraceBlock), 0, ctx->stream,
                         view.nodes, view.tri_isect, (const uint32_t*)nullptr, (uint32_t)n, 0u, kChunkSmall, sio, d_work.p,
                         ctx->lanes[0].spill.p, ctx->spill_stride, none);
  } else if (any_hit)
    hipLaunchKernelGGL((k_trace<true, false, TestIO>), dim3(ctx->trace_grid(n, kChunkSmall, true)), dim3(kTraceBlock), 0, ctx->stream,
                       view.nodes, view.tri_isect, (const uint32_t*)nullptr, (uint32_t)n, 0u, kChunkSmall, io, d_work.p,
                       ctx->lanes[0].spill.p, ctx->spill_stride, none);
  else
    hipLaunchKernelGGL((k_trace<false, false, TestIO>), dim3(ctx->trace_grid(n, kChunkSmall)), dim3(kTraceBlock), 0, ctx->stream,
                       view.nodes, view.tri_isect, (const uint32_t*)nullptr, (uint32_t)n, 0u, kChunkSmall, io, d_work.p,
                       ctx->lanes[0].spill.p, ctx->spill_stride, none);
  CTX_TRY(ctx, hipGetLastError());
  CTX_TRY(ctx, hipMemcpyAsync(hits, d_hits.p, n * sizeof(q4), hipMemcpyDeviceToHost, ctx->stream));
  CTX_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (n > (8u << 20)) {  // a one-off large query does not pin half a gigabyte for the life of the context
    d_rays.release();
    d_hits.release();
  }
  return GSP_OK;
}

}  // extern "C"
