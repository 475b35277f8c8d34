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
#include <cstddef>
#include <cstring>
#include <deque>
#include <mutex>
#include <vector>

#include "pt_hostmath.h"
#include "pt_internal.h"
#include "pt_versions.h"
#include "pt_wavetrace.h"

#include "pt_render_kernels.inc"  // namespace gsp { kernels, queue layouts, ray sources, DevBuf }


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
  bool stage_warm = false;  // the staging buffers have been through one copy + one host read (gsp_frame_begin)
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
  p->struct_size = (uint32_t)sizeof(*p);
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

#include "pt_render_scene.inc"     // scene upload and per-frame edits
#include "pt_render_pipeline.inc"  // the streaming pipeline and the rest of the C ABI
