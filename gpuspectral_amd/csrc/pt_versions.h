// pt_versions.h -- the version-ring bookkeeping of the per-frame scene edits (gsp_update_tables / gsp_update_instances without a
// drain, pt_render.hip), as PURE host logic: no HIP call, no device pointer, no allocation.  pt_render.hip asks these structs
// which slot to write, whether a wait is needed and how large a ring may be, and performs the copies / launches itself;
// tests/emu/versions_model.cpp drives the same structs from a mock pipeline on the CPU (random edit streams under ASan / UBSan:
// no slot is written while a sample that names it is in flight, the byte ledger never wraps, split / ring limits hold).
//
// Behaviour contract (S/renderer/PathTracer.cpp:58-93): the reference re-reads instances, tables and camera every frame, so an
// edit applies to the samples generated AFTER it; the samples in flight finish on the scene they were generated under.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>

namespace gsp {

constexpr uint32_t kMaxTableVersions = 64;  // = kTableVersions (pt_stages.h): the width of the version field of the path flags
constexpr uint32_t kMaxGeoVersions = 64;    // = kGeoVersions
// gsp_update_instances: how often a scene may be split again because an edit touched an instance not edited before (each
// re-split waits for the samples in flight and builds two trees); after that the scene lives in a ring of whole trees
// (include/gpuspectral_pt.h, "Per-frame edits"; gsp_stats.scene_splits)
constexpr uint32_t kMaxSceneSplits = 16;

// device bytes a context accounts for (gsp_stats.device_bytes): a subtraction never wraps -- an accounting slip shows up as
// `underflows`, which the model test and the GPU suite assert to be 0
struct ByteLedger {
  size_t bytes = 0;
  uint64_t underflows = 0;
  void add(size_t n) { bytes += n; }
  void sub(size_t n) {
    if (n > bytes) {
      ++underflows;
      bytes = 0;
    } else {
      bytes -= n;
    }
  }
};

// ---- the ring of BSDF / light table versions ------------------------------------------------------------------------------
// Version v of the tables sits in slot (v - rot) % slots; a sample carries that slot number in its path flags.  While only ONE
// version is live it sits in slot 0 (rot == ver, flags field 0).  r06: the ring is made LAZILY -- gsp_upload_scene allocates one
// slot; the first gsp_update_tables that arrives with samples in flight grows it to `slots_for(...)` slots (the live version is
// copied into slot 0 of the new allocation, the old one is released at the next idle point), capped by bytes: a 1 M-triangle
// mesh emitter (64 MB of light records) gets a ring of 4, not 64 x 64 MB at upload (r05 review, ADVICE medium).
struct TableRing {
  uint32_t slots = 1;     // allocated
  size_t slot_bytes = 0;  // bytes of one slot (256-B aligned image size)
  uint32_t ver = 0;       // version the next sample is generated under (monotonic)
  uint32_t rot = 0;       // version that slot 0 / flags field 0 stand for

  uint32_t slot_of(uint32_t v) const { return (v - rot) % slots; }
  uint32_t current_slot() const { return slot_of(ver); }
  size_t allocation_bytes() const { return slot_bytes * slots; }

  // slots of a grown ring: as many as the flags field can name, at most `budget` bytes (a sixteenth of the free device memory,
  // at most 1 GiB, says the caller); fewer than 2 = no ring to be had
  static uint32_t slots_for(size_t slot_bytes, size_t budget) {
    if (slot_bytes == 0) return kMaxTableVersions;
    const size_t fit = budget / slot_bytes;
    return (uint32_t)std::min<size_t>(kMaxTableVersions, fit);
  }
  static size_t budget_for(size_t free_bytes) { return std::min<size_t>(free_bytes / 16, (size_t)1 << 30); }

  enum class Update {
    kInPlace,          // the new image goes into the slot of ver + 1; no wait
    kGrowThenInPlace,  // the ring has one slot: grow it (on_grown), then as kInPlace
    kDrain,            // the queued samples finish first, then the image replaces the (only live) version
  };
  // same_layout: same record counts and image size as the resident tables; caps_ok: the path pools are small enough for the
  // <VER> kernels' queue-index flag; oldest_live: oldest version a sample in flight carries (== ver when none is)
  Update decide(bool pipe_active, bool same_layout, bool caps_ok, uint32_t oldest_live, size_t free_bytes) const {
    if (!pipe_active || !same_layout || !caps_ok) return Update::kDrain;
    if (slots == 1) return slots_for(slot_bytes, budget_for(free_bytes)) >= 2 ? Update::kGrowThenInPlace : Update::kDrain;
    return ver + 1 - oldest_live < slots ? Update::kInPlace : Update::kDrain;
  }
  // the image of a NEW layout, or any image behind a drain: one live version, in slot 0.  -> true when the allocation must change
  bool upload_behind_drain(size_t new_slot_bytes) {
    const bool realloc = new_slot_bytes != slot_bytes;
    if (realloc) {  // (a new layout starts with one slot again: the ring is made when an edit asks for it)
      slot_bytes = new_slot_bytes;
      slots = 1;
    }
    rot = ver;
    return realloc;
  }
  // the ring has been re-allocated with `n` slots and the live version copied into its slot 0
  void on_grown(uint32_t n) {
    slots = n;
    rot = ver;
  }
  // kInPlace: -> the slot the new image is written to; it becomes the current version
  uint32_t begin_next_version() {
    ++ver;
    return current_slot();
  }
  // every sample in flight belongs to `ver` again and the <VER = false> kernels are about to run: the live version moves into
  // slot 0 (-> the slot it is copied FROM; 0 = it is there already)
  uint32_t collapse() {
    const uint32_t from = current_slot();
    rot = ver;
    return from;
  }
};

// ---- the ring of geometry versions ------------------------------------------------------------------------------------------
// 2^log2 slots of `stride` triangle slots each (node records, intersection triangles, shading packets); version v sits in slot
// v & (slots - 1); a path's stamp s names slot (s + base) & (slots - 1), base = the slot of the one live version the last time
// only one was live (pt_stages.h geo_slot_offset).
struct GeoVersions {
  uint32_t stride = 0;  // triangle slots per version; 0 = no ring
  uint32_t log2 = 0;
  uint32_t ver = 0;   // version the next sample is generated under (monotonic)
  uint32_t base = 0;  // slot that stamp 0 stands for

  uint32_t slots() const { return 1u << log2; }
  uint32_t phys(uint32_t v) const { return v & (slots() - 1u); }
  // the stamp k_generate writes into new samples
  uint32_t stamp_of_current() const { return phys(ver + slots() - base); }
  // may version ver + 1 be written while samples as old as oldest_live are in flight?
  bool next_slot_free(uint32_t oldest_live) const { return stride != 0 && ver + 1 - oldest_live < slots(); }
  // one version live, the plain kernels run: stamp 0 = that version from here on
  void on_single_version() { base = phys(ver); }
  void reset() { *this = GeoVersions{}; }
};

// Size of a geometry ring.  slots_static: triangle slots of a tree every version shares in front of the ring (split scene; 0 = the
// ring holds whole trees); slots_ring: triangle slots of ONE version of the tree that goes through the ring; nodes_*: their node
// counts.  -> log2 of the versions (>= 2) or -1: no ring to be had.  Limits: 32-bit node byte offsets, 2^28 triangle slots, a
// quarter of the free device memory (176 B per triangle slot), the stride field of SceneView::geo.
struct GeoRingPlan {
  int log2 = -1;
  uint64_t stride_static = 0, stride_ring = 0, total_slots = 0;
};
inline GeoRingPlan plan_geo_ring(uint64_t slots_static, uint64_t nodes_static, uint64_t slots_ring, uint64_t nodes_ring, uint32_t want_versions,
                                 size_t free_bytes, uint64_t min_stride, uint64_t max_stride, uint32_t node_bytes) {
  GeoRingPlan p;
  p.stride_static = slots_static ? std::max<uint64_t>(slots_static, min_stride) : 0;
  p.stride_ring = std::max<uint64_t>(slots_ring, min_stride);
  if (p.stride_ring > max_stride || nodes_ring > p.stride_ring || nodes_static > p.stride_static) return p;
  uint32_t lg = 0;
  while ((2u << lg) <= std::min<uint32_t>(std::max<uint32_t>(want_versions, 1u), kMaxGeoVersions)) ++lg;
  auto total = [&](uint32_t l) { return p.stride_static + (p.stride_ring << l); };
  const uint64_t quad = 16;  // bytes of a q4
  while (lg > 0 && (total(lg) * node_bytes >= (1ull << 32) || total(lg) >= (1ull << 28) || total(lg) * 11 * quad > free_bytes / 4)) --lg;
  if (lg < 2) return p;
  p.log2 = (int)lg;
  p.total_slots = total(lg);
  return p;
}

// ---- gsp_update_instances: the wait / no-wait decisions, one predicate each (pt_render_scene.inc walks them in this order) ------
// `live`: samples are in flight (the pipeline is active); `caps_ok`: the path pools are small enough for the <VER> kernels'
// queue-index flag; oldest_live: the oldest geometry version a sample in flight carries (== geo.ver when none is).
//
// (1) split scene, the edit touches only instances that already live in the small tree: the refit goes into the NEXT slot of the ring
//     while the samples in flight finish in theirs -- else (no free slot, pools too large) they finish first and the refit is in place
inline bool edited_tree_refit_needs_no_wait(bool live, bool caps_ok, const GeoVersions& geo, uint32_t oldest_live) {
  return live && caps_ok && geo.next_slot_free(oldest_live);
}
// (2) the edit asks for a split (a first one, or a re-split because it touches an instance not edited before)
inline bool edit_wants_split(bool is_split, bool split_declined, bool live, double refit_growth, uint32_t geometry_versions, uint32_t num_textures) {
  return is_split || (!split_declined && live && refit_growth > 1.0 && geometry_versions >= 4 && num_textures == 0);
}
// ... and the FIRST split of a tree that has never been edited needs no wait: the small tree is built where the instances WERE as
// version 0 -- with the static tree that is the scene the samples in flight were generated under (they carry stamp 0) -- and the
// edit itself becomes version 1
inline bool first_split_needs_no_wait(bool is_split, bool live, bool caps_ok, const GeoVersions& geo) {
  return !is_split && live && geo.ver == 0 && geo.stride == 0 && caps_ok;
}
// (3) ring of whole trees (scenes that do not split): the refit goes into the next slot
inline bool whole_tree_refit_needs_no_wait(bool live, bool caps_ok, double refit_growth, const GeoVersions& geo, uint32_t oldest_live) {
  return live && geo.stride != 0 && refit_growth > 1.0 && caps_ok && geo.next_slot_free(oldest_live);
}
// ... and behind a wait the ring of whole trees is made only where a split is not to be had
inline bool scene_may_still_split(bool split_declined, uint32_t geometry_versions, uint32_t num_textures) {
  return !split_declined && geometry_versions >= 4 && num_textures == 0;
}

// gsp_update_instances, make_split: may this edit ask for a split (or re-split) of the scene?
inline bool may_split(bool split_declined, uint32_t splits_so_far, double refit_growth, uint32_t geometry_versions, uint32_t num_textures,
                      bool ring_failed) {
  return !split_declined && splits_so_far < kMaxSceneSplits && refit_growth > 1.0 && geometry_versions >= 4 && num_textures == 0 && !ring_failed;
}
// ... and is the scene one a split pays for: something edited, something not, at most a quarter of the triangles edited
inline bool split_worthwhile(uint64_t tris_static, uint64_t tris_edited) {
  return tris_edited != 0 && tris_static != 0 && tris_edited * 4 <= tris_static + tris_edited;
}

}  // namespace gsp
