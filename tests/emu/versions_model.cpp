// tests/emu/versions_model.cpp -- TEST HARNESS (CPU only; built by tests/test_versions_model.py with -fsanitize=address,undefined).
//
// Drives the product's version-ring bookkeeping (gpuspectral_amd/csrc/pt_versions.h: the structs pt_render.hip itself asks which
// slot to write and whether to wait) from a MOCK pipeline: random streams of gsp_render / gsp_update_tables /
// gsp_update_instances / drains, with batches of samples in flight that carry the stamps k_generate would write.  The mock
// "device memory" remembers which version every ring slot holds; every in-flight batch reads ITS slot on every iteration.
// Invariants (abort on violation):
//   1. a slot is never written while a batch in flight names it (tables and geometry);
//   2. a batch always reads the version it was generated under -- through collapses, ring growth, base changes and wrap-around;
//   3. the byte ledger never underflows and ends at 0 when everything is released;
//   4. ring plans respect the 32-bit node offsets, the 2^28 slot limit, the memory share and the stride field;
//   5. no more than kMaxSceneSplits re-splits; may_split / split_worthwhile say no where the documentation says they do.
// usage: versions_model <seed> <steps>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <random>
#include <vector>

#include "../../gpuspectral_amd/csrc/pt_versions.h"

using namespace gsp;

#define CHECK(c, ...)                                        \
  do {                                                       \
    if (!(c)) {                                              \
      std::fprintf(stderr, "versions_model: %s:%d: ", __FILE__, __LINE__); \
      std::fprintf(stderr, __VA_ARGS__);                     \
      std::fprintf(stderr, "\n");                            \
      std::abort();                                          \
    }                                                        \
  } while (0)

struct Batch {
  uint32_t tab_ver, tab_field;   // version of the tables it was generated under; the slot number in its path flags
  uint32_t geo_ver, geo_stamp;   // ... of the geometry; its stamp
};

struct Mock {
  std::mt19937 rng;
  TableRing tab;
  GeoVersions geo;
  ByteLedger ledger;
  std::vector<uint32_t> tab_mem;  // version held by each table slot (the allocation itself: its size is on the ledger)
  std::vector<uint32_t> geo_mem;  // version held by each geometry slot (empty: the tree owns its arrays = one implicit slot)
  uint32_t geo_owned_version = 0; // ... the version of that implicit slot
  std::deque<Batch> inflight;
  bool pipe_active = false;
  uint32_t retire_bias = 1;
  uint64_t drains = 0, in_place = 0, grown = 0, geo_in_ring = 0, collapses = 0;
  size_t free_mem = (size_t)200 << 30;

  uint32_t rnd(uint32_t n) { return (uint32_t)(rng() % n); }
  uint32_t oldest_tab() const { return inflight.empty() ? tab.ver : inflight.front().tab_ver; }
  uint32_t oldest_geo() const { return inflight.empty() ? geo.ver : inflight.front().geo_ver; }

  void alloc_tables(size_t slot_bytes, uint32_t slots) {
    ledger.add(slot_bytes * slots);
    tab_mem.assign(slots, 0xffffffffu);
  }
  void free_tables() {
    ledger.sub(tab.slot_bytes * tab_mem.size());
    tab_mem.clear();
  }

  void release_geo() {
    if (geo.stride) ledger.sub((size_t)geo.stride * geo.slots() * 176);
    geo.reset();
    geo_mem.clear();
    geo_owned_version = 0;
  }

  // gsp_upload_scene: behind a drain; the tables start in one slot (or keep a grown ring of the same slot size), the tree owns its arrays
  void upload_scene(size_t slot_bytes) {
    drain();
    if (!tab_mem.empty()) free_tables();
    release_geo();
    tab.upload_behind_drain(slot_bytes);
    alloc_tables(tab.slot_bytes, tab.slots);
    tab_mem[0] = tab.ver;
  }

  // every batch in flight reads its slot: invariant 2
  void iterate(bool multi_version) {
    for (const Batch& b : inflight) {
      const uint32_t ts = multi_version ? b.tab_field : tab.current_slot();
      CHECK(ts < tab_mem.size() && tab_mem[ts] == b.tab_ver, "batch of table version %u reads slot %u which holds %u", b.tab_ver, ts,
            ts < tab_mem.size() ? tab_mem[ts] : 0xdeadu);
      if (geo.stride != 0) {
        const uint32_t gs_ = multi_version ? ((b.geo_stamp + geo.base) & (geo.slots() - 1u)) : geo.phys(geo.ver);
        CHECK(geo_mem[gs_] == b.geo_ver, "batch of geometry version %u reads slot %u which holds %u", b.geo_ver, gs_, geo_mem[gs_]);
      } else {
        CHECK(geo_owned_version == b.geo_ver, "batch of geometry version %u reads the tree's own arrays, which hold %u", b.geo_ver, geo_owned_version);
      }
    }
  }

  void render() {
    // lane_enqueue: are the samples in flight of more than one version?
    const bool multi = oldest_tab() != tab.ver || oldest_geo() != geo.ver;
    if (!multi) {
      if (geo.stride) geo.on_single_version();
      if (tab.rot != tab.ver) {
        const uint32_t from = tab.collapse();
        if (from != 0) {
          for (const Batch& b : inflight) CHECK(b.tab_ver == tab.ver, "collapse with an older version in flight");
          tab_mem[0] = tab_mem[from];
        }
        ++collapses;
      }
      // the <VER = false> kernels write 0 into the version fields of every path they touch -- every path, each iteration
      for (Batch& b : inflight) b.tab_field = 0, b.geo_stamp = 0;
    }
    CHECK(tab.current_slot() < kMaxTableVersions && (!geo.stride || geo.stamp_of_current() < kMaxGeoVersions), "stamp does not fit its field");
    inflight.push_back(Batch{tab.ver, multi ? tab.current_slot() : 0u, geo.ver, multi && geo.stride ? geo.stamp_of_current() : 0u});
    if (!multi) {  // (the new batch's fields under the plain kernels: 0 -- and slot 0 / the current slot is what it reads)
      CHECK(tab.current_slot() == 0, "one live version must sit in slot 0 when the plain kernels run");
    }
    pipe_active = true;
    iterate(multi);
    // batches end in timestamp order (they are folded in that order); a few per iteration -- in phases: while retire_bias is 0 nothing
    // ends (deep paths: 52 bounces = 52 frames of a viewer) and the rings fill up to the point where an edit has to wait
    if (rnd(400) == 0) retire_bias = rnd(3);
    uint32_t k = rnd(1 + retire_bias + (retire_bias ? 1 : 0));
    while (k-- && inflight.size() > 1) inflight.pop_front();
  }

  void drain() {
    while (!inflight.empty()) {
      iterate(oldest_tab() != tab.ver || oldest_geo() != geo.ver);
      inflight.pop_front();
    }
    pipe_active = false;
    ++drains;
  }

  void write_table_slot(uint32_t slot, uint32_t version) {
    for (const Batch& b : inflight)  // (the fields are what the <VER> kernels will read them through from the next iteration on)
      CHECK(b.tab_field != slot, "table slot %u written while a batch of version %u names it", slot, b.tab_ver);
    tab_mem[slot] = version;
  }

  void update_tables() {
    const bool same_layout = rnd(8) != 0;
    const bool caps_ok = rnd(50) != 0;
    TableRing::Update how = tab.decide(pipe_active, same_layout, caps_ok, oldest_tab(), free_mem);
    if (how == TableRing::Update::kGrowThenInPlace) {
      const uint32_t n = TableRing::slots_for(tab.slot_bytes, TableRing::budget_for(free_mem));
      CHECK(n >= 2 && n <= kMaxTableVersions, "grown ring of %u slots", n);
      CHECK((size_t)n * tab.slot_bytes <= TableRing::budget_for(free_mem), "grown ring exceeds its byte budget");
      if (rnd(10) == 0) {
        how = TableRing::Update::kDrain;  // (no memory for it)
      } else {
        // the live version must be the only one in flight's slot 0: with one slot every edit so far waited
        for (const Batch& b : inflight) CHECK(b.tab_ver == tab.ver && b.tab_field == 0, "growth with another version in flight");
        const uint32_t live = tab_mem[0];
        free_tables();
        tab.on_grown(n);
        alloc_tables(tab.slot_bytes, n);
        tab_mem[0] = live;
        ++grown;
        how = TableRing::Update::kInPlace;
      }
    }
    if (how == TableRing::Update::kInPlace) {
      const uint32_t slot = tab.begin_next_version();
      write_table_slot(slot, tab.ver);
      ++in_place;
    } else {
      drain();
      // (another layout: from a few records to half a gigabyte of light records -- the byte budget then allows rings of 2 .. 64 slots)
      const size_t nb = same_layout ? tab.slot_bytes : (size_t)256 * (1 + rnd(1u << (4 + rnd(18))));
      if (tab.slot_bytes != nb) free_tables();
      if (tab.upload_behind_drain(nb)) alloc_tables(tab.slot_bytes, tab.slots);
      tab_mem[0] = tab.ver;  // (the image replaces the one live version: slot 0, same version number -- nothing in flight can tell)
    }
  }

  void update_instances() {
    if (geo.stride == 0) {
      // first edit of a tree: the ring is made behind a drain (plan_geo_ring), or there is none to be had
      drain();
      const uint64_t tris = 1000 + rnd(3000000);
      const GeoRingPlan p = plan_geo_ring(0, 0, tris + 7, tris / 2, 1u << rnd(7), free_mem, 256, (1u << 23) - 1, 64);
      ++geo_owned_version;
      ++geo.ver;
      if (p.log2 >= 2 && rnd(4) != 0) {
        CHECK((p.stride_ring << p.log2) * 64 < (1ull << 32), "node offsets beyond 32 bits");
        geo.stride = (uint32_t)p.stride_ring;
        geo.log2 = (uint32_t)p.log2;
        geo.ver = 0;
        geo.base = 0;
        geo_mem.assign(geo.slots(), 0xffffffffu);
        geo_mem[0] = 0;
        ledger.add((size_t)(p.stride_ring << p.log2) * 176);
      }
      return;
    }
    const bool caps_ok = rnd(50) != 0;
    // (the product asks edited_tree_refit_needs_no_wait for a split scene's small tree and whole_tree_refit_needs_no_wait for a ring of
    // whole trees: the same slot test behind different preconditions -- the model's ring stands for either)
    const bool no_wait = rnd(2) ? edited_tree_refit_needs_no_wait(pipe_active, caps_ok, geo, oldest_geo())
                                : whole_tree_refit_needs_no_wait(pipe_active, caps_ok, 1.25, geo, oldest_geo());
    CHECK(no_wait == (pipe_active && caps_ok && geo.next_slot_free(oldest_geo())), "the two slot predicates disagree");
    if (no_wait) {
      const uint32_t slot = geo.phys(geo.ver + 1);
      for (const Batch& b : inflight) {
        const uint32_t reads = (b.geo_stamp + geo.base) & (geo.slots() - 1u);
        // (a batch whose stamps the plain kernels have reset reads the current version's slot: base == phys(ver) then)
        CHECK(reads != slot, "geometry slot %u written while a batch of version %u names it (stamp %u, base %u)", slot, b.geo_ver, b.geo_stamp, geo.base);
      }
      ++geo.ver;
      geo_mem[slot] = geo.ver;
      ++geo_in_ring;
    } else {
      drain();
      ++geo.ver;
      geo_mem[geo.phys(geo.ver)] = geo.ver;
      if (rnd(6) == 0) release_geo();  // the tree degraded and was rebuilt: it owns its arrays again
    }
  }
};

static void check_plans(std::mt19937& rng) {
  for (int i = 0; i < 20000; ++i) {
    const uint64_t ss = (rng() % 4 == 0) ? 0 : rng() % 40000000, ns = ss ? rng() % (ss + 1) : 0;
    const uint64_t sr = 1 + rng() % 9000000, nr = rng() % (sr + 300);
    const uint32_t want = rng() % 130;
    const size_t free_b = (size_t)(rng() % 256) << 30;
    const GeoRingPlan p = plan_geo_ring(ss, ns, sr, nr, want, free_b, 256, (1u << 23) - 1, 64);
    if (p.log2 < 0) continue;
    CHECK(p.log2 >= 2 && (1u << p.log2) <= kMaxGeoVersions && (1u << p.log2) <= (want < 1 ? 1u : want), "plan of 2^%d versions for %u wanted", p.log2, want);
    CHECK(p.total_slots == p.stride_static + (p.stride_ring << p.log2), "total");
    CHECK(p.total_slots * 64 < (1ull << 32) && p.total_slots < (1ull << 28), "plan beyond the 32-bit node offsets / 2^28 slots");
    CHECK(p.total_slots * 176 <= free_b / 4, "plan beyond a quarter of the free memory");
    CHECK(p.stride_ring <= (1u << 23) - 1 && nr <= p.stride_ring && ns <= p.stride_static, "stride / node counts");
  }
  // split policy (include/gpuspectral_pt.h, "Per-frame edits")
  CHECK(may_split(false, 0, 1.25, 64, 0, false) && !may_split(true, 0, 1.25, 64, 0, false) && !may_split(false, kMaxSceneSplits, 1.25, 64, 0, false) &&
            may_split(false, kMaxSceneSplits - 1, 1.25, 64, 0, false) && !may_split(false, 0, 1.0, 64, 0, false) && !may_split(false, 0, 1.25, 3, 0, false) &&
            !may_split(false, 0, 1.25, 64, 2, false) && !may_split(false, 0, 1.25, 64, 0, true),
        "may_split");
  CHECK(split_worthwhile(300, 100) && !split_worthwhile(299, 100) && !split_worthwhile(0, 5) && !split_worthwhile(5, 0), "split_worthwhile");
  {  // the wait / no-wait predicates of gsp_update_instances
    GeoVersions none, ring;
    ring.stride = 1000, ring.log2 = 2, ring.ver = 5;
    CHECK(edit_wants_split(true, true, false, 1.0, 0, 9) && edit_wants_split(false, false, true, 1.25, 64, 0) && !edit_wants_split(false, true, true, 1.25, 64, 0) &&
              !edit_wants_split(false, false, false, 1.25, 64, 0) && !edit_wants_split(false, false, true, 1.0, 64, 0) && !edit_wants_split(false, false, true, 1.25, 3, 0) &&
              !edit_wants_split(false, false, true, 1.25, 64, 1),
          "edit_wants_split");
    CHECK(first_split_needs_no_wait(false, true, true, none) && !first_split_needs_no_wait(true, true, true, none) && !first_split_needs_no_wait(false, false, true, none) &&
              !first_split_needs_no_wait(false, true, false, none) && !first_split_needs_no_wait(false, true, true, ring),
          "first_split_needs_no_wait");
    CHECK(whole_tree_refit_needs_no_wait(true, true, 1.25, ring, 3) && !whole_tree_refit_needs_no_wait(true, true, 1.25, ring, 2) && !whole_tree_refit_needs_no_wait(true, true, 1.0, ring, 5) &&
              !whole_tree_refit_needs_no_wait(true, true, 1.25, none, 0) && !whole_tree_refit_needs_no_wait(false, true, 1.25, ring, 5),
          "whole_tree_refit_needs_no_wait: 4 slots hold versions oldest .. oldest + 3");
    CHECK(scene_may_still_split(false, 4, 0) && !scene_may_still_split(true, 4, 0) && !scene_may_still_split(false, 3, 0) && !scene_may_still_split(false, 4, 2), "scene_may_still_split");
  }
  // ring sizes by bytes (r05 review: 64 slots of a 64-MB light table are 4 GB)
  CHECK(TableRing::slots_for(64u << 20, TableRing::budget_for((size_t)200 << 30)) == 16, "a 64-MB table gets 16 slots of a 1-GiB budget");
  CHECK(TableRing::slots_for(4096, TableRing::budget_for((size_t)200 << 30)) == kMaxTableVersions, "small tables get the whole field");
  CHECK(TableRing::slots_for((size_t)3 << 30, TableRing::budget_for((size_t)200 << 30)) == 0, "no ring for tables beyond the budget");
  ByteLedger l;
  l.add(10);
  l.sub(4);
  l.sub(7);
  CHECK(l.bytes == 0 && l.underflows == 1, "ledger clamps and counts");
}

int main(int argc, char** argv) {
  const uint32_t seed = argc > 1 ? (uint32_t)std::strtoul(argv[1], nullptr, 10) : 1u;
  const uint32_t steps = argc > 2 ? (uint32_t)std::strtoul(argv[2], nullptr, 10) : 20000u;
  Mock m;
  m.rng.seed(seed);
  check_plans(m.rng);
  m.upload_scene(4096);
  for (uint32_t i = 0; i < steps; ++i) {
    const uint32_t op = m.rnd(100);
    if (op < 55) m.render();
    else if (op < 75) m.update_tables();
    else if (op < 93) m.update_instances();
    else if (op < 98) m.drain();
    else m.upload_scene((size_t)256 * (1 + m.rnd(1u << (4 + m.rnd(16)))));
    CHECK(m.ledger.underflows == 0, "byte ledger underflow at step %u", i);
    CHECK(m.tab.slots == m.tab_mem.size() && m.tab.slots <= kMaxTableVersions, "ring size");
  }
  m.drain();
  m.free_tables();
  m.release_geo();
  CHECK(m.ledger.bytes == 0 && m.ledger.underflows == 0, "ledger ends at %zu (underflows %llu)", m.ledger.bytes, (unsigned long long)m.ledger.underflows);
  std::printf("versions_model seed %u: %u steps, %llu in-place table edits, %llu ring growths, %llu collapses, %llu geometry edits in the ring, %llu drains -- ok\n",
              seed, steps, (unsigned long long)m.in_place, (unsigned long long)m.grown, (unsigned long long)m.collapses, (unsigned long long)m.geo_in_ring,
              (unsigned long long)m.drains);
  return 0;
}
