// pt_internal.h -- declarations shared by the HIP translation units of
// libgpuspectral_pt.so (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/gpuspectral_pt.h"
#include "pt_stages.h"

namespace gsp {

#define GSP_HIP_TRY(expr)                                                                     \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      err = std::string(#expr) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" + std::to_string(__LINE__) + ")"; \
      return e_ == hipErrorOutOfMemory ? GSP_ERR_NOMEM : GSP_ERR_DEVICE;                      \
    }                                                                                         \
  } while (0)

// Result of the device BVH build; all pointers are device memory owned by the caller's context.
struct DeviceBvh {
  q4* nodes = nullptr;      // wide BVH, kNodeQuads quads per node (pt_trace.h); node 0 is the root
  q4* tri_isect = nullptr;  // 3 quads per slot
  q4* tri_shade = nullptr;  // 4 quads per slot
  uint32_t* slot_to_global = nullptr;
  bool arrays_external = false;  // nodes / tri_isect / tri_shade are a slot of the context's geometry ring (pt_render.hip): not free_bvh's
  int32_t root = 0;
  uint32_t num_tris = 0;   // real triangles (0 allowed)
  uint32_t first_slot = 0; // triangle slots in use: [first_slot, first_slot + num_tris); the leading ones are all-zero triangles
  uint32_t num_nodes = 0;
  uint32_t depth = 0;      // levels of the wide tree (bounds the traversal stack: <= 1 node group per level)
  size_t bytes = 0;
  // refit (refit_bvh): the tree is stored level by level -- level l = nodes [level_first[l], level_first[l + 1]) -- so the
  // boxes can be recomputed bottom-up, one launch per level, without parent links
  std::vector<uint32_t> level_first;
  double area_built = 0.0;  // sum of the child-box surface areas right after the last full build (0 = not measured yet)
  uint32_t refits = 0;      // refits since that build
  // scratch of refit_bvh, allocated by the first refit of a tree and kept with it (a per-frame call: seven hipMalloc / hipFree
  // pairs cost more than the kernels): padded triangle boxes by slot, node boxes, per-node areas, reduction space
  q4 *rf_leaf_lo = nullptr, *rf_leaf_hi = nullptr, *rf_box_lo = nullptr, *rf_box_hi = nullptr;
  double *rf_area = nullptr, *rf_total = nullptr;
  void* rf_tmp = nullptr;
  size_t rf_tmp_bytes = 0;
  uint32_t* rf_bounds = nullptr;
};

struct BuildInput {
  const gsp_instance* instances;  // device
  const float* inv_t;             // device, 16 floats per instance: inverse(transpose(M))
  const uint32_t* tri_first;      // device, num_instances + 1 prefix of triangle counts
  uint32_t num_instances;
  const float* positions;  // device, object space
  const float* normals;    // device
  uint32_t num_tris;
  const uint32_t* tri_id_first = nullptr;  // device, per instance: the scene-wide index of its first triangle when `instances` is a
                                           // SUBSET of the scene's (the closest-hit tie-break key must order like the scene's
                                           // triangle list across both trees of a split scene); nullptr = tri_first
  int reinsert_rounds = -1;  // parallel-reinsertion rounds of the build, < 0 = the default (gsp_ctx_options.reinsert_rounds - 1)
};

// ---- internal entry points of a context for pt_multi.hip (same library, not exported) ------------------------------
// Completes everything queued and returns the context's compact RGBA32F accumulate buffer (device memory on the
// context's device, num_pixels records) and the stream its copies are ordered on.
__attribute__((visibility("hidden"))) int gsp_internal_accum(gsp_context* ctx, void** accum, uint64_t* num_pixels, hipStream_t* stream);
// The options a context was created with, defaults filled in (pt_multi.hip divides memory_share among the shares of a device).
__attribute__((visibility("hidden"))) void gsp_internal_resolve_options(const gsp_ctx_options* in, gsp_ctx_options* out);

// Bakes the instances' triangles to world space and builds the wide BVH (PLOC + reinsertion + collapse) on `stream`.
// Returns GSP_OK or an error code with `err` set.
int build_bvh(hipStream_t stream, const BuildInput& in, DeviceBvh& out, std::string& err);
// The instances of `in` have changed (transforms, materials, emission) but not the triangle list: re-bakes every triangle
// packet into its slot and recomputes all node boxes, child orders included, bottom-up in the existing topology.  Results of a
// trace do not depend on the topology (closest-hit rule, pt_trace.h), its speed does: *growth = summed child-box area after /
// right after the last full build; the caller rebuilds when that exceeds its bound.  Synchronises the stream.
int refit_bvh(hipStream_t stream, const BuildInput& in, DeviceBvh& bvh, double* growth, std::string& err);
void free_bvh(DeviceBvh& b);

}  // namespace gsp
