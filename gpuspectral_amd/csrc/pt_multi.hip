// pt_multi.hip -- one frame over several GPUs of one node, behind the C ABI (gsp_multi_*, include/gpuspectral_pt.h).
//
// SURVEY 8(e) / BASELINE north_star: pixels are independent (the seed is a function of the global pixel index and the
// timestamp only, raygen.rgen:37; no pixel filter), so the frame is partitioned by interleaved 32x32 tiles, the scene
// and its BVH are replicated on every GPU, every GPU renders all samples of its own tiles with the single-GPU
// pipeline (pt_render.hip), and the ONLY exchange is one gather of the HDR tiles into GPU 0 at read-back: each
// share's compact RGBA32F buffer travels into a staging buffer on GPU 0, a scatter kernel there places the tiles in
// the frame, and one copy brings the frame to the host.  The gather is ONE RCCL group over xGMI (r03): every share
// ncclSend()s its buffer, GPU 0 posts the matching ncclRecv()s, all inside one ncclGroupStart / ncclGroupEnd of the
// single-process communicator (ncclCommInitAll) -- the north_star's "single RCCL gather of HDR tiles".  A device list
// with repeats (several shares on one GPU: how the one-GPU test box runs partition + gather + scatter with 2 / 3 / 8
// shares) cannot form a communicator -- RCCL wants one rank per device -- and takes the peer-copy route
// (hipMemcpyPeerAsync, a plain copy on the same device); so does a node where librccl is absent or ncclCommInitAll fails
// (no /dev/shm, P2P disabled, ...) unless gsp_ctx_options.gather_route demands RCCL.  librccl is loaded with dlopen at
// the first multi-GPU create: a single-GPU host application links and loads this library without it.
//
// A C++ caller (the reference's host is C++: S/main.cpp:15-30, S/renderer/Renderer.h:22-25,44) gets N GPUs through
// this file without Python or torch; bench.py's one-process-per-GPU path (torch.distributed, RCCL gather) uses the
// same tile partition (gsp_tile_partition) and the same per-GPU pipeline.
//
// Host threading: gsp_render drives its pipeline from the calling thread until every sample is injected, so each
// share's context gets its own host thread for the duration of a call (one std::thread per share per call; the
// calls last milliseconds to minutes).  The same device may appear more than once in `devices` (several shares on
// one GPU): that is how the single-GPU test box exercises partition + gather + scatter.
#include <algorithm>
#include <cstring>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: the entry points are resolved at run time (rccl_api below)

#include "pt_internal.h"

namespace gsp {
namespace {

// The six RCCL entry points of the gather, from librccl.so.1 as the loader finds it (LD_LIBRARY_PATH, ld.so.conf, rpath)
// or from the ROCm default location.  Loaded once; a missing library is an error message, not a load failure of ours.
struct RcclApi {
  void* handle = nullptr;
  std::string err;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
const RcclApi& rccl_api() {
  static const RcclApi api = [] {
    RcclApi a;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      a.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (a.handle) break;
      if (const char* e = dlerror()) a.err = e;
    }
    if (!a.handle) {
      a.err = "librccl not loadable (" + a.err + ")";
      return a;
    }
    a.CommInitAll = (decltype(a.CommInitAll))dlsym(a.handle, "ncclCommInitAll");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.handle, "ncclCommDestroy");
    a.GroupStart = (decltype(a.GroupStart))dlsym(a.handle, "ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd))dlsym(a.handle, "ncclGroupEnd");
    a.Send = (decltype(a.Send))dlsym(a.handle, "ncclSend");
    a.Recv = (decltype(a.Recv))dlsym(a.handle, "ncclRecv");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.handle, "ncclGetErrorString");
    if (!a.CommInitAll || !a.CommDestroy || !a.GroupStart || !a.GroupEnd || !a.Send || !a.Recv || !a.GetErrorString) {
      a.err = "librccl lacks an entry point of the gather";
      dlclose(a.handle);
      a.handle = nullptr;
    }
    return a;
  }();
  return api;
}

// frame[ids[i]] = compact[i]: the tiles of one share into the full frame (16-B records, coalesced reads; writes are
// runs of 32 pixels = 512 B)
__global__ __launch_bounds__(256) void k_scatter_tiles(const q4* __restrict__ compact, const uint32_t* __restrict__ ids,
                                                       uint64_t n, q4* __restrict__ frame) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) frame[ids[i]] = compact[i];
}

}  // namespace
}  // namespace gsp

using namespace gsp;

struct gsp_multi {
  std::vector<int> devices;
  std::vector<gsp_context*> ctx;
  std::vector<std::vector<uint32_t>> ids;  // per share: owned pixels, increasing
  std::vector<uint64_t> offset;            // per share: first record in the staging buffer
  uint32_t width = 0, height = 0, tile = 32;
  bool have_frame = false;
  // on devices[0]
  hipStream_t stream = nullptr;
  q4* staging = nullptr;     // all shares' compact buffers back to back
  q4* frame = nullptr;       // width * height
  uint32_t* d_ids = nullptr; // all shares' pixel ids back to back
  size_t frame_pixels = 0;
  std::string err;
  // RCCL: one communicator rank per share (device lists without repeats), ranks = share indices
  std::vector<ncclComm_t> comms;
  std::vector<hipStream_t> comm_streams;  // one per share, on its device
  std::vector<char> peer_ok;              // per share: peer access towards devices[0] was obtained (copy route)
  bool use_rccl = false;
  std::string route_note;                       // why the gathers take the route they take
  uint64_t rccl_gathers = 0, copy_gathers = 0;  // gathers through each route so far (gsp_multi_gather_route)
};

namespace {

std::mutex g_multi_err_mutex;
std::string g_multi_create_error;
void set_multi_create_error(const std::string& e) {
  std::lock_guard<std::mutex> lk(g_multi_err_mutex);
  g_multi_create_error = e;
}

// Runs fn(share) on one host thread per share and returns the first failing status (GSP_OK if none).
template <class F>
int for_each_share(gsp_multi* m, F fn) {
  const size_t n = m->ctx.size();
  std::vector<int> rc(n, GSP_OK);
  if (n == 1) {
    rc[0] = fn(0);
  } else {
    std::vector<std::thread> th;
    th.reserve(n);
    try {
      for (size_t r = 0; r < n; ++r) th.emplace_back([&, r] { rc[r] = fn(r); });
    } catch (const std::system_error& e) {  // could not start a host thread: the shares that did start still finish
      for (auto& t : th) t.join();
      m->err = std::string("cannot start a host thread per share: ") + e.what();
      return GSP_ERR_NOMEM;
    }
    for (auto& t : th) t.join();
  }
  for (size_t r = 0; r < n; ++r)
    if (rc[r] != GSP_OK) {
      m->err = "share " + std::to_string(r) + " (device " + std::to_string(m->devices[r]) + "): " + gsp_last_error(m->ctx[r]);
      return rc[r];
    }
  return GSP_OK;
}

#define MULTI_TRY(m, expr)                                                                      \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) {                                                                     \
      (m)->err = std::string(#expr) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" + std::to_string(__LINE__) + ")"; \
      return e_ == hipErrorOutOfMemory ? GSP_ERR_NOMEM : GSP_ERR_DEVICE;                        \
    }                                                                                           \
  } while (0)

void free_frame(gsp_multi* m) {
  if (m->devices.empty()) return;
  (void)hipSetDevice(m->devices[0]);
  if (m->staging) (void)hipFree(m->staging);
  if (m->frame) (void)hipFree(m->frame);
  if (m->d_ids) (void)hipFree(m->d_ids);
  m->staging = m->frame = nullptr;
  m->d_ids = nullptr;
  m->have_frame = false;
}

}  // namespace

extern "C" {

// Tile (tx, ty) of a frame cut into tile x tile blocks belongs to share (ty * tiles_x + tx + ty) % world: round-robin
// along a row, staggered by one from row to row so that the shares interleave in both axes (load balance).
uint64_t gsp_tile_partition(uint32_t width, uint32_t height, uint32_t rank, uint32_t world, uint32_t tile, uint32_t* out_ids) {
  if (world == 0 || tile == 0 || rank >= world) return 0;
  const uint32_t tiles_x = (width + tile - 1) / tile;
  uint64_t n = 0;
  for (uint32_t y = 0; y < height; ++y) {
    const uint32_t ty = y / tile;
    for (uint32_t tx = 0; tx < tiles_x; ++tx) {
      if (((uint64_t)ty * tiles_x + tx + ty) % world != rank) continue;
      const uint32_t x0 = tx * tile, x1 = std::min(width, x0 + tile);
      if (out_ids)
        for (uint32_t x = x0; x < x1; ++x) out_ids[n + (x - x0)] = y * width + x;
      n += x1 - x0;
    }
  }
  return n;
}

const char* gsp_multi_last_error(const gsp_multi* m) {
  if (m) return m->err.c_str();
  std::lock_guard<std::mutex> lk(g_multi_err_mutex);
  static thread_local std::string copy;
  copy = g_multi_create_error;
  return copy.c_str();
}

void gsp_multi_destroy(gsp_multi* m) {
  if (!m) return;
  for (size_t r = 0; r < m->comms.size(); ++r) {
    (void)hipSetDevice(m->devices[r]);
    if (m->comms[r]) (void)rccl_api().CommDestroy(m->comms[r]);
  }
  for (size_t r = 0; r < m->comm_streams.size(); ++r) {
    (void)hipSetDevice(m->devices[r]);
    if (m->comm_streams[r]) (void)hipStreamDestroy(m->comm_streams[r]);
  }
  for (gsp_context* c : m->ctx) gsp_ctx_destroy(c);
  free_frame(m);
  if (m->stream) {
    (void)hipSetDevice(m->devices[0]);
    (void)hipStreamDestroy(m->stream);
  }
  delete m;
}

int gsp_multi_create(const int* devices, int n, gsp_multi** out) { return gsp_multi_create_ex(devices, n, nullptr, out); }

int gsp_multi_create_ex(const int* devices, int n, const gsp_ctx_options* options, gsp_multi** out) {
  if (!out) return GSP_ERR_INVALID;
  *out = nullptr;
  if (!devices || n <= 0 || n > 64) {
    set_multi_create_error("gsp_multi_create: need 1..64 devices");
    return GSP_ERR_INVALID;
  }
  if (options && options->struct_size < 8) {
    set_multi_create_error("gsp_ctx_options.struct_size is not set");
    return GSP_ERR_INVALID;
  }
  gsp_ctx_options opt;
  gsp_internal_resolve_options(options, &opt);
  gsp_multi* m = new gsp_multi();
  m->devices.assign(devices, devices + n);
  bool repeats = false;
  for (int r = 0; r < n; ++r) {
    // shares that sit on one device size their path pools concurrently: each takes its part of the share a lone context may
    int same = 0;
    for (int q = 0; q < n; ++q) same += devices[q] == devices[r] ? 1 : 0;
    repeats = repeats || same > 1;
    gsp_ctx_options o = opt;
    o.memory_share = opt.memory_share / same;
    gsp_context* c = nullptr;
    int rc = gsp_ctx_create_ex(devices[r], &o, &c);
    if (rc != GSP_OK) {
      set_multi_create_error(std::string("share ") + std::to_string(r) + ": " + gsp_last_error(nullptr));
      gsp_multi_destroy(m);
      return rc;
    }
    m->ctx.push_back(c);
  }
  // peer access towards the gathering device (xGMI) for the copy route; a share on the gathering device needs none
  m->peer_ok.assign(n, 1);
  for (int r = 1; r < n; ++r) {
    if (devices[r] == devices[0]) continue;
    int can = 0;
    m->peer_ok[r] = 0;
    if (hipDeviceCanAccessPeer(&can, devices[r], devices[0]) == hipSuccess && can) {
      (void)hipSetDevice(devices[r]);
      const hipError_t pe = hipDeviceEnablePeerAccess(devices[0], 0);
      m->peer_ok[r] = pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled;
      (void)hipGetLastError();  // (without peer access hipMemcpyPeerAsync stages through the host: slower, still correct)
    }
  }
  // RCCL communicator over the shares (ranks = share indices): needs one device per rank.  GSP_GATHER_AUTO falls back to
  // the copy route when librccl or the communicator is unavailable; GSP_GATHER_RCCL makes that an error.
  const bool forced = opt.gather_route == GSP_GATHER_RCCL;
  const bool want_rccl = forced || (opt.gather_route == GSP_GATHER_AUTO && n > 1 && !repeats);
  m->route_note = opt.gather_route == GSP_GATHER_COPY ? "copy route requested" : (n == 1 ? "one share: no gather" : "device list with repeats: RCCL wants one rank per device");
  if (want_rccl && repeats) {
    set_multi_create_error("GSP_GATHER_RCCL with a repeated device: RCCL wants one rank per device");
    gsp_multi_destroy(m);
    return GSP_ERR_INVALID;
  }
  if (want_rccl) {
    const RcclApi& api = rccl_api();
    std::string why;
    if (!api.handle) {
      why = api.err;
    } else {
      m->comms.assign(n, nullptr);
      const ncclResult_t nr = api.CommInitAll(m->comms.data(), n, devices);
      if (nr != ncclSuccess) {
        why = std::string("ncclCommInitAll: ") + api.GetErrorString(nr);
        m->comms.clear();
      }
    }
    if (!why.empty()) {
      if (forced) {
        set_multi_create_error(why);
        gsp_multi_destroy(m);
        return GSP_ERR_DEVICE;
      }
      m->route_note = "copy route: " + why;
    } else {
      m->comm_streams.assign(n, nullptr);
      for (int r = 0; r < n; ++r) {
        hipError_t se = hipSetDevice(devices[r]);
        if (se == hipSuccess) se = hipStreamCreateWithFlags(&m->comm_streams[r], hipStreamNonBlocking);
        if (se != hipSuccess) {
          set_multi_create_error(std::string("gsp_multi_create (RCCL stream): ") + hipGetErrorString(se));
          gsp_multi_destroy(m);
          return GSP_ERR_DEVICE;
        }
      }
      m->use_rccl = true;
      m->route_note = "RCCL send / recv group";
    }
  }
  hipError_t e = hipSetDevice(devices[0]);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    set_multi_create_error(std::string("gsp_multi_create: ") + hipGetErrorString(e));
    gsp_multi_destroy(m);
    return GSP_ERR_DEVICE;
  }
  m->err = "gather route: " + m->route_note;  // (readable through gsp_multi_last_error until a call fails)
  *out = m;
  return GSP_OK;
}

int gsp_multi_num_shares(const gsp_multi* m) { return m ? (int)m->ctx.size() : 0; }

int gsp_multi_upload_scene(gsp_multi* m, const gsp_scene_desc* scene) {
  if (!m) return GSP_ERR_INVALID;
  if (!scene) {
    m->err = "gsp_multi_upload_scene: null scene";
    return GSP_ERR_INVALID;
  }
  return for_each_share(m, [&](size_t r) { return gsp_upload_scene(m->ctx[r], scene); });
}

int gsp_multi_update_camera(gsp_multi* m, const gsp_camera* camera) {
  if (!m) return GSP_ERR_INVALID;
  if (!camera) {
    m->err = "gsp_multi_update_camera: null camera";
    return GSP_ERR_INVALID;
  }
  return for_each_share(m, [&](size_t r) { return gsp_update_camera(m->ctx[r], camera); });
}

int gsp_multi_update_instances(gsp_multi* m, const gsp_instance* instances, uint32_t num_instances) {
  if (!m) return GSP_ERR_INVALID;
  if (!instances && num_instances) {
    m->err = "gsp_multi_update_instances: null instances";
    return GSP_ERR_INVALID;
  }
  return for_each_share(m, [&](size_t r) { return gsp_update_instances(m->ctx[r], instances, num_instances); });
}

int gsp_multi_update_tables(gsp_multi* m, const gsp_scene_desc* scene) {
  if (!m) return GSP_ERR_INVALID;
  if (!scene) {
    m->err = "gsp_multi_update_tables: null scene";
    return GSP_ERR_INVALID;
  }
  return for_each_share(m, [&](size_t r) { return gsp_update_tables(m->ctx[r], scene); });
}

int gsp_multi_frame_begin(gsp_multi* m, uint32_t width, uint32_t height) {
  if (!m) return GSP_ERR_INVALID;
  if (width == 0 || height == 0) {
    m->err = "gsp_multi_frame_begin: empty frame";
    return GSP_ERR_INVALID;
  }
  const uint32_t world = (uint32_t)m->ctx.size();
  free_frame(m);
  m->width = width;
  m->height = height;
  m->ids.assign(world, {});
  m->offset.assign(world, 0);
  uint64_t total = 0;
  for (uint32_t r = 0; r < world; ++r) {
    const uint64_t cnt = gsp_tile_partition(width, height, r, world, m->tile, nullptr);
    m->ids[r].resize(cnt);
    gsp_tile_partition(width, height, r, world, m->tile, m->ids[r].data());
    m->offset[r] = total;
    total += cnt;
  }
  m->frame_pixels = (size_t)width * height;
  if (total != m->frame_pixels) {
    m->err = "internal error: tile partition does not cover the frame";
    return GSP_ERR_INVALID;
  }
  if (world > 1 || m->use_rccl) {  // (one share + a forced RCCL route: the self send / recv of the RCCL smoke test)
    MULTI_TRY(m, hipSetDevice(m->devices[0]));
    MULTI_TRY(m, hipMalloc((void**)&m->staging, total * sizeof(q4)));
    MULTI_TRY(m, hipMalloc((void**)&m->frame, total * sizeof(q4)));
    MULTI_TRY(m, hipMalloc((void**)&m->d_ids, total * sizeof(uint32_t)));
    for (uint32_t r = 0; r < world; ++r)
      MULTI_TRY(m, hipMemcpyAsync(m->d_ids + m->offset[r], m->ids[r].data(), m->ids[r].size() * sizeof(uint32_t),
                                  hipMemcpyHostToDevice, m->stream));
    MULTI_TRY(m, hipStreamSynchronize(m->stream));
  }
  int rc = for_each_share(m, [&](size_t r) {
    // a single share owns the whole frame: no subset, no gather
    static const uint32_t none = 0;  // a share without a tile (more shares than tiles) owns NO pixel: non-null list, length 0
    return world == 1 && !m->use_rccl ? gsp_frame_begin(m->ctx[r], width, height, nullptr, 0)
                      : gsp_frame_begin(m->ctx[r], width, height, m->ids[r].empty() ? &none : m->ids[r].data(), m->ids[r].size());
  });
  m->have_frame = rc == GSP_OK;
  return rc;
}

int gsp_multi_render(gsp_multi* m, const gsp_render_params* params) {
  if (!m) return GSP_ERR_INVALID;
  if (!params) {
    m->err = "gsp_multi_render: null parameters";
    return GSP_ERR_INVALID;
  }
  return for_each_share(m, [&](size_t r) { return gsp_render(m->ctx[r], params); });
}

int gsp_multi_sync(gsp_multi* m) {
  if (!m) return GSP_ERR_INVALID;
  return for_each_share(m, [&](size_t r) { return gsp_sync(m->ctx[r]); });
}

// Completes all queued samples, gathers the HDR tiles into devices[0] and leaves the assembled frame there;
// *device_frame (optional) receives the device pointer (width*height RGBA32F on devices[0], valid until the next
// gsp_multi_frame_begin / destroy).
int gsp_multi_gather(gsp_multi* m, void** device_frame) {
  if (!m) return GSP_ERR_INVALID;
  if (!m->have_frame) {
    m->err = "gsp_multi_gather needs gsp_multi_frame_begin first";
    return GSP_ERR_INVALID;
  }
  const uint32_t world = (uint32_t)m->ctx.size();
  if (world == 1 && !m->use_rccl) {
    if (device_frame) *device_frame = nullptr;  // (single share: the frame lives in the context's accumulate buffer)
    int rc = gsp_sync(m->ctx[0]);
    if (rc != GSP_OK) m->err = gsp_last_error(m->ctx[0]);
    return rc;
  }
  // every share finishes its samples; its compact accumulate buffer is the send buffer
  std::vector<void*> src(world, nullptr);
  std::vector<uint64_t> cnt(world, 0);
  int rc = for_each_share(m, [&](size_t r) { return gsp_internal_accum(m->ctx[r], &src[r], &cnt[r], nullptr); });
  if (rc != GSP_OK) return rc;
  for (uint32_t r = 0; r < world; ++r)
    if (cnt[r] != m->ids[r].size()) {
      m->err = "internal error: a share's pixel count differs from its tile list";
      return GSP_ERR_DEVICE;
    }
  if (m->use_rccl) {
    // the one exchange of the job, as ONE RCCL group: share r sends its tiles, GPU 0 receives them side by side
    const RcclApi& api = rccl_api();
    ncclResult_t nr = api.GroupStart();
    for (uint32_t r = 0; r < world && nr == ncclSuccess; ++r) {
      if (cnt[r] == 0) continue;
      nr = api.Send(src[r], cnt[r] * 4, ncclFloat, 0, m->comms[r], m->comm_streams[r]);
      if (nr == ncclSuccess) nr = api.Recv(m->staging + m->offset[r], cnt[r] * 4, ncclFloat, (int)r, m->comms[0], m->comm_streams[0]);
    }
    const ncclResult_t ge = api.GroupEnd();
    if (nr == ncclSuccess) nr = ge;
    if (nr != ncclSuccess) {
      m->err = std::string("RCCL gather: ") + api.GetErrorString(nr);
      return GSP_ERR_DEVICE;
    }
    for (uint32_t r = 0; r < world; ++r) {
      MULTI_TRY(m, hipSetDevice(m->devices[r]));
      MULTI_TRY(m, hipStreamSynchronize(m->comm_streams[r]));
    }
    ++m->rccl_gathers;
  } else {
    // peer copies into the staging buffer (device lists with repeats; GSP_MULTI_GATHER=copy)
    MULTI_TRY(m, hipSetDevice(m->devices[0]));
    for (uint32_t r = 0; r < world; ++r) {
      if (cnt[r] == 0) continue;
      if (m->devices[r] == m->devices[0])
        MULTI_TRY(m, hipMemcpyAsync(m->staging + m->offset[r], src[r], cnt[r] * sizeof(q4), hipMemcpyDeviceToDevice, m->stream));
      else
        MULTI_TRY(m, hipMemcpyPeerAsync(m->staging + m->offset[r], m->devices[0], src[r], m->devices[r], cnt[r] * sizeof(q4), m->stream));
    }
    MULTI_TRY(m, hipStreamSynchronize(m->stream));
    ++m->copy_gathers;
  }
  MULTI_TRY(m, hipSetDevice(m->devices[0]));
  const uint64_t total = m->frame_pixels;
  const uint32_t grid = (uint32_t)std::min<uint64_t>((total + 255) / 256, 256 * 8);
  hipLaunchKernelGGL(k_scatter_tiles, dim3(grid), dim3(256), 0, m->stream, m->staging, m->d_ids, total, m->frame);
  MULTI_TRY(m, hipGetLastError());
  MULTI_TRY(m, hipStreamSynchronize(m->stream));
  if (device_frame) *device_frame = m->frame;
  return GSP_OK;
}

// Which route the gathers of this object take: 1 = RCCL (ncclSend / ncclRecv group), 0 = peer copies.
// *rccl_gathers / *copy_gathers (optional): how many gathers each route has carried so far.
int gsp_multi_gather_route(const gsp_multi* m, uint64_t* rccl_gathers, uint64_t* copy_gathers) {
  if (!m) return -1;
  if (rccl_gathers) *rccl_gathers = m->rccl_gathers;
  if (copy_gathers) *copy_gathers = m->copy_gathers;
  return m->use_rccl ? 1 : 0;
}

int gsp_multi_download(gsp_multi* m, float* out_rgba) {
  if (!m) return GSP_ERR_INVALID;
  if (!out_rgba || !m->have_frame) {
    m->err = !out_rgba ? "gsp_multi_download: null output buffer" : "gsp_multi_download needs gsp_multi_frame_begin first";
    return GSP_ERR_INVALID;
  }
  if (m->ctx.size() == 1 && !m->use_rccl) {
    int rc = gsp_download(m->ctx[0], out_rgba);
    if (rc != GSP_OK) m->err = gsp_last_error(m->ctx[0]);
    return rc;
  }
  int rc = gsp_multi_gather(m, nullptr);
  if (rc != GSP_OK) return rc;
  MULTI_TRY(m, hipSetDevice(m->devices[0]));
  MULTI_TRY(m, hipMemcpyAsync(out_rgba, m->frame, m->frame_pixels * sizeof(q4), hipMemcpyDeviceToHost, m->stream));
  MULTI_TRY(m, hipStreamSynchronize(m->stream));
  return GSP_OK;
}

// total (optional): counters summed over the shares, times = the slowest share's (they run concurrently);
// per_share (optional): gsp_multi_num_shares() records.
int gsp_multi_get_stats(gsp_multi* m, gsp_stats* total, gsp_stats* per_share) {
  if (!m) return GSP_ERR_INVALID;
  std::vector<gsp_stats> st(m->ctx.size());
  int rc = for_each_share(m, [&](size_t r) { return gsp_get_stats(m->ctx[r], &st[r]); });
  if (rc != GSP_OK) return rc;
  if (per_share) std::memcpy(per_share, st.data(), st.size() * sizeof(gsp_stats));
  if (total) {
    gsp_stats t = st[0];
    for (size_t r = 1; r < st.size(); ++r) {
      const gsp_stats& s = st[r];
      t.extension_rays += s.extension_rays;
      t.shadow_rays += s.shadow_rays;
      t.shaded_vertices += s.shaded_vertices;
      t.samples += s.samples;
      t.nodes_visited += s.nodes_visited;
      t.tris_tested += s.tris_tested;
      t.stat_rays += s.stat_rays;
      t.shadow_nodes_visited += s.shadow_nodes_visited;
      t.shadow_tris_tested += s.shadow_tris_tested;
      t.shadow_stat_rays += s.shadow_stat_rays;
      t.nodes_from_lds += s.nodes_from_lds;
      t.shadow_nodes_from_lds += s.shadow_nodes_from_lds;
      t.shadow_stat_occluded += s.shadow_stat_occluded;
      t.shadow_stat_occluded_nodes += s.shadow_stat_occluded_nodes;
      t.shadow_stat_no_triangle += s.shadow_stat_no_triangle;
      t.scene_drains = std::max(t.scene_drains, s.scene_drains);
      t.scene_splits = std::max(t.scene_splits, s.scene_splits);
      t.algorithmic_bytes += s.algorithmic_bytes;
      t.memoised_rays += s.memoised_rays;
      t.memo_build_rays += s.memo_build_rays;
      t.bvh_depth = std::max(t.bvh_depth, s.bvh_depth);
      t.scene_updates = std::max(t.scene_updates, s.scene_updates);
      t.scene_refits = std::max(t.scene_refits, s.scene_refits);
      t.extend_launches += s.extend_launches;
      t.device_bytes += s.device_bytes;
      t.render_seconds = std::max(t.render_seconds, s.render_seconds);
      t.extend_kernel_ms = std::max(t.extend_kernel_ms, s.extend_kernel_ms);
      t.shade_kernel_ms = std::max(t.shade_kernel_ms, s.shade_kernel_ms);
      t.connect_kernel_ms = std::max(t.connect_kernel_ms, s.connect_kernel_ms);
      t.bvh_build_ms = std::max(t.bvh_build_ms, s.bvh_build_ms);
    }
    *total = t;
  }
  return GSP_OK;
}

int gsp_multi_reset_stats(gsp_multi* m) {
  if (!m) return GSP_ERR_INVALID;
  return for_each_share(m, [&](size_t r) { return gsp_reset_stats(m->ctx[r]); });
}

}  // extern "C"
