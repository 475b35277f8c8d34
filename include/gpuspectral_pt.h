/*
 * gpuspectral_pt.h -- C ABI of the MI355X-native path-tracing integrator.
 *
 * This is the drop-in boundary for the reference's PathTracer render pass
 * (sunho/GPUSpectral).  Everything a reference maintainer needs in order to
 * replace `PathTracer::createRenderPass` + the Vulkan ray-tracing pipeline
 * with the HIP wavefront tracer is declared here: plain pointers and sizes,
 * `extern "C"`, no C++/torch types.
 *
 * Reference interfaces each entry point replaces (S/ = src/GPUSpectral/):
 *   gsp_ctx_create / gsp_ctx_destroy
 *        PathTracer::PathTracer(Renderer&) + setup()           S/renderer/PathTracer.h:50-53, PathTracer.cpp:5-7
 *        (the Vulkan device/driver objects it borrows)         S/renderer/Renderer.cpp:19-45
 *   gsp_upload_scene
 *        PathTracer::prepareScene (instance table, 8 BSDF
 *        arrays, light array)                                  S/renderer/PathTracer.cpp:58-93
 *        Renderer::getOrCreateBLAS / HwDriver::createTLAS      S/renderer/Renderer.cpp:122-131, PathTracer.cpp:10-19
 *        (vkCmdBuildAccelerationStructuresKHR)                 S/backend/vulkan/VulkanRays.cpp:6-86,91-181
 *   gsp_update_camera / gsp_update_instances / gsp_update_tables   (ABI 5)
 *        what the reference re-reads on EVERY createRenderPass while the BLAS of a mesh stays cached by mesh id:
 *        renderState.camera                                     S/renderer/PathTracer.cpp:88-90
 *        the TLAS over obj.transform + the per-object Instance
 *        records (transformInvT, emission, bsdf, twofaced)     S/renderer/PathTracer.cpp:10-19,58-70
 *        the eight BSDF tables and the triangle lights          S/renderer/PathTracer.cpp:74-87
 *   gsp_ctx_create_ex / gsp_ctx_options
 *        the knobs of the device objects PathTracer borrows     S/renderer/Renderer.cpp:19-45 (none are per-pass there;
 *        here: path-pool size, result ring, memory share, ...)
 *   gsp_frame_begin
 *        accumulateBuffer = createTexture(RGBA32F, W, H)       S/renderer/PathTracer.cpp:5-7
 *   gsp_render
 *        driver.traceRays(pipeline, W, H) once per sample      S/renderer/PathTracer.cpp:24-39,
 *        (raygen.rgen + rayhit.rchit + miss shaders)           S/backend/vulkan/VulkanDriver.cpp:319-347
 *   gsp_download / gsp_download_compact / gsp_peek / gsp_peek_to_device / gsp_copy_accum_to_device
 *        the RGBA32F accumulateBuffer the blit pass samples    S/renderer/PathTracer.cpp:41-55, raygen.rgen:84-108
 *   gsp_last_error
 *        std::runtime_error thrown by the driver               e.g. S/backend/vulkan/VulkanDevice.cpp:31,58,66
 *   gsp_multi_* (one frame over the N GPUs of a node; the reference has one VkDevice: Renderer.cpp:19-45)
 *        the same calls as above, applied to every GPU's share of the frame: the image is cut into interleaved
 *        32x32 tiles (gsp_tile_partition), the scene is replicated, and gsp_multi_download gathers the HDR tiles
 *        into GPU 0 over xGMI before the one copy to the host                                SURVEY.md 8(e)
 *
 * All structs are POD with the reference's scalar layouts
 * (S/renderer/Scene.h:29-109) so the reference's std::vectors can be passed
 * by pointer without repacking.
 *
 * Threading: one host thread per context.  All device work is queued on the
 * context's own HIP stream; gsp_render is asynchronous, the download calls and
 * gsp_sync block.
 */
#ifndef GPUSPECTRAL_PT_H
#define GPUSPECTRAL_PT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSP_ABI_VERSION 8 /* 8: gsp_render_params.struct_size (first field); 2: gsp_multi_*, gsp_tile_partition, gsp_stats.algorithmic_bytes; 4: gsp_stats.memoised_rays, memo_build_rays, bvh_depth;
                             5: gsp_update_camera / _instances / _tables (+ gsp_multi_*), gsp_ctx_options + gsp_ctx_create_ex /
                                gsp_multi_create_ex, gsp_render_params.nee, gsp_stats.scene_updates;
                             6: gsp_update_instances refits the tree (gsp_ctx_options.refit_growth, gsp_stats.scene_refits);
                             7: gsp_render_params.nee -> disable_nee (a zeroed struct is the reference as shipped),
                                gsp_stats.shadow_stat_no_triangle */

/* ---- status codes (0 = ok); the message is at gsp_last_error(ctx) ---- */
#define GSP_OK 0
#define GSP_ERR_INVALID 1 /* bad argument / call order            */
#define GSP_ERR_DEVICE 2  /* HIP runtime error                    */
#define GSP_ERR_NOMEM 3   /* device or host allocation failed     */
#define GSP_ERR_SCENE 4   /* scene description is inconsistent    */

/* ---- BSDF type ids: S/assets/shaders/BSDF.inc:1-8, rayhit.rchit:332-339 ---- */
#define GSP_BSDF_DIFFUSE 0
#define GSP_BSDF_SMOOTH_DIELECTRIC 1
#define GSP_BSDF_SMOOTH_CONDUCTOR 2
#define GSP_BSDF_SMOOTH_PLASTIC 3
#define GSP_BSDF_ROUGH_CONDUCTOR 4
#define GSP_BSDF_SMOOTH_FLOOR 5
#define GSP_BSDF_ROUGH_FLOOR 6
#define GSP_BSDF_ROUGH_PLASTIC 7
#define GSP_BSDF_TYPE_COUNT 8

/* handle = (type << 16) | index        S/renderer/Scene.h:83-97, pt_common.glsl:30-42 */
#define GSP_BSDF_HANDLE(type, index) ((((uint32_t)(type)) << 16) | ((uint32_t)(index) & 0xFFFFu))

/* ---- BSDF parameter records, scalar layout = S/renderer/Scene.h:29-81 ---- */
typedef struct gsp_diffuse_bsdf {
  float reflectance[3];
  int32_t has_texture; /* never set by the reference loader.  0 = none; k > 0 = gsp_scene_desc.textures[k - 1] replaces
                          `reflectance` (dormant-feature extension below); the same in the two other records with this field */
} gsp_diffuse_bsdf; /* 16 B */

typedef struct gsp_smooth_dielectric_bsdf {
  float ior_in;
  float ior_out;
} gsp_smooth_dielectric_bsdf; /* 8 B */

typedef struct gsp_smooth_conductor_bsdf {
  float ior_in;
  float ior_out;
} gsp_smooth_conductor_bsdf; /* 8 B */

typedef struct gsp_smooth_plastic_bsdf {
  float diffuse[3];
  float ior_in;
  float ior_out;
  float r0;
} gsp_smooth_plastic_bsdf; /* 24 B */

typedef struct gsp_rough_conductor_bsdf {
  float eta[3];
  float k[3];
  float reflectance[3];
  float alpha;
  int32_t has_texture;
} gsp_rough_conductor_bsdf; /* 44 B */

typedef struct gsp_smooth_floor_bsdf {
  float diffuse[3];
  float r0;
} gsp_smooth_floor_bsdf; /* 16 B */

typedef struct gsp_rough_floor_bsdf {
  float diffuse[3];
  float r0;
  float alpha;
} gsp_rough_floor_bsdf; /* 20 B */

typedef struct gsp_rough_plastic_bsdf {
  float diffuse[3];
  float ior_in;
  float ior_out;
  float r0;
  float alpha;
  int32_t has_texture;
} gsp_rough_plastic_bsdf; /* 32 B */

/* S/renderer/Scene.h:106-109; rayhit.rchit:17-20 (world space, built on the host) */
typedef struct gsp_triangle_light {
  float positions[3][4];
  float radiance[4];
} gsp_triangle_light; /* 64 B */

/*
 * One render object = one TLAS instance (S/renderer/PathTracer.cpp:12-17,60-70).
 * `transform` is the glm::mat4 object-to-world matrix in glm's memory order
 * (column-major: transform[4*c + r]).  Vertices [first_vertex, first_vertex +
 * vertex_count) of the shared position/normal arrays are the instance's
 * de-indexed mesh: three consecutive vertices form one triangle
 * (S/renderer/Mesh.cpp:7-51, rayhit.rchit:670).
 */
typedef struct gsp_instance {
  float transform[16];
  float emission[3];
  uint32_t bsdf;     /* GSP_BSDF_HANDLE */
  uint32_t twofaced; /* Material::twofaced */
  uint32_t first_vertex;
  uint32_t vertex_count; /* multiple of 3 */
} gsp_instance; /* 92 B */

/* S/renderer/PathTracer.h:10-15: eye = to_world column 3, fov in radians */
typedef struct gsp_camera {
  float to_world[16]; /* glm memory order (column-major) */
  float fov;
} gsp_camera;

/*
 * ---- Dormant features of the reference, SURVEY 8(f).3: bitmap / checkerboard textures and an environment map ----
 * The reference declares them -- `hasTexture` in DiffuseBSDF / RoughConductorBSDF / RoughPlasticBSDF
 * (S/renderer/Scene.h:31,66,75; rayhit.rchit:24,54,63), `Scene::envMap` + `Envmap` (Scene.h:116-119,182), the loaders
 * loadTexture / loadHdrTexture (S/engine/Loader.cpp:66-116), the texture and emitter branches (:122-143,338-346) --
 * but the branches are commented out, no shader samples a texture and the closest-hit shader passes uv = vec2(0)
 * (rayhit.rchit:716,729).  A scene loaded the reference's way therefore never sets any field below, and all-zero
 * (absent) reproduces the reference exactly.  When a caller does set them, the semantics are the ones the declarations
 * imply, fixed here because no running reference code defines them:
 *   texture    RGBA8 texels, rows bottom-up as loadTexture writes them (Loader.cpp:74-81: image row height-1 first), so
 *              v = 0 is the bottom of the image; channel value = texel_decode[byte] (NULL: byte / 255); repeat wrap,
 *              bilinear filter, texel centres at (i + 0.5) / width; A ignored
 *   uv         per-vertex (Mesh Vertex::uv, Loader.cpp:50-52), interpolated with the hit's barycentrics like the normal
 *   where      a record with has_texture = k > 0 takes its kD (Diffuse::reflectance, RoughConductor::reflectance,
 *              RoughPlastic::diffuse: the `vec3 kD = bsdf.reflectance` lines, rayhit.rchit:342,352,512,526,552,574) from
 *              textures[k - 1] at the hit's uv
 *   envmap     RGBA32F lat-long image, rows bottom-up as loadHdrTexture writes them (Loader.cpp:103-110); a path that
 *              escapes the scene (miss.rmiss) adds weight * texel(direction) behind the firefly test like any emitted term;
 *              light sampling does not see it (sampleLight draws triangle lights only, rayhit.rchit:123-153).  Direction d
 *              (world) -> e = to_local * d; u = atan2(e.x, -e.z) / 2pi + 0.5, v = 1 - atan2(|e.xz|, e.y) / pi; wrap in u,
 *              clamp in v, bilinear.
 */
typedef struct gsp_texture {
  uint32_t width, height;
  uint64_t first_texel; /* index of texel (0, 0) in gsp_scene_desc.texels */
} gsp_texture;          /* 16 B */

typedef struct gsp_envmap {
  const float* texels; /* width * height RGBA32F, NULL = no environment map */
  uint32_t width, height;
  float to_local[16];  /* world -> envmap frame, glm memory order (the inverse of Envmap::transform, Scene.h:118; the
                          host inverts it so that no matrix inversion sits on the parity path) */
} gsp_envmap;

typedef struct gsp_scene_desc {
  const gsp_instance* instances;
  uint32_t num_instances;
  const float* positions; /* tight float3, object space */
  const float* normals;   /* tight float3, object space */
  uint64_t num_vertices;

  const gsp_diffuse_bsdf* diffuse_bsdfs;
  const gsp_smooth_dielectric_bsdf* smooth_dielectric_bsdfs;
  const gsp_smooth_conductor_bsdf* smooth_conductor_bsdfs;
  const gsp_smooth_plastic_bsdf* smooth_plastic_bsdfs;
  const gsp_rough_conductor_bsdf* rough_conductor_bsdfs;
  const gsp_smooth_floor_bsdf* smooth_floor_bsdfs;
  const gsp_rough_floor_bsdf* rough_floor_bsdfs;
  const gsp_rough_plastic_bsdf* rough_plastic_bsdfs;
  uint32_t num_bsdfs[GSP_BSDF_TYPE_COUNT]; /* indexed by GSP_BSDF_* */

  const gsp_triangle_light* lights;
  uint32_t num_lights;

  gsp_camera camera;

  /* dormant-feature extension (see above); all zero = the reference's behaviour */
  const float* uvs;             /* tight float2 per vertex (same indexing as positions), or NULL */
  const gsp_texture* textures;
  uint32_t num_textures;
  const uint32_t* texels;       /* RGBA8, R in bits 0-7 */
  uint64_t num_texels;
  const float* texel_decode;    /* 256 floats (e.g. an sRGB -> linear table), or NULL = byte / 255 */
  gsp_envmap envmap;
} gsp_scene_desc;

/*
 * Integrator constants.  Defaults (gsp_default_render_params) are the
 * reference's shader literals: MAX_DEPTH 50 (raygen.rgen:27), Russian
 * roulette when depth > 10 (raygen.rgen:66), firefly cutoff 20 (raygen.rgen:60).
 */
typedef struct gsp_render_params {
  uint32_t struct_size;     /* (ABI 8) sizeof(gsp_render_params) of the host's header; gsp_default_render_params sets it.  The library
                               reads min(struct_size, its own size) bytes and takes the fields beyond as 0 (= the reference's
                               behaviour, for every field added so far and from here on), so a host built against an older
                               header of ABI >= 8 keeps working.  0 (a zero-initialised struct) = the ABI-8 layout */
  uint32_t spp;             /* samples per pixel to add in this call      */
  uint32_t first_timestamp; /* RenderParams.timestamp of the first sample */
  uint32_t max_depth;
  uint32_t rr_start_depth;
  float clamp;
  uint32_t timestamps_in_flight; /* samples traced concurrently; 0 = auto */
  uint32_t collect_traversal_stats; /* 1: count BVH nodes / triangles per ray (slower); 2: and per-record visit counts of the
                                       closest-hit rays (gsp_debug_visit_histograms) */
  uint32_t collect_kernel_times;    /* 1: HIP-event time every extend/shade/connect launch (2: and print one line per iteration to stderr) */
  uint32_t disable_nee;     /* (ABI 7; ABI 5-6: `nee` with the opposite sense) RenderParams.nee (S/renderer/PathTracer.h:36-41),
                               which the shipped shader replaces by `#define NEE true` (rayhit.rchit:656).  0 (default, and what a
                               zero-initialised struct says) = the reference as shipped.  1 = the other side of its `if (NEE)`
                               branches (rayhit.rchit:733,763-768): no shadow ray, every emitter met counts with full weight,
                               directWeight stays 1; the light sample is still DRAWN (rayhit.rchit:720 is outside the branch),
                               so the random streams of the two settings coincide */
} gsp_render_params;

typedef struct gsp_stats {
  uint64_t extension_rays;
  uint64_t shadow_rays;
  uint64_t shaded_vertices;
  uint64_t samples;          /* pixels * spp completed                           */
  uint64_t nodes_visited;    /* extension rays: BVH node records read (stats mode) */
  uint64_t tris_tested;      /* extension rays: triangle packets read (stats mode) */
  uint64_t stat_rays;        /* extension rays the two counters above cover        */
  uint64_t shadow_nodes_visited; /* same three for shadow rays                     */
  uint64_t shadow_tris_tested;
  uint64_t shadow_stat_rays;
  double render_seconds;     /* host wall time inside gsp_render.. sync          */
  double extend_kernel_ms;   /* sum of HIP-event durations of the extend kernel  */
  uint64_t extend_launches;
  double shade_kernel_ms;
  double connect_kernel_ms;
  double bvh_build_ms;       /* last gsp_upload_scene / gsp_update_instances (build or refit) */
  uint64_t num_triangles;
  uint64_t num_bvh_nodes;
  uint64_t device_bytes;     /* device memory currently held by the context      */
  uint64_t algorithmic_bytes; /* SURVEY 8(d) bytes of the rays the stats mode covered: per extension ray 32 (ray)
                                 + 16 (hit) + 64 per node + 48 per triangle record read, per shadow ray 32 (+ 36 when it is
                                 occluded: the commit) + the same node / triangle terms.  What the traversal ASKS of the memory hierarchy; the
                                 bytes that reach HBM are a quarter of it (rocprofv3 FETCH_SIZE) */
  /* (ABI 4) primary-hit memo: every sample of a pixel shoots the same camera ray (raygen.rgen:31-38, no jitter), so the
     camera rays of a frame are traced once (memo_build_rays) and later samples copy the hit.  extension_rays keeps
     counting PATH SEGMENTS (what the reference traces: it equals the oracle's count); memoised_rays of them were
     answered from the memo.  Rays actually traced = extension_rays - memoised_rays + memo_build_rays + shadow_rays.
     gsp_ctx_options.primary_memo = 2 turns the memo off. */
  uint64_t memoised_rays;
  uint64_t memo_build_rays;
  uint64_t bvh_depth;        /* (ABI 4) levels of the wide BVH: a traversal stacks at most one entry per level; beyond the 20
                                levels a lane keeps in LDS the stack continues in HBM, beyond 34 the tail of a drain is left
                                to the wavefront kernels */
  uint64_t nodes_from_lds;             /* (ABI 5, stats mode) of nodes_visited / shadow_nodes_visited: node records the traversal read ... */
  uint64_t shadow_nodes_from_lds;      /* ... from its LDS copy of the top of the tree (no request on the vector-memory path)         */
  uint64_t shadow_stat_occluded;       /* (ABI 5, stats mode) shadow rays of shadow_stat_rays that were occluded ...          */
  uint64_t shadow_stat_occluded_nodes; /* ... and the node records those read (the rest of shadow_nodes_visited: unoccluded) */
  uint64_t scene_updates;    /* (ABI 5) gsp_update_camera / _instances / _tables calls that changed something since the last
                                gsp_upload_scene (a test hook: which path did the host layer take?) */
  uint64_t scene_refits;     /* (ABI 6) of those, gsp_update_instances calls that kept the tree's topology (refit) */
  uint64_t shadow_stat_no_triangle; /* (ABI 7, stats mode) shadow rays of shadow_stat_rays that ended without ONE triangle test
                                       (the ray's shear constants Sx, Sy were computed for nothing: VERDICT r04 item 7) */
  uint64_t scene_drains;     /* (ABI 7) of scene_updates: calls that first let the samples in flight finish (a test hook: an edited
                                BSDF record / light / transform does not, once the version rings exist: INTEGRATION.md) */
  uint64_t scene_splits;     /* (ABI 7) of scene_updates: gsp_update_instances calls that built the scene as TWO trees -- the instances
                                edited so far and the rest (the first edit that arrives while samples are in flight, and every later
                                one that touches an instance not edited before); edits of those instances then refit the small tree */
} gsp_stats;

typedef struct gsp_context gsp_context;

/* Fill `p` with the reference's literals.  */
void gsp_default_render_params(gsp_render_params* p);

/* ABI version of the loaded library (== GSP_ABI_VERSION of this header).  gsp_stats carries no struct_size (the library FILLS it
 * at its own size): a C host MUST check gsp_abi_version() == GSP_ABI_VERSION once after loading the library;
 * gpuspectral_amd/pt.py and host/PathTracer.cpp do.  gsp_render_params and gsp_ctx_options are sized by their first field. */
int gsp_abi_version(void);

/* What the loaded library was built from: "arch=gfx950 digest=<sha256[:16] of csrc/{pt_render,pt_bvh,pt_multi}.hip + the
 * headers, in the Makefile's order> flags=<compiler flags>".  A deployment (and tests/test_abi_and_host.py) compares the
 * digest with the tree to tell a stale prebuilt .so from a fresh build.  Static storage, never NULL. */
const char* gsp_build_info(void);

/* Number of HIP devices visible (0 if none / runtime unavailable). */
int gsp_device_count(void);

/* Create a context on HIP device `device`.  Fails (GSP_ERR_DEVICE) when no
 * gfx950-capable device is present: there is no CPU fallback. */
int gsp_ctx_create(int device, gsp_context** out);
void gsp_ctx_destroy(gsp_context* ctx);

/*
 * (ABI 5) Per-context resources and scheduling choices.  None of them changes a result: images, ray counts and sample
 * order are the same for every setting (the GPU suite runs the non-default ones).  gsp_default_ctx_options fills in the
 * defaults; a field left 0 also means "default", so a zero-initialised struct with struct_size set is valid.
 * The library never reads the environment: a host application sets these per context.  (The Python test binding,
 * gpuspectral_amd/pt.py, maps the GSP_* variables the A/B scripts of earlier rounds used onto this struct.)
 */
#define GSP_GATHER_AUTO 0 /* RCCL when the shares sit on distinct devices and the communicator comes up, else copies */
#define GSP_GATHER_RCCL 1 /* ncclSend / ncclRecv group; creation fails if RCCL cannot be initialised               */
#define GSP_GATHER_COPY 2 /* hipMemcpyPeerAsync into the gathering device                                         */
typedef struct gsp_ctx_options {
  uint32_t struct_size;      /* sizeof(gsp_ctx_options) of the caller's header                                        */
  uint32_t lanes;            /* independent pipelines per context, 1 (default) or 2                                   */
  uint64_t pool_paths;       /* paths in flight the pool aims at (default 96 Mi; at most 192 per owned pixel)         */
  uint64_t ring_bytes;       /* upper bound of the sample-result ring (default 16 GiB)                                */
  double memory_share;       /* share of the device's TOTAL memory which path pool + ring may take (r05; until r04: of the
                                memory FREE when a render starts, so contexts sized themselves by creation order), never
                                more than 90 % of what is free; 0.01 .. 0.9 (default 0.4): lower it when contexts share a GPU */
  uint32_t primary_memo;     /* 0 default (on), 1 on, 2 off: trace the camera ray of a pixel once per frame           */
  uint32_t finish_paths;     /* k_finish takes over a drain below this many live paths; 0 default (262144),
                                0xffffffff = never                                                                    */
  uint32_t reinsert_rounds;  /* BVH build: parallel-reinsertion rounds + 1 (0 default = 6 rounds, 1 = none, ...)      */
  uint32_t gather_route;     /* gsp_multi_create_ex only: GSP_GATHER_*                                                */
  double refit_growth;       /* (ABI 6) gsp_update_instances keeps the tree's topology and only recomputes its boxes while
                                their summed surface area stays below this multiple of what it was after the last full
                                build; beyond it the tree is rebuilt.  0 default (1.25); <= 1 = always rebuild             */
  uint32_t geometry_versions; /* (r05) slots of the geometry ring that lets gsp_update_instances go on while samples are in
                                flight (176 B of device memory per triangle and slot): 0 default = as many as fit a quarter of
                                the free memory, at most 64; 1 = no ring: a transform edit first completes the samples queued;
                                other values are rounded down to a power of two (2 and 3: no ring)                        */
  uint32_t reserved_;
} gsp_ctx_options;
void gsp_default_ctx_options(gsp_ctx_options* o);
int gsp_ctx_create_ex(int device, const gsp_ctx_options* options, gsp_context** out);

/* Copy the scene to the device, bake world-space triangles and build the BVH
 * on the device.  The caller keeps ownership of every array in `scene`.  Limit: fewer than 2^25 (33.5 M) triangles -- the
 * traversal's packed stack entry holds node indices below 2^25 and a tree has fewer nodes than triangles -- checked here,
 * before any device work (GSP_ERR_SCENE). */
int gsp_upload_scene(gsp_context* ctx, const gsp_scene_desc* scene);

/*
 * (ABI 5) Per-frame scene edits.  The reference re-reads camera, object transforms, materials, BSDF tables and lights on
 * every createRenderPass (S/renderer/PathTracer.cpp:10-19,58-93) and only keeps the BLAS of a mesh (Renderer.cpp:122-131);
 * a host that mirrors it calls these between gsp_render calls instead of uploading the whole scene again.  The samples
 * already queued belong to the scene as it was and finish as such; accumulate buffer and timestamps are left alone -- like the
 * reference, where an edit simply shows up in the next frame's sample; call gsp_frame_begin to restart the running mean -- and
 * the primary-hit memo is invalidated.  All need a prior gsp_upload_scene.  Every call COMPARES first and returns at once,
 * before any wait, when its input equals what the device holds: a host that mirrors the reference makes all three every frame.
 *   gsp_update_camera     new camera; no geometry work and NO WAIT: paths in flight have left the camera behind (only ray
 *                         generation reads it), so a viewer that moves its camera every frame keeps the path pool full.
 *                         The two calls below do not wait either in the usual case (r05): the tables and the geometry
 *                         live in rings of versions, a sample carries the versions it was generated under and finishes on
 *                         them (gsp_stats.scene_drains counts the calls that did have to wait).
 *   gsp_update_instances  new transform / emission / bsdf / twofaced per instance.  `num_instances` and every
 *                         first_vertex / vertex_count must equal the uploaded ones (the meshes stay: they are resident on
 *                         the device); BSDF handles are checked against the resident tables.  Re-bakes the world-space
 *                         packets into their slots and REFITS the tree -- same topology, every node box and child order
 *                         recomputed bottom-up (0.35 ms for a million triangles) -- as the reference rebuilds only its
 *                         TLAS; when the boxes have grown past gsp_ctx_options.refit_growth (an object has moved far
 *                         from where the tree was built for) the BVH is rebuilt instead (16 ms).  Images do not
 *                         depend on which of the two happened: the closest-hit rule is independent of the tree.
 *                         An edit that arrives while samples are in flight (a viewer) builds the scene as TWO trees once --
 *                         the instances edited so far and the rest (gsp_stats.scene_splits; no wait the first time) -- and later
 *                         edits of those instances refit only their small tree (0.08 ms) into the next slot of a ring of
 *                         up to 64 versions while the samples in flight finish in theirs.  Scenes that do not split (more
 *                         than a quarter of the triangles edited, textures, a host that has touched new objects 16 times since the upload --
 *                         kMaxSceneSplits = 16, gpuspectral_amd/csrc/pt_versions.h) keep a ring of whole refitted
 *                         trees instead (176 B per triangle and version, at most a quarter of the free device memory).
 *                         A rebuild, a ring without a free slot or a scene above 8 M triangles first complete the samples
 *                         already queued; an edit that arrives with nothing in flight refits in place.
 *   gsp_update_tables     the eight BSDF arrays + num_bsdfs and the lights + num_lights of `scene` replace the resident
 *                         ones (all other fields of `scene` are ignored); every resident instance's handle must stay in
 *                         range.  No geometry work.  Textured scenes (dormant-feature extension): has_texture values are
 *                         checked against the resident textures.  Same record counts as the resident tables: the next slot
 *                         of a ring of versions, no wait; other counts first complete the samples already queued.  The ring
 *                         is made by the first such edit that arrives with samples in flight (gsp_upload_scene holds ONE
 *                         copy of the tables): up to 64 versions, at most a sixteenth of the free device memory and 1 GiB
 *                         (a 64-MB light table gets 16); where not even two fit, edits wait as they did before the ring.
 */
int gsp_update_camera(gsp_context* ctx, const gsp_camera* camera);
int gsp_update_instances(gsp_context* ctx, const gsp_instance* instances, uint32_t num_instances);
int gsp_update_tables(gsp_context* ctx, const gsp_scene_desc* scene);

/*
 * Allocate (and zero) the accumulate buffer for a width x height frame.
 * `pixel_ids` optionally restricts this context to `num_pixels` pixels of the
 * frame (global index = width*y + x, strictly increasing); NULL = the whole
 * frame.  Seeds depend only on the global pixel index (raygen.rgen:37), so any
 * partition of the frame over contexts/GPUs reproduces the same image.
 */
int gsp_frame_begin(gsp_context* ctx, uint32_t width, uint32_t height,
                    const uint32_t* pixel_ids, uint64_t num_pixels);

/* Trace params->spp more samples for every owned pixel and fold them into the
 * accumulate buffer with the reference's running mean (raygen.rgen:84-108).
 * Returns once every sample has been injected into the path pool; the last
 * paths may still be in flight (they share launches with the next call's
 * samples) and samples are always folded in call / timestamp order.  gsp_sync,
 * the download / copy / upload_accum calls, gsp_get_stats, gsp_reset_stats,
 * gsp_frame_begin, gsp_upload_scene and gsp_ctx_destroy complete them first. */
int gsp_render(gsp_context* ctx, const gsp_render_params* params);

/* Block until all queued work of the context has finished. */
int gsp_sync(gsp_context* ctx);

/* Full frame, RGBA32F row-major, width*height*4 floats; pixels this context
 * does not own are written as 0. */
int gsp_download(gsp_context* ctx, float* out_rgba);
/* Only the owned pixels, in pixel_ids order: num_pixels*4 floats. */
int gsp_download_compact(gsp_context* ctx, float* out_rgba);
/* The accumulate buffer as it stands, WITHOUT waiting for the paths still in
 * flight (compact layout, num_pixels*4 floats): every pixel holds the running
 * mean of its first *samples_folded timestamps.  This is what the reference's
 * presentation pass shows each frame (DrawTexture blit of accumulateBuffer,
 * S/renderer/PathTracer.cpp:41-55) while the next frames are already tracing;
 * a viewer calls it once per displayed frame and gsp_download only at the end. */
int gsp_peek(gsp_context* ctx, float* out_rgba, uint32_t* samples_folded);
/* Device-to-device copy of the compact accumulate buffer into caller-owned
 * device memory (e.g. a tensor handed to an RCCL gather). */
int gsp_copy_accum_to_device(gsp_context* ctx, void* device_dst, uint64_t bytes);
/* (r05) gsp_peek into caller-owned DEVICE memory: the compact accumulate buffer as it stands, no wait for the paths in flight,
 * no trip through host memory -- the reference's blit samples accumulateBuffer on the GPU (S/renderer/PathTracer.cpp:41-55); a
 * viewer that presents from device memory (an interop texture, a tensor) calls this once per displayed frame.  The copy is
 * complete when the call returns.  `bytes` >= num_pixels * 16. */
int gsp_peek_to_device(gsp_context* ctx, void* device_dst, uint64_t bytes, uint32_t* samples_folded);
/* Overwrite the compact accumulate buffer from host memory (checkpoint/resume:
 * buffer + next timestamp are the whole integrator state). */
int gsp_upload_accum(gsp_context* ctx, const float* rgba, uint64_t num_pixels);

int gsp_get_stats(gsp_context* ctx, gsp_stats* out);
int gsp_reset_stats(gsp_context* ctx);

/*
 * Test hook: closest-hit / any-hit queries against the uploaded BVH.
 * rays = n * 8 floats {ox,oy,oz,tmin, dx,dy,dz,tmax}; hits = n * 4 words
 * {t (f32), u (f32), v (f32), prim (i32, -1 = miss)}.  any_hit != 0 runs the
 * shadow-ray kernel (prim = 0 when occluded, -1 when not).
 */
int gsp_trace(gsp_context* ctx, const float* rays, uint64_t n, int any_hit, void* hits);

/* Measurement hook: after renders with collect_traversal_stats = 2, how often the closest-hit rays read each node record
 * (index = position in the level-ordered node array) and tested each triangle slot.  Copies min(count, available) words,
 * then drops the counters.  It answers "which records would an LDS copy have to hold?" (DESIGN.md 4). */
int gsp_debug_visit_histograms(gsp_context* ctx, uint32_t* node_counts, uint64_t num_nodes, uint32_t* slot_counts, uint64_t num_slots);

/* Last error message of this context (or of the failed gsp_ctx_create when
 * ctx == NULL).  Never NULL. */
const char* gsp_last_error(const gsp_context* ctx);

/* ---- several GPUs of one node --------------------------------------------------------------------------------
 * Pixels are independent (the seed depends on the global pixel index and the timestamp only, raygen.rgen:37), so a
 * frame is partitioned by image tile: tile (tx, ty) of the tile x tile grid belongs to share
 * (ty * tiles_x + tx + ty) % world.  gsp_tile_partition writes the pixel ids of one share in increasing order
 * (out_ids may be NULL) and returns their number; it needs no GPU.  bench.py's one-process-per-GPU path hands these
 * lists to gsp_frame_begin; a C++ host uses the gsp_multi_* calls below, which do the same inside one process. */
uint64_t gsp_tile_partition(uint32_t width, uint32_t height, uint32_t rank, uint32_t world, uint32_t tile,
                            uint32_t* out_ids);

typedef struct gsp_multi gsp_multi;

/* One context per entry of `devices` (HIP device indices; an index may repeat: several shares on one GPU).  The
 * first device gathers.  Same error convention as gsp_ctx_create (gsp_multi_last_error(NULL) after a failure). */
int gsp_multi_create(const int* devices, int n, gsp_multi** out);
/* (ABI 5) the same with options: every share's context is created with `options` (memory_share is divided among the shares
 * of one device), options->gather_route picks the gather (GSP_GATHER_*). */
int gsp_multi_create_ex(const int* devices, int n, const gsp_ctx_options* options, gsp_multi** out);
void gsp_multi_destroy(gsp_multi* m);
int gsp_multi_num_shares(const gsp_multi* m);
/* gsp_upload_scene / gsp_frame_begin (32x32 tiles) / gsp_render / gsp_sync on every share, one host thread each. */
int gsp_multi_upload_scene(gsp_multi* m, const gsp_scene_desc* scene);
/* (ABI 5) gsp_update_camera / _instances / _tables on every share */
int gsp_multi_update_camera(gsp_multi* m, const gsp_camera* camera);
int gsp_multi_update_instances(gsp_multi* m, const gsp_instance* instances, uint32_t num_instances);
int gsp_multi_update_tables(gsp_multi* m, const gsp_scene_desc* scene);
int gsp_multi_frame_begin(gsp_multi* m, uint32_t width, uint32_t height);
int gsp_multi_render(gsp_multi* m, const gsp_render_params* params);
int gsp_multi_sync(gsp_multi* m);
/* The one exchange of a render: completes the queued samples, brings every share's HDR tiles into devices[0] and
 * assembles the frame there.  With one share per device the gather is ONE RCCL group over xGMI (every share
 * ncclSend()s its tiles, devices[0] ncclRecv()s them); a device list with repeats (several shares on one GPU, for
 * tests) uses peer copies instead -- RCCL wants one rank per device -- and so does a node on which librccl cannot be
 * loaded or the communicator does not come up (GSP_GATHER_AUTO; the reason is kept in gsp_multi_last_error).
 * gsp_ctx_options.gather_route forces a route (GSP_GATHER_RCCL with ONE share performs a self send / recv: the RCCL smoke
 * test of a one-GPU box).
 * *device_frame (optional) = the RGBA32F frame on devices[0] (NULL for a single share without RCCL, whose frame is its
 * context's accumulate buffer). */
int gsp_multi_gather(gsp_multi* m, void** device_frame);
/* 1: the gathers of `m` go through RCCL, 0: through peer copies (-1: m is NULL).  The optional outputs receive the
 * number of gathers each route has carried so far. */
int gsp_multi_gather_route(const gsp_multi* m, uint64_t* rccl_gathers, uint64_t* copy_gathers);
/* gsp_multi_gather + one copy to the host: width*height*4 floats, identical to a single-GPU gsp_download. */
int gsp_multi_download(gsp_multi* m, float* out_rgba);
/* total (optional): counters summed over the shares, times of the slowest share (they run concurrently);
 * per_share (optional): gsp_multi_num_shares() records. */
int gsp_multi_get_stats(gsp_multi* m, gsp_stats* total, gsp_stats* per_share);
int gsp_multi_reset_stats(gsp_multi* m);
const char* gsp_multi_last_error(const gsp_multi* m);

#ifdef __cplusplus
}
#endif
#endif /* GPUSPECTRAL_PT_H */
