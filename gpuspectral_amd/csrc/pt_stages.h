// pt_stages.h -- per-path bodies of the wavefront stages.
//
//   generate : raygen.rgen:29-48          camera ray + RNG seed + payload init
//   shade    : rayhit.rchit:666-797       one shading vertex (closest-hit shader)
//              raygen.rgen:59-80          firefly clamp, Russian roulette, depth logic
//   connect  : rayhit.rchit:737-757       shadow-ray verdict -> NEE contribution, MIS weight
//   resolve  : raygen.rgen:84-108         running-mean accumulate
//
// The reference runs these as one recursive shader invocation per pixel; here
// they are separate passes over ray queues in HBM.  The state a path carries
// between passes is the reference's HitPayload (pt_common.glsl:4-14).
//
// Sample sum order: the reference adds `prd.emitted` (NEE term first, emission
// term second, rayhit.rchit:752,763-768) to `result` once per bounce behind the
// firefly test (raygen.rgen:60-63).  The NEE term needs the shadow ray, so the
// shade pass hands both terms to the connect pass, which forms the same sum.
#pragma once
#include "pt_shading.h"
#include "pt_trace.h"

namespace gsp {

// The per-instance shading data of RenderState::Instance (emission, bsdf, twofaced; PathTracer.h:27-34) is baked
// into the w components of every triangle's shading packet: one dependent fetch less per shaded vertex.
GSP_HD uint32_t pack_material(uint32_t bsdf, uint32_t twofaced) {
  return (bsdf & 0x7fffffffu) | (twofaced == 1u ? 0x80000000u : 0u);
}

struct RenderConsts {
  uint32_t width, height;
  uint32_t max_depth, rr_start_depth;
  float clamp;
  uint32_t nee;        // RenderParams.nee (PathTracer.h:36-41); the shipped shader hard-wires 1 (`#define NEE true`, rchit:656)
  float zplane;        // (max(W,H)/2) / tan(fov/2), raygen.rgen:22 (tan evaluated on the host)
  float cam_origin[3]; // camera.eye = toWorld[3]
  float cam_to_world[16];
};

struct SceneView {
  const q4* nodes;
  const q4* tri_isect;  // 3 quads per slot
  const q4* tri_shade;  // 4 quads per slot: {N, bits(bsdf | twofaced << 31)}, {n0, emission.r}, {n1, .g}, {n2, .b}
  BsdfTables bsdf;
  const gsp_triangle_light* lights;  // baked records (pt_shading.h bake_light), as is the diffuse table (bake_diffuse)
  uint32_t num_lights;
  float inv_num_lights = 0.0f;       // 1.0f / (float)num_lights (rayhit.rchit:151), formed once on the host
  uint32_t static_slots = 0;        // <VER = true>, split scene: triangle slots (and 64-B node records) of the STATIC tree that sits in front
                                    // of the geometry ring -- slots below it are the same in every version; 0 = the ring holds whole trees
  uint32_t geo = 0;                 // <VER = true>: geometry ring (below): phys slot that stamp 0 stands for << 29 | triangle slots per version
  const uint8_t* tables = nullptr;  // the BSDF tables + lights back to back (device: one allocation), for LDS staging
  uint32_t tables_bytes = 0;
  uint32_t ver_stride = 0;          // <VER = true> instantiations: bytes between two versions of the tables (slot 0 is what the pointers name)
  TextureView tex;                  // dormant-feature extension; read only by the <TEX = true> instantiations
};

// path flags word: depth [0,7] | wasDelta << 8 | countEmitted << 9 | table version << 10 [10,15] | geometry version << 16 [16,21]
GSP_HD uint32_t pack_flags(uint32_t depth, uint32_t wasDelta, uint32_t countEmitted) {
  return depth | (wasDelta << 8) | (countEmitted << 9);
}
// r05: a path carries the VERSION of the BSDF / light tables it was generated under (gsp_update_tables without a drain: the
// samples in flight finish on the tables they started with, new samples read the new ones; the versions live in a ring of
// kTableVersions slots, SceneView::ver_stride bytes apart).  Stamped by k_generate, handed on by shade_vertex<VER = true>.
// While every sample in flight belongs to ONE version -- always, for a scene that is not being edited -- that version sits in
// slot 0, the field is 0 and the <VER = false> kernels neither read nor write it: their code is what it was before versions existed.
constexpr uint32_t kVerShift = 10, kTableVersions = 64, kVerMask = (kTableVersions - 1u) << kVerShift;
// ... and the version of the GEOMETRY (gsp_update_instances without a drain: bits [16,21]): the node records, intersection
// triangles and shading packets of up to kGeoVersions edits live in a ring of G = 2^g slots -- version slot p at nodes + p * stride
// * 64 B (a tree of n triangles has fewer than n nodes), tri_isect + 3 * p * stride, tri_shade + 4 * p * stride quads, stride =
// triangle slots per version -- and a path's stamp s names slot (s + base) % G, base = the slot of the one live version the last
// time only one was (then the field is 0, as for the tables).  A path of 52 bounces lives 52 iterations -- 52 frames of a viewer
// that renders one sample per frame -- so the ring is as long as the tables' when memory and the 32-bit node offsets allow
// (G * stride * 64 B < 4 GB: 64 versions up to a million triangles, 8 up to eight million).  g, base and stride travel in one
// word (SceneView::geo / GeoRing).
constexpr uint32_t kGeoShift = 16, kGeoVersions = 64, kGeoMask = (kGeoVersions - 1u) << kGeoShift;
constexpr uint32_t kStampMask = kVerMask | kGeoMask;
constexpr uint32_t kGeoStrideBits = 23, kGeoStrideMask = (1u << kGeoStrideBits) - 1u;  // stride | base << 23 | g << 29
constexpr uint32_t kGeoMaxStride = kGeoStrideMask;
GSP_HD uint32_t pack_geo(uint32_t base, uint32_t stride, uint32_t log2_versions) {
  return stride | (base << kGeoStrideBits) | (log2_versions << 29);
}
// slot offset (in triangle slots) of the geometry version a stamp names
GSP_HD uint32_t geo_slot_offset(uint32_t geo, uint32_t stamp) {
  const uint32_t mask = (1u << (geo >> 29)) - 1u;
  return ((stamp + ((geo >> kGeoStrideBits) & (kGeoVersions - 1u))) & mask) * (geo & kGeoStrideMask);
}
GSP_HD uint32_t geo_stamp(uint32_t flags) { return (flags & kGeoMask) >> kGeoShift; }

struct PathState {
  f3 o, d;
  f3 weight;
  float directWeight;
  uint32_t seed;
  uint32_t flags;
  uint32_t sid;  // sample slot: k * num_pixels + local pixel
};

struct ShadowRay {
  f3 o, d;
  float tmax;     // Ldist - 0.01 (tmin is the literal 0.01)
  f3 nee;         // contribution if unoccluded
  f3 emis;        // emission term of the same vertex
  float dw_nee;   // directWeight of the continuing path if NEE happens
  uint32_t sid;
  int32_t next;   // index of the continuing path in the next queue, -1 if the path ended here
};

// raygen.rgen:20-25,31-48
GSP_HD void generate_path(const RenderConsts& rc, uint32_t gid, uint32_t timestamp, uint32_t sid, PathState& p) {
  const uint32_t px = gid % rc.width, py = gid / rc.width;
  const float x = (float)px - (float)rc.width / 2.0f;
  const float y = (float)py - (float)rc.height / 2.0f;
  f3 dl = normalize(mk3(-x, y, rc.zplane));
  f3 dir = xform_dir(rc.cam_to_world, dl);
  dir.y = dir.y * -1.0f;
  p.o = mk3(rc.cam_origin[0], rc.cam_origin[1], rc.cam_origin[2]);
  p.d = dir;
  p.weight = splat(1.0f);
  p.directWeight = 1.0f;
  p.seed = pcg_hash(tea(rc.width * py + px, timestamp));
  p.flags = pack_flags(0, 0, 1);
  p.sid = sid;
}

// firefly test + add, raygen.rgen:60-63
GSP_HD void add_emitted(float clampv, f3 e, q4& result) {
  if (e.x < clampv && e.y < clampv && e.z < clampv) {
    result.x = result.x + e.x;
    result.y = result.y + e.y;
    result.z = result.z + e.z;
  }
}

struct ShadeOut {
  bool alive;        // path continues: `next` is valid
  PathState next;
  bool has_shadow;   // a shadow ray must be traced: `shadow` is valid (sid/next filled by the caller)
  ShadowRay shadow;
  f3 emitted;        // when !has_shadow: prd.emitted of this bounce (emission term only)
};

// TEX: the scene carries textures (include/gpuspectral_pt.h, dormant-feature extension).  The reference's shader
// passes uv = vec2(0) and never reads a texture (rayhit.rchit:716,729): TEX = false is that code, instruction for
// instruction; TEX = true interpolates the hit's uv and, for a record with has_texture, takes kD from the texture.
// VER: tables of several versions are live (an edit through gsp_update_tables while samples were in flight): the vertex reads
// the version its path carries.  VER = false is the code of a scene that is not being edited, instruction for instruction.
template <bool TEX = false, bool VER = false>
GSP_HD void shade_vertex(const SceneView& S, const RenderConsts& rc, const PathState& in, const HitRec& hit,
                         ShadeOut& out) {
  uint32_t rng = in.seed;                                                 // rchit:668
  BsdfTables Tv;  // (VER only: this lane's version of the tables)
  const gsp_triangle_light* lights_v = nullptr;
  if (VER) {
    const size_t voff = (size_t)((in.flags & kVerMask) >> kVerShift) * S.ver_stride;
    Tv.diffuse = (const gsp_diffuse_bsdf*)((const char*)S.bsdf.diffuse + voff);
    Tv.smooth_dielectric = (const gsp_smooth_dielectric_bsdf*)((const char*)S.bsdf.smooth_dielectric + voff);
    Tv.smooth_conductor = (const gsp_smooth_conductor_bsdf*)((const char*)S.bsdf.smooth_conductor + voff);
    Tv.smooth_plastic = (const gsp_smooth_plastic_bsdf*)((const char*)S.bsdf.smooth_plastic + voff);
    Tv.rough_conductor = (const gsp_rough_conductor_bsdf*)((const char*)S.bsdf.rough_conductor + voff);
    Tv.smooth_floor = (const gsp_smooth_floor_bsdf*)((const char*)S.bsdf.smooth_floor + voff);
    Tv.rough_floor = (const gsp_rough_floor_bsdf*)((const char*)S.bsdf.rough_floor + voff);
    Tv.rough_plastic = (const gsp_rough_plastic_bsdf*)((const char*)S.bsdf.rough_plastic + voff);
    lights_v = (const gsp_triangle_light*)((const char*)S.lights + voff);
  }
  const BsdfTables& T = VER ? Tv : S.bsdf;
  const gsp_triangle_light* lights = VER ? lights_v : S.lights;
  GSP_PROF_BEGIN(PR_PACKET);
  const q4* sp = S.tri_shade + 4ll * ((long long)hit.slot +
                                      (VER && (uint32_t)hit.slot >= S.static_slots ? (long long)geo_slot_offset(S.geo, geo_stamp(in.flags)) : 0ll));
  const q4 s0 = sp[0], s1 = sp[1], s2 = sp[2], s3 = sp[3];
  const uint32_t material = f2u(s0.w);                                    // :672 (instance record, baked per triangle)
  const uint32_t bsdf = material & 0x7fffffffu;
  const bool twofaced = (material >> 31) != 0u;
  const f3 emission = mk3(s1.w, s2.w, s3.w);
  const f3 rayDir = in.d;                                                 // :696
  const f3 position = in.o + rayDir * hit.t;                              // :692
  const float b0 = (1.0f - hit.u) - hit.v;                                // :690
  f3 SN = normalize((b0 * mk3(s1.x, s1.y, s1.z) + hit.u * mk3(s2.x, s2.y, s2.z)) + hit.v * mk3(s3.x, s3.y, s3.z));
  f3 N = mk3(s0.x, s0.y, s0.z);                                           // :694 (precomputed at bake)
  if (dot(N, -rayDir) < 0.0f) {                                           // :698-707
    if (twofaced && emission.x == 0.0f && emission.y == 0.0f && emission.z == 0.0f) {
      N = N * -1.0f;
      SN = SN * -1.0f;
    }
  }
  const Frame onb = make_frame(SN);                                       // :712
  const f3 wo = normalize(to_local(onb, -rayDir));                        // :713
  bool kd_on = false;
  f3 kd = splat(0.0f);
  if (TEX) {
    const uint32_t tid = bsdf_texture(T, bsdf);
    if (tid != 0u && tid <= S.tex.num_textures && S.tex.tri_uv != nullptr) {
      const q4* uv = (const q4*)S.tex.tri_uv + 2ll * hit.slot;  // {u0 v0 u1 v1}, {u2 v2 - -}
      const q4 ua = uv[0], ub2 = uv[1];
      const float tu = (b0 * ua.x + hit.u * ua.z) + hit.v * ub2.x;
      const float tv = (b0 * ua.y + hit.u * ua.w) + hit.v * ub2.y;
      kd = sample_texture(S.tex, tid - 1u, tu, tv);
      kd_on = true;
    }
  }
  BsdfResult bs;
  f3 wi_l;
  GSP_PROF_END(PR_PACKET);
  GSP_PROF_BEGIN(PR_SAMPLE);
  BsdfCarry cy;
  bsdf_sample(T, bsdf, rng, wo, wi_l, bs, cy, kd_on, kd);            // :716
  GSP_PROF_END(PR_SAMPLE);
  GSP_PROF_BEGIN(PR_LIGHT);
  const float NoW = gabs(wi_l.z);                                         // :717
  const f3 wi = to_world(onb, wi_l);                                      // :718

  const LightSample ls = sample_light(lights, S.num_lights, S.inv_num_lights, rng, position);  // :720
  const f3 toL = ls.position - position;
  const f3 L = normalize(toL);                                            // :722
  const f3 wL = to_local(onb, L);                                         // :723
  const float Ldist = length(toL);                                        // :724
  const float NoL = gabs(dot(SN, L));                                     // :725
  const float lightPdf = ls.pdf;
  BsdfResult lb;
  GSP_PROF_END(PR_LIGHT);
  GSP_PROF_BEGIN(PR_EVAL);
  bsdf_eval(T, bsdf, wo, wL, lb, cy, kd_on, kd);                     // :729
  GSP_PROF_END(PR_EVAL);
  GSP_PROF_BEGIN(PR_TAIL);

  const bool transmits = bsdf_transmits(bsdf);
  const float NdotV = dot(N, -rayDir);
  // :733-736 gate (`if (NEE)` first); `lightPdf != 0` (:750) is known before tracing
  const bool nee_on = rc.nee != 0u;
  const bool want_shadow = nee_on && !bs.delta && ((NdotV > 0.0f && dot(N, L) > 0.0f) || transmits) && (lightPdf != 0.0f);
  f3 nee = splat(0.0f);
  if (want_shadow) {
    const float w = power_heuristic(lightPdf, bs.pdf);                    // :751
    nee = ((((w * NoL) * lb.f) * in.weight) * ls.emission) / lightPdf;    // :752
  }
  const uint32_t depth = in.flags & 0xffu;
  const uint32_t wasDelta = (in.flags >> 8) & 1u;
  const uint32_t countEmitted = (in.flags >> 9) & 1u;
  const float lightFlag = NdotV > 0.0f ? 1.0f : 0.0f;                     // :760
  f3 emis = splat(0.0f);
  if (nee_on && countEmitted == 0 && wasDelta == 0) emis = emis + ((in.directWeight * emission) * lightFlag) * in.weight;  // :763-765
  if (!nee_on || countEmitted == 1 || wasDelta == 1) emis = emis + (emission * lightFlag) * in.weight;                      // :766-768

  bool done = false;
  if (dot(wi, N) <= 0.0f && !transmits) done = true;                      // :770-773
  if (NdotV <= 0.0f && !transmits) done = true;                           // :776-779
  if (!gisvalid(bs.pdf) || !gisvalid(bs.f.x) || !gisvalid(bs.f.y) || !gisvalid(bs.f.z) || bs.pdf == 0.0f)
    done = true;                                                          // :781-784

  PathState nx = in;
  nx.seed = rng;                                                          // :759
  if (!done) {
    nx.directWeight = 1.0f;                                               // :785-790 (connect overwrites when NEE happens)
    nx.o = position + 0.0001f * faceforward(N, -wi, N);                   // :793
    nx.d = wi;                                                            // :794
    nx.weight = in.weight * ((bs.f * NoW) / bs.pdf);                      // :795
  }
  // back in raygen: Russian roulette reads the NEXT vertex's first variate (rgen:59,66-71)
  bool alive = true;
  if (depth > rc.rr_start_depth) {
    const float q = gclamp(gmax(gmax(nx.weight.x, nx.weight.y), nx.weight.z), 0.05f, 1.0f);
    uint32_t peek = nx.seed;
    if (rand_uniform(peek) > q) alive = false;
    nx.weight = nx.weight / q;
  }
  if (depth > rc.max_depth) alive = false;                                // rgen:73-75
  if (done) alive = false;                                                // rgen:77-78
  nx.flags = pack_flags(depth + 1u, bs.delta ? 1u : 0u, 0u) | (VER ? (in.flags & kStampMask) : 0u);  // rgen:80, rchit:792,796; VER: the versions ride along

  out.alive = alive;
  out.next = nx;
  out.has_shadow = want_shadow;
  out.emitted = emis;                                                     // (0 + nee) + emis with nee absent
  if (want_shadow) {
    out.shadow.o = position;                                              // :744 (un-offset origin, tmin 0.01)
    out.shadow.d = L;
    out.shadow.tmax = Ldist - 0.01f;                                      // :747
    out.shadow.nee = nee;
    out.shadow.emis = emis;
    out.shadow.dw_nee = power_heuristic(bs.pdf, lightPdf);                // :786
    out.shadow.sid = in.sid;
    out.shadow.next = -1;
  }
  GSP_PROF_END(PR_TAIL);
}

// dormant-feature extension: a path that leaves the scene (miss.rmiss:15-18 ends it) first picks up the environment
// map along its direction.  MIS: sampleLight never draws the environment, so the weight is the one an emitter met after a
// delta bounce gets (rayhit.rchit:766-768): the full path weight.
GSP_HD f3 miss_emitted(const SceneView& S, const PathState& in) { return sample_envmap(S.tex, in.d) * in.weight; }

// rayhit.rchit:750-754 + raygen.rgen:60-63 for a vertex that traced a shadow ray
GSP_HD void connect_vertex(float clampv, const ShadowRay& s, bool occluded, q4& result, bool& nee_done) {
  f3 e = splat(0.0f);
  if (!occluded) e = e + s.nee;
  e = e + s.emis;
  add_emitted(clampv, e, result);
  nee_done = !occluded;
}

// raygen.rgen:84-108: fold one finished sample into the running mean
GSP_HD void resolve_sample(uint32_t timestamp, q4 sample, q4& accum) {
  f3 c = mk3(sample.x, sample.y, sample.z);
  if (timestamp > 0) {
    const float a = 1.0f / (float)(timestamp + 1u);
    const f3 prev = mk3(accum.x, accum.y, accum.z);
    c = prev * (1.0f - a) + c * a;  // mix(prev, c, a)
  }
  if (!(gisnan(c.x) || gisnan(c.y) || gisnan(c.z))) {
    accum.x = c.x;
    accum.y = c.y;
    accum.z = c.z;
    accum.w = 1.0f;
  }
}

}  // namespace gsp
