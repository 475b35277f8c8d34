// tests/emu/pt_emu.cpp -- TEST HARNESS, NOT A PRODUCT PATH.
//
// Compiles the product's per-path stage headers (gpuspectral_amd/csrc/pt_*.h,
// the GSP_HD functions the HIP kernels call) for the host with g++ and drives
// them path by path, so that `pytest -m "not gpu"` can compare the exact
// shading / traversal / accumulate code of the kernels with the oracle on a
// machine without a GPU.  It is built into tests/emu/libpt_emu.so by the tests
// and is never imported by the gpuspectral_amd package, bench.py's timed path
// or anything shipped: the product renders on the GPU only.
//
// The BVH here is a plain median-split wide tree (kWide children per node, contiguous children) written with the
// product's own node encoder (encode_node_w4, pt_trace.h) and walked with the product's own per-ray
// traversal (trace_ray: the node step node_step + intersect_tri that k_trace and k_finish run, with the product's
// step table); closest-hit results do not depend on the BVH topology (pt_trace.h).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <vector>

#include "../../gpuspectral_amd/csrc/pt_hostmath.h"
#include "../../gpuspectral_amd/csrc/pt_stages.h"

using namespace gsp;

namespace {

q4 mkq(float x, float y, float z, float w) {
  q4 r;
  r.x = x;
  r.y = y;
  r.z = z;
  r.w = w;
  return r;
}

struct Emu {
  gsp_scene_desc sc;
  std::vector<gsp_instance> instances;
  std::vector<float> positions, normals, inv_t;
  // A resident BSDF table = one derived quad per record, in reverse order, in FRONT of the records (pt_shading.h derived_of):
  // n quads, then the n records; data() points at the first record.
  template <class R>
  struct Table {
    std::vector<q4s> store;
    size_t n = 0;
    void assign(const R* src, size_t count, uint32_t type) {
      n = count;
      store.assign(n + (n * sizeof(R) + 15) / 16 + 1, q4s{});
      if (n) std::memcpy((void*)data(), src, n * sizeof(R));
      for (size_t i = 0; i < n; ++i) store[n - 1 - i] = bake_bsdf(type, data() + i);  // as k_bake_tables (records as uploaded)
    }
    R* data() { return (R*)(store.data() + n); }
    R* begin() { return data(); }
    R* end() { return data() + n; }
  };
  Table<gsp_diffuse_bsdf> b0;
  Table<gsp_smooth_dielectric_bsdf> b1;
  Table<gsp_smooth_conductor_bsdf> b2;
  Table<gsp_smooth_plastic_bsdf> b3;
  Table<gsp_rough_conductor_bsdf> b4;
  Table<gsp_smooth_floor_bsdf> b5;
  Table<gsp_rough_floor_bsdf> b6;
  Table<gsp_rough_plastic_bsdf> b7;
  std::vector<gsp_triangle_light> lights;
  std::vector<q4> nodes, isect, shade;
  std::vector<uint32_t> slot_to_global;
  // dormant-feature extension (include/gpuspectral_pt.h)
  std::vector<float> uvs, tri_uv, decode, env_texels;
  std::vector<gsp_texture> textures;
  std::vector<uint32_t> texels;
  bool textured = false;
  std::vector<float> lo, hi;  // per slot padded boxes
  SceneView view;

  void bake() {
    std::vector<q4> gi, gs;
    std::vector<float> glo, ghi, guv;
    uint32_t g = 0;
    for (uint32_t a = 0; a < sc.num_instances; ++a) {
      const gsp_instance& I = instances[a];
      const float* T = &inv_t[16ull * a];
      for (uint32_t k = 0; k + 3 <= I.vertex_count; k += 3, ++g) {
        const float* P = &positions[3ull * (I.first_vertex + k)];
        const float* Nn = &normals[3ull * (I.first_vertex + k)];
        f3 p0 = xform_point(I.transform, mk3(P[0], P[1], P[2]));
        f3 p1 = xform_point(I.transform, mk3(P[3], P[4], P[5]));
        f3 p2 = xform_point(I.transform, mk3(P[6], P[7], P[8]));
        f3 n0 = xform_dir(T, mk3(Nn[0], Nn[1], Nn[2]));
        f3 n1 = xform_dir(T, mk3(Nn[3], Nn[4], Nn[5]));
        f3 n2 = xform_dir(T, mk3(Nn[6], Nn[7], Nn[8]));
        f3 e1 = p1 - p0, e2 = p2 - p0;
        f3 N = normalize(cross(e1, e2));
        gi.push_back(mkq(p0.x, p0.y, p0.z, u2f((g << 3) | ((I.bsdf >> 16) & 7u))));  // as k_bake (pt_bvh.hip)
        gi.push_back(mkq(p1.x, p1.y, p1.z, 0));
        gi.push_back(mkq(p2.x, p2.y, p2.z, 0));
        gs.push_back(mkq(N.x, N.y, N.z, u2f(pack_material(I.bsdf, I.twofaced))));
        gs.push_back(mkq(n0.x, n0.y, n0.z, I.emission[0]));
        gs.push_back(mkq(n1.x, n1.y, n1.z, I.emission[1]));
        gs.push_back(mkq(n2.x, n2.y, n2.z, I.emission[2]));
        if (!uvs.empty()) {
          const float* U = &uvs[2ull * (I.first_vertex + k)];
          guv.insert(guv.end(), {U[0], U[1], U[2], U[3], U[4], U[5], 0.0f, 0.0f});
        }
        const float px[3] = {p0.x, p0.y, p0.z}, qx[3] = {p1.x, p1.y, p1.z}, rx[3] = {p2.x, p2.y, p2.z};
        float l3[3], h3[3];
        for (int c = 0; c < 3; ++c) {
          l3[c] = std::min(px[c], std::min(qx[c], rx[c]));
          h3[c] = std::max(px[c], std::max(qx[c], rx[c]));
        }
        const float diag = std::max(h3[0] - l3[0], std::max(h3[1] - l3[1], h3[2] - l3[2]));
        const f3 e3 = p2 - p1, cr = cross(e1, e2);  // sliver factor, as k_bake (pt_bvh.hip)
        const float l2 = std::max(std::max(dot(e1, e1), dot(e2, e2)), dot(e3, e3));
        const float aspect = l2 / std::max(gsqrt(dot(cr, cr)), 1e-30f);
        const float sliver = aspect < 1.0e6f ? std::min(std::max(aspect * (1.0f / 32.0f), 1.0f), 1024.0f) : 1.0f;
        for (int c = 0; c < 3; ++c) {
          float pad = 1e-5f * std::max(std::max(std::fabs(l3[c]), std::fabs(h3[c])), std::max(diag, 1e-3f)) * sliver;
          glo.push_back(l3[c] - pad);
          ghi.push_back(h3[c] + pad);
        }
      }
    }
    const uint32_t n = g;
    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; ++i) order[i] = i;
    // median split into a kWide-wide tree with one triangle per leaf child; the nodes are numbered breadth-first so
    // that the inner children of a node are consecutive nodes and its leaf children consecutive triangle slots
    // (pt_trace.h), and written by the product's encoder
    nodes.clear();
    std::vector<uint32_t> slots;  // final slot order
    auto split = [&](uint32_t first, uint32_t count) -> uint32_t {  // returns the size of the left part
      float cl[3] = {1e30f, 1e30f, 1e30f}, ch[3] = {-1e30f, -1e30f, -1e30f};
      for (uint32_t i = first; i < first + count; ++i)
        for (int c = 0; c < 3; ++c) {
          float m = 0.5f * (glo[3ull * order[i] + c] + ghi[3ull * order[i] + c]);
          cl[c] = std::min(cl[c], m);
          ch[c] = std::max(ch[c], m);
        }
      int ax = 0;
      if (ch[1] - cl[1] > ch[ax] - cl[ax]) ax = 1;
      if (ch[2] - cl[2] > ch[ax] - cl[ax]) ax = 2;
      uint32_t mid = first + count / 2;
      std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count,
                       [&](uint32_t a, uint32_t b) {
                         return glo[3ull * a + ax] + ghi[3ull * a + ax] < glo[3ull * b + ax] + ghi[3ull * b + ax];
                       });
      return mid - first;
    };
    auto range_box = [&](uint32_t first, uint32_t count, WideChild& w) {
      float lo3[3] = {3e38f, 3e38f, 3e38f}, hi3[3] = {-3e38f, -3e38f, -3e38f};
      for (uint32_t i = first; i < first + count; ++i)
        for (int c = 0; c < 3; ++c) {
          lo3[c] = std::min(lo3[c], glo[3ull * order[i] + c]);
          hi3[c] = std::max(hi3[c], ghi[3ull * order[i] + c]);
        }
      w.lo = mkq(lo3[0], lo3[1], lo3[2], 0);
      w.hi = mkq(hi3[0], hi3[1], hi3[2], 0);
    };
    struct Range {
      uint32_t first, count;
    };
    std::vector<Range> queue;  // node i of the output = queue[i]
    queue.push_back(Range{0, n});
    for (size_t qi = 0; qi < queue.size(); ++qi) {
      const Range me = queue[qi];
      // up to kWide ranges: split the largest range until the node is full or only single triangles are left
      std::vector<Range> ch;
      ch.push_back(me);
      if (me.count >= 2) {
        const uint32_t l = split(me.first, me.count);
        ch[0] = Range{me.first, l};
        ch.push_back(Range{me.first + l, me.count - l});
      }
      while ((int)ch.size() < kWide) {
        int best = -1;
        for (int k = 0; k < (int)ch.size(); ++k)
          if (ch[k].count >= 2 && (best < 0 || ch[k].count > ch[best].count)) best = k;
        if (best < 0) break;
        const Range r = ch[best];
        const uint32_t l = split(r.first, r.count);
        ch[best] = Range{r.first, l};
        ch.push_back(Range{r.first + l, r.count - l});
      }
      const uint32_t child_base = (uint32_t)queue.size(), tri_base = (uint32_t)slots.size();
      nodes.resize(nodes.size() + kNodeQuads);
      q4* out = &nodes[(size_t)kNodeQuads * qi];
      if (n == 0) ch.clear();
      WideChild wc[4];
      int ni = 0, nl = 0;
      for (const Range& r : ch)
        if (r.count >= 2) {
          range_box(r.first, r.count, wc[ni++]);
          queue.push_back(r);
        }
      for (const Range& r : ch)
        if (r.count == 1) {
          range_box(r.first, r.count, wc[ni + nl++]);
          slots.push_back(order[r.first]);
        }
      encode_node_w4(out, wc, ni, nl, child_base, tri_base);
    }
    isect.assign(3ull * (n + kWide), mkq(0, 0, 0, 0));  // slots n..: all-zero triangles (det == 0: never hit), as build_bvh appends
    shade.assign(4ull * (n + kWide), mkq(0, 0, 0, 0));
    slot_to_global.assign(n + kWide, 0);
    for (uint32_t s = 0; s < n; ++s) {
      uint32_t gg = slots[s];
      slot_to_global[s] = gg;
      for (int k = 0; k < 3; ++k) isect[3ull * s + k] = gi[3ull * gg + k];
      for (int k = 0; k < 4; ++k) shade[4ull * s + k] = gs[4ull * gg + k];
    }
    if (!guv.empty()) {
      tri_uv.assign(8ull * (n + kWide), 0.0f);
      for (uint32_t s = 0; s < n; ++s)
        for (int k = 0; k < 8; ++k) tri_uv[8ull * s + k] = guv[8ull * slots[s] + k];
    }
    view.nodes = nodes.data();
    view.tri_isect = isect.data();
    view.tri_shade = shade.data();
    view.bsdf.diffuse = b0.data();
    view.bsdf.smooth_dielectric = b1.data();
    view.bsdf.smooth_conductor = b2.data();
    view.bsdf.smooth_plastic = b3.data();
    view.bsdf.rough_conductor = b4.data();
    view.bsdf.smooth_floor = b5.data();
    view.bsdf.rough_floor = b6.data();
    view.bsdf.rough_plastic = b7.data();
    view.lights = lights.data();
    view.num_lights = sc.num_lights;
    view.inv_num_lights = sc.num_lights ? 1.0f / (float)sc.num_lights : 0.0f;
    view.static_slots = 0;
    if (textured) {  // what gsp_context::view() fills
      view.tex.tri_uv = textures.empty() ? nullptr : tri_uv.data();
      view.tex.textures = textures.data();
      view.tex.texels = texels.data();
      view.tex.decode = decode.data();
      view.tex.num_textures = (uint32_t)textures.size();
      view.tex.env_texels = env_texels.empty() ? nullptr : env_texels.data();
      view.tex.env_width = sc.envmap.width;
      view.tex.env_height = sc.envmap.height;
      for (int k = 0; k < 16; ++k) view.tex.env_to_local[k] = sc.envmap.to_local[k];
    }
  }
};

struct HostStack {  // the per-lane stack of trace_ray (k_finish keeps it in LDS)
  uint32_t w[4096];
  uint32_t top = 0;
  void push(uint32_t v) { w[top++] = v; }
  uint32_t pop() { return w[--top]; }
};
const StepTableRef kTab{&kStepTable};
template <bool ANY>
bool trace1(const SceneView& S, f3 o, f3 d, float tmin, float tmax, HitRec& h, uint32_t& aux) {
  HostStack stk;
  return trace_ray<ANY>(S.nodes, S.tri_isect, o, d, tmin, tmax, h, aux, stk, kTab);
}

template <class T>
void copyv(std::vector<T>& d, const T* s, size_t n) {
  d.assign(s, s + (s ? n : 0));
}

}  // namespace

extern "C" {

// transformInvT is computed with the product's own host routine (pt_hostmath.h),
// exactly as gsp_upload_scene does.
void* emu_create(const gsp_scene_desc* sc) {
  Emu* e = new Emu();
  e->sc = *sc;
  copyv(e->instances, sc->instances, sc->num_instances);
  copyv(e->positions, sc->positions, 3 * (size_t)sc->num_vertices);
  copyv(e->normals, sc->normals, 3 * (size_t)sc->num_vertices);
  e->b0.assign(sc->diffuse_bsdfs, sc->num_bsdfs[0], 0);
  e->b1.assign(sc->smooth_dielectric_bsdfs, sc->num_bsdfs[1], 1);
  e->b2.assign(sc->smooth_conductor_bsdfs, sc->num_bsdfs[2], 2);
  e->b3.assign(sc->smooth_plastic_bsdfs, sc->num_bsdfs[3], 3);
  e->b4.assign(sc->rough_conductor_bsdfs, sc->num_bsdfs[4], 4);
  e->b5.assign(sc->smooth_floor_bsdfs, sc->num_bsdfs[5], 5);
  e->b6.assign(sc->rough_floor_bsdfs, sc->num_bsdfs[6], 6);
  e->b7.assign(sc->rough_plastic_bsdfs, sc->num_bsdfs[7], 7);
  copyv(e->lights, sc->lights, sc->num_lights);
  for (gsp_triangle_light& L : e->lights) bake_light(L);  // the resident records, as k_bake_tables leaves them
  for (gsp_diffuse_bsdf& b : e->b0) bake_diffuse(b);
  e->inv_t.resize(16ull * sc->num_instances);
  for (uint32_t i = 0; i < sc->num_instances; ++i) {
    float tr[16];
    transpose4(e->instances[i].transform, tr);
    inverse4(tr, &e->inv_t[16ull * i]);
  }
  if (sc->num_textures && sc->textures && sc->texels && sc->uvs) {  // as gsp_upload_scene
    copyv(e->uvs, sc->uvs, 2 * (size_t)sc->num_vertices);
    copyv(e->textures, sc->textures, sc->num_textures);
    copyv(e->texels, sc->texels, (size_t)sc->num_texels);
    e->decode.resize(256);
    for (int b = 0; b < 256; ++b) e->decode[b] = sc->texel_decode ? sc->texel_decode[b] : (float)b / 255.0f;
    e->textured = true;
  }
  if (sc->envmap.texels) {
    copyv(e->env_texels, sc->envmap.texels, 4 * (size_t)sc->envmap.width * sc->envmap.height);
    e->textured = true;
  }
  e->bake();
  return e;
}
void emu_destroy(void* h) { delete (Emu*)h; }

// same contract as oracle_render / gsp_render + gsp_download_compact
int emu_render(void* h, uint32_t width, uint32_t height, const uint32_t* pixel_ids, uint64_t num_pixels,
               const gsp_render_params* rp, float* accum) {
  Emu* e = (Emu*)h;
  RenderConsts rc;
  rc.width = width;
  rc.height = height;
  rc.max_depth = rp->max_depth;
  rc.rr_start_depth = rp->rr_start_depth;
  rc.clamp = rp->clamp;
  rc.nee = rp->disable_nee != 0 ? 0u : 1u;
  rc.zplane = (std::max((float)width, (float)height) / 2.0f) / tanf(e->sc.camera.fov / 2.0f);
  for (int i = 0; i < 16; ++i) rc.cam_to_world[i] = e->sc.camera.to_world[i];
  for (int i = 0; i < 3; ++i) rc.cam_origin[i] = e->sc.camera.to_world[12 + i];
  const uint64_t npix = pixel_ids ? num_pixels : (uint64_t)width * height;
  const SceneView& S = e->view;
  for (uint64_t lp = 0; lp < npix; ++lp) {
    const uint32_t gid = pixel_ids ? pixel_ids[lp] : (uint32_t)lp;
    q4 acc = mkq(accum[4 * lp], accum[4 * lp + 1], accum[4 * lp + 2], accum[4 * lp + 3]);
    for (uint32_t s = 0; s < rp->spp; ++s) {
      const uint32_t ts = rp->first_timestamp + s;
      PathState p;
      generate_path(rc, gid, ts, 0, p);
      q4 result = mkq(0, 0, 0, 0);
      bool alive = true;
      while (alive) {
        HitRec hit;
        uint32_t aux;
        trace1<false>(S, p.o, p.d, 0.0f, 1e10f, hit, aux);
        if (hit.slot < 0 || e->sc.num_vertices == 0) {  // miss (k_shade / k_finish: the <TEX> branch)
          if (e->textured && S.tex.env_texels != nullptr) add_emitted(rc.clamp, miss_emitted(S, p), result);
          break;
        }
        ShadeOut out;
        if (e->textured) shade_vertex<true>(S, rc, p, hit, out);
        else shade_vertex<false>(S, rc, p, hit, out);
        if (out.has_shadow) {
          HitRec sh;
          uint32_t aux2;
          bool occ = trace1<true>(S, out.shadow.o, out.shadow.d, 0.01f, out.shadow.tmax, sh, aux2);
          bool nee_done;
          connect_vertex(rc.clamp, out.shadow, occ, result, nee_done);
          if (nee_done && out.alive) out.next.directWeight = out.shadow.dw_nee;
        } else {
          add_emitted(rc.clamp, out.emitted, result);
        }
        alive = out.alive;
        p = out.next;
      }
      resolve_sample(ts, result, acc);
    }
    accum[4 * lp] = acc.x;
    accum[4 * lp + 1] = acc.y;
    accum[4 * lp + 2] = acc.z;
    accum[4 * lp + 3] = acc.w;
  }
  return 0;
}

int emu_trace(void* h, const float* rays, uint64_t n, int any_hit, void* hits_out) {
  Emu* e = (Emu*)h;
  struct HR {
    float t, u, v;
    int32_t prim;
  };
  HR* out = (HR*)hits_out;
  const SceneView& S = e->view;
  const bool empty = e->sc.num_vertices == 0;
  for (uint64_t i = 0; i < n; ++i) {
    const float* r = rays + 8 * i;
    HitRec hh;
    uint32_t aux;
    bool hit;
    if (any_hit)
      hit = trace1<true>(S, mk3(r[0], r[1], r[2]), mk3(r[4], r[5], r[6]), r[3], r[7], hh, aux);
    else
      hit = trace1<false>(S, mk3(r[0], r[1], r[2]), mk3(r[4], r[5], r[6]), r[3], r[7], hh, aux);
    if (empty) hit = false;
    if (any_hit) out[i] = HR{0, 0, 0, hit ? 0 : -1};
    else out[i] = hit ? HR{hh.t, hh.u, hh.v, (int32_t)e->slot_to_global[hh.slot]} : HR{0, 0, 0, -1};
  }
  return 0;
}

void emu_bsdf_sample(void* h, uint32_t handle, const float* wo, uint32_t seed, float* out9) {
  Emu* e = (Emu*)h;
  uint32_t rng = seed;
  f3 wi;
  BsdfResult r;
  BsdfCarry cy;
  bsdf_sample(e->view.bsdf, handle, rng, mk3(wo[0], wo[1], wo[2]), wi, r, cy);
  out9[0] = wi.x;
  out9[1] = wi.y;
  out9[2] = wi.z;
  out9[3] = r.f.x;
  out9[4] = r.f.y;
  out9[5] = r.f.z;
  out9[6] = r.pdf;
  out9[7] = r.delta ? 1.0f : 0.0f;
  out9[8] = u2f(rng);
}
void emu_bsdf_eval(void* h, uint32_t handle, const float* wo, const float* wi, float* out5) {
  Emu* e = (Emu*)h;
  BsdfResult r;
  // evalBSDF of a vertex reads what sampleBSDF of the SAME vertex (same record, same wo) left behind (BsdfCarry, pt_shading.h):
  // run the sampler first, with any random stream
  BsdfCarry cy;
  {
    uint32_t rng = 1u;
    f3 wi0;
    BsdfResult r0;
    bsdf_sample(e->view.bsdf, handle, rng, mk3(wo[0], wo[1], wo[2]), wi0, r0, cy);
  }
  bsdf_eval(e->view.bsdf, handle, mk3(wo[0], wo[1], wo[2]), mk3(wi[0], wi[1], wi[2]), r, cy);
  out5[0] = r.f.x;
  out5[1] = r.f.y;
  out5[2] = r.f.z;
  out5[3] = r.pdf;
  out5[4] = r.delta ? 1.0f : 0.0f;
}
void emu_sample_light(void* h, const float* pos, uint32_t seed, float* out8) {
  Emu* e = (Emu*)h;
  uint32_t rng = seed;
  LightSample r = sample_light(e->view.lights, e->view.num_lights, e->view.inv_num_lights, rng, mk3(pos[0], pos[1], pos[2]));
  out8[0] = r.position.x;
  out8[1] = r.position.y;
  out8[2] = r.position.z;
  out8[3] = r.emission.x;
  out8[4] = r.emission.y;
  out8[5] = r.emission.z;
  out8[6] = r.pdf;
  out8[7] = u2f(rng);
}
void emu_det_math(const float* x, uint64_t n, float* s, float* c, float* lg, float* ex) {
  for (uint64_t i = 0; i < n; ++i) {
    det_sincosf(x[i], s[i], c[i]);
    lg[i] = det_logf(x[i]);
    ex[i] = det_expf(x[i]);
  }
}
// The two statements of the watertight triangle test (pt_trace.h): intersect_tri as the oracle states it (per-ray axis
// permutation with the kx / ky exchange, early exits) and intersect_tri_rot as k_trace's leaf step runs it (axes selected by two
// lane masks, no exchange, straight-line).  rays: n x {o.xyz, d.xyz, tmin, tmax}, tris: n x 9; out: n x {hit, t, u, v} twice.
void emu_tri_tests(const float* rays, const float* tris, uint64_t n, float* out_ref, float* out_rot) {
  for (uint64_t i = 0; i < n; ++i) {
    const float* r = rays + 8 * i;
    const float* p = tris + 9 * i;
    const f3 o = mk3(r[0], r[1], r[2]), d = mk3(r[3], r[4], r[5]);
    const f3 v0 = mk3(p[0], p[1], p[2]), v1 = mk3(p[3], p[4], p[5]), v2 = mk3(p[6], p[7], p[8]);
    float t = 0.0f, u = 0.0f, v = 0.0f;
    const RayShear rs = make_shear(d);
    const bool h0 = intersect_tri(v0, v1, v2, o, rs, r[6], r[7], t, u, v);
    out_ref[4 * i + 0] = h0 ? 1.0f : 0.0f;
    out_ref[4 * i + 1] = h0 ? t : 0.0f;
    out_ref[4 * i + 2] = h0 ? u : 0.0f;
    out_ref[4 * i + 3] = h0 ? v : 0.0f;
    const RayShearRot rr = make_shear_rot(d);
    const bool h1 = intersect_tri_rot(v0, v1, v2, o, rr, r[6], r[7], t, u, v);
    out_rot[4 * i + 0] = h1 ? 1.0f : 0.0f;
    out_rot[4 * i + 1] = h1 ? t : 0.0f;
    out_rot[4 * i + 2] = h1 ? u : 0.0f;
    out_rot[4 * i + 3] = h1 ? v : 0.0f;
  }
}
// The two statements of the 4-wide node encoder (pt_trace.h): encode_node_w4_ref (loops, arrays: the definition) and
// encode_node_w4 (what the build and the refit run: positions in registers, the insertion sort as compare-exchanges, closed-form
// order codes).  boxes: n x 4 x {lo.xyz, hi.xyz}; counts: n x {ni, nl}; out: n x 16 words each.
void emu_encode_nodes(const float* boxes, const int32_t* counts, uint64_t n, uint32_t* out_ref, uint32_t* out_fast) {
  for (uint64_t i = 0; i < n; ++i) {
    WideChild wc[4];
    for (int k = 0; k < 4; ++k) {
      const float* b = boxes + 24 * i + 6 * k;
      wc[k].lo = make_q4(b[0], b[1], b[2], 0.0f);
      wc[k].hi = make_q4(b[3], b[4], b[5], 0.0f);
    }
    q4 a[4], c[4];
    const int ni = counts[2 * i], nl = counts[2 * i + 1];
    encode_node_w4_ref(a, wc, ni, nl, 1000u + (uint32_t)i, 77u + (uint32_t)i);
    encode_node_w4(c, wc, ni, nl, 1000u + (uint32_t)i, 77u + (uint32_t)i);
    std::memcpy(out_ref + 16 * i, a, 64);
    std::memcpy(out_fast + 16 * i, c, 64);
  }
}
void emu_transform_inv_t(const float* m, float* out) {
  float tr[16];
  transpose4(m, tr);
  inverse4(tr, out);
}
// the product's frame / RNG / helper functions one by one (tests/test_glsl_vectors.py)
void emu_onb(const float* nrm, const float* v, float* out15) {
  Frame f = make_frame(mk3(nrm[0], nrm[1], nrm[2]));
  f3 a = to_local(f, mk3(v[0], v[1], v[2])), b = to_world(f, mk3(v[0], v[1], v[2]));
  const f3 r[5] = {f.t, f.b, f.n, a, b};
  for (int k = 0; k < 5; ++k) {
    out15[3 * k] = r[k].x;
    out15[3 * k + 1] = r[k].y;
    out15[3 * k + 2] = r[k].z;
  }
}
// out6 = {tea(a, b), pcgHash(a), then from state a: randPcg, randPcg, randUniform bits, state afterwards}
void emu_rng(uint32_t a, uint32_t b, uint32_t* out6) {
  out6[0] = tea(a, b);
  out6[1] = pcg_hash(a);
  uint32_t st = a;
  out6[2] = pcg_next(st);
  out6[3] = pcg_next(st);
  out6[4] = f2u(rand_uniform(st));
  out6[5] = st;
}
float emu_power_heuristic(float f, float g) { return power_heuristic(f, g); }
float emu_cosine_pdf(float z) { return cosine_pdf(mk3(0.0f, 0.0f, z)); }
int emu_is_transmission(uint32_t handle) { return bsdf_transmits(handle) ? 1 : 0; }
uint32_t emu_seed(uint32_t width, uint32_t px, uint32_t py, uint32_t timestamp) {
  return pcg_hash(tea(width * py + px, timestamp));
}
}
