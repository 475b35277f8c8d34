// scripts/experiments/wide_bvh_probe.cpp -- CPU experiment, not product code.
//
// Question: how many node visits / box tests / triangle tests per ray does an 8-wide BVH
// traversed in octant order need, against the 4-wide tree with exact distance ordering the product uses?
// Builds a binned-SAH binary tree over a triangle soup (1 triangle per leaf, like the device builder), collapses it
// several ways and traverses a set of incoherent closest-hit rays with each variant, counting work.
//
//   g++ -O3 -march=native -fopenmp -o /tmp/wide_bvh_probe scripts/experiments/wide_bvh_probe.cpp
//   /tmp/wide_bvh_probe tris.bin [num_rays]
// tris.bin: N x 9 float32 (world-space vertices), written by scripts/experiments/wide_bvh_probe.py
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <random>
#include <vector>

struct V3 {
  float x, y, z;
};
static inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static inline V3 vmin(V3 a, V3 b) { return {std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z)}; }
static inline V3 vmax(V3 a, V3 b) { return {std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z)}; }
static inline float comp(V3 v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); }

struct Box {
  V3 lo{3e38f, 3e38f, 3e38f}, hi{-3e38f, -3e38f, -3e38f};
  void grow(V3 p) { lo = vmin(lo, p), hi = vmax(hi, p); }
  void grow(const Box& b) { lo = vmin(lo, b.lo), hi = vmax(hi, b.hi); }
  float area() const {
    V3 d = hi - lo;
    return d.x * d.y + d.y * d.z + d.z * d.x;
  }
  V3 centre() const { return (lo + hi) * 0.5f; }
};

struct Tri {
  V3 a, b, c;
};

// ---- binary tree -------------------------------------------------------------------------------------------------
struct BNode {
  Box box;
  int left = -1, right = -1;  // children (node index) ; leaf: left = -1, tri = right
  bool leaf() const { return left < 0; }
};
static std::vector<BNode> g_bin;
static int g_root = 0;
static std::vector<Tri> g_tris;
static std::vector<Box> g_tbox;

static int build_binary(std::vector<int>& idx, int first, int count) {
  int me = (int)g_bin.size();
  g_bin.emplace_back();
  Box b, cb;
  for (int i = first; i < first + count; ++i) {
    b.grow(g_tbox[idx[i]]);
    cb.grow(g_tbox[idx[i]].centre());
  }
  g_bin[me].box = b;
  if (count == 1) {
    g_bin[me].right = idx[first];
    return me;
  }
  // binned SAH, 16 bins, best of three axes
  constexpr int NB = 16;
  float best = 3e38f;
  int best_axis = -1, best_split = 0;
  for (int ax = 0; ax < 3; ++ax) {
    float lo = comp(cb.lo, ax), hi = comp(cb.hi, ax);
    if (!(hi > lo)) continue;
    Box bins[NB];
    int cnt[NB] = {0};
    float k = NB / (hi - lo);
    for (int i = first; i < first + count; ++i) {
      int bi = std::min(NB - 1, (int)((comp(g_tbox[idx[i]].centre(), ax) - lo) * k));
      bins[bi].grow(g_tbox[idx[i]]);
      ++cnt[bi];
    }
    float ra[NB];
    Box acc;
    int rc[NB];
    int c = 0;
    for (int i = NB - 1; i > 0; --i) {
      acc.grow(bins[i]);
      c += cnt[i];
      ra[i] = c ? acc.area() : 0.0f;
      rc[i] = c;
    }
    acc = Box();
    c = 0;
    for (int i = 0; i < NB - 1; ++i) {
      acc.grow(bins[i]);
      c += cnt[i];
      if (c == 0 || rc[i + 1] == 0) continue;
      float cost = acc.area() * c + ra[i + 1] * rc[i + 1];
      if (cost < best) best = cost, best_axis = ax, best_split = i;
    }
  }
  int mid;
  if (best_axis < 0) {
    mid = first + count / 2;
  } else {
    float lo = comp(cb.lo, best_axis), hi = comp(cb.hi, best_axis);
    float k = NB / (hi - lo);
    mid = (int)(std::partition(idx.begin() + first, idx.begin() + first + count,
                               [&](int t) {
                                 int bi = std::min(NB - 1, (int)((comp(g_tbox[t].centre(), best_axis) - lo) * k));
                                 return bi <= best_split;
                               }) -
                idx.begin());
    if (mid == first || mid == first + count) mid = first + count / 2;
  }
  int l = build_binary(idx, first, mid - first);
  int r = build_binary(idx, mid, first + count - mid);
  g_bin[me].left = l;
  g_bin[me].right = r;
  return me;
}

// PLOC (Meister & Bittner 2018) as the device builder does it (pt_bvh.hip): Morton order, mutual nearest neighbours by
// merged half-area inside a +-R window, repeated until one cluster is left.  Replaces g_bin; returns the root.
static uint64_t expand21(uint64_t v) {
  v &= 0x1fffffull;
  v = (v | (v << 32)) & 0x001f00000000ffffull;
  v = (v | (v << 16)) & 0x001f0000ff0000ffull;
  v = (v | (v << 8)) & 0x100f00f00f00f00full;
  v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}
static int build_ploc(int R) {
  const size_t n = g_tris.size();
  Box sb;
  for (size_t i = 0; i < n; ++i) sb.grow(g_tbox[i]);
  std::vector<std::pair<uint64_t, int>> keys(n);
  for (size_t i = 0; i < n; ++i) {
    V3 c = g_tbox[i].centre();
    uint64_t code = 0;
    const float cc[3] = {c.x, c.y, c.z}, lo[3] = {sb.lo.x, sb.lo.y, sb.lo.z}, hi[3] = {sb.hi.x, sb.hi.y, sb.hi.z};
    for (int k = 0; k < 3; ++k) {
      float t = hi[k] > lo[k] ? (cc[k] - lo[k]) / (hi[k] - lo[k]) : 0.0f;
      t = std::min(std::max(t, 0.0f), 1.0f);
      code |= expand21((uint64_t)std::min(t * 2097152.0f, 2097151.0f)) << (2 - k);
    }
    keys[i] = {code, (int)i};
  }
  std::sort(keys.begin(), keys.end());
  g_bin.clear();
  g_bin.reserve(2 * n);
  std::vector<int> cl(n);
  for (size_t i = 0; i < n; ++i) {
    BNode b;
    b.box = g_tbox[keys[i].second];
    b.right = keys[i].second;
    g_bin.push_back(b);
    cl[i] = (int)i;
  }
  std::vector<int> nn, next;
  while (cl.size() > 1) {
    const int m = (int)cl.size();
    nn.assign(m, 0);
    // PLOC_ADAPT="m1 f1 m2 f2": the window grows to f1 * R once m <= m1 clusters are left, f2 * R below m2 (the top of the tree)
    int Re = R;
    if (const char* ad = getenv("PLOC_ADAPT")) {
      int m1 = 0, f1 = 1, m2 = 0, f2 = 1;
      sscanf(ad, "%d %d %d %d", &m1, &f1, &m2, &f2);
      if (m <= m2) Re = R * f2; else if (m <= m1) Re = R * f1;
    }
#pragma omp parallel for schedule(dynamic, 1024)
    for (int i = 0; i < m; ++i) {
      float best = 3e38f;
      int bj = i > 0 ? i - 1 : i + 1;
      for (int j = std::max(0, i - Re); j <= std::min(m - 1, i + Re); ++j) {
        if (j == i) continue;
        Box u = g_bin[cl[i]].box;
        u.grow(g_bin[cl[j]].box);
        float a = u.area();
        if (a < best) best = a, bj = j;
      }
      nn[i] = bj;
    }
    next.clear();
    for (int i = 0; i < m; ++i) {
      const int j = nn[i];
      const bool mutual = nn[j] == i;
      if (mutual && i > j) continue;
      if (mutual) {
        BNode b;
        b.box = g_bin[cl[i]].box;
        b.box.grow(g_bin[cl[j]].box);
        b.left = cl[i];
        b.right = cl[j];
        g_bin.push_back(b);
        next.push_back((int)g_bin.size() - 1);
      } else next.push_back(cl[i]);
    }
    cl.swap(next);
  }
  return cl[0];
}
// SAH cost of the binary tree rooted at r (node visit 1, triangle test 1), relative to the root area
static double sah_cost(int r) {
  double c = 0;
  const double ra = g_bin[r].box.area();
  std::vector<int> st = {r};
  while (!st.empty()) {
    int b = st.back();
    st.pop_back();
    c += g_bin[b].box.area() / ra;
    if (!g_bin[b].leaf()) st.push_back(g_bin[b].left), st.push_back(g_bin[b].right);
  }
  return c;
}

// ---- insertion-based optimisation (Bittner, Hapala, Havran 2013), REINSERT=passes --------------------------------------
// For every node N (largest first): take N and its parent out (the sibling moves up), find by branch and bound the
// node X for which inserting N as X's new sibling adds the least surface area (direct: area(N u X); induced: the growth
// of X's ancestors), and put it there.  What would a tree-quality pass behind the PLOC build be worth?
static std::vector<int> g_parent;
static void refit_up(int n) {
  while (n >= 0) {
    BNode& b = g_bin[n];
    Box nb = g_bin[b.left].box;
    nb.grow(g_bin[b.right].box);
    b.box = nb;
    n = g_parent[n];
  }
}
static void reinsertion_pass() {
  const int nn = (int)g_bin.size();
  g_parent.assign(nn, -1);
  for (int i = 0; i < nn; ++i)
    if (!g_bin[i].leaf()) g_parent[g_bin[i].left] = i, g_parent[g_bin[i].right] = i;
  std::vector<int> order;
  for (int i = 0; i < nn; ++i)
    if (i != g_root && g_parent[i] >= 0 && g_parent[i] != g_root) order.push_back(i);
  std::sort(order.begin(), order.end(), [](int a, int b) { return g_bin[a].box.area() > g_bin[b].box.area(); });
  float min_area = 0.0f;  // REINSERT_SEARCH_TOP=K2: candidate positions only among the K2 largest nodes (a top tree cut out of the whole)
  if (const char* st = getenv("REINSERT_SEARCH_TOP")) {
    const size_t k2 = std::min<size_t>(order.size() - 1, (size_t)atol(st));
    min_area = g_bin[order[k2]].box.area();
  }
  if (const char* tk = getenv("REINSERT_TOP")) order.resize(std::min<size_t>(order.size(), (size_t)atol(tk)));  // only the K largest nodes
  if (getenv("REINSERT_LEAVES_ONLY")) {  // only the leaves among them (the room-sized triangles)
    std::vector<int> lv;
    for (int i : order) if (g_bin[i].leaf()) lv.push_back(i);
    order.swap(lv);
  }
  size_t moved = 0;
  struct Cand {
    float induced;
    int node;
    bool operator<(const Cand& o) const { return induced > o.induced; }
  };
  std::vector<Cand> pq;
  for (int N : order) {
    const int P = g_parent[N];
    if (P < 0 || P == g_root) continue;
    const int G = g_parent[P];
    const int S = g_bin[P].left == N ? g_bin[P].right : g_bin[P].left;
    // is N an ancestor position we must not insert under? (N's own subtree is excluded from the search)
    // remove: S takes P's place
    if (g_bin[G].left == P) g_bin[G].left = S; else g_bin[G].right = S;
    g_parent[S] = G;
    refit_up(G);
    const Box nb = g_bin[N].box;
    const float na = nb.area();
    float best = 3e38f;
    int bx = -1;
    pq.clear();
    pq.push_back({0.0f, g_root});
    while (!pq.empty()) {
      std::pop_heap(pq.begin(), pq.end());
      const Cand c = pq.back();
      pq.pop_back();
      if (c.induced + na >= best) break;  // nothing below can beat the best (direct cost >= area(N))
      const BNode& x = g_bin[c.node];
      Box u = x.box;
      u.grow(nb);
      const float direct = u.area();
      const float total = c.induced + direct;
      if (total < best) best = total, bx = c.node;
      const float ind = total - x.box.area();  // induced cost for the children of x
      if (!x.leaf() && ind + na < best && x.box.area() >= min_area) {
        pq.push_back({ind, x.left});
        std::push_heap(pq.begin(), pq.end());
        pq.push_back({ind, x.right});
        std::push_heap(pq.begin(), pq.end());
      }
    }
    // insert N as the sibling of bx under the recycled parent P
    const int X = bx, XP = g_parent[X];
    if (X != S || XP != G) ++moved;
    g_bin[P].left = X;
    g_bin[P].right = N;
    g_parent[X] = P;
    g_parent[N] = P;
    g_parent[P] = XP;
    if (XP >= 0) {
      if (g_bin[XP].left == X) g_bin[XP].left = P; else g_bin[XP].right = P;
    } else g_root = P;
    refit_up(P);
  }
  printf("reinsertion pass: %zu of %zu nodes moved, SAH cost %.2f\n", moved, order.size(), sah_cost(g_root));
}

// ---- parallel reinsertion (Meister & Bittner 2018), PREINSERT="rounds": every node searches on the FROZEN tree (no removal:
// the search climbs from the node's parent and accounts for the shrinking of the path), candidates lock the nodes whose
// links they would change (largest gain wins), the winners are applied, all boxes are refitted.  What a device pass would do.
struct PMove {
  int x;
  float gain;
};
static inline int sibling_of(int n) {
  const int p = g_parent[n];
  return g_bin[p].left == n ? g_bin[p].right : g_bin[p].left;
}
static long g_budget = 0;
static std::vector<long> g_visit_hist(8, 0);
static PMove find_best(int in) {
  PMove r{-1, 0.0f};
  long visits = 0;
  const int P = g_parent[in];
  const Box inb = g_bin[in].box;
  const float a_parent = g_bin[P].box.area();
  float d_bound = 0.0f;  // area freed on the path from P up to (not including) the pivot
  int pivot = P, sib = sibling_of(in);
  Box pivot_box;  // what the pivot's box becomes once `in` is gone
  int stack_n[128];
  float stack_d[128];
  while (true) {
    int sp = 0;
    stack_n[sp] = sib, stack_d[sp] = d_bound, ++sp;
    while (sp > 0) {
      if (g_budget > 0 && ++visits > g_budget) return r;  // PREINSERT_BUDGET: node visits a search may spend
      --sp;
      const int out = stack_n[sp];
      const float d_par = stack_d[sp];
      if (d_par + a_parent <= r.gain) continue;  // even a free insertion below cannot beat the best
      Box m = g_bin[out].box;
      m.grow(inb);
      const float d_direct = a_parent - m.area();  // P is recycled: its old box goes, the merged one comes
      if (d_par + d_direct > r.gain && out != sib + 0 * in) r.gain = d_par + d_direct, r.x = out;
      if (!g_bin[out].leaf()) {
        const float d = d_par + g_bin[out].box.area() - m.area();  // `out` grows to the merged box if we go below it
        if (d + a_parent > r.gain && sp + 2 <= 128) {
          stack_n[sp] = g_bin[out].left, stack_d[sp] = d, ++sp;
          stack_n[sp] = g_bin[out].right, stack_d[sp] = d, ++sp;
        }
      }
    }
    // climb: the pivot loses `in`
    pivot_box.grow(g_bin[sib].box);
    if (pivot != P) d_bound += g_bin[pivot].box.area() - pivot_box.area();
    if (pivot == g_root) break;
    sib = sibling_of(pivot);
    pivot = g_parent[pivot];
  }
  return r;
}
static void parallel_reinsertion(int rounds) {
  const int nn = (int)g_bin.size();
  std::vector<PMove> mv(nn);
  std::vector<unsigned long long> lock(nn);
  for (int round = 0; round < rounds; ++round) {
    g_parent.assign(nn, -1);
    for (int i = 0; i < nn; ++i)
      if (!g_bin[i].leaf()) g_parent[g_bin[i].left] = i, g_parent[g_bin[i].right] = i;
    const int k0 = getenv("PREINSERT_K0") ? atoi(getenv("PREINSERT_K0")) : 9;
    const int k = std::max(1, k0 - round);  // sparse selection, denser every round
#pragma omp parallel for schedule(dynamic, 256)
    for (int i = 0; i < nn; ++i) {
      mv[i] = PMove{-1, 0.0f};
      if (i == g_root || g_parent[i] == g_root || (i % k) != (round % k)) continue;
      mv[i] = find_best(i);
      if (mv[i].x == sibling_of(i)) mv[i].x = -1;  // where it is
    }
    std::fill(lock.begin(), lock.end(), 0ull);
    auto key = [&](int i) { unsigned u; memcpy(&u, &mv[i].gain, 4); return ((unsigned long long)u << 32) | (unsigned)i; };
    auto touched = [&](int i, int* t) {
      const int P = g_parent[i], S = sibling_of(i), G = g_parent[P], X = mv[i].x, XP = g_parent[X];
      t[0] = i, t[1] = P, t[2] = S, t[3] = G, t[4] = X, t[5] = XP;
    };
    for (int i = 0; i < nn; ++i) {
      if (mv[i].x < 0 || mv[i].gain <= 0.0f) continue;
      int t[6];
      touched(i, t);
      for (int q = 0; q < 6; ++q)
        if (t[q] >= 0) lock[t[q]] = std::max(lock[t[q]], key(i));
    }
    size_t applied = 0, cands = 0;
    for (int i = 0; i < nn; ++i) {
      if (mv[i].x < 0 || mv[i].gain <= 0.0f) continue;
      ++cands;
      int t[6];
      touched(i, t);
      bool ok = true;
      for (int q = 0; q < 6; ++q)
        if (t[q] >= 0 && lock[t[q]] != key(i)) ok = false;
      if (!ok) continue;
      const int N = i, P = t[1], S = t[2], G = t[3], X = t[4], XP = t[5];
      // X inside N's subtree cannot happen (the search never enters it); X == P is excluded by the locks (P is N's parent: key equal) -- skip it
      if (X == P) continue;
      if (g_bin[G].left == P) g_bin[G].left = S; else g_bin[G].right = S;
      g_parent[S] = G;
      const int XP2 = (XP == P) ? G : XP;  // X was the sibling's child?  (X == S is filtered above; XP == P only if X == S)
      if (g_bin[XP2].left == X) g_bin[XP2].left = P; else g_bin[XP2].right = P;
      g_parent[P] = XP2;
      g_bin[P].left = X;
      g_bin[P].right = N;
      g_parent[X] = P;
      g_parent[N] = P;
      ++applied;
    }
    // refit everything bottom-up (post-order over the new topology)
    {
      std::vector<int> st = {g_root}, post;
      while (!st.empty()) {
        const int b = st.back();
        st.pop_back();
        if (g_bin[b].leaf()) continue;
        post.push_back(b);
        st.push_back(g_bin[b].left), st.push_back(g_bin[b].right);
      }
      for (size_t q = post.size(); q-- > 0;) {
        BNode& b = g_bin[post[q]];
        Box nb = g_bin[b.left].box;
        nb.grow(g_bin[b.right].box);
        b.box = nb;
      }
      if (post.size() != (size_t)(nn - 1) / 2 + 0 && round == 0) printf("(inner nodes reachable: %zu)\n", post.size());
    }
    printf("parallel reinsertion round %d (every %d-th node): %zu candidates, %zu applied, SAH cost %.2f\n", round, k, cands, applied, sah_cost(g_root));
  }
}

// ---- wide tree ---------------------------------------------------------------------------------------------------
struct WChild {
  Box box;
  int node = -1;  // >= 0 wide node, < 0: leaf, tri = ~node
};
struct WNode {
  int n = 0;
  WChild c[8];
};
struct Wide {
  int W = 4;
  std::vector<WNode> nodes;
  bool quantised = false;
};

// r05: what would planes stored as FP8 E4M3 cost in node visits?  gfx950 converts TWO fp8 values per
// half-rate instruction (v_cvt_scalef32_pk_f32_fp8: 12 instead of the node step's 24 v_cvt_f32_ubyte).  Encoding probed
// (QUANT=fp8): `lo` planes as e4m3 offsets UP from the node's low corner, `hi` planes as e4m3 offsets DOWN from the node's high
// reference corner lo + 256 * scale (scale = the power of two with extent <= 256 * scale), both rounded towards zero = outward.
// E4M3 holds integers exactly up to 16 and then in steps of 2 / 4 / 8 / 16 up to 256: a plane q grid units from its corner
// moves outward by up to q / 16 units (8 bits: never more than one unit).
static float e4m3_floor(float v) {  // largest e4m3 value (OCP "fn": 3 mantissa bits, subnormal step 2^-9, max 448) <= v, v >= 0
  if (!(v > 0.0f)) return 0.0f;
  if (v >= 448.0f) return 448.0f;
  int e;
  (void)frexpf(v, &e);                        // v = f * 2^e, f in [0.5, 1)
  float step = ldexpf(1.0f, std::max(e - 4, -9));  // 3 mantissa bits below the leading one; subnormals: 2^-9
  return floorf(v / step) * step;
}
static void quantise_fp8(WNode& nd) {
  Box b;
  for (int k = 0; k < nd.n; ++k) b.grow(nd.c[k].box);
  float lo[3] = {b.lo.x, b.lo.y, b.lo.z}, hi[3] = {b.hi.x, b.hi.y, b.hi.z}, sc[3];
  for (int a = 0; a < 3; ++a) {
    float ext = hi[a] - lo[a];
    int ex = -100;
    if (ext > 0) (void)frexpf(ext / 256.0f, &ex);
    sc[a] = ldexpf(1.0f, ex);
    while (lo[a] + 256.0f * sc[a] < hi[a]) sc[a] *= 2;
  }
  for (int k = 0; k < nd.n; ++k) {
    float* cl = &nd.c[k].box.lo.x;
    float* ch = &nd.c[k].box.hi.x;
    for (int a = 0; a < 3; ++a) {
      const float ref = lo[a] + 256.0f * sc[a];
      const float ql = e4m3_floor(std::max(0.0f, (cl[a] - lo[a]) / sc[a]));
      const float qh = e4m3_floor(std::max(0.0f, (ref - ch[a]) / sc[a]));
      cl[a] = lo[a] + ql * sc[a];
      ch[a] = ref - qh * sc[a];
    }
  }
}

// quantise the child boxes of a node to an 8-bit grid of the node box (outward), like encode_node4
static void quantise(WNode& nd) {
  if (getenv("QUANT") && !strcmp(getenv("QUANT"), "fp8")) return quantise_fp8(nd);
  Box b;
  for (int k = 0; k < nd.n; ++k) b.grow(nd.c[k].box);
  float lo[3] = {b.lo.x, b.lo.y, b.lo.z}, hi[3] = {b.hi.x, b.hi.y, b.hi.z}, sc[3];
  for (int a = 0; a < 3; ++a) {
    float ext = hi[a] - lo[a];
    int ex = -100;
    if (ext > 0) (void)frexpf(ext / 255.0f, &ex);
    sc[a] = ldexpf(1.0f, ex);
    while (lo[a] + 255.0f * sc[a] < hi[a]) sc[a] *= 2;
    if (getenv("TIGHT_SCALE") && ext > 0) {  // any float step instead of a power of two: extent / 255, nudged up until 255 steps reach the far side
      sc[a] = ext / 255.0f;
      while (lo[a] + 255.0f * sc[a] < hi[a]) sc[a] = nextafterf(sc[a], 3e38f);
    }
  }
  for (int k = 0; k < nd.n; ++k) {
    float* cl = &nd.c[k].box.lo.x;
    float* ch = &nd.c[k].box.hi.x;
    for (int a = 0; a < 3; ++a) {
      float ql = std::min(255.0f, std::max(0.0f, floorf((cl[a] - lo[a]) / sc[a])));
      float qh = std::min(255.0f, std::max(0.0f, ceilf((ch[a] - lo[a]) / sc[a])));
      cl[a] = lo[a] + ql * sc[a];
      ch[a] = lo[a] + qh * sc[a];
    }
  }
}

// greedy collapse: open the inner child with the largest area until W children (or none left to open)
static int collapse_greedy(Wide& T, int bnode) {
  int me = (int)T.nodes.size();
  T.nodes.emplace_back();
  std::vector<int> ch = {g_bin[bnode].left, g_bin[bnode].right};
  while ((int)ch.size() < T.W) {
    int best = -1;
    float ba = -1;
    for (int k = 0; k < (int)ch.size(); ++k)
      if (!g_bin[ch[k]].leaf() && g_bin[ch[k]].box.area() > ba) ba = g_bin[ch[k]].box.area(), best = k;
    if (best < 0) break;
    int o = ch[best];
    ch[best] = g_bin[o].left;
    ch.push_back(g_bin[o].right);
  }
  WNode nd;
  nd.n = (int)ch.size();
  for (int k = 0; k < nd.n; ++k) nd.c[k].box = g_bin[ch[k]].box;
  std::vector<int> kids(nd.n);
  for (int k = 0; k < nd.n; ++k) kids[k] = g_bin[ch[k]].leaf() ? ~g_bin[ch[k]].right : collapse_greedy(T, ch[k]);
  for (int k = 0; k < nd.n; ++k) nd.c[k].node = kids[k];
  T.nodes[me] = nd;
  return me;
}
// parity collapse (the product's default for W = 4): children = grandchildren
static int collapse_parity(Wide& T, int bnode) {
  int me = (int)T.nodes.size();
  T.nodes.emplace_back();
  std::vector<int> ch;
  for (int c : {g_bin[bnode].left, g_bin[bnode].right}) {
    if (g_bin[c].leaf()) ch.push_back(c);
    else ch.push_back(g_bin[c].left), ch.push_back(g_bin[c].right);
  }
  WNode nd;
  nd.n = (int)ch.size();
  for (int k = 0; k < nd.n; ++k) nd.c[k].box = g_bin[ch[k]].box;
  std::vector<int> kids(nd.n);
  for (int k = 0; k < nd.n; ++k) kids[k] = g_bin[ch[k]].leaf() ? ~g_bin[ch[k]].right : collapse_parity(T, ch[k]);
  for (int k = 0; k < nd.n; ++k) nd.c[k].node = kids[k];
  T.nodes[me] = nd;
  return me;
}

// SAH-optimal collapse by dynamic programming (Ylitie, Karras, Laine 2017, section 3.1, leaves of one triangle):
//   C(n, 1) = leaf ? A_n : A_n + D(n, W)                  n as ONE root: a triangle, or a wide node over W roots below it
//   C(n, i) = min(D(n, i), C(n, i - 1))                    a forest of at most i roots
//   D(n, j) = min over 0 < k < j of C(left, k) + C(right, j - k)
static std::vector<float> g_cost;    // C: [node * 8 + (i - 1)]
static std::vector<int8_t> g_split;  // for C(n, i), i >= 2: 0 = "use C(n, i - 1)", k > 0 = D(n, i) with k roots on the left
static std::vector<int8_t> g_dsplit; // best k of D(n, j): [node * 8 + (j - 1)]
static void dp_cost(int W, int n, float root_area) {
  BNode& b = g_bin[n];
  float* C = &g_cost[(size_t)n * 8];
  int8_t* S = &g_split[(size_t)n * 8];
  int8_t* DS = &g_dsplit[(size_t)n * 8];
  const float an = b.box.area() / root_area;
  if (b.leaf()) {
    for (int i = 0; i < W; ++i) C[i] = an, S[i] = 0;
    return;
  }
  dp_cost(W, b.left, root_area);
  dp_cost(W, b.right, root_area);
  const float* L = &g_cost[(size_t)b.left * 8];
  const float* R = &g_cost[(size_t)b.right * 8];
  float D[9];
  for (int j = 2; j <= W; ++j) {
    float best = 3e38f;
    int bs = 1;
    for (int k = 1; k < j; ++k) {
      float c = L[k - 1] + R[j - k - 1];
      if (c < best) best = c, bs = k;
    }
    D[j] = best;
    DS[j - 1] = (int8_t)bs;
  }
  C[0] = an + D[W];
  S[0] = 0;
  for (int i = 2; i <= W; ++i) {
    if (D[i] < C[i - 2]) C[i - 1] = D[i], S[i - 1] = DS[i - 1];
    else C[i - 1] = C[i - 2], S[i - 1] = 0;
  }
}
static void dp_gather(int n, int i, std::vector<int>& roots) {
  BNode& b = g_bin[n];
  if (b.leaf() || i == 1) {
    roots.push_back(n);
    return;
  }
  int8_t s = g_split[(size_t)n * 8 + (i - 1)];
  if (s == 0) {
    dp_gather(n, i - 1, roots);
    return;
  }
  dp_gather(b.left, s, roots);
  dp_gather(b.right, i - s, roots);
}
static int collapse_dp(Wide& T, int bnode) {
  int me = (int)T.nodes.size();
  T.nodes.emplace_back();
  std::vector<int> ch;
  const int ks = g_dsplit[(size_t)bnode * 8 + (T.W - 1)];
  dp_gather(g_bin[bnode].left, ks, ch);
  dp_gather(g_bin[bnode].right, T.W - ks, ch);
  WNode nd;
  nd.n = (int)ch.size();
  if (nd.n > 8) {
    fprintf(stderr, "dp gather overflow %d\n", nd.n);
    exit(1);
  }
  for (int k = 0; k < nd.n; ++k) nd.c[k].box = g_bin[ch[k]].box;
  std::vector<int> kids(nd.n);
  for (int k = 0; k < nd.n; ++k) kids[k] = g_bin[ch[k]].leaf() ? ~g_bin[ch[k]].right : collapse_dp(T, ch[k]);
  for (int k = 0; k < nd.n; ++k) nd.c[k].node = kids[k];
  T.nodes[me] = nd;
  return me;
}

// octant slot assignment (Ylitie et al. 3.2): child k -> slot s maximising sum of dot(centre_k - centre_node, dir_s),
// dir_s = (s&1 ? +1 : -1, s&2 ? +1 : -1, s&4 ? +1 : -1); greedy on the cost table.  Slots of a W=4 node: 8 virtual
// slots as well (only the ORDER matters for the counts here).
static void assign_slots(Wide& T) {
  for (WNode& nd : T.nodes) {
    Box b;
    for (int k = 0; k < nd.n; ++k) b.grow(nd.c[k].box);
    V3 c0 = b.centre();
    float cost[8][8];
    for (int k = 0; k < nd.n; ++k) {
      V3 d = nd.c[k].box.centre() - c0;
      for (int s = 0; s < 8; ++s) cost[k][s] = (s & 1 ? d.x : -d.x) + (s & 2 ? d.y : -d.y) + (s & 4 ? d.z : -d.z);
    }
    int slot_of[8];
    bool cu[8] = {false}, su[8] = {false};
    for (int it = 0; it < nd.n; ++it) {
      float best = -3e38f;
      int bk = -1, bs = -1;
      for (int k = 0; k < nd.n; ++k)
        if (!cu[k])
          for (int s = 0; s < 8; ++s)
            if (!su[s] && cost[k][s] > best) best = cost[k][s], bk = k, bs = s;
      cu[bk] = su[bs] = true;
      slot_of[bk] = bs;
    }
    WChild out[8];
    for (int s = 0; s < 8; ++s) out[s].node = 0x7fffffff;  // empty
    for (int k = 0; k < nd.n; ++k) out[slot_of[k]] = nd.c[k];
    for (int s = 0; s < 8; ++s) nd.c[s] = out[s];
    nd.n = 8;
  }
}

// ---- traversal ---------------------------------------------------------------------------------------------------
struct Ray {
  V3 o, d;
  float tmax = 1e10f;
};
static bool g_any = false;  // any-hit rays: stop at the first triangle hit inside (0.01, tmax)
struct Counts {
  double nodes = 0, boxes = 0, tris = 0, maxstack = 0, pushes = 0;
};
static inline bool tri_hit(const Tri& t, const Ray& r, float tmax, float& tout) {
  V3 e1 = t.b - t.a, e2 = t.c - t.a;
  V3 p = cross(r.d, e2);
  float det = dot(e1, p);
  if (det == 0.0f) return false;
  float inv = 1.0f / det;
  V3 s = r.o - t.a;
  float u = dot(s, p) * inv;
  if (u < 0 || u > 1) return false;
  V3 q = cross(s, e1);
  float v = dot(r.d, q) * inv;
  if (v < 0 || u + v > 1) return false;
  float tt = dot(e2, q) * inv;
  if (tt > 1e-6f && tt < tmax) {
    tout = tt;
    return true;
  }
  return false;
}
static inline bool slab(const Box& b, const Ray& r, V3 inv, float tmax, float& tn) {
  float tx0 = (b.lo.x - r.o.x) * inv.x, tx1 = (b.hi.x - r.o.x) * inv.x;
  float ty0 = (b.lo.y - r.o.y) * inv.y, ty1 = (b.hi.y - r.o.y) * inv.y;
  float tz0 = (b.lo.z - r.o.z) * inv.z, tz1 = (b.hi.z - r.o.z) * inv.z;
  float lo = std::max(std::max(std::min(tx0, tx1), std::min(ty0, ty1)), std::max(std::min(tz0, tz1), 0.0f));
  float hi = std::min(std::min(std::max(tx0, tx1), std::max(ty0, ty1)), std::min(std::max(tz0, tz1), tmax));
  tn = lo;
  return lo <= hi;
}

// distance-sorted traversal (the product's rule): children hit are visited nearest first, the others pushed far -> near
static float trace_sorted(const Wide& T, const Ray& r, Counts& C) {
  V3 inv{1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z};
  float best = r.tmax;
  int stack[256];
  int sp = 0;
  int cur = 0;
  int maxsp = 0;
  for (;;) {
    if (cur >= 0) {
      const WNode& nd = T.nodes[cur];
      C.nodes += 1;
      C.boxes += T.W;
      float tn[8];
      int id[8];
      int h = 0;
      for (int k = 0; k < nd.n; ++k) {
        float t;
        if (nd.c[k].node != 0x7fffffff && slab(nd.c[k].box, r, inv, best, t)) tn[h] = t, id[h] = nd.c[k].node, ++h;
      }
      for (int i = 1; i < h; ++i)
        for (int j = i; j > 0 && tn[j] < tn[j - 1]; --j) std::swap(tn[j], tn[j - 1]), std::swap(id[j], id[j - 1]);
      for (int k = h - 1; k >= 1; --k) stack[sp++] = id[k];
      C.pushes += h > 1 ? h - 1 : 0;
      maxsp = std::max(maxsp, sp);
      if (h) {
        cur = id[0];
        continue;
      }
    } else {
      float t;
      C.tris += 1;
      if (tri_hit(g_tris[~cur], r, best, t)) {
        best = t;
        if (g_any) return 0.0f;
      }
    }
    if (!sp) break;
    cur = stack[--sp];
  }
  C.maxstack = std::max(C.maxstack, (double)maxsp);
  return best;
}

// octant-order traversal: a node's hit children are visited in the order slot ^ oct ascending; one stack entry per
// node (the remaining-children mask), as in the compressed wide BVH
static float trace_octant(const Wide& T, const Ray& r, Counts& C, bool leaves_first) {
  V3 inv{1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z};
  const int oct = (r.d.x < 0 ? 1 : 0) | (r.d.y < 0 ? 2 : 0) | (r.d.z < 0 ? 4 : 0);
  float best = 1e10f;
  struct Ent {
    int node;
    uint32_t mask;
  };
  Ent stack[128];
  int sp = 0, maxsp = 0;
  Ent cur{0, 0};
  bool fresh = true;  // cur.node not yet tested
  for (;;) {
    if (fresh) {
      const WNode& nd = T.nodes[cur.node];
      C.nodes += 1;
      C.boxes += T.W;
      uint32_t m = 0;
      for (int s = 0; s < 8; ++s) {
        float t;
        if (nd.c[s].node != 0x7fffffff && slab(nd.c[s].box, r, inv, best, t)) m |= 1u << (s ^ oct);  // priority position
      }
      if (leaves_first) {  // all leaf children of the node are tested right away (triangle group), in priority order
        for (uint32_t mm = m; mm;) {
          int p = __builtin_ctz(mm);
          mm &= mm - 1;
          int c = nd.c[p ^ oct].node;
          if (c < 0) {
            float t;
            C.tris += 1;
            if (tri_hit(g_tris[~c], r, best, t)) best = t;
            m &= ~(1u << p);
          }
        }
      }
      cur.mask = m;
      fresh = false;
    }
    if (cur.mask == 0) {
      if (!sp) break;
      cur = stack[--sp];
      continue;
    }
    int p = __builtin_ctz(cur.mask);
    cur.mask &= cur.mask - 1;
    const WNode& nd = T.nodes[cur.node];
    int c = nd.c[p ^ oct].node;
    if (c < 0) {
      float t;
      C.tris += 1;
      if (tri_hit(g_tris[~c], r, best, t)) best = t;
      continue;
    }
    // the child box was tested against the `best` of that time; re-test is not done by CWBVH either
    if (cur.mask) {
      stack[sp++] = cur;
      C.pushes += 1;
      maxsp = std::max(maxsp, sp);
    }
    cur.node = c;
    fresh = true;
  }
  C.maxstack = std::max(C.maxstack, (double)maxsp);
  return best;
}

// per-node, per-octant visiting order from a static key (W4T: the order is a table in the node): children hit are visited in
// increasing key for the ray's octant.  key_mode 0: box centre projected on the octant diagonal; 1: the corner of the box
// the octant's rays enter through, projected on the diagonal; 2: centre projection weighted by the node's extent^-1
static int g_key_mode = 0;
static float trace_table(const Wide& T, const Ray& r, Counts& C) {
  V3 inv{1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z};
  const V3 sg{r.d.x < 0 ? -1.0f : 1.0f, r.d.y < 0 ? -1.0f : 1.0f, r.d.z < 0 ? -1.0f : 1.0f};
  float best = r.tmax;
  struct Ent {
    int node;
    int order[8];
    int cnt, pos;
  };
  static thread_local Ent stack[128];
  int sp = 0, maxsp = 0;
  Ent cur;
  cur.node = 0;
  bool fresh = true;
  for (;;) {
    if (fresh) {
      const WNode& nd = T.nodes[cur.node];
      C.nodes += 1;
      C.boxes += T.W;
      float key[8];
      int id[8], h = 0;
      Box nb;
      for (int k = 0; k < nd.n; ++k) if (nd.c[k].node != 0x7fffffff) nb.grow(nd.c[k].box);
      V3 ext = nb.hi - nb.lo;
      for (int k = 0; k < nd.n; ++k) {
        float t;
        if (nd.c[k].node != 0x7fffffff && slab(nd.c[k].box, r, inv, best, t)) {
          const Box& b = nd.c[k].box;
          V3 p = g_key_mode == 1 ? V3{sg.x > 0 ? b.lo.x : b.hi.x, sg.y > 0 ? b.lo.y : b.hi.y, sg.z > 0 ? b.lo.z : b.hi.z} : b.centre();
          if (g_key_mode == 2) p = V3{ext.x > 0 ? p.x / ext.x : 0, ext.y > 0 ? p.y / ext.y : 0, ext.z > 0 ? p.z / ext.z : 0};
          key[h] = p.x * sg.x + p.y * sg.y + p.z * sg.z;
          if (g_key_mode == 3) {  // dominant axis of the ray only (6 classes)
            float ax = std::fabs(r.d.x), ay = std::fabs(r.d.y), az = std::fabs(r.d.z);
            key[h] = ax >= ay && ax >= az ? p.x * sg.x : (ay >= az ? p.y * sg.y : p.z * sg.z);
          }
          if (g_key_mode == 4) key[h] = p.x * r.d.x + p.y * r.d.y + p.z * r.d.z;  // exact direction (not static: bound)
          if (g_key_mode == 5) key[h] = -b.area();  // largest box first
          if (g_key_mode == 6) key[h] = t;  // entry distance (what the sort does)
          id[h] = k;
          ++h;
        }
      }
      for (int i = 1; i < h; ++i)
        for (int j = i; j > 0 && key[j] < key[j - 1]; --j) std::swap(key[j], key[j - 1]), std::swap(id[j], id[j - 1]);
      // leaves first (triangle group), then inner children in order
      cur.cnt = 0;
      for (int i = 0; i < h; ++i) {
        int c = nd.c[id[i]].node;
        if (c < 0) {
          float t;
          C.tris += 1;
          if (tri_hit(g_tris[~c], r, best, t)) {
            best = t;
            if (g_any) return 0.0f;
          }
        } else cur.order[cur.cnt++] = c;
      }
      cur.pos = 0;
      fresh = false;
    }
    if (cur.pos == cur.cnt) {
      if (!sp) break;
      cur = stack[--sp];
      continue;
    }
    int c = cur.order[cur.pos++];
    if (cur.pos < cur.cnt) {
      stack[sp++] = cur;
      C.pushes += 1;
      maxsp = std::max(maxsp, sp);
    }
    cur.node = c;
    fresh = true;
  }
  C.maxstack = std::max(C.maxstack, (double)maxsp);
  return best;
}

int main(int argc, char** argv) {
  if (argc < 2) return 1;
  FILE* f = fopen(argv[1], "rb");
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  size_t n = sz / 36;
  g_tris.resize(n);
  if (fread(g_tris.data(), 36, n, f) != n) return 1;
  fclose(f);
  const int nrays = argc > 2 ? atoi(argv[2]) : 200000;
  g_tbox.resize(n);
  for (size_t i = 0; i < n; ++i) {
    g_tbox[i].grow(g_tris[i].a);
    g_tbox[i].grow(g_tris[i].b);
    g_tbox[i].grow(g_tris[i].c);
  }
  // SPLIT_GRID=G (r03): triangles larger than a cell of a G^3 grid over the scene's longest extent are entered once per
  // cell they reach, each copy with the box of its clipped part (early split clipping): what does the tree gain?
  const std::vector<Tri> orig_tris = g_tris;  // rays start on the triangles, not on their references
  if (const char* sg = getenv("SPLIT_GRID")) {
    const int G = atoi(sg);
    Box sb;
    for (size_t i = 0; i < n; ++i) sb.grow(g_tbox[i]);
    const float ext = std::max(sb.hi.x - sb.lo.x, std::max(sb.hi.y - sb.lo.y, sb.hi.z - sb.lo.z));
    const float cell = ext / G;
    std::vector<Tri> nt;
    std::vector<Box> nb;
    size_t big = 0;
    for (size_t i = 0; i < n; ++i) {
      const Box& b = g_tbox[i];
      const float e = std::max(b.hi.x - b.lo.x, std::max(b.hi.y - b.lo.y, b.hi.z - b.lo.z));
      if (G <= 0 || e <= cell) {
        nt.push_back(g_tris[i]);
        nb.push_back(b);
        continue;
      }
      ++big;
      int c0[3], c1[3];
      for (int k = 0; k < 3; ++k) {
        c0[k] = std::max(0, std::min(G - 1, (int)std::floor((comp(b.lo, k) - comp(sb.lo, k)) / cell)));
        c1[k] = std::max(0, std::min(G - 1, (int)std::floor((comp(b.hi, k) - comp(sb.lo, k)) / cell)));
      }
      for (int cz = c0[2]; cz <= c1[2]; ++cz)
        for (int cy = c0[1]; cy <= c1[1]; ++cy)
          for (int cx = c0[0]; cx <= c1[0]; ++cx) {
            const int cc[3] = {cx, cy, cz};
            double poly[16][3], tmp[16][3];
            int m = 3;
            const V3 vv[3] = {g_tris[i].a, g_tris[i].b, g_tris[i].c};
            for (int q = 0; q < 3; ++q) poly[q][0] = vv[q].x, poly[q][1] = vv[q].y, poly[q][2] = vv[q].z;
            for (int k = 0; k < 3 && m > 0; ++k)
              for (int side = 0; side < 2 && m > 0; ++side) {
                const double plane = comp(sb.lo, k) + (double)cell * (cc[k] + side);
                const double sgn = side ? -1.0 : 1.0;  // keep sgn * (x - plane) >= 0
                int o = 0;
                for (int q = 0; q < m; ++q) {
                  const double* A = poly[q];
                  const double* Bq = poly[(q + 1) % m];
                  const double da = sgn * (A[k] - plane), db = sgn * (Bq[k] - plane);
                  if (da >= 0) { tmp[o][0] = A[0]; tmp[o][1] = A[1]; tmp[o][2] = A[2]; ++o; }
                  if ((da >= 0) != (db >= 0)) {
                    const double t = da / (da - db);
                    for (int c = 0; c < 3; ++c) tmp[o][c] = A[c] + t * (Bq[c] - A[c]);
                    tmp[o][k] = plane;
                    ++o;
                  }
                }
                m = o;
                for (int q = 0; q < m; ++q) for (int c = 0; c < 3; ++c) poly[q][c] = tmp[q][c];
              }
            if (m < 3) continue;
            Box cb;
            for (int q = 0; q < m; ++q) cb.grow(V3{(float)poly[q][0], (float)poly[q][1], (float)poly[q][2]});
            nt.push_back(g_tris[i]);
            nb.push_back(cb);
          }
    }
    printf("SPLIT_GRID %d: %zu of %zu triangles larger than a cell -> %zu references\n", G, big, n, nt.size());
    g_tris.swap(nt);
    g_tbox.swap(nb);
    n = g_tris.size();
  }
  std::vector<int> idx(n);
  for (size_t i = 0; i < n; ++i) idx[i] = (int)i;
  g_bin.reserve(2 * n);
  const int ploc_r = argc > 3 ? atoi(argv[3]) : 0;  // > 0: PLOC with this radius instead of the binned-SAH builder
  if (ploc_r > 0) g_root = build_ploc(ploc_r);
  else build_binary(idx, 0, (int)n);
  printf("%zu triangles, %zu binary nodes, %s, SAH cost %.2f\n", n, g_bin.size(), ploc_r > 0 ? "PLOC" : "binned SAH", sah_cost(g_root));
  if (const char* bg = getenv("PREINSERT_BUDGET")) g_budget = atol(bg);
  if (const char* pr = getenv("PREINSERT")) parallel_reinsertion(atoi(pr));
  if (const char* rp = getenv("REINSERT"))
    for (int k = 0; k < atoi(rp); ++k) reinsertion_pass();

  // rays: diffuse bounces -- a point on a random triangle, cosine-distributed direction about its (randomly flipped) normal
  std::mt19937 rng(12345);
  std::uniform_real_distribution<float> U(0.0f, 1.0f);
  std::vector<Ray> rays;
  // area-weighted triangle choice
  const size_t on = orig_tris.size();
  std::vector<double> cdf(on);
  double acc = 0;
  for (size_t i = 0; i < on; ++i) {
    V3 c = cross(orig_tris[i].b - orig_tris[i].a, orig_tris[i].c - orig_tris[i].a);
    acc += 0.5 * std::sqrt((double)dot(c, c));
    cdf[i] = acc;
  }
  while ((int)rays.size() < nrays) {
    size_t ti = std::lower_bound(cdf.begin(), cdf.end(), U(rng) * acc) - cdf.begin();
    if (ti >= on) ti = on - 1;
    const Tri& t = orig_tris[ti];
    float u = U(rng), v = U(rng);
    if (u + v > 1) u = 1 - u, v = 1 - v;
    V3 p = t.a + (t.b - t.a) * u + (t.c - t.a) * v;
    V3 nn = cross(t.b - t.a, t.c - t.a);
    float l = std::sqrt(dot(nn, nn));
    if (l == 0) continue;
    nn = nn * (1.0f / l);
    if (U(rng) < 0.5f) nn = nn * -1.0f;
    V3 a = std::fabs(nn.x) > 0.9f ? V3{0, 1, 0} : V3{1, 0, 0};
    V3 t1 = cross(nn, a);
    t1 = t1 * (1.0f / std::sqrt(dot(t1, t1)));
    V3 t2 = cross(nn, t1);
    float r1 = U(rng), r2 = U(rng), rr = std::sqrt(r1), ph = 6.2831853f * r2;
    V3 d = t1 * (rr * std::cos(ph)) + t2 * (rr * std::sin(ph)) + nn * std::sqrt(std::max(0.0f, 1 - r1));
    rays.push_back({p + nn * 1e-4f, d});
  }

  auto run = [&](const char* name, const Wide& T, int mode) {
    Counts C;
    double sum = 0;
    for (const Ray& r : rays) sum += mode == 0 ? trace_sorted(T, r, C) : mode >= 3 ? (g_key_mode = mode - 3, trace_table(T, r, C)) : trace_octant(T, r, C, mode == 2);
    printf("%-56s nodes %7zu | per ray: visits %6.2f  box tests %6.1f  tris %5.2f  pushes %5.2f  max stack %3.0f  (checksum %.6g)\n", name,
           T.nodes.size(), C.nodes / nrays, C.boxes / nrays, C.tris / nrays, C.pushes / nrays, C.maxstack, sum / nrays);
  };
  for (int quant = 0; quant < 2; ++quant) {
    printf("---- %s child boxes\n", quant ? "8-bit quantised" : "float");
    {
      Wide T;
      T.W = 4;
      collapse_parity(T, g_root);
      if (quant) for (auto& nd : T.nodes) quantise(nd);
      run("BVH4 parity collapse, distance-sorted", T, 0);
      run("BVH4 parity, per-octant table (centre)", T, 3);
      run("BVH4 parity, per-octant table (entry corner)", T, 4);
      run("BVH4 parity, per-octant table (centre/extent)", T, 5);
      assign_slots(T);
      run("BVH4 parity collapse, octant order", T, 1);
    }
    {
      Wide T;
      T.W = 4;
      collapse_greedy(T, g_root);
      if (quant) for (auto& nd : T.nodes) quantise(nd);
      run("BVH4 greedy collapse, distance-sorted", T, 0);
      assign_slots(T);
      run("BVH4 greedy collapse, octant order", T, 1);
    }
    {
      Wide T;
      T.W = 8;
      collapse_greedy(T, g_root);
      if (quant) for (auto& nd : T.nodes) quantise(nd);
      run("BVH8 greedy collapse, distance-sorted", T, 0);
      assign_slots(T);
      run("BVH8 greedy collapse, octant order", T, 1);
      run("BVH8 greedy collapse, octant, leaves first", T, 2);
    }
    for (int W : {4, 8}) {
      g_cost.assign(g_bin.size() * 8, 0.0f);
      g_split.assign(g_bin.size() * 8, 0);
      g_dsplit.assign(g_bin.size() * 8, 0);
      dp_cost(W, g_root, g_bin[g_root].box.area());
      Wide T;
      T.W = W;
      collapse_dp(T, g_root);
      if (quant) for (auto& nd : T.nodes) quantise(nd);
      char nm[64];
      snprintf(nm, sizeof nm, "BVH%d SAH-optimal (DP) collapse, sorted", W);
      run(nm, T, 0);
      snprintf(nm, sizeof nm, "BVH%d SAH-optimal (DP), table (centre)", W);
      run(nm, T, 3);
      snprintf(nm, sizeof nm, "BVH%d SAH-optimal (DP), table (entry corner)", W);
      run(nm, T, 4);
      assign_slots(T);
      snprintf(nm, sizeof nm, "BVH%d SAH-optimal (DP) collapse, octant", W);
      run(nm, T, 1);
      snprintf(nm, sizeof nm, "BVH%d SAH-optimal (DP), octant, leaves first", W);
      run(nm, T, 2);
    }
  }
  // ---- any-hit (shadow) rays: from the same surface points towards points below the ceiling ----
  {
    Box sb;
    for (size_t i = 0; i < n; ++i) sb.grow(g_tbox[i]);
    std::vector<Ray> srays;
    for (const Ray& r : rays) {
      V3 tgt{sb.lo.x + (0.2f + 0.6f * U(rng)) * (sb.hi.x - sb.lo.x), sb.lo.y + 0.97f * (sb.hi.y - sb.lo.y),
             sb.lo.z + (0.2f + 0.6f * U(rng)) * (sb.hi.z - sb.lo.z)};
      V3 d = tgt - r.o;
      float l = std::sqrt(dot(d, d));
      if (l < 0.05f) continue;
      Ray s;
      s.o = r.o;
      s.d = d * (1.0f / l);
      s.tmax = l - 0.01f;
      srays.push_back(s);
    }
    rays = srays;
    g_any = true;
    printf("==== any-hit rays (%zu), 8-bit quantised boxes\n", rays.size());
    for (int variant = 0; variant < 2; ++variant) {
      Wide T;
      T.W = 4;
      if (variant == 0) collapse_parity(T, g_root); else collapse_greedy(T, g_root);
      for (auto& nd : T.nodes) quantise(nd);
      const char* vn = variant == 0 ? "BVH4 parity" : "BVH4 greedy";
      char nm[96];
      const char* keys[7] = {"centre . octant diagonal", "entry corner . diagonal", "centre/extent . diagonal", "centre on the dominant axis",
                             "centre . exact direction (bound)", "largest box first", "entry distance (table machinery)"};
      snprintf(nm, sizeof nm, "%s, distance-sorted", vn);
      run(nm, T, 0);
      for (int k = 0; k < 7; ++k) {
        snprintf(nm, sizeof nm, "%s, %s", vn, keys[k]);
        run(nm, T, 3 + k);
      }
    }
    g_any = false;
  }
  return 0;
}
