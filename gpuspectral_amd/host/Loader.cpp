// Loader.cpp -- Mitsuba 0.5 XML + Wavefront OBJ -> Scene.
//
// Behaviour follows the reference loader S/engine/Loader.cpp:
//   loadMesh        :19-64    de-index pos/normal/uv, concatenate all shapes
//   loadMaterial    :145-234  twosided / diffuse / roughplastic / dielectric /
//                             conductor / plastic / roughconductor, recursing into child bsdfs
//   loadScene       :253-349  shapes (obj, rectangle, cube, disk), to_world, center,
//                             area emitters -> one TriangleLight per triangle, sensor
// The two parsing libraries it calls (TinyParser-Mitsuba, tinyobjloader) are absent
// from the reference tree; the subset of their behaviour the call sites rely on
//  is implemented here: camelCase -> snake_case property names,
// <ref id> resolution into anonymous children, typed property getters with defaults
// (number 0, colour 0, bool default), fan triangulation, relative OBJ indices.
#include "Loader.h"
#include "Image.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <unordered_map>

namespace GPUSpectral {

namespace {

std::string g_defaultAssetDir;

// ---------------------------------------------------------------------------
// minimal XML reader (elements, attributes, comments, <?...?>, self-closing tags)
// ---------------------------------------------------------------------------
struct XmlNode {
  std::string tag;
  std::vector<std::pair<std::string, std::string>> attrs;
  std::vector<std::unique_ptr<XmlNode>> children;
  const std::string* attr(const char* name) const {
    for (auto& a : attrs)
      if (a.first == name) return &a.second;
    return nullptr;
  }
  std::string get(const char* name, const std::string& def = "") const {
    auto* a = attr(name);
    return a ? *a : def;
  }
};

class XmlParser {
 public:
  explicit XmlParser(const std::string& text) : s(text) {}
  std::unique_ptr<XmlNode> parseDocument() {
    skipMisc();
    auto root = parseElement();
    if (!root) throw std::runtime_error("XML: no root element");
    return root;
  }

 private:
  const std::string& s;
  size_t p = 0;
  void skipWs() {
    while (p < s.size() && std::isspace((unsigned char)s[p])) ++p;
  }
  bool starts(const char* lit) const { return s.compare(p, std::strlen(lit), lit) == 0; }
  void skipMisc() {
    for (;;) {
      skipWs();
      if (starts("<?")) {
        size_t e = s.find("?>", p);
        if (e == std::string::npos) throw std::runtime_error("XML: unterminated <?");
        p = e + 2;
      } else if (starts("<!--")) {
        size_t e = s.find("-->", p);
        if (e == std::string::npos) throw std::runtime_error("XML: unterminated comment");
        p = e + 3;
      } else if (starts("<!")) {
        size_t e = s.find('>', p);
        if (e == std::string::npos) throw std::runtime_error("XML: unterminated <!");
        p = e + 1;
      } else {
        return;
      }
    }
  }
  static std::string unescape(const std::string& v) {
    std::string o;
    for (size_t i = 0; i < v.size(); ++i) {
      if (v[i] == '&') {
        if (v.compare(i, 4, "&lt;") == 0) { o += '<'; i += 3; continue; }
        if (v.compare(i, 4, "&gt;") == 0) { o += '>'; i += 3; continue; }
        if (v.compare(i, 5, "&amp;") == 0) { o += '&'; i += 4; continue; }
        if (v.compare(i, 6, "&quot;") == 0) { o += '"'; i += 5; continue; }
        if (v.compare(i, 6, "&apos;") == 0) { o += '\''; i += 5; continue; }
      }
      o += v[i];
    }
    return o;
  }
  std::string name() {
    size_t b = p;
    while (p < s.size() && (std::isalnum((unsigned char)s[p]) || s[p] == '_' || s[p] == '-' || s[p] == ':' || s[p] == '.'))
      ++p;
    return s.substr(b, p - b);
  }
  std::unique_ptr<XmlNode> parseElement() {
    skipMisc();
    if (p >= s.size() || s[p] != '<') return nullptr;
    ++p;
    auto n = std::make_unique<XmlNode>();
    n->tag = name();
    if (n->tag.empty()) throw std::runtime_error("XML: bad tag near offset " + std::to_string(p));
    for (;;) {
      skipWs();
      if (p >= s.size()) throw std::runtime_error("XML: unexpected end in <" + n->tag + ">");
      if (starts("/>")) {
        p += 2;
        return n;
      }
      if (s[p] == '>') {
        ++p;
        break;
      }
      std::string an = name();
      skipWs();
      if (an.empty() || p >= s.size() || s[p] != '=') throw std::runtime_error("XML: bad attribute in <" + n->tag + ">");
      ++p;
      skipWs();
      char q = s[p];
      if (q != '"' && q != '\'') throw std::runtime_error("XML: unquoted attribute in <" + n->tag + ">");
      size_t e = s.find(q, p + 1);
      if (e == std::string::npos) throw std::runtime_error("XML: unterminated attribute value");
      n->attrs.emplace_back(an, unescape(s.substr(p + 1, e - p - 1)));
      p = e + 1;
    }
    for (;;) {
      // text content is irrelevant for Mitsuba scenes: skip to the next tag
      size_t lt = s.find('<', p);
      if (lt == std::string::npos) throw std::runtime_error("XML: missing </" + n->tag + ">");
      p = lt;
      if (starts("</")) {
        p += 2;
        std::string cn = name();
        skipWs();
        if (p < s.size() && s[p] == '>') ++p;
        if (cn != n->tag) throw std::runtime_error("XML: </" + cn + "> closes <" + n->tag + ">");
        return n;
      }
      if (starts("<!--") || starts("<?") || starts("<!")) {
        skipMisc();
        continue;
      }
      n->children.push_back(parseElement());
    }
  }
};

// ---------------------------------------------------------------------------
// stand-in for tinyparser_mitsuba::Object
// ---------------------------------------------------------------------------
enum PropType { PT_NUMBER, PT_BOOL, PT_STRING, PT_COLOR, PT_VECTOR, PT_TRANSFORM };
struct Prop {
  PropType type;
  double number = 0;
  bool flag = false;
  std::string str;
  float v[3] = {0, 0, 0};
  float matrix[16];  // row-major, as written in the file
};
struct Object {
  std::string kind;    // tag: shape, bsdf, sensor, emitter, texture, ...
  std::string plugin;  // type attribute
  std::map<std::string, Prop> props;
  std::vector<std::shared_ptr<Object>> children;                        // anonymousChildren()
  std::vector<std::pair<std::string, std::shared_ptr<Object>>> named;   // namedChildren()

  bool has(const std::string& n) const { return props.count(n) != 0; }
  float number(const std::string& n, float def = 0.0f) const {
    auto it = props.find(n);
    return it != props.end() && it->second.type == PT_NUMBER ? (float)it->second.number : def;
  }
  vec3 color(const std::string& n) const {
    auto it = props.find(n);
    if (it != props.end() && it->second.type == PT_COLOR) return vec3{it->second.v[0], it->second.v[1], it->second.v[2]};
    return vec3{};
  }
  std::string string(const std::string& n) const {
    auto it = props.find(n);
    return it != props.end() && it->second.type == PT_STRING ? it->second.str : std::string();
  }
  bool boolean(const std::string& n, bool def) const {
    auto it = props.find(n);
    return it != props.end() && it->second.type == PT_BOOL ? it->second.flag : def;
  }
};

// toWorld -> to_world, intIOR -> int_ior, diffuseReflectance -> diffuse_reflectance
std::string snake(const std::string& s) {
  std::string o;
  for (size_t i = 0; i < s.size(); ++i) {
    unsigned char c = s[i];
    if (std::isupper(c) && i > 0 && (std::islower((unsigned char)s[i - 1]) || std::isdigit((unsigned char)s[i - 1]))) o += '_';
    o += (char)std::tolower(c);
  }
  return o;
}

std::vector<double> numbers(const std::string& text) {
  std::vector<double> out;
  const char* c = text.c_str();
  while (*c) {
    while (*c && (std::isspace((unsigned char)*c) || *c == ',')) ++c;
    if (!*c) break;
    char* e = nullptr;
    double v = std::strtod(c, &e);
    if (e == c) break;
    out.push_back(v);
    c = e;
  }
  return out;
}

bool isObjectTag(const std::string& t) {
  static const char* tags[] = {"shape", "bsdf", "sensor", "emitter", "texture", "medium", "integrator", "sampler", "film", "rfilter"};
  for (auto* k : tags)
    if (t == k) return true;
  return false;
}

// <ref id="x"/> that names an object defined later in the file: the slot in the parent's child list is kept and
// patched once the whole document has been read (resolveRefs).
struct PendingRef {
  std::shared_ptr<Object> parent;
  size_t slot;
  std::string id;
};
struct ParseState {
  std::unordered_map<std::string, std::shared_ptr<Object>> ids;
  std::vector<PendingRef> pending;
};

std::shared_ptr<Object> parseObject(const XmlNode& el, ParseState& ps) {
  auto& ids = ps.ids;
  auto o = std::make_shared<Object>();
  o->kind = el.tag;
  o->plugin = el.get("type");
  for (auto& chp : el.children) {
    const XmlNode& ch = *chp;
    const std::string nm = snake(ch.get("name"));
    Prop pr;
    if (ch.tag == "float" || ch.tag == "integer") {
      pr.type = PT_NUMBER;
      pr.number = std::strtod(ch.get("value", "0").c_str(), nullptr);
      o->props[nm] = pr;
    } else if (ch.tag == "boolean") {
      pr.type = PT_BOOL;
      std::string v = ch.get("value");
      for (auto& c : v) c = (char)std::tolower((unsigned char)c);
      pr.flag = v.find("true") != std::string::npos;
      o->props[nm] = pr;
    } else if (ch.tag == "string") {
      pr.type = PT_STRING;
      pr.str = ch.get("value");
      o->props[nm] = pr;
    } else if (ch.tag == "rgb" || ch.tag == "srgb") {
      pr.type = PT_COLOR;
      auto v = numbers(ch.get("value"));
      if (v.size() == 1) v = {v[0], v[0], v[0]};
      for (size_t k = 0; k < 3 && k < v.size(); ++k) pr.v[k] = (float)v[k];
      o->props[nm] = pr;
    } else if (ch.tag == "point" || ch.tag == "vector") {
      pr.type = PT_VECTOR;
      if (ch.attr("value")) {
        auto v = numbers(ch.get("value"));
        for (size_t k = 0; k < 3 && k < v.size(); ++k) pr.v[k] = (float)v[k];
      } else {
        pr.v[0] = (float)std::strtod(ch.get("x", "0").c_str(), nullptr);
        pr.v[1] = (float)std::strtod(ch.get("y", "0").c_str(), nullptr);
        pr.v[2] = (float)std::strtod(ch.get("z", "0").c_str(), nullptr);
      }
      o->props[nm] = pr;
    } else if (ch.tag == "transform") {
      pr.type = PT_TRANSFORM;
      for (int k = 0; k < 16; ++k) pr.matrix[k] = (k % 5 == 0) ? 1.0f : 0.0f;
      for (auto& t : ch.children)
        if (t->tag == "matrix") {
          auto v = numbers(t->get("value"));
          for (size_t k = 0; k < 16 && k < v.size(); ++k) pr.matrix[k] = (float)v[k];
        }
      o->props[nm] = pr;
    } else if (ch.tag == "ref") {
      auto it = ids.find(ch.get("id"));
      if (it != ids.end()) {
        o->children.push_back(it->second);
      } else {  // forward (or dangling) reference
        ps.pending.push_back(PendingRef{o, o->children.size(), ch.get("id")});
        o->children.push_back(nullptr);
      }
    } else if (isObjectTag(ch.tag)) {
      auto child = parseObject(ch, ps);
      if (ch.attr("id")) ids[ch.get("id")] = child;
      if (ch.attr("name")) o->named.emplace_back(nm, child);
      else o->children.push_back(child);
    }
  }
  return o;
}

// Second pass over the references that pointed forward; what is still unknown is dropped with a warning (the shape
// then keeps the default BSDF, as with the reference's parser).
void resolveRefs(ParseState& ps, std::vector<std::string>& warnings) {
  for (auto& pr : ps.pending) {
    auto it = ps.ids.find(pr.id);
    if (it != ps.ids.end()) pr.parent->children[pr.slot] = it->second;
    else warnings.push_back("unresolved <ref id=\"" + pr.id + "\"> dropped");
  }
  for (auto& pr : ps.pending) {
    auto& ch = pr.parent->children;
    ch.erase(std::remove(ch.begin(), ch.end(), nullptr), ch.end());
  }
  ps.pending.clear();
}

std::string dirOf(const std::string& path) {
  size_t s = path.find_last_of("/\\");
  return s == std::string::npos ? std::string(".") : path.substr(0, s);
}
std::string joinPath(const std::string& a, const std::string& b) {
  if (a.empty()) return b;
  if (!b.empty() && b[0] == '/') return b;
  return a + "/" + b;
}
bool fileExists(const std::string& p) {
  std::ifstream f(p);
  return f.good();
}

// S/assets/rect.obj / box.obj restated from their definition: used when
// the asset directory does not hold the files.
MeshPtr builtinQuadMesh(uint32_t id, bool box) {
  struct Face {
    float n[3];
    float c[4][3];
  };
  static const Face rect[] = {{{0, 0, 1}, {{-1, 1, 0}, {1, 1, 0}, {-1, -1, 0}, {1, -1, 0}}}};
  static const Face cube[] = {
      {{1, 0, 0}, {{1, 1, 1}, {1, 1, -1}, {1, -1, 1}, {1, -1, -1}}},
      {{-1, 0, 0}, {{-1, 1, -1}, {-1, 1, 1}, {-1, -1, -1}, {-1, -1, 1}}},
      {{0, 1, 0}, {{-1, 1, -1}, {1, 1, -1}, {-1, 1, 1}, {1, 1, 1}}},
      {{0, -1, 0}, {{-1, -1, 1}, {1, -1, 1}, {-1, -1, -1}, {1, -1, -1}}},
      {{0, 0, 1}, {{-1, 1, 1}, {1, 1, 1}, {-1, -1, 1}, {1, -1, 1}}},
      {{0, 0, -1}, {{1, 1, -1}, {-1, 1, -1}, {1, -1, -1}, {-1, -1, -1}}},
  };
  static const float uv[4][2] = {{0, 1}, {1, 1}, {0, 0}, {1, 0}};
  static const int order[6] = {0, 2, 1, 2, 3, 1};  // f 1 3 2 / f 3 4 2
  const Face* faces = box ? cube : rect;
  const int nf = box ? 6 : 1;
  std::vector<Mesh::Vertex> v;
  for (int f = 0; f < nf; ++f)
    for (int k : order) {
      Mesh::Vertex x;
      x.pos = vec3{faces[f].c[k][0], faces[f].c[k][1], faces[f].c[k][2]};
      x.normal = vec3{faces[f].n[0], faces[f].n[1], faces[f].n[2]};
      x.uv = vec2{uv[k][0], uv[k][1]};
      v.push_back(x);
    }
  return std::make_shared<Mesh>(id, std::move(v));
}

// LoadOptions::builtinShapes.  Points of the unit circle / sphere come from repeated NORMALISED MIDPOINTS of the four axis
// points / the octahedron -- float32 additions, multiplications, one square root and three divisions per point, all correctly
// rounded on every IEEE machine and commutative where two triangles share an edge (watertight), so that the numpy
// restatement (oracle/mitsuba_loader.py builtin_disk / builtin_sphere) produces the same bits.
static vec3 unitMid(const vec3& a, const vec3& b) {
  const float x = a.x + b.x, y = a.y + b.y, z = a.z + b.z;
  const float l = std::sqrt((x * x + y * y) + z * z);
  return vec3{x / l, y / l, z / l};
}
MeshPtr builtinDiskMesh(uint32_t id) {
  constexpr int N = 64;
  vec3 p[N];
  p[0] = vec3{1, 0, 0}, p[N / 4] = vec3{0, 1, 0}, p[N / 2] = vec3{-1, 0, 0}, p[3 * N / 4] = vec3{0, -1, 0};
  for (int step = N / 4; step > 1; step /= 2)
    for (int i = 0; i < N; i += step) p[i + step / 2] = unitMid(p[i], p[(i + step) % N]);
  std::vector<Mesh::Vertex> v;
  for (int k = 0; k < N; ++k) {  // fan, counter-clockwise seen from +z: the geometric normal is +z (an area emitter shines that way)
    const vec3 c[3] = {vec3{0, 0, 0}, p[k], p[(k + 1) % N]};
    for (const vec3& q : c) {
      Mesh::Vertex x;
      x.pos = q;
      x.normal = vec3{0, 0, 1};
      x.uv = vec2{0.5f + 0.5f * q.x, 0.5f + 0.5f * q.y};
      v.push_back(x);
    }
  }
  return std::make_shared<Mesh>(id, std::move(v));
}
MeshPtr builtinSphereMesh(uint32_t id) {
  struct Tri {
    vec3 a, b, c;
  };
  const vec3 px{1, 0, 0}, nx{-1, 0, 0}, py{0, 1, 0}, ny{0, -1, 0}, pz{0, 0, 1}, nz{0, 0, -1};
  // the octahedron, every face counter-clockwise seen from outside
  std::vector<Tri> t = {{px, py, pz}, {py, nx, pz}, {nx, ny, pz}, {ny, px, pz}, {py, px, nz}, {nx, py, nz}, {ny, nx, nz}, {px, ny, nz}};
  for (int level = 0; level < 3; ++level) {
    std::vector<Tri> n;
    n.reserve(t.size() * 4);
    for (const Tri& f : t) {
      const vec3 ab = unitMid(f.a, f.b), bc = unitMid(f.b, f.c), ca = unitMid(f.c, f.a);
      n.push_back({f.a, ab, ca});
      n.push_back({ab, f.b, bc});
      n.push_back({ca, bc, f.c});
      n.push_back({ab, bc, ca});
    }
    t.swap(n);
  }
  std::vector<Mesh::Vertex> v;
  for (const Tri& f : t)
    for (const vec3& q : {f.a, f.b, f.c}) {
      Mesh::Vertex x;
      x.pos = q;
      x.normal = q;  // unit sphere: the smooth normal is the position
      x.uv = vec2{0.5f + 0.5f * q.x, 0.5f + 0.5f * q.y};
      v.push_back(x);
    }
  return std::make_shared<Mesh>(id, std::move(v));
}

struct LoadContext {
  std::string parentPath;
  LoadOptions options;
  std::map<std::string, uint32_t> textureCache;  // file / checkerboard key -> hasTexture value
};

// Loader.cpp:122-143 (dormant there): the texture child named `slot` of a BSDF -> hasTexture value, 0 = none
int loadTexture(Scene& scene, LoadContext& lc, const Object& bsdf, const std::string& slot, const char* what) {
  for (auto& nc : bsdf.named) {
    if (nc.first != slot || nc.second->kind != "texture") continue;
    if (!lc.options.dormantFeatures) {
      scene.warnings.push_back(std::string(what) + ": textured " + slot + " unsupported (Loader.cpp:122-143), colour default used");
      return 0;
    }
    const Object& tex = *nc.second;
    if (tex.plugin == "bitmap") {
      const std::string file = joinPath(lc.parentPath, tex.string("filename"));
      auto it = lc.textureCache.find(file);
      if (it != lc.textureCache.end()) return (int)it->second;
      if (!fileExists(file)) {  // (the reference's own living-room names a WoodFloor.jpg its tree does not hold)
        scene.warnings.push_back(std::string(what) + ": bitmap '" + file + "' not found, colour default used");
        return 0;
      }
      Image8 img = loadBitmap(file);  // throws like the reference's loadTexture (Loader.cpp:70-72)
      const uint32_t id = scene.addTexture(Texture{img.width, img.height, std::move(img.texels)});
      lc.textureCache.emplace(file, id);
      return (int)id;
    }
    if (tex.plugin == "checkerboard") {  // Loader.cpp:127-139
      const uint32_t uSize = (uint32_t)tex.number("uscale", 1.0f), vSize = (uint32_t)tex.number("vscale", 1.0f);
      const vec3 on = tex.color("color0"), off = tex.color("color1");
      const float c0[3] = {on.x, on.y, on.z}, c1[3] = {off.x, off.y, off.z};
      Image8 img = makeCheckerboard(uSize, vSize, c0, c1);
      return (int)scene.addTexture(Texture{img.width, img.height, std::move(img.texels)});
    }
    scene.warnings.push_back(std::string(what) + ": texture type '" + tex.plugin + "' unsupported, colour default used");
    return 0;
  }
  return 0;
}

// Loader.cpp:145-234
void loadMaterial(Scene& scene, Material* material, const Object& obj, LoadContext& lc) {
  const std::string& type = obj.plugin;
  if (type == "twosided") material->twofaced = true;
  if (type == "diffuse") {
    vec3 rgb = obj.color("reflectance");
    const int tex = loadTexture(scene, lc, obj, "reflectance", "diffuse");
    material->bsdf = scene.addDiffuseBSDF(DiffuseBSDF{{rgb.x, rgb.y, rgb.z}, tex});
  } else if (type == "roughplastic") {
    vec3 rgb = obj.color("diffuse_reflectance");
    float alpha = obj.number("alpha");
    if (obj.has("ext_ior") && std::fabs(obj.number("ext_ior") - 1.0f) > 0.001f) scene.warnings.push_back("unsupported ext ior of plastic");
    float ior = obj.has("int_ior") ? obj.number("int_ior") : 1.3f;
    float R0 = (ior - 1.0f) / (ior + 1.0f);
    R0 *= R0;
    RoughPlasticBSDF b{};
    b.diffuse[0] = rgb.x;
    b.diffuse[1] = rgb.y;
    b.diffuse[2] = rgb.z;
    b.ior_in = ior;
    b.ior_out = 1.0f;
    b.r0 = R0;
    b.alpha = (float)std::sqrt(2.0f) * alpha;
    b.has_texture = loadTexture(scene, lc, obj, "diffuse_reflectance", "roughplastic");
    material->bsdf = scene.addRoughPlasticBSDF(b);
  } else if (type == "dielectric") {
    float intIOR = obj.number("int_ior");
    float extIOR = obj.number("ext_ior");
    if (!obj.has("int_ior") || !obj.has("ext_ior")) scene.warnings.push_back("dielectric without numeric int_ior/ext_ior: the reference reads 0");
    material->bsdf = scene.addSmoothDielectricBSDF(SmoothDielectricBSDF{intIOR, extIOR});
  } else if (type == "conductor") {
    float ior = obj.has("eta") ? obj.number("eta") : 0.0f;
    material->bsdf = scene.addSmoothConductorBSDF(SmoothConductorBSDF{ior, 1.0f});
  } else if (type == "plastic") {
    vec3 rgb = obj.color("diffuse_reflectance");
    if (obj.has("ext_ior") && std::fabs(obj.number("ext_ior") - 1.0f) > 0.001f) scene.warnings.push_back("unsupported ext ior of plastic");
    float ior = obj.has("int_ior") ? obj.number("int_ior") : 1.3f;
    float R0 = (ior - 1.0f) / (ior + 1.0f);
    R0 *= R0;
    SmoothPlasticBSDF b{};
    b.diffuse[0] = rgb.x;
    b.diffuse[1] = rgb.y;
    b.diffuse[2] = rgb.z;
    b.ior_in = ior;
    b.ior_out = 1.0f;
    b.r0 = R0;
    for (auto& nc : obj.named)  // SmoothPlasticBSDF has no hasTexture field (Scene.h:48-53): never textured
      if (nc.first == "diffuse_reflectance") scene.warnings.push_back("plastic: textured diffuse_reflectance unsupported (no hasTexture field), colour default used");
    material->bsdf = scene.addSmoothPlasticBSDF(b);
  } else if (type == "roughconductor") {
    vec3 eta = obj.color("eta"), k = obj.color("k"), refl = obj.color("specular_reflectance");
    float alpha = obj.number("alpha");
    RoughConductorBSDF b{};
    b.eta[0] = eta.x; b.eta[1] = eta.y; b.eta[2] = eta.z;
    b.k[0] = k.x; b.k[1] = k.y; b.k[2] = k.z;
    b.reflectance[0] = refl.x; b.reflectance[1] = refl.y; b.reflectance[2] = refl.z;
    b.alpha = (float)std::sqrt(2) * alpha;
    b.has_texture = loadTexture(scene, lc, obj, "specular_reflectance", "roughconductor");
    material->bsdf = scene.addRoughConductorBSDF(b);
  }
  for (auto& child : obj.children)
    if (child->kind == "bsdf") loadMaterial(scene, material, *child, lc);
}

}  // namespace

void setDefaultAssetDir(const std::string& dir) { g_defaultAssetDir = dir; }

// Loader.cpp:19-64
MeshPtr loadMesh(const std::string& path, uint32_t id) {
  std::ifstream in(path);
  if (!in) throw std::runtime_error("cannot open OBJ file: " + path);
  std::vector<float> v, vt, vn;
  std::vector<Mesh::Vertex> out;
  std::string line;
  bool missingNormals = false;
  struct Corner {
    int v, t, n;
  };
  std::vector<Corner> corners;
  while (std::getline(in, line)) {
    const char* c = line.c_str();
    while (*c == ' ' || *c == '\t') ++c;
    // `c` stays inside the line: every component is one whitespace-delimited token; a missing or unparsable one
    // reads as 0 (what tinyobjloader's parseReal does with its default), never past the terminating NUL
    auto reals = [&](const char* p, int count, std::vector<float>& dst) {
      for (int k = 0; k < count; ++k) {
        while (*p == ' ' || *p == '\t') ++p;
        float val = 0.0f;
        if (*p && *p != '\r' && *p != '\n') {
          char* e = nullptr;
          const double d = std::strtod(p, &e);
          if (e != p) val = (float)d;
          while (*p && *p != ' ' && *p != '\t' && *p != '\r' && *p != '\n') ++p;  // skip the whole token
        }
        dst.push_back(val);
      }
    };
    const bool ws2 = c[0] && (c[1] == ' ' || c[1] == '\t');
    const bool ws3 = c[0] && c[1] && (c[2] == ' ' || c[2] == '\t' || c[2] == '\0' || c[2] == '\r');
    if (c[0] == 'v' && ws2) {
      reals(c + 2, 3, v);
    } else if (c[0] == 'v' && c[1] == 'n' && ws3) {
      reals(c + 2, 3, vn);
    } else if (c[0] == 'v' && c[1] == 't' && ws3) {
      reals(c + 2, 2, vt);
    } else if (c[0] == 'f' && (c[1] == ' ' || c[1] == '\t')) {
      corners.clear();
      c += 2;
      for (;;) {
        while (*c == ' ' || *c == '\t') ++c;
        if (!*c || *c == '\r' || *c == '\n') break;
        Corner k{0, 0, 0};
        char* e;
        long a = std::strtol(c, &e, 10);
        if (e == c) break;
        c = e;
        k.v = a > 0 ? (int)a - 1 : (int)(v.size() / 3) + (int)a;
        k.t = k.n = -1;
        if (*c == '/') {
          ++c;
          if (*c != '/') {
            long t = std::strtol(c, &e, 10);
            if (e != c) k.t = t > 0 ? (int)t - 1 : (int)(vt.size() / 2) + (int)t;
            c = e;
          }
          if (*c == '/') {
            ++c;
            long n = std::strtol(c, &e, 10);
            if (e != c) k.n = n > 0 ? (int)n - 1 : (int)(vn.size() / 3) + (int)n;
            c = e;
          }
        }
        corners.push_back(k);
      }
      for (size_t k = 2; k < corners.size(); ++k) {  // fan triangulation
        const Corner tri[3] = {corners[0], corners[k - 1], corners[k]};
        for (const Corner& q : tri) {
          Mesh::Vertex x;
          if (q.v < 0 || (size_t)q.v * 3 + 2 >= v.size()) throw std::runtime_error("OBJ index out of range in " + path);
          x.pos = vec3{v[3 * q.v], v[3 * q.v + 1], v[3 * q.v + 2]};
          if (q.t >= 0 && (size_t)q.t * 2 + 1 < vt.size()) x.uv = vec2{vt[2 * q.t], vt[2 * q.t + 1]};
          if (q.n >= 0 && (size_t)q.n * 3 + 2 < vn.size()) x.normal = vec3{vn[3 * q.n], vn[3 * q.n + 1], vn[3 * q.n + 2]};
          else missingNormals = true;
          out.push_back(x);
        }
      }
    }
  }
  if (missingNormals) {
    // outside the reference's contract (it indexes attrib.normals unchecked, Loader.cpp:56-59):
    // substitute flat face normals so the mesh is still renderable
    for (size_t i = 0; i + 2 < out.size(); i += 3) {
      vec3 a = out[i].pos, b = out[i + 1].pos, c = out[i + 2].pos;
      float e1[3] = {b.x - a.x, b.y - a.y, b.z - a.z}, e2[3] = {c.x - a.x, c.y - a.y, c.z - a.z};
      float n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
      float l = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
      vec3 fn = l > 0 ? vec3{n[0] / l, n[1] / l, n[2] / l} : vec3{};
      out[i].normal = out[i + 1].normal = out[i + 2].normal = fn;
    }
  }
  return std::make_shared<Mesh>(id, std::move(out));
}

// Loader.cpp:253-349
Scene loadScene(const std::string& path, const std::string& assetDirArg, const LoadOptions& options) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("cannot open scene file: " + path);
  std::stringstream ss;
  ss << f.rdbuf();
  const std::string text = ss.str();
  auto root = XmlParser(text).parseDocument();
  const std::string parentPath = dirOf(path);
  const std::string assetDir = !assetDirArg.empty() ? assetDirArg : (!g_defaultAssetDir.empty() ? g_defaultAssetDir : parentPath);

  ParseState ps;
  auto top = parseObject(*root, ps);

  std::unordered_map<std::string, MeshPtr> meshCache;
  uint32_t nextMeshId = 1;
  Scene outScene;
  outScene.srgbTextures = options.srgbTextures;
  LoadContext lc;
  lc.parentPath = parentPath;
  lc.options = options;
  resolveRefs(ps, outScene.warnings);
  auto loadOrGetMesh = [&](const std::string& objPath, int builtin) -> MeshPtr {
    auto it = meshCache.find(objPath);
    if (it != meshCache.end()) return it->second;
    MeshPtr m;
    if (builtin == 3) m = builtinDiskMesh(nextMeshId++);
    else if (builtin == 4) m = builtinSphereMesh(nextMeshId++);
    else if (builtin && !fileExists(objPath)) m = builtinQuadMesh(nextMeshId++, builtin == 2);
    else m = loadMesh(objPath, nextMeshId++);
    meshCache.emplace(objPath, m);
    return m;
  };

  for (auto& obj : top->children) {
    if (obj->kind == "shape") {
      std::string filename;
      int builtin = 0;
      const std::string& pt = obj->plugin;
      if (pt == "obj") filename = joinPath(parentPath, obj->string("filename"));
      else if (pt == "rectangle") { filename = joinPath(assetDir, "rect.obj"); builtin = 1; }
      else if (pt == "cube") { filename = joinPath(assetDir, "box.obj"); builtin = 2; }
      else if (pt == "disk") {
        filename = joinPath(assetDir, "disk.obj");  // Loader.cpp:276 (no such asset in the reference tree: skipped below by default)
        if (options.builtinShapes && !fileExists(filename)) { filename = "<builtin disk>"; builtin = 3; }
      } else if (pt == "sphere" && options.builtinShapes) { filename = "<builtin sphere>"; builtin = 4; }
      else {
        outScene.warnings.push_back("unsupported shape type '" + pt + "' skipped");
        continue;
      }
      if (!builtin && !fileExists(filename)) {
        outScene.warnings.push_back("missing mesh '" + filename + "' skipped");
        continue;
      }
      MeshPtr mesh = loadOrGetMesh(filename, builtin);
      float rowMajor[16];
      for (int k = 0; k < 16; ++k) rowMajor[k] = (k % 5 == 0) ? 1.0f : 0.0f;
      auto tw = obj->props.find("to_world");
      if (tw != obj->props.end() && tw->second.type == PT_TRANSFORM) std::memcpy(rowMajor, tw->second.matrix, sizeof(rowMajor));
      bool faceNormals = obj->boolean("face_normals", false);
      mat4 matrix = transpose(make_mat4(rowMajor));  // Loader.cpp:286
      auto ce = obj->props.find("center");
      if (ce != obj->props.end() && ce->second.type == PT_VECTOR) {  // Loader.cpp:287-293
        matrix[3][0] = ce->second.v[0];
        matrix[3][1] = ce->second.v[1];
        matrix[3][2] = ce->second.v[2];
        matrix[3][3] = 1.0f;
      }
      if (builtin == 4) {  // (extension: the reference never reads `radius`) the unit sphere scaled about its centre
        const float r = obj->number("radius", 1.0f);
        for (int c = 0; c < 3; ++c)
          for (int k = 0; k < 3; ++k) matrix[c][k] = matrix[c][k] * r;
      }
      bool emitting = false;
      RenderObject renderObject;
      Material material;
      for (auto& child : obj->children) {
        if (child->kind == "bsdf") {
          loadMaterial(outScene, &material, *child, lc);
        } else if (child->kind == "emitter" && child->plugin == "area") {
          material.emission = child->color("radiance");
          emitting = true;
        }
      }
      renderObject.mesh = mesh;
      renderObject.transform = matrix;
      renderObject.material = outScene.addMaterial(material);
      outScene.getMaterial(renderObject.material).facenormals = faceNormals;
      outScene.addRenderObject(renderObject);
      if (emitting) {  // Loader.cpp:316-330
        const auto& vertices = mesh->getVertices();
        for (size_t i = 0; i + 2 < vertices.size(); i += 3) {
          TriangleLight light{};
          for (int k = 0; k < 3; ++k) {
            const vec3& p = vertices[i + k].pos;
            vec4 w = renderObject.transform * vec4{p.x, p.y, p.z, 1.0f};
            light.positions[k][0] = w.x;
            light.positions[k][1] = w.y;
            light.positions[k][2] = w.z;
            light.positions[k][3] = w.w;
          }
          light.radiance[0] = material.emission.x;
          light.radiance[1] = material.emission.y;
          light.radiance[2] = material.emission.z;
          light.radiance[3] = 1.0f;
          outScene.addTriangleLight(light);
        }
      }
    } else if (obj->kind == "sensor") {  // Loader.cpp:331-337
      float rowMajor[16];
      for (int k = 0; k < 16; ++k) rowMajor[k] = (k % 5 == 0) ? 1.0f : 0.0f;
      auto tw = obj->props.find("to_world");
      if (tw != obj->props.end() && tw->second.type == PT_TRANSFORM) std::memcpy(rowMajor, tw->second.matrix, sizeof(rowMajor));
      float fov = obj->number("fov");
      outScene.camera.setFov((float)(fov * M_PI / 180.f));
      outScene.camera.setToWorld(transpose(make_mat4(rowMajor)));
    } else if (obj->kind == "emitter") {
      if (!options.dormantFeatures || obj->plugin != "envmap") {
        outScene.warnings.push_back("top-level emitter (envmap) ignored, as in the reference (Loader.cpp:338-346)");
      } else {  // Loader.cpp:339-345, dormant there
        ImageF img = loadHdrBitmap(joinPath(parentPath, obj->string("filename")));
        float rowMajor[16];
        for (int k = 0; k < 16; ++k) rowMajor[k] = (k % 5 == 0) ? 1.0f : 0.0f;
        auto tw = obj->props.find("to_world");
        if (tw != obj->props.end() && tw->second.type == PT_TRANSFORM) std::memcpy(rowMajor, tw->second.matrix, sizeof(rowMajor));
        outScene.envMap.width = img.width;
        outScene.envMap.height = img.height;
        outScene.envMap.texels = std::move(img.texels);
        // (the dormant line is make_mat4 without the transpose the shapes and the sensor get, :286,335 vs :343; a
        // row-major file matrix read as column-major would be the inverse rotation -- the transpose is applied here)
        outScene.envMap.transform = transpose(make_mat4(rowMajor));
        outScene.hasEnvMap = true;
      }
    }
  }
  return outScene;
}

void flattenInstances(const Scene& scene, std::vector<gsp_instance>& out) {
  out.clear();
  out.reserve(scene.renderObjects.size());
  std::unordered_map<const Mesh*, std::pair<uint32_t, uint32_t>> placed;
  uint32_t next = 0;
  for (const RenderObject& obj : scene.renderObjects) {
    const Mesh* m = obj.mesh.get();
    auto it = placed.find(m);
    if (it == placed.end()) {
      const uint32_t count = (uint32_t)(m->getVertices().size() / 3 * 3);
      it = placed.emplace(m, std::make_pair(next, count)).first;
      next += count;
    }
    const Material& material = scene.getMaterial(obj.material);
    gsp_instance in{};
    std::memcpy(in.transform, obj.transform.data(), sizeof(in.transform));
    in.emission[0] = material.emission.x;
    in.emission[1] = material.emission.y;
    in.emission[2] = material.emission.z;
    in.bsdf = material.bsdf.handle;
    in.twofaced = material.twofaced ? 1u : 0u;
    in.first_vertex = it->second.first;
    in.vertex_count = it->second.second;
    out.push_back(in);
  }
}

void describeTables(const Scene& scene, gsp_scene_desc& d) {
  d.diffuse_bsdfs = scene.diffuseBSDFs.data();
  d.smooth_dielectric_bsdfs = scene.smoothDielectricBSDFs.data();
  d.smooth_conductor_bsdfs = scene.smoothConductorBSDFs.data();
  d.smooth_plastic_bsdfs = scene.smoothPlasticBSDFs.data();
  d.rough_conductor_bsdfs = scene.roughConductorBSDFs.data();
  d.smooth_floor_bsdfs = scene.smoothFloorBSDFs.data();
  d.rough_floor_bsdfs = scene.roughFloorBSDFs.data();
  d.rough_plastic_bsdfs = scene.roughPlasticBSDFs.data();
  d.num_bsdfs[GSP_BSDF_DIFFUSE] = (uint32_t)scene.diffuseBSDFs.size();
  d.num_bsdfs[GSP_BSDF_SMOOTH_DIELECTRIC] = (uint32_t)scene.smoothDielectricBSDFs.size();
  d.num_bsdfs[GSP_BSDF_SMOOTH_CONDUCTOR] = (uint32_t)scene.smoothConductorBSDFs.size();
  d.num_bsdfs[GSP_BSDF_SMOOTH_PLASTIC] = (uint32_t)scene.smoothPlasticBSDFs.size();
  d.num_bsdfs[GSP_BSDF_ROUGH_CONDUCTOR] = (uint32_t)scene.roughConductorBSDFs.size();
  d.num_bsdfs[GSP_BSDF_SMOOTH_FLOOR] = (uint32_t)scene.smoothFloorBSDFs.size();
  d.num_bsdfs[GSP_BSDF_ROUGH_FLOOR] = (uint32_t)scene.roughFloorBSDFs.size();
  d.num_bsdfs[GSP_BSDF_ROUGH_PLASTIC] = (uint32_t)scene.roughPlasticBSDFs.size();
  d.lights = scene.triangleLights.data();
  d.num_lights = (uint32_t)scene.triangleLights.size();
  std::memcpy(d.camera.to_world, scene.camera.getToWorld().data(), sizeof(d.camera.to_world));
  d.camera.fov = scene.camera.getFov();
}

void flattenScene(const Scene& scene, FlatScene& out) {
  out = FlatScene{};
  flattenInstances(scene, out.instances);  // (vertex ranges in order of first use, as below)
  std::unordered_map<const Mesh*, bool> placed;
  for (const RenderObject& obj : scene.renderObjects) {
    const Mesh* m = obj.mesh.get();
    if (!placed.emplace(m, true).second) continue;
    const auto& vs = m->getVertices();
    const uint32_t count = (uint32_t)(vs.size() / 3 * 3);
    for (uint32_t i = 0; i < count; ++i) {
      out.positions.insert(out.positions.end(), {vs[i].pos.x, vs[i].pos.y, vs[i].pos.z});
      out.normals.insert(out.normals.end(), {vs[i].normal.x, vs[i].normal.y, vs[i].normal.z});
      if (!scene.textures.empty()) out.uvs.insert(out.uvs.end(), {vs[i].uv.x, vs[i].uv.y});
    }
  }
  gsp_scene_desc& d = out.desc;
  d.instances = out.instances.data();
  d.num_instances = (uint32_t)out.instances.size();
  d.positions = out.positions.data();
  d.normals = out.normals.data();
  d.num_vertices = out.positions.size() / 3;
  describeTables(scene, d);
  // ---- dormant features ----
  if (!scene.textures.empty()) {
    for (const Texture& t : scene.textures) {
      out.textures.push_back(gsp_texture{t.width, t.height, (uint64_t)out.texels.size()});
      out.texels.insert(out.texels.end(), t.texels.begin(), t.texels.end());
    }
    out.texelDecode.resize(256);
    for (int b = 0; b < 256; ++b) {
      const double c = b / 255.0;  // IEC 61966-2-1 sRGB -> linear; the table is DATA for the renderer and the oracle alike
      out.texelDecode[b] = scene.srgbTextures ? (float)(c <= 0.04045 ? c / 12.92 : std::pow((c + 0.055) / 1.055, 2.4)) : (float)b / 255.0f;
    }
    d.uvs = out.uvs.data();
    d.textures = out.textures.data();
    d.num_textures = (uint32_t)out.textures.size();
    d.texels = out.texels.data();
    d.num_texels = out.texels.size();
    d.texel_decode = out.texelDecode.data();
  }
  if (scene.hasEnvMap) {
    out.envTexels = scene.envMap.texels;
    d.envmap.texels = out.envTexels.data();
    d.envmap.width = scene.envMap.width;
    d.envmap.height = scene.envMap.height;
    const mat4 inv = inverse(scene.envMap.transform);
    std::memcpy(d.envmap.to_local, inv.data(), sizeof(d.envmap.to_local));
  }
}

mat4 inverse(const mat4& a) {
  // Gauss-Jordan with partial pivoting in double precision; a singular matrix yields the identity
  double m[4][8];
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      m[r][c] = a[c][r];
      m[r][4 + c] = r == c ? 1.0 : 0.0;
    }
  for (int col = 0; col < 4; ++col) {
    int piv = col;
    for (int r = col + 1; r < 4; ++r)
      if (std::fabs(m[r][col]) > std::fabs(m[piv][col])) piv = r;
    if (std::fabs(m[piv][col]) < 1e-30) return mat4::identity();
    for (int c = 0; c < 8; ++c) std::swap(m[col][c], m[piv][c]);
    const double inv = 1.0 / m[col][col];
    for (int c = 0; c < 8; ++c) m[col][c] *= inv;
    for (int r = 0; r < 4; ++r)
      if (r != col) {
        const double f = m[r][col];
        for (int c = 0; c < 8; ++c) m[r][c] -= f * m[col][c];
      }
  }
  mat4 out;
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) out[c][r] = (float)m[r][4 + c];
  return out;
}

}  // namespace GPUSpectral
