// Image.cpp -- see Image.h.  Formats follow their public specifications: RFC 1950 / 1951 (zlib, deflate), the PNG
// specification (W3C, 2nd ed.), ITU-T T.81 (JPEG) + JFIF 1.02, the PFM convention (P. Debevec) and the Radiance
// picture format (G. Ward, "Real Pixels", Graphics Gems II).
#include "Image.h"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>

namespace GPUSpectral {

namespace {

[[noreturn]] void fail(const std::string& what) { throw std::runtime_error(what); }

std::vector<uint8_t> readFile(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) fail("cannot open image " + path);
  std::vector<uint8_t> data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  if (data.empty()) fail("empty image file " + path);
  return data;
}

uint32_t rgba(uint32_t r, uint32_t g, uint32_t b) { return r | (g << 8) | (b << 16) | 0xff000000u; }

// ---------------------------------------------------------------------------------------------------------------
// deflate (RFC 1951): LSB-first bit stream, canonical Huffman codes decoded length by length
// ---------------------------------------------------------------------------------------------------------------
struct LsbBits {
  const uint8_t* p;
  size_t n, pos = 0;
  uint32_t acc = 0;
  int have = 0;
  uint32_t take(int k) {
    while (have < k) {
      if (pos >= n) fail("deflate: stream ends early");
      acc |= (uint32_t)p[pos++] << have;
      have += 8;
    }
    const uint32_t v = k == 32 ? acc : (acc & ((1u << k) - 1u));
    acc >>= k;
    have -= k;
    return v;
  }
  void alignByte() {
    acc = 0;
    have = 0;
  }
};

// canonical code: how many codes of each length, and the symbols in code order
struct CodeBook {
  uint16_t perLength[17] = {};  // deflate codes are at most 15 bits long, JPEG codes 16
  std::vector<uint16_t> symbols;
  void build(const uint8_t* lengths, int n) {
    std::fill(perLength, perLength + 17, (uint16_t)0);
    for (int s = 0; s < n; ++s) ++perLength[lengths[s]];
    perLength[0] = 0;
    uint16_t start[17] = {};
    for (int l = 1; l < 17; ++l) start[l] = (uint16_t)(start[l - 1] + perLength[l - 1]);
    symbols.assign((size_t)n, 0);
    for (int s = 0; s < n; ++s)
      if (lengths[s]) symbols[start[lengths[s]]++] = (uint16_t)s;
  }
  template <class Bits>
  int decode(Bits& in) const {
    int code = 0, first = 0, index = 0;
    for (int l = 1; l < 17; ++l) {
      code |= (int)in.take(1);
      const int c = perLength[l];
      if (code - c < first) return symbols[(size_t)(index + (code - first))];
      index += c;
      first = (first + c) << 1;
      code <<= 1;
    }
    fail("bad Huffman code");
  }
};

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

void inflateBlock(LsbBits& in, const CodeBook& lit, const CodeBook& dist, std::vector<uint8_t>& out, size_t limit) {
  for (;;) {
    if (out.size() > limit) fail("deflate: more data than the image holds");
    const int s = lit.decode(in);
    if (s < 256) {
      out.push_back((uint8_t)s);
    } else if (s == 256) {
      return;
    } else {
      if (s > 285) fail("deflate: bad length symbol");
      const size_t len = kLenBase[s - 257] + in.take(kLenExtra[s - 257]);
      const int ds = dist.decode(in);
      if (ds > 29) fail("deflate: bad distance symbol");
      const size_t d = kDistBase[ds] + in.take(kDistExtra[ds]);
      if (d > out.size()) fail("deflate: distance beyond the output");
      const size_t from = out.size() - d;
      for (size_t k = 0; k < len; ++k) out.push_back(out[from + k]);
    }
  }
}

std::vector<uint8_t> inflateRaw(const uint8_t* data, size_t size, size_t expected) {
  LsbBits in{data, size};
  std::vector<uint8_t> out;
  out.reserve(std::min(expected, size * 1032 + 1024));  // deflate expands at most 1032 : 1
  const size_t limit = expected + 1024;  // a stream that inflates to more than the caller can use is rejected, not stored
  CodeBook fixedLit, fixedDist;
  {
    uint8_t l[288];
    for (int s = 0; s < 288; ++s) l[s] = s < 144 ? 8 : (s < 256 ? 9 : (s < 280 ? 7 : 8));
    fixedLit.build(l, 288);
    uint8_t d[30];
    std::fill(d, d + 30, (uint8_t)5);
    fixedDist.build(d, 30);
  }
  bool last = false;
  while (!last) {
    last = in.take(1) != 0;
    const uint32_t type = in.take(2);
    if (type == 0) {
      in.alignByte();
      if (in.pos + 4 > in.n) fail("deflate: stored block header cut");
      const uint32_t len = in.p[in.pos] | (in.p[in.pos + 1] << 8), nlen = in.p[in.pos + 2] | (in.p[in.pos + 3] << 8);
      in.pos += 4;
      if ((len ^ 0xffffu) != nlen || in.pos + len > in.n || out.size() + len > limit) fail("deflate: bad stored block");
      out.insert(out.end(), in.p + in.pos, in.p + in.pos + len);
      in.pos += len;
    } else if (type == 1) {
      inflateBlock(in, fixedLit, fixedDist, out, limit);
    } else if (type == 2) {
      const int nlit = (int)in.take(5) + 257, ndist = (int)in.take(5) + 1, ncode = (int)in.take(4) + 4;
      static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
      uint8_t cl[19] = {};
      for (int k = 0; k < ncode; ++k) cl[order[k]] = (uint8_t)in.take(3);
      CodeBook lengthsCode;
      lengthsCode.build(cl, 19);
      uint8_t lengths[288 + 32] = {};
      int k = 0;
      while (k < nlit + ndist) {
        const int s = lengthsCode.decode(in);
        if (s < 16) {
          lengths[k++] = (uint8_t)s;
        } else {
          int rep;
          uint8_t v = 0;
          if (s == 16) {
            if (k == 0) fail("deflate: repeat with no previous length");
            v = lengths[k - 1];
            rep = 3 + (int)in.take(2);
          } else if (s == 17) {
            rep = 3 + (int)in.take(3);
          } else {
            rep = 11 + (int)in.take(7);
          }
          if (k + rep > nlit + ndist) fail("deflate: too many code lengths");
          while (rep--) lengths[k++] = v;
        }
      }
      CodeBook lit, dist;
      lit.build(lengths, nlit);
      dist.build(lengths + nlit, ndist);
      inflateBlock(in, lit, dist, out, limit);
    } else {
      fail("deflate: reserved block type");
    }
  }
  return out;
}

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
uint32_t be16(const uint8_t* p) { return ((uint32_t)p[0] << 8) | p[1]; }

}  // namespace

std::vector<uint8_t> inflateZlib(const uint8_t* data, size_t size, size_t expected) {
  if (size < 6) fail("zlib: stream too short");
  if ((data[0] & 0x0f) != 8 || ((data[0] << 8) | data[1]) % 31 != 0 || (data[1] & 0x20)) fail("zlib: bad header");
  return inflateRaw(data + 2, size - 2, expected);  // the Adler-32 trailer is not checked
}

// ---------------------------------------------------------------------------------------------------------------
// PNG
// ---------------------------------------------------------------------------------------------------------------
Image8 decodePng(const uint8_t* data, size_t size) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (size < 8 || std::memcmp(data, sig, 8) != 0) fail("png: bad signature");
  uint32_t w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
  std::vector<uint8_t> idat, palette;
  size_t pos = 8;
  bool end = false;
  while (!end) {
    if (pos + 12 > size) fail("png: chunk cut");
    const uint32_t len = be32(data + pos);
    const uint8_t* type = data + pos + 4;
    const uint8_t* body = data + pos + 8;
    if (len > size || pos + 12 + len > size) fail("png: chunk beyond the file");
    if (!std::memcmp(type, "IHDR", 4)) {
      if (len < 13) fail("png: short IHDR");
      w = be32(body);
      h = be32(body + 4);
      depth = body[8];
      ctype = body[9];
      interlace = body[12];
    } else if (!std::memcmp(type, "PLTE", 4)) {
      palette.assign(body, body + len);
    } else if (!std::memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), body, body + len);
    } else if (!std::memcmp(type, "IEND", 4)) {
      end = true;
    }
    pos += 12 + (size_t)len;
  }
  if (w == 0 || h == 0 || w > (1u << 15) || h > (1u << 15)) fail("png: size 0 or above 32768");
  if (interlace) fail("png: interlaced files are not supported");
  int channels;
  switch (ctype) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: fail("png: bad colour type");
  }
  if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4)))) fail("png: bad bit depth");
  if (ctype == 3 && depth == 16) fail("png: bad bit depth");
  const size_t stride = ((size_t)w * channels * depth + 7) / 8;
  const size_t bpp = std::max<size_t>(1, (size_t)channels * depth / 8);
  if ((stride + 1) * h > idat.size() * 1032 + 1024) fail("png: image data cut");  // (no allocation from a forged header)
  std::vector<uint8_t> raw = inflateZlib(idat.data(), idat.size(), (stride + 1) * h);
  if (raw.size() < (stride + 1) * h) fail("png: image data cut");
  // undo the scanline filters in place (PNG 9.2)
  std::vector<uint8_t> zero(stride, 0);
  for (uint32_t y = 0; y < h; ++y) {
    uint8_t* cur = raw.data() + (stride + 1) * y + 1;
    const uint8_t* up = y ? raw.data() + (stride + 1) * (y - 1) + 1 : zero.data();
    const int filter = cur[-1];
    for (size_t x = 0; x < stride; ++x) {
      const int a = x >= bpp ? cur[x - bpp] : 0, b = up[x], c = x >= bpp ? up[x - bpp] : 0;
      int pred = 0;
      switch (filter) {
        case 0: break;
        case 1: pred = a; break;
        case 2: pred = b; break;
        case 3: pred = (a + b) / 2; break;
        case 4: {
          const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
          pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
        } break;
        default: fail("png: bad filter type");
      }
      cur[x] = (uint8_t)(cur[x] + pred);
    }
  }
  Image8 img;
  img.width = w;
  img.height = h;
  img.texels.resize((size_t)w * h);
  auto sample = [&](const uint8_t* row, size_t index) -> uint32_t {  // sample `index` of the row, scaled to 8 bits
    if (depth == 8) return row[index];
    if (depth == 16) return row[2 * index];
    const uint32_t per = 8 / depth, v = (row[index / per] >> (8 - depth * (index % per + 1))) & ((1u << depth) - 1u);
    return ctype == 3 ? v : v * 255u / ((1u << depth) - 1u);
  };
  for (uint32_t y = 0; y < h; ++y) {
    const uint8_t* row = raw.data() + (stride + 1) * y + 1;
    uint32_t* out = img.texels.data() + (size_t)(h - 1 - y) * w;  // last image row first (Loader.cpp:74)
    for (uint32_t x = 0; x < w; ++x) {
      if (ctype == 3) {
        const uint32_t i = sample(row, x);
        if (3 * (size_t)i + 2 >= palette.size()) fail("png: palette index out of range");
        out[x] = rgba(palette[3 * i], palette[3 * i + 1], palette[3 * i + 2]);
      } else if (channels <= 2) {
        const uint32_t g = sample(row, (size_t)x * channels);
        out[x] = rgba(g, g, g);
      } else {
        out[x] = rgba(sample(row, (size_t)x * channels), sample(row, (size_t)x * channels + 1), sample(row, (size_t)x * channels + 2));
      }
    }
  }
  return img;
}

// ---------------------------------------------------------------------------------------------------------------
// JPEG: sequential Huffman (SOF0 / SOF1), 8-bit samples
// ---------------------------------------------------------------------------------------------------------------
namespace {

struct MsbBits {  // entropy-coded segment: MSB first, 0xFF00 = a data byte 0xFF, any other 0xFFxx ends the segment
  const uint8_t* p;
  size_t n, pos;
  uint32_t acc = 0;
  int have = 0;
  bool hitMarker = false;
  uint32_t take(int k) {
    if (k == 0) return 0;
    while (have < k) {
      uint32_t byte = 0;
      if (!hitMarker && pos < n) {
        byte = p[pos];
        if (byte == 0xff) {
          if (pos + 1 < n && p[pos + 1] == 0x00) {
            pos += 2;
          } else {
            hitMarker = true;  // feed zeros from here on (T.81 F.2.2.5)
            byte = 0;
          }
        } else {
          ++pos;
        }
      }
      acc = (acc << 8) | byte;
      have += 8;
    }
    const uint32_t v = (acc >> (have - k)) & ((1u << k) - 1u);
    have -= k;
    return v;
  }
  void restart() {  // drop the bits left over before an RSTn marker and step over it
    acc = 0;
    have = 0;
    hitMarker = false;
    while (pos + 1 < n && !(p[pos] == 0xff && p[pos + 1] >= 0xd0 && p[pos + 1] <= 0xd7)) ++pos;
    if (pos + 1 < n) pos += 2;
  }
};

struct JpegComponent {
  int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
  int pred = 0;
  int planeW = 0, planeH = 0;
  std::vector<uint8_t> plane;
};

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

int extendSign(uint32_t bits, int size) {  // T.81 F.2.2.1
  return size && bits < (1u << (size - 1)) ? (int)bits - (1 << size) + 1 : (int)bits;
}

// 8x8 inverse DCT (T.81 A.3.3), separable, double precision; output level-shifted and clamped
void inverseDct(const int coef[64], const uint16_t quant[64], uint8_t* out, int stride) {
  struct Basis {  // built once, by the first caller, under the language's own guard (scenes may load on several threads)
    double b[8][8];
    Basis() {
      for (int x = 0; x < 8; ++x)
        for (int u = 0; u < 8; ++u) b[x][u] = (u == 0 ? std::sqrt(0.5) : 1.0) * std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0) * 0.5;
    }
  };
  static const Basis table;
  const double (&basis)[8][8] = table.b;
  double f[64], tmp[64];
  for (int k = 0; k < 64; ++k) f[k] = (double)coef[k] * quant[k];
  for (int v = 0; v < 8; ++v)  // rows: over u
    for (int x = 0; x < 8; ++x) {
      double s = 0;
      for (int u = 0; u < 8; ++u) s += basis[x][u] * f[8 * v + u];
      tmp[8 * v + x] = s;
    }
  for (int x = 0; x < 8; ++x)  // columns: over v
    for (int y = 0; y < 8; ++y) {
      double s = 0;
      for (int v = 0; v < 8; ++v) s += basis[y][v] * tmp[8 * v + x];
      const double px = std::floor(s + 128.5);
      out[y * stride + x] = (uint8_t)(px < 0 ? 0 : (px > 255 ? 255 : px));
    }
}

}  // namespace

Image8 decodeJpeg(const uint8_t* data, size_t size) {
  if (size < 4 || data[0] != 0xff || data[1] != 0xd8) fail("jpeg: no SOI marker");
  uint16_t quant[4][64] = {};
  CodeBook dc[4], ac[4];
  bool haveDc[4] = {}, haveAc[4] = {};
  std::vector<JpegComponent> comps;
  uint32_t w = 0, h = 0;
  int restartInterval = 0, adobeTransform = -1;
  size_t pos = 2;
  bool decoded = false;
  while (!decoded) {
    while (pos < size && data[pos] != 0xff) ++pos;
    while (pos < size && data[pos] == 0xff) ++pos;
    if (pos >= size) fail("jpeg: no image data");
    const uint8_t marker = data[pos++];
    if (marker == 0xd9) fail("jpeg: EOI before any scan");
    if (marker == 0x01 || (marker >= 0xd0 && marker <= 0xd7)) continue;
    if (pos + 2 > size) fail("jpeg: segment cut");
    const size_t len = be16(data + pos);
    if (len < 2 || pos + len > size) fail("jpeg: segment beyond the file");
    const uint8_t* seg = data + pos + 2;
    const size_t n = len - 2;
    if (marker == 0xdb) {  // DQT
      size_t k = 0;
      while (k < n) {
        const int pq = seg[k] >> 4, tq = seg[k] & 15;
        ++k;
        if (tq > 3 || k + (pq ? 128 : 64) > n) fail("jpeg: bad DQT");
        for (int i = 0; i < 64; ++i) {
          quant[tq][kZigzag[i]] = (uint16_t)(pq ? be16(seg + k + 2 * i) : seg[k + i]);
        }
        k += pq ? 128 : 64;
      }
    } else if (marker == 0xc4) {  // DHT
      size_t k = 0;
      while (k < n) {
        if (k + 17 > n) fail("jpeg: bad DHT");
        const int tc = seg[k] >> 4, th = seg[k] & 15;
        if (tc > 1 || th > 3) fail("jpeg: bad DHT table id");
        int total = 0;
        for (int l = 1; l <= 16; ++l) total += seg[k + l];
        if (total > 256 || k + 17 + total > n) fail("jpeg: bad DHT");
        CodeBook& cb = tc ? ac[th] : dc[th];  // BITS = codes per length, HUFFVAL = the symbols in code order (T.81 B.2.4.2)
        cb.perLength[0] = 0;
        for (int l = 1; l <= 16; ++l) cb.perLength[l] = seg[k + l];
        cb.symbols.assign(seg + k + 17, seg + k + 17 + total);
        (tc ? haveAc[th] : haveDc[th]) = true;
        k += 17 + (size_t)total;
      }
    } else if (marker == 0xc0 || marker == 0xc1) {  // SOF0 / SOF1
      if (n < 6) fail("jpeg: bad SOF");
      if (seg[0] != 8) fail("jpeg: only 8-bit samples are supported");
      h = be16(seg + 1);
      w = be16(seg + 3);
      const int nc = seg[5];
      if ((nc != 1 && nc != 3) || n < 6 + 3 * (size_t)nc) fail("jpeg: only 1- or 3-component images are supported");
      comps.resize((size_t)nc);
      for (int c = 0; c < nc; ++c) {
        comps[c].id = seg[6 + 3 * c];
        comps[c].h = seg[7 + 3 * c] >> 4;
        comps[c].v = seg[7 + 3 * c] & 15;
        comps[c].tq = seg[8 + 3 * c];
        if (comps[c].h < 1 || comps[c].h > 4 || comps[c].v < 1 || comps[c].v > 4 || comps[c].tq > 3) fail("jpeg: bad component");
      }
    } else if (marker == 0xc2 || (marker >= 0xc5 && marker <= 0xcf && marker != 0xc8 && marker != 0xcc)) {
      fail("jpeg: progressive / lossless / arithmetic-coded files are not supported");
    } else if (marker == 0xdd) {  // DRI
      if (n < 2) fail("jpeg: bad DRI");
      restartInterval = (int)be16(seg);
    } else if (marker == 0xee && n >= 12 && !std::memcmp(seg, "Adobe", 5)) {
      adobeTransform = seg[11];
    } else if (marker == 0xda) {  // SOS: decode the (single, interleaved) scan
      if (comps.empty() || w == 0 || h == 0) fail("jpeg: SOS before SOF");
      if (w > (1u << 15) || h > (1u << 15)) fail("jpeg: size above 32768");
      const int ns = seg[0];
      if (ns != (int)comps.size() || n < 1 + 2 * (size_t)ns + 3) fail("jpeg: only one interleaved scan is supported");
      for (int s = 0; s < ns; ++s) {
        JpegComponent* c = nullptr;
        for (auto& cc : comps)
          if (cc.id == seg[1 + 2 * s]) c = &cc;
        if (!c) fail("jpeg: scan names an unknown component");
        c->td = seg[2 + 2 * s] >> 4;
        c->ta = seg[2 + 2 * s] & 15;
        if (c->td > 3 || c->ta > 3 || !haveDc[c->td] || !haveAc[c->ta]) fail("jpeg: scan uses a missing Huffman table");
      }
      int hmax = 1, vmax = 1;
      for (auto& c : comps) {
        hmax = std::max(hmax, c.h);
        vmax = std::max(vmax, c.v);
      }
      if (comps.size() == 1) comps[0].h = comps[0].v = hmax = vmax = 1;  // a single component is never interleaved (A.2.2)
      const int mcuW = 8 * hmax, mcuH = 8 * vmax;
      const int mcusX = ((int)w + mcuW - 1) / mcuW, mcusY = ((int)h + mcuH - 1) / mcuH;
      {  // a block costs at least two bits of entropy-coded data (DC size 0 + EOB): no allocation from a forged header
        size_t blocks = 0;
        for (auto& c : comps) blocks += (size_t)mcusX * mcusY * c.h * c.v;
        if (size - (pos + len) < blocks / 4) fail("jpeg: image data cut");
      }
      for (auto& c : comps) {
        c.planeW = mcusX * c.h * 8;
        c.planeH = mcusY * c.v * 8;
        c.plane.assign((size_t)c.planeW * c.planeH, 0);
        c.pred = 0;
      }
      MsbBits in{data, size, pos + len};
      int untilRestart = restartInterval;
      for (int my = 0; my < mcusY; ++my)
        for (int mx = 0; mx < mcusX; ++mx) {
          if (restartInterval && untilRestart == 0) {
            in.restart();
            for (auto& c : comps) c.pred = 0;
            untilRestart = restartInterval;
          }
          for (auto& c : comps)
            for (int by = 0; by < c.v; ++by)
              for (int bx = 0; bx < c.h; ++bx) {
                int coef[64] = {};
                const int t = dc[c.td].decode(in);  // F.2.2.1
                if (t > 11) fail("jpeg: bad DC size");
                c.pred += extendSign(in.take(t), t);
                // 8-bit samples: |DC| < 2^11 after quantisation (T.81 F.1.2.1); a forged stream must not walk the
                // predictor towards INT_MAX one difference at a time
                if (c.pred < -32768 || c.pred > 32767) fail("jpeg: DC out of range");
                coef[0] = c.pred;
                for (int k = 1; k < 64;) {  // F.2.2.2
                  const int rs = ac[c.ta].decode(in);
                  const int run = rs >> 4, sz = rs & 15;
                  if (sz == 0) {
                    if (run != 15) break;  // EOB
                    k += 16;
                    continue;
                  }
                  k += run;
                  if (k > 63) fail("jpeg: AC run beyond the block");
                  coef[kZigzag[k]] = extendSign(in.take(sz), sz);
                  ++k;
                }
                inverseDct(coef, quant[c.tq], c.plane.data() + (size_t)(my * c.v + by) * 8 * c.planeW + (size_t)(mx * c.h + bx) * 8,
                           c.planeW);
              }
          --untilRestart;
        }
      // ---- up-sample to full resolution: the triangle filter of libjpeg's "fancy upsampling" for the 2:1 horizontal
      // (h2v1) and 2:1 x 2:1 (h2v2) cases that 4:2:2 / 4:2:0 files use, replication otherwise ----
      std::vector<std::vector<uint8_t>> full(comps.size());
      for (size_t ci = 0; ci < comps.size(); ++ci) {
        const JpegComponent& cc = comps[ci];
        std::vector<uint8_t>& out = full[ci];
        out.resize((size_t)w * h);
        const int fx = hmax / cc.h, fy = vmax / cc.v;
        const bool exact = hmax % cc.h == 0 && vmax % cc.v == 0;
        const int dw = ((int)w * cc.h + hmax - 1) / hmax, dh = ((int)h * cc.v + vmax - 1) / vmax;  // down-sampled size
        auto in = [&](int x, int y) { return (int)cc.plane[(size_t)std::min(std::max(y, 0), dh - 1) * cc.planeW + std::min(std::max(x, 0), dw - 1)]; };
        if (exact && fx == 1 && fy == 1) {
          for (uint32_t y = 0; y < h; ++y)
            for (uint32_t x = 0; x < w; ++x) out[(size_t)y * w + x] = (uint8_t)in((int)x, (int)y);
        } else if (exact && fx == 2 && fy == 1) {
          for (uint32_t y = 0; y < h; ++y)
            for (uint32_t x = 0; x < w; ++x) {
              const int i = (int)x / 2;
              int v;
              if ((x & 1u) == 0) v = i == 0 ? in(0, (int)y) : (3 * in(i, (int)y) + in(i - 1, (int)y) + 1) >> 2;
              else v = i == dw - 1 ? in(i, (int)y) : (3 * in(i, (int)y) + in(i + 1, (int)y) + 2) >> 2;
              out[(size_t)y * w + x] = (uint8_t)v;
            }
        } else if (exact && fx == 2 && fy == 2) {
          for (uint32_t y = 0; y < h; ++y) {
            const int j = (int)y / 2, far = (y & 1u) ? j + 1 : j - 1;
            auto col = [&](int i) { return 3 * in(i, j) + in(i, far); };  // vertical pass, scaled by 4
            for (uint32_t x = 0; x < w; ++x) {
              const int i = (int)x / 2;
              int v;
              if ((x & 1u) == 0) v = i == 0 ? (col(0) * 4 + 8) >> 4 : (3 * col(i) + col(i - 1) + 8) >> 4;
              else v = i == dw - 1 ? (col(i) * 4 + 7) >> 4 : (3 * col(i) + col(i + 1) + 7) >> 4;
              out[(size_t)y * w + x] = (uint8_t)v;
            }
          }
        } else {
          for (uint32_t y = 0; y < h; ++y)
            for (uint32_t x = 0; x < w; ++x) out[(size_t)y * w + x] = (uint8_t)in((int)x * cc.h / hmax, (int)y * cc.v / vmax);
        }
      }
      // ---- colour convert (JFIF: YCbCr, full range), flip ----
      Image8 img;
      img.width = w;
      img.height = h;
      img.texels.resize((size_t)w * h);
      const bool ycc = comps.size() == 3 && adobeTransform != 0;
      for (uint32_t y = 0; y < h; ++y) {
        uint32_t* out = img.texels.data() + (size_t)(h - 1 - y) * w;
        for (uint32_t x = 0; x < w; ++x) {
          int s[3] = {0, 0, 0};
          for (size_t c = 0; c < comps.size(); ++c) s[c] = full[c][(size_t)y * w + x];
          if (comps.size() == 1) {
            out[x] = rgba((uint32_t)s[0], (uint32_t)s[0], (uint32_t)s[0]);
          } else if (ycc) {
            const double Y = s[0], cb = s[1] - 128.0, cr = s[2] - 128.0;
            auto clamp8 = [](double v) { return (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : std::floor(v + 0.5))); };
            out[x] = rgba(clamp8(Y + 1.402 * cr), clamp8(Y - 0.344136 * cb - 0.714136 * cr), clamp8(Y + 1.772 * cb));
          } else {
            out[x] = rgba((uint32_t)s[0], (uint32_t)s[1], (uint32_t)s[2]);
          }
        }
      }
      return img;
    }
    pos += len;
  }
  fail("jpeg: no scan");
}

// ---------------------------------------------------------------------------------------------------------------
// PFM, Radiance RGBE
// ---------------------------------------------------------------------------------------------------------------
namespace {
std::string headerToken(const uint8_t* data, size_t size, size_t& pos) {
  while (pos < size && std::isspace(data[pos])) ++pos;
  std::string t;
  while (pos < size && !std::isspace(data[pos])) t.push_back((char)data[pos++]);
  return t;
}
}  // namespace

ImageF decodePfm(const uint8_t* data, size_t size) {
  size_t pos = 0;
  const std::string magic = headerToken(data, size, pos);
  if (magic != "PF" && magic != "Pf") fail("pfm: bad magic");
  const int channels = magic == "PF" ? 3 : 1;
  const long w = std::atol(headerToken(data, size, pos).c_str()), h = std::atol(headerToken(data, size, pos).c_str());
  const double scale = std::atof(headerToken(data, size, pos).c_str());
  ++pos;  // the single whitespace byte behind the scale
  if (w <= 0 || h <= 0 || w > (1 << 15) || h > (1 << 15) || scale == 0.0) fail("pfm: bad header");
  const size_t count = (size_t)w * h * channels;
  if (pos + 4 * count > size) fail("pfm: pixel data cut");
  ImageF img;
  img.width = (uint32_t)w;
  img.height = (uint32_t)h;
  img.texels.resize((size_t)w * h * 4);
  const bool little = scale < 0.0;
  for (size_t i = 0; i < (size_t)w * h; ++i) {  // PFM rows run bottom to top already
    float px[3];
    for (int c = 0; c < channels; ++c) {
      const uint8_t* b = data + pos + 4 * (i * channels + c);
      const uint32_t bits = little ? ((uint32_t)b[0] | (b[1] << 8) | (b[2] << 16) | ((uint32_t)b[3] << 24))
                                   : ((uint32_t)b[3] | (b[2] << 8) | (b[1] << 16) | ((uint32_t)b[0] << 24));
      std::memcpy(&px[c], &bits, 4);
    }
    if (channels == 1) px[1] = px[2] = px[0];
    img.texels[4 * i] = px[0];
    img.texels[4 * i + 1] = px[1];
    img.texels[4 * i + 2] = px[2];
    img.texels[4 * i + 3] = 1.0f;
  }
  return img;
}

ImageF decodeRgbe(const uint8_t* data, size_t size) {
  size_t pos = 0;
  auto line = [&]() {
    std::string s;
    while (pos < size && data[pos] != '\n') s.push_back((char)data[pos++]);
    ++pos;
    return s;
  };
  const std::string magic = line();
  if (magic.rfind("#?", 0) != 0) fail("hdr: bad magic");
  bool rgbe = false;
  for (;;) {
    if (pos >= size) fail("hdr: header cut");
    const std::string s = line();
    if (s.empty()) break;
    if (s.rfind("FORMAT=", 0) == 0) rgbe = s == "FORMAT=32-bit_rle_rgbe";
  }
  if (!rgbe) fail("hdr: only FORMAT=32-bit_rle_rgbe is supported");
  long w = 0, h = 0;
  {
    const std::string res = line();
    char sy = 0, sx = 0;
    if (std::sscanf(res.c_str(), "%cY %ld %cX %ld", &sy, &h, &sx, &w) != 4 || sy != '-' || sx != '+') fail("hdr: only '-Y h +X w' orientation is supported");
  }
  if (w <= 0 || h <= 0 || w > (1 << 15) || h > (1 << 15)) fail("hdr: bad size");
  if (size - std::min(pos, size) < (size_t)h * (4 + (size_t)w / 16)) fail("hdr: pixel data cut");  // shortest run-length coding
  ImageF img;
  img.width = (uint32_t)w;
  img.height = (uint32_t)h;
  img.texels.resize((size_t)w * h * 4);
  std::vector<uint8_t> scan((size_t)w * 4);
  for (long y = 0; y < h; ++y) {
    if (pos + 4 > size) fail("hdr: pixel data cut");
    if (w >= 8 && w < 32768 && data[pos] == 2 && data[pos + 1] == 2 && (((long)data[pos + 2] << 8) | data[pos + 3]) == w) {
      pos += 4;  // adaptive run-length encoding, one channel after the other
      for (int c = 0; c < 4; ++c) {
        long x = 0;
        while (x < w) {
          if (pos >= size) fail("hdr: pixel data cut");
          int count = data[pos++];
          if (count > 128) {
            count -= 128;
            if (pos >= size || x + count > w) fail("hdr: bad run");
            const uint8_t v = data[pos++];
            while (count--) scan[(size_t)4 * x++ + c] = v;
          } else {
            if (count == 0 || pos + (size_t)count > size || x + count > w) fail("hdr: bad run");
            while (count--) scan[(size_t)4 * x++ + c] = data[pos++];
          }
        }
      }
    } else {
      if (pos + (size_t)4 * w > size) fail("hdr: pixel data cut");
      std::memcpy(scan.data(), data + pos, (size_t)4 * w);
      pos += (size_t)4 * w;
    }
    float* out = img.texels.data() + (size_t)(h - 1 - y) * w * 4;  // last image row first (Loader.cpp:103)
    for (long x = 0; x < w; ++x) {
      const uint8_t* p = &scan[(size_t)4 * x];
      const float f = p[3] ? std::ldexp(1.0f, (int)p[3] - (128 + 8)) : 0.0f;
      out[4 * x] = p[0] * f;
      out[4 * x + 1] = p[1] * f;
      out[4 * x + 2] = p[2] * f;
      out[4 * x + 3] = 1.0f;
    }
  }
  return img;
}

Image8 loadBitmap(const std::string& path) {
  const std::vector<uint8_t> data = readFile(path);
  try {
    if (data.size() >= 2 && data[0] == 0xff && data[1] == 0xd8) return decodeJpeg(data.data(), data.size());
    if (data.size() >= 4 && data[0] == 0x89 && data[1] == 'P') return decodePng(data.data(), data.size());
  } catch (const std::runtime_error& e) {
    fail(path + ": " + e.what());
  }
  fail(path + ": not a PNG or JPEG file");
}

ImageF loadHdrBitmap(const std::string& path) {
  const std::vector<uint8_t> data = readFile(path);
  try {
    if (data.size() >= 2 && data[0] == 'P' && (data[1] == 'F' || data[1] == 'f')) return decodePfm(data.data(), data.size());
    if (data.size() >= 2 && data[0] == '#' && data[1] == '?') return decodeRgbe(data.data(), data.size());
  } catch (const std::runtime_error& e) {
    fail(path + ": " + e.what());
  }
  fail(path + ": not a PFM or Radiance .hdr file");
}

Image8 makeCheckerboard(uint32_t uSize, uint32_t vSize, const float color0[3], const float color1[3]) {
  if (uSize == 0 || vSize == 0 || uSize > 64 || vSize > 64) fail("checkerboard: uscale / vscale must be in [1, 64]");
  auto byte = [](float c) { return (uint32_t)std::floor(std::min(std::max(c, 0.0f), 1.0f) * 255.0f + 0.5f); };  // float32 throughout
  const uint32_t on = rgba(byte(color0[0]), byte(color0[1]), byte(color0[2]));
  const uint32_t off = rgba(byte(color1[0]), byte(color1[1]), byte(color1[2]));
  Image8 img;
  img.width = uSize * 100 * 2;   // Loader.cpp:132-133
  img.height = vSize * 100 * 2;
  img.texels.resize((size_t)img.width * img.height);
  for (uint32_t y = 0; y < img.height; ++y)
    for (uint32_t x = 0; x < img.width; ++x) img.texels[(size_t)y * img.width + x] = ((x / 100 + y / 100) & 1u) ? off : on;
  return img;
}

}  // namespace GPUSpectral
