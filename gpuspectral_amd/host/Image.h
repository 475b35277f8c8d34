// Image.h -- image file readers of the scene loader's dormant features.
//
// The reference reads bitmaps with stb_image (loadTexture, S/engine/Loader.cpp:66-86: any format stb knows, rows
// flipped so that the last image row comes first, RGBA8 with A = 0xFF) and environment maps with loadPfm / stbi_loadf
// (loadHdrTexture, :88-116, RGBA32F with A = 1, flipped the same way).  stb_image is a third-party header that is absent
// from the reference tree (external/ submodules are empty), so the formats the shipped scenes use are decoded here:
// PNG (8-bit grey / grey+alpha / RGB / RGBA / palette, non-interlaced) over an own inflate, baseline + extended
// sequential Huffman JPEG (8-bit, 1 or 3 components, any sampling factors, restart intervals), PFM (PF / Pf, either
// byte order) and Radiance RGBE (.hdr, flat or run-length encoded scanlines).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace GPUSpectral {

struct Image8 {
  uint32_t width = 0, height = 0;
  std::vector<uint32_t> texels;  // RGBA8, R in bits 0-7; row 0 = BOTTOM image row (Loader.cpp:74-81)
};
struct ImageF {
  uint32_t width = 0, height = 0;
  std::vector<float> texels;  // RGBA32F; row 0 = BOTTOM image row (Loader.cpp:103-110)
};

// Throw std::runtime_error (with the path) on unreadable or unsupported files, like the reference's loaders.
Image8 loadBitmap(const std::string& path);
ImageF loadHdrBitmap(const std::string& path);

// In-memory entry points (tests, fuzzing).
Image8 decodePng(const uint8_t* data, size_t size);
Image8 decodeJpeg(const uint8_t* data, size_t size);
ImageF decodePfm(const uint8_t* data, size_t size);
ImageF decodeRgbe(const uint8_t* data, size_t size);
// zlib stream (RFC 1950 wrapper around RFC 1951 deflate) -> bytes; `expected` = output size hint
std::vector<uint8_t> inflateZlib(const uint8_t* data, size_t size, size_t expected);

// The reference's dormant checkerboard (Loader.cpp:127-139): a (2*uSize*100) x (2*vSize*100) RGBA8 image of
// 2*uSize x 2*vSize cells alternating color0 / color1, cell (0, 0) at the bottom left = color0.
Image8 makeCheckerboard(uint32_t uSize, uint32_t vSize, const float color0[3], const float color1[3]);

}  // namespace GPUSpectral
