// Microbenchmark behind DESIGN.md's "bound by divergent vector-memory instructions" claim: how many random
// 64-B node fetches per second does one MI355X sustain when (A) every lane of a wave fetches its own node with
// 4 x 16-B loads (the k_trace layout) versus (B) 4 adjacent lanes share a node and load 16 B each, versus
// (C) like B plus a second 16-B load of a shared header (80-B node), (D) 8 lanes share a 128-B node.
// Dependent chain: the next node index comes from the loaded data, as in a traversal.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k_chase(const uint4* __restrict__ nodes, uint32_t mask, int iters, uint32_t* out) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63;
  constexpr int G = MODE == 0 ? 1 : ((MODE == 3 || MODE == 5) ? 8 : 4);  // lanes per node
  uint32_t ray = tid / G, sub = lane % G;
  uint32_t idx = (ray * 2654435761u) & mask;
  uint32_t acc = 0;
  for (int i = 0; i < iters; ++i) {
    uint32_t next;
    if (MODE == 0) {
      const uint4* n = nodes + 4ull * idx;
      uint4 a = n[0], b = n[1], c = n[2], d = n[3];
      next = a.x ^ b.y ^ c.z ^ d.w;
    } else if (MODE == 1) {
      uint4 a = nodes[4ull * idx + sub];
      uint32_t v = a.x ^ a.y;
      v ^= __shfl_xor(v, 1); v ^= __shfl_xor(v, 2);
      next = v;
    } else if (MODE == 2) {
      uint4 h = nodes[5ull * idx];
      uint4 a = nodes[5ull * idx + 1 + sub];
      uint32_t v = a.x ^ a.y;
      v ^= __shfl_xor(v, 1); v ^= __shfl_xor(v, 2);
      next = v ^ h.x;
    } else if (MODE == 4) {  // 64-B node: shared 16-B header + 12 B per lane
      const char* base = (const char*)(nodes + 4ull * idx);
      uint4 h = *(const uint4*)base;
      const uint32_t* c = (const uint32_t*)(base + 16 + 12 * sub);
      uint32_t v = c[0] ^ c[1] ^ c[2];
      v ^= __shfl_xor(v, 1); v ^= __shfl_xor(v, 2);
      next = v ^ h.x;
    } else if (MODE == 5) {  // 128-B node: shared 16-B header + 12 B per lane, 8 lanes
      const char* base = (const char*)(nodes + 8ull * idx);
      uint4 h = *(const uint4*)base;
      const uint32_t* c = (const uint32_t*)(base + 16 + 12 * sub);
      uint32_t v = c[0] ^ c[1] ^ c[2];
      v ^= __shfl_xor(v, 1); v ^= __shfl_xor(v, 2); v ^= __shfl_xor(v, 4);
      next = v ^ h.x;
    } else if (MODE == 6) {  // 128-B stride, quad, header + own 16 B (C without the 80-B stride)
      uint4 h = nodes[8ull * idx];
      uint4 a = nodes[8ull * idx + 1 + sub];
      uint32_t v = a.x ^ a.y;
      v ^= __shfl_xor(v, 1); v ^= __shfl_xor(v, 2);
      next = v ^ h.x;
    } else {
      uint4 a = nodes[8ull * idx + sub];
      uint32_t v = a.x ^ a.y;
      v ^= __shfl_xor(v, 1); v ^= __shfl_xor(v, 2); v ^= __shfl_xor(v, 4);
      next = v;
    }
    acc += next;
    idx = (next * 2246822519u + ray) & mask;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char** argv) {
  const int iters = 400;
  for (int lg = 14; lg <= 22; lg += 4) {  // 16K, 256K, 4M nodes  (1 MB, 16 MB, 256 MB at 64 B)
    const uint32_t N = 1u << lg;
    std::vector<uint32_t> h((size_t)N * 8 * 4);
    uint32_t s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s >> 3; }
    uint4* d; uint32_t* out;
    CHECK(hipMalloc(&d, h.size() * 4)); CHECK(hipMalloc(&out, 4));
    CHECK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int waves = 4; waves <= 8; waves += 4)
    for (int mode = 0; mode < 7; ++mode) {
      const int blocks = 256 * waves;  // 256-thread blocks: `waves` waves per SIMD
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        switch (mode) {
          case 0: hipLaunchKernelGGL(k_chase<0>, dim3(blocks), dim3(256), 0, 0, d, N - 1, iters, out); break;
          case 1: hipLaunchKernelGGL(k_chase<1>, dim3(blocks), dim3(256), 0, 0, d, N - 1, iters, out); break;
          case 2: hipLaunchKernelGGL(k_chase<2>, dim3(blocks), dim3(256), 0, 0, d, N - 1, iters, out); break;
          case 3: hipLaunchKernelGGL(k_chase<3>, dim3(blocks), dim3(256), 0, 0, d, N - 1, iters, out); break;
          case 4: hipLaunchKernelGGL(k_chase<4>, dim3(blocks), dim3(256), 0, 0, d, N - 1, iters, out); break;
          case 5: hipLaunchKernelGGL(k_chase<5>, dim3(blocks), dim3(256), 0, 0, d, N - 1, iters, out); break;
          case 6: hipLaunchKernelGGL(k_chase<6>, dim3(blocks), dim3(256), 0, 0, d, N - 1, iters, out); break;
        }
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
      }
      const int G = mode == 0 ? 1 : ((mode == 3 || mode == 5) ? 8 : 4);
      const double visits = (double)blocks * 256 / G * iters;
      const char* names[7] = {"A lane/node 4x16B", "B quad/node 1x16B", "C quad/node 2x16B (80B)", "D oct/node 1x16B (128B)",
                              "E quad/node hdr+12B (64B)", "F oct/node hdr+12B (128B)", "G quad hdr+16B (128B str)"};
      printf("nodes 2^%d waves/SIMD %d  %-26s %8.3f ms  %7.2f Gvisits/s  (%.2f TB/s node bytes)\n", lg, waves, names[mode], ms,
             visits / ms / 1e6, visits * (mode == 2 ? 80 : (mode == 3 || mode == 5) ? 128 : mode == 6 ? 80 : 64) / ms / 1e9);
    }
    CHECK(hipFree(d)); CHECK(hipFree(out));
  }
  return 0;
}
