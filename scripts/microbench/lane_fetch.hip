// How does the rate of a lane-per-node pointer chase (every lane of a wave fetches its own random node, the next index
// comes from the data: k_trace's access pattern) depend on the NUMBER of 16-B loads a lane issues per node?
// If the vector-memory path processes one lane-request per cycle, a node of K quads costs K of them whatever the
// cache line holds.  K = 1..4 quads of a 64-B node (stride 64 B), K = 3 of a 48-B node (stride 48 B), K = 5 of an 80-B node.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int K, int STRIDE_Q, int FILL>
__global__ void __launch_bounds__(256) k_chase(const uint4* __restrict__ nodes, uint32_t mask, int iters, uint32_t* out) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t idx = (tid * 2654435761u) & mask;
  uint32_t acc = 0;
  float f = (float)tid;
  for (int i = 0; i < iters; ++i) {
    const uint4* n = nodes + (size_t)STRIDE_Q * idx;
    uint32_t next = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const uint4 a = n[k];
      next ^= k == 0 ? a.x : k == 1 ? a.y : k == 2 ? a.z : a.w;
    }
#pragma unroll
    for (int k = 0; k < FILL; ++k) f = f * 1.0001f + (float)next;  // dependent VALU work per visit (k_trace: ~125 per node step)
    acc += next;
    idx = (next * 2246822519u + tid) & mask;
  }
  if (acc == 0x12345678u || f == 1.2345f) out[0] = acc;
}

template <int K, int STRIDE_Q, int FILL>
static void run(const char* name, const uint4* d, uint32_t N, int waves, uint32_t* out) {
  const int iters = 300, blocks = 256 * waves;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_chase<K, STRIDE_Q, FILL>), dim3(blocks), dim3(256), 0, 0, d, N - 1, iters, out);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
  }
  const double visits = (double)blocks * 256 * iters;
  printf("nodes %8u (%5.1f MB) waves/SIMD %d  %-34s %8.3f ms  %7.2f G visits/s  %7.2f G lane-loads/s\n", N, N * 16.0 * STRIDE_Q / 1e6, waves, name, ms,
         visits / ms / 1e6, visits * K / ms / 1e6);
}

int main() {
  for (int lg = 15; lg <= 23; lg += 4) {  // 32 K, 512 K (the bench scene's BVH: 494 k nodes), 8 M nodes
    const uint32_t N = 1u << lg;
    std::vector<uint32_t> h((size_t)N * 5 * 4);
    uint32_t s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s >> 3; }
    uint4* d; uint32_t* out;
    CHECK(hipMalloc(&d, h.size() * 4)); CHECK(hipMalloc(&out, 4));
    CHECK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (int waves = 7; waves <= 7; ++waves) {
      run<1, 4, 0>("1 x 16 B of a 64-B node", d, N, waves, out);
      run<2, 4, 0>("2 x 16 B of a 64-B node", d, N, waves, out);
      run<3, 4, 0>("3 x 16 B of a 64-B node", d, N, waves, out);
      run<4, 4, 0>("4 x 16 B of a 64-B node", d, N, waves, out);
      run<3, 3, 0>("3 x 16 B of a 48-B node", d, N, waves, out);
      run<5, 5, 0>("5 x 16 B of an 80-B node", d, N, waves, out);
      run<4, 4, 64>("4 x 16 B + 64 dependent fma", d, N, waves, out);
      run<3, 3, 64>("3 x 16 B (48-B) + 64 dependent fma", d, N, waves, out);
      run<4, 4, 128>("4 x 16 B + 128 dependent fma", d, N, waves, out);
      run<3, 3, 128>("3 x 16 B (48-B) + 128 dependent fma", d, N, waves, out);
      run<3, 3, 140>("3 x 16 B (48-B) + 140 dependent fma", d, N, waves, out);
    }
    CHECK(hipFree(d)); CHECK(hipFree(out));
  }
  return 0;
}
