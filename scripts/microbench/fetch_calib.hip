// Calibration of rocprofv3's FETCH_SIZE for the traversal's access pattern: every lane of a wave reads its own random
// 64-B record with 4 x 16-B loads (the k_trace node fetch), from a table far larger than L2 + Infinity Cache, so
// (almost) every record comes from HBM and the bytes that must cross the memory side are known: visits x 64 B.
// MI355X_MICROARCH.md (HBM section) gives the x2 correction only for wide coalesced streaming reads and says other
// widths are uncalibrated; this prints the known byte count to compare with the counter:
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- ./fetch_calib          (FETCH_SIZE is in KiB)
// Also runs the coalesced streaming read of the same table as the control (expected: counter = 1/2 of the bytes).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_gather64(const uint4* __restrict__ nodes, uint32_t mask, int iters, uint32_t* out) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t idx = (tid * 2654435761u) & mask, acc = 0;
  for (int i = 0; i < iters; ++i) {
    const uint4* n = nodes + 4ull * idx;
    const uint4 a = n[0], b = n[1], c = n[2], d = n[3];
    const uint32_t next = a.x ^ b.y ^ c.z ^ d.w;
    acc += next;
    idx = (next * 2246822519u + tid * 40503u + i * 7919u) & mask;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_stream(const uint4* __restrict__ p, size_t n, uint32_t* out) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc ^= p[i].x;
  if (acc == 0x12345678u) out[0] = acc;
}
// WRITE_SIZE controls: a coalesced 16-B-per-lane streaming write of the table, and scattered 16-B writes (one random
// quad per lane per iteration: the pattern of the sample-result ring and of k_trace's hit records)
__global__ void __launch_bounds__(256) k_wstream(uint4* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}
__global__ void __launch_bounds__(256) k_wscatter16(uint4* __restrict__ p, uint32_t mask, int iters) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t idx = (tid * 2654435761u) & mask;
  for (int i = 0; i < iters; ++i) {
    p[idx] = make_uint4(idx, tid, (uint32_t)i, 7u);
    idx = (idx * 2246822519u + tid * 40503u + i * 7919u + 1u) & mask;
  }
}
__global__ void k_fill(uint4* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    uint32_t s = (uint32_t)i * 1664525u + 1013904223u;
    p[i] = make_uint4(s, s * 3u + 1u, s * 5u + 2u, s * 7u + 3u);
  }
}
int main() {
  const uint32_t N = 1u << 26;  // 64 Mi records x 64 B = 4 GiB (16x the Infinity Cache)
  uint4* d; uint32_t* out;
  CHECK(hipMalloc(&d, (size_t)N * 64)); CHECK(hipMalloc(&out, 4));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, d, (size_t)N * 4);
  CHECK(hipDeviceSynchronize());
  const int iters = 64, blocks = 256 * 8;
  hipLaunchKernelGGL(k_gather64, dim3(blocks), dim3(256), 0, 0, d, N - 1, iters, out);
  CHECK(hipDeviceSynchronize());
  printf("k_gather64: %d lanes x %d records x 64 B = %.1f MiB must come from memory (minus ~6 %% Infinity-Cache hits)\n",
         blocks * 256, iters, (double)blocks * 256 * iters * 64 / 1048576.0);
  hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, 0, d, (size_t)N * 4, out);
  CHECK(hipDeviceSynchronize());
  printf("k_stream: %.1f MiB read once, 16 B per lane, coalesced\n", (double)N * 64 / 1048576.0);
  hipLaunchKernelGGL(k_wstream, dim3(4096), dim3(256), 0, 0, d, (size_t)N * 4);
  CHECK(hipDeviceSynchronize());
  printf("k_wstream: %.1f MiB written once, 16 B per lane, coalesced\n", (double)N * 64 / 1048576.0);
  hipLaunchKernelGGL(k_wscatter16, dim3(blocks), dim3(256), 0, 0, d, 4u * N - 1u, iters);
  CHECK(hipDeviceSynchronize());
  printf("k_wscatter16: %d lanes x %d quads x 16 B = %.1f MiB of scattered 16-B writes\n", blocks * 256, iters,
         (double)blocks * 256 * iters * 16 / 1048576.0);
  return 0;
}
