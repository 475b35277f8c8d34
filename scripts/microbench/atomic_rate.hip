// Rate of RETURNING global atomics that many blocks aim at few addresses -- the pattern of a wavefront tracer's queue
// tails (k_shade: every resident block reserves its tile's room once per tile and waits for the answer at a barrier).
//   hipcc --offload-arch=gfx950 -O3 atomic_rate.hip -o atomic_rate && ./atomic_rate
// Each of `blocks` blocks of 256 threads loops: thread 0 (and 1) issue the atomic(s), everyone meets at a barrier, the
// base goes through LDS (as in k_shade); `work` dependent FMAs per thread stand for the tile's arithmetic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e = (x);                                                           \
    if (e != hipSuccess) {                                                        \
      printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e));             \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

// mode 0: one u32 counter              1: two u32 counters in one line (lanes 0, 1 of one instruction)
// mode 2: two u32 counters 4 KB apart   3: one u64 counter
// mode 4: one u32 counter per shard (block % shards), shards 4 KB apart
// mode 5: as 0 but the result is not used (non-returning atomic)
template <int MODE>
__global__ __launch_bounds__(256) void k_atomics(uint32_t* ctr, int iters, int work, int shards, float* sink) {
  __shared__ uint32_t s_base[2];
  float acc = threadIdx.x;
  uint32_t sum = 0;
  for (int it = 0; it < iters; ++it) {
    for (int k = 0; k < work; ++k) acc = __builtin_fmaf(acc, 1.0000001f, 0.5f);
    if (MODE == 0 || MODE == 4 || MODE == 5) {
      if (threadIdx.x == 0) {
        uint32_t* a = ctr + (MODE == 4 ? (blockIdx.x % shards) * 1024 : 0);
        if (MODE == 5) {
          atomicAdd(a, 1u);
          s_base[0] = it;
        } else {
          s_base[0] = atomicAdd(a, 1u);
        }
      }
    } else if (MODE == 1 || MODE == 2) {
      if (threadIdx.x < 2) s_base[threadIdx.x] = atomicAdd(ctr + threadIdx.x * (MODE == 2 ? 1024 : 1), 1u);
    } else if (MODE == 3) {
      if (threadIdx.x == 0) {
        const unsigned long long b = atomicAdd((unsigned long long*)ctr, (1ull << 32) | 1ull);
        s_base[0] = (uint32_t)b;
        s_base[1] = (uint32_t)(b >> 32);
      }
    }
    __syncthreads();
    sum += s_base[0];
    __syncthreads();
  }
  if (acc == 12345.678f || sum == 0x12345678u) sink[0] = acc;
}

template <int MODE>
static void run(const char* name, int blocks, int iters, int work, int shards, uint32_t* ctr, float* sink) {
  CHECK(hipMemset(ctr, 0, 64 * 4096));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_atomics<MODE>, dim3(blocks), dim3(256), 0, 0, ctr, 8, work, shards, sink);  // warm-up
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_atomics<MODE>, dim3(blocks), dim3(256), 0, 0, ctr, iters, work, shards, sink);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double per_it = (MODE == 1 || MODE == 2) ? 2.0 : 1.0;
  const double n = (double)blocks * iters;
  printf("%-44s blocks %4d work %5d shards %2d: %7.2f us per iteration and block, %7.1f M iterations/s, %7.1f M atomics/s\n", name, blocks, work,
         shards, ms * 1e3 / iters, n / ms / 1e3, n * per_it / ms / 1e3);
}

int main() {
  uint32_t* ctr;
  float* sink;
  CHECK(hipMalloc(&ctr, 64 * 4096));
  CHECK(hipMalloc(&sink, 64));
  const int iters = 2000;
  for (int blocks : {256, 1024}) {
    for (int work : {0, 2000}) {
      run<0>("one u32 counter", blocks, iters, work, 1, ctr, sink);
      run<5>("one u32 counter, result unused", blocks, iters, work, 1, ctr, sink);
      run<1>("two u32 counters, one line, one instruction", blocks, iters, work, 1, ctr, sink);
      run<2>("two u32 counters, 4 KB apart", blocks, iters, work, 1, ctr, sink);
      run<3>("one u64 counter", blocks, iters, work, 1, ctr, sink);
      for (int shards : {2, 4, 8, 16, 64}) run<4>("u32 counter per shard (block % shards)", blocks, iters, work, shards, ctr, sink);
    }
  }
  return 0;
}
