// What does hipcc (ROCm 7.2) wait for when an LDS-DMA load (global_load_lds_dwordx4) is in flight?  Compile only:
//   hipcc --offload-arch=gfx950 -O3 --cuda-device-only -S lds_dma_waits.hip -o - | grep -n "global_load\|s_waitcnt"
// k_order<N, TAB>: a plain load issued BEFORE N DMAs is answered with s_waitcnt vmcnt(0) at its first use (not vmcnt(N));
// k_plainloads: the same with ordinary loads gets vmcnt(6), vmcnt(5), ... (in-order counting).  k_shade therefore waits for its
// shading packet before it issues the DMA of the next tile.
#include <hip/hip_runtime.h>
typedef __attribute__((address_space(1))) const void glb_cvoid;
typedef __attribute__((address_space(3))) void lds_void;
template <int NDMA, bool TAB>
__global__ void k_order(const float4* src, const float4* pk, float* out, float* out2, int idx) {
  __shared__ float4 buf[5][256];
  __shared__ float tab[64];
  if (threadIdx.x < 64) tab[threadIdx.x] = out2[threadIdx.x];
  __syncthreads();
  const int i = threadIdx.x;
  const float4 a = pk[idx + i * 7];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NDMA; ++k)
    __builtin_amdgcn_global_load_lds((glb_cvoid*)(src + i + 256 * k), (lds_void*)(&buf[k][threadIdx.x & ~63]), 16, 0, 2);
  float x = (TAB ? tab[i & 63] : 1.5f) * a.x;
  for (int k = 0; k < 64; ++k) x = __builtin_fmaf(x, 1.0001f, a.z);
  out[i] = x;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  out2[threadIdx.x] = buf[1][255 - threadIdx.x].x + buf[4][threadIdx.x].y;
}
template __global__ void k_order<5, false>(const float4*, const float4*, float*, float*, int);
template __global__ void k_order<1, false>(const float4*, const float4*, float*, float*, int);
template __global__ void k_order<5, true>(const float4*, const float4*, float*, float*, int);
// the same with plain loads instead of DMA: what does the pass do for ordinary in-order loads?
__global__ void k_plainloads(const float4* src, const float4* pk, float* out, float* out2, int idx) {
  const int i = threadIdx.x;
  const float4 a = pk[idx + i * 7];
  float4 b[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) b[k] = src[i + 256 * k];
  float x = 1.5f * a.x;
  for (int k = 0; k < 64; ++k) x = __builtin_fmaf(x, 1.0001f, a.z);
  out[i] = x;
  out2[i] = b[0].x + b[1].y + b[2].z + b[3].w + b[4].x;
}
