// The long-filter tap loop of k_if_fir (fir_long_b128_asm, 16-byte window reads, D = 46) on its own:
// one workgroup of four waves per CU with a 127 KB window of ones in LDS, every lane's 4064-tap sum,
// no staging, no stores but one -- what the loop itself costs per 16 taps, on one CU and on all of them
// (the taps come through the scalar cache, which CUs share).  Dev aid, not part of the product.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/ubench/fir_long.hip -o fir_long
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "../../pvr.rtl.radiofm_amd/csrc/fmd_design.hpp"
#include "../../pvr.rtl.radiofm_amd/csrc/fmd_kernels.hip.h"

__global__ __launch_bounds__(256) void k(const float* __restrict__ coeff, long long* cyc, float2* sink, unsigned reps,
                                         unsigned order, unsigned D)
{
  extern __shared__ __attribute__((aligned(16))) float2 smem[];
  float2* win = smem + 32;
  const unsigned nslots = 255u * D + order + 8u;
  for (unsigned i = threadIdx.x; i < nslots + 32; i += 256)
    smem[i] = make_float2(1.0f, 0.5f);
  __syncthreads();
  const unsigned tid = threadIdx.x;
  fmd::fmd_f2v acc2 = {0.0f, 0.0f};
  const long long t0 = __builtin_readcyclecounter();
  for (unsigned r = 0; r < reps; r++)
  {
    const float2* w = win + tid * D + order;
    unsigned j = 1; // the lowest slot of a batch (w - j - 15) is even: 16-byte aligned reads
    unsigned cnt = (unsigned)__builtin_amdgcn_readfirstlane((int)((order + 1u - j) >> 5));
    unsigned a = (unsigned)(size_t)(w - (int)j - 15);
    const size_t ka = (size_t)(coeff + j);
    const unsigned klo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ka);
    const unsigned khi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ka >> 32));
    fmd::fir_long_b128_asm(acc2, a, klo, khi, cnt);
  }
  const long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0)
    cyc[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 256 + threadIdx.x] = make_float2(acc2.x, acc2.y);
}

int main()
{
  const unsigned order = 4096, D = 46, reps = 6;
  std::vector<float> h(order + 128, 1e-4f);
  float* d_c;
  long long* d_cyc;
  float2* d_sink;
  hipMalloc(&d_c, h.size() * 4);
  hipMemcpy(d_c, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&d_cyc, 4096 * 8);
  hipMalloc(&d_sink, 4096 * 256 * 8);
  const size_t lds = (size_t(255) * D + order + 8 + 64) * sizeof(float2);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (unsigned grid : {1u, 64u, 256u, 1024u})
  {
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, d_c, d_cyc, d_sink, reps, order, D);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, d_c, d_cyc, d_sink, reps, order, D);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> c(grid);
    hipMemcpy(c.data(), d_cyc, grid * 8, hipMemcpyDeviceToHost);
    double s = 0, mx = 0;
    for (auto v : c)
    {
      s += double(v);
      mx = double(v) > mx ? double(v) : mx;
    }
    const double halves = double(reps) * (order >> 5) * 2;
    printf("grid %4u: %.3f ms, %.1f cycles per 16 taps and wave (mean over workgroups; max %.1f)\n", grid, ms,
           s / grid / halves, mx / halves);
  }
  return 0;
}
