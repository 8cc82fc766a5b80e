// Does a latency-bound kernel that owns its CUs slow down while other CUs stream from HBM, and if so
// is it the shader clock (cycle count unchanged) or stalls (cycle count up)?  Dev aid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void chain(int iters, long long* out, float* sink)
{
  asm volatile("" ::: "v255", "a255"); // whole register file: nothing else fits on this CU's SIMDs
  float x = threadIdx.x * 1e-3f, y = 1.0f;
  long long c0 = __builtin_readcyclecounter();
  long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; i++)
  {
#pragma unroll
    for (int k = 0; k < 16; k++)
    {
      x = __builtin_fmaf(x, 0.999f, y);
      y = __builtin_fmaf(y, 0.5f, x);
    }
  }
  long long c1 = __builtin_readcyclecounter();
  long long r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0)
  {
    out[2 * blockIdx.x] = c1 - c0;
    out[2 * blockIdx.x + 1] = r1 - r0;
  }
  sink[blockIdx.x * 256 + threadIdx.x] = x + y;
}
__global__ __launch_bounds__(256) void stream(const float4* __restrict__ a, float4* __restrict__ b, size_t n, int reps)
{
  for (int r = 0; r < reps; r++)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    {
      float4 v = a[i];
      v.x += r;
      b[i] = v;
    }
}
__global__ __launch_bounds__(256) void valu(float* sink, int iters)
{ // pure VALU load on the other CUs (no memory)
  float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3;
  for (int i = 0; i < iters; i++)
  {
#pragma unroll
    for (int k = 0; k < 16; k++)
    {
      a0 = __builtin_fmaf(a0, 0.999f, 1.0f); a1 = __builtin_fmaf(a1, 0.999f, 1.0f);
      a2 = __builtin_fmaf(a2, 0.999f, 1.0f); a3 = __builtin_fmaf(a3, 0.999f, 1.0f);
    }
  }
  sink[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;
}
int main()
{
  hipStream_t s1, s2;
  hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  long long* out; float* sink; float4 *a, *b;
  const size_t n = (size_t)1 << 27; // 2 GiB per array
  hipMalloc(&out, 64 * 16); hipMalloc(&sink, 4 << 20); hipMalloc(&a, n * 16); hipMalloc(&b, n * 16);
  hipMemset(a, 0, n * 16);
  const int iters = 60000; // ~ 2 ms of dependent FMAs
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 3; mode++)
  {
    for (int rep = 0; rep < 3; rep++)
    {
      if (mode == 1)
        hipLaunchKernelGGL(stream, dim3(192 * 8), dim3(256), 0, s2, a, b, n, 3);
      if (mode == 2)
        hipLaunchKernelGGL(valu, dim3(192 * 8), dim3(256), 0, s2, sink + (1 << 18), 400000);
      hipEventRecord(e0, s1);
      hipLaunchKernelGGL(chain, dim3(64), dim3(256), 0, s1, iters, out, sink);
      hipEventRecord(e1, s1);
      hipStreamSynchronize(s1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipEvent_t f0, f1; hipEventCreate(&f0); hipEventCreate(&f1);
      hipDeviceSynchronize();
      std::vector<long long> h(128);
      hipMemcpy(h.data(), out, 128 * 8, hipMemcpyDeviceToHost);
      double cyc = 0, rt = 0;
      for (int i = 0; i < 64; i++) { cyc += h[2 * i]; rt += h[2 * i + 1]; }
      cyc /= 64; rt /= 64;
      printf("%-28s chain %.3f ms  cycles %.0f  realtime ticks %.0f (100 MHz -> %.3f ms)  => %.0f MHz, %.2f cycles per dependent FMA\n",
             mode == 0 ? "alone" : mode == 1 ? "beside an HBM stream (192x8)" : "beside a VALU kernel (192x8)", ms, cyc, rt, rt / 1e5,
             cyc / (rt / 100.0), cyc / (iters * 32.0));
    }
  }
  return 0;
}
