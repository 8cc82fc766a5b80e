// Issue rate of packed vs plain f32 VALU ops on gfx950 (dev aid, not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int MODE>
__global__ void k(long long* out, float* sink, int iters)
{
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
  float2 m = {1.0000001f, 0.9999999f};
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++)
  {
    if (MODE == 0)
    { // 8 independent v_pk_mul_f32
      asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                   "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(m));
    }
    else if (MODE == 1)
    { // 8 independent v_pk_add_f32
      asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                   "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(m));
    }
    else if (MODE == 2)
    { // 8 independent v_mul_f32
      asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                   "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m.x));
    }
    else if (MODE == 3)
    { // 8 independent v_pk_fma_f32
      asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                   "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(m));
    }
    else if (MODE == 4)
    { // pk_mul with an SGPR pair operand and op_sel (the FIR / resampler form)
      asm volatile("s_mov_b32 s20, 0x3f800001\n s_mov_b32 s21, 0x3f7fffff\n"
                   "v_pk_mul_f32 %0, %0, s[20:21] op_sel_hi:[1,0]\n v_pk_mul_f32 %1, %1, s[20:21] op_sel:[0,1]\n"
                   "v_pk_mul_f32 %2, %2, s[20:21] op_sel_hi:[1,0]\n v_pk_mul_f32 %3, %3, s[20:21] op_sel:[0,1]\n"
                   "v_pk_mul_f32 %4, %4, s[20:21] op_sel_hi:[1,0]\n v_pk_mul_f32 %5, %5, s[20:21] op_sel:[0,1]\n"
                   "v_pk_mul_f32 %6, %6, s[20:21] op_sel_hi:[1,0]\n v_pk_mul_f32 %7, %7, s[20:21] op_sel:[0,1]\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : : "s20", "s21");
    }
    else if (MODE == 5)
    { // dependent chain: pk_mul -> pk_add alternating on one accumulator pair (tap loop form, products ahead)
      asm volatile("v_pk_mul_f32 %1, %1, %8\n v_pk_add_f32 %0, %0, %2\n v_pk_mul_f32 %2, %2, %8\n v_pk_add_f32 %0, %0, %3\n"
                   "v_pk_mul_f32 %3, %3, %8\n v_pk_add_f32 %0, %0, %4\n v_pk_mul_f32 %4, %4, %8\n v_pk_add_f32 %0, %0, %1\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(m));
    }
    else if (MODE == 6)
    { // same work with plain ops: 8 mul + 8 add (one chain per component)
      asm volatile("v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %5\n"
                   "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_add_f32 %0, %0, %6\n v_add_f32 %1, %1, %7\n"
                   "v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %3\n"
                   "v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %5\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m.x));
    }
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0)
    out[0] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}
template <int MODE>
void run(const char* name, int nops, int threads)
{
  long long* d; float* sink; hipMalloc(&d, 8); hipMalloc(&sink, 4 * 1024 * 1024);
  const int iters = 20000;
  hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, d, sink, 100);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, d, sink, iters);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, d, 8, hipMemcpyDeviceToHost);
  printf("%-34s threads=%4d  %.2f ns per op per wave-stream (%.0f us total; s_memtime ticks/op %.3f)\n", name, threads,
         ms * 1e6 / ((double)iters * nops), ms * 1e3, (double)c / ((double)iters * nops));
  hipFree(d); hipFree(sink);
}
int main()
{
  for (int th : {64, 256, 512, 1024})
  {
    run<2>("v_mul_f32 x8", 8, th);
    run<0>("v_pk_mul_f32 x8", 8, th);
    run<1>("v_pk_add_f32 x8", 8, th);
    run<3>("v_pk_fma_f32 x8", 8, th);
    run<4>("v_pk_mul_f32 sgpr op_sel x8", 8, th);
    run<5>("pk mul/add tap chain (8 ops=4 taps)", 8, th);
    run<6>("plain mul/add tap chain (16 ops=4 taps)", 16, th);
  }
  return 0;
}
