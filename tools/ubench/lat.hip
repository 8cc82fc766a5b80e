// Issue / dependent-issue cost of the instruction kinds in the serial stage's sample loops for a LONE
// wave on its SIMD (gfx950): what one instruction costs when the next one needs its result, and when
// it does not.  Dev aid (not part of the product).  Every body is 240 instructions of straight-line
// code per loop trip, so the loop's back edge (~28 cycles) is < 3 % of a trip.
//
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/lat.hip -o tools/ubench/lat && ./tools/ubench/lat
#include <hip/hip_runtime.h>

#include <cstdio>

#define R2(x) x x
#define R3(x) x x x
#define R4(x) x x x x
#define R5(x) x x x x x
#define R6(x) x x x x x x
#define R8(x) R4(x) R4(x)
#define R10(x) R5(x) R5(x)
#define R12(x) R6(x) R6(x)
#define R15(x) R5(x) R5(x) R5(x)
#define R20(x) R10(x) R10(x)
#define R30(x) R10(x) R10(x) R10(x)
#define R40(x) R20(x) R20(x)
#define R60(x) R30(x) R30(x)
#define R120(x) R60(x) R60(x)
#define R240(x) R120(x) R120(x)

typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(long long* out, float* sink, int iters)
{
  __shared__ __attribute__((aligned(16))) unsigned lds[8192]; // 32 KB
  for (unsigned i = threadIdx.x; i < 1024; i += blockDim.x)
    lds[i] = (i * 4u) & 0xffcu; // a pointer chain inside the array
  __syncthreads();
  float a = threadIdx.x * 1e-3f + 1.0f, b = a + 1, c = a + 2, d = a + 3;
  float m = 0.99999f, n = 1e-6f;
  v2f pa = {a, b}, pb = {c, d}, pc = {b, c}, pd = {d, a}, pm = {m, m}, pn = {n, n};
  double da = a, db = b, dc = c, dd = d, dm = m, dn = n;
  unsigned ia = threadIdx.x, ib = ia + 1, ic = ia + 2, id = ia + 3;
  unsigned la = (threadIdx.x * 4u) & 0xffcu;
  const unsigned la46 = (threadIdx.x & 63u) * 368u;
  asm volatile("" : "+v"(m), "+v"(n), "+v"(pm), "+v"(pn), "+v"(dm), "+v"(dn));
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++)
  {
    if (MODE == 0) // dependent v_fma_f32
      asm volatile(R240("v_fma_f32 %0, %0, %1, %2\n") : "+v"(a) : "v"(m), "v"(n));
    else if (MODE == 1) // four independent v_fma_f32 chains
      asm volatile(R60("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n")
                   : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(n));
    else if (MODE == 2) // two independent chains
      asm volatile(R120("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n") : "+v"(a), "+v"(b) : "v"(m), "v"(n));
    else if (MODE == 3) // dependent v_pk_fma_f32
      asm volatile(R240("v_pk_fma_f32 %0, %0, %1, %2\n") : "+v"(pa) : "v"(pm), "v"(pn));
    else if (MODE == 4) // four independent v_pk_fma_f32 chains
      asm volatile(R60("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n")
                   : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd) : "v"(pm), "v"(pn));
    else if (MODE == 5) // dependent v_pk_mul_f32 / v_pk_add_f32 alternating (polynomial form)
      asm volatile(R120("v_pk_mul_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %2\n") : "+v"(pa) : "v"(pm), "v"(pn));
    else if (MODE == 6) // packed -> plain -> packed dependent (plain reads the packed result's low half)
      asm volatile("v_mov_b32 v100, 1.0\n v_mov_b32 v101, 1.0\n" R120("v_pk_mul_f32 v[100:101], v[100:101], %0\n v_add_f32 v100, v100, %1\n") : : "v"(pm), "v"(n) : "v100", "v101");
    else if (MODE == 7) // dependent v_fma_f64
      asm volatile(R240("v_fma_f64 %0, %0, %1, %2\n") : "+v"(da) : "v"(dm), "v"(dn));
    else if (MODE == 8) // four independent v_fma_f64 chains
      asm volatile(R60("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n")
                   : "+v"(da), "+v"(db), "+v"(dc), "+v"(dd) : "v"(dm), "v"(dn));
    else if (MODE == 9) // dependent v_add_u32
      asm volatile(R240("v_add_u32 %0, %0, %1\n") : "+v"(ia) : "v"(ib));
    else if (MODE == 10) // v_cmp -> v_cndmask dependent (the compiler's s_nop 1 in between)
      asm volatile(R120("v_cmp_lt_u32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %1, %2, vcc\n") : "+v"(ia) : "v"(ib), "v"(ic) : "vcc");
    else if (MODE == 11) // dependent v_rcp_f32 (with the s_nop 0 the compiler puts behind a transcendental)
      asm volatile(R120("v_rcp_f32 %0, %0\n s_nop 0\n v_add_f32 %0, %0, %1\n") : "+v"(a) : "v"(n));
    else if (MODE == 12) // float -> double -> float
      asm volatile(R120("v_cvt_f64_f32 %1, %0\n v_cvt_f32_f64 %0, %1\n") : "+v"(a), "+v"(da));
    else if (MODE == 13) // LDS pointer chase: ds_read_b32 -> its own address
      asm volatile(R60("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n") : "+v"(la)::"memory");
    else if (MODE == 14) // dependent v_fma_f32 with an s_nop 0 behind each
      asm volatile(R120("v_fma_f32 %0, %0, %1, %2\n s_nop 0\n") : "+v"(a) : "v"(m), "v"(n));
    else if (MODE == 15) // dependent v_pk_fma_f32 with the s_nop 0 the compiler puts between them
      asm volatile(R120("v_pk_fma_f32 %0, %0, %1, %2\n s_nop 0\n") : "+v"(pa) : "v"(pm), "v"(pn));
    else if (MODE == 16) // dependent chain, every other instruction an independent one
      asm volatile(R120("v_fma_f32 %0, %0, %2, %3\n v_add_u32 %1, %1, %1\n") : "+v"(a), "+v"(ia) : "v"(m), "v"(n));
    else if (MODE == 17) // dependent chain with two independent ones in between
      asm volatile(R60("v_fma_f32 %0, %0, %3, %4\n v_add_u32 %1, %1, %1\n v_add_u32 %2, %2, %2\n") : "+v"(a), "+v"(ia), "+v"(ib) : "v"(m), "v"(n));
    else if (MODE == 18) // dependent f64 chain with one independent f32 op in between
      asm volatile(R120("v_fma_f64 %0, %0, %2, %3\n v_add_f32 %1, %1, %1\n") : "+v"(da), "+v"(a) : "v"(dm), "v"(dn));
    else if (MODE == 19) // dependent v_mul_f32 -> v_add_f32 (two-operand encodings)
      asm volatile(R120("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2\n") : "+v"(a) : "v"(m), "v"(n));
    else if (MODE == 20) // dependent packed chain with one independent plain op in between
      asm volatile(R120("v_pk_fma_f32 %0, %0, %2, %3\n v_add_u32 %1, %1, %1\n") : "+v"(pa), "+v"(ia) : "v"(pm), "v"(pn));
    else if (MODE == 21) // v_div_scale -> v_div_fmas -> v_div_fixup dependent
      asm volatile(R60("v_div_scale_f32 %0, vcc, %0, %1, %0\n s_nop 1\n v_div_fmas_f32 %0, %0, %1, %1\n v_div_fixup_f32 %0, %0, %1, %1\n") : "+v"(a) : "v"(m) : "vcc");
    else if (MODE == 22) // s_nop 0 alone
      asm volatile(R240("s_nop 0\n"));
    else if (MODE == 24 || MODE == 25)
    { // k_if_fir's long-filter half iteration (16 taps): 16 packed multiplies by SGPR taps, 16 packed
      // adds in one chain, products two ahead; MODE 25 with the eight 16-byte LDS reads of the next
      // batch issued in front and waited for at the end, as in fir_long_b128_asm
#define TAPMUL(t, x) "v_pk_mul_f32 v[" #t ":" #t "+1], v[" #x ":" #x "+1], s[20:21] op_sel_hi:[1,0]\n"
#define TAPADD(t) "v_pk_add_f32 %0, %0, v[" #t ":" #t "+1]\n"
#define HALF_READS "ds_read_b128 v[64:67], %1 offset:112\n ds_read_b128 v[68:71], %1 offset:96\n ds_read_b128 v[72:75], %1 offset:80\n ds_read_b128 v[76:79], %1 offset:64\n ds_read_b128 v[80:83], %1 offset:48\n ds_read_b128 v[84:87], %1 offset:32\n ds_read_b128 v[88:91], %1 offset:16\n ds_read_b128 v[92:95], %1\n"
#define HALF_MAC TAPMUL(128, 96) TAPMUL(130, 98) TAPADD(128) TAPMUL(132, 100) TAPADD(130) TAPMUL(134, 102) TAPADD(132) TAPMUL(128, 104) TAPADD(134) TAPMUL(130, 106) TAPADD(128) TAPMUL(132, 108) TAPADD(130) TAPMUL(134, 110) TAPADD(132) TAPMUL(128, 112) TAPADD(134) TAPMUL(130, 114) TAPADD(128) TAPMUL(132, 116) TAPADD(130) TAPMUL(134, 118) TAPADD(132) TAPMUL(128, 120) TAPADD(134) TAPMUL(130, 122) TAPADD(128) TAPMUL(132, 124) TAPADD(130) TAPMUL(134, 126) TAPADD(132) TAPADD(134)
      if (MODE == 24)
        asm volatile("s_mov_b32 s20, 0x3f800000\n s_mov_b32 s21, 0x3f800000\n" R4(HALF_MAC)
                     : "+v"(pa) : "v"(la)
                     : "s20", "s21", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106",
                       "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118",
                       "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130",
                       "v131", "v132", "v133", "v134", "v135");
      else
        asm volatile("s_mov_b32 s20, 0x3f800000\n s_mov_b32 s21, 0x3f800000\n" R4(HALF_READS HALF_MAC "s_waitcnt lgkmcnt(0)\n")
                     : "+v"(pa) : "v"(la46) // lanes 46 samples = 368 bytes apart: conflict-free 16-byte reads
                     : "s20", "s21", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75",
                       "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89",
                       "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102",
                       "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114",
                       "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126",
                       "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "memory");
    }
    else if (MODE == 23) // scalar ALU dependent
      asm volatile(R240("s_add_u32 s20, s20, 1\n") ::: "s20", "scc");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0)
    out[0] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + pa.x + pa.y + pb.x + pc.x + pd.x + (float)(da + db + dc + dd) + ia + ib + ic + id + la;
}

template <int MODE>
void run(const char* name, int per_trip, int threads)
{
  long long* d;
  float* sink;
  hipMalloc(&d, 8);
  hipMalloc(&sink, 4 * 4096);
  const int iters = 2000;
  hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, d, sink, 50);
  hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, d, sink, iters);
  hipDeviceSynchronize();
  long long c;
  hipMemcpy(&c, d, 8, hipMemcpyDeviceToHost);
  printf("%-78s waves/SIMD=%d  %6.2f cycles per instruction\n", name, threads / 256 ? threads / 256 : 1, (double)c / ((double)iters * per_trip));
  hipFree(d);
  hipFree(sink);
}

int main()
{
  for (int th : {64, 256, 512})
  {
    run<0>("v_fma_f32, each needs the previous one's result", 240, th);
    run<2>("v_fma_f32, two independent chains", 240, th);
    run<1>("v_fma_f32, four independent chains", 240, th);
    run<19>("v_mul_f32 -> v_add_f32 dependent", 240, th);
    run<14>("v_fma_f32 dependent + s_nop 0 (per pair)", 120, th);
    run<16>("v_fma_f32 dependent + 1 independent v_add_u32 (per pair)", 120, th);
    run<17>("v_fma_f32 dependent + 2 independent v_add_u32 (per triple)", 60, th);
    run<3>("v_pk_fma_f32 dependent", 240, th);
    run<15>("v_pk_fma_f32 dependent + s_nop 0 (per pair)", 120, th);
    run<20>("v_pk_fma_f32 dependent + 1 independent v_add_u32 (per pair)", 120, th);
    run<4>("v_pk_fma_f32, four independent chains", 240, th);
    run<5>("v_pk_mul_f32 -> v_pk_add_f32 dependent", 240, th);
    run<6>("v_pk_mul_f32 -> v_add_f32 dependent", 240, th);
    run<7>("v_fma_f64 dependent", 240, th);
    run<8>("v_fma_f64, four independent chains", 240, th);
    run<18>("v_fma_f64 dependent + 1 independent v_add_f32 (per pair)", 120, th);
    run<9>("v_add_u32 dependent", 240, th);
    run<10>("v_cmp -> s_nop 1 -> v_cndmask dependent (per triple)", 120, th);
    run<11>("v_rcp_f32 -> s_nop 0 -> v_add_f32 dependent (per triple)", 120, th);
    run<12>("v_cvt_f64_f32 -> v_cvt_f32_f64 dependent", 240, th);
    run<21>("v_div_scale -> s_nop 1 -> v_div_fmas -> v_div_fixup (per group of 4)", 60, th);
    run<13>("ds_read_b32 -> s_waitcnt -> its own address (per read)", 60, th);
    run<22>("s_nop 0", 240, th);
    run<23>("s_add_u32 dependent", 240, th);
    run<24>("long-filter half iteration, 16 taps = 32 packed instructions (per half)", 4, th);
    run<25>("the same with its 8 ds_read_b128 + s_waitcnt (per half)", 4, th);
  }
  return 0;
}
