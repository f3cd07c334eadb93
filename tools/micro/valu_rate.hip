// micro-test: issue rate of vector instructions on gfx950 by ENCODING (VOP2 / VOP3 / VOP3P) and by waves per SIMD.
// One workgroup of 64 .. 1024 threads on one CU (1024 threads = 4 waves per SIMD), then two workgroups of 1024 (8 waves per SIMD where
// the CU admits both).  16 independent chains per instruction kind, 64 instructions per loop iteration; wall time by HIP events and
// in-kernel cycles by s_memtime.  Output: ns and cycles per wave-instruction PER SIMD (the issue interval the roofline of bench.py uses).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
template <int KIND>
__global__ __launch_bounds__(1024) void k(unsigned* out, long long* cyc, int iters, unsigned seed)
{
  unsigned a[16], b = seed + threadIdx.x, c = seed * 3 + threadIdx.x;
  for (int i = 0; i < 16; i++) a[i] = threadIdx.x + i;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++)
  {
#define OP(i)                                                                                       \
    if (KIND == 0) asm volatile("v_sad_u16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));           \
    else if (KIND == 1) asm volatile("v_alignbit_b32 %0, %1, %0, 16" : "+v"(a[i]) : "v"(b));         \
    else if (KIND == 2) asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b));              \
    else if (KIND == 3) asm volatile("v_pk_sub_i16 %0, %1, %0" : "+v"(a[i]) : "v"(b));               \
    else if (KIND == 4) asm volatile("v_sad_u16 %0, %1, %2, %0" : "+v"(a[i]) : "s"(seed), "v"(c));    \
    else if (KIND == 5) asm volatile("v_add_u32_e64 %0, %1, %0" : "+v"(a[i]) : "v"(b));              \
    else if (KIND == 6) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));        \
    else if (KIND == 7) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(a[i]) : "s"(seed), "v"(c));    \
    else if (KIND == 8) { if ((i) & 1) asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b)); else asm volatile("v_sad_u16 %0, %1, %2, %0" : "+v"(a[i]) : "s"(seed), "v"(c)); } \
    else if (KIND == 9) asm volatile("v_max_u32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b));              \
    else if (KIND == 10) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));   \
    else if (KIND == 11) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));  \
    else if (KIND == 12) asm volatile("v_pk_mad_i16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));    \
    else if (KIND == 13) asm volatile("v_and_b32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b));             \
    else if (KIND == 14) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b)); \
    else if (KIND == 15) asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(*(unsigned long long*)&a[(i) & 14]) : "v"(*(unsigned long long*)&a[((i) + 2) & 14]), "v"(c));
    REP16(OP) REP16(OP) REP16(OP) REP16(OP)
#undef OP
  }
  long long t1 = __builtin_readcyclecounter();
  unsigned s = 0;
  for (int i = 0; i < 16; i++) s ^= a[i];
  out[blockIdx.x * 1024 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int KIND> void run(const char* name, const char* enc, unsigned* out, long long* cyc)
{
  // grid 256 x 8: every CU busy (the clock the chip holds under load), waves per SIMD = threads / 256 (x2 for the last line if both fit)
  const int cfgT[6] = { 64, 256, 512, 1024, 1024, 1024 }, cfgG[6] = { 1, 1, 1, 1, 2, 512 };
  for (int ci = 0; ci < 6; ci++)
  {
    const int threads = cfgT[ci], grid = cfgG[ci];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(threads), 0, 0, out, cyc, 2000, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(threads), 0, 0, out, cyc, 20000, 12345u);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n = 20000.0 * 64;                       // wave-instructions per wave
    const int wavesPerSimd = threads >= 256 ? threads / 256 : 1;
    const int wgPerCu = grid == 512 ? 2 : 1;             // 512 workgroups of 1024 on 256 CUs: two rounds or two residents
    const double perWaveNs = ms * 1e6 / n / (grid == 512 ? 1.0 : 1.0);
    printf("%-18s %-6s wg %4d x %4d thr (%d waves/SIMD): wall %.3f ms = %.2f ns per wave-instr per wave; counter ticks per instr per wave %.3f -> per SIMD %.3f ticks\n",
           name, enc, grid, threads, wavesPerSimd, ms, perWaveNs, c / n, c / n / wavesPerSimd);
    (void)wgPerCu;
  }
}
int main()
{
  unsigned* out; long long* cyc;
  hipMalloc(&out, 4096 * 1024); hipMalloc(&cyc, 8);
  run<2>("v_add_u32", "VOP2", out, cyc);
  run<5>("v_add_u32_e64", "VOP3", out, cyc);
  run<9>("v_max_u32", "VOP2", out, cyc);
  run<13>("v_and_b32", "VOP2", out, cyc);
  run<0>("v_sad_u16", "VOP3", out, cyc);
  run<4>("v_sad_u16 sgpr", "VOP3", out, cyc);
  run<8>("sad_u16+add_u32 mix", "mix", out, cyc);
  run<7>("v_bfi_b32 sgpr", "VOP3", out, cyc);
  run<1>("v_alignbit_b32", "VOP3", out, cyc);
  run<10>("v_mad_u32_u24", "VOP3", out, cyc);
  run<3>("v_pk_sub_i16", "VOP3P", out, cyc);
  run<12>("v_pk_mad_i16", "VOP3P", out, cyc);
  run<11>("v_dot2_i32_i16", "VOP3P", out, cyc);
  run<6>("v_sad_u8", "VOP3", out, cyc);
  run<14>("v_mov_b32_dpp", "DPP", out, cyc);
  run<15>("v_qsad_pk_u16_u8", "VOP3", out, cyc);
  return 0;
}
