// micro-test: issue rate of v_sad_u16 / v_alignbit_b32 / v_add_u32 / v_pk_sub_i16 on gfx950 (one wave, then 4 waves on one SIMD each
// -- 256 threads = one wave per SIMD).  16 independent chains per instruction kind, wall time by s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
template <int KIND>
__global__ void k(unsigned* out, long long* cyc, int iters, unsigned seed)
{
  unsigned a[16], b = seed + threadIdx.x, c = seed * 3 + threadIdx.x;
  for (int i = 0; i < 16; i++) a[i] = threadIdx.x + i;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++)
  {
#define OP(i)                                                                                       \
    if (KIND == 0) asm volatile("v_sad_u16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));           \
    else if (KIND == 1) asm volatile("v_alignbit_b32 %0, %1, %0, 16" : "+v"(a[i]) : "v"(b));         \
    else if (KIND == 2) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));                  \
    else if (KIND == 3) asm volatile("v_pk_sub_i16 %0, %1, %0" : "+v"(a[i]) : "v"(b));               \
    else if (KIND == 4) asm volatile("v_sad_u16 %0, %1, %2, %0" : "+v"(a[i]) : "s"(seed), "v"(c));    \
    else if (KIND == 5) asm volatile("v_sad_u32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));       \
    else if (KIND == 6) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
    REP16(OP) REP16(OP) REP16(OP) REP16(OP)
#undef OP
  }
  long long t1 = __builtin_readcyclecounter();
  unsigned s = 0;
  for (int i = 0; i < 16; i++) s ^= a[i];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int KIND> void run(const char* name, unsigned* out, long long* cyc)
{
  for (int threads = 64; threads <= 512; threads *= 2)
  {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, out, cyc, 2000, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, out, cyc, 20000, 12345u);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n = 20000.0 * 64;
    printf("%-16s %3d threads: %.2f ns per wave-instruction (wall %.3f ms), counter ticks per instr %.3f\n", name, threads, ms * 1e6 / n, ms, c / n);
  }
}
int main()
{
  unsigned* out; long long* cyc;
  hipMalloc(&out, 4096); hipMalloc(&cyc, 8);
  run<2>("v_add_u32", out, cyc);
  run<0>("v_sad_u16", out, cyc);
  run<4>("v_sad_u16 sgpr", out, cyc);
  run<1>("v_alignbit_b32", out, cyc);
  run<3>("v_pk_sub_i16", out, cyc);
  run<5>("v_sad_u32", out, cyc);
  run<6>("v_sad_u8", out, cyc);
  return 0;
}
