// micro-test: what one SIMD sustains on the raster kernels' stage body (csrc/raster7.hip, csrc/dist.hip quad form) -- per stage 8 ds_read_b64 of a
// window span + one 64-byte scalar load of the packed original row + 32 v_sad_u16 (four accumulators) + 2 v_bfi_b32, software-pipelined one
// stage ahead -- with 4 waves per SIMD (one 1024-thread workgroup per CU, all 256 CUs busy).  Variants switch parts off:
//   bit 0: no LDS reads   bit 1: no scalar load   bit 2: accumulators interleaved (no 8-deep dependent chains)   bit 3: VGPR original (no SGPR operand)
// Prints ns per stage and per vector instruction per SIMD, and the shader clock during the loop (s_memtime / s_memrealtime, 100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int V>
__global__ __launch_bounds__(1024, 4) void k(const unsigned* __restrict__ org, unsigned* __restrict__ out, unsigned long long* __restrict__ clk, int stages)
{
  extern __shared__ __align__(16) unsigned lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 32768; i += 1024) lds[i] = i * 2654435761u;
  __syncthreads();
  // lane -> unit u: 8-byte slot 5 u (mod 32) as in the kernels (row pitch 148 dwords, 10 quads per row)
  const int u = lane + 64 * ((tid >> 6) & 3);
  const unsigned a0 = (unsigned)((u / 10) * 5 * 592 + (u % 10) * 40);
  unsigned acc[4] = { 0, 0, 0, 0 };
  unsigned long long d[2][8];
  unsigned o[2][16];
  const unsigned* op = org + (blockIdx.x & 63) * 4096;
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  auto issue = [&](int b, int s)
  {
    const unsigned a = a0 + (unsigned)((s & 31) * 592 * 2);
    if (!(V & 1))
      asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:8\n\tds_read_b64 %2, %8 offset:16\n\tds_read_b64 %3, %8 offset:24\n\t"
                   "ds_read_b64 %4, %8 offset:32\n\tds_read_b64 %5, %8 offset:40\n\tds_read_b64 %6, %8 offset:48\n\tds_read_b64 %7, %8 offset:56"
                   : "=&v"(d[b][0]), "=&v"(d[b][1]), "=&v"(d[b][2]), "=&v"(d[b][3]), "=&v"(d[b][4]), "=&v"(d[b][5]), "=&v"(d[b][6]), "=&v"(d[b][7]) : "v"(a) : "memory");
    if (!(V & 2))
    {
#pragma unroll
      for (int k = 0; k < 16; k++) o[b][k] = op[(s & 63) * 16 + k];
    }
  };
  auto compute = [&](int b)
  {
    unsigned dd[16];
#pragma unroll
    for (int k = 0; k < 8; k++) { asm volatile("" :: "v"(d[b][k])); dd[2 * k] = (unsigned)d[b][k]; dd[2 * k + 1] = (unsigned)(d[b][k] >> 32); }
    if (V & 8)
    {
      unsigned vo[16];
#pragma unroll
      for (int k = 0; k < 16; k++) { vo[k] = o[b][k]; asm volatile("v_mov_b32 %0, %0" : "+v"(vo[k])); }
#pragma unroll
      for (int m = 0; m < 4; m++)
#pragma unroll
        for (int k = 0; k < 8; k++) acc[m] = __builtin_amdgcn_sad_u16(vo[(m & 1) * 8 + k], dd[2 * m + k], acc[m]);
    }
    else if (V & 4)
    {
#pragma unroll
      for (int k = 0; k < 8; k++)
#pragma unroll
        for (int m = 0; m < 4; m++) acc[m] = __builtin_amdgcn_sad_u16(o[b][(m & 1) * 8 + k], dd[2 * m + k], acc[m]);
    }
    else
    {
#pragma unroll
      for (int m = 0; m < 4; m++)
#pragma unroll
        for (int k = 0; k < 8; k++) acc[m] = __builtin_amdgcn_sad_u16(o[b][(m & 1) * 8 + k], dd[2 * m + k], acc[m]);
    }
    acc[1] = __builtin_amdgcn_sad_u16(o[b][3], (dd[9] & 0xFFFFu) | (dd[1] & 0xFFFF0000u), acc[1]);
    acc[3] = __builtin_amdgcn_sad_u16(o[b][5], (dd[11] & 0xFFFFu) | (dd[3] & 0xFFFF0000u), acc[3]);
  };
  if (V & 2) { for (int k = 0; k < 16; k++) { o[0][k] = op[k]; o[1][k] = op[16 + k]; } }
  if (V & 1) { for (int k = 0; k < 8; k++) { d[0][k] = tid * 77 + k; d[1][k] = tid * 91 + k; } }
  issue(0, 0);
  for (int s = 0; s < stages; s += 2)
  {
    __builtin_amdgcn_s_waitcnt(0xC07F);
    issue(1, s + 1);
    __builtin_amdgcn_sched_barrier(0);
    compute(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    issue(0, s + 2);
    __builtin_amdgcn_sched_barrier(0);
    compute(1);
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 1024 + tid] = acc[0] + acc[1] + acc[2] + acc[3];
  if (tid == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}
template <int V> void run(const char* name, const unsigned* org, unsigned* out, unsigned long long* clk)
{
  const int stages = 20000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipLaunchKernelGGL(k<V>, dim3(256), dim3(1024), 131072, 0, org, out, clk, 2000);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<V>, dim3(256), dim3(1024), 131072, 0, org, out, clk, stages);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c[512]; hipMemcpy(c, clk, sizeof c, hipMemcpyDeviceToHost);
  double tc = 0, rc = 0; for (int i = 0; i < 256; i++) { tc += c[2 * i]; rc += c[2 * i + 1]; }
  const double nsStage = ms * 1e6 / stages / 4.0;                 // four waves per SIMD: ns per wave-stage per SIMD
  printf("%-58s %.1f ns per wave-stage per SIMD = %.2f ns per vector instruction (34 per stage); counter ticks / 10 ns = %.2f\n", name, nsStage, nsStage / 34.0, tc / rc);
}
int main()
{
  unsigned *org, *out; unsigned long long* clk;
  hipMalloc(&org, 64 * 4096 * 4 + 65536); hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&clk, 512 * 8);
  hipMemset(org, 0x11, 64 * 4096 * 4 + 65536);
  run<0>("full stage body", org, out, clk);
  run<1>("no LDS reads", org, out, clk);
  run<2>("no scalar load", org, out, clk);
  run<3>("no LDS reads, no scalar load (v_sad_u16 + v_bfi only)", org, out, clk);
  run<4>("accumulators interleaved", org, out, clk);
  run<7>("interleaved, no LDS reads, no scalar load", org, out, clk);
  run<8>("original rows in VGPRs", org, out, clk);
  return 0;
}
