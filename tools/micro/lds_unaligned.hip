// RESULT (MI355X): correct data, but ~9x slower than the aligned read (112k vs 1024k clocks for 4000 reads): not usable to
// replace v_alignbit in the raster kernel.
// micro-test: ds_read_b32 at a 2-byte-aligned LDS address -- correct? how fast?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned* out, long long* cyc, int iters, int misalign)
{
  __shared__ __align__(16) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (unsigned short)i;
  __syncthreads();
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned short*)lds + threadIdx.x * 8u + (misalign ? 2u : 0u);
  unsigned acc = 0;
  long long t0 = clock64();
  for (int it = 0; it < iters; it++)
  {
    unsigned v0, v1, v2, v3;
    asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:16\n\tds_read_b32 %2, %4 offset:32\n\tds_read_b32 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(base) : "memory");
    acc += v0 ^ v1 ^ v2 ^ v3;
  }
  long long t1 = clock64();
  out[threadIdx.x] = acc;
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; unsigned v; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(base) : "memory"); out[256] = v; }
}
int main()
{
  unsigned* out; long long* cyc;
  hipMalloc(&out, 4096); hipMalloc(&cyc, 8);
  for (int mis = 0; mis < 2; mis++)
  {
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, out, cyc, 1000, mis);
    hipDeviceSynchronize();
    long long c; unsigned v;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    hipMemcpy(&v, out + 256, 4, hipMemcpyDeviceToHost);
    printf("misalign=%d: %lld clocks for 1000 x 4 reads (256 threads), lane0 value 0x%08x (expect %s)\n", mis, c, v, mis ? "0x00020001" : "0x00010000");
  }
  return 0;
}
