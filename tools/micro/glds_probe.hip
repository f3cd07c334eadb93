// micro-test: LDS-DMA (global_load_lds_dwordx4 through __builtin_amdgcn_global_load_lds) on gfx950.
//  (1) layout: LDS destination = wave-uniform base + 16 * lane id; lanes masked off by EXEC leave their 16 bytes untouched;
//  (2) rate: one 1024-thread workgroup per CU streams 592-byte rows of a picture-like buffer into a 150 KB LDS ring, rows shared between
//      neighbouring workgroups (L2 hits) -- the fill pattern of the ring raster kernel (csrc/raster7.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* gbl_ptr;
__device__ __forceinline__ void glds16(const void* g, void* l)
{
  __builtin_amdgcn_global_load_lds((gbl_ptr)g, (lds_ptr)l, 16, 0, 0);
}
__global__ __launch_bounds__(64) void layout_kernel(const unsigned* __restrict__ src, unsigned* __restrict__ dst, unsigned long long mask)
{
  extern __shared__ __align__(16) unsigned lds[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 512; i += 64) lds[i] = 0xDEAD0000u + i;
  __syncthreads();
  // lane l reads the 16 bytes at src + 4 * perm(l) dwords (a per-lane gather), destination base = lds + 64 dwords
  const int perm = (lane * 7) & 63;
  if ((mask >> lane) & 1) glds16(src + 4 * perm, lds + 64);
  __builtin_amdgcn_s_waitcnt(0x0F70);                               // vmcnt(0)
  __syncthreads();
  for (int i = lane; i < 512; i += 64) dst[i] = lds[i];
}
__global__ __launch_bounds__(1024) void rate_kernel(const unsigned char* __restrict__ pic, size_t rowBytes, int rowsPerStep, int steps, int ringRows,
                                                    unsigned* __restrict__ sink, int colStrideBytes)
{
  extern __shared__ __align__(16) unsigned char ring[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned char* base = pic + (size_t)(blockIdx.x % 120) * colStrideBytes + (size_t)(blockIdx.x / 120) * 700 * rowBytes;
  int top = 0;
  unsigned acc = 0;
  for (int s = 0; s < steps; s++)
  {
    const int nPieces = rowsPerStep * 37;
    for (int c = wave; c * 64 < nPieces; c += 16)
    {
      const int pi = c * 64 + lane;
      const int r = (int)__umulhi((unsigned)pi, 116080198u);      // pi / 37 (pi < 2^16)
      const int col = pi - 37 * r;
      int rr = top + r; if (rr >= ringRows) rr -= ringRows;
      (void)rr;
      if (pi < nPieces) glds16(base + (size_t)(s * rowsPerStep + r) * rowBytes + col * 16, ring + (size_t)((top + (c * 64) / 37) % ringRows) * 592 + ((c * 64) % 37) * 16);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    acc += reinterpret_cast<unsigned*>(ring)[(top * 148 + tid) % (ringRows * 148)];
    top += rowsPerStep; if (top >= ringRows) top -= ringRows;
    __syncthreads();
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
int main()
{
  unsigned *src, *dst;
  hipMalloc(&src, 4096); hipMalloc(&dst, 4096);
  std::vector<unsigned> h(1024);
  for (int i = 0; i < 1024; i++) h[i] = 0x10000u + i;
  hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice);
  for (unsigned long long mask : { ~0ull, 0x00FF00FF0F0F3355ull })
  {
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 2048, 0, src, dst, mask);
    hipDeviceSynchronize();
    std::vector<unsigned> o(512);
    hipMemcpy(o.data(), dst, 2048, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 512; i++)
    {
      unsigned want = 0xDEAD0000u + i;
      if (i >= 64 && i < 64 + 256)
      {
        const int l = (i - 64) >> 2, perm = (l * 7) & 63;
        if ((mask >> l) & 1) want = 0x10000u + 4 * perm + ((i - 64) & 3);
      }
      if (o[i] != want) { if (bad < 8) printf("  mismatch at dword %d: got %08x want %08x\n", i, o[i], want); bad++; }
    }
    printf("layout test mask %016llx: %s (%d mismatches): destination = base + 16 * lane id, masked lanes untouched\n", mask, bad ? "FAIL" : "ok", bad);
  }
  // rate: picture 4096 x 2400 samples (2 bytes), workgroup b streams rows of a 592-byte column band
  const size_t rowBytes = 8192, rows = 2400;
  unsigned char* pic; unsigned* sink;
  hipMalloc(&pic, rowBytes * rows); hipMalloc(&sink, 64);
  hipMemset(pic, 1, rowBytes * rows);
  hipFuncSetAttribute(reinterpret_cast<const void*>(rate_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 254 * 592 + 2048);
  for (int rps : { 32, 64, 222 })
  {
    const int steps = 600 / rps * 1;
    for (int rep = 0; rep < 2; rep++)
    {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(rate_kernel, dim3(256), dim3(1024), 254 * 592 + 2048, 0, pic, rowBytes, rps, steps, 254, sink, 64);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double bytes = 256.0 * steps * rps * 592;
      if (rep) printf("rate: %3d rows per step, %d steps, serial (wait + barrier after every step): %.3f ms, %.1f GB/s per CU, %.2f TB/s chip, %.2f us per step\n",
                      rps, steps, ms, bytes / 256 / ms * 1e-6, bytes / ms * 1e-9, ms * 1e3 / steps);
    }
  }
  return 0;
}
