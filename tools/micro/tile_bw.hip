// tile_bw.hip -- how fast a picture pass can move a 4K plane on gfx950, as a function of the tile shape a workgroup walks.
// Every in-loop kernel of this library (deblock, SAO, ALF) and the window staging of the searches sits at ~2 TB/s for ~50 MB; this probe separates
// "the access pattern" from "the kernel": plain read + write of 16-bit samples, 16 bytes per lane and access, one workgroup (256 threads) per tile
// of TW x TH samples, tiles in raster order or dealt XCD-contiguous.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/tile_bw tools/micro/tile_bw.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int ROWS_PER_ITER>
__global__ __launch_bounds__(256) void copy_tiles(const uint4* __restrict__ src, uint4* __restrict__ dst, int pitch16, int tw16, int th, int tilesX, int nTiles, int xcd)
{
  int b = blockIdx.x;
  if (xcd) { const int chunk = (nTiles + 7) >> 3; b = (b & 7) * chunk + (b >> 3); if (b >= nTiles) return; }
  const int ty = b / tilesX, tx = b - ty * tilesX;
  const size_t base = (size_t)ty * th * pitch16 + (size_t)tx * tw16;
  const int n = tw16 * th;
  // every thread issues all its loads before its first store (as the library's kernels do)
  uint4 v[ROWS_PER_ITER];
  for (int i0 = threadIdx.x; i0 < n; i0 += 256 * ROWS_PER_ITER)
  {
#pragma unroll
    for (int u = 0; u < ROWS_PER_ITER; u++)
    {
      const int i = i0 + u * 256;
      if (i < n) { const int r = i / tw16, c = i - r * tw16; v[u] = src[base + (size_t)r * pitch16 + c]; }
    }
#pragma unroll
    for (int u = 0; u < ROWS_PER_ITER; u++)
    {
      const int i = i0 + u * 256;
      if (i < n) { const int r = i / tw16, c = i - r * tw16; uint4 q = v[u]; q.x ^= 1u; dst[base + (size_t)r * pitch16 + c] = q; }
    }
  }
}

// the access pattern of the SAO pass: a thread owns 8 samples x R rows, loads rows y-1 .. y+R (16 bytes each) and the two halo samples of every row
// (2 bytes each, HALO != 0), stores R rows; a wave covers 512 samples x R rows (WIDE) or 64 samples x 8 R rows
template <int R, int HALO, int WIDE>
__global__ __launch_bounds__(256) void sao_like(const unsigned short* __restrict__ src, unsigned short* __restrict__ dst, int w, int h)
{
  const int wv = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int colsPerWave = WIDE ? 512 : 64, rowsPerWave = WIDE ? R : 8 * R;
  const int wavesX = w / colsPerWave;
  const int wx = wv % wavesX, wy = wv / wavesX;
  const int x = wx * colsPerWave + (WIDE ? lane : (lane & 7)) * 8, y = wy * rowsPerWave + (WIDE ? 0 : (lane >> 3) * R);
  if (y >= h) return;
  uint4 v[R + 2]; unsigned short hl[R + 2], hr[R + 2];
#pragma unroll
  for (int i = 0; i < R + 2; i++)
  {
    const int yy = min(max(y - 1 + i, 0), h - 1);
    v[i] = *reinterpret_cast<const uint4*>(src + (size_t)yy * w + x);
    if (HALO) { hl[i] = src[(size_t)yy * w + max(x - 1, 0)]; hr[i] = src[(size_t)yy * w + min(x + 8, w - 1)]; }
  }
#pragma unroll
  for (int i = 0; i < R; i++)
  {
    uint4 q = v[i + 1];
    q.x ^= v[i].x ^ v[i + 2].y;
    if (HALO) q.y += hl[i + 1] + hr[i + 1];
    if (y + i < h) *reinterpret_cast<uint4*>(dst + (size_t)(y + i) * w + x) = q;
  }
}

int main()
{
  const int W = 3840, H = 2160;                       // samples (2 bytes)
  const int pitch16 = W / 8;
  void *s, *d;
  CK(hipMalloc(&s, (size_t)W * H * 2)); CK(hipMalloc(&d, (size_t)W * H * 2));
  CK(hipMemset(s, 1, (size_t)W * H * 2));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  // tiles of 2048 samples: one 16-byte access per thread (a register array filled under run-time predicates ends in scratch memory and every
  // sample then makes a round trip through it: the "register array" lines at the end show what that costs)
  const int shapes[][2] = { { 128, 16 }, { 256, 8 }, { 64, 32 }, { 32, 64 }, { 16, 128 }, { 1280, 1 } };
  for (auto& sh : shapes)
    for (int xcd = 0; xcd < 2; xcd++)
    {
      const int tw = sh[0], th = sh[1];
      if (W % tw || H % th) { if ((H % th) != 0) { /* ragged rows: skip the tail */ } }
      const int tilesX = W / tw, tilesY = H / th, nTiles = tilesX * tilesY;
      const int grid = xcd ? ((nTiles + 7) / 8) * 8 : nTiles;
      auto run = [&]() { hipLaunchKernelGGL(copy_tiles<1>, dim3(grid), dim3(256), 0, 0, (const uint4*)s, (uint4*)d, pitch16, tw / 8, th, tilesX, nTiles, xcd); };
      for (int i = 0; i < 3; i++) run();
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(a));
      const int reps = 20;
      for (int i = 0; i < reps; i++) run();
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
      const double bytes = 2.0 * tilesX * tw * (double)tilesY * th * 2;
      printf("tile %4d x %3d  %s  %6d workgroups: %.1f us  %.2f TB/s (read + write)\n", tw, th, xcd ? "XCD-contiguous" : "raster order  ", nTiles, ms * 1e3, bytes / ms / 1e9);
    }
  {
    auto timeit = [&](const char* name, auto launch)
    {
      for (int i = 0; i < 3; i++) launch();
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(a));
      for (int i = 0; i < 20; i++) launch();
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 20;
      printf("%s: %.1f us  %.2f TB/s (plane read once + written once)\n", name, ms * 1e3, 2.0 * W * H * 2 / ms / 1e9);
    };
#define SL(R_, HALO_, WIDE_) timeit("SAO-like pass, " #R_ " rows per thread, halo loads " #HALO_ ", wide waves " #WIDE_, [&]() { \
      const int waves = (W / (WIDE_ ? 512 : 64)) * ((H + (WIDE_ ? R_ : 8 * R_) - 1) / (WIDE_ ? R_ : 8 * R_)); \
      hipLaunchKernelGGL((sao_like<R_, HALO_, WIDE_>), dim3((waves + 3) / 4), dim3(256), 0, 0, (const unsigned short*)s, (unsigned short*)d, W, H); })
    SL(4, 1, 0); SL(4, 0, 0); SL(4, 1, 1); SL(4, 0, 1); SL(1, 0, 0); SL(1, 0, 1); SL(2, 0, 1); SL(8, 0, 1); SL(2, 1, 1);
#undef SL
  }
  // the same copy (tile 128 x 16) on planes of growing size, and reading only: where the rate stops being a property of the launch
  const int sizes[][2] = { { 1920, 1080 }, { 3840, 2160 }, { 7680, 4320 }, { 15360, 8640 } };
  for (auto& sz : sizes)
  {
    const int w = sz[0], h = sz[1], tw = 128, th = 16, tilesX = w / tw, tilesY = h / th, nTiles = tilesX * tilesY;
    void *s2, *d2;
    CK(hipMalloc(&s2, (size_t)w * h * 2)); CK(hipMalloc(&d2, (size_t)w * h * 2));
    CK(hipMemset(s2, 1, (size_t)w * h * 2));
    for (int deep = 0; deep < 2; deep++)
    {
      auto run = [&]()
      {
        if (deep) hipLaunchKernelGGL(copy_tiles<8>, dim3(nTiles), dim3(256), 0, 0, (const uint4*)s2, (uint4*)d2, w / 8, tw / 8, th, tilesX, nTiles, 0);
        else      hipLaunchKernelGGL(copy_tiles<1>, dim3(nTiles), dim3(256), 0, 0, (const uint4*)s2, (uint4*)d2, w / 8, tw / 8, th, tilesX, nTiles, 0);
      };
      for (int i = 0; i < 3; i++) run();
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(a));
      const int reps = 10;
      for (int i = 0; i < reps; i++) run();
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
      const double bytes = 2.0 * (double)tilesX * tw * tilesY * th * 2;
      printf("plane %5d x %4d, tile 128 x 16, %s: %.1f us  %.2f TB/s (read + write)\n", w, h, deep ? "register array under predicates (scratch)" : "one access per thread                    ", ms * 1e3, bytes / ms / 1e9);
    }
    CK(hipFree(s2)); CK(hipFree(d2));
  }
  return 0;
}
