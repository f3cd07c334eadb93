// sao.hip -- SAO apply (S1) for gfx950.
//
// Reference behaviour reproduced (bit-exact): SampleAdaptiveOffset::offsetBlock / offsetCTU / SAOProcess
// (CommonLib/SampleAdaptiveOffset.cpp:292-612).  The reference's running sign buffers are a CPU
// optimisation; per sample the result is  clip(c + offset[2 + sgn(c-a) + sgn(c-b)])  when both neighbours
// are available (availability of a neighbour in an adjacent CTU = that CTU-direction's flag), c otherwise.
//
// Design: pure streaming stencil.  One thread = 8 horizontally adjacent samples (one 16-byte load/store)
// x 4 rows; its six rows (one above, one below) are loaded up front into a register window, the two horizontal halo
// samples per row with 2-byte loads that hit L1/L2.  dst receives the complete picture (offset or copied
// samples), so the reference's whole-picture temp copy (:587) never touches HBM.
#include "common.h"

namespace {

// rows per thread.  4 rows (6 loaded) made 6112 waves of 94 registers for a 4K picture: ONE round of five waves per SIMD, in which every wave loads, then
// computes, then stores at the same time (19.2 us); 2 rows (4 loaded) are 12 k smaller waves in two to three rounds whose phases overlap: 17.5 us.  1 row: 19.7.
constexpr int SAO_ROWS = 2;

struct SaoGeom { int gx, gy, rows, x0, y0, x1, y1, w, h, avail, clpMin, clpMax; };

// Edge-offset body with the class direction as a compile-time constant (keeps the 3-row register window statically indexed -> no scratch).
// Neighbour a = (x+DXA, y+DYA), b = (x-DXA, y-DYA).
// Which samples take an offset depends on the thread only through a few bits: neighbour a (b) of column k lies left of / inside / right of the
// CTU -- three 8-bit column masks per thread -- and above / inside / below it per row.  They are folded into one 8-bit "use" mask per row before
// the sample loop, which is then: two differences, two v_med3 (sign), the offset by a bit-field extract from the five offsets packed as 6-bit
// fields, add, clip, and a v_bfi that keeps the sample where the mask bit is clear.  (The form this replaces evaluated the position tests and a
// compare-select sign per sample: ~40 vector instructions per sample, the kernel's arithmetic took as long as its memory traffic.)
// Offsets outside [-32, 31] (not reachable from the reference at 8 - 10 bits, but the ABI takes any int16) go through a select chain instead.
template <int DXA, int DYA, typename StoreRow>
__device__ __forceinline__ void eo_rows(const SaoGeom& g, int off0, int off1, int off2, int off3, int off4,
                                        const int (&win)[SAO_ROWS + 2][10], StoreRow store_row)
{
  // 0 / -1 masks of the neighbour CTUs' availability
  const int avL = -(g.avail & 1), avR = -((g.avail >> 1) & 1), avA = -((g.avail >> 2) & 1), avB = -((g.avail >> 3) & 1);
  const int avAL = -((g.avail >> 4) & 1), avAR = -((g.avail >> 5) & 1), avBL = -((g.avail >> 6) & 1), avBR = -((g.avail >> 7) & 1);
  // columns k whose neighbour a (b) lies left of / inside / right of the CTU
  int aL = 0, aR = 0, bL = 0, bR = 0;
#pragma unroll
  for (int k = 0; k < 8; k++)
  {
    const int x = g.gx + k;
    if (DXA != 0)
    {
      if (x + DXA < g.x0) aL |= 1 << k; else if (x + DXA >= g.x1) aR |= 1 << k;
      if (x - DXA < g.x0) bL |= 1 << k; else if (x - DXA >= g.x1) bR |= 1 << k;
    }
  }
  const int aIn = 0xFF & ~(aL | aR), bIn = 0xFF & ~(bL | bR);
  const bool fits = (unsigned)(off0 + 32) < 64u && (unsigned)(off1 + 32) < 64u && (unsigned)(off2 + 32) < 64u && (unsigned)(off3 + 32) < 64u && (unsigned)(off4 + 32) < 64u;
  const unsigned packed = (unsigned)(off0 & 63) | (unsigned)(off1 & 63) << 6 | (unsigned)(off2 & 63) << 12 | (unsigned)(off3 & 63) << 18 | (unsigned)(off4 & 63) << 24;
#pragma unroll
  for (int r = 0; r < SAO_ROWS; r++)
  {
    if (r >= g.rows) break;
    const int y = g.gy + r;
    const bool topOut = (DYA != 0) && (y - 1 < g.y0);     // a's row is in the CTU above
    const bool botOut = (DYA != 0) && (y + 1 >= g.y1);    // b's row is in the CTU below
    const int okA = topOut ? ((aIn & avA) | (aL & avAL) | (aR & avAR)) : (aIn | (aL & avL) | (aR & avR));
    const int okB = botOut ? ((bIn & avB) | (bL & avBL) | (bR & avBR)) : (bIn | (bL & avL) | (bR & avR));
    const int use = okA & okB;
    int o[8];
#pragma unroll
    for (int k = 0; k < 8; k++)
    {
      const int c = win[r + 1][1 + k];
      const int a = DYA == 0 ? win[r + 1][1 + k + DXA] : win[r][1 + k + DXA];
      const int b = DYA == 0 ? win[r + 1][1 + k - DXA] : win[r + 2][1 + k - DXA];
      int sa, sb;
      asm("v_med3_i32 %0, %1, -1, 1" : "=v"(sa) : "v"(c - a));
      asm("v_med3_i32 %0, %1, -1, 1" : "=v"(sb) : "v"(c - b));
      const int e = sa + sb;
      int of;
      if (fits) of = __builtin_amdgcn_sbfe((int)packed, (unsigned)(e * 6 + 12), 6u);
      else      of = e == -2 ? off0 : e == -1 ? off1 : e == 0 ? off2 : e == 1 ? off3 : off4;
      const int t = clip3(g.clpMin, g.clpMax, c + of);
      const int m = __builtin_amdgcn_sbfe(use, (unsigned)k, 1u);            // -1: this sample takes the offset
      o[k] = (t & m) | (c & ~m);
    }
    store_row(y, o);
  }
}

__device__ __forceinline__ void sao_apply_body(const int bidx, const Pel* __restrict__ src, int sstride,
                                                        Pel* __restrict__ dst, int dstride, int w, int h,
                                                        int ctuW, int ctuH, int wCtu, int boShift,
                                                        const vvcgpu_sao_ctu* __restrict__ params,
                                                        int clpMin, int clpMax, int tpcShift, int nWaveCols)
{
  // a wave covers (8 << tpcShift) samples x (64 >> tpcShift) thread rows: with a power-of-two CTU width that is exactly
  // one CTU column, so the SAO type is wave-uniform and only one of the five type bodies runs (a wave that straddles
  // several CTUs executes them one after the other)
  const int wv = bidx * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wcol = wv % nWaveCols, wrow = wv / nWaveCols;
  const int gx = ((wcol << tpcShift) + (lane & ((1 << tpcShift) - 1))) * 8;           // first sample of this thread
  const int gy = (wrow * (64 >> tpcShift) + (lane >> tpcShift)) * SAO_ROWS;           // first row
  if (gx >= w || gy >= h) return;
  const int n = min(8, w - gx);                                 // valid samples (w need not be a multiple of 8)
  const int cx = gx / ctuW, cy = gy / ctuH;
  const vvcgpu_sao_ctu* prm = params + cy * wCtu + cx;
  const int type = prm->type;
  const int avail = prm->avail;
  const int x0 = cx * ctuW, y0 = cy * ctuH;
  const int x1 = min(x0 + ctuW, w), y1 = min(y0 + ctuH, h);
  const bool vec = (n == 8) && ((sstride & 7) == 0) && ((dstride & 7) == 0) &&
                   ((reinterpret_cast<uintptr_t>(src) & 15) == 0) && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0);

  // all SAO_ROWS + 2 rows of the thread (gy-1 .. gy+SAO_ROWS, clamped to the picture) are loaded up front: one exposure to
  // memory latency instead of one per row (a chroma plane is only ~1 workgroup per CU, nothing else hides it)
  int win[SAO_ROWS + 2][10];                        // [row][samples gx-1 .. gx+8]
#pragma unroll
  for (int i = 0; i < SAO_ROWS + 2; i++)
  {
    const int y = min(max(gy - 1 + i, 0), h - 1);
    const Pel* row = src + (size_t)y * sstride;
    if (vec)
    {
      const pel8 v = *reinterpret_cast<const pel8*>(row + gx);
#pragma unroll
      for (int k = 0; k < 8; k++) win[i][1 + k] = v[k];
    }
    else
    {
#pragma unroll
      for (int k = 0; k < 8; k++) win[i][1 + k] = row[min(gx + k, w - 1)];
    }
    win[i][0] = row[max(gx - 1, 0)];
    win[i][9] = row[min(gx + 8, w - 1)];
  }
  auto store_row = [&](int y, const int* o) {
    Pel* row = dst + (size_t)y * dstride;
    if (vec)
    {
      pel8 v;
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = (short)o[k];
      *reinterpret_cast<pel8*>(row + gx) = v;
    }
    else
    {
#pragma unroll
      for (int k = 0; k < 8; k++) if (k < n) row[gx + k] = (short)o[k];
    }
  };

  const int rows = min(SAO_ROWS, h - gy);
  if (type < 0)
  {
#pragma unroll
    for (int r = 0; r < SAO_ROWS; r++) if (r < rows) store_row(gy + r, &win[r + 1][1]);
    return;
  }
  if (type == 4)
  {
    const int16_t* off = prm->offset;
#pragma unroll
    for (int r = 0; r < SAO_ROWS; r++)
    {
      if (r >= rows) break;
      int o[8];
#pragma unroll
      for (int k = 0; k < 8; k++) o[k] = clip3(clpMin, clpMax, win[r + 1][1 + k] + off[win[r + 1][1 + k] >> boShift]);
      store_row(gy + r, o);
    }
    return;
  }
  const int16_t* op = prm->offset;
  SaoGeom g{ gx, gy, rows, x0, y0, x1, y1, w, h, avail, clpMin, clpMax };
  const int off0 = op[0], off1 = op[1], off2 = op[2], off3 = op[3], off4 = op[4];
  switch (type)
  {
  case 0:  eo_rows<-1, 0>(g, off0, off1, off2, off3, off4, win, store_row); break;
  case 1:  eo_rows<0, -1>(g, off0, off1, off2, off3, off4, win, store_row); break;
  case 2:  eo_rows<-1, -1>(g, off0, off1, off2, off3, off4, win, store_row); break;
  default: eo_rows<1, -1>(g, off0, off1, off2, off3, off4, win, store_row); break;
  }
}

__global__ __launch_bounds__(256) void sao_apply_kernel(const Pel* __restrict__ src, int sstride, Pel* __restrict__ dst, int dstride, int w, int h,
                                                        int ctuW, int ctuH, int wCtu, int boShift, const vvcgpu_sao_ctu* __restrict__ params,
                                                        int clpMin, int clpMax, int tpcShift, int nWaveCols)
{
  sao_apply_body((int)blockIdx.x, src, sstride, dst, dstride, w, h, ctuW, ctuH, wCtu, boShift, params, clpMin, clpMax, tpcShift, nWaveCols);
}

// the three planes of a picture in one launch (a chroma plane alone is ~1 workgroup per CU: its own launch costs a latency floor)
struct SaoPlaneArgs { const Pel* src; Pel* dst; const vvcgpu_sao_ctu* params; int sstride, dstride, w, h, ctuW, ctuH, wCtu, tpcShift, nWaveCols, wgEnd; };
struct SaoApply3 { SaoPlaneArgs a[3]; int boShift, clpMin, clpMax, total, xcd; };
__global__ __launch_bounds__(256) void sao_apply_picture_kernel(SaoApply3 p)
{
  const int b = vvc_xcd_index2((int)blockIdx.x, p.a[0].wgEnd, p.total, p.xcd);
  if (b < 0) return;
  const int c = b < p.a[0].wgEnd ? 0 : b < p.a[1].wgEnd ? 1 : 2;
  const SaoPlaneArgs& a = c == 0 ? p.a[0] : c == 1 ? p.a[1] : p.a[2];
  const int first = c == 0 ? 0 : c == 1 ? p.a[0].wgEnd : p.a[1].wgEnd;
  sao_apply_body(b - first, a.src, a.sstride, a.dst, a.dstride, a.w, a.h, a.ctuW, a.ctuH, a.wCtu, p.boShift, a.params, p.clpMin, p.clpMax, a.tpcShift,
                 a.nWaveCols);
}


}  // namespace

extern "C" int vvcgpu_sao_apply(const vvc_pel* src, int src_stride, vvc_pel* dst, int dst_stride,
                                int width, int height, int ctu_w, int ctu_h, int bit_depth,
                                const vvcgpu_sao_ctu* params, int clp_min, int clp_max, void* stream)
{
  VVC_CHECK_ARG(src && dst && params, "sao_apply: null pointer");
  VVC_CHECK_ARG(src != dst, "sao_apply: src must not alias dst (reference reads the deblocked copy)");
  VVC_CHECK_ARG(width > 0 && height > 0, "sao_apply: bad size %dx%d", width, height);
  VVC_CHECK_ARG(src_stride >= width && dst_stride >= width, "sao_apply: stride < width");
  VVC_CHECK_ARG(ctu_w >= 8 && ctu_h >= 4 && (ctu_w & 7) == 0 && (ctu_h & 3) == 0,
                "sao_apply: CTU %dx%d must be a multiple of 8x4", ctu_w, ctu_h);
  VVC_CHECK_ARG(bit_depth >= 8 && bit_depth <= 10, "sao_apply: bit depth %d outside 8..10", bit_depth);
  int tpcShift = 6;                                                    // threads across a wave: one CTU width when that is a power of two
  if ((ctu_w & (ctu_w - 1)) == 0 && ctu_w <= 512) { tpcShift = 0; while ((8 << tpcShift) < ctu_w) tpcShift++; }
  const int nWaveCols = cdiv(width, 8 << tpcShift), nWaveRows = cdiv(height, (64 >> tpcShift) * SAO_ROWS);
  hipLaunchKernelGGL(sao_apply_kernel, dim3(cdiv(nWaveCols * nWaveRows, 4)), dim3(256), 0, (hipStream_t)stream, src, src_stride, dst,
                     dst_stride, width, height, ctu_w, ctu_h, cdiv(width, ctu_w), bit_depth - 5, params, clp_min, clp_max, tpcShift,
                     nWaveCols);
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}

extern "C" int vvcgpu_sao_apply_picture(const vvcgpu_planes* src, const vvcgpu_planes* dst, int width, int height, int ctu_size, int bit_depth,
                                        const vvcgpu_sao_ctu* params_y, const vvcgpu_sao_ctu* params_cb, const vvcgpu_sao_ctu* params_cr,
                                        int clp_min, int clp_max, void* stream)
{
  VVC_CHECK_ARG(src && dst && params_y && params_cb && params_cr, "sao_apply_picture: null pointer");
  VVC_CHECK_ARG(width > 0 && height > 0 && (width & 1) == 0 && (height & 1) == 0, "sao_apply_picture: bad size %dx%d", width, height);
  VVC_CHECK_ARG(ctu_size >= 16 && (ctu_size & 15) == 0, "sao_apply_picture: CTU size %d", ctu_size);
  VVC_CHECK_ARG(bit_depth >= 8 && bit_depth <= 10, "sao_apply_picture: bit depth %d outside 8..10", bit_depth);
  SaoApply3 p;
  const vvcgpu_sao_ctu* prm[3] = { params_y, params_cb, params_cr };
  int end = 0;
  for (int c = 0; c < 3; c++)
  {
    const int w = c ? width >> 1 : width, h = c ? height >> 1 : height, ctu = c ? ctu_size >> 1 : ctu_size;
    VVC_CHECK_ARG(src->p[c] && dst->p[c] && src->p[c] != dst->p[c] && src->stride[c] >= w && dst->stride[c] >= w, "sao_apply_picture: plane %d", c);
    int tpcShift = 6;
    if ((ctu & (ctu - 1)) == 0 && ctu <= 512) { tpcShift = 0; while ((8 << tpcShift) < ctu) tpcShift++; }
    const int nWaveCols = cdiv(w, 8 << tpcShift), nWaveRows = cdiv(h, (64 >> tpcShift) * SAO_ROWS);
    end += cdiv(nWaveCols * nWaveRows, 4);
    p.a[c] = SaoPlaneArgs{ src->p[c], dst->p[c], prm[c], src->stride[c], dst->stride[c], w, h, ctu, ctu, cdiv(w, ctu), tpcShift, nWaveCols, end };
  }
  p.boShift = bit_depth - 5; p.clpMin = clp_min; p.clpMax = clp_max;
  p.total = end; p.xcd = vvc_xcd_on();
  hipLaunchKernelGGL(sao_apply_picture_kernel, dim3(vvc_xcd_grid2(p.a[0].wgEnd, end, p.xcd)), dim3(256), 0, (hipStream_t)stream, p);
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}
