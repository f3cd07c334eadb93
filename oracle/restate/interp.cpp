// oracle/restate/interp.cpp -- TEST INFRASTRUCTURE: CPU restatement of DCTIF interpolation, MC and PelBuffer ops.
// I1 follows InterpolationFilter::filter<N,isVertical,isFirst,isLast> (CommonLib/InterpolationFilter.cpp:290-379),
//    filterCopy<isFirst,isLast> (:205-264), tables m_lumaFilter/m_chromaFilter (:59-138).
// I3 follows InterPrediction::xPredInterBlk (CommonLib/InterPrediction.cpp:480-547) + AreaBuf::addAvg (Buffer.cpp:114-151).
// B* follow addAvgCore/reconstructCore/linTfCore (Buffer.cpp:50-94), subtract (Buffer.h:321-339),
//    removeHighFreq (Buffer.h:389-416), copyClip (Buffer.cpp:197-222).
#include "orc_common.h"
#include <vector>

static const int16_t lumaFilter[16][8] = {
  {  0, 0,   0, 64,  0,   0,  0,  0 }, {  0, 1,  -3, 63,  4,  -2,  1,  0 }, { -1, 2,  -5, 62,  8,  -3,  1,  0 },
  { -1, 3,  -8, 60, 13,  -4,  1,  0 }, { -1, 4, -10, 58, 17,  -5,  1,  0 }, { -1, 4, -11, 52, 26,  -8,  3, -1 },
  { -1, 3,  -9, 47, 31, -10,  4, -1 }, { -1, 4, -11, 45, 34, -10,  4, -1 }, { -1, 4, -11, 40, 40, -11,  4, -1 },
  { -1, 4, -10, 34, 45, -11,  4, -1 }, { -1, 4, -10, 31, 47,  -9,  3, -1 }, { -1, 3,  -8, 26, 52, -11,  4, -1 },
  {  0, 1,  -5, 17, 58, -10,  4, -1 }, {  0, 1,  -4, 13, 60,  -8,  3, -1 }, {  0, 1,  -3,  8, 62,  -5,  2, -1 },
  {  0, 1,  -2,  4, 63,  -3,  1,  0 } };
static const int16_t chromaFilter[32][4] = {
  {  0, 64,  0,  0 }, { -1, 63,  2,  0 }, { -2, 62,  4,  0 }, { -2, 60,  7, -1 }, { -2, 58, 10, -2 }, { -3, 57, 12, -2 },
  { -4, 56, 14, -2 }, { -4, 55, 15, -2 }, { -4, 54, 16, -2 }, { -5, 53, 18, -2 }, { -6, 52, 20, -2 }, { -6, 49, 24, -3 },
  { -6, 46, 28, -4 }, { -5, 44, 29, -4 }, { -4, 42, 30, -4 }, { -4, 39, 33, -4 }, { -4, 36, 36, -4 }, { -4, 33, 39, -4 },
  { -4, 30, 42, -4 }, { -4, 29, 44, -5 }, { -4, 28, 46, -6 }, { -3, 24, 49, -6 }, { -2, 20, 52, -6 }, { -2, 18, 53, -5 },
  { -2, 16, 54, -4 }, { -2, 15, 55, -4 }, { -2, 14, 56, -4 }, { -2, 12, 57, -3 }, { -2, 10, 58, -2 }, { -1,  7, 60, -2 },
  {  0,  4, 62, -2 }, {  0,  2, 63, -1 } };
ORC_API const int16_t* orc_luma_filter(int frac) { return lumaFilter[frac]; }
ORC_API const int16_t* orc_chroma_filter(int frac) { return chromaFilter[frac]; }

static const int IF_INTERNAL_PREC = 14, IF_FILTER_PREC = 6, IF_INTERNAL_OFFS = 1 << 13;

ORC_API void orc_if_filter(int N, int isVertical, int isFirst, int isLast, const Pel* src, int sstride, Pel* dst,
                           int dstride, int w, int h, const int16_t* coeff, int bd, int clpMin, int clpMax)
{
  if (N == 0)                                       // filterCopy (:205-264)
  {
    const int shift = std::max(2, IF_INTERNAL_PREC - bd);
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++)
      {
        const Pel s = src[y * sstride + x];
        if (isFirst == isLast) dst[y * dstride + x] = s;
        else if (isFirst) { const Pel v = (Pel)(s << shift); dst[y * dstride + x] = (Pel)(v - (Pel)IF_INTERNAL_OFFS); }
        else { Pel v = s; v = (Pel)((v + IF_INTERNAL_OFFS + (1 << (shift - 1))) >> shift); dst[y * dstride + x] = (Pel)clip3i(clpMin, clpMax, v); }
      }
    return;
  }
  const int cStride = isVertical ? sstride : 1;
  src -= (N / 2 - 1) * cStride;
  const int headRoom = std::max(2, IF_INTERNAL_PREC - bd);
  int shift = IF_FILTER_PREC, offset;
  if (isLast) { shift += isFirst ? 0 : headRoom; offset = 1 << (shift - 1); offset += isFirst ? 0 : IF_INTERNAL_OFFS << IF_FILTER_PREC; }
  else { shift -= isFirst ? headRoom : 0; offset = isFirst ? -IF_INTERNAL_OFFS << shift : 0; }
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++)
    {
      int sum = 0;
      for (int k = 0; k < N; k++) sum += src[y * sstride + x + k * cStride] * coeff[k];
      Pel val = (Pel)((sum + offset) >> shift);
      if (isLast) val = (Pel)clip3i(clpMin, clpMax, val);
      dst[y * dstride + x] = val;
    }
}

ORC_API int orc_if_batch(const Pel* srcBase, Pel* dstBase, const vvcgpu_if_desc* d, int n, int bd, int clpMin, int clpMax)
{
  for (int i = 0; i < n; i++)
    orc_if_filter(d[i].taps, d[i].is_vertical, d[i].is_first, d[i].is_last, srcBase + d[i].src_off, d[i].src_stride,
                  dstBase + d[i].dst_off, d[i].dst_stride, d[i].w, d[i].h, d[i].coeff, bd, clpMin, clpMax);
  return 0;
}

// xPredInterBlk (:529-546): rndRes = !bi
static void predBlk(const Pel* ref, int rs, Pel* dst, int ds, int w, int h, int xFrac, int yFrac, bool luma, bool rndRes,
                    int bd, int cmin, int cmax)
{
  const int N = luma ? 8 : 4;
  const int16_t* cx = luma ? lumaFilter[xFrac] : chromaFilter[xFrac];
  const int16_t* cy = luma ? lumaFilter[yFrac] : chromaFilter[yFrac];
  if (yFrac == 0) orc_if_filter(xFrac ? N : 0, 0, 1, rndRes, ref, rs, dst, ds, w, h, cx, bd, cmin, cmax);
  else if (xFrac == 0) orc_if_filter(N, 1, 1, rndRes, ref, rs, dst, ds, w, h, cy, bd, cmin, cmax);
  else
  {
    std::vector<Pel> tmp((size_t)w * (h + N - 1));
    orc_if_filter(N, 0, 1, 0, ref - ((N >> 1) - 1) * rs, rs, tmp.data(), w, w, h + N - 1, cx, bd, cmin, cmax);
    orc_if_filter(N, 1, 0, rndRes, tmp.data() + ((N >> 1) - 1) * w, w, dst, ds, w, h, cy, bd, cmin, cmax);
  }
}

ORC_API int orc_mc_batch(const Pel* ref0Base, const Pel* ref1Base, Pel* dstBase, const vvcgpu_mc_desc* d, int n, int bd,
                         int clpMin, int clpMax)
{
  for (int i = 0; i < n; i++)
  {
    const vvcgpu_mc_desc& m = d[i];
    Pel* dst = dstBase + m.dst_off;
    if (m.bi == 0) predBlk(ref0Base + m.ref0_off, m.ref0_stride, dst, m.dst_stride, m.w, m.h, m.frac_x0, m.frac_y0, m.is_luma, true, bd, clpMin, clpMax);
    else if (m.bi == 2) predBlk(ref0Base + m.ref0_off, m.ref0_stride, dst, m.dst_stride, m.w, m.h, m.frac_x0, m.frac_y0, m.is_luma, false, bd, clpMin, clpMax);
    else
    {
      std::vector<Pel> p0((size_t)m.w * m.h), p1((size_t)m.w * m.h);
      predBlk(ref0Base + m.ref0_off, m.ref0_stride, p0.data(), m.w, m.w, m.h, m.frac_x0, m.frac_y0, m.is_luma, false, bd, clpMin, clpMax);
      predBlk(ref1Base + m.ref1_off, m.ref1_stride, p1.data(), m.w, m.w, m.h, m.frac_x1, m.frac_y1, m.is_luma, false, bd, clpMin, clpMax);
      const int shiftNum = std::max(2, IF_INTERNAL_PREC - bd) + 1, offset = (1 << (shiftNum - 1)) + 2 * IF_INTERNAL_OFFS;
      for (int y = 0; y < m.h; y++)
        for (int x = 0; x < m.w; x++)
          dst[y * m.dst_stride + x] = (Pel)clip3i(clpMin, clpMax, (p0[y * m.w + x] + p1[y * m.w + x] + offset) >> shiftNum);
    }
  }
  return 0;
}

ORC_API int orc_pelop_batch(int op, const Pel* s0Base, const Pel* s1Base, Pel* dstBase, const vvcgpu_pelop_desc* d, int n,
                            const vvcgpu_pelop_cfg* c)
{
  for (int i = 0; i < n; i++)
  {
    const Pel* s0 = s0Base + d[i].src0_off; const Pel* s1 = s1Base ? s1Base + d[i].src1_off : nullptr;
    Pel* dst = dstBase + d[i].dst_off;
    for (int y = 0; y < d[i].h; y++)
      for (int x = 0; x < d[i].w; x++)
      {
        const int a = s0[y * d[i].src0_stride + x];
        const int b = s1 ? s1[y * d[i].src1_stride + x] : 0;
        int v;
        switch (op)
        {
        case 0: v = clip3i(c->clp_min, c->clp_max, (a + b + c->offset) >> c->shift); break;
        case 1: v = clip3i(c->clp_min, c->clp_max, a + b); break;
        case 2: { const int t = (c->shift >= 0 ? (c->scale * a) >> c->shift : (c->scale * a) << -c->shift) + c->offset;
                  v = c->clip ? clip3i(c->clp_min, c->clp_max, t) : t; } break;
        case 3: v = a - b; break;
        case 4: v = c->clip ? clip3i(c->clp_min, c->clp_max, 2 * a - b) : 2 * a - b; break;
        default: v = clip3i(c->clp_min, c->clp_max, a); break;
        }
        dst[y * d[i].dst_stride + x] = (Pel)v;
      }
  }
  return 0;
}
