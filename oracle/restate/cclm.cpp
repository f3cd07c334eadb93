// oracle/restate/cclm.cpp -- TEST INFRASTRUCTURE: scalar restatement of the cross-component linear model (CCLM, next row N4).
//   IntraPrediction::xGetLumaRecPixels   CommonLib/IntraPrediction.cpp:1283-1581 (JVET_K0190 branch: one neighbour line, [1 2 1; 1 2 1] / 8)
//   IntraPrediction::xGetLMParameters    :1597-1857
//   IntraPrediction::predIntraChromaLM   :390-403  (= a * recLuma' >> shift + b, clipped: AreaBuf::linearTransform, Buffer.cpp:83-94)
// Pinned by tests/golden/cclm.npz: inputs and outputs of the reference's own predIntraChromaLM calls, captured inside reference
// encoder runs by the drop-in shim (tests/golden/gen_cclm.py).
#include "orc_common.h"

namespace {
int floorLog2(unsigned x) { int b = -1; while (x) { b++; x >>= 1; } return b; }
}

// luma: co-located luma block top-left (reconstruction, rows -2.. and columns -3.. are read when the neighbour is available)
// nbAbove[w], nbLeft[h]: reconstructed chroma neighbours of this component
ORC_API int orc_cclm_pred(const Pel* luma, int lumaStride, const Pel* nbAbove, const Pel* nbLeft, Pel* dst, int dstStride, int w, int h,
                          int aboveAvail, int leftAvail, int bdLuma, int bdChroma, int clpMin, int clpMax)
{
  const int rs = lumaStride, rs2 = lumaStride * 2;
  auto six = [&](const Pel* p) { return (p[0] * 2 + p[-1] + p[1] + p[rs] * 2 + p[rs - 1] + p[rs + 1] + 4) >> 3; };
  auto two = [&](const Pel* p) { return (p[0] + p[rs] + 1) >> 1; };
  auto inner = [&](int i, int j) { const Pel* p = luma + (ptrdiff_t)j * rs2 + 2 * i; return (i == 0 && !leftAvail) ? two(p) : six(p); };
  auto above = [&](int i) { const Pel* p = luma - rs2 + 2 * i; return (i == 0 && !leftAvail) ? two(p) : six(p); };     // :1381-1395
  auto left = [&](int j) { return six(luma + (ptrdiff_t)j * rs2 - 2); };                                                // :1455-1469

  int a = 0, b = 1 << (bdChroma - 1), shift = 0;
  if (aboveAvail || leftAvail)                                                                                            // :1677-1856
  {
    int x = 0, y = 0, xx = 0, xy = 0, countShift = 0;
    const int minDim = (leftAvail && aboveAvail) ? std::min(w, h) : (leftAvail ? h : w);
    if (aboveAvail)
    {
      for (int j = 0; j < minDim; j++)
      {
        const int idx = (j * w) / minDim, s = above(idx), c = nbAbove[idx];
        x += s; y += c; xx += s * s; xy += s * c;
      }
      countShift = floorLog2(minDim);
    }
    if (leftAvail)
    {
      for (int i = 0; i < minDim; i++)
      {
        const int idx = (i * h) / minDim, s = left(idx), c = nbLeft[idx];
        x += s; y += c; xx += s * s; xy += s * c;
      }
      countShift += aboveAvail ? 1 : floorLog2(minDim);
    }
    const int tempShift = bdChroma + countShift - 15;
    if (tempShift > 0)
    {
      const int r = 1 << (tempShift - 1);
      x = (x + r) >> tempShift; y = (y + r) >> tempShift; xx = (xx + r) >> tempShift; xy = (xy + r) >> tempShift;
      countShift -= tempShift;
    }
    const int avgX = x >> countShift, avgY = y >> countShift;
    const int rErrX = x & ((1 << countShift) - 1), rErrY = y & ((1 << countShift) - 1);
    const int iB = 7;
    shift = 13 - iB;
    if (countShift == 0) { a = 0; b = 1 << (bdChroma - 1); shift = 0; }
    else
    {
      const int a1 = xy - (avgX * avgY << countShift) - avgX * rErrY - avgY * rErrX;
      const int a2 = xx - (avgX * avgX << countShift) - 2 * avgX * rErrX;
      int sA1 = a1 == 0 ? 0 : floorLog2(std::abs(a1)) - (bdChroma - 2);
      int sA2 = a2 == 0 ? 0 : floorLog2(std::abs(a2)) - 5;
      if (sA1 < 0) sA1 = 0;
      if (sA2 < 0) sA2 = 0;
      const int sA = sA2 + (bdChroma + 4) - shift - sA1;
      const int a2s = a2 >> sA2, a1s = a1 >> sA1;
      if (a2s >= 32) a = a1s * (int)(uint32_t)(((1 << (bdLuma + 4)) + a2s / 2) / a2s);          // m_auShiftLM[a2s - 32], :153-157
      else a = 0;
      if (sA < 0) a = a << -sA; else a = a >> sA;
      a = clip3i(-(1 << (15 - iB)), (1 << (15 - iB)) - 1, a);
      a = a << iB;
      int n = 0;
      if (a != 0) n = floorLog2(std::abs(a) + ((a < 0 ? -1 : 1) - 1) / 2) - 5;
      n = (int16_t)n;
      shift = (shift + iB) - n;
      a = a >> n;
      b = avgY - ((a * avgX) >> shift);
    }
  }
  for (int j = 0; j < h; j++)
    for (int i = 0; i < w; i++)
      dst[j * dstStride + i] = (Pel)clip3i(clpMin, clpMax, ((a * (Pel)inner(i, j)) >> shift) + b);
  return 0;
}
