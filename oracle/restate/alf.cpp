// oracle/restate/alf.cpp -- TEST INFRASTRUCTURE: CPU restatement of ALF classification + filtering.
// Follows: AdaptiveLoopFilter::deriveClassificationBlk  (CommonLib/AdaptiveLoopFilter.cpp:292-463)
//          AdaptiveLoopFilter::filterBlk<5|7>           (CommonLib/AdaptiveLoopFilter.cpp:465-650)
//          AdaptiveLoopFilter::ALFProcess CTU loop      (CommonLib/AdaptiveLoopFilter.cpp:68-139)
// Pinned against the compiled reference by tests/test_oracle_vs_ref.py and tests/golden/alf_*.npz.
#include "orc_common.h"

namespace {
// sample with the border replication ALFProcess obtains from extendBorderPel(3) (:90-92)
struct Plane {
  const Pel* p; int stride, w, h;
  inline int at(int x, int y) const {
    x = x < 0 ? 0 : (x >= w ? w - 1 : x);
    y = y < 0 ? 0 : (y >= h ? h - 1 : y);
    return p[y * stride + x];
  }
};
}

// A1.  cls[(y/4)*(w/4)+(x/4)] = classIdx | transposeIdx<<8
ORC_API int orc_alf_classify(const Pel* src, int stride, int w, int h, int bitDepth, uint16_t* cls)
{
  static const int th[16] = { 0, 1, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3, 4 };   // :294
  static const int transposeTable[8] = { 0, 1, 0, 2, 2, 3, 1, 3 };                 // :447
  const int shift = bitDepth + 4;                                                  // :287
  Plane P{ src, stride, w, h };
  for (int by = 0; by < h; by += 4)
    for (int bx = 0; bx < w; bx += 4)
    {
      int sumV = 0, sumH = 0, sumD0 = 0, sumD1 = 0;
      // 8x8 window starting 2 above/left of the block, Laplacians on the 2x2-subsampled grid (:312-357):
      // cell (r,c) with r,c even covers rows r,r+1 / cols c,c+1 but only pixels (r,c),(r,c+1),(r+1,c),(r+1,c+1)
      // in the pattern y0,y1 (row r) and yup0,yup1 (row r+1) -- all four pixels of the cell.
      for (int r = by - 2; r < by + 6; r += 2)
        for (int c = bx - 2; c < bx + 6; c += 2)
          for (int dy = 0; dy < 2; dy++)
            for (int dx = 0; dx < 2; dx++)
            {
              const int y = r + dy, x = c + dx;
              const int c2 = P.at(x, y) << 1;
              sumV  += abs(c2 - P.at(x, y - 1) - P.at(x, y + 1));
              sumH  += abs(c2 - P.at(x + 1, y) - P.at(x - 1, y));
              sumD0 += abs(c2 - P.at(x - 1, y - 1) - P.at(x + 1, y + 1));
              sumD1 += abs(c2 - P.at(x - 1, y + 1) - P.at(x + 1, y - 1));
            }
      const int tempAct = sumV + sumH;
      const int activity = (Pel)clip3i(0, 15, (tempAct * 32) >> shift);          // :391
      int classIdx = th[activity];
      int hv1, hv0, d1, d0, hvd1, hvd0, dirTempHV, dirTempD, mainDirection, secondaryDirection;
      if (sumV > sumH) { hv1 = sumV; hv0 = sumH; dirTempHV = 1; } else { hv1 = sumH; hv0 = sumV; dirTempHV = 3; }
      if (sumD0 > sumD1) { d1 = sumD0; d0 = sumD1; dirTempD = 0; } else { d1 = sumD1; d0 = sumD0; dirTempD = 2; }
      // int products wrap modulo 2^32 in the reference build (imul / _mm_mullo_epi32, AdaptiveLoopFilterX86.h:241)
      if ((int32_t)((uint32_t)d1 * (uint32_t)hv0) > (int32_t)((uint32_t)hv1 * (uint32_t)d0)) { hvd1 = d1; hvd0 = d0; mainDirection = dirTempD; secondaryDirection = dirTempHV; }
      else                     { hvd1 = hv1; hvd0 = hv0; mainDirection = dirTempHV; secondaryDirection = dirTempD; }
      int directionStrength = 0;
      if (hvd1 > 2 * hvd0) directionStrength = 1;
      if (hvd1 * 2 > 9 * hvd0) directionStrength = 2;
      if (directionStrength) classIdx += (((mainDirection & 0x1) << 1) + directionStrength) * 5;
      const int transposeIdx = transposeTable[mainDirection * 2 + (secondaryDirection >> 1)];
      cls[(by >> 2) * (w >> 2) + (bx >> 2)] = (uint16_t)(classIdx | (transposeIdx << 8));
    }
  return 0;
}

static void permute7(const int16_t* c, int t, int* f)   // :565-580
{
  static const int8_t m[4][7] = { {0,1,2,3,4,5,6}, {4,1,5,3,0,2,6}, {0,3,2,1,4,5,6}, {4,3,5,1,0,2,6} };
  for (int i = 0; i < 7; i++) f[i] = c[m[t][i]];
}
static void permute13(const int16_t* c, int t, int* f)  // :545-560
{
  static const int8_t m[4][13] = { {0,1,2,3,4,5,6,7,8,9,10,11,12}, {9,4,10,8,1,5,11,7,3,0,2,6,12},
                                   {0,3,2,1,8,7,6,5,4,9,10,11,12}, {9,8,10,4,3,7,11,5,1,0,2,6,12} };
  for (int i = 0; i < 13; i++) f[i] = c[m[t][i]];
}

static inline int filt5(const Plane& P, int x, int y, const int* f)
{
  int s = 0;
  s += f[0] * (P.at(x, y + 2) + P.at(x, y - 2));
  s += f[1] * (P.at(x + 1, y + 1) + P.at(x - 1, y - 1));
  s += f[2] * (P.at(x, y + 1) + P.at(x, y - 1));
  s += f[3] * (P.at(x - 1, y + 1) + P.at(x + 1, y - 1));
  s += f[4] * (P.at(x + 2, y) + P.at(x - 2, y));
  s += f[5] * (P.at(x + 1, y) + P.at(x - 1, y));
  s += f[6] * P.at(x, y);
  return s;
}
static inline int filt7(const Plane& P, int x, int y, const int* f)
{
  int s = 0;
  s += f[0] * (P.at(x, y + 3) + P.at(x, y - 3));
  s += f[1] * (P.at(x + 1, y + 2) + P.at(x - 1, y - 2));
  s += f[2] * (P.at(x, y + 2) + P.at(x, y - 2));
  s += f[3] * (P.at(x - 1, y + 2) + P.at(x + 1, y - 2));
  s += f[4] * (P.at(x + 2, y + 1) + P.at(x - 2, y - 1));
  s += f[5] * (P.at(x + 1, y + 1) + P.at(x - 1, y - 1));
  s += f[6] * (P.at(x, y + 1) + P.at(x, y - 1));
  s += f[7] * (P.at(x - 1, y + 1) + P.at(x + 1, y - 1));
  s += f[8] * (P.at(x - 2, y + 1) + P.at(x + 2, y - 1));
  s += f[9] * (P.at(x + 3, y) + P.at(x - 3, y));
  s += f[10] * (P.at(x + 2, y) + P.at(x - 2, y));
  s += f[11] * (P.at(x + 1, y) + P.at(x - 1, y));
  s += f[12] * P.at(x, y);
  return s;
}

// A2 luma.  coeff: 25 x 13 int16 (m_coeffFinal).  ctuEnable may be NULL.
ORC_API int orc_alf_filter_luma(const Pel* src, int sstride, Pel* dst, int dstride, int w, int h, int ctu,
                                const uint16_t* cls, int filterType, const int16_t* coeff,
                                const uint8_t* ctuEnable, int clpMin, int clpMax)
{
  Plane P{ src, sstride, w, h };
  const int wCtu = (w + ctu - 1) / ctu;
  for (int by = 0; by < h; by += 4)
    for (int bx = 0; bx < w; bx += 4)
    {
      if (ctuEnable && !ctuEnable[(by / ctu) * wCtu + bx / ctu]) continue;
      const uint16_t c = cls[(by >> 2) * (w >> 2) + (bx >> 2)];
      const int16_t* cf = coeff + (c & 0xff) * 13;
      int f[13];
      if (filterType) permute13(cf, c >> 8, f); else permute7(cf, c >> 8, f);
      for (int y = by; y < by + 4; y++)
        for (int x = bx; x < bx + 4; x++)
        {
          int s = filterType ? filt7(P, x, y, f) : filt5(P, x, y, f);
          s = (s + 256) >> 9;                                                     // :625
          dst[y * dstride + x] = (Pel)clip3i(clpMin, clpMax, s);
        }
    }
  return 0;
}

// A2 chroma: single 5x5 filter, transposeIdx = 0 (:522-537 bChroma path).
ORC_API int orc_alf_filter_chroma(const Pel* src, int sstride, Pel* dst, int dstride, int w, int h, int ctu,
                                  const int16_t* coeff, const uint8_t* ctuEnable, int clpMin, int clpMax)
{
  Plane P{ src, sstride, w, h };
  const int wCtu = (w + ctu - 1) / ctu;
  int f[7];
  permute7(coeff, 0, f);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++)
    {
      if (ctuEnable && !ctuEnable[(y / ctu) * wCtu + x / ctu]) continue;
      int s = (filt5(P, x, y, f) + 256) >> 9;
      dst[y * dstride + x] = (Pel)clip3i(clpMin, clpMax, s);
    }
  return 0;
}
