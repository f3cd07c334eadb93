// oracle/restate/stats.cpp -- TEST INFRASTRUCTURE: CPU restatement of the encoder-side statistics kernels.
// S2 follows EncSampleAdaptiveOffset::getBlkStats (EncoderLib/EncSampleAdaptiveOffset.cpp:1122-1490, the
//    isCalculatePreDeblockSamples == false branches) and getStatistics' flag derivation (:296-306).
// A3 follows EncAdaptiveLoopFilter::getBlkStats (EncoderLib/EncAdaptiveLoopFilter.cpp:1394-1440) and
//    calcCovariance (:1442-1515) with the filter patterns of AlfFilterShape (CommonLib/TypeDef.h:1436-1509).
#include "orc_common.h"
#include <vector>

ORC_API int orc_sao_stats(const Pel* org, int ostride, const Pel* rec, int rstride, int w, int h, int ctuW, int ctuH,
                          int bitDepth, const uint8_t* avail, int skipR, int skipB, int64_t* out)
{
  const int wCtu = (w + ctuW - 1) / ctuW, hCtu = (h + ctuH - 1) / ctuH;
  memset(out, 0, sizeof(int64_t) * 320 * wCtu * hCtu);
  for (int cy = 0; cy < hCtu; cy++)
    for (int cx = 0; cx < wCtu; cx++)
    {
      const int x0 = cx * ctuW, y0 = cy * ctuH;
      const int width = std::min(ctuW, w - x0), height = std::min(ctuH, h - y0);
      const int a = avail ? avail[cy * wCtu + cx] : ((cx > 0 ? 1 : 0) | (cy > 0 ? 4 : 0) | (cx > 0 && cy > 0 ? 16 : 0));
      const bool left = a & 1, above = (a >> 2) & 1, aboveLeft = (a >> 4) & 1;
      const bool right = x0 + ctuW < w, below = y0 + ctuH < h;           // :300-306
      int64_t* st = out + (int64_t)(cy * wCtu + cx) * 320;
      const Pel* s = rec + y0 * rstride + x0;
      const Pel* o = org + y0 * ostride + x0;
      auto S = [&](int x, int y) { return (int)s[y * rstride + x]; };
      auto acc = [&](int type, int cls, int x, int y) {
        st[type * 64 + cls] += o[y * ostride + x] - s[y * rstride + x];
        st[type * 64 + 32 + cls]++;
      };
      const int endXe = right ? width - skipR : width - 1;               // EO_0/135/45
      const int startXe = left ? 0 : 1;
      // EO_0 (:1146-1170)
      { const int endY = below ? height - skipB : height;
        for (int y = 0; y < endY; y++) for (int x = startXe; x < endXe; x++)
          acc(0, 2 + sgni(S(x, y) - S(x - 1, y)) + sgni(S(x, y) - S(x + 1, y)), x, y); }
      // EO_90 (:1198-1237)
      { const int endX = right ? width - skipR : width, startY = above ? 0 : 1, endY = below ? height - skipB : height - 1;
        for (int y = startY; y < endY; y++) for (int x = 0; x < endX; x++)
          acc(1, 2 + sgni(S(x, y) - S(x, y - 1)) + sgni(S(x, y) - S(x, y + 1)), x, y); }
      // EO_135 (:1263-1320): first line has its own start/end
      { const int endY = below ? height - skipB : height - 1;
        const int fls = aboveLeft ? 0 : 1, fle = above ? endXe : 1;
        for (int x = fls; x < fle; x++) acc(2, 2 + sgni(S(x, 0) - S(x - 1, -1)) + sgni(S(x, 0) - S(x + 1, 1)), x, 0);
        for (int y = 1; y < endY; y++) for (int x = startXe; x < endXe; x++)
          acc(2, 2 + sgni(S(x, y) - S(x - 1, y - 1)) + sgni(S(x, y) - S(x + 1, y + 1)), x, y); }
      // EO_45 (:1346-1400); (!isRightAvail && isAboveRightAvail) is never true (:306)
      { const int endY = below ? height - skipB : height - 1;
        if (above) for (int x = startXe; x < endXe; x++) acc(3, 2 + sgni(S(x, 0) - S(x + 1, -1)) + sgni(S(x, 0) - S(x - 1, 1)), x, 0);
        for (int y = 1; y < endY; y++) for (int x = startXe; x < endXe; x++)
          acc(3, 2 + sgni(S(x, y) - S(x + 1, y - 1)) + sgni(S(x, y) - S(x - 1, y + 1)), x, y); }
      // BO (:1428-1450)
      { const int endX = right ? width - skipR : width, endY = below ? height - skipB : height, sh = bitDepth - 5;
        for (int y = 0; y < endY; y++) for (int x = 0; x < endX; x++) acc(4, S(x, y) >> sh, x, y); }
    }
  return 0;
}

// ------------------------------------------------------------------------------------------------
static const int pattern5[13] = { 0, 1, 2, 3, 4, 5, 6, 5, 4, 3, 2, 1, 0 };
static const int pattern7[25] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0 };

namespace {
struct PlaneC {
  const Pel* p; int stride, w, h;
  inline int at(int x, int y) const {
    x = x < 0 ? 0 : (x >= w ? w - 1 : x);
    y = y < 0 ? 0 : (y >= h ? h - 1 : y);
    return p[y * stride + x];
  }
};
}

// calcCovariance (:1442-1515) with R(dx,dy) = rec sample at (x+dx, y+dy)
static void calcCov(int* E, const PlaneC& P, int x, int y, const int* pat, int half, int t)
{
  int k = 0;
  auto R = [&](int dx, int dy) { return P.at(x + dx, y + dy); };
  if (t == 0)
  {
    for (int i = -half; i < 0; i++) for (int j = -half - i; j <= half + i; j++) E[pat[k++]] += R(j, i) + R(-j, -i);
    for (int j = -half; j < 0; j++) E[pat[k++]] += R(j, 0) + R(-j, 0);
  }
  else if (t == 1)
  {
    for (int j = -half; j < 0; j++) for (int i = -half - j; i <= half + j; i++) E[pat[k++]] += R(j, i) + R(-j, -i);
    for (int i = -half; i < 0; i++) E[pat[k++]] += R(0, i) + R(0, -i);
  }
  else if (t == 2)
  {
    for (int i = -half; i < 0; i++) for (int j = half + i; j >= -half - i; j--) E[pat[k++]] += R(j, i) + R(-j, -i);
    for (int j = -half; j < 0; j++) E[pat[k++]] += R(j, 0) + R(-j, 0);
  }
  else
  {
    for (int j = -half; j < 0; j++) for (int i = half + j; i >= -half - j; i--) E[pat[k++]] += R(j, i) + R(-j, -i);
    for (int i = -half; i < 0; i++) E[pat[k++]] += R(0, i) + R(0, -i);
  }
  E[pat[k++]] += R(0, 0);
}

ORC_API int orc_alf_stats(const Pel* org, int ostride, const Pel* rec, int rstride, int w, int h, int ctu,
                          const uint16_t* cls, int filterType, int64_t* out)
{
  const int N = filterType ? 13 : 7, half = filterType ? 3 : 2, rec_sz = N * N + N + 1;
  const int* pat = filterType ? pattern7 : pattern5;
  const int nCls = cls ? 25 : 1;
  const int wCtu = (w + ctu - 1) / ctu, hCtu = (h + ctu - 1) / ctu;
  memset(out, 0, sizeof(int64_t) * (size_t)rec_sz * nCls * wCtu * hCtu);
  PlaneC P{ rec, rstride, w, h };
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++)
    {
      int classIdx = 0, t = 0;
      if (cls) { const uint16_t c = cls[(y >> 2) * (w >> 2) + (x >> 2)]; classIdx = c & 0xff; t = c >> 8; }
      int E[13] = { 0 };
      calcCov(E, P, x, y, pat, half, t);
      const int yl = org[y * ostride + x] - rec[y * rstride + x];
      int64_t* a = out + ((int64_t)((y / ctu) * wCtu + x / ctu) * nCls + classIdx) * rec_sz;
      for (int k = 0; k < N; k++)
      {
        for (int l = k; l < N; l++) a[k * N + l] += E[k] * E[l];
        a[N * N + k] += E[k] * yl;
      }
      a[N * N + N] += yl * yl;
    }
  for (int64_t i = 0; i < (int64_t)nCls * wCtu * hCtu; i++)      // mirror the upper triangle (:1430-1439)
  {
    int64_t* a = out + i * rec_sz;
    for (int k = 1; k < N; k++) for (int l = 0; l < k; l++) a[k * N + l] = a[l * N + k];
  }
  return 0;
}
