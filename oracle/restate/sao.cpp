// oracle/restate/sao.cpp -- TEST INFRASTRUCTURE: CPU restatement of SAO apply.
// Follows: SampleAdaptiveOffset::offsetBlock (CommonLib/SampleAdaptiveOffset.cpp:292-508),
//          offsetCTU (:510-562), SAOProcess CTU loop (:564-612).
// The reference walks each CTU with running sign buffers; per sample that is equivalent to
//   out = clip(c + offset[2 + sgn(c-a) + sgn(c-b)])   when both neighbours a,b are "available",
// where availability of a neighbour outside the CTU is the CTU's flag for that direction
// (left/right/above/below/4 corners) -- derived line by line from the start/end indices of
// each case (:307-311, :331-337, :363-415, :420-470).
#include "orc_common.h"

ORC_API int orc_sao_apply(const Pel* src, int sstride, Pel* dst, int dstride, int w, int h,
                          int ctuW, int ctuH, int bitDepth, const vvcgpu_sao_ctu* params,
                          int clpMin, int clpMax)
{
  const int wCtu = (w + ctuW - 1) / ctuW;
  static const int dxa[4] = { -1, 0, -1, 1 }, dya[4] = { 0, -1, -1, -1 };   // neighbour a of EO_0/90/135/45
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++)
    {
      const int cx = x / ctuW, cy = y / ctuH;
      const vvcgpu_sao_ctu& p = params[cy * wCtu + cx];
      if (p.type < 0) continue;
      const int c = src[y * sstride + x];
      if (p.type == 4)
      {
        dst[y * dstride + x] = (Pel)clip3i(clpMin, clpMax, c + p.offset[c >> (bitDepth - 5)]);
        continue;
      }
      const int x0 = cx * ctuW, y0 = cy * ctuH;
      const int x1 = std::min(x0 + ctuW, w), y1 = std::min(y0 + ctuH, h);
      bool ok = true;
      int nb[2];
      for (int k = 0; k < 2; k++)
      {
        const int nx = x + (k ? -dxa[p.type] : dxa[p.type]);
        const int ny = y + (k ? -dya[p.type] : dya[p.type]);
        const int hx = nx < x0 ? -1 : (nx >= x1 ? 1 : 0);
        const int hy = ny < y0 ? -1 : (ny >= y1 ? 1 : 0);
        int bit = -1;
        if (hx == -1 && hy == 0) bit = 0; else if (hx == 1 && hy == 0) bit = 1;
        else if (hx == 0 && hy == -1) bit = 2; else if (hx == 0 && hy == 1) bit = 3;
        else if (hx == -1 && hy == -1) bit = 4; else if (hx == 1 && hy == -1) bit = 5;
        else if (hx == -1 && hy == 1) bit = 6; else if (hx == 1 && hy == 1) bit = 7;
        if (bit >= 0 && !((p.avail >> bit) & 1)) { ok = false; break; }
        nb[k] = src[ny * sstride + nx];
      }
      if (!ok) continue;
      const int edgeType = sgni(c - nb[0]) + sgni(c - nb[1]);
      dst[y * dstride + x] = (Pel)clip3i(clpMin, clpMax, c + p.offset[2 + edgeType]);
    }
  return 0;
}
