// oracle/restate/tzsearch.cpp -- TEST INFRASTRUCTURE: scalar restatement of the integer TZ search (next row N2).
//   InterSearch::xTZSearch            EncoderLib/InterSearch.cpp:1971-2252
//   InterSearch::xTZSearchHelp        :249-343   (subShiftMode 0/2 branch :320-342)
//   InterSearch::xTZ2PointSearch      :349-374
//   InterSearch::xTZ8PointDiamondSearch :431-632
//   InterSearch::xSetSearchRange      :1820-1883 (composite reference off)
//   clipMv                            CommonLib/Mv.cpp:64-80;  Mv::divideByPowerOf2 (ME_ENABLE_ROUNDING_OF_MVS 1) Mv.h:142-151
// Pinned against the compiled reference's own xTZSearch (oracle/_ref, vtmref_tz_search) by tests/golden/tzsearch.npz.
//
// The diamond pattern is written as "candidate c of a round" (c < 16, reference visiting order), the form the device kernel
// evaluates in parallel; visiting the valid candidates in index order reproduces the nested ifs of :431-632 (the in-range fast
// paths of :520-531 / :574-592 are the same list with every test true).
#include "orc_common.h"

extern "C" uint64_t orc_mvcost(const vvcgpu_mvcost* m, int x, int y);
extern "C" uint64_t orc_sad(const Pel* org, int os, const Pel* cur, int cs, int w, int h, int subShift);

// visit statistics of the last orc_tz_search call (probes, rounds, raster probes) -- for the measurement notes in DESIGN.md
static uint64_t g_stat[3];
ORC_API void orc_tz_stats(uint64_t* out) { out[0] = g_stat[0]; out[1] = g_stat[1]; out[2] = g_stat[2]; }

namespace {

struct Range { int left, right, top, bottom; };

struct Tz
{
  const Pel* org; int os; const Pel* ref; int rs;
  const vvcgpu_tz_pu* pu; const vvcgpu_tz_cfg* cfg; vvcgpu_mvcost mc;
  Range sr;
  uint64_t bestSad; int bestX, bestY; unsigned bestDist, bestRound; int pointNr;
  bool countRaster = false;

  void clip(int& hor, int& ver) const                       // Mv.cpp:64-80, quarter units
  {
    const int off = 8;
    const int horMax = (cfg->pic_w + off - pu->pos_x - 1) << 2, horMin = (-cfg->max_cu_w - off - pu->pos_x + 1) << 2;
    const int verMax = (cfg->pic_h + off - pu->pos_y - 1) << 2, verMin = (-cfg->max_cu_h - off - pu->pos_y + 1) << 2;
    hor = std::min(horMax, std::max(horMin, hor));
    ver = std::min(verMax, std::max(verMin, ver));
  }
  static int div4(int v) { return (v + 2) >> 2; }             // Mv.h:142-151 with rounding

  void probe(int x, int y, int pn, unsigned dist)             // :320-342
  {
    // the read position is clamped to the rectangle the caller declared readable (include/vvcgpu.h, N2)
    g_stat[0]++; if (dist >= 5 && pn == 0 && countRaster) g_stat[2]++;
    const int px = clip3i(cfg->ref_x0, cfg->ref_x1 - pu->w, pu->ref_x + x), py = clip3i(cfg->ref_y0, cfg->ref_y1 - pu->h, pu->ref_y + y);
    uint64_t sad = orc_sad(org + (ptrdiff_t)pu->org_y * os + pu->org_x, os, ref + (ptrdiff_t)py * rs + px, rs, pu->w, pu->h, pu->sub_shift);
    if (sad < bestSad)
    {
      sad += orc_mvcost(&mc, x, y);
      if (sad < bestSad) { bestSad = sad; bestX = x; bestY = y; bestDist = dist; bestRound = 0; pointNr = pn; }
    }
  }

  // candidate c of the diamond round (sx, sy, d); false = not visited
  bool candidate(int c, int sx, int sy, int d, bool corners, int& x, int& y, int& pn, unsigned& dd) const
  {
    const int top = sy - d, bottom = sy + d, left = sx - d, right = sx + d;
    if (d == 1)                                                // :446-492
    {
      if (c >= 8) return false;
      static const signed char ox[8] = { -1, 0, 1, -1, 1, -1, 0, 1 }, oy[8] = { -1, -1, -1, 0, 0, 1, 1, 1 };
      const bool corner = ox[c] != 0 && oy[c] != 0;
      if (corner && !corners) return false;
      x = sx + ox[c]; y = sy + oy[c]; pn = c + 1; dd = 1;
      if (oy[c] < 0 && !(top >= sr.top)) return false;
      if (oy[c] > 0 && !(bottom <= sr.bottom)) return false;
      if (ox[c] < 0 && !(left >= sr.left)) return false;
      if (ox[c] > 0 && !(right <= sr.right)) return false;
      return true;
    }
    if (d <= 8)                                                // :496-569
    {
      if (c >= 8) return false;
      const int h2 = d >> 1;
      static const signed char ox[8] = { 0, -1, 1, -2, 2, -1, 1, 0 }, oy[8] = { -2, -1, -1, 0, 0, 1, 1, 2 };  // 2 = d, 1 = d >> 1
      static const signed char pnr[8] = { 2, 1, 3, 4, 5, 6, 8, 7 };
      const bool half = (ox[c] & 1) != 0;
      x = sx + (ox[c] == 2 ? d : ox[c] == -2 ? -d : ox[c] * h2);
      y = sy + (oy[c] == 2 ? d : oy[c] == -2 ? -d : oy[c] * h2);
      pn = pnr[c]; dd = half ? (unsigned)h2 : (unsigned)d;
      if (oy[c] < 0 && !(y >= sr.top)) return false;
      if (oy[c] > 0 && !(y <= sr.bottom)) return false;
      if (ox[c] < 0 && !(x >= sr.left)) return false;
      if (ox[c] > 0 && !(x <= sr.right)) return false;
      return true;
    }
    pn = 0; dd = (unsigned)d;                                  // :571-630
    if (c < 4)
    {
      x = c == 1 ? left : (c == 2 ? right : sx);
      y = c == 0 ? top : (c == 3 ? bottom : sy);
      return c == 0 ? top >= sr.top : c == 1 ? left >= sr.left : c == 2 ? right <= sr.right : bottom <= sr.bottom;
    }
    const int index = ((c - 4) >> 2) + 1, q = (c - 4) & 3, off = (d >> 2) * index;
    x = (q & 1) ? sx + off : sx - off;
    y = (q & 2) ? bottom - off : top + off;
    if ((q & 2) ? !(y <= sr.bottom) : !(y >= sr.top)) return false;
    if ((q & 1) ? !(x <= sr.right) : !(x >= sr.left)) return false;
    return true;
  }

  void diamond(int sx, int sy, int d, bool corners)
  {
    bestRound += 1;                                            // :444
    g_stat[1]++;
    for (int c = 0; c < 16; c++)
    {
      int x, y, pn; unsigned dd;
      if (candidate(c, sx, sy, d, corners, x, y, pn, dd)) probe(x, y, pn, dd);
    }
  }

  void twoPoint()                                              // :349-374: the two untested neighbours of the best point
  {
    static const signed char off[9][2][2] = {
      { { 0, 0 }, { 0, 0 } },   { { -1, 0 }, { 0, -1 } }, { { -1, -1 }, { 1, -1 } }, { { 0, -1 }, { 1, 0 } }, { { -1, 1 }, { -1, -1 } },
      { { 1, -1 }, { 1, 1 } },  { { -1, 0 }, { 0, 1 } },  { { -1, 1 }, { 1, 1 } },   { { 1, 0 }, { 0, 1 } } };
    const int bx = bestX, by = bestY, p = pointNr;
    for (int k = 0; k < 2; k++)
    {
      const int x = bx + off[p][k][0], y = by + off[p][k][1];
      if (x >= sr.left && x <= sr.right && y >= sr.top && y <= sr.bottom) probe(x, y, 0, 2);
    }
  }

  void setRange(int bx, int by, int range)                    // :1820-1853
  {
    int hor = bx << 2, ver = by << 2;
    clip(hor, ver);
    int l = hor - (range << 2), t = ver - (range << 2), r = hor + (range << 2), b = ver + (range << 2);
    clip(l, t); clip(r, b);
    sr.left = div4(l); sr.top = div4(t); sr.right = div4(r); sr.bottom = div4(b);
  }

  void run(vvcgpu_search_best* out)
  {
    const bool ext = (pu->flags & VVCGPU_TZ_EXTENDED) != 0, fast = (pu->flags & VVCGPU_TZ_FAST) != 0;
    const int raster = fast ? 8 : 5, range = cfg->search_range;
    mc.lambda = cfg->lambda; mc.pred_hor = pu->pred_hor; mc.pred_ver = pu->pred_ver; mc.cost_scale = cfg->cost_scale; mc.imv_shift = cfg->imv_shift;

    int mx = pu->start_x, my = pu->start_y;
    clip(mx, my); mx = div4(mx); my = div4(my);               // :2005-2006
    bestSad = ~0ull; bestX = bestY = 0; bestDist = 0; bestRound = 0; pointNr = 0;
    probe(mx, my, 0, 0);                                       // :2023
    if (!fast && (mx != 0 || my != 0) && (bestX != 0 || bestY != 0)) probe(0, 0, 0, 0);   // :2026-2034
    if (pu->flags & VVCGPU_TZ_PRED2)                           // :2038-2051
    {
      int px = pu->pred2_x << 2, py = pu->pred2_y << 2;
      clip(px, py); px = div4(px); py = div4(py);
      if ((mx != px || my != py) && (px != bestX || py != bestY)) probe(px, py, 0, 0);
    }
    setRange(bestX, bestY, range >> (fast ? 1 : 0));           // :2052-2061

    int startX = bestX, startY = bestY;
    const bool bestCandidateZero = bestX == 0 && bestY == 0;
    for (int d = 1; d <= range; d *= 2)                        // :2072-2088
    {
      diamond(startX, startY, d, ext);
      if (cfg->first_search_stop && bestRound >= 3) break;
    }
    if (ext && !bestCandidateZero)                             // :2111-2126 (the :2090-2109 branch is dead: both flags are bExtendedSettings)
      for (int d = 1; d <= (range >> 1); d *= 2) diamond(0, 0, d, false);

    if (bestDist == 1) { bestDist = 0; twoPoint(); }           // :2129-2133

    if (ext)                                                   // :2136-2157 adaptive raster
    {
      int win = raster; Range l = sr;
      if (!((int)bestDist >= raster)) { win++; l.left /= 2; l.right /= 2; l.top /= 2; l.bottom /= 2; }
      bestDist = win; countRaster = true; g_stat[1]++;
      for (int y = l.top; y <= l.bottom; y += win)
        for (int x = l.left; x <= l.right; x += win) probe(x, y, 0, win);
    }
    else if ((int)bestDist >= raster)                          // :2158-2171
    {
      bestDist = raster; countRaster = true; g_stat[1]++;
      for (int y = sr.top; y <= sr.bottom; y += raster)
        for (int x = sr.left; x <= sr.right; x += raster) probe(x, y, 0, raster);
    }

    countRaster = false;
    while (bestDist > 0)                                       // :2207-2241 star refinement
    {
      startX = bestX; startY = bestY; bestDist = 0; pointNr = 0;
      for (int d = 1; d < range + 1; d *= 2)
      {
        diamond(startX, startY, d, ext);
        if (fast && bestRound >= 2) break;
      }
      if (bestDist == 1) { bestDist = 0; if (pointNr != 0) twoPoint(); }
    }
    out->x = bestX; out->y = bestY; out->cost = bestSad; out->sad = bestSad - orc_mvcost(&mc, bestX, bestY);   // :2247-2251
  }
};

}  // namespace

ORC_API int orc_tz_search(const Pel* org, int os, const Pel* ref, int rs, const vvcgpu_tz_pu* pus, int n, const vvcgpu_tz_cfg* cfg,
                          vvcgpu_search_best* out)
{
  g_stat[0] = g_stat[1] = g_stat[2] = 0;
  for (int i = 0; i < n; i++)
  {
    Tz t; t.org = org; t.os = os; t.ref = ref; t.rs = rs; t.pu = pus + i; t.cfg = cfg;
    t.run(out + i);
  }
  return 0;
}


// ---- InterSearch::xPatternSearchIntRefine :2408-2501 (AMVR) ------------------------------------------------------------------
extern "C" uint64_t orc_satd(const Pel* org, int os, const Pel* cur, int cs, int w, int h);
extern "C" uint32_t orc_expgolomb_bits(int v);

ORC_API int orc_imv_refine(const Pel* org, int os, const Pel* ref, int rs, const vvcgpu_imv_pu* pus, int n, const vvcgpu_tz_cfg* cfg,
                           int useHad, double weight, vvcgpu_imv_result* out)
{
  static const int testPos[9][2] = { { 0, 0 }, { -1, -1 }, { -1, 0 }, { -1, 1 }, { 0, -1 }, { 0, 1 }, { 1, -1 }, { 1, 0 }, { 1, 1 } };
  for (int i = 0; i < n; i++)
  {
    const vvcgpu_imv_pu& p = pus[i];
    const int sh = cfg->imv_shift, mvOffset = 1 << sh;
    auto clipq = [&](int& hor, int& ver)                          // clipMv, quarter units
    {
      const int horMax = (cfg->pic_w + 8 - p.pos_x - 1) << 2, horMin = (-cfg->max_cu_w - 8 - p.pos_x + 1) << 2;
      const int verMax = (cfg->pic_h + 8 - p.pos_y - 1) << 2, verMin = (-cfg->max_cu_h - 8 - p.pos_y + 1) << 2;
      hor = std::min(horMax, std::max(horMin, hor)); ver = std::min(verMax, std::max(verMin, ver));
    };
    auto bitsOf = [&](int x, int y, int c) { return orc_expgolomb_bits((x - p.cand_x[c]) >> sh) + orc_expgolomb_bits((y - p.cand_y[c]) >> sh); };   // cost scale 0
    const int mvx = p.mv_x << 2, mvy = p.mv_y << 2;               // :2418
    int baseX[2], baseY[2];
    for (int c = 0; c < 2; c++)                                   // cBaseMvd, rounded (roundMV, Mv.cpp:44-52)
    {
      const int off = 1 << (sh - 1);
      baseX[c] = (((mvx - p.cand_x[c]) + off) >> sh) << sh;
      baseY[c] = (((mvy - p.cand_y[c]) + off) >> sh) << sh;
    }
    uint64_t bestDist = ~0ull, satd = 0;
    int bestX = mvx, bestY = mvy, bestIdx = p.mvp_idx, bestBits = 0;
    for (int pos = 0; pos < 9; pos++)
    {
      int tx[2] = { 0, 0 }, ty[2] = { 0, 0 };
      for (int c = 0; c < p.num_cand; c++)
      {
        tx[c] = testPos[pos][0] * mvOffset + baseX[c] + p.cand_x[c];
        ty[c] = testPos[pos][1] * mvOffset + baseY[c] + p.cand_y[c];
        uint64_t dist;
        if (c == 0 || tx[0] != tx[1] || ty[0] != ty[1])
        {
          int cx = tx[c], cy = ty[c];
          clipq(cx, cy);
          const int px = clip3i(cfg->ref_x0, cfg->ref_x1 - p.w, p.ref_x + (cx >> 2)), py = clip3i(cfg->ref_y0, cfg->ref_y1 - p.h, p.ref_y + (cy >> 2));
          const Pel* o = org + (ptrdiff_t)p.org_y * os + p.org_x;
          const Pel* r = ref + (ptrdiff_t)py * rs + px;
          const uint64_t d = useHad ? orc_satd(o, os, r, rs, p.w, p.h) : orc_sad(o, os, r, rs, p.w, p.h, 0);
          dist = satd = (uint64_t)((double)d * weight);           // :2456
        }
        else dist = satd;
        const uint32_t mvBits = bitsOf(tx[c], ty[c], c);
        const int iMvBits = (int)(p.idx_cost[c] + mvBits);
        dist += (uint64_t)(cfg->lambda * mvBits);
        if (dist < bestDist) { bestDist = dist; bestX = tx[c]; bestY = ty[c]; bestIdx = c; bestBits = iMvBits; }
      }
    }
    uint32_t bits = p.bits - p.idx_cost[p.mvp_idx];               // :2427
    bits += (uint32_t)bestBits;
    out[i].cost = bestDist - (uint64_t)(cfg->lambda * (uint32_t)bestBits) + (uint64_t)(cfg->lambda * bits);   // :2493
    bits += bitsOf(bestX, bestY, bestIdx);                        // :2496
    out[i].mv_x = bestX; out[i].mv_y = bestY; out[i].mvp_idx = bestIdx; out[i].bits = bits;
  }
  return 0;
}
