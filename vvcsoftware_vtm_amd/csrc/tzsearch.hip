// tzsearch.hip -- integer-sample TZ search of whole PUs (next row N2) for gfx950.
//
// Reference behaviour reproduced (bit-exact position, cost and SAD):
//   InterSearch::xTZSearch               EncoderLib/InterSearch.cpp:1971-2252
//   InterSearch::xTZSearchHelp           :249-343 (subShiftMode 0/2 branch)
//   InterSearch::xTZ2PointSearch         :349-374
//   InterSearch::xTZ8PointDiamondSearch  :431-632
//   InterSearch::xSetSearchRange         :1820-1853,  clipMv CommonLib/Mv.cpp:64-80,  Mv::divideByPowerOf2 Mv.h:142-151
//   RdCost::getCostOfVectorWithPredictor CommonLib/RdCost.h:172-199
//
// Design: the search is a short, data-dependent chain of "rounds" (one probe, a diamond of <= 16 probes, two neighbours, a
// raster of up to ~1500 probes).  One wavefront owns one PU and keeps the whole search state wave-uniform; within a round
// the probes are independent, so the wave evaluates them together: LX lanes span a row of the block (coalesced reads),
// 64 / LX probes run side by side, and a round becomes a 64-bit min over  cost << 16 | visiting index  -- the reference's
// strict '<' in visiting order.  No data leaves the wave, no barrier is needed, and PUs of any size mix in one launch.
#include "common.h"

namespace {

__device__ __forceinline__ unsigned tz_expgolomb_bits(int v)   // RdCost.h:172-184
{
  unsigned len = 1, t = (v <= 0) ? ((unsigned)(-v) << 1) + 1 : (unsigned)(v << 1);
  while (t > 128u) { len += 14; t >>= 7; }
  return len + ((31 - __clz((int)t)) << 1);
}

// small signed tables packed into immediates: entry i holds v[i] + bias in `bits` bits
template <int N> constexpr unsigned pack_tab(const int (&v)[N], int bits, int bias)
{
  unsigned r = 0;
  for (int i = 0; i < N; i++) r |= (unsigned)(v[i] + bias) << (bits * i);
  return r;
}
// diamond at distance 1: the eight neighbours row by row, point numbers 1..8 (:446-492)
constexpr int kD1X[8] = { -1, 0, 1, -1, 1, -1, 0, 1 }, kD1Y[8] = { -1, -1, -1, 0, 0, 1, 1, 1 };
// diamond at 2 <= d <= 8, visiting order T, (L2,T2), (R2,T2), L, R, (L2,B2), (R2,B2), B (:496-569): offset signs and point numbers
constexpr int kD8X[8] = { 0, -1, 1, -1, 1, -1, 1, 0 }, kD8Y[8] = { -1, -1, -1, 0, 0, 1, 1, 1 }, kD8P[8] = { 2, 1, 3, 4, 5, 6, 8, 7 };
// the two untested neighbours of the best point by point number (:349-374)
constexpr int k2X0[9] = { 0, -1, -1, 0, -1, 1, -1, -1, 1 }, k2X1[9] = { 0, 0, 1, 1, -1, 1, 0, 1, 0 };
constexpr int k2Y0[9] = { 0, 0, -1, -1, 1, -1, 0, 1, 0 },   k2Y1[9] = { 0, -1, -1, 0, -1, 1, 1, 1, 1 };
constexpr unsigned D1X = pack_tab(kD1X, 2, 1), D1Y = pack_tab(kD1Y, 2, 1), D8X = pack_tab(kD8X, 2, 1), D8Y = pack_tab(kD8Y, 2, 1),
                   D8P = pack_tab(kD8P, 4, 0), P2X0 = pack_tab(k2X0, 2, 1), P2X1 = pack_tab(k2X1, 2, 1), P2Y0 = pack_tab(k2Y0, 2, 1),
                   P2Y1 = pack_tab(k2Y1, 2, 1);

struct TzRange { int left, right, top, bottom; };

struct TzWave
{
  // per PU, wave-uniform
  const Pel* org; const Pel* ref; int os, rs;
  int w, h, subShift, refX, refY;
  int rx0, ry0, rx1, ry1;               // clamp rectangle for the block origin
  int horMin, horMax, verMin, verMax;   // clipMv bounds, quarter units
  double lambda; int predHor, predVer, costScale, imvShift;
  TzRange sr;
  unsigned long long bestSad; int bestX, bestY, pointNr; unsigned bestDist, bestRound;
  // lane mapping
  int lx, lc, LX, CP;

  __device__ __forceinline__ void clip(int& hor, int& ver) const
  {
    hor = min(horMax, max(horMin, hor));
    ver = min(verMax, max(verMin, ver));
  }
  __device__ __forceinline__ unsigned long long mvcost(int x, int y) const
  {
    const unsigned bits = tz_expgolomb_bits(((x << costScale) - predHor) >> imvShift) + tz_expgolomb_bits(((y << costScale) - predVer) >> imvShift);
    return (unsigned long long)(lambda * (double)bits);
  }

  // this lane's share of the SAD of the probe at (x, y): columns lx, lx + LX, ... of every (1 << subShift)-th row
  __device__ __forceinline__ unsigned partial_sad(int x, int y) const
  {
    const int px = min(max(refX + x, rx0), rx1), py = min(max(refY + y, ry0), ry1);
    const Pel* o = org + lx;
    const Pel* r = ref + (ptrdiff_t)py * rs + px + lx;
    const int ostep = os << subShift, rstep = rs << subShift, rows = h >> subShift;
    unsigned acc = 0;
    if (w <= LX)
    {
      if (lx < w)
        for (int j = 0; j < rows; j++, o += ostep, r += rstep) acc += (unsigned)abs((int)o[0] - (int)r[0]);
    }
    else
    {
      for (int j = 0; j < rows; j++, o += ostep, r += rstep)
        for (int k = 0; k + lx < w; k += LX) acc += (unsigned)abs((int)o[k] - (int)r[k]);
    }
    return acc;
  }

  // one round: candidates 0 .. n-1 in visiting order; cand(c, x, y, pn, dd) -> visited?
  template <class F>
  __device__ __forceinline__ void round(int n, F cand)
  {
    unsigned long long key = ~0ull;
    for (int c0 = 0; c0 < n; c0 += CP)
    {
      const int c = c0 + lc;
      int x = 0, y = 0, pn = 0; unsigned dd = 0;
      const bool valid = c < n && cand(c, x, y, pn, dd);
      if (__ballot(valid) == 0ull) continue;
      unsigned s = valid ? partial_sad(x, y) : 0u;
      for (int m = 1; m < LX; m <<= 1) s += (unsigned)__shfl_xor((int)s, m);
      if (valid)
      {
        const unsigned long long cost = ((unsigned long long)s << subShift) + mvcost(x, y);
        key = min(key, (cost << 16) | (unsigned)c);
      }
    }
    for (int m = LX; m < 64; m <<= 1)
    {
      const unsigned long long o = __shfl_xor(key, m);
      key = min(key, o);
    }
    key = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(key >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)key);
    if (key != ~0ull && (key >> 16) < bestSad)
    {
      int x, y, pn; unsigned dd;
      cand((int)(key & 0xFFFFu), x, y, pn, dd);
      bestSad = key >> 16; bestX = x; bestY = y; bestDist = dd; bestRound = 0; pointNr = pn;
    }
  }

  __device__ __forceinline__ void probe(int x, int y)
  {
    round(1, [=](int, int& cx, int& cy, int& pn, unsigned& dd) { cx = x; cy = y; pn = 0; dd = 0; return true; });
  }

  // candidate c of the diamond round (sx, sy, d): the nested range tests of :431-632 in visiting order
  __device__ __forceinline__ bool diamond_cand(int c, int sx, int sy, int d, bool corners, int& x, int& y, int& pn, unsigned& dd) const
  {
    bool ok;
    int ox, oy;          // direction of the candidate relative to the start: decides which range tests apply
    if (d <= 8)
    {
      ok = c < 8;
      const int cc = c & 7;
      if (d == 1)
      {
        ox = (int)((D1X >> (2 * cc)) & 3u) - 1;
        oy = (int)((D1Y >> (2 * cc)) & 3u) - 1;
        if (ox != 0 && oy != 0 && !corners) ok = false;
        x = sx + ox; y = sy + oy; pn = cc + 1; dd = 1;
      }
      else
      {
        ox = (int)((D8X >> (2 * cc)) & 3u) - 1;
        oy = (int)((D8Y >> (2 * cc)) & 3u) - 1;
        const int mag = (ox != 0 && oy != 0) ? (d >> 1) : d;       // tips at d, diagonals at d >> 1
        x = sx + ox * mag; y = sy + oy * mag;
        pn = (int)((D8P >> (4 * cc)) & 15u);
        dd = (unsigned)mag;
      }
    }
    else
    {
      ok = c < 16;
      pn = 0; dd = (unsigned)d;
      if (c < 4)
      {
        ox = c == 1 ? -1 : (c == 2 ? 1 : 0);
        oy = c == 0 ? -1 : (c == 3 ? 1 : 0);
        x = sx + ox * d; y = sy + oy * d;
      }
      else
      {
        const int index = ((c - 4) >> 2) + 1, q = (c - 4) & 3, off = (d >> 2) * index;
        ox = (q & 1) ? 1 : -1; oy = (q & 2) ? 1 : -1;
        x = sx + ox * off; y = sy + oy * (d - off);
      }
    }
    if (oy < 0 && y < sr.top) ok = false;
    if (oy > 0 && y > sr.bottom) ok = false;
    if (ox < 0 && x < sr.left) ok = false;
    if (ox > 0 && x > sr.right) ok = false;
    return ok;
  }

  __device__ __forceinline__ void diamond(int sx, int sy, int d, bool corners)
  {
    bestRound += 1;
    round(d <= 8 ? 8 : 16, [=](int c, int& x, int& y, int& pn, unsigned& dd) { return diamond_cand(c, sx, sy, d, corners, x, y, pn, dd); });
  }

  __device__ __forceinline__ void two_point()
  {
    const int p = pointNr, bx = bestX, by = bestY;
    const TzRange r = sr;
    round(2, [=](int c, int& x, int& y, int& pn, unsigned& dd) {
      x = bx + (int)(((c == 0 ? P2X0 : P2X1) >> (2 * p)) & 3u) - 1;
      y = by + (int)(((c == 0 ? P2Y0 : P2Y1) >> (2 * p)) & 3u) - 1;
      pn = 0; dd = 2;
      return x >= r.left && x <= r.right && y >= r.top && y <= r.bottom;
    });
  }

  __device__ __forceinline__ void raster(TzRange l, int win)
  {
    if (l.right < l.left || l.bottom < l.top) return;
    const int nx = (l.right - l.left) / win + 1, ny = (l.bottom - l.top) / win + 1;
    round(nx * ny, [=](int c, int& x, int& y, int& pn, unsigned& dd) {
      const int j = c / nx, i = c - j * nx;
      x = l.left + i * win; y = l.top + j * win; pn = 0; dd = (unsigned)win;
      return true;
    });
  }

  __device__ __forceinline__ void set_range(int bx, int by, int range)
  {
    int hor = bx << 2, ver = by << 2;
    clip(hor, ver);
    int l = hor - (range << 2), t = ver - (range << 2), r = hor + (range << 2), b = ver + (range << 2);
    clip(l, t); clip(r, b);
    sr.left = (l + 2) >> 2; sr.top = (t + 2) >> 2; sr.right = (r + 2) >> 2; sr.bottom = (b + 2) >> 2;
  }
};

__global__ __launch_bounds__(256) void tz_search_kernel(const Pel* __restrict__ org, int os, const Pel* __restrict__ ref, int rs,
                                                        const vvcgpu_tz_pu* __restrict__ pus, int n, vvcgpu_tz_cfg cfg,
                                                        vvcgpu_search_best* __restrict__ results)
{
  const int lane = threadIdx.x & 63;
  const int b = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (b >= n) return;
  const vvcgpu_tz_pu pu = pus[b];

  TzWave s;
  s.org = org + (ptrdiff_t)pu.org_y * os + pu.org_x; s.ref = ref; s.os = os; s.rs = rs;
  s.w = pu.w; s.h = pu.h; s.subShift = pu.sub_shift; s.refX = pu.ref_x; s.refY = pu.ref_y;
  s.rx0 = cfg.ref_x0; s.ry0 = cfg.ref_y0; s.rx1 = cfg.ref_x1 - pu.w; s.ry1 = cfg.ref_y1 - pu.h;
  s.horMax = (cfg.pic_w + 8 - pu.pos_x - 1) << 2; s.horMin = (-cfg.max_cu_w - 8 - pu.pos_x + 1) << 2;
  s.verMax = (cfg.pic_h + 8 - pu.pos_y - 1) << 2; s.verMin = (-cfg.max_cu_h - 8 - pu.pos_y + 1) << 2;
  s.lambda = cfg.lambda; s.predHor = pu.pred_hor; s.predVer = pu.pred_ver; s.costScale = cfg.cost_scale; s.imvShift = cfg.imv_shift;
  int LX = 64; while (LX > 4 && LX > pu.w) LX >>= 1;         // lanes along a row: largest power of two <= min(w, 64), >= 4
  if (LX > pu.w) LX = 4;
  s.LX = LX; s.CP = 64 / LX; s.lx = lane & (LX - 1); s.lc = lane / LX;

  const bool ext = (pu.flags & VVCGPU_TZ_EXTENDED) != 0, fast = (pu.flags & VVCGPU_TZ_FAST) != 0;
  const int rasterStep = fast ? 8 : 5, range = cfg.search_range;

  int mx = pu.start_x, my = pu.start_y;
  s.clip(mx, my); mx = (mx + 2) >> 2; my = (my + 2) >> 2;
  s.bestSad = ~0ull >> 16; s.bestX = s.bestY = 0; s.bestDist = 0; s.bestRound = 0; s.pointNr = 0;
  s.sr = TzRange{ 0, 0, 0, 0 };
  s.probe(mx, my);
  if (!fast && (mx != 0 || my != 0) && (s.bestX != 0 || s.bestY != 0)) s.probe(0, 0);
  if (pu.flags & VVCGPU_TZ_PRED2)
  {
    int px = pu.pred2_x << 2, py = pu.pred2_y << 2;
    s.clip(px, py); px = (px + 2) >> 2; py = (py + 2) >> 2;
    if ((mx != px || my != py) && (px != s.bestX || py != s.bestY)) s.probe(px, py);
  }
  s.set_range(s.bestX, s.bestY, range >> (fast ? 1 : 0));

  int startX = s.bestX, startY = s.bestY;
  const bool bestCandidateZero = s.bestX == 0 && s.bestY == 0;
  for (int d = 1; d <= range; d *= 2)
  {
    s.diamond(startX, startY, d, ext);
    if (cfg.first_search_stop && s.bestRound >= 3) break;
  }
  if (ext && !bestCandidateZero)
    for (int d = 1; d <= (range >> 1); d *= 2) s.diamond(0, 0, d, false);

  if (s.bestDist == 1) { s.bestDist = 0; s.two_point(); }

  if (ext)
  {
    int win = rasterStep; TzRange l = s.sr;
    if (!((int)s.bestDist >= rasterStep)) { win++; l.left /= 2; l.right /= 2; l.top /= 2; l.bottom /= 2; }
    s.bestDist = (unsigned)win;
    s.raster(l, win);
  }
  else if ((int)s.bestDist >= rasterStep)
  {
    s.bestDist = (unsigned)rasterStep;
    s.raster(s.sr, rasterStep);
  }

  while (s.bestDist > 0)
  {
    startX = s.bestX; startY = s.bestY; s.bestDist = 0; s.pointNr = 0;
    for (int d = 1; d < range + 1; d *= 2)
    {
      s.diamond(startX, startY, d, ext);
      if (fast && s.bestRound >= 2) break;
    }
    if (s.bestDist == 1) { s.bestDist = 0; if (s.pointNr != 0) s.two_point(); }
  }

  if (lane == 0)
  {
    vvcgpu_search_best r;
    r.x = s.bestX; r.y = s.bestY; r.cost = s.bestSad; r.sad = s.bestSad - s.mvcost(s.bestX, s.bestY);
    results[b] = r;
  }
}

}  // namespace

extern "C" {

int vvcgpu_tz_search_batch(const vvc_pel* org, int org_stride, const vvc_pel* ref, int ref_stride,
                           const vvcgpu_tz_pu* pus, int n, const vvcgpu_tz_cfg* cfg_host,
                           vvcgpu_search_best* results, void* stream)
{
  VVC_CHECK_ARG(n >= 0, "tz_search_batch: n %d", n);
  if (n == 0) return VVCGPU_OK;
  VVC_CHECK_ARG(org && ref && pus && cfg_host && results, "tz_search_batch: null pointer");
  const vvcgpu_tz_cfg c = *cfg_host;
  VVC_CHECK_ARG(org_stride > 0 && ref_stride > 0, "tz_search_batch: strides %d %d", org_stride, ref_stride);
  VVC_CHECK_ARG(c.search_range >= 1 && c.search_range <= 512, "tz_search_batch: search_range %d", c.search_range);
  VVC_CHECK_ARG(c.cost_scale >= 0 && c.cost_scale <= 4 && c.imv_shift >= 0 && c.imv_shift <= 4, "tz_search_batch: cost_scale %d imv_shift %d",
                c.cost_scale, c.imv_shift);
  VVC_CHECK_ARG(c.lambda >= 0.0 && c.lambda < 1048576.0, "tz_search_batch: lambda out of range");
  VVC_CHECK_ARG(c.pic_w > 0 && c.pic_h > 0 && c.max_cu_w > 0 && c.max_cu_h > 0, "tz_search_batch: picture geometry");
  VVC_CHECK_ARG(c.ref_x1 - c.ref_x0 >= 128 && c.ref_y1 - c.ref_y0 >= 128 && c.ref_x0 >= 0 && c.ref_y0 >= 0 && c.ref_x1 <= ref_stride,
                "tz_search_batch: readable rectangle [%d,%d)x[%d,%d) (stride %d) must hold a 128x128 block", c.ref_x0, c.ref_x1, c.ref_y0, c.ref_y1,
                ref_stride);
  hipLaunchKernelGGL(tz_search_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, org, org_stride, ref, ref_stride, pus, n, c, results);
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}

}  // extern "C"
