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
// raster of up to ~1500 probes).  A team of lanes -- one wavefront (TEAM 1, small PUs) or one workgroup of four (TEAM 4, large
// PUs) -- owns one PU and keeps the whole search state uniform; within a round the probes are independent, so the team
// evaluates them together: the (sub-sampled) original block sits in LDS as packed pairs, LX = w / 4 lanes span a row with one
// 4-sample quad each (8-byte LDS read, aligned dword reads of the reference row + v_alignbit for odd positions, two
// v_sad_u16), RP lane groups split the rows when a round has few probes, and the remaining lanes take further probes.  A
// round then is a 64-bit min over  cost << 16 | visiting index  -- the reference's strict '<' in visiting order.
// PUs of any size mix in one launch; blocks that do not fit the team's LDS slice fall back to sample-wise reads.
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

// one round of probes: 0 single (x, y) | 1 diamond around (x, y) at distance d | 2 the two neighbours of the best point (x, y)
// with point number d | 3 raster over `win` with step d, nx columns
struct TzRound { int kind, n, x, y, d, corners, nx; unsigned rnx; TzRange win; };   // rnx = ceil(2^32 / nx)

constexpr int TZ_SEG_REGS = 6;            // raster rounds: dwords per lane of one staged chunk (6 x 64 dwords of LDS per wavefront)
constexpr int TZ_SEG_DWORDS = 64 * TZ_SEG_REGS / 2;   // longest reference segment of one block row (two rows per chunk at least)
constexpr int TZ_LDS_DWORDS = 8192;     // 32 KB: 8 KB per wavefront (TEAM 1: 64x64 / 64x128 sub-sampled) or all of it (TEAM 4: 128x128)

template <int TEAM>
struct TzTeam
{
  // per PU, team-uniform
  const Pel* org; const Pel* ref; int os, rs;
  int w, h, subShift, refX, refY;
  int rx0, ry0, rx1, ry1;               // clamp rectangle for the block origin
  int horMin, horMax, verMin, verMax;   // clipMv bounds, quarter units
  double lambda; int predHor, predVer, costScale, imvShift;
  TzRange sr;
  unsigned long long bestSad; int bestX, bestY, pointNr; unsigned bestDist, bestRound;
  // team mapping
  int tl;                               // lane within the team (TL = 64 * TEAM lanes)
  int LX;                               // lanes along a row (power of two >= w / 4)
  const unsigned* orgL;                 // LDS copy of the sub-sampled block (packed pairs, row pitch w / 2 dwords), or nullptr
  unsigned bias;                        // 0x80008000 when the block holds negative samples (both sides are biased then)
  unsigned long long* keyL;             // TEAM 4: one slot per wavefront
  unsigned* segL;                       // this wavefront's raster chunk (64 * TZ_SEG_REGS dwords)
  static constexpr int TL = 64 * TEAM;

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

  // candidate c of the round: position, point number, distance; false = not visited (the nested range tests of :431-632, :349-374)
  __device__ __forceinline__ bool candidate(const TzRound& R, int c, int& x, int& y, int& pn, unsigned& dd) const
  {
    if (R.kind == 0) { x = R.x; y = R.y; pn = 0; dd = 0; return true; }
    if (R.kind == 3)
    {
      const int j = R.nx > 1 ? (int)__umulhi((unsigned)c, R.rnx) : c, i = c - j * R.nx;     // exact for c, nx < 2^16
      x = R.win.left + i * R.d; y = R.win.top + j * R.d; pn = 0; dd = (unsigned)R.d;
      return true;
    }
    if (R.kind == 2)
    {
      const int p = R.d;
      x = R.x + (int)(((c == 0 ? P2X0 : P2X1) >> (2 * p)) & 3u) - 1;
      y = R.y + (int)(((c == 0 ? P2Y0 : P2Y1) >> (2 * p)) & 3u) - 1;
      pn = 0; dd = 2;
      return x >= sr.left && x <= sr.right && y >= sr.top && y <= sr.bottom;
    }
    const int sx = R.x, sy = R.y, d = R.d;
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
        if (ox != 0 && oy != 0 && !R.corners) ok = false;
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

  // this lane's share of the SADs of NP probes: quad lx of rows rp, rp + RP, ... of the sub-sampled block.  The NP probes share
  // the LDS read of the original quad and keep NP x (unroll) row reads in flight.
  template <int NP>
  __device__ __forceinline__ void partial_sad(const int (&x)[NP], const int (&y)[NP], int lx, int rp, int RP, unsigned (&acc)[NP]) const
  {
    const int rows = h >> subShift, rstep = rs << subShift;
#pragma unroll
    for (int u = 0; u < NP; u++) acc[u] = 0;
    if (orgL)
    {
      if (4 * lx >= w) return;
      const int halfW = w >> 1;
      const Pel* r[NP];
#pragma unroll
      for (int u = 0; u < NP; u++)
      {
        const int px = min(max(refX + x[u], rx0), rx1), py = min(max(refY + y[u], ry0), ry1);
        r[u] = ref + (ptrdiff_t)py * rs + px + 4 * lx + (ptrdiff_t)rp * rstep;
      }
      const unsigned* o = orgL + rp * halfW + 2 * lx;
      const ptrdiff_t rinc = (ptrdiff_t)RP * rstep; const int oinc = RP * halfW;
      if ((rs & 1) == 0)
      {
        // even row pitch: a probe keeps its dword phase on every row -> aligned pointer and shift are set up once
        const unsigned* g[NP]; unsigned sh[NP];
#pragma unroll
        for (int u = 0; u < NP; u++)
        {
          const uintptr_t a = reinterpret_cast<uintptr_t>(r[u]);
          g[u] = reinterpret_cast<const unsigned*>(a & ~(uintptr_t)3);
          sh[u] = (unsigned)(a & 2) << 3;
        }
        const ptrdiff_t ginc = rinc >> 1;
        if (bias == 0u)
        {
#pragma unroll 2
          for (int j = rp; j < rows; j += RP, o += oinc)
          {
            const uint2 ov = *reinterpret_cast<const uint2*>(o);
#pragma unroll
            for (int u = 0; u < NP; u++)
            {
              const unsigned g0 = g[u][0], g1 = g[u][1], g2 = sh[u] ? g[u][2] : 0u;
              acc[u] = __builtin_amdgcn_sad_u16(ov.x, __builtin_amdgcn_alignbit(g1, g0, sh[u]), acc[u]);
              acc[u] = __builtin_amdgcn_sad_u16(ov.y, __builtin_amdgcn_alignbit(g2, g1, sh[u]), acc[u]);
              g[u] += ginc;
            }
          }
        }
        else
        {
          for (int j = rp; j < rows; j += RP, o += oinc)
          {
            const uint2 ov = *reinterpret_cast<const uint2*>(o);
#pragma unroll
            for (int u = 0; u < NP; u++)
            {
              const unsigned g0 = g[u][0], g1 = g[u][1], g2 = sh[u] ? g[u][2] : 0u;
              acc[u] = __builtin_amdgcn_sad_u16(ov.x, __builtin_amdgcn_alignbit(g1, g0, sh[u]) ^ 0x80008000u, acc[u]);
              acc[u] = __builtin_amdgcn_sad_u16(ov.y, __builtin_amdgcn_alignbit(g2, g1, sh[u]) ^ 0x80008000u, acc[u]);
              g[u] += ginc;
            }
          }
        }
        return;
      }
      for (int j = rp; j < rows; j += RP, o += oinc)           // odd row pitch: the phase alternates
      {
        const uint2 ov = *reinterpret_cast<const uint2*>(o);
#pragma unroll
        for (int u = 0; u < NP; u++)
        {
          const uintptr_t a = reinterpret_cast<uintptr_t>(r[u]);
          const unsigned* g = reinterpret_cast<const unsigned*>(a & ~(uintptr_t)3);
          const unsigned sh = (unsigned)(a & 2) << 3;
          const unsigned g0 = g[0], g1 = g[1], g2 = sh ? g[2] : 0u;
          acc[u] = __builtin_amdgcn_sad_u16(ov.x, __builtin_amdgcn_alignbit(g1, g0, sh) ^ bias, acc[u]);
          acc[u] = __builtin_amdgcn_sad_u16(ov.y, __builtin_amdgcn_alignbit(g2, g1, sh) ^ bias, acc[u]);
          r[u] += rinc;
        }
      }
      return;
    }
    // block too large for the LDS slice: sample-wise
    const int ostep = os << subShift;
#pragma unroll
    for (int u = 0; u < NP; u++)
    {
      const int px = min(max(refX + x[u], rx0), rx1), py = min(max(refY + y[u], ry0), ry1);
      for (int j = rp; j < rows; j += RP)
      {
        const Pel* o = org + (ptrdiff_t)j * ostep + 4 * lx;
        const Pel* rr = ref + (ptrdiff_t)py * rs + px + (ptrdiff_t)j * rstep + 4 * lx;
        for (int k = 0; k + 4 * lx < w; k += 4 * LX)
#pragma unroll
          for (int v = 0; v < 4; v++) acc[u] += (unsigned)abs((int)o[k + v] - (int)rr[k + v]);
      }
    }
  }

  // candidates c0 + lc + u * CPT (u < NP) of the round -> running key
  template <int NP>
  __device__ __forceinline__ void pass(const TzRound& R, int c0, int CPT, int G, int lc, int lx, int rp, int RP, unsigned long long& key) const
  {
    int x[NP], y[NP], c[NP]; bool valid[NP]; unsigned s[NP];
    bool any = false;
#pragma unroll
    for (int u = 0; u < NP; u++)
    {
      int pn; unsigned dd;
      c[u] = c0 + u * CPT + lc; x[u] = 0; y[u] = 0;
      valid[u] = c[u] < R.n && candidate(R, c[u], x[u], y[u], pn, dd);
      any |= valid[u];
    }
    if (__ballot(any) == 0ull) return;
    partial_sad<NP>(x, y, lx, rp, RP, s);                   // probes that are not visited read a clamped position and are dropped
#pragma unroll
    for (int u = 0; u < NP; u++)
    {
      for (int m = 1; m < G; m <<= 1) s[u] += (unsigned)__shfl_xor((int)s[u], m);
      if (valid[u])
      {
        const unsigned long long cost = ((unsigned long long)s[u] << subShift) + mvcost(x[u], y[u]);
        key = min(key, (cost << 16) | (unsigned)c[u]);
      }
    }
  }


  // Raster round through wave-private LDS.  The probes of a raster row (fixed y) are step samples apart, so their block rows overlap in
  // memory: a pass takes K raster rows x T probes (K * T <= 64 lanes; long raster rows are cut into tiles of T, short ones are taken
  // K at a time), stages for several block rows the K contiguous reference segments ((T - 1) * step + w samples each) in LDS with
  // coalesced dword loads (the next chunk is in flight in registers while the current one is consumed), and lane (jj, i) evaluates
  // its probe from LDS: broadcast 16-byte reads of the original row, dword reads of the segment at its own offset + v_alignbit for
  // odd offsets.  A wavefront executes in order, so its private chunk needs no second buffer.  TEAM 4 deals the passes to its four
  // wavefronts.  Returns false (generic path) when the block is not LDS-resident, narrower than 32 (measured: the generic path with
  // its full lane use is faster there) or not a multiple of 8 wide, a segment does not fit, or a probe would be clamped.
  __device__ __forceinline__ bool raster_rows(const TzRound& R, unsigned long long& key)
  {
    const int step = R.d, nx = R.nx, ny = R.n / R.nx;
    if (!orgL || w < 32 || (w & 7) || (rs & 1) || step < 1) return false;
    if (refX + R.win.left < rx0 || refX + R.win.left + (nx - 1) * step > rx1 || refY + R.win.top < ry0 || refY + R.win.top + (ny - 1) * step > ry1) return false;
    const int ntiles = (nx + 63) >> 6, T = (nx + ntiles - 1) / ntiles;     // probes per tile, balanced
    const int nd = ((T - 1) * step + w + 2) >> 1;                          // dwords of one segment, whatever its sub-dword phase
    if (nd > TZ_SEG_DWORDS) return false;
    const int K = max(1, min(min(64 / T, ny), (64 * TZ_SEG_REGS) / (2 * nd)));     // raster rows per pass (at least two block rows per chunk)
    const int lane = tl & 63, wave = TEAM == 4 ? tl >> 6 : 0;
    const int rows = h >> subShift, halfW = w >> 1;
    const ptrdiff_t gstep = ((ptrdiff_t)rs << subShift) >> 1;             // dwords between block rows
    const Pel* first = ref + (ptrdiff_t)(refY + R.win.top) * rs + refX + R.win.left;
    const int RB = min(rows, (64 * TZ_SEG_REGS) / (K * nd));              // block rows per chunk
    const unsigned rnd = (unsigned)(0x100000000ull / (unsigned)nd) + 1u;  // i / nd == umulhi(i, rnd) for the i used here
    const unsigned rK = (unsigned)(0x100000000ull / (unsigned)K) + 1u, rT = (unsigned)(0x100000000ull / (unsigned)T) + 1u;
    // staging: dword lane + 64 k of the chunk = (block row, segment, column)
    int ldRow[TZ_SEG_REGS], ldSeg[TZ_SEG_REGS], ldCol[TZ_SEG_REGS];
#pragma unroll
    for (int k = 0; k < TZ_SEG_REGS; k++)
    {
      const int i = lane + 64 * k, rsg = (int)__umulhi((unsigned)i, rnd);
      ldCol[k] = i - rsg * nd;
      ldRow[k] = K > 1 ? (int)__umulhi((unsigned)rsg, rK) : rsg; ldSeg[k] = rsg - ldRow[k] * K;
      if (ldRow[k] >= RB) ldRow[k] = -1;
    }
    const int jj = T < 64 ? (int)__umulhi((unsigned)lane, rT) : 0, ii = lane - jj * T;     // this lane's probe of a pass
    const int groups = (ny + K - 1) / K, chunks = (rows + RB - 1) / RB;
    unsigned pf[TZ_SEG_REGS];
    // a wavefront walks (pass, chunk) pairs; the pair after the current one is tracked by increments (no divisions in the loop)
    struct Pos { int g, q, c; };
    auto advance = [&](Pos& p) { if (++p.c == chunks) { p.c = 0; p.q += TEAM; while (p.q >= ntiles) { p.q -= ntiles; p.g++; } } };
    auto fetch = [&](const Pos& p)                                          // chunk -> registers
    {
      const int r0 = p.c * RB;
      const Pel* org0 = first + (ptrdiff_t)(p.g * K) * step * rs + p.q * T * step;
      const int nxq = min(T, nx - p.q * T), pOff = (int)((reinterpret_cast<uintptr_t>(org0) & 2) >> 1);
      const int need = (pOff + (nxq - 1) * step + w + 1) >> 1;             // dwords that hold samples some probe of the tile reads
      const unsigned* gp = reinterpret_cast<const unsigned*>(reinterpret_cast<uintptr_t>(org0) & ~(uintptr_t)3);
#pragma unroll
      for (int k = 0; k < TZ_SEG_REGS; k++)
      {
        const bool on = ldRow[k] >= 0 && r0 + ldRow[k] < rows && p.g * K + ldSeg[k] < ny;
        pf[k] = on ? gp[(ptrdiff_t)ldSeg[k] * step * (rs >> 1) + (ptrdiff_t)(r0 + ldRow[k]) * gstep + min(ldCol[k], need - 1)] : 0u;
      }
    };
    Pos cur = { 0, wave, 0 };
    while (cur.q >= ntiles) { cur.q -= ntiles; cur.g++; }
    Pos nxt = cur;
    if (cur.g < groups) fetch(cur);
    unsigned acc = 0;
    while (cur.g < groups)
    {
      const int r0 = cur.c * RB, g = cur.g, q = cur.q, c = cur.c;
#pragma unroll
      for (int k = 0; k < TZ_SEG_REGS; k++) if (ldRow[k] >= 0) segL[lane + 64 * k] = pf[k] ^ bias;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
      advance(nxt);
      if (nxt.g < groups) fetch(nxt);
      const int j = g * K + jj, i = q * T + ii;
      if (jj < K && j < ny && ii < T && i < nx)
      {
        const Pel* org0 = first + (ptrdiff_t)(g * K) * step * rs + q * T * step;
        const int pOff = (int)((reinterpret_cast<uintptr_t>(org0) & 2) >> 1);
        const int myOff = ii * step + pOff, myDw = myOff >> 1;
        const unsigned sh = (unsigned)(myOff & 1) << 4;
        const int nr = min(RB, rows - r0);
        for (int r = 0; r < nr; r++)
        {
          const unsigned* o = orgL + (r0 + r) * halfW;
          const unsigned* sp = segL + (r * K + jj) * nd + myDw;
          unsigned d0 = sp[0];
          for (int k = 0; k < halfW; k += 4)
          {
            const uint4 ov = *reinterpret_cast<const uint4*>(o + k);
            const unsigned d1 = sp[k + 1], d2 = sp[k + 2], d3 = sp[k + 3], d4 = sp[k + 4];
            acc = __builtin_amdgcn_sad_u16(ov.x, __builtin_amdgcn_alignbit(d1, d0, sh), acc);
            acc = __builtin_amdgcn_sad_u16(ov.y, __builtin_amdgcn_alignbit(d2, d1, sh), acc);
            acc = __builtin_amdgcn_sad_u16(ov.z, __builtin_amdgcn_alignbit(d3, d2, sh), acc);
            acc = __builtin_amdgcn_sad_u16(ov.w, __builtin_amdgcn_alignbit(d4, d3, sh), acc);
            d0 = d4;
          }
        }
        if (c == chunks - 1)
        {
          const unsigned long long cost = ((unsigned long long)acc << subShift) + mvcost(R.win.left + i * step, R.win.top + j * step);
          key = min(key, (cost << 16) | (unsigned)(j * nx + i));
          acc = 0;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); __builtin_amdgcn_wave_barrier();
      cur = nxt;
    }
    // fold the lanes of this wavefront (the tail of round() folds across groups of G lanes only)
    for (int m = 1; m < 64; m <<= 1) { const unsigned long long ok = __shfl_xor(key, m); key = min(key, ok); }
    return true;
  }

  __device__ __forceinline__ void round(const TzRound& R)
  {
    // lanes per probe G = LX * RP: split the rows RP ways while the team has lanes to spare for this round
    const int rows = h >> subShift, n = R.n;
    int RP = 1;
    while (LX * RP * 2 <= 64 && RP * 2 <= rows && LX * RP * 2 * n <= TL) RP <<= 1;
    const int G = LX * RP, CPT = TL / G;
    const int sub = tl & (G - 1), lx = sub & (LX - 1), rp = sub / LX, lc = tl / G;
    unsigned long long key = ~0ull;
    int c0 = 0;
    if (R.kind == 3 && raster_rows(R, key)) c0 = n;           // raster round through per-row LDS segments
    for (; c0 + 4 * CPT <= n; c0 += 4 * CPT) pass<4>(R, c0, CPT, G, lc, lx, rp, RP, key);
    for (; c0 < n; c0 += CPT) pass<1>(R, c0, CPT, G, lc, lx, rp, RP, key);
    for (int m = G; m < 64; m <<= 1)
    {
      const unsigned long long o = __shfl_xor(key, m);
      key = min(key, o);
    }
    if (TEAM == 4)
    {
      const int wave = threadIdx.x >> 6;
      __syncthreads();                                   // the slots of the previous round have been read
      if ((threadIdx.x & 63) == 0) keyL[wave] = key;
      __syncthreads();
      key = min(min(keyL[0], keyL[1]), min(keyL[2], keyL[3]));
    }
    key = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(key >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)key);
    if (key != ~0ull && (key >> 16) < bestSad)
    {
      int x, y, pn; unsigned dd;
      candidate(R, (int)(key & 0xFFFFu), x, y, pn, dd);
      bestSad = key >> 16; bestX = x; bestY = y; bestDist = dd; bestRound = 0; pointNr = pn;
    }
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

// state of a PU whose raster stage runs as its own launch (split form, see vvcgpu_tz_search_batch)
struct TzSave { unsigned long long bestSad; int bestX, bestY; unsigned bestDist, bestRound; int pointNr, deferred; int left, top, right, bottom; int x0, y0, nx, ny; int reserved; int pad; };

template <int TEAM>
__global__ __launch_bounds__(256) void tz_search_kernel(const Pel* __restrict__ org, int os, const Pel* __restrict__ ref, int rs,
                                                        const vvcgpu_tz_pu* __restrict__ pus, int n, vvcgpu_tz_cfg cfg,
                                                        vvcgpu_search_best* __restrict__ results, int ldsDwords, int phase, TzSave* __restrict__ save,
                                                        vvcgpu_search_blk* __restrict__ rblk, VvcRasterPer* __restrict__ rper,
                                                        const vvcgpu_search_best* __restrict__ rbest)
{
  extern __shared__ __attribute__((aligned(16))) unsigned orgL[];          // ldsDwords dwords: the sub-sampled original block(s) of the team
  __shared__ unsigned long long keyL[4];
  __shared__ int negL[4];
  __shared__ unsigned segS[4][64 * TZ_SEG_REGS + 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = __builtin_amdgcn_readfirstlane(TEAM == 4 ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave);
  if (b >= n) return;                                          // TEAM 4: the whole workgroup leaves; TEAM 1: no barrier is used
  const vvcgpu_tz_pu pu = pus[b];
  if (phase == 2 && !save[b].deferred) return;                 // finished in the first launch (team-uniform)

  TzTeam<TEAM> s;
  s.org = org + (ptrdiff_t)pu.org_y * os + pu.org_x; s.ref = ref; s.os = os; s.rs = rs;
  s.w = pu.w; s.h = pu.h; s.subShift = pu.sub_shift; s.refX = pu.ref_x; s.refY = pu.ref_y;
  s.rx0 = cfg.ref_x0; s.ry0 = cfg.ref_y0; s.rx1 = cfg.ref_x1 - pu.w; s.ry1 = cfg.ref_y1 - pu.h;
  s.horMax = (cfg.pic_w + 8 - pu.pos_x - 1) << 2; s.horMin = (-cfg.max_cu_w - 8 - pu.pos_x + 1) << 2;
  s.verMax = (cfg.pic_h + 8 - pu.pos_y - 1) << 2; s.verMin = (-cfg.max_cu_h - 8 - pu.pos_y + 1) << 2;
  s.lambda = cfg.lambda; s.predHor = pu.pred_hor; s.predVer = pu.pred_ver; s.costScale = cfg.cost_scale; s.imvShift = cfg.imv_shift;
  s.tl = TEAM == 4 ? (int)threadIdx.x : lane;
  int LX = 1; while (4 * LX < pu.w) LX <<= 1;                 // quads along a row, rounded up to a power of two (w <= 128 -> LX <= 32)
  s.LX = LX;
  s.keyL = keyL;
  s.segL = segS[wave];

  // the sub-sampled original block -> LDS as packed pairs (w is a multiple of 4: VVC block widths are 4, 8, 12, 16, 24, ...)
  {
    const int rows = pu.h >> pu.sub_shift, halfW = pu.w >> 1, ndw = rows * halfW;
    const int slice = TEAM == 4 ? ldsDwords : ldsDwords / 4;
    unsigned* dst = orgL + (TEAM == 4 ? 0 : wave * slice);
    const bool fits = ndw <= slice && (pu.w & 3) == 0;
    int neg = 0;
    if (fits)
    {
      const int ostep = os << pu.sub_shift;
      for (int i = s.tl; i < ndw; i += s.TL)
      {
        const int j = i / halfW, k = i - j * halfW;
        const Pel* o = s.org + (ptrdiff_t)j * ostep + 2 * k;
        const int a = o[0], c = o[1];
        neg |= (a | c) < 0;
        dst[i] = ((unsigned)a & 0xFFFFu) | ((unsigned)c << 16);
      }
    }
    neg = __ballot(neg != 0) != 0ull;
    if (TEAM == 4)
    {
      if (lane == 0) negL[wave] = neg;
      __syncthreads();
      neg = negL[0] | negL[1] | negL[2] | negL[3];
    }
    s.bias = neg ? 0x80008000u : 0u;
    if (fits && neg)
      for (int i = s.tl; i < ndw; i += s.TL) dst[i] ^= 0x80008000u;
    if (TEAM == 4) __syncthreads();
    else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }   // one wavefront: LDS is in order
    s.orgL = fits ? dst : nullptr;
  }

  const bool ext = (pu.flags & VVCGPU_TZ_EXTENDED) != 0, fast = (pu.flags & VVCGPU_TZ_FAST) != 0;
  const int rasterStep = fast ? 8 : 5, range = pu.reserved[0] > 0 ? min(pu.reserved[0], cfg.search_range) : cfg.search_range;   // per-PU range (the adaptive search range is per reference picture), never beyond cfg.search_range: the split form sizes its raster launch from that

  int mx = pu.start_x, my = pu.start_y;
  s.clip(mx, my); mx = (mx + 2) >> 2; my = (my + 2) >> 2;
  int p2x = pu.pred2_x << 2, p2y = pu.pred2_y << 2;
  s.clip(p2x, p2y); p2x = (p2x + 2) >> 2; p2y = (p2y + 2) >> 2;
  s.bestSad = ~0ull >> 16; s.bestX = s.bestY = 0; s.bestDist = 0; s.bestRound = 0; s.pointNr = 0;
  s.sr = TzRange{ 0, 0, 0, 0 };

  // xTZSearch (:1971-2252) as a state machine: every state prepares at most one round, so the probe code exists once
  enum { START, ZERO, PRED2, RANGE, FIRST, FIRST_STOP, ZERO_NBH, TWO_POINT, RASTER, STAR_BEGIN, STAR, STAR_STOP, STAR_TWO_POINT, DONE };
  int state = START, d = 1, startX = 0, startY = 0;
  bool bestCandidateZero = false;
  if (phase == 2)
  {
    // resume behind the raster stage: the state of the first launch, then the raster launch's best candidate under the round rule
    // (strictly better than what the earlier rounds found; the raster's own ties were resolved in visiting order by its key)
    const TzSave sv = save[b];
    s.bestSad = sv.bestSad; s.bestX = sv.bestX; s.bestY = sv.bestY; s.bestDist = sv.bestDist; s.bestRound = sv.bestRound; s.pointNr = sv.pointNr;
    s.sr.left = sv.left; s.sr.top = sv.top; s.sr.right = sv.right; s.sr.bottom = sv.bottom;
    const unsigned long long key = rbest[b].cost;
    if (key != ~0ull && (key >> 24) < s.bestSad)
    {
      const int idx = (int)(key & 0xFFFFFFu), j = idx / sv.nx, i = idx - j * sv.nx;
      s.bestSad = key >> 24; s.bestX = sv.x0 + 5 * i; s.bestY = sv.y0 + 5 * j; s.bestDist = 5u; s.bestRound = 0; s.pointNr = 0;
    }
    state = STAR_BEGIN;
  }
  while (state != DONE)
  {
    TzRound R;
    R.kind = 0; R.n = 0; R.x = 0; R.y = 0; R.d = 0; R.corners = 0; R.nx = 1; R.rnx = 0; R.win = s.sr;
    switch (state)
    {
    case START:                                                // :2023
      R.n = 1; R.x = mx; R.y = my; state = ZERO; break;
    case ZERO:                                                 // :2026-2034
      if (!fast && (mx != 0 || my != 0) && (s.bestX != 0 || s.bestY != 0)) { R.n = 1; }
      state = PRED2; break;
    case PRED2:                                                // :2038-2051
      if ((pu.flags & VVCGPU_TZ_PRED2) && (mx != p2x || my != p2y) && (p2x != s.bestX || p2y != s.bestY)) { R.n = 1; R.x = p2x; R.y = p2y; }
      state = RANGE; break;
    case RANGE:                                                // :2052-2070
      s.set_range(s.bestX, s.bestY, range >> (fast ? 1 : 0));
      startX = s.bestX; startY = s.bestY; bestCandidateZero = s.bestX == 0 && s.bestY == 0; d = 1;
      state = FIRST; break;
    case FIRST:                                                // :2072-2088
      if (d <= range) { s.bestRound += 1; R.kind = 1; R.n = d <= 8 ? 8 : 16; R.x = startX; R.y = startY; R.d = d; R.corners = ext; d *= 2; state = FIRST_STOP; }
      else { d = 1; state = ZERO_NBH; }
      break;
    case FIRST_STOP:
      if (cfg.first_search_stop && s.bestRound >= 3) { d = 1; state = ZERO_NBH; } else state = FIRST;
      break;
    case ZERO_NBH:                                             // :2111-2126 (the :2090-2109 branch is dead: both of its flags are bExtendedSettings)
      if (ext && !bestCandidateZero && d <= (range >> 1)) { s.bestRound += 1; R.kind = 1; R.n = d <= 8 ? 8 : 16; R.d = d; d *= 2; }
      else state = TWO_POINT;
      break;
    case TWO_POINT:                                            // :2129-2133
      if (s.bestDist == 1) { s.bestDist = 0; R.kind = 2; R.n = 2; R.x = s.bestX; R.y = s.bestY; R.d = s.pointNr; }
      state = RASTER; break;
    case RASTER:                                               // :2136-2171
    {
      int step = 0; TzRange l = s.sr;
      if (ext)
      {
        step = rasterStep;
        if (!((int)s.bestDist >= rasterStep)) { step++; l.left /= 2; l.right /= 2; l.top /= 2; l.bottom /= 2; }
      }
      else if ((int)s.bestDist >= rasterStep) step = rasterStep;
      if (step)
      {
        s.bestDist = (unsigned)step;
        if (l.right >= l.left && l.bottom >= l.top)
        {
          R.kind = 3; R.d = step; R.win = l; R.nx = (l.right - l.left) / step + 1;
          R.n = R.nx * ((l.bottom - l.top) / step + 1);
          R.rnx = R.nx > 1 ? (unsigned)(0x100000000ull / (unsigned)R.nx) + 1u : 0u;
          // first launch of the split form: a plain step-5 raster whose every probe lies inside the readable rectangle (no clamping of the
          // block origin) is left to the raster launch; this PU resumes behind it in the third launch
          const int ny = (l.bottom - l.top) / step + 1;
          // (the raster kernel stages whole 16-byte words of the window rows: 8 samples of slack on both sides)
          if (phase == 1 && step == 5 && R.nx <= 40 && ny <= 40 && pu.w == (cfg.uniform_pu & 0xFFFF) && pu.h == ((cfg.uniform_pu >> 16) & 0xFFFF) && pu.sub_shift == 1 &&
              s.refX + l.left - 8 >= s.rx0 && s.refX + l.left + (R.nx - 1) * 5 + 8 <= s.rx1 && s.refY + l.top >= s.ry0 && s.refY + l.top + (ny - 1) * 5 <= s.ry1)
          {
            if (s.tl == 0)
            {
              TzSave sv;
              sv.bestSad = s.bestSad; sv.bestX = s.bestX; sv.bestY = s.bestY; sv.bestDist = s.bestDist; sv.bestRound = s.bestRound; sv.pointNr = s.pointNr;
              sv.left = s.sr.left; sv.top = s.sr.top; sv.right = s.sr.right; sv.bottom = s.sr.bottom;
              sv.x0 = l.left; sv.y0 = l.top; sv.nx = R.nx; sv.ny = ny; sv.deferred = 1; sv.reserved = 0;
              save[b] = sv;
              rblk[b] = vvcgpu_search_blk{ pu.org_x, pu.org_y, pu.ref_x, pu.ref_y };       // reference position of the zero vector, as in vvcgpu_sad_search
              rper[b] = VvcRasterPer{ 1, R.nx, ny, l.left, l.top, pu.pred_hor, pu.pred_ver, 0 };
            }
            return;                                            // team-uniform
          }
        }
      }
      state = STAR_BEGIN; break;
    }
    case STAR_BEGIN:                                           // :2207-2214
      if (s.bestDist > 0) { startX = s.bestX; startY = s.bestY; s.bestDist = 0; s.pointNr = 0; d = 1; state = STAR; }
      else state = DONE;
      break;
    case STAR:                                                 // :2215-2229
      if (d < range + 1) { s.bestRound += 1; R.kind = 1; R.n = d <= 8 ? 8 : 16; R.x = startX; R.y = startY; R.d = d; R.corners = ext; d *= 2; state = STAR_STOP; }
      else state = STAR_TWO_POINT;
      break;
    case STAR_STOP:
      state = (fast && s.bestRound >= 2) ? STAR_TWO_POINT : STAR; break;
    case STAR_TWO_POINT:                                       // :2231-2240
      if (s.bestDist == 1) { s.bestDist = 0; if (s.pointNr != 0) { R.kind = 2; R.n = 2; R.x = s.bestX; R.y = s.bestY; R.d = s.pointNr; } }
      state = STAR_BEGIN; break;
    default: state = DONE; break;
    }
    if (R.n > 0) s.round(R);
  }

  if (s.tl == 0)
  {
    vvcgpu_search_best r;
    r.x = s.bestX; r.y = s.bestY; r.cost = s.bestSad; r.sad = s.bestSad - s.mvcost(s.bestX, s.bestY);
    results[b] = r;
    if (phase == 1) { save[b].deferred = 0; rper[b].active = 0; rblk[b] = vvcgpu_search_blk{ pu.org_x, pu.org_y, pu.ref_x, pu.ref_y }; }
  }
}

// TZ results -> the block list of the fractional refinement (reference block displaced by the integer MV, :1816) + per-PU predictors
__global__ __launch_bounds__(256) void tz_to_frac_kernel(const vvcgpu_tz_pu* __restrict__ pus, const vvcgpu_search_best* __restrict__ best, int n,
                                                         vvcgpu_frac_blk* __restrict__ blk, int* __restrict__ preds)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const vvcgpu_tz_pu p = pus[i];
  const int bx = best[i].x, by = best[i].y;
  vvcgpu_frac_blk f;
  f.org_x = p.org_x; f.org_y = p.org_y; f.ref_x = p.ref_x + bx; f.ref_y = p.ref_y + by; f.mv_x = bx; f.mv_y = by;
  blk[i] = f;
  preds[2 * i] = p.pred_hor; preds[2 * i + 1] = p.pred_ver;
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
  hipStream_t st = (hipStream_t)stream;
  // LDS for the teams' original blocks: 32 KB serve any PU (8 KB per wavefront: 64x128 sub-sampled; one workgroup per PU: 128x128); with the
  // caller's word that the PUs are w x h the allocation shrinks to what they need (16x16: 1 KB per workgroup), which doubles the resident
  // wavefronts of this latency-bound kernel (a PU that is larger after all reads its block sample-wise: correct, slower)
  int ldsDwords = TZ_LDS_DWORDS;
  {
    const int uw0 = c.uniform_pu & 0xFFFF, uh0 = (c.uniform_pu >> 16) & 0xFFFF;
    if (c.uniform_pu != 0 && uw0 >= 4 && uh0 >= 4 && uw0 <= 128 && uh0 <= 128)
    {
      const int per = (((uh0 >> 1) * (uw0 >> 1)) + 63) & ~63;
      ldsDwords = c.wg_per_pu ? per : 4 * per;
      if (ldsDwords > TZ_LDS_DWORDS) ldsDwords = TZ_LDS_DWORDS;
    }
  }
  auto launch = [&](int phase, TzSave* save, vvcgpu_search_blk* rblk, VvcRasterPer* rper, const vvcgpu_search_best* rbest)
  {
    if (c.wg_per_pu)
      hipLaunchKernelGGL(tz_search_kernel<4>, dim3(n), dim3(256), (size_t)ldsDwords * 4, st, org, org_stride, ref, ref_stride, pus, n, c, results, ldsDwords, phase, save, rblk, rper, rbest);
    else
      hipLaunchKernelGGL(tz_search_kernel<1>, dim3((n + 3) / 4), dim3(256), (size_t)ldsDwords * 4, st, org, org_stride, ref, ref_stride, pus, n, c, results, ldsDwords, phase, save, rblk, rper, rbest);
  };
  // Split form (cfg.uniform_pu = h << 16 | w: the caller states that EVERY PU of the batch is w x h with 2:1 row sub-sampling): the raster stage,
  // 86 % of the probes of a search that enters it, runs as the quad raster kernel of dist.hip between two launches of the state machine --
  // the in-kernel raster round works one wavefront per PU at ~8 % of the v_sad_u16 issue rate, the raster kernel at ~60 %.
  const int uw = c.uniform_pu & 0xFFFF, uh = (c.uniform_pu >> 16) & 0xFFFF;
  const int gridMax = (2 * c.search_range) / 5 + 1;
  if (c.uniform_pu != 0 && (uw == 16 || uw == 32 || uw == 64) && (uh == 16 || uh == 32 || uh == 64) && gridMax <= 40 &&
      (org_stride & 1) == 0 && (ref_stride & 7) == 0 && ((uintptr_t)org & 3) == 0 && ((uintptr_t)ref & 15) == 0)
  {
    const size_t packedDw = (size_t)n * 2 * (uh >> 1) * (uw >> 1);
    const size_t bytes = (size_t)n * (sizeof(TzSave) + sizeof(vvcgpu_search_blk) + sizeof(VvcRasterPer) + sizeof(vvcgpu_search_best)) + packedDw * 4 + 256;
    unsigned char* ws = static_cast<unsigned char*>(vvcgpu_scratch(st, bytes));
    if (!ws) return VVCGPU_E_DEVICE;
    TzSave* save = reinterpret_cast<TzSave*>(ws);
    vvcgpu_search_best* rbest = reinterpret_cast<vvcgpu_search_best*>(save + n);
    VvcRasterPer* rper = reinterpret_cast<VvcRasterPer*>(rbest + n);
    vvcgpu_search_blk* rblk = reinterpret_cast<vvcgpu_search_blk*>(rper + n);
    unsigned* packed = reinterpret_cast<unsigned*>((reinterpret_cast<uintptr_t>(rblk + n) + 63) & ~(uintptr_t)63);
    launch(1, save, rblk, rper, nullptr);
    VVC_LAUNCH_CHECK();
    vvcgpu_mvcost mv;
    mv.lambda = c.lambda; mv.pred_hor = 0; mv.pred_ver = 0; mv.cost_scale = c.cost_scale; mv.imv_shift = c.imv_shift;
    const int rc = vvcgpu_raster_per_block_launch(org, org_stride, ref, ref_stride, rblk, rper, n, uw, uh, 1, gridMax, gridMax, &mv, rbest, packed, st);
    if (rc != VVCGPU_OK) return rc;
    launch(2, save, rblk, rper, rbest);
    VVC_LAUNCH_CHECK();
    return VVCGPU_OK;
  }
  launch(0, nullptr, nullptr, nullptr, nullptr);
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}

int vvcgpu_me_batch(const vvc_pel* org, int org_stride, const vvc_pel* ref, int ref_stride, const vvcgpu_tz_pu* pus, int n, int w, int h,
                    const vvcgpu_tz_cfg* cfg_host, int bit_depth, int clp_min, int clp_max, int use_hadamard,
                    vvcgpu_search_best* int_results, vvcgpu_frac_result* frac_results, void* stream)
{
  VVC_CHECK_ARG(n >= 0, "me_batch: n %d", n);
  if (n == 0) return VVCGPU_OK;
  VVC_CHECK_ARG(cfg_host && int_results && frac_results, "me_batch: null pointer");
  vvcgpu_tz_cfg cfgu = *cfg_host;
  if (cfgu.uniform_pu == 0 && (w == 16 || w == 32 || w == 64) && (h == 16 || h == 32 || h == 64)) cfgu.uniform_pu = (h << 16) | w;   // the chain's PUs are w x h
  int rc = vvcgpu_tz_search_batch(org, org_stride, ref, ref_stride, pus, n, &cfgu, int_results, stream);
  if (rc != VVCGPU_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  unsigned char* scratch = static_cast<unsigned char*>(vvcgpu_scratch(st, (size_t)n * (sizeof(vvcgpu_frac_blk) + 2 * sizeof(int))));
  if (!scratch) return VVCGPU_E_DEVICE;
  vvcgpu_frac_blk* blk = reinterpret_cast<vvcgpu_frac_blk*>(scratch);
  int* preds = reinterpret_cast<int*>(scratch + (size_t)n * sizeof(vvcgpu_frac_blk));
  hipLaunchKernelGGL(tz_to_frac_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, pus, int_results, n, blk, preds);
  VVC_LAUNCH_CHECK();
  vvcgpu_mvcost mv;
  mv.lambda = cfg_host->lambda; mv.pred_hor = 0; mv.pred_ver = 0; mv.cost_scale = 0; mv.imv_shift = 0;
  return vvcgpu_frac_refine_launch(org, org_stride, ref, ref_stride, blk, n, w, h, bit_depth, clp_min, clp_max, use_hadamard, &mv, preds,
                                   frac_results, stream);
}

}  // extern "C"
