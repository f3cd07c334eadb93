// oracle/restate/deblock.cpp -- TEST INFRASTRUCTURE: CPU restatement of the deblocking sample filters.
// Follows LoopFilter::xEdgeFilterLuma (CommonLib/LoopFilter.cpp:543-681), xEdgeFilterChroma (:684-838),
// xPelFilterLuma (:856-916), xPelFilterChroma (:928-949), xUseStrongFiltering (:960-970), xCalcDP/DQ
// (:972-980), tables sm_tcTable/sm_betaTable (:66-80), g_aucChromaScale[CHROMA_420] (Rom.cpp:528);
// pass order of loopFilterPic (:149-230): every vertical edge of the picture, then every horizontal edge.
// Edge selection / boundary strength arrive as maps (see include/vvcgpu.h, vvcgpu_deblock).
#include "orc_common.h"

static const uint8_t tcTable[66] = {
  0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,1,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,5,5,6,6,7,8,9,10,11,13,14,16,18,20,22,24,
  26,28,30,32,34,36,38,40,42,44,46,48 };
static const uint8_t betaTable[64] = {
  0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,6,7,8,9,10,11,12,13,14,15,16,17,18,20,22,24,26,28,30,32,34,36,38,40,42,44,46,48,50,52,
  54,56,58,60,62,64,66,68,70,72,74,76,78,80,82,84,86,88 };
static const uint8_t chromaScale420[70] = {
  0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,29,30,31,32,33,33,34,34,35,35,36,36,
  37,37,38,39,40,41,42,43,44,45,46,47,48,49,50,51,52,53,54,55,56,57,58,59,60,61,62,63 };
static const int MAXQP = 63, TCOFF = 2, QPMAPSZ = 70;

static inline void pelFilterLuma(Pel* s, int o, int tc, bool sw, bool noP, bool noQ, int thrCut, bool fP, bool fQ,
                                 int cmin, int cmax)
{
  const int m4 = s[0], m3 = s[-o], m5 = s[o], m2 = s[-2 * o], m6 = s[2 * o], m1 = s[-3 * o], m7 = s[3 * o], m0 = s[-4 * o];
  if (sw)
  {
    s[-o]     = (Pel)clip3i(m3 - 2 * tc, m3 + 2 * tc, (m1 + 2 * m2 + 2 * m3 + 2 * m4 + m5 + 4) >> 3);
    s[0]      = (Pel)clip3i(m4 - 2 * tc, m4 + 2 * tc, (m2 + 2 * m3 + 2 * m4 + 2 * m5 + m6 + 4) >> 3);
    s[-2 * o] = (Pel)clip3i(m2 - 2 * tc, m2 + 2 * tc, (m1 + m2 + m3 + m4 + 2) >> 2);
    s[o]      = (Pel)clip3i(m5 - 2 * tc, m5 + 2 * tc, (m3 + m4 + m5 + m6 + 2) >> 2);
    s[-3 * o] = (Pel)clip3i(m1 - 2 * tc, m1 + 2 * tc, (2 * m0 + 3 * m1 + m2 + m3 + m4 + 4) >> 3);
    s[2 * o]  = (Pel)clip3i(m6 - 2 * tc, m6 + 2 * tc, (m3 + m4 + m5 + 3 * m6 + 2 * m7 + 4) >> 3);
  }
  else
  {
    int delta = (9 * (m4 - m3) - 3 * (m5 - m2) + 8) >> 4;
    if (abs(delta) < thrCut)
    {
      delta = clip3i(-tc, tc, delta);
      s[-o] = (Pel)clip3i(cmin, cmax, m3 + delta);
      s[0]  = (Pel)clip3i(cmin, cmax, m4 - delta);
      const int tc2 = tc >> 1;
      if (fP) { const int d1 = clip3i(-tc2, tc2, ((((m1 + m3 + 1) >> 1) - m2 + delta) >> 1)); s[-2 * o] = (Pel)clip3i(cmin, cmax, m2 + d1); }
      if (fQ) { const int d2 = clip3i(-tc2, tc2, ((((m6 + m4 + 1) >> 1) - m5 - delta) >> 1)); s[o] = (Pel)clip3i(cmin, cmax, m5 + d2); }
    }
  }
  if (noP) { s[-o] = (Pel)m3; s[-2 * o] = (Pel)m2; s[-3 * o] = (Pel)m1; }
  if (noQ) { s[0] = (Pel)m4; s[o] = (Pel)m5; s[2 * o] = (Pel)m6; }
}

static inline bool useStrong(const Pel* s, int o, int d, int beta, int tc)
{
  const int m4 = s[0], m3 = s[-o], m7 = s[3 * o], m0 = s[-4 * o];
  const int ds = abs(m0 - m3) + abs(m7 - m4);
  return (ds < (beta >> 3)) && (d < (beta >> 2)) && (abs(m3 - m4) < ((tc * 5 + 1) >> 1));
}
static inline int calcDP(const Pel* s, int o) { return abs(s[-3 * o] - 2 * s[-2 * o] + s[-o]); }
static inline int calcDQ(const Pel* s, int o) { return abs(s[0] - 2 * s[o] + s[2 * o]); }

// one 4-line luma segment; s points at line 0, Q-side sample 0; o = step across the edge, ls = step along it
static void lumaSegment(Pel* s, int o, int ls, int bs, int qpP, int qpQ, bool noP, bool noQ, const vvcgpu_deblock_cfg& c)
{
  const int qp = (qpP + qpQ + 1) >> 1;
  const int scale = 1 << (c.bit_depth_luma - 8);
  const int idxTc = clip3i(0, MAXQP + TCOFF, qp + TCOFF * (bs - 1) + (c.tc_offset_div2 << 1));
  const int idxB = clip3i(0, MAXQP, qp + (c.beta_offset_div2 << 1));
  const int tc = tcTable[idxTc] * scale, beta = betaTable[idxB] * scale;
  const int side = (beta + (beta >> 1)) >> 3, thrCut = tc * 10;
  const int dp0 = calcDP(s, o), dq0 = calcDQ(s, o), dp3 = calcDP(s + 3 * ls, o), dq3 = calcDQ(s + 3 * ls, o);
  const int d0 = dp0 + dq0, d3 = dp3 + dq3, dp = dp0 + dp3, dq = dq0 + dq3, d = d0 + d3;
  if (d < beta)
  {
    const bool fP = dp < side, fQ = dq < side;
    const bool sw = useStrong(s, o, 2 * d0, beta, tc) && useStrong(s + 3 * ls, o, 2 * d3, beta, tc);
    for (int i = 0; i < 4; i++) pelFilterLuma(s + i * ls, o, tc, sw, noP, noQ, thrCut, fP, fQ, c.clp_min[0], c.clp_max[0]);
  }
}

static void chromaSegment(Pel* s, int o, int ls, int n, int qpP, int qpQ, int qpOff, bool noP, bool noQ, int comp,
                          const vvcgpu_deblock_cfg& c)
{
  int qp = ((qpP + qpQ + 1) >> 1) + qpOff;
  if (qp >= QPMAPSZ) qp -= 6;                            // 4:2:0 (:812-817)
  else if (qp >= 0) qp = chromaScale420[qp];             // getScaledChromaQP clips to [0, size-1]
  const int idxTc = clip3i(0, MAXQP + TCOFF, qp + TCOFF * (2 - 1) + (c.tc_offset_div2 << 1));
  const int tc = tcTable[idxTc] * (1 << (c.bit_depth_chroma - 8));
  for (int i = 0; i < n; i++)
  {
    Pel* q = s + i * ls;
    const int m4 = q[0], m3 = q[-o], m5 = q[o], m2 = q[-2 * o];
    const int delta = clip3i(-tc, tc, ((((m4 - m3) << 2) + m2 - m5 + 4) >> 3));
    if (!noP) q[-o] = (Pel)clip3i(c.clp_min[comp], c.clp_max[comp], m3 + delta);
    if (!noQ) q[0] = (Pel)clip3i(c.clp_min[comp], c.clp_max[comp], m4 - delta);
  }
}

ORC_API int orc_deblock(Pel* Y, int strideY, Pel* Cb, Pel* Cr, int strideC, int w, int h,
                        const uint8_t* edgeV, const uint8_t* edgeH, const int8_t* qpY, const int8_t* qpC,
                        const vvcgpu_deblock_cfg* cfg)
{
  const vvcgpu_deblock_cfg& c = *cfg;
  const int w4 = w >> 2, h4 = h >> 2;
  for (int dir = 0; dir < 2; dir++)
  {
    const uint8_t* em = dir == 0 ? edgeV : edgeH;
    for (int uy = 0; uy < h4; uy++)
      for (int ux = 0; ux < w4; ux++)
      {
        const int u = uy * w4 + ux;
        const int e = em[u];
        if (!e) continue;
        if (dir == 0 ? ux == 0 : uy == 0) continue;           // picture border: never an edge (leftEdge/topEdge, :408-415)
        const int uP = dir == 0 ? u - 1 : u - w4;
        const bool noP = (e >> 4) & 1, noQ = (e >> 5) & 1;
        const int bsY = e & 3, bsC = (e >> 2) & 3;
        const int x = ux * 4, y = uy * 4;
        if (bsY && (dir == 0 ? (x & 7) == 0 : (y & 7) == 0))
        {
          Pel* s = Y + y * strideY + x;
          lumaSegment(s, dir == 0 ? 1 : strideY, dir == 0 ? strideY : 1, bsY, qpY[uP], qpY[u], noP, noQ, c);
        }
        if (bsC > 1 && (dir == 0 ? (x & 15) == 0 : (y & 15) == 0))
        {
          Pel* planes[2] = { Cb, Cr };
          for (int k = 0; k < 2; k++)
          {
            Pel* s = planes[k] + (y >> 1) * strideC + (x >> 1);
            chromaSegment(s, dir == 0 ? 1 : strideC, dir == 0 ? strideC : 1, 2, qpC[uP], qpC[u],
                          k == 0 ? c.cb_qp_offset : c.cr_qp_offset, noP, noQ, 1 + k, c);
          }
        }
      }
  }
  return 0;
}
