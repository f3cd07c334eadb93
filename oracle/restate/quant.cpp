// oracle/restate/quant.cpp -- TEST INFRASTRUCTURE: scalar restatement of the de-quantisers (next row N1).
//   scan order      g_scanOrder[SCAN_GROUPED_4x4][SCAN_DIAG]            CommonLib/Rom.cpp:357-405 (ScanGenerator, diagonal)
//   Quant::dequant  (flat scaling, HM_QTBT_AS_IN_JEM_QUANT)             CommonLib/Quant.cpp:277-428
//   dependent quantisation  DQIntern::Quantizer::dequantBlock           CommonLib/DepQuant.cpp:708-785
//   getTransformShift                                                    CommonLib/ChromaFormat.h:117-120
// Pinned against the compiled reference by tests/golden/dequant.npz (tests/test_oracle_golden.py).
#include "orc_common.h"
#include <vector>

static inline int ilog2q(int v) { int l = 0; while ((1 << l) < v) l++; return l; }
static const int kInvQuantScales[6] = { 40, 45, 51, 57, 64, 72 };               // g_invQuantScales, Rom.cpp:470-473

// diagonal scan of a cw x ch grid: diagonals x + y = d, each walked from its bottom-left end upwards
static void diagScan(int cw, int ch, std::vector<int>& xs, std::vector<int>& ys)
{
  xs.clear(); ys.clear();
  for (int d = 0; d < cw + ch - 1; d++)
    for (int y = std::min(d, ch - 1); y >= 0; y--)
    {
      const int x = d - y;
      if (x < cw) { xs.push_back(x); ys.push_back(y); }
    }
}

ORC_API int orc_scan_order(int w, int h, uint32_t* out)
{
  const int lg = ((w & 3) + (h & 3)) > 0 ? 1 : 2;                                 // 2x2 groups when a side is 2 (:360-361)
  const int gw = 1 << lg, gh = 1 << lg;
  std::vector<int> gx, gy, px, py;
  diagScan(w >> lg, h >> lg, gx, gy);
  diagScan(gw, gh, px, py);
  int n = 0;
  for (size_t g = 0; g < gx.size(); g++)
    for (size_t k = 0; k < px.size(); k++)
      out[n++] = (uint32_t)((gy[g] * gh + py[k]) * w + gx[g] * gw + px[k]);
  return 0;
}

ORC_API int orc_dequant(int depQuant, int bd, int qp, int transformSkip, const TCoeff* level, TCoeff* out, int w, int h)
{
  (void)transformSkip;                                                            // only matters with extended precision (off)
  const int n = w * h, lw = ilog2q(w), lh = ilog2q(h);
  const int transformShift = 15 - bd - ((lw + lh) >> 1);
  const bool sqrt2 = ((lw + lh) & 1) != 0;                                        // needsSqrt2Scale == needsBlockSizeTrafoScale here
  const int64_t minT = -(1 << 15), maxT = (1 << 15) - 1;
  if (!depQuant)
  {
    const int per = qp / 6, rem = qp % 6;
    const int rightShift = (sqrt2 ? 8 : 0) + (6 - (transformShift + per));
    const int64_t scale = (int64_t)kInvQuantScales[rem] * (sqrt2 ? 181 : 1);
    const int targetBits = std::min(16, 32 + rightShift - 7);
    const int64_t inMin = -(1ll << (targetBits - 1)), inMax = (1ll << (targetBits - 1)) - 1;
    for (int i = 0; i < n; i++)
    {
      const int64_t c = std::min(std::max((int64_t)level[i], inMin), inMax);
      const int64_t v = rightShift > 0 ? (c * scale + (1ll << (rightShift - 1))) >> rightShift : (c * scale) << -rightShift;
      out[i] = (TCoeff)std::min(std::max(v, minT), maxT);
    }
    return 0;
  }
  std::vector<uint32_t> scan(n);
  orc_scan_order(w, h, scan.data());
  for (int i = 0; i < n; i++) out[i] = 0;
  const int qpDQ = qp + 1, qpPer = qpDQ / 6, qpRem = qpDQ - 6 * qpPer;
  int shift = 6 + 1 - qpPer - transformShift + (sqrt2 ? 8 : 0);
  int64_t invQScale = (int64_t)kInvQuantScales[qpRem] * (sqrt2 ? 181 : 1);
  if (shift < 0) { invQScale <<= -shift; shift = 0; }
  const int64_t add = (1ll << shift) >> 1;
  int state = 0;
  for (int scanIdx = n - 1; scanIdx >= 0; scanIdx--)                              // zeros above the last level keep state 0
  {
    const int pos = (int)scan[scanIdx];
    const int lv = level[pos];
    if (lv)
    {
      const int64_t qIdx = ((int64_t)lv << 1) + (lv > 0 ? -(state >> 1) : (state >> 1));
      out[pos] = (TCoeff)std::min(std::max((qIdx * invQScale + add) >> shift, minT), maxT);
    }
    state = (32040 >> ((state << 2) + ((lv & 1) << 1))) & 3;
  }
  return 0;
}

extern "C" int orc_tr_inv(const TCoeff* coeff, Pel* resi, int stride, int w, int h, int trHor, int trVer, int bd);

ORC_API int orc_dequant_tr_inv_batch(const TCoeff* levelBase, Pel* resiBase, const vvcgpu_dqtr_desc* d, int n, int bd, TCoeff* coeffOut)
{
  std::vector<TCoeff> tmp;
  for (int i = 0; i < n; i++)
  {
    const int cnt = d[i].w * d[i].h;
    tmp.resize(cnt);
    TCoeff* c = coeffOut ? coeffOut + d[i].level_off : tmp.data();
    orc_dequant(d[i].dep_quant, bd, d[i].qp, d[i].tr_hor == 3, levelBase + d[i].level_off, c, d[i].w, d[i].h);
    orc_tr_inv(c, resiBase + d[i].resi_off, d[i].resi_stride, d[i].w, d[i].h, d[i].tr_hor, d[i].tr_ver, bd);
  }
  return 0;
}


// ---- forward scalar quantisation without RDOQ: Quant::quant (Quant.cpp:721-834) + sign bit hiding xSignBitHidingHDQ (:142-273) ----
// flat scaling (g_quantScales, Quant.cpp), HM_QTBT_AS_IN_JEM_QUANT block-size scale for sqrt(2) shapes; scan = diagonal 4x4-grouped.

ORC_API uint32_t orc_quant(const TCoeff* coef, TCoeff* level, int w, int h, int bd, int qp, int intraSlice, int signHiding)
{
  static const int quantScales[6] = { 26214, 23302, 20560, 18396, 16384, 14564 };
  int lw = 0, lh = 0; while ((1 << (lw + 1)) <= w) lw++; while ((1 << (lh + 1)) <= h) lh++;
  const int per = qp / 6, rem = qp % 6, n = w * h;
  int transformShift = 15 - bd - ((lw + lh) >> 1);
  int whScale = 1;
  if ((lw + lh) & 1) { transformShift += 7; whScale = 181; }
  const int qBits = 14 + per + transformShift, qBits8 = qBits - 8;
  const int64_t add = (int64_t)(intraSlice ? 171 : 85) << (qBits - 9);
  std::vector<TCoeff> deltaU(n);
  uint32_t absSum = 0;
  for (int i = 0; i < n; i++)
  {
    const TCoeff c = coef[i];
    const int64_t tmp = (int64_t)std::abs(c) * quantScales[rem];
    const TCoeff q = (TCoeff)((tmp * whScale + add) >> qBits);
    deltaU[i] = (TCoeff)((tmp * whScale - ((int64_t)q << qBits)) >> qBits8);
    absSum += q;
    level[i] = clip3i(-32768, 32767, c < 0 ? -q : q);
  }
  if (!(signHiding && w >= 4 && h >= 4 && (int32_t)absSum >= 2)) return absSum;     // uiAbsSum is a TCoeff (int) in the reference
  std::vector<uint32_t> scan(n);
  orc_scan_order(w, h, scan.data());
  const int TMAX = 0x7fffffff;
  int lastCG = -1;
  for (int subSet = (n - 1) >> 4; subSet >= 0; subSet--)
  {
    const int subPos = subSet << 4;
    int first = 16, last = -1, sum = 0;
    for (int k = 15; k >= 0; k--) if (level[scan[k + subPos]]) { last = k; break; }
    for (int k = 0; k < 16; k++) if (level[scan[k + subPos]]) { first = k; break; }
    for (int k = first; k <= last; k++) sum += level[scan[k + subPos]];
    if (last >= 0 && lastCG == -1) lastCG = 1;
    if (last - first >= 4)
    {
      const unsigned signbit = level[scan[subPos + first]] > 0 ? 0 : 1;
      if (signbit != (unsigned)(sum & 1))
      {
        int curCost = TMAX, minCostInc = TMAX, minPos = -1, finalChange = 0, curChange = 0;
        for (int k = (lastCG == 1 ? last : 15); k >= 0; k--)
        {
          const int pos = scan[k + subPos];
          if (level[pos] != 0)
          {
            if (deltaU[pos] > 0) { curCost = -deltaU[pos]; curChange = 1; }
            else if (k == first && std::abs(level[pos]) == 1) curCost = TMAX;
            else { curCost = deltaU[pos]; curChange = -1; }
          }
          else if (k < first)
          {
            const unsigned thisSign = coef[pos] >= 0 ? 0 : 1;
            if (thisSign != signbit) curCost = TMAX;
            else { curCost = -deltaU[pos]; curChange = 1; }
          }
          else { curCost = -deltaU[pos]; curChange = 1; }
          if (curCost < minCostInc) { minCostInc = curCost; finalChange = curChange; minPos = pos; }
        }
        if (level[minPos] == 32767 || level[minPos] == -32768) finalChange = -1;
        if (coef[minPos] >= 0) level[minPos] += finalChange; else level[minPos] -= finalChange;
      }
    }
    if (lastCG == 1) lastCG = 0;
  }
  return absSum;
}

ORC_API int orc_quant_batch(const TCoeff* coeffBase, TCoeff* levelBase, const vvcgpu_quant_desc* d, int n, int bd, uint32_t* absSum)
{
  for (int i = 0; i < n; i++)
    absSum[i] = orc_quant(coeffBase + d[i].coeff_off, levelBase + d[i].level_off, d[i].w, d[i].h, bd, d[i].qp, d[i].intra_slice, d[i].sign_hiding);
  return 0;
}
