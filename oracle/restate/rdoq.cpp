// oracle/restate/rdoq.cpp -- TEST INFRASTRUCTURE: scalar restatement of the rate-distortion optimised quantiser (next row N1).
//   QuantRDOQ::xRateDistOptQuant  (JVET_K0072 template contexts, flat scaling, HM_QTBT_AS_IN_JEM_QUANT)  CommonLib/QuantRDOQ.cpp:694-1409
//   QuantRDOQ::xGetCodedLevel :107-162, xGetICRate :235-313, xGetRateLast :407-421, xGetErrScaleCoeff :482-506
//   CoeffCodingContext::sigCtxIdAbs / ctxOffsetAbs / GoRiceParAbs   CommonLib/ContextModelling.h:135-219, initSubblock ContextModelling.cpp:353-370
// Pinned against the compiled reference's own QuantRDOQ::quant by tests/golden/rdoq.npz (tests/test_oracle_golden.py).
//
// All costs are IEEE doubles evaluated in the reference's order (the reference is built without FMA contraction: -msse4.1).
// The CABAC side enters as the fractional-bit tables of include/vvcgpu.h (vvcgpu_rdoq_rates).
#include "orc_common.h"
#include "../../include/vvcgpu.h"
#include <vector>
#include <cmath>
#include <limits>

extern "C" int orc_scan_order(int w, int h, uint32_t* out);

namespace {

const int kQuantScales[6] = { 26214, 23302, 20560, 18396, 16384, 14564 };       // g_quantScales, Rom.cpp:465-468
const int kInvQuantScalesR[6] = { 40, 45, 51, 57, 64, 72 };
const int kGoRicePars[32] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2 };   // Rom.cpp:644-650
const int kGoRiceRange[3] = { 6, 5, 6 };                                         // g_auiGoRiceRange[0..2], Rom.cpp:652-655
const int kGroupIdx[64] = { 0, 1, 2, 3, 4, 4, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 8, 8, 8, 8, 8, 8, 8, 8, 9, 9, 9, 9, 9, 9, 9, 9,
                            10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 11, 11, 11, 11, 11, 11, 11, 11, 11, 11, 11, 11, 11, 11, 11, 11 };
int ilog2r(int v) { int l = 0; while ((1 << l) < v) l++; return l; }

struct Bits2 { int b[2]; };

int icRate(unsigned absLevel, const int* par, const int* gt1, const int* gt2, int rice)      // xGetICRate :235-313
{
  int rate = 32768;
  if (absLevel >= 5)
  {
    unsigned symbol = (absLevel - 5) >> 1, length;
    const int threshold = kGoRiceRange[rice];
    if (symbol < (unsigned)(threshold << rice)) { length = symbol >> rice; rate += (length + 1 + rice) << 15; }
    else
    {
      length = rice; symbol -= threshold << rice;
      while (symbol >= (1u << length)) symbol -= 1u << (length++);
      rate += (threshold + length + 1 - rice + length) << 15;
    }
    rate += par[(absLevel - 1) & 1] + gt1[1] + gt2[1];
  }
  else if (absLevel == 1) rate += par[0] + gt1[0];
  else if (absLevel == 2) rate += par[1] + gt1[0];
  else if (absLevel == 3) rate += par[0] + gt1[1] + gt2[0];
  else if (absLevel == 4) rate += par[1] + gt1[1] + gt2[0];
  else rate = 0;
  return rate;
}

}  // namespace

ORC_API uint32_t orc_rdoq(const TCoeff* src, TCoeff* dst, int w, int h, int luma, int bd, int qp, double lambda, int signHiding,
                          const vvcgpu_rdoq_rates* rt)
{
  const int n = w * h, lw = ilog2r(w), lh = ilog2r(h);
  const int per = qp / 6, rem = qp % 6;
  const int maxLog2 = 15;
  const int transformShift = maxLog2 - bd - ((lw + lh) >> 1);
  const bool sqrt2 = ((lw + lh) & 1) != 0;
  const int qBits = 14 + per + transformShift;
  const int quantCoef = sqrt2 ? (kQuantScales[rem] * 181) >> 7 : kQuantScales[rem];
  // xGetErrScaleCoeff :482-506 (DISTORTION_PRECISION_ADJUSTMENT == 0): 2^15 * 2^(-2 * (shift - 0.5 * sqrt2)) / QStep / QStep
  const double errScale = std::ldexp(1.0, 15 - 2 * transformShift + (sqrt2 ? 1 : 0)) / quantCoef / quantCoef / 1;
  const TCoeff entMax = (1 << maxLog2) - 1, entMin = -(1 << maxLog2);
  std::vector<uint32_t> scan(n);
  orc_scan_order(w, h, scan.data());
  const int wig = w >> 2, hig = h >> 2, numCG = n >> 4;
  std::vector<int> scanCG(numCG);
  for (int g = 0; g < numCG; g++) { const int p = (int)scan[g << 4]; scanCG[g] = ((p / w) >> 2) * wig + ((p % w) >> 2); }
  std::vector<double> costCoeff(n, 0.0), costSig(n, 0.0), costCoeff0(n, 0.0), costCGSig(numCG, 0.0);
  std::vector<int> rateIncUp(n, 0), rateIncDown(n, 0), sigRateDelta(n, 0), deltaU(n, 0);
  std::vector<char> sigGroup(numCG, 0);                                            // indexed by CG raster position

  auto tmpl = [&](int pos, int& sumAbs, int& numPos, int& sumGo)                   // sigCtxIdAbs / GoRiceParAbs neighbourhood
  {
    const int y = pos / w, x = pos % w;
    sumAbs = numPos = sumGo = 0;
    auto upd = [&](int v) { const int a = std::abs(v); sumAbs += std::min(4 - (a & 1), a); numPos += a != 0; sumGo += a - (a != 0); };
    if (x < w - 1) { upd(dst[pos + 1]); if (x < w - 2) upd(dst[pos + 2]); if (y < h - 1) upd(dst[pos + w + 1]); }
    if (y < h - 1) { upd(dst[pos + w]); if (y < h - 2) upd(dst[pos + 2 * w]); }
  };

  double blockUncoded = 0, baseCost = 0;
  int cgLastScanPos = -1, lastScanPos = -1;
  for (int subSet = numCG - 1; subSet >= 0; subSet--)
  {
    const int cgPos = scanCG[subSet], cgY = cgPos / wig, cgX = cgPos % wig;
    const int sigRight = cgX + 1 < wig ? sigGroup[cgPos + 1] : 0, sigLower = cgY + 1 < hig ? sigGroup[cgPos + wig] : 0;
    const int* sgBits = rt->sig_group[sigRight | sigLower];
    double sigCost = 0, sigCost0 = 0, codedLevelAndDist = 0, uncodedDist = 0; int nnzBeforePos0 = 0;
    for (int k = 15; k >= 0; k--)
    {
      const int sp = (subSet << 4) + k, pos = (int)scan[sp];
      const int64_t tmpLevel = (int64_t)std::abs(src[pos]) * quantCoef;
      const int levelDouble = (int)std::min<int64_t>(tmpLevel, (int64_t)std::numeric_limits<int>::max() - (1 << (qBits - 1)));
      unsigned maxAbs = std::min<unsigned>((unsigned)entMax, (unsigned)((levelDouble + (1 << (qBits - 1))) >> qBits));
      const double err0 = (double)levelDouble;
      costCoeff0[sp] = err0 * err0 * errScale;
      blockUncoded += costCoeff0[sp];
      dst[pos] = (TCoeff)maxAbs;
      if (maxAbs > 0 && lastScanPos < 0) { lastScanPos = sp; cgLastScanPos = subSet; }
      if (lastScanPos >= 0)
      {
        const bool isLast = sp == lastScanPos;
        int ctxSig = 0, ofs = 0, sumAbs, numPos, sumGo;
        tmpl(pos, sumAbs, numPos, sumGo);
        if (!isLast)
        {
          const int diag = pos / w + pos % w;
          ctxSig = std::min(sumAbs, 5) + (diag < 2 ? 6 : 0) + ((luma && diag < 5) ? 6 : 0);
          ofs = std::min(sumAbs - numPos, 4) + 1 + (diag == 0 ? (luma ? 15 : 5) : (luma ? (diag < 3 ? 10 : (diag < 10 ? 5 : 0)) : 0));
        }
        // at the last position sigCtxIdAbs has never been called: the template state is still -1 and the offset 0 (ContextModelling.h:175-184)
        const int rice = kGoRicePars[std::min(sumGo, 31)];
        const int* par = rt->par[ofs]; const int* gt1 = rt->gt1[ofs]; const int* gt2 = rt->gt2[ofs];
        const int* sig = rt->sig[ctxSig];
        // xGetCodedLevel :107-162
        double codedCost, codedCostSig = costSig[sp]; unsigned best = 0;
        bool done = false;
        if (!isLast && maxAbs < 3)
        {
          codedCostSig = lambda * sig[0];
          codedCost = costCoeff0[sp] + codedCostSig;
          if (maxAbs == 0) done = true;
        }
        else codedCost = std::numeric_limits<double>::max();
        if (!done)
        {
          const double currSig = isLast ? 0.0 : lambda * sig[1];
          const unsigned minAbs = maxAbs > 1 ? maxAbs - 1 : 1;
          for (int a = (int)maxAbs; a >= (int)minAbs; a--)
          {
            const double err = (double)(levelDouble - (int)((unsigned)a << qBits));
            double cost = err * err * errScale + lambda * icRate((unsigned)a, par, gt1, gt2, rice);
            cost += currSig;
            if (cost < codedCost) { best = (unsigned)a; codedCost = cost; codedCostSig = currSig; }
          }
        }
        costCoeff[sp] = codedCost; costSig[sp] = codedCostSig;
        if (!isLast) sigRateDelta[pos] = sig[1] - sig[0];
        deltaU[pos] = (TCoeff)((levelDouble - (int)(best << qBits)) >> (qBits - 8));
        if (best > 0)
        {
          const int now = icRate(best, par, gt1, gt2, rice);
          rateIncUp[pos] = icRate(best + 1, par, gt1, gt2, rice) - now;
          rateIncDown[pos] = icRate(best - 1, par, gt1, gt2, rice) - now;
        }
        else rateIncUp[pos] = par[0] + gt1[0];
        dst[pos] = (TCoeff)best;
        baseCost += costCoeff[sp];
      }
      else baseCost += costCoeff0[sp];
      sigCost += costSig[sp];
      if (k == 0) sigCost0 = costSig[sp];
      if (dst[pos])
      {
        sigGroup[cgPos] = 1;
        codedLevelAndDist += costCoeff[sp] - costSig[sp];
        uncodedDist += costCoeff0[sp];
        if (k != 0) nnzBeforePos0++;
      }
    }
    if (cgLastScanPos >= 0)
    {
      if (subSet)
      {
        if (!sigGroup[cgPos])
        {
          baseCost += lambda * sgBits[0] - sigCost;
          costCGSig[subSet] = lambda * sgBits[0];
        }
        else if (subSet < cgLastScanPos)
        {
          if (nnzBeforePos0 == 0) { baseCost -= sigCost0; sigCost -= sigCost0; }
          double costZeroCG = baseCost;
          baseCost += lambda * sgBits[1];
          costZeroCG += lambda * sgBits[0];
          costCGSig[subSet] = lambda * sgBits[1];
          costZeroCG += uncodedDist;
          costZeroCG -= codedLevelAndDist;
          costZeroCG -= sigCost;
          if (costZeroCG < baseCost)
          {
            sigGroup[cgPos] = 0;
            baseCost = costZeroCG;
            costCGSig[subSet] = lambda * sgBits[0];
            for (int k = 15; k >= 0; k--)
            {
              const int sp = (subSet << 4) + k, pos = (int)scan[sp];
              if (dst[pos]) { dst[pos] = 0; costCoeff[sp] = costCoeff0[sp]; costSig[sp] = 0; }
            }
          }
        }
      }
      else sigGroup[cgPos] = 1;
    }
  }
  if (lastScanPos < 0) return 0;

  // ---- last position :1127-1262
  double bestCost = blockUncoded + lambda * rt->cbf[0];
  baseCost += lambda * rt->cbf[1];
  int bestLastIdxP1 = 0;
  bool foundLast = false;
  for (int cg = cgLastScanPos; cg >= 0 && !foundLast; cg--)
  {
    baseCost -= costCGSig[cg];
    if (!sigGroup[scanCG[cg]]) continue;
    for (int k = 15; k >= 0; k--)
    {
      const int sp = (cg << 4) + k;
      if (sp > lastScanPos) continue;
      const int pos = (int)scan[sp];
      if (dst[pos])
      {
        const int py = pos >> lw, px = pos - (py << lw);
        const int cx = kGroupIdx[px], cy = kGroupIdx[py];
        double c = rt->last_x[cx] + rt->last_y[cy];                                // xGetRateLast :407-421
        if (cx > 3) c += 32768.0 * ((cx - 2) >> 1);
        if (cy > 3) c += 32768.0 * ((cy - 2) >> 1);
        const double costLast = lambda * c;
        const double total = baseCost + costLast - costSig[sp];
        if (total < bestCost) { bestLastIdxP1 = sp + 1; bestCost = total; }
        if (dst[pos] > 1) { foundLast = true; break; }
        baseCost -= costCoeff[sp];
        baseCost += costCoeff0[sp];
      }
      else baseCost -= costSig[sp];
    }
  }
  uint32_t absSum = 0;
  for (int sp = 0; sp < bestLastIdxP1; sp++)
  {
    const int pos = (int)scan[sp];
    const TCoeff level = dst[pos];
    absSum += (uint32_t)level;
    dst[pos] = src[pos] < 0 ? -level : level;
  }
  for (int sp = bestLastIdxP1; sp <= lastScanPos; sp++) dst[scan[sp]] = 0;

  // ---- sign bit hiding :1264-1406
  if (signHiding && (int32_t)absSum >= 2)
  {
    const double inv = (double)kInvQuantScalesR[rem];
    const int64_t rdFactor = (int64_t)(inv * inv * (1 << (2 * per)) / lambda / 16 / (1 << 0) + 0.5);
    int lastCG = -1;
    for (int subSet = (n - 1) >> 4; subSet >= 0; subSet--)
    {
      const int subPos = subSet << 4;
      int firstNZ = 16, lastNZ = -1, sum = 0, k;
      for (k = 15; k >= 0; --k) if (dst[scan[k + subPos]]) { lastNZ = k; break; }
      for (k = 0; k <= 15; k++) if (dst[scan[k + subPos]]) { firstNZ = k; break; }
      for (k = firstNZ; k <= lastNZ; k++) sum += (int)dst[scan[k + subPos]];
      if (lastNZ >= 0 && lastCG == -1) lastCG = 1;
      if (lastNZ - firstNZ >= 4)
      {
        const unsigned signbit = dst[scan[subPos + firstNZ]] > 0 ? 0 : 1;
        if (signbit != (unsigned)(sum & 1))
        {
          int64_t minCostInc = std::numeric_limits<int64_t>::max(), curCost = std::numeric_limits<int64_t>::max();
          int minPos = -1, finalChange = 0, curChange = 0;
          for (k = (lastCG == 1 ? lastNZ : 15); k >= 0; --k)
          {
            const int pos = (int)scan[k + subPos];
            if (dst[pos] != 0)
            {
              const int64_t costUp = rdFactor * (-deltaU[pos]) + rateIncUp[pos];
              int64_t costDown = rdFactor * (deltaU[pos]) + rateIncDown[pos] - ((std::abs(dst[pos]) == 1) ? sigRateDelta[pos] : 0);
              if (lastCG == 1 && lastNZ == k && std::abs(dst[pos]) == 1) costDown -= (4 << 15);
              if (costUp < costDown) { curCost = costUp; curChange = 1; }
              else
              {
                curChange = -1;
                curCost = (k == firstNZ && std::abs(dst[pos]) == 1) ? std::numeric_limits<int64_t>::max() : costDown;
              }
            }
            else
            {
              curCost = rdFactor * (-(std::abs(deltaU[pos]))) + (1 << 15) + rateIncUp[pos] + sigRateDelta[pos];
              curChange = 1;
              if (k < firstNZ)
              {
                const unsigned thisSign = src[pos] >= 0 ? 0 : 1;
                if (thisSign != signbit) curCost = std::numeric_limits<int64_t>::max();
              }
            }
            if (curCost < minCostInc) { minCostInc = curCost; finalChange = curChange; minPos = pos; }
          }
          if (dst[minPos] == entMax || dst[minPos] == entMin) finalChange = -1;
          if (src[minPos] >= 0) dst[minPos] += finalChange; else dst[minPos] -= finalChange;
        }
      }
      if (lastCG == 1) lastCG = 0;
    }
  }
  return absSum;
}
