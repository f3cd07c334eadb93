// oracle/restate/depquant.cpp -- TEST INFRASTRUCTURE: scalar restatement of the dependent-quantisation trellis (next row N1).
//   DQIntern::DepQuant::quant / xDecideAndUpdate / xDecide        CommonLib/DepQuant.cpp:1222-1391
//   DQIntern::State (rate checks, updateState, updateStateEOS)     :861-1102
//   DQIntern::CommonCtx::update                                    :1104-1164
//   DQIntern::Quantizer::initQuantBlock / preQuantCoeff            :647-706, :786-808
//   DQIntern::ScanData (per-position scan info)                    :511-580
//   DQIntern::Rom::xInitScanArrays (template neighbourhoods)       :89-229
// The rate tables (RateEstimator :335-485: last-position bits, significance / greater-than / parity bits of the current CABAC
// states) are an INPUT, vvcgpu_dq_rates in include/vvcgpu.h.
// Pinned against the compiled reference's own DepQuant::quant (vtmref_depquant) by tests/golden/depquant.npz.
#include "orc_common.h"
#include <vector>
#include <cstdint>

extern "C" int orc_scan_order(int w, int h, uint32_t* out);

namespace {

const int kGoRicePars[32] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2 };
const unsigned kGoRiceRange[10] = { 6, 5, 6, 3, 3, 3, 3, 3, 3, 3 };
const int kQuantScales[6] = { 26214, 23302, 20560, 18396, 16384, 14564 };
const int SCALE_BITS = 15;

struct NbSbb { int num; int inPos[5]; };
struct NbOut { int maxDist, num; int outPos[5]; };
struct PQ { int64_t deltaDist; int absLevel; };
struct Decision { int64_t rdCost; int absLevel; int prevId; };

int ilog2u(unsigned v) { int l = 0; while ((2u << l) <= v) l++; return l; }
int ceilLog2(uint64_t x) { int y = 0; while ((1ull << y) < x && y < 63) y++; return y; }     // :632-645

struct State
{
  int64_t rdCost;
  uint8_t absLevels[16];
  uint16_t ctxInit[16];
  int numSigSbb, refSbbCtxId;
  int32_t sbbBits[2], sigBits[2], coeffBits[7];
  int goRicePar;
};

struct Trellis
{
  // block
  int w, h, numCoeff, sbbSize, sbbMask, log2Sbb, numSbb, widthInSbb, heightInSbb;
  bool luma;
  std::vector<uint32_t> scan, sbbScan;           // scan id -> raster position; sub-block scan id -> sub-block position
  std::vector<int> posX, posY;
  std::vector<NbSbb> nbSbb;
  std::vector<NbOut> nbOut;
  const vvcgpu_dq_rates* rt;
  // quantiser
  int qShift; int64_t qAdd, qScale; int maxQIdx, thresLast;
  int distShift; int64_t distAdd, distStepAdd, distOrgFact;
  // states: [0..3] current, [4..7] previous, [8..11] skip, one start state
  State st[12], start;
  int cur, prv, skp;
  // common context: 8 x { sbbFlags[numSbb], levels[numCoeff] }, first four current, last four previous
  std::vector<uint8_t> mem; int chunk; int curCtx, prvCtx;
  uint8_t* sbbFlags(int set, int id) { return mem.data() + (size_t)(set * 4 + id) * chunk; }
  uint8_t* levels(int set, int id) { return sbbFlags(set, id) + numSbb; }

  const int32_t* sigArr(int stateId) const { return &rt->sig[std::max(stateId - 1, 0)][0][0]; }

  void initBlock(int w_, int h_, bool luma_)
  {
    w = w_; h = h_; luma = luma_; numCoeff = w * h;
    const bool no4x4 = (w & 3) || (h & 3);
    log2Sbb = no4x4 ? 2 : 4; sbbSize = 1 << log2Sbb; sbbMask = sbbSize - 1;
    const int lg = no4x4 ? 1 : 2;
    widthInSbb = w >> lg; heightInSbb = h >> lg; numSbb = widthInSbb * heightInSbb;
    scan.resize(numCoeff); orc_scan_order(w, h, scan.data());
    posX.resize(numCoeff); posY.resize(numCoeff);
    std::vector<int> raster2id(numCoeff);
    for (int i = 0; i < numCoeff; i++) { posX[i] = scan[i] % w; posY[i] = scan[i] / w; raster2id[scan[i]] = i; }
    // sub-block scan (SCAN_UNGROUPED diagonal over the sub-block grid): the order in which the grouped scan visits the sub-blocks
    sbbScan.resize(numSbb);
    for (int s = 0; s < numSbb; s++) sbbScan[s] = (posY[s << log2Sbb] >> lg) * widthInSbb + (posX[s << log2Sbb] >> lg);
    // template neighbourhoods :135-216: right, right+1, below-right, below, below+1 -- inside the sub-block (as in-sub-block
    // positions, ascending) and outside it (as scan ids relative to the sub-block start, ascending)
    nbSbb.assign(numCoeff, NbSbb()); nbOut.assign(numCoeff, NbOut());
    for (int id = 0; id < numCoeff; id++)
    {
      const int x = posX[id], y = posY[id], r = scan[id], beg = id - (id & sbbMask);
      int cand[5] = { x < w - 1 ? raster2id[r + 1] : 0, x < w - 2 ? raster2id[r + 2] : 0, (x < w - 1 && y < h - 1) ? raster2id[r + 1 + w] : 0,
                      y < h - 1 ? raster2id[r + w] : 0, y < h - 2 ? raster2id[r + 2 * w] : 0 };
      const bool have[5] = { x < w - 1, x < w - 2, x < w - 1 && y < h - 1, y < h - 1, y < h - 2 };
      int in[5], out[5];
      for (int k = 0; k < 5; k++)
      {
        const int rel = cand[k] - beg;
        in[k] = (have[k] && rel < sbbSize) ? rel : 0;
        out[k] = (have[k] && rel >= sbbSize) ? cand[k] : 0;
      }
      NbSbb& a = nbSbb[id]; a.num = 0;
      for (;;) { int nk = -1; for (int k = 0; k < 5; k++) if (in[k] != 0 && (nk < 0 || in[k] < in[nk])) nk = k; if (nk < 0) break; a.inPos[a.num++] = in[nk]; in[nk] = 0; }
      for (int k = a.num; k < 5; k++) a.inPos[k] = 0;
      NbOut& b = nbOut[id]; b.num = 0;
      for (;;) { int nk = -1; for (int k = 0; k < 5; k++) if (out[k] != 0 && (nk < 0 || out[k] < out[nk])) nk = k; if (nk < 0) break; b.outPos[b.num++] = out[nk]; out[nk] = 0; }
      for (int k = b.num; k < 5; k++) b.outPos[k] = 0;
      b.maxDist = id == 0 ? 0 : nbOut[id - 1].maxDist;
      for (int k = 0; k < b.num; k++) b.maxDist = std::max(b.maxDist, b.outPos[k]);
    }
    for (int id = 0; id < numCoeff; id++)                       // make it relative :218-228
    {
      const int beg = id - (id & sbbMask);
      for (int k = 0; k < nbOut[id].num; k++) nbOut[id].outPos[k] -= beg;
      nbOut[id].maxDist -= id;
    }
  }

  void initQuant(int bd, int qp, double lambda)                 // :647-706
  {
    const int qpDQ = qp + 1, qpPer = qpDQ / 6, qpRem = qpDQ - 6 * qpPer;
    const int lw = ilog2u(w), lh = ilog2u(h);
    const bool sqrt2 = ((lw + lh) & 1) != 0;
    const int transformShift = 15 - bd - ((lw + lh) >> 1);
    qShift = 14 - 1 + qpPer + transformShift;
    qAdd = -(((int64_t)3 << qShift) >> 1);
    const int invShift = 6 + 1 - qpPer - transformShift + (sqrt2 ? 8 : 0);
    qScale = sqrt2 ? (kQuantScales[qpRem] * 181) >> 7 : kQuantScales[qpRem];
    const unsigned qIdxBD = std::min<unsigned>(15 + 1, 8 * sizeof(int) + invShift - 6 - 1);
    maxQIdx = (1 << (qIdxBD - 1)) - 4;
    thresLast = (int)(((int64_t)3 << qShift) / (4 * qScale));
    const int64_t qs = kQuantScales[qpRem];
    const int nomDShift = SCALE_BITS - 2 * transformShift + qShift;       // DISTORTION_PRECISION_ADJUSTMENT == 0
    const double qScale2 = (double)(qs * qs);
    const double nomDistFactor = nomDShift < 0 ? 1.0 / ((double)((int64_t)1 << (-nomDShift)) * qScale2 * lambda) : (double)((int64_t)1 << nomDShift) / (qScale2 * lambda);
    const int64_t pow2dfShift = (int64_t)(nomDistFactor * qScale2) + 1;
    const int dfShift = ceilLog2((uint64_t)pow2dfShift);
    distShift = 62 + qShift - 2 * 15 - dfShift;
    distAdd = ((int64_t)1 << distShift) >> 1;
    distStepAdd = (int64_t)(nomDistFactor * (double)((int64_t)1 << (distShift + qShift)) + .5);
    distOrgFact = (int64_t)(nomDistFactor * (double)((int64_t)1 << (distShift + 1)) + .5);
  }

  void preQuant(int absCoeff, PQ* pq) const                     // :786-808
  {
    const int64_t scaledOrg = (int64_t)absCoeff * qScale;
    int qIdx = std::max(1, std::min(maxQIdx, (int)((scaledOrg + qAdd) >> qShift)));
    int64_t scaledAdd = qIdx * distStepAdd - scaledOrg * distOrgFact;
    for (int k = 0; k < 4; k++)
    {
      PQ& p = pq[qIdx & 3];
      p.deltaDist = (scaledAdd * qIdx + distAdd) >> distShift;
      p.absLevel = (++qIdx) >> 1;
      scaledAdd += distStepAdd;
    }
  }

  static void initState(State& s, const int32_t* sig0, const int32_t* gtx0)
  {
    s.rdCost = INT64_MAX >> 1; s.numSigSbb = 0; s.refSbbCtxId = -1; s.goRicePar = 0;
    memcpy(s.sigBits, sig0, sizeof s.sigBits); memcpy(s.coeffBits, gtx0, sizeof s.coeffBits);
    s.sbbBits[0] = s.sbbBits[1] = 0;
    memset(s.absLevels, 0, 16); memset(s.ctxInit, 0, 32);
  }

  static int32_t levelBits(const State& s, unsigned level)       // :909-931
  {
    if (level < 5) return s.coeffBits[level];
    const unsigned value = (level - 5) >> 1;
    const int32_t bits = s.coeffBits[level - (value << 1)];
    const unsigned thres = kGoRiceRange[s.goRicePar] << s.goRicePar;
    if (value < thres) return bits + (((value >> s.goRicePar) + 1 + s.goRicePar) << SCALE_BITS);
    unsigned length = s.goRicePar, delta = 1u << length, valLeft = value - thres;
    while (valLeft >= delta) { valLeft -= delta; delta = 1u << (++length); }
    return bits + ((kGoRiceRange[s.goRicePar] + 1 + (length << 1) - s.goRicePar) << SCALE_BITS);
  }

  // spt: 0 inside a sub-block, 1 start of a coded sub-block (socsbb), 2 end of a coded sub-block (eocsbb)
  static void checkNonZero(const State& s, int id, int spt, const PQ& p, Decision& d)
  {
    int64_t c = s.rdCost + p.deltaDist + levelBits(s, p.absLevel);
    if (spt == 0) c += s.sigBits[1];
    else if (spt == 1) c += s.sbbBits[1] + s.sigBits[1];
    else if (s.numSigSbb) c += s.sigBits[1];
    if (c < d.rdCost) { d.rdCost = c; d.absLevel = p.absLevel; d.prevId = id; }
  }
  static void checkZero(const State& s, int id, int spt, Decision& d)
  {
    int64_t c = s.rdCost;
    if (spt == 0) c += s.sigBits[0];
    else if (spt == 1) c += s.sbbBits[1] + s.sigBits[0];
    else if (s.numSigSbb) c += s.sigBits[0];
    else return;
    if (c < d.rdCost) { d.rdCost = c; d.absLevel = 0; d.prevId = id; }
  }

  struct Info { int scanIdx, lastOffset, sigCtxOffsetNext, gtxCtxOffsetNext, insidePos, nextInsidePos; NbSbb nextNb; bool eosbb, socsbb, eocsbb; int sbbPos, nextSbbRight, nextSbbBelow; };

  Info info(int scanIdx) const                                   // ScanData::xSet :553-589
  {
    Info f; memset(&f, 0, sizeof f);
    f.scanIdx = scanIdx;
    f.sbbPos = sbbScan[scanIdx >> log2Sbb];
    f.lastOffset = rt->last_x[posX[scanIdx]] + rt->last_y[posY[scanIdx]];
    f.insidePos = scanIdx & sbbMask;
    const bool sosbb = f.insidePos == sbbMask;
    f.eosbb = f.insidePos == 0;
    f.socsbb = sosbb && scanIdx > sbbSize && scanIdx < numCoeff - 1;
    f.eocsbb = f.eosbb && scanIdx > 0 && scanIdx < numCoeff - sbbSize;
    if (scanIdx)
    {
      const int nxt = scanIdx - 1, diag = posX[nxt] + posY[nxt];
      if (luma) { f.sigCtxOffsetNext = diag < 2 ? 12 : diag < 5 ? 6 : 0; f.gtxCtxOffsetNext = diag < 1 ? 16 : diag < 3 ? 11 : diag < 10 ? 6 : 1; }
      else { f.sigCtxOffsetNext = diag < 2 ? 6 : 0; f.gtxCtxOffsetNext = diag < 1 ? 6 : 1; }
      f.nextInsidePos = nxt & sbbMask;
      f.nextNb = nbSbb[nxt];
      if (f.eosbb)
      {
        const int np = sbbScan[nxt >> log2Sbb], ny = np / widthInSbb, nx = np - ny * widthInSbb;
        f.nextSbbRight = nx < widthInSbb - 1 ? np + 1 : 0;
        f.nextSbbBelow = ny < heightInSbb - 1 ? np + widthInSbb : 0;
      }
    }
    return f;
  }

  void setRates(State& s, int id, const Info& f, int sumAbs, int sumAbs1, int sumNum)
  {
    const int sumGt1 = sumAbs1 - sumNum;
    sumAbs -= sumNum;
    memcpy(s.sigBits, sigArr(id) + 2 * (f.sigCtxOffsetNext + (sumAbs1 < 5 ? sumAbs1 : 5)), sizeof s.sigBits);
    memcpy(s.coeffBits, rt->gtx[f.gtxCtxOffsetNext + (sumGt1 < 4 ? sumGt1 : 4)], sizeof s.coeffBits);
    s.goRicePar = kGoRicePars[sumAbs < 31 ? sumAbs : 31];
  }

  void updateState(State& s, int id, const Info& f, const State* prev, const Decision& d)        // :1004-1068
  {
    s.rdCost = d.rdCost;
    if (d.prevId <= -2) return;
    if (d.prevId >= 0)
    {
      const State& p = prev[d.prevId];
      s.numSigSbb = p.numSigSbb + !!d.absLevel; s.refSbbCtxId = p.refSbbCtxId;
      memcpy(s.sbbBits, p.sbbBits, sizeof s.sbbBits); memcpy(s.absLevels, p.absLevels, 16); memcpy(s.ctxInit, p.ctxInit, 32);
    }
    else { s.numSigSbb = 1; s.refSbbCtxId = -1; memset(s.absLevels, 0, 16); memset(s.ctxInit, 0, 32); }
    s.absLevels[f.insidePos] = (uint8_t)std::min(255, d.absLevel);
    const int tinit = s.ctxInit[f.nextInsidePos];
    int sumAbs = tinit >> 8, sumAbs1 = (tinit >> 3) & 31, sumNum = tinit & 7;
    const int num = std::min(f.nextNb.num, 5);
    for (int k = 0; k < num; k++) { const int t = s.absLevels[f.nextNb.inPos[k]]; sumAbs += t; sumAbs1 += std::min(4 - (t & 1), t); sumNum += !!t; }
    setRates(s, id, f, sumAbs, sumAbs1, sumNum);
  }

  void commonUpdate(const Info& f, const State* prevState, State& s, int id)                     // :1104-1164
  {
    uint8_t* flags = sbbFlags(curCtx, id);
    uint8_t* lev = levels(curCtx, id);
    const int setCp = nbOut[f.scanIdx - 1].maxDist;
    if (prevState && prevState->refSbbCtxId >= 0)
    {
      memcpy(flags, sbbFlags(prvCtx, prevState->refSbbCtxId), numSbb);
      memcpy(lev + f.scanIdx, levels(prvCtx, prevState->refSbbCtxId) + f.scanIdx, setCp);
    }
    else { memset(flags, 0, numSbb); memset(lev + f.scanIdx, 0, setCp); }
    flags[f.sbbPos] = !!s.numSigSbb;
    memcpy(lev + f.scanIdx, s.absLevels, sbbSize);
    const int sigNSbb = ((f.nextSbbRight ? flags[f.nextSbbRight] : 0) || (f.nextSbbBelow ? flags[f.nextSbbBelow] : 0)) ? 1 : 0;
    s.numSigSbb = 0; s.refSbbCtxId = id;
    s.sbbBits[0] = rt->sig_sbb[sigNSbb][0]; s.sbbBits[1] = rt->sig_sbb[sigNSbb][1];
    uint16_t tpl[16];
    const int scanBeg = f.scanIdx - sbbSize;
    const uint8_t* abs = lev + scanBeg;
    for (int i = 0; i < sbbSize; i++)
    {
      const NbOut& nb = nbOut[scanBeg + i];
      if (nb.num)
      {
        int sumAbs = 0, sumAbs1 = 0, sumNum = 0;
        for (int k = 0; k < nb.num; k++) { const int t = abs[nb.outPos[k]]; sumAbs += t; sumAbs1 += std::min(4 - (t & 1), t); sumNum += !!t; }
        tpl[i] = (uint16_t)(sumNum + (sumAbs1 << 3) + (std::min(127, sumAbs) << 8));
      }
      else tpl[i] = 0;
    }
    memset(s.absLevels, 0, 16);
    memset(s.ctxInit, 0, 32);
    memcpy(s.ctxInit, tpl, sbbSize * sizeof(uint16_t));
  }

  void updateStateEOS(State& s, int id, const Info& f, const State* prev, const State* skip, const Decision& d)   // :1071-1102
  {
    s.rdCost = d.rdCost;
    if (d.prevId <= -2) return;
    const State* p = nullptr;
    if (d.prevId >= 0)
    {
      p = d.prevId < 4 ? prev + d.prevId : skip + (d.prevId - 4);
      s.numSigSbb = p->numSigSbb + !!d.absLevel;
      memcpy(s.absLevels, p->absLevels, 16);
    }
    else { s.numSigSbb = 1; memset(s.absLevels, 0, 16); }
    s.absLevels[f.insidePos] = (uint8_t)std::min(255, d.absLevel);
    commonUpdate(f, p, s, id);
    const int tinit = s.ctxInit[f.nextInsidePos];
    const int sumNum = tinit & 7, sumAbs1 = (tinit >> 3) & 31, sumAbs = tinit >> 8;
    setRates(s, id, f, sumAbs, sumAbs1, sumNum);
  }

  uint32_t run(const TCoeff* coef, TCoeff* level, const vvcgpu_dq_rates* rates, int bd, int qp, double lambda)
  {
    rt = rates;
    initQuant(bd, qp, lambda);
    memset(level, 0, sizeof(TCoeff) * numCoeff);
    int first = numCoeff - 1;
    for (; first >= 0; first--) if (std::abs(coef[scan[first]]) > thresLast) break;
    if (first < 0) return 0;
    chunk = numSbb + numCoeff; mem.assign((size_t)8 * chunk, 0); curCtx = 0; prvCtx = 1;
    for (int k = 0; k < 12; k++) initState(st[k], sigArr(k & 3), rt->gtx[0]);
    initState(start, sigArr(0), rt->gtx[0]);
    cur = 0; prv = 4; skp = 8;
    std::vector<Decision> trellis((size_t)numCoeff * 8);
    for (int scanIdx = first; scanIdx >= 0; scanIdx--)
    {
      const Info f = info(scanIdx);
      Decision* dec = &trellis[(size_t)scanIdx * 8];
      std::swap(prv, cur);
      const int spt = f.socsbb ? 1 : (f.eocsbb ? 2 : 0);
      for (int k = 0; k < 4; k++) dec[k] = { INT64_MAX >> 2, -1, -2 };
      for (int k = 4; k < 8; k++) dec[k] = { INT64_MAX >> 2, 0, k };
      PQ pq[4];
      preQuant(std::abs(coef[scan[scanIdx]]), pq);
      const State* P = st + prv;
      checkNonZero(P[0], 0, spt, pq[0], dec[0]); checkNonZero(P[0], 0, spt, pq[2], dec[2]); checkZero(P[0], 0, spt, dec[0]);
      checkNonZero(P[1], 1, spt, pq[2], dec[0]); checkNonZero(P[1], 1, spt, pq[0], dec[2]); checkZero(P[1], 1, spt, dec[2]);
      checkNonZero(P[2], 2, spt, pq[3], dec[1]); checkNonZero(P[2], 2, spt, pq[1], dec[3]); checkZero(P[2], 2, spt, dec[1]);
      checkNonZero(P[3], 3, spt, pq[1], dec[1]); checkNonZero(P[3], 3, spt, pq[3], dec[3]); checkZero(P[3], 3, spt, dec[3]);
      if (spt == 2)
        for (int k = 0; k < 4; k++)
        {
          const State& s = st[skp + k];
          const int64_t c = s.rdCost + s.sbbBits[0];
          if (c < dec[k].rdCost) { dec[k].rdCost = c; dec[k].absLevel = 0; dec[k].prevId = 4 + k; }
        }
      for (int k = 0; k < 4; k += 2)                               // checkRdCostStart on decisions 0 and 2
      {
        const PQ& p = pq[k];
        const int64_t c = p.deltaDist + f.lastOffset + levelBits(start, p.absLevel);
        if (c < dec[k].rdCost) { dec[k].rdCost = c; dec[k].absLevel = p.absLevel; dec[k].prevId = -1; }
      }
      if (scanIdx)
      {
        if (f.eosbb)
        {
          std::swap(curCtx, prvCtx);
          for (int k = 0; k < 4; k++) updateStateEOS(st[cur + k], k, f, st + prv, st + skp, dec[k]);
          memcpy(dec + 4, dec, 4 * sizeof(Decision));
        }
        else
          for (int k = 0; k < 4; k++) updateState(st[cur + k], k, f, st + prv, dec[k]);
        if (f.socsbb) std::swap(prv, skp);
      }
    }
    Decision d = { INT64_MAX, -1, -2 };
    int64_t minCost = 0;
    for (int k = 0; k < 4; k++) if (trellis[k].rdCost < minCost) { d.prevId = k; minCost = trellis[k].rdCost; }
    uint32_t absSum = 0;
    for (int scanIdx = 0; d.prevId >= 0; scanIdx++)
    {
      d = trellis[(size_t)scanIdx * 8 + d.prevId];
      const int pos = scan[scanIdx];
      level[pos] = coef[pos] < 0 ? -d.absLevel : d.absLevel;
      absSum += d.absLevel;
    }
    return absSum;
  }
};

}  // namespace

ORC_API uint32_t orc_depquant(const TCoeff* coef, TCoeff* level, int w, int h, int luma, int bd, int qp, double lambda, const vvcgpu_dq_rates* rates)
{
  Trellis t;
  t.initBlock(w, h, luma != 0);
  return t.run(coef, level, rates, bd, qp, lambda);
}
