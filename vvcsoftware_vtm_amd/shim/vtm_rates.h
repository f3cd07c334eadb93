// vtm_rates.h -- reference-side glue of the drop-in binding (compiled only where the reference headers exist).
// The two quantiser entry points take their rate tables as plain arrays (vvcgpu_dq_rates / vvcgpu_rdoq_rates); inside the reference the rates
// live in a CABAC context object.  These two helpers gather them, following what DQIntern::RateEstimator::initCtx (DepQuant.cpp:371-485) and
// QuantRDOQ::xRateDistOptQuant (QuantRDOQ.cpp:694-1409) read.  The shim (vtm_hip_shim.cpp) calls them for every served TU; the oracle's wrapper
// of the compiled reference (oracle/ref_wrap_kernels.h) includes this header too, so that fixtures and drop-in runs build their tables alike.
#pragma once
#include "CommonLib/CommonDef.h"
#include "CommonLib/Unit.h"
#include "CommonLib/UnitTools.h"
#include "CommonLib/Contexts.h"
#include "CommonLib/ContextModelling.h"
#include "CommonLib/Rom.h"
#include "../../include/vvcgpu.h"

// ---------------------------------------------------------------------------------------------
// rate tables of the dependent-quantisation trellis from a CABAC context object: DQIntern::RateEstimator::initCtx (DepQuant.cpp:371-485)
// re-expressed with Ctx's public FracBitsAccess (the estimator class itself is private to DepQuant.cpp).  Shared with the drop-in shim.
inline void vtmref_dq_rates_from_ctx(const TransformUnit& tu, ComponentID compID, const Ctx& ctxRef, vvcgpu_dq_rates* rt)
{
  {
    const Ctx* ctx = &ctxRef;
    const ChannelType chType = toChannelType(compID);
    const int w = tu.blocks[compID].width, h = tu.blocks[compID].height;
    const FracBitsAccess& fb = ctx->getFracBitsAcess();
    memset(rt, 0, sizeof *rt);
    // :379-426
    int32_t cbfDelta;
    if (compID == COMPONENT_Y && !CU::isIntra(*tu.cu) && !tu.depth) { const BinFracBits b = fb.getFracBitsArray(Ctx::QtRootCbf()); cbfDelta = int32_t(b.intBits[1]) - int32_t(b.intBits[0]); }
    else { const BinFracBits b = fb.getFracBitsArray(Ctx::QtCbf[compID](DeriveCtx::CtxQtCbf(compID, tu.depth, tu.cbf[COMPONENT_Cb]))); cbfDelta = int32_t(b.intBits[1]) - int32_t(b.intBits[0]); }
    static const unsigned prefixCtx[] = { 0, 0, 0, 3, 6, 10, 15, 21 };
    for (unsigned xy = 0; xy < 2; xy++)
    {
      const int32_t bitOffset = xy ? cbfDelta : 0;
      int32_t* lastBits = xy ? rt->last_y : rt->last_x;
      const unsigned size = xy ? h : w, log2Size = g_aucNextLog2[size];
      const CtxSet& set = (xy ? Ctx::LastY : Ctx::LastX)[chType];
      const unsigned lastShift = compID == COMPONENT_Y ? (log2Size + 1) >> 2 : Clip3<unsigned>(0, 2, size >> 3);
      const unsigned lastOffset = compID == COMPONENT_Y ? prefixCtx[log2Size] : 0;
      uint32_t ctxBits[LAST_SIGNIFICANT_GROUPS], sum = 0;
      const unsigned maxCtxId = g_uiGroupIdx[size - 1];
      for (unsigned id = 0; id < maxCtxId; id++)
      {
        const BinFracBits b = fb.getFracBitsArray(set(lastOffset + (id >> lastShift)));
        ctxBits[id] = sum + b.intBits[0] + (id > 3 ? ((id - 2) >> 1) << SCALE_BITS : 0) + bitOffset;
        sum += b.intBits[1];
      }
      ctxBits[maxCtxId] = sum + (maxCtxId > 3 ? ((maxCtxId - 2) >> 1) << SCALE_BITS : 0) + bitOffset;
      for (unsigned pos = 0; pos < size; pos++) lastBits[pos] = ctxBits[g_uiGroupIdx[pos]];
    }
    for (unsigned c = 0; c < 2; c++) { const BinFracBits b = fb.getFracBitsArray(Ctx::SigCoeffGroup[chType](c)); rt->sig_sbb[c][0] = b.intBits[0]; rt->sig_sbb[c][1] = b.intBits[1]; }
    const unsigned numSig = compID == COMPONENT_Y ? 18 : 12, numGtx = compID == COMPONENT_Y ? 21 : 11;
    for (unsigned s = 0; s < 3; s++)
      for (unsigned c = 0; c < numSig; c++) { const BinFracBits b = fb.getFracBitsArray(Ctx::SigFlag[chType + 2 * s](c)); rt->sig[s][c][0] = b.intBits[0]; rt->sig[s][c][1] = b.intBits[1]; }
    for (unsigned c = 0; c < numGtx; c++)
    {
      const BinFracBits par = fb.getFracBitsArray(Ctx::ParFlag[chType](c)), gt1 = fb.getFracBitsArray(Ctx::GtxFlag[2 + chType](c)),
                        gt2 = fb.getFracBitsArray(Ctx::GtxFlag[chType](c));
      const int32_t par0 = (1 << SCALE_BITS) + int32_t(par.intBits[0]), par1 = (1 << SCALE_BITS) + int32_t(par.intBits[1]);
      int32_t* o = rt->gtx[c];
      o[0] = 0; o[1] = par0 + gt1.intBits[0]; o[2] = par1 + gt1.intBits[0];
      o[3] = par0 + gt1.intBits[1] + gt2.intBits[0]; o[4] = par1 + gt1.intBits[1] + gt2.intBits[0];
      o[5] = par0 + gt1.intBits[1] + gt2.intBits[1]; o[6] = par1 + gt1.intBits[1] + gt2.intBits[1];
    }
  }
}

// fractional-bit tables QuantRDOQ::xRateDistOptQuant reads (QuantRDOQ.cpp:694-1409), gathered from a CABAC context object.  Shared with the drop-in shim.
inline void vtmref_rdoq_rates_from_ctx(const TransformUnit& tu, ComponentID compID, const Ctx& ctxRef, vvcgpu_rdoq_rates* rt)
{
  const ChannelType chType = toChannelType(compID);
  const FracBitsAccess& fb = ctxRef.getFracBitsAcess();
  memset(rt, 0, sizeof *rt);
  auto put = [&](int32_t* o, unsigned ctxId) { const BinFracBits b = fb.getFracBitsArray(ctxId); o[0] = b.intBits[0]; o[1] = b.intBits[1]; };
  const unsigned numSig = chType == CHANNEL_TYPE_LUMA ? 18 : 12, numGtx = chType == CHANNEL_TYPE_LUMA ? 21 : 11;
  for (unsigned c = 0; c < numSig; c++) put(rt->sig[c], Ctx::SigFlag[chType](c));
  for (unsigned c = 0; c < numGtx; c++) { put(rt->par[c], Ctx::ParFlag[chType](c)); put(rt->gt1[c], Ctx::GtxFlag[2 + chType](c)); put(rt->gt2[c], Ctx::GtxFlag[chType](c)); }
  for (unsigned c = 0; c < 2; c++) put(rt->sig_group[c], Ctx::SigCoeffGroup[chType](c));
  if (compID == COMPONENT_Y && !CU::isIntra(*tu.cu) && tu.depth == 0) put(rt->cbf, Ctx::QtRootCbf());
  else put(rt->cbf, Ctx::QtCbf[compID](DeriveCtx::CtxQtCbf(compID, tu.depth, tu.cbf[COMPONENT_Cb])));
  CoeffCodingContext cctx(tu, compID, false);
  const int dim[2] = { (int)tu.blocks[compID].width, (int)tu.blocks[compID].height };
  for (int xy = 0; xy < 2; xy++)                                                   // :1172-1200
  {
    int32_t* o = xy ? rt->last_y : rt->last_x;
    int bits = 0, id;
    for (id = 0; id < (int)g_uiGroupIdx[dim[xy] - 1]; id++)
    {
      const BinFracBits b = fb.getFracBitsArray(xy ? cctx.lastYCtxId(id) : cctx.lastXCtxId(id));
      o[id] = bits + b.intBits[0]; bits += b.intBits[1];
    }
    o[id] = bits;
  }
}
