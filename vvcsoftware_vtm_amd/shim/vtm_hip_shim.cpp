// vtm_hip_shim.cpp -- reference-side binding of libvvcgpu.so for the picture-level in-loop filters (host C++).
//
// This is the `InitHIP` translation unit INTEGRATION.md describes, in the form that needs NO change to the reference
// sources: it is linked with the reference objects using GNU ld --wrap, so that the calls
//     LoopFilter::loopFilterPic(cs)                       DecoderLib/DecLib.cpp:516, EncoderLib/EncGOP.cpp:2122
//     SampleAdaptiveOffset::SAOProcess(cs, saoBlkParams)  DecoderLib/DecLib.cpp:524
//     AdaptiveLoopFilter::ALFProcess(cs, alfSliceParam)   DecoderLib/DecLib.cpp:530
// land here.  The CodingStructure walk (which edges, which boundary strength, SAO merge resolution, ALF coefficient
// reconstruction) is done by calling the reference's OWN private helpers (the file is compiled with
// -fno-access-control); only the sample arithmetic moves to the GPU.  Compiled only where the reference headers exist
// (oracle/Makefile, target `ref`); the C ABI of vvcgpu.h is the only interface to the device.
// Environment: VVCGPU_SHIM=0 falls through to the reference implementation (A/B runs with one binary).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "CommonLib/CommonDef.h"
#include "CommonLib/CodingStructure.h"
#include "CommonLib/Picture.h"
#include "CommonLib/UnitTools.h"
#include "CommonLib/LoopFilter.h"
#include "CommonLib/SampleAdaptiveOffset.h"
#include "CommonLib/AdaptiveLoopFilter.h"
#include "vvcgpu.h"

#define VVCGPU(call) do { if ((call) != 0) THROW("vvcgpu: " << vvcgpu_last_error()); } while (0)

void real_loopFilterPic(LoopFilter*, CodingStructure&) asm("__real__ZN10LoopFilter13loopFilterPicER15CodingStructure");
void wrap_loopFilterPic(LoopFilter*, CodingStructure&) asm("__wrap__ZN10LoopFilter13loopFilterPicER15CodingStructure");
void real_SAOProcess(SampleAdaptiveOffset*, CodingStructure&, SAOBlkParam*) asm("__real__ZN20SampleAdaptiveOffset10SAOProcessER15CodingStructureP11SAOBlkParam");
void wrap_SAOProcess(SampleAdaptiveOffset*, CodingStructure&, SAOBlkParam*) asm("__wrap__ZN20SampleAdaptiveOffset10SAOProcessER15CodingStructureP11SAOBlkParam");
void real_ALFProcess(AdaptiveLoopFilter*, CodingStructure&, AlfSliceParam&) asm("__real__ZN18AdaptiveLoopFilter10ALFProcessER15CodingStructureR13AlfSliceParam");
void wrap_ALFProcess(AdaptiveLoopFilter*, CodingStructure&, AlfSliceParam&) asm("__wrap__ZN18AdaptiveLoopFilter10ALFProcessER15CodingStructureR13AlfSliceParam");

namespace {

bool shimEnabled()
{
  static int on = -1;
  if (on < 0) { const char* e = getenv("VVCGPU_SHIM"); on = (e && e[0] == '0') ? 0 : 1; }
  return on == 1;
}
long g_calls[3] = { 0, 0, 0 };
struct Report { ~Report() { if (shimEnabled()) fprintf(stderr, "[vvcgpu shim] GPU calls: deblock %ld, SAO %ld, ALF %ld\n", g_calls[0], g_calls[1], g_calls[2]); } } g_report;

// ---- device-resident picture (three planes), re-used across calls
struct DevPlanes
{
  vvc_pel* p[3] = { nullptr, nullptr, nullptr };
  int w[3] = { 0, 0, 0 }, h[3] = { 0, 0, 0 }, stride[3] = { 0, 0, 0 };
  void ensure(const CPelUnitBuf& b)
  {
    for (int c = 0; c < 3; c++)
    {
      const int cw = b.bufs[c].width, ch = b.bufs[c].height;
      if (cw != w[c] || ch != h[c])
      {
        if (p[c]) VVCGPU(vvcgpu_free(p[c]));
        stride[c] = (cw + 63) & ~63;
        VVCGPU(vvcgpu_malloc((void**)&p[c], (size_t)stride[c] * ch * sizeof(vvc_pel)));
        w[c] = cw; h[c] = ch;
      }
    }
  }
  void upload(const CPelUnitBuf& b)
  {
    ensure(b);
    for (int c = 0; c < 3; c++)
      VVCGPU(vvcgpu_memcpy2d_h2d(p[c], stride[c] * sizeof(vvc_pel), b.bufs[c].buf, b.bufs[c].stride * sizeof(Pel), w[c] * sizeof(Pel), h[c], nullptr));
  }
  void download(PelUnitBuf b)
  {
    for (int c = 0; c < 3; c++)
      VVCGPU(vvcgpu_memcpy2d_d2h(b.bufs[c].buf, b.bufs[c].stride * sizeof(Pel), p[c], stride[c] * sizeof(vvc_pel), w[c] * sizeof(Pel), h[c], nullptr));
    VVCGPU(vvcgpu_stream_sync(nullptr));
  }
};
DevPlanes g_a, g_b;

template <typename T> struct DevArray
{
  T* ptr = nullptr; size_t cap = 0;
  void upload(const T* host, size_t n)
  {
    if (n > cap) { if (ptr) VVCGPU(vvcgpu_free(ptr)); VVCGPU(vvcgpu_malloc((void**)&ptr, n * sizeof(T))); cap = n; }
    VVCGPU(vvcgpu_memcpy_h2d(ptr, host, n * sizeof(T), nullptr));
  }
  void reserve(size_t n) { if (n > cap) { if (ptr) VVCGPU(vvcgpu_free(ptr)); VVCGPU(vvcgpu_malloc((void**)&ptr, n * sizeof(T))); cap = n; } }
};
DevArray<uint8_t> g_edgeV, g_edgeH, g_flags[3];
DevArray<int8_t> g_qpY, g_qpC;
DevArray<vvcgpu_sao_ctu> g_sao;
DevArray<uint16_t> g_cls;

inline uint32_t rasterIdx(const Position& pos, const PreCalcValues& pcv)
{
  return ((pos.x & pcv.maxCUWidthMask) >> pcv.minCUWidthLog2) + ((pos.y & pcv.maxCUHeightMask) >> pcv.minCUHeightLog2) * pcv.partsInCtuWidth;
}

// The edge / boundary-strength derivation of LoopFilter::xDeblockCU (LoopFilter.cpp:243-369), with the two sample-filter
// calls (:340-347) replaced by recording what they would have filtered.  Every decision is made by the reference's own
// xSetLoopfilterParam / xSetEdgefilterMultiple / xGetBoundaryStrengthSingle on its own scratch arrays.
void recordCU(LoopFilter& lf, CodingUnit& cu, const DeblockEdgeDir edgeDir, std::vector<uint8_t>& emap, int w4)
{
  const PreCalcValues& pcv = *cu.cs->pcv;
  const Area area = cu.Y().valid() ? cu.Y() : Area(recalcPosition(cu.chromaFormat, cu.chType, CHANNEL_TYPE_LUMA, cu.blocks[cu.chType].pos()),
                                                   recalcSize(cu.chromaFormat, cu.chType, CHANNEL_TYPE_LUMA, cu.blocks[cu.chType].size()));
  lf.xSetLoopfilterParam(cu);
  for (auto& currTU : CU::traverseTUs(cu))
  {
    const Area& areaTu = cu.Y().valid() ? currTU.block(COMPONENT_Y) : area;
    lf.xSetEdgefilterMultiple(cu, EDGE_VER, areaTu, lf.m_stLFCUParam.internalEdge);
    lf.xSetEdgefilterMultiple(cu, EDGE_HOR, areaTu, lf.m_stLFCUParam.internalEdge);
  }
  for (auto& currPU : CU::traversePUs(cu))
  {
    const Area& areaPu = cu.Y().valid() ? currPU.block(COMPONENT_Y) : area;
    const bool xOff = currPU.blocks[cu.chType].x != cu.blocks[cu.chType].x;
    const bool yOff = currPU.blocks[cu.chType].y != cu.blocks[cu.chType].y;
    lf.xSetEdgefilterMultiple(cu, EDGE_VER, areaPu, (xOff ? lf.m_stLFCUParam.internalEdge : lf.m_stLFCUParam.leftEdge), xOff);
    lf.xSetEdgefilterMultiple(cu, EDGE_HOR, areaPu, (yOff ? lf.m_stLFCUParam.internalEdge : lf.m_stLFCUParam.topEdge), yOff);
  }
  if (cu.affine)
  {
    const int widthInBaseUnits = cu.Y().width >> pcv.minCUWidthLog2;
    for (uint32_t edgeIdx = 1; edgeIdx < (uint32_t)widthInBaseUnits; edgeIdx++)
      lf.xSetEdgefilterMultiple(cu, EDGE_VER, Area(cu.Y().x + edgeIdx * pcv.minCUWidth, cu.Y().y, pcv.minCUWidth, cu.Y().height), lf.m_stLFCUParam.internalEdge, 1);
    const int heightInBaseUnits = cu.Y().height >> pcv.minCUHeightLog2;
    for (uint32_t edgeIdx = 1; edgeIdx < (uint32_t)heightInBaseUnits; edgeIdx++)
      lf.xSetEdgefilterMultiple(cu, EDGE_HOR, Area(cu.Y().x, cu.Y().y + edgeIdx * pcv.minCUHeight, cu.Y().width, pcv.minCUHeight), lf.m_stLFCUParam.internalEdge, 1);
  }
  const unsigned uiPelsInPart = pcv.minCUWidth;
  for (int y = 0; y < (int)area.height; y += uiPelsInPart)
    for (int x = 0; x < (int)area.width; x += uiPelsInPart)
    {
      unsigned uiBSCheck = 1;
      if (!pcv.noRQT && uiPelsInPart == 4)
        uiBSCheck = ((edgeDir == EDGE_VER) && (x % 8 == 0)) || ((edgeDir == EDGE_HOR) && (y % 8 == 0));
      const Position localPos{ area.x + x, area.y + y };
      const unsigned idx = rasterIdx(localPos, pcv);
      if (lf.m_aapbEdgeFilter[edgeDir][idx] && uiBSCheck)
        lf.m_aapucBS[edgeDir][idx] = lf.xGetBoundaryStrengthSingle(cu, edgeDir, localPos);
    }
  // 8x8 deblocking grid (:313-324)
  if (edgeDir == EDGE_HOR) { if ((cu.block(COMPONENT_Y).y % 8) != 0) return; }
  else                     { if ((cu.block(COMPONENT_Y).x % 8) != 0) return; }

  const unsigned shiftFactor = edgeDir == EDGE_VER ? ::getComponentScaleX(COMPONENT_Cb, pcv.chrFormat) : ::getComponentScaleY(COMPONENT_Cb, pcv.chrFormat);
  const bool bAlwaysDoChroma = pcv.chrFormat == CHROMA_444 || pcv.noRQT;
  unsigned orthogonalLength = 1, orthogonalIncrement = 1;
  if (cu.blocks[COMPONENT_Y].valid())
  {
    if ((cu.blocks[COMPONENT_Y].height > 64) && (edgeDir == EDGE_HOR)) { orthogonalIncrement = 64 / 4; orthogonalLength = cu.blocks[COMPONENT_Y].height / 4; }
    if ((cu.blocks[COMPONENT_Y].width > 64) && (edgeDir == EDGE_VER))  { orthogonalIncrement = 64 / 4; orthogonalLength = cu.blocks[COMPONENT_Y].width / 4; }
  }
  const SPS& sps = *cu.cs->sps;
  const PPS& pps = *cu.cs->pps;
  const bool bPCMFilter = sps.getUsePCM() && sps.getPCMFilterDisableFlag();
  const Position lumaPos = area.pos();
  const Size lumaSize = area.size();
  const unsigned numParts = (edgeDir == EDGE_VER) ? lumaSize.height / pcv.minCUHeight : lumaSize.width / pcv.minCUWidth;
  for (unsigned edge = 0; edge < orthogonalLength; edge += orthogonalIncrement)
  {
    const bool doLuma = cu.blocks[COMPONENT_Y].valid();
    const bool doChroma = cu.blocks[COMPONENT_Cb].valid() && pcv.chrFormat != CHROMA_400 &&
                          (bAlwaysDoChroma || (uiPelsInPart > DEBLOCK_SMALLEST_BLOCK) || (edge % ((DEBLOCK_SMALLEST_BLOCK << shiftFactor) / uiPelsInPart)) == 0);
    for (unsigned iIdx = 0; iIdx < numParts; iIdx++)
    {
      const Position pos = (edgeDir == EDGE_VER) ? Position{ lumaPos.x + (int)edge * 4, lumaPos.y + (int)iIdx * 4 }
                                                 : Position{ lumaPos.x + (int)iIdx * 4, lumaPos.y + (int)edge * 4 };
      const unsigned bs = lf.m_aapucBS[edgeDir][rasterIdx(pos, pcv)];
      if (!bs) continue;
      uint8_t e = 0;
      if (doLuma) e |= (uint8_t)(bs & 3);
      if (doChroma && bs > 1) e |= (uint8_t)((bs & 3) << 2);
      if (bPCMFilter || pps.getTransquantBypassEnabledFlag())
      {
        const Position posP = (edgeDir == EDGE_VER) ? pos.offset(-1, 0) : pos.offset(0, -1);
        const CodingUnit& cuP = *cu.cs->getCU(cu.Y().valid() ? posP : recalcPosition(cu.chromaFormat, CHANNEL_TYPE_LUMA, cu.chType, posP), cu.chType);
        bool noP = bPCMFilter && cuP.ipcm, noQ = bPCMFilter && cu.ipcm;
        if (pps.getTransquantBypassEnabledFlag()) { noP = noP || cuP.transQuantBypass; noQ = noQ || cu.transQuantBypass; }
        e |= (noP ? 0x10 : 0) | (noQ ? 0x20 : 0);
      }
      // in the dual tree the luma pass and the chroma pass record into different bit fields of the same unit
      emap[(pos.y >> 2) * w4 + (pos.x >> 2)] |= e;
    }
  }
}

void buildDeblockMaps(LoopFilter& lf, CodingStructure& cs, std::vector<uint8_t>& ev, std::vector<uint8_t>& eh,
                      std::vector<int8_t>& qy, std::vector<int8_t>& qc)
{
  const PreCalcValues& pcv = *cs.pcv;
  const int w4 = pcv.lumaWidth >> 2, h4 = pcv.lumaHeight >> 2;
  ev.assign((size_t)w4 * h4, 0); eh.assign((size_t)w4 * h4, 0);
  qy.assign((size_t)w4 * h4, 0); qc.assign((size_t)w4 * h4, 0);
  const bool dual = CS::isDualITree(cs);
  for (int dir = 0; dir < 2; dir++)
  {
    const DeblockEdgeDir edgeDir = dir == 0 ? EDGE_VER : EDGE_HOR;
    std::vector<uint8_t>& emap = dir == 0 ? ev : eh;
    for (int y = 0; y < (int)pcv.heightInCtus; y++)
      for (int x = 0; x < (int)pcv.widthInCtus; x++)
      {
        const UnitArea ctuArea(pcv.chrFormat, Area(x << pcv.maxCUWidthLog2, y << pcv.maxCUHeightLog2, pcv.maxCUWidth, pcv.maxCUWidth));
        for (int tree = 0; tree < (dual ? 2 : 1); tree++)
        {
          memset(lf.m_aapucBS[edgeDir].data(), 0, lf.m_aapucBS[edgeDir].byte_size());
          memset(lf.m_aapbEdgeFilter[edgeDir].data(), false, lf.m_aapbEdgeFilter[edgeDir].byte_size());
          const ChannelType ch = tree == 0 ? CH_L : CH_C;
          for (auto& currCU : cs.traverseCUs(CS::getArea(cs, ctuArea, ch), ch))
          {
            if (dir == 0)
            {
              const Area a = currCU.Y().valid() ? (Area)currCU.Y()
                           : Area(recalcPosition(currCU.chromaFormat, currCU.chType, CHANNEL_TYPE_LUMA, currCU.blocks[currCU.chType].pos()),
                                  recalcSize(currCU.chromaFormat, currCU.chType, CHANNEL_TYPE_LUMA, currCU.blocks[currCU.chType].size()));
              for (int uy = a.y >> 2; uy < std::min<int>((a.y + a.height) >> 2, h4); uy++)
                for (int ux = a.x >> 2; ux < std::min<int>((a.x + a.width) >> 2, w4); ux++)
                {
                  if (tree == 0) { qy[uy * w4 + ux] = (int8_t)currCU.qp; if (!dual) qc[uy * w4 + ux] = (int8_t)currCU.qp; }
                  else qc[uy * w4 + ux] = (int8_t)currCU.qp;
                }
            }
            recordCU(lf, currCU, edgeDir, emap, w4);
          }
        }
      }
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
void wrap_loopFilterPic(LoopFilter* self, CodingStructure& cs)
{
  if (!shimEnabled()) { real_loopFilterPic(self, cs); return; }
  const PreCalcValues& pcv = *cs.pcv;
  CHECK(pcv.chrFormat != CHROMA_420, "vvcgpu shim: only 4:2:0");
  std::vector<uint8_t> ev, eh; std::vector<int8_t> qy, qc;
  buildDeblockMaps(*self, cs, ev, eh, qy, qc);
  PelUnitBuf rec = cs.getRecoBuf();
  g_a.upload(rec);
  g_edgeV.upload(ev.data(), ev.size()); g_edgeH.upload(eh.data(), eh.size());
  g_qpY.upload(qy.data(), qy.size()); g_qpC.upload(qc.data(), qc.size());
  vvcgpu_deblock_cfg cfg;
  cfg.bit_depth_luma = cs.sps->getBitDepth(CHANNEL_TYPE_LUMA); cfg.bit_depth_chroma = cs.sps->getBitDepth(CHANNEL_TYPE_CHROMA);
  cfg.beta_offset_div2 = cs.slice->getDeblockingFilterBetaOffsetDiv2(); cfg.tc_offset_div2 = cs.slice->getDeblockingFilterTcOffsetDiv2();
  cfg.cb_qp_offset = cs.pps->getQpOffset(COMPONENT_Cb); cfg.cr_qp_offset = cs.pps->getQpOffset(COMPONENT_Cr);
  for (int c = 0; c < 3; c++) { cfg.clp_min[c] = cs.slice->clpRng(ComponentID(c)).min; cfg.clp_max[c] = cs.slice->clpRng(ComponentID(c)).max; }
  if (!cs.slice->getDeblockingFilterDisable())
    VVCGPU(vvcgpu_deblock(g_a.p[0], g_a.stride[0], g_a.p[1], g_a.p[2], g_a.stride[1], pcv.lumaWidth, pcv.lumaHeight,
                          g_edgeV.ptr, g_edgeH.ptr, g_qpY.ptr, g_qpC.ptr, &cfg, nullptr));
  g_a.download(rec);
  g_calls[0]++;
}

void wrap_SAOProcess(SampleAdaptiveOffset* self, CodingStructure& cs, SAOBlkParam* saoBlkParams)
{
  if (!shimEnabled()) { real_SAOProcess(self, cs, saoBlkParams); return; }
  CHECK(!saoBlkParams, "No parameters present");
  self->xReconstructBlkSAOParams(cs, saoBlkParams);                      // merge resolution + de-quantisation (:262-290)
  bool any = false;
  for (int c = 0; c < 3; c++) any = any || self->m_picSAOEnabled[c];
  if (!any) return;
  const PreCalcValues& pcv = *cs.pcv;
  PelUnitBuf rec = cs.getRecoBuf();
  g_a.upload(rec);
  g_b.ensure(rec);
  const int nCtu = pcv.sizeInCtus;
  std::vector<vvcgpu_sao_ctu> prm(nCtu);
  std::vector<uint8_t> avail(nCtu);
  int idx = 0;
  for (uint32_t yPos = 0; yPos < pcv.lumaHeight; yPos += pcv.maxCUHeight)
    for (uint32_t xPos = 0; xPos < pcv.lumaWidth; xPos += pcv.maxCUWidth, idx++)
    {
      bool l, r, a, b, al, ar, bl, br;
      self->deriveLoopFilterBoundaryAvailibility(cs, Position(xPos, yPos), l, r, a, b, al, ar, bl, br);
      avail[idx] = (uint8_t)((l ? 1 : 0) | (r ? 2 : 0) | (a ? 4 : 0) | (b ? 8 : 0) | (al ? 16 : 0) | (ar ? 32 : 0) | (bl ? 64 : 0) | (br ? 128 : 0));
    }
  for (int c = 0; c < 3; c++)
  {
    const ComponentID compID = ComponentID(c);
    for (int i = 0; i < nCtu; i++)
    {
      const SAOOffset& o = cs.picture->getSAO()[i][compID];
      // offsetCTU skips the whole CTU when every component is off (:513-525); a per-component OFF is skipped at :541
      prm[i].type = (int8_t)(o.modeIdc == SAO_MODE_OFF ? -1 : o.typeIdc);
      prm[i].avail = avail[i];
      for (int k = 0; k < 32; k++) prm[i].offset[k] = 0;
      if (o.modeIdc != SAO_MODE_OFF)
      {
        if (o.typeIdc == SAO_TYPE_BO) for (int k = 0; k < 32; k++) prm[i].offset[k] = (int16_t)o.offset[k];
        else for (int k = 0; k < NUM_SAO_EO_CLASSES; k++) prm[i].offset[k] = (int16_t)o.offset[k];
      }
    }
    g_sao.upload(prm.data(), prm.size());
    const int cw = pcv.maxCUWidth >> (c ? 1 : 0), ch = pcv.maxCUHeight >> (c ? 1 : 0);
    VVCGPU(vvcgpu_sao_apply(g_a.p[c], g_a.stride[c], g_b.p[c], g_b.stride[c], g_a.w[c], g_a.h[c], cw, ch,
                            cs.sps->getBitDepth(toChannelType(compID)), g_sao.ptr, cs.slice->clpRng(compID).min, cs.slice->clpRng(compID).max, nullptr));
    VVCGPU(vvcgpu_stream_sync(nullptr));       // prm / g_sao are re-used for the next component
  }
  g_b.download(rec);
  self->xPCMLFDisableProcess(cs);
  g_calls[1]++;
}

void wrap_ALFProcess(AdaptiveLoopFilter* self, CodingStructure& cs, AlfSliceParam& alfSliceParam)
{
  if (!shimEnabled()) { real_ALFProcess(self, cs, alfSliceParam); return; }
  if (!alfSliceParam.enabledFlag[COMPONENT_Y] && !alfSliceParam.enabledFlag[COMPONENT_Cb] && !alfSliceParam.enabledFlag[COMPONENT_Cr]) return;
  alfSliceParam.filterShapes = self->m_filterShapes;
  self->m_clpRngs = cs.slice->getClpRngs();
  self->reconstructCoeff(alfSliceParam, CHANNEL_TYPE_LUMA);
  self->reconstructCoeff(alfSliceParam, CHANNEL_TYPE_CHROMA);
  const PreCalcValues& pcv = *cs.pcv;
  PelUnitBuf rec = cs.getRecoBuf();
  g_a.upload(rec);
  g_b.ensure(rec);
  const int nCtu = pcv.sizeInCtus;
  for (int c = 0; c < 3; c++) g_flags[c].upload(cs.picture->getAlfCtuEnableFlag(c), nCtu);
  g_cls.reserve((size_t)(pcv.lumaWidth >> 2) * (pcv.lumaHeight >> 2));
  const int bd = cs.sps->getBitDepth(CHANNEL_TYPE_LUMA);
  VVCGPU(vvcgpu_alf_classify(g_a.p[0], g_a.stride[0], g_a.w[0], g_a.h[0], bd, g_cls.ptr, nullptr));
  VVCGPU(vvcgpu_alf_filter_luma(g_a.p[0], g_a.stride[0], g_b.p[0], g_b.stride[0], g_a.w[0], g_a.h[0], pcv.maxCUWidth, g_cls.ptr,
                                alfSliceParam.lumaFilterType == ALF_FILTER_7 ? 1 : 0, self->m_coeffFinal, g_flags[0].ptr,
                                self->m_clpRngs.comp[0].min, self->m_clpRngs.comp[0].max, nullptr));
  for (int c = 1; c < 3; c++)
    VVCGPU(vvcgpu_alf_filter_chroma(g_a.p[c], g_a.stride[c], g_b.p[c], g_b.stride[c], g_a.w[c], g_a.h[c], pcv.maxCUWidth >> 1,
                                    alfSliceParam.chromaCoeff, g_flags[c].ptr, self->m_clpRngs.comp[c].min, self->m_clpRngs.comp[c].max, nullptr));
  g_b.download(rec);
  g_calls[2]++;
}
