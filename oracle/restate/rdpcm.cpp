// oracle/restate/rdpcm.cpp -- TEST INFRASTRUCTURE (CPU restatement, the checker of the HIP path; never part of the product).
// Residual DPCM of transform-skipped / lossless TUs, restated from the arithmetic of
//   TrQuant::applyForwardRDPCM (CommonLib/TrQuant.cpp:991-1045), Quant::transformSkipQuantOneSample / invTrSkipDeQuantOneSample
//   (CommonLib/Quant.cpp:911-1090, flat scaling lists, maxLog2TrDynamicRange 15), TrQuant::invRdpcmNxN (TrQuant.cpp:632-688),
// and the affine sub-block vector derivation of InterPrediction::xPredAffineBlk (CommonLib/InterPrediction.cpp:618-701, roundAffineMv Mv.cpp:56-61).
// Pinned against the compiled reference by tests/golden/rdpcm.npz and tests/golden/affine_mv.npz (tests/test_oracle_golden.py).
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include "orc_common.h"

static int log2i(int v) { int l = 0; while ((1 << (l + 1)) <= v) l++; return l; }
static const int kQuant[6] = { 26214, 23302, 20560, 18396, 16384, 14564 }, kInvQuant[6] = { 40, 45, 51, 57, 64, 72 };

extern "C" int orc_rdpcm_fwd_batch(const int16_t* resiBase, int32_t* coeffBase, const vvcgpu_rdpcm_desc* descs, int n, int bd, uint32_t* absSum)
{
  for (int i = 0; i < n; i++)
  {
    const vvcgpu_rdpcm_desc& d = descs[i];
    const int w = d.w, h = d.h, per = d.qp / 6, rem = d.qp % 6;
    const int trShift = 15 - bd - ((log2i(w) + log2i(h)) >> 1);
    const int qBits = 14 + per + trShift;
    const int add = (int)((int64_t)(d.mode != 0 ? 256 : (d.intra_slice ? 171 : 85)) << (int64_t)(qBits - 9));
    const int rightShift = 6 - (trShift + per);
    const int targetBits = std::min(16, 32 + rightShift - 7);
    const int inMin = -(1 << (targetBits - 1)), inMax = (1 << (targetBits - 1)) - 1;
    const int16_t* resi = resiBase + d.resi_off;
    int32_t* coeff = coeffBase + d.coeff_off;
    uint32_t sum = 0;
    const int nMajor = d.mode == 2 ? w : h, nMinor = d.mode == 2 ? h : w;
    for (int major = 0; major < nMajor; major++)
    {
      int32_t acc = 0;
      for (int minor = 0; minor < nMinor; minor++)
      {
        const int x = d.mode == 2 ? major : minor, y = d.mode == 2 ? minor : major;
        const int si = y * w + x, ci = d.rotate ? w * h - 1 - si : si;
        const int32_t delta = (int32_t)resi[(ptrdiff_t)y * d.resi_stride + x] - acc;
        int32_t lv; int16_t rec;
        if (d.lossless) { lv = delta; rec = (int16_t)delta; }
        else
        {
          const int32_t tc = trShift >= 0 ? (int32_t)((uint32_t)delta << trShift) : (delta + (1 << (-trShift - 1))) >> -trShift;
          const int32_t sign = tc < 0 ? -1 : 1;
          const int64_t tmp = (int64_t)std::abs(tc) * kQuant[rem];
          lv = std::min(std::max((int32_t)((tmp + add) >> qBits) * sign, -32768), 32767);
          const int32_t c = std::min(std::max(lv, inMin), inMax);
          int32_t v;
          if (rightShift > 0) v = (int32_t)((uint32_t)(c * kInvQuant[rem]) + (1u << (rightShift - 1))) >> rightShift;
          else v = (int32_t)((uint32_t)(c * kInvQuant[rem]) << -rightShift);
          v = std::min(std::max(v, -32768), 32767);
          rec = trShift >= 0 ? (int16_t)((v + (trShift == 0 ? 0 : 1 << (trShift - 1))) >> trShift) : (int16_t)(v << -trShift);
        }
        coeff[ci] = lv;
        sum += (uint32_t)std::abs(lv);
        if (d.mode != 0) acc += rec;
      }
    }
    absSum[i] = sum;
  }
  return 0;
}

extern "C" int orc_rdpcm_inv_batch(int16_t* resiBase, const vvcgpu_rdpcm_desc* descs, int n)
{
  for (int i = 0; i < n; i++)
  {
    const vvcgpu_rdpcm_desc& d = descs[i];
    if (d.mode == 0) continue;
    int16_t* resi = resiBase + d.resi_off;
    const int nMajor = d.mode == 2 ? d.w : d.h, nMinor = d.mode == 2 ? d.h : d.w;
    const ptrdiff_t sMinor = d.mode == 2 ? d.resi_stride : 1, sMajor = d.mode == 2 ? 1 : d.resi_stride;
    for (int major = 0; major < nMajor; major++)
    {
      int16_t* p = resi + major * sMajor;
      int32_t acc = p[0];
      for (int minor = 1; minor < nMinor; minor++) { acc += p[minor * sMinor]; p[minor * sMinor] = (int16_t)std::min(std::max(acc, -32768), 32767); }
    }
  }
  return 0;
}

extern "C" int orc_affine_subblock_descs(const vvcgpu_affine_pu* pus, int n, int comp, int picW, int picH, int maxCuW, int maxCuH,
                                         int orgX, int orgY, int rs0, int rs1, vvcgpu_mc_desc* out)
{
  for (int pi = 0; pi < n; pi++)
  {
    const vvcgpu_affine_pu& pu = pus[pi];
    const int sc = comp ? 1 : 0, bw = 4 >> sc, bh = 4 >> sc, cxW = pu.w >> sc, cxH = pu.h >> sc;
    const int iBit = 7, shift = iBit - 4 + 2 + 2;
    const int horMax = (picW + 8 - pu.pos_x - 1) << 4, horMin = (-maxCuW - 8 - pu.pos_x + 1) << 4;
    const int verMax = (picH + 8 - pu.pos_y - 1) << 4, verMin = (-maxCuH - 8 - pu.pos_y + 1) << 4;
    int k = 0;
    for (int hq = 0; hq < cxH; hq += bh)
      for (int wq = 0; wq < cxW; wq += bw, k++)
      {
        vvcgpu_mc_desc d = {};
        d.w = (int16_t)bw; d.h = (int16_t)bh; d.is_luma = comp ? 0 : 1; d.bi = pu.bi ? 1 : 0;
        d.dst_off = pu.dst_off + (int64_t)hq * pu.dst_stride + wq; d.dst_stride = pu.dst_stride; d.ref0_stride = rs0; d.ref1_stride = rs1;
        for (int l = 0; l < (pu.bi ? 2 : 1); l++)
        {
          const int ltx = pu.mv[l][0][0], lty = pu.mv[l][0][1];
          const int dHorX = (pu.mv[l][1][0] - ltx) << (iBit - log2i(cxW)), dHorY = (pu.mv[l][1][1] - lty) << (iBit - log2i(cxW));
          int dVerX, dVerY;
          if (pu.six_param) { dVerX = (pu.mv[l][2][0] - ltx) << (iBit - log2i(cxH)); dVerY = (pu.mv[l][2][1] - lty) << (iBit - log2i(cxH)); }
          else { dVerX = -dHorY; dVerY = dHorX; }
          int mh = (ltx << iBit) + dHorX * ((bw >> 1) + wq) + dVerX * ((bh >> 1) + hq);
          int mv = (lty << iBit) + dHorY * ((bw >> 1) + wq) + dVerY * ((bh >> 1) + hq);
          const int off = 1 << (shift - 1);
          mh = mh >= 0 ? (mh + off) >> shift : -((-mh + off) >> shift);
          mv = mv >= 0 ? (mv + off) >> shift : -((-mv + off) >> shift);
          mh = std::min(horMax, std::max(horMin, mh));
          mv = std::min(verMax, std::max(verMin, mv));
          const int xInt = mh >> (4 + sc), xFrac = mh & (sc ? 31 : 15), yInt = mv >> (4 + sc), yFrac = mv & (sc ? 31 : 15);
          const int64_t ro = (int64_t)((pu.pos_y >> sc) + hq + yInt + orgY) * (l ? rs1 : rs0) + (pu.pos_x >> sc) + wq + xInt + orgX;
          if (l == 0) { d.ref0_off = ro; d.frac_x0 = (int8_t)xFrac; d.frac_y0 = (int8_t)yFrac; }
          else        { d.ref1_off = ro; d.frac_x1 = (int8_t)xFrac; d.frac_y1 = (int8_t)yFrac; }
        }
        out[pu.first_desc + k] = d;
      }
  }
  return 0;
}
