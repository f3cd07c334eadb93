// oracle/restate/fracsearch.cpp -- TEST INFRASTRUCTURE: CPU restatement of the fractional-sample refinement.
// Follows InterSearch::xPatternSearchFracDIF (EncoderLib/InterSearch.cpp:2503-2552), xPatternRefinement (:634-689) with
// the candidate tables s_acMvRefineH/Q (:59-83).  The planes of xExtDIFUpSamplingH/Q (:3813-4093) are not materialised:
// the block the reference reads for a candidate at quarter-sample offset (qx,qy) is, sample for sample, the separable
// interpolation  filterVer<isFirst=false,isLast=true>(qy & 3)( filterHor<isFirst=true,isLast=false>(qx & 3)( ref ) )
// taken at integer offset (qx >> 2, qy >> 2) (frac 0 = the filterCopy variants) -- checked against the compiled reference
// (planes + pointer arithmetic of xPatternRefinement) by tests/test_oracle_vs_ref.py.
#include "orc_common.h"
#include <vector>

extern "C" void orc_if_filter(int N, int isVertical, int isFirst, int isLast, const Pel* src, int sstride, Pel* dst,
                              int dstride, int w, int h, const int16_t* coeff, int bd, int clpMin, int clpMax);
extern "C" const int16_t* orc_luma_filter(int frac);
extern "C" uint64_t orc_satd(const Pel* org, int os, const Pel* cur, int cs, int w, int h);
extern "C" uint64_t orc_sad(const Pel* org, int os, const Pel* cur, int cs, int w, int h, int subShift);
extern "C" uint64_t orc_mvcost(const vvcgpu_mvcost* m, int x, int y);

// candidate block at quarter-sample offset (qx,qy) from the integer-MV position `ref`
ORC_API void orc_frac_block(const Pel* ref, int rs, int w, int h, int qx, int qy, int bd, int cmin, int cmax, Pel* out)
{
  const int ix = qx >> 2, iy = qy >> 2, fx = (qx & 3) << 2, fy = (qy & 3) << 2;
  const Pel* src = ref + iy * rs + ix;
  std::vector<Pel> tmp((size_t)w * (h + 7));
  // horizontal, first stage, h+7 rows starting 3 rows above
  orc_if_filter(fx ? 8 : 0, 0, 1, 0, src - 3 * rs, rs, tmp.data(), w, w, h + 7, orc_luma_filter(fx), bd, cmin, cmax);
  // vertical, last stage
  orc_if_filter(fy ? 8 : 0, 1, 0, 1, tmp.data() + 3 * w, w, out, w, w, h, orc_luma_filter(fy), bd, cmin, cmax);
}

ORC_API int orc_frac_refine(const Pel* org, int os, const Pel* ref, int rs, const vvcgpu_frac_blk* blk, int n, int w, int h,
                            int bd, int cmin, int cmax, int useHad, const vvcgpu_mvcost* mv, vvcgpu_frac_result* res)
{
  static const int refH[9][2] = { {0,0},{0,-1},{0,1},{-1,0},{1,0},{-1,-1},{1,-1},{-1,1},{1,1} };
  static const int refQ[9][2] = { {0,0},{0,-1},{0,1},{-1,-1},{1,-1},{-1,0},{1,0},{-1,1},{1,1} };
  std::vector<Pel> cand((size_t)w * h);
  for (int b = 0; b < n; b++)
  {
    const Pel* o = org + blk[b].org_y * os + blk[b].org_x;
    const Pel* r = ref + (int64_t)blk[b].ref_y * rs + blk[b].ref_x;
    vvcgpu_mvcost m = *mv;
    m.imv_shift = 0;
    auto dist = [&](int qx, int qy) {
      orc_frac_block(r, rs, w, h, qx, qy, bd, cmin, cmax, cand.data());
      return useHad ? orc_satd(o, os, cand.data(), w, w, h) : orc_sad(o, os, cand.data(), w, w, h, 0);
    };
    // half-sample stage: cost scale 1, MV in half units = (mvInt << 1) + offset   (:2533-2538)
    m.cost_scale = 1;
    uint64_t best = ~0ull; int bi = 0;
    for (int i = 0; i < 9; i++)
    {
      const uint64_t c = dist(2 * refH[i][0], 2 * refH[i][1]) + orc_mvcost(&m, (blk[b].mv_x << 1) + refH[i][0], (blk[b].mv_y << 1) + refH[i][1]);
      if (c < best) { best = c; bi = i; }
    }
    const int hx = refH[bi][0], hy = refH[bi][1];
    res[b].half_x = hx; res[b].half_y = hy; res[b].cost_half = best;
    // quarter-sample stage: cost scale 0, MV in quarter units = ((mvInt << 1) + half) << 1 + offset   (:2541-2549)
    m.cost_scale = 0;
    best = ~0ull; bi = 0;
    for (int i = 0; i < 9; i++)
    {
      const int qx = 2 * hx + refQ[i][0], qy = 2 * hy + refQ[i][1];
      const uint64_t c = dist(qx, qy) + orc_mvcost(&m, (((blk[b].mv_x << 1) + hx) << 1) + refQ[i][0], (((blk[b].mv_y << 1) + hy) << 1) + refQ[i][1]);
      if (c < best) { best = c; bi = i; }
    }
    res[b].qter_x = refQ[bi][0]; res[b].qter_y = refQ[bi][1]; res[b].cost = best;
  }
  return 0;
}
