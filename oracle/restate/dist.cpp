// oracle/restate/dist.cpp -- TEST INFRASTRUCTURE: CPU restatement of the block distortion functions.
// SAD  follows RdCost::xGetSAD           (CommonLib/RdCost.cpp:450-492; SIMD twin ignores the early exit, RdCostX86.h:216-310)
// HAD  follows RdCost::xGetHADs          (:2855-2974) and xCalcHADs2x2/4x4/8x8/16x8/8x16/4x8/8x4 (:2205-2853)
// SSE  follows RdCost::xGetSSE           (:1820-1857)
// MV cost follows RdCost::xGetExpGolombNumberOfBits / getCostOfVectorWithPredictor (CommonLib/RdCost.h:172-199)
// search follows InterSearch::xPatternSearch (EncoderLib/InterSearch.cpp:1887-1935).
#include "orc_common.h"
#include <cmath>
#include <vector>

ORC_API uint64_t orc_sad(const Pel* org, int os, const Pel* cur, int cs, int w, int h, int subShift)
{
  uint64_t sum = 0;
  const int step = 1 << subShift;
  for (int y = 0; y < h; y += step)
    for (int x = 0; x < w; x++) sum += abs(org[y * os + x] - cur[y * cs + x]);
  return sum << subShift;
}

ORC_API uint64_t orc_sse(const Pel* org, int os, const Pel* cur, int cs, int w, int h)
{
  uint64_t sum = 0;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) { const int d = org[y * os + x] - cur[y * cs + x]; sum += (uint64_t)(d * d); }
  return sum;
}

// sum of |2-D Walsh-Hadamard coefficients| of a tw x th tile (the butterfly order of the reference does not
// change the multiset of |coefficients|)
static int64_t hadAbsSum(const Pel* org, int os, const Pel* cur, int cs, int tw, int th)
{
  int d[16 * 16];
  for (int y = 0; y < th; y++) for (int x = 0; x < tw; x++) d[y * tw + x] = org[y * os + x] - cur[y * cs + x];
  for (int y = 0; y < th; y++)                      // rows
    for (int len = 1; len < tw; len <<= 1)
      for (int i = 0; i < tw; i += 2 * len)
        for (int j = i; j < i + len; j++)
        { const int a = d[y * tw + j], b = d[y * tw + j + len]; d[y * tw + j] = a + b; d[y * tw + j + len] = a - b; }
  for (int x = 0; x < tw; x++)                      // columns
    for (int len = 1; len < th; len <<= 1)
      for (int i = 0; i < th; i += 2 * len)
        for (int j = i; j < i + len; j++)
        { const int a = d[j * tw + x], b = d[(j + len) * tw + x]; d[j * tw + x] = a + b; d[(j + len) * tw + x] = a - b; }
  int64_t s = 0;
  for (int i = 0; i < tw * th; i++) s += abs(d[i]);
  return s;
}

static uint64_t hadTile(const Pel* org, int os, const Pel* cur, int cs, int tw, int th)
{
  const int sad = (int)hadAbsSum(org, os, cur, cs, tw, th);
  if (tw == 2 && th == 2) return (uint64_t)sad;                                   // :2205-2225
  if (tw == 4 && th == 4) return (uint64_t)((sad + 1) >> 1);                      // :2318
  if (tw == 8 && th == 8) return (uint64_t)((sad + 2) >> 2);                      // :2415
  if (tw * th == 128) return (uint64_t)(int)(sad / sqrt(16.0 * 8) * 2);           // :2561, :2698
  return (uint64_t)(int)(sad / sqrt(4.0 * 8) * 2);                                // :2771, :2850
}

ORC_API uint64_t orc_satd(const Pel* org, int os, const Pel* cur, int cs, int w, int h)
{
  int tw, th;                                                                       // xGetHADs tile selection, isQtbt = true
  if (w > h && (h & 7) == 0 && (w & 15) == 0) { tw = 16; th = 8; }
  else if (w < h && (w & 7) == 0 && (h & 15) == 0) { tw = 8; th = 16; }
  else if (w > h && (h & 3) == 0 && (w & 7) == 0) { tw = 8; th = 4; }
  else if (w < h && (w & 3) == 0 && (h & 7) == 0) { tw = 4; th = 8; }
  else if ((h % 8 == 0) && (w % 8 == 0)) { tw = th = 8; }
  else if ((h % 4 == 0) && (w % 4 == 0)) { tw = th = 4; }
  else if ((h % 2 == 0) && (w % 2 == 0)) { tw = th = 2; }
  else return ~0ull;
  uint64_t sum = 0;
  for (int y = 0; y < h; y += th)
    for (int x = 0; x < w; x += tw) sum += hadTile(org + y * os + x, os, cur + y * cs + x, cs, tw, th);
  return sum;
}

// D4: mean-removed SAD (RdCost::xGetMRSAD, RdCost.cpp:1008-1052; the size-specific twins xGetMRSAD4..64 :1055-1815 unroll the same sums;
// the early exit only fires for a finite maximumDistortionForEarlyExit and is not part of the value) and mean-removed SATD
// (xGetMRHADs :3433-3446: org - Pel(meanDiff) through xGetHADs).  Both divisions truncate towards zero (C integer division).
ORC_API uint64_t orc_mrsad(const Pel* org, int os, const Pel* cur, int cs, int w, int h, int subShift)
{
  const int step = 1 << subShift;
  int32_t deltaSum = 0;
  for (int y = 0; y < h; y += step)
    for (int x = 0; x < w; x++) deltaSum += org[y * os + x] - cur[y * cs + x];
  const Pel offset = (Pel)(deltaSum / (w * (h >> subShift)));
  uint64_t sum = 0;
  for (int y = 0; y < h; y += step)
    for (int x = 0; x < w; x++) sum += abs(org[y * os + x] - cur[y * cs + x] - offset);
  return sum << subShift;
}

ORC_API uint64_t orc_mrsatd(const Pel* org, int os, const Pel* cur, int cs, int w, int h)
{
  int64_t acc = 0;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) acc += org[y * os + x] - cur[y * cs + x];
  const Pel offset = (Pel)(acc / (w * h));                                           // AreaBuf::meanDiff, Buffer.h:469-491
  std::vector<Pel> mod((size_t)w * h);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) mod[(size_t)y * w + x] = (Pel)(org[y * os + x] - offset);   // AreaBuf::subtract on Pel
  return orc_satd(mod.data(), w, cur, cs, w, h);
}

ORC_API int orc_dist_batch(int kind, const Pel* orgBase, const Pel* curBase, const vvcgpu_dist_desc* d, int n, uint64_t* out)
{
  for (int i = 0; i < n; i++)
  {
    const Pel* o = orgBase + d[i].org_off; const Pel* c = curBase + d[i].cur_off;
    out[i] = kind == 0 ? orc_sad(o, d[i].org_stride, c, d[i].cur_stride, d[i].w, d[i].h, d[i].sub_shift)
           : kind == 1 ? orc_satd(o, d[i].org_stride, c, d[i].cur_stride, d[i].w, d[i].h)
           : kind == 2 ? orc_sse(o, d[i].org_stride, c, d[i].cur_stride, d[i].w, d[i].h)
           : kind == 3 ? orc_mrsad(o, d[i].org_stride, c, d[i].cur_stride, d[i].w, d[i].h, d[i].sub_shift)
                       : orc_mrsatd(o, d[i].org_stride, c, d[i].cur_stride, d[i].w, d[i].h);
  }
  return 0;
}

ORC_API uint32_t orc_expgolomb_bits(int v)           // RdCost.h:172-184 (MAX_CU_SIZE 128, MAX_CU_DEPTH 7)
{
  unsigned len = 1, t = (v <= 0) ? ((unsigned)(-v) << 1) + 1 : (unsigned)(v << 1);
  while (t > 128) { len += 14; t >>= 7; }
  int lg = 0; while ((2u << lg) <= t) lg++;
  return len + (lg << 1);
}
ORC_API uint64_t orc_mvcost(const vvcgpu_mvcost* m, int x, int y)
{
  const uint32_t bits = orc_expgolomb_bits(((x << m->cost_scale) - m->pred_hor) >> m->imv_shift) +
                        orc_expgolomb_bits(((y << m->cost_scale) - m->pred_ver) >> m->imv_shift);
  return (uint64_t)(m->lambda * bits);
}

ORC_API int orc_sad_search(const Pel* org, int os, const Pel* ref, int rs, const vvcgpu_search_blk* blk, int nblk,
                           int w, int h, int subShift, int dx0, int dy0, int nx, int ny, int sx, int sy,
                           uint32_t* sadOut, const vvcgpu_mvcost* mv, vvcgpu_search_best* best)
{
  for (int b = 0; b < nblk; b++)
  {
    const Pel* o = org + blk[b].org_y * os + blk[b].org_x;
    uint64_t bestCost = ~0ull; int bx = 0, by = 0; uint64_t bsad = 0;
    for (int j = 0; j < ny; j++)
      for (int i = 0; i < nx; i++)
      {
        const int x = dx0 + i * sx, y = dy0 + j * sy;
        const Pel* c = ref + (int64_t)(blk[b].ref_y + y) * rs + blk[b].ref_x + x;
        const uint64_t sad = orc_sad(o, os, c, rs, w, h, subShift);
        if (sadOut) sadOut[((int64_t)b * ny + j) * nx + i] = (uint32_t)sad;
        if (mv && best)
        {
          const uint64_t cost = sad + orc_mvcost(mv, x, y);
          if (cost < bestCost) { bestCost = cost; bx = x; by = y; bsad = sad; }
        }
      }
    if (mv && best) { best[b].x = bx; best[b].y = by; best[b].cost = bestCost; best[b].sad = bsad; }
  }
  return 0;
}
