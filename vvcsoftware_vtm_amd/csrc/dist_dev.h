// dist_dev.h -- device functions of the block distortions shared by dist.hip and the fused "predict -> distortion" entries (intra.hip).
// Reference behaviour: RdCost::xGetHADs CommonLib/RdCost.cpp:2855-2974, xCalcHADs* :2205-2853 (see dist.hip).
#pragma once
#include "common.h"

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// sum over an aligned group of G lanes (G = 16: four descriptors side by side in a wave; G = 64: the wave)
template <int G>
__device__ __forceinline__ unsigned long long group_sum_u64(unsigned long long v)
{
#pragma unroll
  for (int o = G >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---------------------------------------------------------------------------------------------------
// Hadamard tile: TW columns in registers, TH rows across TH consecutive lanes.  `lane` = lane index inside a group of G lanes that share the block.
// CurPtr: the type of the second block's pointer (plain, or address-space qualified when the caller keeps it in LDS: a plain pointer to LDS costs
// flat loads).
template <int TW, int TH, int G = 64, class CurPtr = const Pel*>
__device__ __forceinline__ unsigned long long satd_tiles(const Pel* org, int os, CurPtr cur, int cs, int w, int h, int lane, int offset = 0)
{
  constexpr int GROUPS = G / TH;
  const int row = lane % TH, grp = lane / TH;
  const int tilesX = w / TW, nTiles = tilesX * (h / TH);
  unsigned long long total = 0;
  for (int t0 = 0; t0 < nTiles; t0 += GROUPS)
  {
    const int t = t0 + grp;
    const bool active = t < nTiles;
    int v[TW];
    if (active)
    {
      const int ty = t / tilesX, tx = t - ty * tilesX;
      const Pel* o = org + (size_t)(ty * TH + row) * os + tx * TW;
      CurPtr c = cur + (size_t)(ty * TH + row) * cs + tx * TW;
#pragma unroll
      for (int x = 0; x < TW; x++) v[x] = (int)(Pel)((int)o[x] - offset) - (int)c[x];      // offset != 0: D4, org - Pel(meanDiff) kept as Pel
    }
    else
    {
#pragma unroll
      for (int x = 0; x < TW; x++) v[x] = 0;
    }
    // horizontal WHT in registers
#pragma unroll
    for (int len = 1; len < TW; len <<= 1)
#pragma unroll
      for (int i = 0; i < TW; i += 2 * len)
#pragma unroll
        for (int j = i; j < i + len; j++) { const int a = v[j], b = v[j + len]; v[j] = a + b; v[j + len] = a - b; }
    // vertical WHT across the TH lanes of the group
#pragma unroll
    for (int len = 1; len < TH; len <<= 1)
    {
      const bool upper = row & len;
#pragma unroll
      for (int x = 0; x < TW; x++)
      {
        const int p = __shfl_xor(v[x], len);
        v[x] = upper ? p - v[x] : v[x] + p;
      }
    }
    int s = 0;
#pragma unroll
    for (int x = 0; x < TW; x++) s += abs(v[x]);
#pragma unroll
    for (int len = 1; len < TH; len <<= 1) s += __shfl_xor(s, len);
    if (active && row == 0)
    {
      unsigned long long n;
      if (TW == 2) n = (unsigned long long)s;
      else if (TW == 4 && TH == 4) n = (unsigned long long)((s + 1) >> 1);
      else if (TW == 8 && TH == 8) n = (unsigned long long)((s + 2) >> 2);
      else if (TW * TH == 128) n = (unsigned long long)(int)((double)s / sqrt(16.0 * 8) * 2);   // RdCost.cpp:2561,2698
      else n = (unsigned long long)(int)((double)s / sqrt(4.0 * 8) * 2);                        // :2771,2850
      total += n;
    }
  }
  return group_sum_u64<G>(total);
}

// Hadamard SATD of a w x h block with the reference's tile choice (xGetHADs :2855-2974); lane = index inside a group of G lanes.  hSel (default h):
// the height the tile choice is made for, when [org, cur] is a band of h rows of a taller block (h a multiple of the chosen tile height).
template <int G, class CurPtr = const Pel*>
__device__ __forceinline__ unsigned long long satd_block(const Pel* org, int os, CurPtr cur, int cs, int w, int h, int lane, int offset = 0, int hSel = 0)
{
  const int hs = hSel ? hSel : h;
  if (w > hs && (hs & 7) == 0 && (w & 15) == 0)      return satd_tiles<16, 8, G, CurPtr>(org, os, cur, cs, w, h, lane, offset);
  else if (w < hs && (w & 7) == 0 && (hs & 15) == 0) return satd_tiles<8, 16, G, CurPtr>(org, os, cur, cs, w, h, lane, offset);
  else if (w > hs && (hs & 3) == 0 && (w & 7) == 0)  return satd_tiles<8, 4, G, CurPtr>(org, os, cur, cs, w, h, lane, offset);
  else if (w < hs && (w & 3) == 0 && (hs & 7) == 0)  return satd_tiles<4, 8, G, CurPtr>(org, os, cur, cs, w, h, lane, offset);
  else if ((hs & 7) == 0 && (w & 7) == 0)            return satd_tiles<8, 8, G, CurPtr>(org, os, cur, cs, w, h, lane, offset);
  else if ((hs & 3) == 0 && (w & 3) == 0)            return satd_tiles<4, 4, G, CurPtr>(org, os, cur, cs, w, h, lane, offset);
  return satd_tiles<2, 2, G, CurPtr>(org, os, cur, cs, w, h, lane, offset);
}
