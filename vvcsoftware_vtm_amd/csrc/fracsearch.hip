// fracsearch.hip -- fused fractional-sample refinement of a PU (I2 + D2 + D5) for gfx950.
//
// Reference behaviour reproduced (bit-exact): InterSearch::xPatternSearchFracDIF (EncoderLib/InterSearch.cpp:2503-2552):
// xExtDIFUpSamplingH (:3813-3869), xPatternRefinement (:634-689, candidate order s_acMvRefineH/Q :59-83, strict '<'),
// xExtDIFUpSamplingQ (:3882-4093); distortion = xGetHADs (CommonLib/RdCost.cpp:2855-2974) or SAD; MV cost RdCost.h:172-199
// with cost scale 1 (half stage) / 0 (quarter stage).
//
// Design: the reference writes up to 12 fractional planes per PU to memory and re-reads them 18 times.  Here one
// (sub-)workgroup owns a PU: org block and the (W+9)x(H+9) reference window are staged once in LDS; per stage and per
// distinct horizontal phase the 14-bit first-stage plane is built in LDS, each candidate's block is produced by a
// sliding-window vertical pass (one LDS read per output sample) and consumed immediately by the Hadamard
// (tile row per lane, vertical butterflies with wave shuffles).  Nothing but the 32-byte result leaves the CU.
#include "common.h"

namespace {

__constant__ short c_lumaF[16][8] = {
  {  0, 0,   0, 64,  0,   0,  0,  0 }, {  0, 1,  -3, 63,  4,  -2,  1,  0 }, { -1, 2,  -5, 62,  8,  -3,  1,  0 },
  { -1, 3,  -8, 60, 13,  -4,  1,  0 }, { -1, 4, -10, 58, 17,  -5,  1,  0 }, { -1, 4, -11, 52, 26,  -8,  3, -1 },
  { -1, 3,  -9, 47, 31, -10,  4, -1 }, { -1, 4, -11, 45, 34, -10,  4, -1 }, { -1, 4, -11, 40, 40, -11,  4, -1 },
  { -1, 4, -10, 34, 45, -11,  4, -1 }, { -1, 4, -10, 31, 47,  -9,  3, -1 }, { -1, 3,  -8, 26, 52, -11,  4, -1 },
  {  0, 1,  -5, 17, 58, -10,  4, -1 }, {  0, 1,  -4, 13, 60,  -8,  3, -1 }, {  0, 1,  -3,  8, 62,  -5,  2, -1 },
  {  0, 1,  -2,  4, 63,  -3,  1,  0 } };
__constant__ signed char c_refH[9][2] = { {0,0},{0,-1},{0,1},{-1,0},{1,0},{-1,-1},{1,-1},{-1,1},{1,1} };
__constant__ signed char c_refQ[9][2] = { {0,0},{0,-1},{0,1},{-1,-1},{1,-1},{-1,0},{1,0},{-1,1},{1,1} };

constexpr int OFFS = 1 << 13;

__device__ __forceinline__ unsigned eg_bits(int v)
{
  unsigned len = 1, t = (v <= 0) ? ((unsigned)(-v) << 1) + 1 : (unsigned)(v << 1);
  while (t > 128u) { len += 14; t >>= 7; }
  return len + ((31 - __clz((int)t)) << 1);
}
__device__ __forceinline__ unsigned long long mv_cost(double lambda, int predH, int predV, int scale, int x, int y)
{
  const unsigned bits = eg_bits((x << scale) - predH) + eg_bits((y << scale) - predV);
  return (unsigned long long)(lambda * (double)bits);
}

// Hadamard SATD of (org - pred), both in LDS with pitch w; tiles spread over the lanes of `nw` waves; returns the
// partial sum of THIS wave's tiles in every lane (caller combines the waves).
template <int TW, int TH>
__device__ __forceinline__ unsigned long long satd_lds(const short* org, const short* pred, int w, int h, int lane, int wave, int nw)
{
  constexpr int GROUPS = 64 / TH;
  const int row = lane % TH, grp = lane / TH;
  const int tilesX = w / TW, nTiles = tilesX * (h / TH);
  unsigned long long total = 0;
  for (int t0 = wave * GROUPS; t0 < nTiles; t0 += nw * GROUPS)
  {
    const int t = t0 + grp;
    const bool act = t < nTiles;
    int v[TW];
    if (act)
    {
      const int ty = t / tilesX, tx = t - ty * tilesX;
      const int o = (ty * TH + row) * w + tx * TW;
#pragma unroll
      for (int x = 0; x < TW; x++) v[x] = (int)org[o + x] - (int)pred[o + x];
    }
    else
    {
#pragma unroll
      for (int x = 0; x < TW; x++) v[x] = 0;
    }
#pragma unroll
    for (int len = 1; len < TW; len <<= 1)
#pragma unroll
      for (int i = 0; i < TW; i += 2 * len)
#pragma unroll
        for (int j = i; j < i + len; j++) { const int a = v[j], b = v[j + len]; v[j] = a + b; v[j + len] = a - b; }
#pragma unroll
    for (int len = 1; len < TH; len <<= 1)
    {
      const bool upper = row & len;
#pragma unroll
      for (int x = 0; x < TW; x++) { const int p = __shfl_xor(v[x], len); v[x] = upper ? p - v[x] : v[x] + p; }
    }
    int s = 0;
#pragma unroll
    for (int x = 0; x < TW; x++) s += abs(v[x]);
#pragma unroll
    for (int len = 1; len < TH; len <<= 1) s += __shfl_xor(s, len);
    if (act && row == 0)
    {
      unsigned long long n;
      if (TW == 2) n = (unsigned long long)s;
      else if (TW == 4 && TH == 4) n = (unsigned long long)((s + 1) >> 1);
      else if (TW == 8 && TH == 8) n = (unsigned long long)((s + 2) >> 2);
      else if (TW * TH == 128) n = (unsigned long long)(int)((double)s / sqrt(16.0 * 8) * 2);
      else n = (unsigned long long)(int)((double)s / sqrt(4.0 * 8) * 2);
      total += n;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o);
  return total;
}

__device__ __forceinline__ unsigned long long dist_lds(const short* org, const short* pred, int w, int h, int useHad, int lane, int wave, int nw)
{
  if (!useHad)
  {
    unsigned long long acc = 0;
    for (int i = wave * 64 + lane; i < w * h; i += nw * 64) acc += (unsigned)abs((int)org[i] - (int)pred[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    return acc;
  }
  if (w > h && (h & 7) == 0 && (w & 15) == 0)      return satd_lds<16, 8>(org, pred, w, h, lane, wave, nw);
  else if (w < h && (w & 7) == 0 && (h & 15) == 0) return satd_lds<8, 16>(org, pred, w, h, lane, wave, nw);
  else if (w > h && (h & 3) == 0 && (w & 7) == 0)  return satd_lds<8, 4>(org, pred, w, h, lane, wave, nw);
  else if (w < h && (w & 3) == 0 && (h & 7) == 0)  return satd_lds<4, 8>(org, pred, w, h, lane, wave, nw);
  else if ((h & 7) == 0 && (w & 7) == 0)           return satd_lds<8, 8>(org, pred, w, h, lane, wave, nw);
  else if ((h & 3) == 0 && (w & 3) == 0)           return satd_lds<4, 4>(org, pred, w, h, lane, wave, nw);
  return satd_lds<2, 2>(org, pred, w, h, lane, wave, nw);
}

// one wave per PU needs no workgroup barrier: LDS operations of a wave execute in order; the fence keeps the compiler from
// moving LDS accesses across the point.  Four-wave groups (large PUs) use the real barrier.
#define GROUP_SYNC() do { if (nw == 1) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } else __syncthreads(); } while (0)

struct FracLds { short* org; short* win; short* hpl; short* pred; unsigned long long* cost; int* sel; };

// gsz lanes (1 or 4 waves) cooperate on one PU; all groups of the workgroup execute the same barrier sequence.
__global__ __launch_bounds__(256) void frac_refine_kernel(const Pel* __restrict__ org, int os, const Pel* __restrict__ ref, int rs,
                                                          const vvcgpu_frac_blk* __restrict__ blocks, int nblocks, int w, int h,
                                                          int bd, int cmin, int cmax, int useHad, vvcgpu_mvcost mv, int groups,
                                                          int groupBytes, vvcgpu_frac_result* __restrict__ results)
{
  extern __shared__ __align__(16) unsigned char smem[];
  const int gsz = 256 / groups, grp = threadIdx.x / gsz, tid = threadIdx.x - grp * gsz;
  const int lane = tid & 63, wave = tid >> 6, nw = gsz >> 6;
  const int b = blockIdx.x * groups + grp;
  const bool active = b < nblocks;
  const int wp = w + 9 + 1;                        // window pitch (cols -4 .. w+4, +1 pad)
  const int WR = h + 9;                            // window rows -4 .. h+4
  unsigned char* base = smem + (size_t)grp * groupBytes;
  FracLds L;
  L.cost = reinterpret_cast<unsigned long long*>(base);                 // [0..8] candidate distortions, [16 + 4i + wave] partials
  L.sel = reinterpret_cast<int*>(base + 512);
  L.org = reinterpret_cast<short*>(base + 544);
  L.win = L.org + ((w * h + 7) & ~7);
  L.hpl = L.win + ((wp * WR + 7) & ~7);
  L.pred = L.hpl + ((w * (h + 8) + 7) & ~7);

  vvcgpu_frac_blk blk = { 0, 0, 0, 0, 0, 0 };
  if (active)
  {
    blk = blocks[b];
    const Pel* o = org + (size_t)blk.org_y * os + blk.org_x;
    for (int i = tid; i < w * h; i += gsz) { const int y = i / w, x = i - y * w; L.org[i] = o[(size_t)y * os + x]; }
    const Pel* r0 = ref + (ptrdiff_t)(blk.ref_y - 4) * rs + blk.ref_x - 4;
    for (int i = tid; i < (w + 9) * WR; i += gsz) { const int y = i / (w + 9), x = i - y * (w + 9); L.win[y * wp + x] = r0[(ptrdiff_t)y * rs + x]; }
  }
  const int headRoom = max(2, 14 - bd);
  int hx = 0, hy = 0;
  for (int stage = 0; stage < 2; stage++)
  {
    // candidate i of this stage sits at quarter offset (bx + dx_i * step, by + dy_i * step)
    const int step = stage == 0 ? 2 : 1;
    const int bx = stage == 0 ? 0 : 2 * hx, by = stage == 0 ? 0 : 2 * hy;
    for (int cxi = -1; cxi <= 1; cxi++)
    {
      const int qx = bx + cxi * step;
      const int ix = qx >> 2, fx = (qx & 3) << 2;
      GROUP_SYNC();                                           // window / previous users of hpl, pred done
      if (active)
      {
        // first-stage horizontal plane, rows -4 .. h+3 (h+8 rows), cols 0 .. w-1 at integer offset ix
        const short* cf = c_lumaF[fx];
        const int shift1 = 6 - headRoom, off1 = -(OFFS << shift1);
        for (int i = tid; i < w * (h + 8); i += gsz)
        {
          const int r = i / w, x = i - r * w;
          const short* s = L.win + r * wp + x + ix + 1;         // sample (x + ix - 3) of row r-4
          int v;
          if (fx == 0) v = (short)((short)(s[3] << headRoom) - (short)OFFS);
          else
          {
            int sum = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) sum += s[k] * cf[k];
            v = (short)((sum + off1) >> shift1);
          }
          L.hpl[i] = (short)v;
        }
      }
      GROUP_SYNC();
      for (int cyi = -1; cyi <= 1; cyi++)
      {
        const int qy = by + cyi * step;
        const int iy = qy >> 2, fy = (qy & 3) << 2;
        if (active)
        {
          // last-stage vertical pass with a sliding 8-row window per column segment of 4 rows
          const short* cf = c_lumaF[fy];
          const int shift2 = 6 + headRoom, off2 = (1 << (shift2 - 1)) + (OFFS << 6);
          const int nseg = (h + 3) >> 2;
          for (int i = tid; i < w * nseg; i += gsz)
          {
            const int seg = i / w, x = i - seg * w;
            const int y0 = seg * 4;
            const short* hp = L.hpl + (y0 + iy + 1) * w + x;    // row (y0 + iy - 3) of the plane (plane row 0 = picture row -4)
            if (fy == 0)
            {
              for (int y = y0; y < min(y0 + 4, h); y++)
              {
                const int s = hp[(y - y0 + 3) * w];
                L.pred[y * w + x] = (short)clip3(cmin, cmax, (short)((s + OFFS + (1 << (headRoom - 1))) >> headRoom));
              }
            }
            else
            {
              int win8[8];
#pragma unroll
              for (int k = 0; k < 7; k++) win8[k + 1] = hp[k * w];
#pragma unroll
              for (int yy = 0; yy < 4; yy++)
              {
#pragma unroll
                for (int k = 0; k < 7; k++) win8[k] = win8[k + 1];
                if (y0 + yy < h)
                {
                  win8[7] = hp[(yy + 7) * w];
                  int sum = 0;
#pragma unroll
                  for (int k = 0; k < 8; k++) sum += win8[k] * cf[k];
                  L.pred[(y0 + yy) * w + x] = (short)clip3(cmin, cmax, (short)((sum + off2) >> shift2));
                }
              }
            }
          }
        }
        GROUP_SYNC();
        if (active)
        {
          const unsigned long long d = dist_lds(L.org, L.pred, w, h, useHad, lane, wave, nw);
          // which candidate index has offsets (cxi, cyi)?
          int ci = 0;
          for (int i = 0; i < 9; i++)
          {
            const int dx = stage == 0 ? c_refH[i][0] : c_refQ[i][0], dy = stage == 0 ? c_refH[i][1] : c_refQ[i][1];
            if (dx == cxi && dy == cyi) ci = i;
          }
          if (lane == 0) L.cost[16 + ci * 4 + wave] = d;              // per-wave partials, summed after the barrier
        }
        GROUP_SYNC();
        if (active && tid == 0)
        {
          int ci = 0;
          for (int i = 0; i < 9; i++)
          {
            const int dx = stage == 0 ? c_refH[i][0] : c_refQ[i][0], dy = stage == 0 ? c_refH[i][1] : c_refQ[i][1];
            if (dx == cxi && dy == cyi) ci = i;
          }
          unsigned long long s = 0;
          for (int k = 0; k < nw; k++) s += L.cost[16 + ci * 4 + k];
          L.cost[ci] = s;
        }
      }
    }
    GROUP_SYNC();
    if (active && tid == 0)
    {
      unsigned long long best = ~0ull;
      int bi = 0;
      for (int i = 0; i < 9; i++)
      {
        const int dx = stage == 0 ? c_refH[i][0] : c_refQ[i][0], dy = stage == 0 ? c_refH[i][1] : c_refQ[i][1];
        unsigned long long c;
        if (stage == 0) c = L.cost[i] + mv_cost(mv.lambda, mv.pred_hor, mv.pred_ver, 1, (blk.mv_x << 1) + dx, (blk.mv_y << 1) + dy);
        else c = L.cost[i] + mv_cost(mv.lambda, mv.pred_hor, mv.pred_ver, 0, (((blk.mv_x << 1) + hx) << 1) + dx, (((blk.mv_y << 1) + hy) << 1) + dy);
        if (c < best) { best = c; bi = i; }
      }
      const int dx = stage == 0 ? c_refH[bi][0] : c_refQ[bi][0], dy = stage == 0 ? c_refH[bi][1] : c_refQ[bi][1];
      L.sel[0] = dx; L.sel[1] = dy;
      if (stage == 0) { results[b].half_x = dx; results[b].half_y = dy; results[b].cost_half = best; }
      else { results[b].qter_x = dx; results[b].qter_y = dy; results[b].cost = best; }
    }
    GROUP_SYNC();
    if (stage == 0) { hx = L.sel[0]; hy = L.sel[1]; }
  }
}

}  // namespace

extern "C" int vvcgpu_frac_refine(const vvc_pel* org, int org_stride, const vvc_pel* ref, int ref_stride,
                                  const vvcgpu_frac_blk* blocks, int nblocks, int w, int h, int bit_depth, int clp_min,
                                  int clp_max, int use_hadamard, const vvcgpu_mvcost* mvcost_host,
                                  vvcgpu_frac_result* results, void* stream)
{
  VVC_CHECK_ARG(nblocks >= 0, "frac_refine: nblocks %d", nblocks);
  if (nblocks == 0) return VVCGPU_OK;
  VVC_CHECK_ARG(org && ref && blocks && mvcost_host && results, "frac_refine: null pointer");
  VVC_CHECK_ARG(w >= 4 && w <= 128 && h >= 4 && h <= 128 && (w & 3) == 0 && (h & 3) == 0, "frac_refine: block %dx%d unsupported", w, h);
  if (bit_depth < 8 || bit_depth > 10) { vvcgpu_set_error("frac_refine: bit depth %d outside 8..10", bit_depth); return VVCGPU_E_UNSUPPORTED; }
  const int wp = w + 10, WR = h + 9;
  auto al8 = [](size_t v) { return (v + 7) & ~(size_t)7; };
  const size_t shorts = al8((size_t)w * h) + al8((size_t)wp * WR) + al8((size_t)w * (h + 8)) + al8((size_t)w * h);
  const size_t groupBytes = (544 + shorts * 2 + 15) & ~(size_t)15;
  const int groups = (w * h <= 1024) ? 4 : 1;
  const size_t smem = groupBytes * groups;
  VVC_CHECK_ARG(smem <= 160 * 1024, "frac_refine: LDS need %zu too large", smem);
  hipStream_t st = (hipStream_t)stream;
  if (smem > 64 * 1024)
    VVC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(frac_refine_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL(frac_refine_kernel, dim3(cdiv(nblocks, groups)), dim3(256), smem, st, org, org_stride, ref, ref_stride, blocks,
                     nblocks, w, h, bit_depth, clp_min, clp_max, use_hadamard, *mvcost_host, groups, (int)groupBytes, results);
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}
