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
#include "mfma_tr.h"
#include <mutex>

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
                                                          int bd, int cmin, int cmax, int useHad, vvcgpu_mvcost mv,
                                                          const int* __restrict__ preds, int groups,
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
    if (preds) { mv.pred_hor = preds[2 * b]; mv.pred_ver = preds[2 * b + 1]; }     // per-PU predictor (vvcgpu_me_batch)
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


// ---------------------------------------------------------------------------------------------------
// 16x16 specialisation (the PU size the canonical workload refines; 8x8 Hadamard tiles or SAD): one WAVE per PU, no
// workgroup barriers, the candidate blocks never touch LDS.
//   lane = (tile t = lane >> 4, row-quad rq = (lane >> 3) & 1, column c = lane & 7): the lane owns column x = 8 (t & 1) + c,
//   rows y0 .. y0+3 with y0 = 8 (t >> 1) + 4 rq, so one 16-lane DPP row holds one 8x8 Hadamard tile.
//   * window 24x24 and the 14-bit first-stage planes live in LDS (per wave); a plane is built by 48 lanes, each
//     filtering 8 or 9 neighbouring outputs of one row from 8 dword reads; the two half-sample planes at integer
//     offsets -1 / 0 are one 17-column plane, and the quarter stage reuses the half stage's plane for dx = 0.
//   * per (plane, column) the lane loads its 12 plane rows ONCE and derives the three vertical candidates from
//     registers; the residual goes straight into the Hadamard: rows in registers, the remaining row stage and the
//     column stages with DPP (row_ror:8 for the row halves, quad_perm for xor 1 / 2, a row_shl:4 + row_shr:4
//     pair for xor 4): the coefficients are those of RdCost::xCalcHADs8x8; the transposed tile orientation is harmless
//     because H D H^T and H D^T H^T have the same absolute sum.
//   * the quarter stage's centre candidate is the half stage's winner: its distortion is reused, not recomputed.
template <int CTRL> __device__ __forceinline__ int dpp_mov(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_ROR8 = 0x128;

template <int CTRL>
__device__ __forceinline__ void had_cross(int (&v)[4], bool upper)
{
#pragma unroll
  for (int j = 0; j < 4; j++) { const int p = dpp_mov<CTRL>(v[j]); v[j] = (upper ? -v[j] : v[j]) + p; }
}

// distortion of the candidate whose 4-row column segment is pred[]: wave-uniform result
// wide (wave-uniform): some |org - pred| of this PU may exceed 1023 (a bi-predictive original 2 org - otherPred reaches [-1023, 2046]):
// the Hadamard then runs in 32-bit registers (had_cross), the packed form below holds only for |d| <= 1023.
template <bool HAD>
__device__ __forceinline__ unsigned f16_dist(const int (&orgv)[4], const int (&pred)[4], int lane, bool wide)
{
  int d[4];
#pragma unroll
  for (int j = 0; j < 4; j++) d[j] = orgv[j] - pred[j];
  int s;
  if (HAD && wide)
  {
    int v[4] = { d[0] + d[1], d[0] - d[1], d[2] + d[3], d[2] - d[3] };
    { const int a0 = v[0] + v[2], a2 = v[0] - v[2], a1 = v[1] + v[3], a3 = v[1] - v[3]; v[0] = a0; v[1] = a1; v[2] = a2; v[3] = a3; }
    had_cross<DPP_ROR8>(v, (lane & 8) != 0);
    had_cross<DPP_XOR1>(v, (lane & 1) != 0);
    had_cross<DPP_XOR2>(v, (lane & 2) != 0);
#pragma unroll
    for (int j = 0; j < 4; j++)                    // xor 4: row_shl:4 into banks 0, 2 and row_shr:4 into banks 1, 3
    {
      int p = __builtin_amdgcn_update_dpp(0, v[j], 0x104, 0xF, 0x5, false);
      p = __builtin_amdgcn_update_dpp(p, v[j], 0x114, 0xF, 0xA, false);
      v[j] = ((lane & 4) ? -v[j] : v[j]) + p;
    }
    s = abs(v[0]) + abs(v[1]) + abs(v[2]) + abs(v[3]);
  }
  else if (HAD)
  {
    // 8x8 Hadamard in PACKED 16-bit: |d| <= 1023 grows by 2 per stage, so five of the six stages fit int16 (32 x 1023 = 32736), and the
    // sixth is never formed: |a + b| + |a - b| = 2 max(|a|, |b|), i.e. every lane adds max(|own|, |partner|) and the pair is counted twice.
    // The lane's four rows are two dwords; a butterfly with the partner lane is one DPP move + one v_pk_mad_i16 (own x (+-1) + partner).
    const pel2 one = { 1, 1 }, mone = { -1, -1 }, pm = { 1, -1 };
    pel2 q0 = { (short)d[0], (short)d[1] }, q1 = { (short)d[2], (short)d[3] };
    auto rot = [](pel2 v) { const unsigned u = __builtin_bit_cast(unsigned, v); return __builtin_bit_cast(pel2, __builtin_amdgcn_alignbit(u, u, 16)); };
    q0 = rot(q0) + q0 * pm;                                               // (d0 + d1, d0 - d1)
    q1 = rot(q1) + q1 * pm;
    { const pel2 a = q0 + q1, b = q0 - q1; q0 = a; q1 = b; }
    auto cross = [&](auto mov, bool upper)
    {
      const pel2 sg = upper ? mone : one;
      q0 = __builtin_bit_cast(pel2, mov(__builtin_bit_cast(int, q0))) + q0 * sg;
      q1 = __builtin_bit_cast(pel2, mov(__builtin_bit_cast(int, q1))) + q1 * sg;
    };
    cross([](int v) { return dpp_mov<DPP_ROR8>(v); }, (lane & 8) != 0);
    cross([](int v) { return dpp_mov<DPP_XOR1>(v); }, (lane & 1) != 0);
    cross([](int v) { return dpp_mov<DPP_XOR2>(v); }, (lane & 2) != 0);
    auto xor4 = [](int v)                                                 // row_shl:4 into banks 0, 2 and row_shr:4 into banks 1, 3
    {
      int p = __builtin_amdgcn_update_dpp(0, v, 0x104, 0xF, 0x5, false);
      return __builtin_amdgcn_update_dpp(p, v, 0x114, 0xF, 0xA, false);
    };
    const pel2 a0 = __builtin_elementwise_max(q0, -q0), a1 = __builtin_elementwise_max(q1, -q1);
    const pel2 m0 = __builtin_elementwise_max(a0, __builtin_bit_cast(pel2, xor4(__builtin_bit_cast(int, a0))));
    const pel2 m1 = __builtin_elementwise_max(a1, __builtin_bit_cast(pel2, xor4(__builtin_bit_cast(int, a1))));
    s = (int)m0[0] + (int)m0[1] + (int)m1[0] + (int)m1[1];
  }
  else
    s = abs(d[0]) + abs(d[1]) + abs(d[2]) + abs(d[3]);
  s += dpp_mov<DPP_XOR1>(s); s += dpp_mov<DPP_XOR2>(s); s += dpp_mov<DPP_HALF_MIRROR>(s); s += dpp_mov<DPP_ROR8>(s);   // tile sum in all 16 lanes
  unsigned t0 = (unsigned)__builtin_amdgcn_readlane(s, 0), t1 = (unsigned)__builtin_amdgcn_readlane(s, 16),
           t2 = (unsigned)__builtin_amdgcn_readlane(s, 32), t3 = (unsigned)__builtin_amdgcn_readlane(s, 48);
  if (HAD) return ((t0 + 2) >> 2) + ((t1 + 2) >> 2) + ((t2 + 2) >> 2) + ((t3 + 2) >> 2);           // xCalcHADs8x8: (sad + 2) >> 2 per tile
  return t0 + t1 + t2 + t3;
}

// The first-stage planes are kept TRANSPOSED in LDS (hpT[column + 1][plane row], 24 rows per column, plane row 0 = picture row -4), so the
// lane's column is 12 consecutive samples = three aligned ds_read_b64, and they arrive as the dword pairs that v_dot2_i32_i16 wants:
// D[m] = (col[2m], col[2m+1]); the odd pairs E[m] = (col[2m+1], col[2m+2]) cost one v_alignbit each, once per column for its three candidates.
constexpr int HPT = 24;                                                   // samples per column of a transposed plane
struct F16Col { unsigned D[6], E[5]; };
__device__ __forceinline__ void f16_load_col(const short* __restrict__ pcT, F16Col& c)
{
  const uint2* q = reinterpret_cast<const uint2*>(pcT);
#pragma unroll
  for (int m = 0; m < 3; m++) { const uint2 v = q[m]; c.D[2 * m] = v.x; c.D[2 * m + 1] = v.y; }
#pragma unroll
  for (int m = 0; m < 5; m++) c.E[m] = __builtin_amdgcn_alignbit(c.D[m + 1], c.D[m], 16);
}

// vertical (last-stage) filter of the lane's column: col[k] = plane row y0 + k, outputs rows y0 .. y0+3 at integer row offset IY (-1 / 0)
// and quarter phase fy (0..3): out[yy] = sum_k col[IY + 1 + yy + k] c[k]
template <int IY>
__device__ __forceinline__ void f16_vert(const F16Col& col, int fy, int headRoom, int cmin, int cmax, int (&out)[4])
{
  if (fy == 0)                                     // IY == 0 here: samples col[4 .. 7]
  {
#pragma unroll
    for (int yy = 0; yy < 4; yy++)
    {
      const unsigned d = col.D[2 + (yy >> 1)];
      const int v = (yy & 1) ? (int)d >> 16 : (int)(short)(d & 0xFFFF);
      out[yy] = clip3(cmin, cmax, (int)(short)((v + OFFS + (1 << (headRoom - 1))) >> headRoom));
    }
    return;
  }
  const unsigned* cf = reinterpret_cast<const unsigned*>(c_lumaF[fy << 2]);
  const int shift2 = 6 + headRoom, off2 = (1 << (shift2 - 1)) + (OFFS << 6);
  int sum[4] = { off2, off2, off2, off2 };
#pragma unroll
  for (int m = 0; m < 4; m++)
  {
    const pel2 cm = __builtin_bit_cast(pel2, cf[m]);
    if (IY < 0)                                    // starts col[0], col[1], col[2], col[3]
    {
      sum[0] = __builtin_amdgcn_sdot2(__builtin_bit_cast(pel2, col.D[m]), cm, sum[0], false);
      sum[1] = __builtin_amdgcn_sdot2(__builtin_bit_cast(pel2, col.E[m]), cm, sum[1], false);
      sum[2] = __builtin_amdgcn_sdot2(__builtin_bit_cast(pel2, col.D[m + 1]), cm, sum[2], false);
      sum[3] = __builtin_amdgcn_sdot2(__builtin_bit_cast(pel2, col.E[m + 1]), cm, sum[3], false);
    }
    else                                           // starts col[1] .. col[4]
    {
      sum[0] = __builtin_amdgcn_sdot2(__builtin_bit_cast(pel2, col.E[m]), cm, sum[0], false);
      sum[1] = __builtin_amdgcn_sdot2(__builtin_bit_cast(pel2, col.D[m + 1]), cm, sum[1], false);
      sum[2] = __builtin_amdgcn_sdot2(__builtin_bit_cast(pel2, col.E[m + 1]), cm, sum[2], false);
      sum[3] = __builtin_amdgcn_sdot2(__builtin_bit_cast(pel2, col.D[m + 2]), cm, sum[3], false);
    }
  }
#pragma unroll
  for (int yy = 0; yy < 4; yy++) out[yy] = clip3(cmin, cmax, (int)(short)(sum[yy] >> shift2));
}

// first-stage plane into hpT (transposed, column index = plane column + 1): lanes 0..47 = (row r, segment); 16-column planes at
// integer offset ix use outputs x = 0..15, the 17-column half plane (WIDE) x = -1..15.  A lane filters 8 (9) neighbouring outputs of its
// row from 8 dword reads: output with first tap at sample st of the lane's 16 is four v_dot2 over D[st/2 ..] (st even) or E[(st-1)/2 ..].
template <bool WIDE>
__device__ __forceinline__ void f16_hplane(const short* __restrict__ win, short* __restrict__ hpT, int ix, int fx, int headRoom, int lane)
{
  if (lane < 48)
  {
    const int r = lane >> 1, seg = lane & 1;
    const unsigned* wr = reinterpret_cast<const unsigned*>(win + r * 26 + seg * 8);      // 16 samples = window cols 8 seg .. 8 seg + 15
    unsigned D[8], E[7];
#pragma unroll
    for (int k = 0; k < 8; k++) D[k] = wr[k];
#pragma unroll
    for (int k = 0; k < 7; k++) E[k] = __builtin_amdgcn_alignbit(D[k + 1], D[k], 16);
    short* o = hpT + (1 + seg * 8) * HPT + r;                                            // column x of this segment: o[x * HPT]
    const int shift1 = 6 - headRoom, off1 = -(OFFS << shift1);
    const unsigned* cf = reinterpret_cast<const unsigned*>(c_lumaF[fx << 2]);
    pel2 c2[4];
#pragma unroll
    for (int m = 0; m < 4; m++) c2[m] = __builtin_bit_cast(pel2, cf[m]);
    auto tap8 = [&](int st) -> int                                                     // st: compile-time after unrolling
    {
      int sum = off1;
#pragma unroll
      for (int m = 0; m < 4; m++)
        sum = __builtin_amdgcn_sdot2(__builtin_bit_cast(pel2, (st & 1) ? E[(st >> 1) + m] : D[(st >> 1) + m]), c2[m], sum, false);
      return sum >> shift1;
    };
    if (WIDE)                                      // ix = -1..0 folded into the 17 columns: column x' starts at sample x' - 8 seg + 1
    {
      if (seg == 0)
      {
#pragma unroll
        for (int x = -1; x < 8; x++) o[x * HPT] = (short)tap8(x + 1);
      }
      else
      {
#pragma unroll
        for (int x = 0; x < 8; x++) o[x * HPT] = (short)tap8(x + 1);
      }
    }
    else if (fx == 0)
    {
#pragma unroll
      for (int x = 0; x < 8; x++)                                                        // ix == 0: sample x + 4
      {
        const unsigned d = D[(x + 4) >> 1];
        const int sv = (x & 1) ? (int)d >> 16 : (int)(short)(d & 0xFFFF);
        o[x * HPT] = (short)((short)(sv << headRoom) - (short)OFFS);
      }
    }
    else if (ix == 0)
    {
#pragma unroll
      for (int x = 0; x < 8; x++) o[x * HPT] = (short)tap8(x + 1);
    }
    else
    {
#pragma unroll
      for (int x = 0; x < 8; x++) o[x * HPT] = (short)tap8(x);
    }
  }
}

#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

__device__ __forceinline__ int f16_idx(int dx, int dy, bool quarter)        // position of (dx, dy) in s_acMvRefineH / s_acMvRefineQ
{
  if (dx == 0) return dy == 0 ? 0 : dy < 0 ? 1 : 2;
  if (!quarter) return dx < 0 ? (dy == 0 ? 3 : dy < 0 ? 5 : 7) : (dy == 0 ? 4 : dy < 0 ? 6 : 8);
  return dx < 0 ? (dy < 0 ? 3 : dy == 0 ? 5 : 7) : (dy < 0 ? 4 : dy == 0 ? 6 : 8);
}

// arg-min of dist[i] + mvcost(candidate i) over the 9 candidates, first index on ties; lanes 0..8 carry one candidate each
__device__ __forceinline__ void f16_best(const unsigned* dist, bool quarter, const vvcgpu_mvcost& mv, int baseX, int baseY, int scale,
                                         int lane, int& bdx, int& bdy, unsigned long long& bcost, unsigned& bdist)
{
  const int li = lane < 9 ? lane : 0;
  const int dx = quarter ? c_refQ[li][0] : c_refH[li][0], dy = quarter ? c_refQ[li][1] : c_refH[li][1];
  const unsigned dl = dist[li];
  unsigned long long c = lane < 9 ? (unsigned long long)dl + mv_cost(mv.lambda, mv.pred_hor, mv.pred_ver, scale, baseX + dx, baseY + dy) : ~0ull;
  int bi = lane;
#pragma unroll
  for (int o = 1; o < 16; o <<= 1)
  {
    const unsigned long long oc = __shfl_xor(c, o);
    const int oi = __shfl_xor(bi, o);
    if (oc < c || (oc == c && oi < bi)) { c = oc; bi = oi; }
  }
  bi = __builtin_amdgcn_readfirstlane(bi);
  bcost = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(c >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)c);
  bdx = quarter ? c_refQ[bi][0] : c_refH[bi][0];
  bdy = quarter ? c_refQ[bi][1] : c_refH[bi][1];
  bdist = dist[bi];
}

// LDS of one wave of the vector-pipe form
struct F16Lds { short win[24 * 26]; short hpl[3][24 * 18]; unsigned dist[16]; };      // hpl: [0] integer plane, [1] half plane (17 cols), [2] quarter planes

// one PU on the vector pipes (the round-2..4 form): the whole kernel when HAD is off; with HAD on, the PUs the matrix-core form below cannot take
template <bool HAD>
__device__ __forceinline__ void frac16_pu_valu(F16Lds& L, const Pel* __restrict__ org, int os, const Pel* __restrict__ ref, int rs,
                                               const vvcgpu_frac_blk& blk, int b, int bd, int cmin, int cmax, const vvcgpu_mvcost& mv,
                                               vvcgpu_frac_result* __restrict__ results, int lane)
{
  short* win = L.win;
  short* hp0 = L.hpl[0];
  short* hp8 = L.hpl[1];
  short* hpq = L.hpl[2];
  const int t = lane >> 4, x = 8 * (t & 1) + (lane & 7), y0 = 8 * (t >> 1) + 4 * ((lane >> 3) & 1);
  int orgv[4];
  {
    const Pel* o = org + (size_t)(blk.org_y + y0) * os + blk.org_x + x;
#pragma unroll
    for (int j = 0; j < 4; j++) orgv[j] = o[(size_t)j * os];
    const Pel* r0 = ref + (ptrdiff_t)(blk.ref_y - 4) * rs + blk.ref_x - 4;
#pragma unroll
    for (int u = 0; u < 9; u++)
    {
      const int i = lane + 64 * u, r = (i * 2731) >> 16, cc = i - r * 24;      // i / 24 for i < 576
      win[r * 26 + cc] = r0[(ptrdiff_t)r * rs + cc];
    }
  }
  // predictions are clipped to [cmin, cmax]: |org - pred| <= 1023 for every candidate iff org lies in [cmax - 1023, cmin + 1023]
  const bool wide = __ballot(min(min(orgv[0], orgv[1]), min(orgv[2], orgv[3])) < cmax - 1023 || max(max(orgv[0], orgv[1]), max(orgv[2], orgv[3])) > cmin + 1023) != 0ull;
  const int headRoom = max(2, 14 - bd);
  WAVE_SYNC();
  f16_hplane<false>(win, hp0, 0, 0, headRoom, lane);
  f16_hplane<true>(win, hp8, 0, 2, headRoom, lane);
  WAVE_SYNC();

  unsigned* dist = L.dist;
  F16Col col;
  int pred[4];
  // ---- half stage: quarter offsets qx, qy in {-2, 0, 2}
#pragma unroll
  for (int dx = -1; dx <= 1; dx++)
  {
    const short* pc = dx == 0 ? hp0 + ((1 + x) * HPT + y0) : hp8 + ((1 + x + (dx < 0 ? -1 : 0)) * HPT + y0);
    f16_load_col(pc, col);
    f16_vert<0>(col, 0, headRoom, cmin, cmax, pred);  dist[f16_idx(dx, 0, false)] = f16_dist<HAD>(orgv, pred, lane, wide);
    f16_vert<-1>(col, 2, headRoom, cmin, cmax, pred); dist[f16_idx(dx, -1, false)] = f16_dist<HAD>(orgv, pred, lane, wide);
    f16_vert<0>(col, 2, headRoom, cmin, cmax, pred);  dist[f16_idx(dx, 1, false)] = f16_dist<HAD>(orgv, pred, lane, wide);
  }
  int hx, hy;
  unsigned long long costH;
  unsigned distH;
  WAVE_SYNC();
  f16_best(dist, false, mv, blk.mv_x << 1, blk.mv_y << 1, 1, lane, hx, hy, costH, distH);

  // ---- quarter stage around (hx, hy): qx = 2 hx + dx, qy = 2 hy + dy
#pragma unroll
  for (int dx = -1; dx <= 1; dx++)
  {
    const int qx = 2 * hx + dx, ix = qx >> 2, fx = qx & 3;
    const short* pc;
    if (dx == 0) pc = hx == 0 ? hp0 + ((1 + x) * HPT + y0) : hp8 + ((1 + x + (hx < 0 ? -1 : 0)) * HPT + y0);
    else
    {
      WAVE_SYNC();                                          // previous readers of hpq are done
      f16_hplane<false>(win, hpq, ix, fx, headRoom, lane);
      WAVE_SYNC();
      pc = hpq + ((1 + x) * HPT + y0);
    }
    f16_load_col(pc, col);
#pragma unroll
    for (int dy = -1; dy <= 1; dy++)
    {
      const int ci = f16_idx(dx, dy, true);
      if (dx == 0 && dy == 0) { dist[ci] = distH; continue; }
      const int qy = 2 * hy + dy, iy = qy >> 2, fy = qy & 3;
      if (iy == 0) f16_vert<0>(col, fy, headRoom, cmin, cmax, pred); else f16_vert<-1>(col, fy, headRoom, cmin, cmax, pred);
      dist[ci] = f16_dist<HAD>(orgv, pred, lane, wide);
    }
  }
  int qdx, qdy;
  unsigned long long costQ;
  unsigned distQ;
  WAVE_SYNC();
  f16_best(dist, true, mv, ((blk.mv_x << 1) + hx) << 1, ((blk.mv_y << 1) + hy) << 1, 0, lane, qdx, qdy, costQ, distQ);
  if (lane == 0)
  {
    vvcgpu_frac_result r;
    r.half_x = hx; r.half_y = hy; r.qter_x = qdx; r.qter_y = qdy; r.cost_half = costH; r.cost = costQ;
    results[b] = r;
  }
  WAVE_SYNC();                                                // the next PU of this wave reuses the buffers
}

template <bool HAD>
__global__ __launch_bounds__(256) void frac16_kernel(const Pel* __restrict__ org, int os, const Pel* __restrict__ ref, int rs,
                                                     const vvcgpu_frac_blk* __restrict__ blocks, int nblocks, int bd, int cmin, int cmax,
                                                     vvcgpu_mvcost mv, const int* __restrict__ preds,
                                                     vvcgpu_frac_result* __restrict__ results, int nWg, int xcd)
{
  __shared__ __align__(16) F16Lds ldsS[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wg = vvc_xcd_index((int)blockIdx.x, nWg, xcd);
  const int b = wg * 4 + wave;
  if (wg < 0 || b >= nblocks) return;                       // no workgroup barrier below
  const vvcgpu_frac_blk blk = blocks[b];
  if (preds) { mv.pred_hor = preds[2 * b]; mv.pred_ver = preds[2 * b + 1]; }
  frac16_pu_valu<HAD>(ldsS[wave], org, os, ref, rs, blk, b, bd, cmin, cmax, mv, results, lane);
}

// the vector-pipe form of one PU as a REAL call: the matrix-core kernel below serves the PUs it cannot take (reference samples outside the bit depth, an
// original that is no picture) on the spot with it -- the separate launch that walked a flag array cost 6.4 us per 4K picture for an empty list (round 6)
__device__ __noinline__ void frac16_pu_valu_call(F16Lds* L, const Pel* __restrict__ org, int os, const Pel* __restrict__ ref, int rs,
                                                 const vvcgpu_frac_blk* __restrict__ blocks, int b, int bd, int cmin, int cmax, vvcgpu_mvcost mv, const int* __restrict__ preds,
                                                 vvcgpu_frac_result* __restrict__ results, int lane)
{
  if (preds) { mv.pred_hor = preds[2 * b]; mv.pred_ver = preds[2 * b + 1]; }
  frac16_pu_valu<true>(*L, org, os, ref, rs, blocks[b], b, bd, cmin, cmax, mv, results, lane);
}
constexpr int FM_FB_MAX = 32;                                // PUs a wave of frac16m_kernel can set aside (the launcher keeps a wave's walk within it)

// ---------------------------------------------------------------------------------------------------
// 16x16 PUs with Hadamard cost ON THE MATRIX CORES (round 5).  The vector-pipe form above runs ~2100 vector instructions per PU at an issue
// bound of 0.95 with the matrix pipe idle; here both DCTIF passes and the x direction of the 8x8 Hadamards are exact f16 products
// (v_mfma_f32_16x16x32_f16 / 16x16x16_f16) and the lane sums of the 8x8 tiles are f32 products (v_mfma_f32_16x16x4_f32), ~700 vector instructions
// per PU.  One wave per PU, a result tile is the next product's operand WITHOUT leaving the lane (the trick of mfma_tr.h): an MFMA contracts the
// index its operands hold in registers, and feeding a result tile as the A operand moves its lane index into registers.
//   window  W[r][c], r, c = 0..23 (picture rows / columns -4..19), 10-bit samples v as f16 bit patterns 0x6400 | v (= 1024 + v: no conversion)
//   stage A (horizontal pass, InterpolationFilter.cpp:290-379 with isLast false): plane[r][x] = (sum_k W[r][x + ix + 1 + k] c_fx[k] + off1) >> shift1
//           = W (A operand: lane row, registers columns) x Toeplitz matrix of the taps (table TA); the accumulator starts at the constant that
//           turns the sum into T - (S - 1) / 2, T = S (plane + 16384) + remainder, S = 2^shift1; one v_add_f32 with 2^23 S then leaves
//           u = plane + 16384 (15 bits, unsigned) in the low mantissa bits -- round-to-nearest at that exponent IS the floor, because the offset
//           -(S - 1) / 2 keeps every remainder off the ties.  u is split into limbs lo = u & 127, hi = u >> 7, again as 0x6400 | limb.
//   stage B (vertical pass, isLast true): pred^T[x][y] = plane^T (A operand: the stage-A result registers as they are) x Toeplitz matrix of the
//           taps per limb (table TB: c and 128 c, scaled by 2^-shift2; the scale is exact: powers of two).  The accumulator starts at the constant
//           that removes the biases (1024 per limb, 16384 of u), adds the rounding offset and 1024: the result is 1024 + (sum + off2) / 2^shift2
//           as a real number.  v_cvt_pkrtz_f16_f32 truncates: in [1024, 2048) f16 has unit spacing, so that IS the floor; values outside land
//           outside and the packed clamp to [1024 + clpMin, 1024 + clpMax] takes them.  All partial sums stay below 2^24 in units of 2^-shift2
//           whatever the order (largest positive part 88 x 128 x 1279 + 88 x 1151 = 14.5 M on a start value of about -8 M): exact in f32.
//   residual d = (1024 + org) - (1024 + pred), |d| <= 1023 (PUs with other originals -- bi-predictive 2 org - otherPred -- take the vector form)
//   Hadamard (RdCost.cpp:2205-2853, xCalcHADs8x8): over y one butterfly before the product (lane ^ 8, packed f16, |.| <= 2046 still exact),
//           over x as a product with blockdiag(H8, H8) (the residual as A operand: y moves into registers), the remaining two y stages in
//           registers on f32, the last of them as |a + b| + |a - b| = 2 max(|a|, |b|).
//   tile sums: p = the lane's part of sum |coefficient| / 2; R1[x'][slot] += p x selector (slot = candidate + 8 (tile row)), one product per
//           candidate into ONE accumulator for the eight candidates of a stage; at the end of the stage four registers are added, a second
//           product adds the lane groups of a tile column, SATD tile = (P + 1) >> 1 (= (2 P + 2) >> 2), and a row_ror:8 adds the tile rows:
//           lane i then holds the distortion of candidate i, which is where the arg-min wants it.
constexpr int FM_MAT = 512;                                  // halves per lane-indexed operand image (64 lanes x 8)
constexpr int FM_NA = 7, FM_NB = 14;                         // TA: (fx, ix) in {(0,0), (1,-1), (1,0), (2,-1), (2,0), (3,-1), (3,0)}; TB: the same seven (fy, iy) x two row chunks
constexpr int FM_TAB_HALVES = (FM_NA + FM_NB) * FM_MAT;
typedef _Float16 fh2 __attribute__((ext_vector_type(2)));

// result column slot n of stage B -> sample row y: slot bit 3 = y bit 2 (the partner of the lane ^ 8 butterfly), bit 2 = y bit 3 (the tile row)
__host__ __device__ constexpr int fm_ymap(int n) { return (n & 3) | (((n >> 3) & 1) << 2) | (((n >> 2) & 1) << 3); }
__host__ __device__ constexpr int fm_combo(int f, int i) { return f == 0 ? 0 : 1 + (f - 1) * 2 + (i + 1); }     // (phase f in quarter samples, integer offset i in {-1, 0})
__host__ __device__ constexpr int fm_idx(int dx, int dy, bool quarter)                                           // f16_idx as a constant expression
{
  return dx == 0 ? (dy == 0 ? 0 : dy < 0 ? 1 : 2)
       : !quarter ? (dx < 0 ? (dy == 0 ? 3 : dy < 0 ? 5 : 7) : (dy == 0 ? 4 : dy < 0 ? 6 : 8))
                  : (dx < 0 ? (dy < 0 ? 3 : dy == 0 ? 5 : 7) : (dy < 0 ? 4 : dy == 0 ? 6 : 8));
}

__global__ void fm_build_tables_kernel(_Float16* __restrict__ tab, int shift2)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= FM_TAB_HALVES) return;
  const int mat = i / FM_MAT, lane = (i % FM_MAT) >> 3, e = i & 7, c16 = lane & 15, g = lane >> 4;
  float v = 0.f;
  if (mat < FM_NA)
  {
    const int fx = mat == 0 ? 0 : 1 + ((mat - 1) >> 1), ix = mat == 0 ? 0 : ((mat - 1) & 1) - 1;
    const int t = 8 * g + e - (c16 + ix + 1);                // window column k = 8 g + e, output column x = c16
    if (g < 3 && t >= 0 && t <= 7) v = (float)c_lumaF[fx << 2][t];
    // k = 24, 25 meet the constant 1.0 the window operand carries there: the start value 8192 S - 65536 - (S - 1) / 2 of the sum, in two exact pieces
    const float S = (float)(1 << (12 - shift2));
    if (g == 3 && e == 0) v = 8192.f * S - 65536.f;
    if (g == 3 && e == 1) v = -0.5f * (S - 1.f);
  }
  else
  {
    const int m = mat - FM_NA, cb = m >> 1, ch = m & 1;
    const int fy = cb == 0 ? 0 : 1 + ((cb - 1) >> 1), iy = cb == 0 ? 0 : ((cb - 1) & 1) - 1;
    const int t = 16 * ch + 4 * g + (e & 3) - (fm_ymap(c16) + iy + 1);   // plane row 16 ch + 4 g + (e & 3), limb e >> 2
    if (t >= 0 && t <= 7) v = (float)c_lumaF[fy << 2][t] * (e >= 4 ? 128.f : 1.f) / (float)(1 << shift2);
    // rows 24, 25 of the plane do not exist: their operand slots carry 1.0, and these two the start value of the sum (bias removal + rounding offset + 1024)
    // as an f16 value and its exact remainder
    if (ch == 1 && g == 2 && e < 2)
    {
      const float cinB = (float)(-8454144 - 1048576 + (1 << (shift2 - 1)) + (OFFS << 6) + (1024 << shift2)) / (float)(1 << shift2);
      const float p1 = (float)(_Float16)cinB;
      v = e == 0 ? p1 : cinB - p1;
    }
  }
  tab[i] = (_Float16)v;
}

template <int T> __device__ __forceinline__ unsigned fm_sel(unsigned sp0)
{
  if (T == 0) return sp0;
  return (unsigned)__builtin_amdgcn_mov_dpp((int)sp0, 0x120 + (T ? T : 1), 0xF, 0xF, true);   // row_ror:T (every lane has a source: no old value to set up)
}

struct FmPu                                                 // per-PU operands that every candidate shares
{
  h8 wA[2];                                                 // window rows 16 ch + (lane & 15) (clamped to 23), columns 8 min(g, 2) .. + 7, biased
  fh2 o2[2];                                                // 1024 + original, row fm_ymap(lane & 15), columns 4 g .. 4 g + 3
};
struct FmK                                                  // wave constants (lane-varying ones in vector registers: v_and_or_b32 takes them as they are)
{
  float magicA;
  unsigned sp0;                                             // selector of slot 0 as the f16 pair (1, 2048): the weights of a sum's two limbs
  fh2 pmin, pmax, sg;
  h4 hx;
  unsigned m7[2], m8[2], orX[2], orR[2];                    // limb masks / f16 exponent patterns per row chunk: chunk 1 has no rows 24..31, its lanes g >= 2 carry constants
};

// stage A: the two row chunks of the first-stage plane `a` (TA index) as limb operands
__device__ __forceinline__ void fm_plane(const _Float16* __restrict__ tabS, int a, const FmPu& pu, const FmK& K, int lane, h8 (&pl)[2])
{
  const h8 tb = *reinterpret_cast<const h8*>(tabS + (a * 64 + lane) * 8);
#pragma unroll
  for (int ch = 0; ch < 2; ch++)
  {
    const f4 acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(pu.wA[ch], tb, f4{ 0.f, 0.f, 0.f, 0.f }, 0, 0, 0);
    unsigned u[4];
#pragma unroll
    for (int j = 0; j < 4; j++) u[j] = __builtin_bit_cast(unsigned, acc[j] + K.magicA);
    const unsigned p01 = __builtin_amdgcn_perm(u[1], u[0], 0x05040100u), p23 = __builtin_amdgcn_perm(u[3], u[2], 0x05040100u);
    uint4 o;
    o.x = (p01 & K.m7[ch]) | K.orX[ch];
    o.y = (p23 & K.m7[ch]) | K.orR[ch];
    o.z = ((p01 >> 7) & K.m8[ch]) | K.orR[ch];
    o.w = ((p23 >> 7) & K.m8[ch]) | K.orR[ch];
    pl[ch] = __builtin_bit_cast(h8, o);
  }
}

// N candidates (TB indices cb[]) from one plane, side by side: the lane's parts P of sum |Hadamard coefficient| / 2 over its tile.  The products of
// the N candidates are issued together (a candidate's second product needs the first one's result: N - 1 independent products in between hide that),
// and the vector work of the N candidates is independent.
typedef float f2v __attribute__((ext_vector_type(2)));
template <int N>
__device__ __forceinline__ void fm_cands(const _Float16* __restrict__ tabS, const int (&cb)[N], const h8 (&pl)[2], const FmPu& pu, const FmK& K, int lane, float (&P)[N])
{
  h8 b0[N], b1[N];
#pragma unroll
  for (int i = 0; i < N; i++)
  {
    b0[i] = *reinterpret_cast<const h8*>(tabS + ((FM_NA + 2 * cb[i]) * 64 + lane) * 8);
    b1[i] = *reinterpret_cast<const h8*>(tabS + ((FM_NA + 2 * cb[i] + 1) * 64 + lane) * 8);
  }
  f4 acc[N];
#pragma unroll
  for (int i = 0; i < N; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pl[0], b0[i], f4{ 0.f, 0.f, 0.f, 0.f }, 0, 0, 0);
#pragma unroll
  for (int i = 0; i < N; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pl[1], b1[i], acc[i], 0, 0, 0);
  h4 dv[N];
#pragma unroll
  for (int i = 0; i < N; i++)
  {
    fh2 p0 = __builtin_bit_cast(fh2, __builtin_amdgcn_cvt_pkrtz(acc[i][0], acc[i][1])), p1 = __builtin_bit_cast(fh2, __builtin_amdgcn_cvt_pkrtz(acc[i][2], acc[i][3]));
    p0 = __builtin_elementwise_min(__builtin_elementwise_max(p0, K.pmin), K.pmax);
    p1 = __builtin_elementwise_min(__builtin_elementwise_max(p1, K.pmin), K.pmax);
    fh2 d0 = pu.o2[0] - p0, d1 = pu.o2[1] - p1;
    const fh2 q0 = __builtin_bit_cast(fh2, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, d0), DPP_ROR8, 0xF, 0xF, true));
    const fh2 q1 = __builtin_bit_cast(fh2, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, d1), DPP_ROR8, 0xF, 0xF, true));
    d0 = __builtin_elementwise_fma(d0, K.sg, q0);
    d1 = __builtin_elementwise_fma(d1, K.sg, q1);
    dv[i] = __builtin_shufflevector(d0, d1, 0, 1, 2, 3);
  }
  f4 e[N];
#pragma unroll
  for (int i = 0; i < N; i++) e[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(dv[i], K.hx, f4{ 0.f, 0.f, 0.f, 0.f }, 0, 0, 0);
#pragma unroll
  for (int i = 0; i < N; i++)
  {
    const f2v lo = { e[i][0], e[i][1] }, hi = { e[i][2], e[i][3] };
    const f2v sm = lo + hi, df = lo - hi;
    P[i] = fmaxf(fabsf(sm[0]), fabsf(sm[1])) + fmaxf(fabsf(df[0]), fabsf(df[1]));
  }
}

// The lane's part P (an integer below 2^17 in f32) as two f16 limbs (P mod 2048, P div 2048) in one register
__device__ __forceinline__ unsigned fm_limbs(float P)
{
  const float hi = floorf(P * (1.f / 2048.f));
  const float lo = __builtin_fmaf(hi, -2048.f, P);
  return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(lo, hi));
}
// R1[x'][slot] += sum over the lane groups of up to three candidates' parts, slot = T + 8 (tile row): ONE f16 product (the f32-input MFMA this
// replaces holds the SIMD's vector pipe for its whole 32 cycles -- measured: 9 us of 72 per picture for 20 of them per PU)
template <int T0, int T1, int T2>
__device__ __forceinline__ f4 fm_acc(f4 R, float P0, float P1, float P2, const FmK& K)
{
  uint4 a, b;
  a.x = fm_limbs(P0); b.x = fm_sel<T0 & 7>(K.sp0);
  a.y = T1 >= 0 ? fm_limbs(P1) : 0u; b.y = T1 >= 0 ? fm_sel<T1 & 7>(K.sp0) : 0u;
  a.z = T2 >= 0 ? fm_limbs(P2) : 0u; b.z = T2 >= 0 ? fm_sel<T2 & 7>(K.sp0) : 0u;
  a.w = 0u; b.w = 0u;
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), R, 0, 0, 0);
}

// end of a stage: R1[x'][slot] (lane: slot, lane group g': x' = 4 g' .. 4 g' + 3 in the four registers) -> distortion of candidate (lane & 7) in every lane:
// the lane groups of a tile column meet through the LDS crossbar (no memory: ds_swizzle / ds_bpermute), SATD tile = (P + 1) >> 1, tile rows by row_ror:8
__device__ __forceinline__ float fm_finalize(const f4& R1, int lane)
{
  const float rr = (R1[0] + R1[1]) + (R1[2] + R1[3]);
  const float col = rr + __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, rr), 0x401F));       // lane ^ 16: the other half of the tile column
  const float sat = floorf((col + 1.f) * 0.5f);
  float s = sat + __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __builtin_bit_cast(int, sat)));  // lane ^ 32: the other tile column
  s += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s), DPP_ROR8, 0xF, 0xF, true));           // the other tile row
  return s;
}

// exp-Golomb length of xGetExpGolombNumberOfBits (RdCost.h:172-199) without its loop: every pass of `while (t > 128) { len += 14; t >>= 7; }` is one of
// four monotone tests
__device__ __forceinline__ unsigned fm_eg_bits(int v)
{
  const unsigned t = (v <= 0) ? ((unsigned)(-v) << 1) + 1 : (unsigned)(v << 1);
  const unsigned k = (unsigned)(t > 128u) + (unsigned)((t >> 7) > 128u) + (unsigned)((t >> 14) > 128u) + (unsigned)((t >> 21) > 128u);
  const unsigned tf = t >> (7 * k);
  return 1 + 14 * k + ((31 - __clz((int)tf)) << 1);
}
constexpr int FM_COST_N = 160;                               // two lengths of at most 71 bits each
// the candidate offsets of s_acMvRefineH / s_acMvRefineQ as 2-bit fields (offset + 1), index i at bits 2 i
constexpr unsigned FM_HX = 1u | 1u << 2 | 1u << 4 | 0u << 6 | 2u << 8 | 0u << 10 | 2u << 12 | 0u << 14 | 2u << 16;
constexpr unsigned FM_HY = 1u | 0u << 2 | 2u << 4 | 1u << 6 | 1u << 8 | 0u << 10 | 0u << 12 | 2u << 14 | 2u << 16;
constexpr unsigned FM_QX = 1u | 1u << 2 | 1u << 4 | 0u << 6 | 2u << 8 | 0u << 10 | 2u << 12 | 0u << 14 | 2u << 16;
constexpr unsigned FM_QY = 1u | 0u << 2 | 2u << 4 | 0u << 6 | 0u << 8 | 1u << 10 | 1u << 12 | 2u << 14 | 2u << 16;

// arg-min of dist + mvcost over the 9 candidates, first index on ties (xPatternRefinement's strict '<'); lane i < 9 carries candidate i's distortion.
// costS[n] = (uint64)(lambda n) (built once per workgroup).  Costs that fit 32 bits (every real one) meet in four DPP minima; otherwise the 64-bit form.
__device__ __forceinline__ void fm_best(unsigned dl, bool quarter, const unsigned long long* __restrict__ costS, double lambda, int predH, int predV, int baseX, int baseY, int scale,
                                        int lane, int& bdx, int& bdy, unsigned long long& bcost, unsigned& bdist)
{
  const int li = lane < 9 ? lane : 0;
  const int dx = (int)(((quarter ? FM_QX : FM_HX) >> (2 * li)) & 3u) - 1, dy = (int)(((quarter ? FM_QY : FM_HY) >> (2 * li)) & 3u) - 1;
  const unsigned bits = fm_eg_bits(((baseX + dx) << scale) - predH) + fm_eg_bits(((baseY + dy) << scale) - predV);
  const unsigned long long mc = bits < (unsigned)FM_COST_N ? costS[bits] : (unsigned long long)(lambda * (double)bits);
  unsigned long long c = lane < 9 ? (unsigned long long)dl + mc : ~0ull;
  int bi;
  if (__ballot(lane < 9 && (c >> 32) != 0) == 0ull)
  {
    unsigned m = (unsigned)c;                                // lanes 9..63: all ones
    m = min(m, (unsigned)__builtin_amdgcn_mov_dpp((int)m, DPP_XOR1, 0xF, 0xF, true));
    m = min(m, (unsigned)__builtin_amdgcn_mov_dpp((int)m, DPP_XOR2, 0xF, 0xF, true));
    m = min(m, (unsigned)__builtin_amdgcn_mov_dpp((int)m, DPP_HALF_MIRROR, 0xF, 0xF, true));
    m = min(m, (unsigned)__builtin_amdgcn_mov_dpp((int)m, 0x140, 0xF, 0xF, true));          // row_mirror: the minimum of the 16 lanes in each of them
    const unsigned m0 = (unsigned)__builtin_amdgcn_readfirstlane((int)m);
    bi = __builtin_ctzll(__ballot((unsigned)c == m0 && lane < 9));
    bcost = m0;
  }
  else
  {
    bi = lane;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1)
    {
      const unsigned long long oc = __shfl_xor(c, o);
      const int oi = __shfl_xor(bi, o);
      if (oc < c || (oc == c && oi < bi)) { c = oc; bi = oi; }
    }
    bi = __builtin_amdgcn_readfirstlane(bi);
    bcost = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(c >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)c);
  }
  bdx = (int)(((quarter ? FM_QX : FM_HX) >> (2 * bi)) & 3u) - 1;
  bdy = (int)(((quarter ? FM_QY : FM_HY) >> (2 * bi)) & 3u) - 1;
  bdist = (unsigned)__builtin_amdgcn_readlane((int)dl, bi);
}


// WV waves per workgroup: 16 (one workgroup per CU: the 21 KB table image is copied once per CU instead of four times -- 1024 workgroups of four waves
// fetched 22 MB of tables in the first microseconds of the launch) or 4 (short lists)
template <int WPS, int WV>
__global__ __launch_bounds__(64 * WV, WV == 16 ? 1 : WPS) void frac16m_kernel(const Pel* __restrict__ org, int os, const Pel* __restrict__ ref, int rs,
                                                      const vvcgpu_frac_blk* __restrict__ blocks, int nblocks, int bd, int cmin, int cmax,
                                                      vvcgpu_mvcost mv0, const int* __restrict__ preds,
                                                      vvcgpu_frac_result* __restrict__ results, const _Float16* __restrict__ image,
                                                      int nWg, int xcd)
{
  __shared__ __align__(16) _Float16 tabS[FM_TAB_HALVES];
  __shared__ __align__(16) F16Lds ldsV[WV];                  // the vector-pipe form's planes (PUs this kernel cannot take: frac16_pu_valu_call)
  __shared__ int fbk[WV][FM_FB_MAX];                         // ... which a wave sets aside and serves BEHIND its walk: a call inside the pipelined loop kept the
  int nFb = 0;                                               // loop's lane constants in scratch memory (73.1 against 64.8 us)
  __shared__ unsigned long long costS[FM_COST_N];
  const int wg = vvc_xcd_index((int)blockIdx.x, nWg, xcd);
  if (wg < 0) return;                                        // whole workgroup
  for (int i = threadIdx.x; i < FM_TAB_HALVES / 8; i += 64 * WV) reinterpret_cast<uint4*>(tabS)[i] = reinterpret_cast<const uint4*>(image)[i];
  if (threadIdx.x < FM_COST_N) costS[threadIdx.x] = (unsigned long long)(mv0.lambda * (double)threadIdx.x);       // RdCost.h:172-199 per bit count
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), c16 = lane & 15, g = lane >> 4;   // (the PU index is wave-uniform: descriptors through the scalar cache)
  const int headRoom = max(2, 14 - bd), shift1 = 6 - headRoom, S = 1 << shift1;
  FmK K;
  K.magicA = 8388608.f * (float)S;
  K.sp0 = c16 == 8 * (g & 1) ? 0x68003C00u : 0u;             // (1.0, 2048.0)
  K.pmin = fh2{ (_Float16)(short)(1024 + cmin), (_Float16)(short)(1024 + cmin) };
  K.pmax = fh2{ (_Float16)(short)(1024 + cmax), (_Float16)(short)(1024 + cmax) };
  K.sg = (c16 & 8) ? fh2{ (_Float16)-1.f, (_Float16)-1.f } : fh2{ (_Float16)1.f, (_Float16)1.f };
#pragma unroll
  for (int j = 0; j < 4; j++)
  {
    const int k = 4 * g + j;
    K.hx[j] = (c16 >> 3) != (k >> 3) ? (_Float16)0.f : (__popc((c16 & 7) & (k & 7)) & 1) ? (_Float16)-1.f : (_Float16)1.f;
  }
  K.m7[0] = 0x007F007Fu; K.m8[0] = 0x00FF00FFu; K.orX[0] = K.orR[0] = 0x64006400u;
  asm("" : "+v"(K.m7[0]), "+v"(K.m8[0]), "+v"(K.orX[0]));    // held in vector registers: v_and_or_b32 takes no literal
  K.orR[0] = K.orX[0];
  K.m7[1] = g < 2 ? 0x007F007Fu : 0u; K.m8[1] = g < 2 ? 0x00FF00FFu : 0u;
  K.orX[1] = g < 2 ? 0x64006400u : g == 2 ? 0x3C003C00u : 0u; K.orR[1] = g < 2 ? 0x64006400u : 0u;
  const unsigned winAnd = g == 3 ? 0u : 0xFFFFFFFFu, winOrX = g == 3 ? 0x3C003C00u : 0x64006400u, winOrR = g == 3 ? 0u : 0x64006400u;   // columns 24..31: 1.0, 1.0, 0 ..
  const unsigned rangeMask = (unsigned)((1 << bd) - 1) * 0x10001u;
  const int orgLo = max(cmax - 1023, -1024), orgHi = min(cmin + 1023, 1023);   // |org - pred| <= 1023 for every clipped prediction, and 1024 + org exact in f16
  const int yrow = fm_ymap(c16);
  const int stride = nWg * WV;

  struct Raw { uint4 w[2]; pel4 o; };                        // a PU's samples as loaded: fetched one PU ahead, behind the half stage of the PU in front
  auto fetch = [&](const vvcgpu_frac_blk& blk, Raw& r)
  {
    const Pel* r0 = ref + (ptrdiff_t)(blk.ref_y - 4) * rs + blk.ref_x - 4;
#pragma unroll
    for (int ch = 0; ch < 2; ch++)
    {
      const Pel* q = r0 + (ptrdiff_t)min(16 * ch + c16, 23) * rs + 8 * min(g, 2);
      pel8 v;
#pragma unroll
      for (int e = 0; e < 8; e++) v[e] = q[e];
      r.w[ch] = __builtin_bit_cast(uint4, v);
    }
    const Pel* o = org + (size_t)(blk.org_y + yrow) * os + blk.org_x + 4 * g;
#pragma unroll
    for (int j = 0; j < 4; j++) r.o[j] = o[j];
  };

  int b = wg * WV + wave;
  if (b >= nblocks) return;
  vvcgpu_frac_blk blk = blocks[b];
  int predHN = preds ? preds[2 * b] : mv0.pred_hor, predVN = preds ? preds[2 * b + 1] : mv0.pred_ver;
  Raw raw;
  fetch(blk, raw);
  for (; b < nblocks; b += stride)
  {
    const int bn = b + stride < nblocks ? b + stride : b;
    // the next PU's descriptor (and predictor) by VECTOR loads, every lane the same address: a scalar load in flight would turn every LDS wait of the
    // half stage -- table reads, the crossbar of the arg-min -- into an lgkmcnt(0) that also waits for it (scalar loads return out of order)
    const uint2* bq = reinterpret_cast<const uint2*>(blocks + bn);
    const uint2 bv0 = bq[0], bv1 = bq[1], bv2 = bq[2];
    const int2 pv = preds ? *reinterpret_cast<const int2*>(preds + 2 * bn) : make_int2(mv0.pred_hor, mv0.pred_ver);
    Raw rawN;
    const int predH = predHN, predV = predVN;
    FmPu pu;
    unsigned bad = 0;
#pragma unroll
    for (int ch = 0; ch < 2; ch++)
    {
      uint4 u = raw.w[ch];
      bad |= (u.x | u.y | u.z | u.w) & ~rangeMask;
      u.x = (u.x & winAnd) | winOrX; u.y = (u.y & winAnd) | winOrR; u.z = (u.z & winAnd) | winOrR; u.w = (u.w & winAnd) | winOrR;
      pu.wA[ch] = __builtin_bit_cast(h8, u);
    }
    {
      const pel4 ov = raw.o;
      const int omin = min(min((int)ov[0], (int)ov[1]), min((int)ov[2], (int)ov[3])), omax = max(max((int)ov[0], (int)ov[1]), max((int)ov[2], (int)ov[3]));
      bad |= (omin < orgLo || omax > orgHi) ? 1u : 0u;
      pu.o2[0] = fh2{ (_Float16)(short)(ov[0] + 1024), (_Float16)(short)(ov[1] + 1024) };
      pu.o2[1] = fh2{ (_Float16)(short)(ov[2] + 1024), (_Float16)(short)(ov[3] + 1024) };
    }
    vvcgpu_frac_blk blkN;
    blkN.org_x = __builtin_amdgcn_readfirstlane((int)bv0.x); blkN.org_y = __builtin_amdgcn_readfirstlane((int)bv0.y);
    blkN.ref_x = __builtin_amdgcn_readfirstlane((int)bv1.x); blkN.ref_y = __builtin_amdgcn_readfirstlane((int)bv1.y);
    blkN.mv_x = __builtin_amdgcn_readfirstlane((int)bv2.x); blkN.mv_y = __builtin_amdgcn_readfirstlane((int)bv2.y);
    predHN = __builtin_amdgcn_readfirstlane(pv.x); predVN = __builtin_amdgcn_readfirstlane(pv.y);
    const bool fallBack = __ballot(bad != 0) != 0ull;        // reference samples outside the bit depth, or an original that is not a picture:
    if (fallBack)                                            // set aside for the vector-pipe form
    {
      fetch(blkN, rawN);
      if (lane == 0) fbk[wave][nFb] = b;
      nFb++;
    }
    else
    {
      h8 pl[2];
      // ---- half stage: planes (fx, ix) = (0, 0), (2, -1), (2, 0); vertical candidates (fy, iy) = (0, 0), (2, -1), (2, 0)
      f4 Ra = { 0.f, 0.f, 0.f, 0.f }, Rb = { 0.f, 0.f, 0.f, 0.f };
      const int cbH[3] = { fm_combo(0, 0), fm_combo(2, -1), fm_combo(2, 0) };      // dy = 0, -1, +1
      float P[3];
      fm_plane(tabS, fm_combo(0, 0), pu, K, lane, pl);
      fm_cands<3>(tabS, cbH, pl, pu, K, lane, P);
      Ra = fm_acc<fm_idx(0, 0, false), fm_idx(0, -1, false), fm_idx(0, 1, false)>(Ra, P[0], P[1], P[2], K);
      fm_plane(tabS, fm_combo(2, -1), pu, K, lane, pl);
      fm_cands<3>(tabS, cbH, pl, pu, K, lane, P);
      Ra = fm_acc<fm_idx(-1, 0, false), fm_idx(-1, -1, false), fm_idx(-1, 1, false)>(Ra, P[0], P[1], P[2], K);
      fetch(blkN, rawN);                                     // the next PU's samples travel behind the rest of this one
      fm_plane(tabS, fm_combo(2, 0), pu, K, lane, pl);
      fm_cands<3>(tabS, cbH, pl, pu, K, lane, P);
      Ra = fm_acc<fm_idx(1, 0, false), fm_idx(1, -1, false), -1>(Ra, P[0], P[1], 0.f, K);
      Rb = fm_acc<fm_idx(1, 1, false), -1, -1>(Rb, P[2], 0.f, 0.f, K);                       // candidate 8: slot 0 of its own accumulator
      const float da = fm_finalize(Ra, lane), db = fm_finalize(Rb, lane);
      int hx, hy;
      unsigned long long costH;
      unsigned distH;
      fm_best((unsigned)(lane == 8 ? db : da), false, costS, mv0.lambda, predH, predV, blk.mv_x << 1, blk.mv_y << 1, 1, lane, hx, hy, costH, distH);

      // ---- quarter stage around (hx, hy): qx = 2 hx + dx, qy = 2 hy + dy; the centre is the half stage's winner
      f4 Rq = { 0.f, 0.f, 0.f, 0.f };
      const int qy0 = 2 * hy - 1, qy1 = 2 * hy, qy2 = 2 * hy + 1;
      const int cb0 = fm_combo(qy0 & 3, (qy0 & 3) ? qy0 >> 2 : 0), cb1 = fm_combo(qy1 & 3, (qy1 & 3) ? qy1 >> 2 : 0), cb2 = fm_combo(qy2 & 3, (qy2 & 3) ? qy2 >> 2 : 0);
      const int cbQ[3] = { cb0, cb1, cb2 }, cbQ2[2] = { cb0, cb2 };
      {
        const int qx = 2 * hx - 1;
        fm_plane(tabS, fm_combo(qx & 3, (qx & 3) ? qx >> 2 : 0), pu, K, lane, pl);
        fm_cands<3>(tabS, cbQ, pl, pu, K, lane, P);
        Rq = fm_acc<fm_idx(-1, -1, true), fm_idx(-1, 0, true), fm_idx(-1, 1, true)>(Rq, P[0], P[1], P[2], K);
      }
      {
        const int qx = 2 * hx;
        float P2[2];
        fm_plane(tabS, fm_combo(qx & 3, (qx & 3) ? qx >> 2 : 0), pu, K, lane, pl);
        fm_cands<2>(tabS, cbQ2, pl, pu, K, lane, P2);
        Rq = fm_acc<fm_idx(0, -1, true), fm_idx(0, 1, true), -1>(Rq, P2[0], P2[1], 0.f, K);
      }
      {
        const int qx = 2 * hx + 1;
        fm_plane(tabS, fm_combo(qx & 3, (qx & 3) ? qx >> 2 : 0), pu, K, lane, pl);
        fm_cands<3>(tabS, cbQ, pl, pu, K, lane, P);
        Rq = fm_acc<fm_idx(1, -1, true), fm_idx(1, 0, true), fm_idx(1, 1, true)>(Rq, P[0], P[1], P[2], K);
      }
      const float dq = fm_finalize(Rq, lane);                   // lane i (1..7) = candidate i, lane 8 = candidate 8 (slot 0 once more), lane 0 := the centre
      int qdx, qdy;
      unsigned long long costQ;
      unsigned distQ;
      fm_best(lane == 0 ? distH : (unsigned)dq, true, costS, mv0.lambda, predH, predV, ((blk.mv_x << 1) + hx) << 1, ((blk.mv_y << 1) + hy) << 1, 0, lane, qdx, qdy, costQ, distQ);
      if (lane == 0)
      {
        vvcgpu_frac_result r;
        r.half_x = hx; r.half_y = hy; r.qter_x = qdx; r.qter_y = qdy; r.cost_half = costH; r.cost = costQ;
        results[b] = r;
      }
    }
    blk = blkN;
    raw = rawN;
  }
  for (int i = 0; i < nFb; i++)
    frac16_pu_valu_call(&ldsV[wave], org, os, ref, rs, blocks, __builtin_amdgcn_readfirstlane(fbk[wave][i]), bd, cmin, cmax, mv0, preds, results, lane);
}

// TA / TB images per device and bit depth, built on first use
const _Float16* fm_image(int bd)
{
  static std::mutex mtx;
  static _Float16* images[64][3] = { { nullptr } };
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { vvcgpu_set_error("frac image: device index"); return nullptr; }
  std::lock_guard<std::mutex> lock(mtx);
  _Float16*& slot = images[dev][bd - 8];
  if (!slot)
  {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, FM_TAB_HALVES * sizeof(_Float16));
    if (e != hipSuccess) { (void)hipGetLastError(); vvcgpu_set_error("frac image: hipMalloc failed: %s", hipGetErrorString(e)); return nullptr; }
    hipLaunchKernelGGL(fm_build_tables_kernel, dim3(cdiv(FM_TAB_HALVES, 256)), dim3(256), 0, (hipStream_t)0, static_cast<_Float16*>(p), 6 + (14 - bd > 2 ? 14 - bd : 2));
    e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();         // other streams may use the image right after this call returns
    if (e != hipSuccess) { (void)hipFree(p); vvcgpu_set_error("building the fractional-search table image failed: %s", hipGetErrorString(e)); return nullptr; }
    slot = static_cast<_Float16*>(p);
  }
  return slot;
}
}  // namespace

__attribute__((visibility("hidden"))) int vvcgpu_frac_image_build(int bit_depth) { return fm_image(bit_depth) ? VVCGPU_OK : VVCGPU_E_DEVICE; }

// shared by vvcgpu_frac_refine and vvcgpu_me_batch (tzsearch.hip); preds: optional per-block MV predictors (hor, ver) on the device
int vvcgpu_frac_refine_launch(const vvc_pel* org, int org_stride, const vvc_pel* ref, int ref_stride,
                              const vvcgpu_frac_blk* blocks, int nblocks, int w, int h, int bit_depth, int clp_min,
                              int clp_max, int use_hadamard, const vvcgpu_mvcost* mvcost_host, const int* preds,
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
  if (w == 16 && h == 16)
  {
    const int xcd = vvc_xcd_on();
    if (use_hadamard && !vvcgpu_no_mfma())                                  // (VVCGPU_NO_MFMA, common.h: the vector-pipe form for every PU)
    {
      const _Float16* image = fm_image(bit_depth);
      if (!image) return VVCGPU_E_DEVICE;
      const int cap = 256 * 4;                                             // sixteen waves per CU (the kernel is built for four waves per SIMD: five spill); a wave walks its PUs
      // (lists beyond 4096 waves x FM_FB_MAX PUs -- four 4K pictures -- go as several launches: a wave sets aside at most FM_FB_MAX PUs)
      for (int first = 0; first < nblocks; first += cap * 4 * FM_FB_MAX)
      {
        const int nb = nblocks - first < cap * 4 * FM_FB_MAX ? nblocks - first : cap * 4 * FM_FB_MAX;
        if (nb >= cap * 4)                                                 // every CU gets sixteen waves: as ONE workgroup per CU
        {
          const int nWg = cap / 4;
          hipLaunchKernelGGL((frac16m_kernel<4, 16>), dim3(vvc_xcd_grid(nWg, xcd)), dim3(1024), 0, st, org, org_stride, ref, ref_stride, blocks + first, nb,
                             bit_depth, clp_min, clp_max, *mvcost_host, preds ? preds + 2 * first : nullptr, results + first, image, nWg, xcd);
        }
        else
        {
          const int nWg = cdiv(nb, 4);
          hipLaunchKernelGGL((frac16m_kernel<4, 4>), dim3(vvc_xcd_grid(nWg, xcd)), dim3(256), 0, st, org, org_stride, ref, ref_stride, blocks + first, nb,
                             bit_depth, clp_min, clp_max, *mvcost_host, preds ? preds + 2 * first : nullptr, results + first, image, nWg, xcd);
        }
      }
    }
    else if (use_hadamard)
      hipLaunchKernelGGL(frac16_kernel<true>, dim3(vvc_xcd_grid(cdiv(nblocks, 4), xcd)), dim3(256), 0, st, org, org_stride, ref, ref_stride, blocks, nblocks,
                         bit_depth, clp_min, clp_max, *mvcost_host, preds, results, cdiv(nblocks, 4), xcd);
    else
      hipLaunchKernelGGL(frac16_kernel<false>, dim3(vvc_xcd_grid(cdiv(nblocks, 4), xcd)), dim3(256), 0, st, org, org_stride, ref, ref_stride, blocks, nblocks,
                         bit_depth, clp_min, clp_max, *mvcost_host, preds, results, cdiv(nblocks, 4), xcd);
    VVC_LAUNCH_CHECK();
    return VVCGPU_OK;
  }
  if (smem > 64 * 1024)
    VVC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(frac_refine_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL(frac_refine_kernel, dim3(cdiv(nblocks, groups)), dim3(256), smem, st, org, org_stride, ref, ref_stride, blocks,
                     nblocks, w, h, bit_depth, clp_min, clp_max, use_hadamard, *mvcost_host, preds, groups, (int)groupBytes, results);
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}

extern "C" int vvcgpu_frac_refine(const vvc_pel* org, int org_stride, const vvc_pel* ref, int ref_stride,
                                  const vvcgpu_frac_blk* blocks, int nblocks, int w, int h, int bit_depth, int clp_min,
                                  int clp_max, int use_hadamard, const vvcgpu_mvcost* mvcost_host,
                                  vvcgpu_frac_result* results, void* stream)
{
  return vvcgpu_frac_refine_launch(org, org_stride, ref, ref_stride, blocks, nblocks, w, h, bit_depth, clp_min, clp_max, use_hadamard,
                                   mvcost_host, nullptr, results, stream);
}
