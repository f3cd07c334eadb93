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

template <bool HAD>
__global__ __launch_bounds__(256) void frac16_kernel(const Pel* __restrict__ org, int os, const Pel* __restrict__ ref, int rs,
                                                     const vvcgpu_frac_blk* __restrict__ blocks, int nblocks, int bd, int cmin, int cmax,
                                                     vvcgpu_mvcost mv, const int* __restrict__ preds,
                                                     vvcgpu_frac_result* __restrict__ results, int nWg, int xcd)
{
  __shared__ __align__(16) short winS[4][24 * 26];
  __shared__ __align__(16) short hplS[4][3][24 * 18];      // [0] integer plane, [1] half plane (17 cols), [2] quarter planes
  __shared__ unsigned distS[4][16];                        // candidate distortions of the current stage (wave-uniform values)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wg = vvc_xcd_index((int)blockIdx.x, nWg, xcd);
  const int b = wg * 4 + wave;
  if (wg < 0 || b >= nblocks) return;                       // no workgroup barrier below
  short* win = winS[wave];
  short* hp0 = hplS[wave][0];
  short* hp8 = hplS[wave][1];
  short* hpq = hplS[wave][2];
  const vvcgpu_frac_blk blk = blocks[b];
  if (preds) { mv.pred_hor = preds[2 * b]; mv.pred_ver = preds[2 * b + 1]; }
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

  unsigned* dist = distS[wave];
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
}

}  // namespace

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
  static const int f16Off = getenv("VVCGPU_NO_FRAC16") ? 1 : 0;           // A/B timing switch
  if (w == 16 && h == 16 && !f16Off)
  {
    const int xcd = vvc_xcd_on();
    if (use_hadamard)
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
