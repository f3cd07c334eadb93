// sao.hip -- SAO apply (S1) for gfx950.
//
// Reference behaviour reproduced (bit-exact): SampleAdaptiveOffset::offsetBlock / offsetCTU / SAOProcess
// (CommonLib/SampleAdaptiveOffset.cpp:292-612).  The reference's running sign buffers are a CPU
// optimisation; per sample the result is  clip(c + offset[2 + sgn(c-a) + sgn(c-b)])  when both neighbours
// are available (availability of a neighbour in an adjacent CTU = that CTU-direction's flag), c otherwise.
//
// Design: pure streaming stencil.  One thread = 8 horizontally adjacent samples (one 16-byte load/store)
// x 4 rows; its six rows (one above, one below) are loaded up front into a register window, the two horizontal halo
// samples per row with 2-byte loads that hit L1/L2.  dst receives the complete picture (offset or copied
// samples), so the reference's whole-picture temp copy (:587) never touches HBM.
#include "common.h"

namespace {

// rows per thread.  4 rows (6 loaded) made 6112 waves of 94 registers for a 4K picture: ONE round of five waves per SIMD, in which every wave loads, then
// computes, then stores at the same time (19.2 us); 2 rows (4 loaded) are 12 k smaller waves in two to three rounds whose phases overlap: 17.5 us.  1 row: 19.7.
constexpr int SAO_ROWS = 2;

struct SaoGeom { int gx, gy, rows, x0, y0, x1, y1, w, h, avail, clpMin, clpMax; };

// Edge-offset body with the class direction as a compile-time constant (keeps the 3-row register window statically indexed -> no scratch).
// Neighbour a = (x+DXA, y+DYA), b = (x-DXA, y-DYA).
// Which samples take an offset depends on the thread only through a few bits: neighbour a (b) of column k lies left of / inside / right of the
// CTU -- three 8-bit column masks per thread -- and above / inside / below it per row.  They are folded into one 8-bit "use" mask per row before
// the sample loop, which is then: two differences, two v_med3 (sign), the offset by a bit-field extract from the five offsets packed as 6-bit
// fields, add, clip, and a v_bfi that keeps the sample where the mask bit is clear.  (The form this replaces evaluated the position tests and a
// compare-select sign per sample: ~40 vector instructions per sample, the kernel's arithmetic took as long as its memory traffic.)
// Offsets outside [-32, 31] (not reachable from the reference at 8 - 10 bits, but the ABI takes any int16) go through a select chain instead.
template <int DXA, int DYA, typename StoreRow>
__device__ __forceinline__ void eo_rows(const SaoGeom& g, int off0, int off1, int off2, int off3, int off4,
                                        const int (&win)[SAO_ROWS + 2][10], StoreRow store_row)
{
  // 0 / -1 masks of the neighbour CTUs' availability
  const int avL = -(g.avail & 1), avR = -((g.avail >> 1) & 1), avA = -((g.avail >> 2) & 1), avB = -((g.avail >> 3) & 1);
  const int avAL = -((g.avail >> 4) & 1), avAR = -((g.avail >> 5) & 1), avBL = -((g.avail >> 6) & 1), avBR = -((g.avail >> 7) & 1);
  // columns k whose neighbour a (b) lies left of / inside / right of the CTU
  int aL = 0, aR = 0, bL = 0, bR = 0;
#pragma unroll
  for (int k = 0; k < 8; k++)
  {
    const int x = g.gx + k;
    if (DXA != 0)
    {
      if (x + DXA < g.x0) aL |= 1 << k; else if (x + DXA >= g.x1) aR |= 1 << k;
      if (x - DXA < g.x0) bL |= 1 << k; else if (x - DXA >= g.x1) bR |= 1 << k;
    }
  }
  const int aIn = 0xFF & ~(aL | aR), bIn = 0xFF & ~(bL | bR);
  const bool fits = (unsigned)(off0 + 32) < 64u && (unsigned)(off1 + 32) < 64u && (unsigned)(off2 + 32) < 64u && (unsigned)(off3 + 32) < 64u && (unsigned)(off4 + 32) < 64u;
  const unsigned packed = (unsigned)(off0 & 63) | (unsigned)(off1 & 63) << 6 | (unsigned)(off2 & 63) << 12 | (unsigned)(off3 & 63) << 18 | (unsigned)(off4 & 63) << 24;
#pragma unroll
  for (int r = 0; r < SAO_ROWS; r++)
  {
    if (r >= g.rows) break;
    const int y = g.gy + r;
    const bool topOut = (DYA != 0) && (y - 1 < g.y0);     // a's row is in the CTU above
    const bool botOut = (DYA != 0) && (y + 1 >= g.y1);    // b's row is in the CTU below
    const int okA = topOut ? ((aIn & avA) | (aL & avAL) | (aR & avAR)) : (aIn | (aL & avL) | (aR & avR));
    const int okB = botOut ? ((bIn & avB) | (bL & avBL) | (bR & avBR)) : (bIn | (bL & avL) | (bR & avR));
    const int use = okA & okB;
    int o[8];
#pragma unroll
    for (int k = 0; k < 8; k++)
    {
      const int c = win[r + 1][1 + k];
      const int a = DYA == 0 ? win[r + 1][1 + k + DXA] : win[r][1 + k + DXA];
      const int b = DYA == 0 ? win[r + 1][1 + k - DXA] : win[r + 2][1 + k - DXA];
      int sa, sb;
      asm("v_med3_i32 %0, %1, -1, 1" : "=v"(sa) : "v"(c - a));
      asm("v_med3_i32 %0, %1, -1, 1" : "=v"(sb) : "v"(c - b));
      const int e = sa + sb;
      int of;
      if (fits) of = __builtin_amdgcn_sbfe((int)packed, (unsigned)(e * 6 + 12), 6u);
      else      of = e == -2 ? off0 : e == -1 ? off1 : e == 0 ? off2 : e == 1 ? off3 : off4;
      const int t = clip3(g.clpMin, g.clpMax, c + of);
      const int m = __builtin_amdgcn_sbfe(use, (unsigned)k, 1u);            // -1: this sample takes the offset
      o[k] = (t & m) | (c & ~m);
    }
    store_row(y, o);
  }
}

__device__ __forceinline__ void sao_apply_body(const int bidx, const Pel* __restrict__ src, int sstride,
                                                        Pel* __restrict__ dst, int dstride, int w, int h,
                                                        int ctuW, int ctuH, int wCtu, int boShift,
                                                        const vvcgpu_sao_ctu* __restrict__ params,
                                                        int clpMin, int clpMax, int tpcShift, int nWaveCols)
{
  // a wave covers (8 << tpcShift) samples x (64 >> tpcShift) thread rows: with a power-of-two CTU width that is exactly
  // one CTU column, so the SAO type is wave-uniform and only one of the five type bodies runs (a wave that straddles
  // several CTUs executes them one after the other)
  const int wv = bidx * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wcol = wv % nWaveCols, wrow = wv / nWaveCols;
  const int gx = ((wcol << tpcShift) + (lane & ((1 << tpcShift) - 1))) * 8;           // first sample of this thread
  const int gy = (wrow * (64 >> tpcShift) + (lane >> tpcShift)) * SAO_ROWS;           // first row
  if (gx >= w || gy >= h) return;
  const int n = min(8, w - gx);                                 // valid samples (w need not be a multiple of 8)
  const int cx = gx / ctuW, cy = gy / ctuH;
  const vvcgpu_sao_ctu* prm = params + cy * wCtu + cx;
  const int type = prm->type;
  const int avail = prm->avail;
  const int x0 = cx * ctuW, y0 = cy * ctuH;
  const int x1 = min(x0 + ctuW, w), y1 = min(y0 + ctuH, h);
  const bool vec = (n == 8) && ((sstride & 7) == 0) && ((dstride & 7) == 0) &&
                   ((reinterpret_cast<uintptr_t>(src) & 15) == 0) && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0);

  // all SAO_ROWS + 2 rows of the thread (gy-1 .. gy+SAO_ROWS, clamped to the picture) are loaded up front: one exposure to
  // memory latency instead of one per row (a chroma plane is only ~1 workgroup per CU, nothing else hides it)
  int win[SAO_ROWS + 2][10];                        // [row][samples gx-1 .. gx+8]
#pragma unroll
  for (int i = 0; i < SAO_ROWS + 2; i++)
  {
    const int y = min(max(gy - 1 + i, 0), h - 1);
    const Pel* row = src + (size_t)y * sstride;
    if (vec)
    {
      const pel8 v = *reinterpret_cast<const pel8*>(row + gx);
#pragma unroll
      for (int k = 0; k < 8; k++) win[i][1 + k] = v[k];
    }
    else
    {
#pragma unroll
      for (int k = 0; k < 8; k++) win[i][1 + k] = row[min(gx + k, w - 1)];
    }
    win[i][0] = row[max(gx - 1, 0)];
    win[i][9] = row[min(gx + 8, w - 1)];
  }
  auto store_row = [&](int y, const int* o) {
    Pel* row = dst + (size_t)y * dstride;
    if (vec)
    {
      pel8 v;
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = (short)o[k];
      *reinterpret_cast<pel8*>(row + gx) = v;
    }
    else
    {
#pragma unroll
      for (int k = 0; k < 8; k++) if (k < n) row[gx + k] = (short)o[k];
    }
  };

  const int rows = min(SAO_ROWS, h - gy);
  if (type < 0)
  {
#pragma unroll
    for (int r = 0; r < SAO_ROWS; r++) if (r < rows) store_row(gy + r, &win[r + 1][1]);
    return;
  }
  if (type == 4)
  {
    const int16_t* off = prm->offset;
#pragma unroll
    for (int r = 0; r < SAO_ROWS; r++)
    {
      if (r >= rows) break;
      int o[8];
#pragma unroll
      for (int k = 0; k < 8; k++) o[k] = clip3(clpMin, clpMax, win[r + 1][1 + k] + off[win[r + 1][1 + k] >> boShift]);
      store_row(gy + r, o);
    }
    return;
  }
  const int16_t* op = prm->offset;
  SaoGeom g{ gx, gy, rows, x0, y0, x1, y1, w, h, avail, clpMin, clpMax };
  const int off0 = op[0], off1 = op[1], off2 = op[2], off3 = op[3], off4 = op[4];
  switch (type)
  {
  case 0:  eo_rows<-1, 0>(g, off0, off1, off2, off3, off4, win, store_row); break;
  case 1:  eo_rows<0, -1>(g, off0, off1, off2, off3, off4, win, store_row); break;
  case 2:  eo_rows<-1, -1>(g, off0, off1, off2, off3, off4, win, store_row); break;
  default: eo_rows<1, -1>(g, off0, off1, off2, off3, off4, win, store_row); break;
  }
}

__global__ __launch_bounds__(256) void sao_apply_kernel(const Pel* __restrict__ src, int sstride, Pel* __restrict__ dst, int dstride, int w, int h,
                                                        int ctuW, int ctuH, int wCtu, int boShift, const vvcgpu_sao_ctu* __restrict__ params,
                                                        int clpMin, int clpMax, int tpcShift, int nWaveCols)
{
  sao_apply_body((int)blockIdx.x, src, sstride, dst, dstride, w, h, ctuW, ctuH, wCtu, boShift, params, clpMin, clpMax, tpcShift, nWaveCols);
}

// the three planes of a picture in one launch (a chroma plane alone is ~1 workgroup per CU: its own launch costs a latency floor)
struct SaoPlaneArgs { const Pel* src; Pel* dst; const vvcgpu_sao_ctu* params; int sstride, dstride, w, h, ctuW, ctuH, wCtu, tpcShift, nWaveCols, wgEnd; };
struct SaoApply3 { SaoPlaneArgs a[3]; int boShift, clpMin, clpMax, total, xcd; };
__global__ __launch_bounds__(256) void sao_apply_picture_kernel(SaoApply3 p)
{
  const int b = vvc_xcd_index2((int)blockIdx.x, p.a[0].wgEnd, p.total, p.xcd);
  if (b < 0) return;
  const int c = b < p.a[0].wgEnd ? 0 : b < p.a[1].wgEnd ? 1 : 2;
  const SaoPlaneArgs& a = c == 0 ? p.a[0] : c == 1 ? p.a[1] : p.a[2];
  const int first = c == 0 ? 0 : c == 1 ? p.a[0].wgEnd : p.a[1].wgEnd;
  sao_apply_body(b - first, a.src, a.sstride, a.dst, a.dstride, a.w, a.h, a.ctuW, a.ctuH, a.wCtu, p.boShift, a.params, p.clpMin, p.clpMax, a.tpcShift,
                 a.nWaveCols);
}


// ---------------------------------------------------------------------------------------------------
// Strip form of the picture entry (round 6).  The form above re-reads: a thread's two output rows need four row loads and eight 2-byte halo
// loads (six load instructions per output row), and its 12 k short waves run their load / compute / store phases in lockstep.  Here a WAVE
// owns a tile of up to 128 columns (at most a CTU's width) and walks DOWN it: lane (l, g) = columns 8 l .. 8 l + 7 of the band of SS_RB rows g, in two
// steps with the second step's rows in flight; a row is loaded once (the row above / below a band once more: SS_RB + 2 loads for SS_RB rows), the left / right
// neighbour samples of a lane come from the neighbouring LANE (DPP wave shift; the two edge lanes of the tile load theirs), and the arithmetic
// works on sample pairs (v_pk_*_i16: sign = clamp(c - n, -1, 1), the five offsets as a byte table of v_perm_b32): two load instructions and
// ~70 vector instructions per row of eight samples.  The SAO type is a per-lane value (a chroma tile spans two CTUs): the row bodies branch on it.  Results identical to the form above (tests/test_gpu_inloop.py, the workload tests).
// ---------------------------------------------------------------------------------------------------
#ifndef SS_RB_D
#define SS_RB_D 8
#define SS_CH_D 5
#endif
constexpr int SS_RB = SS_RB_D, SS_CH = SS_CH_D, SS_NCH = (SS_RB + 2) / SS_CH;           // rows of a lane's band; rows per step; steps (the band + one row above and below)
static_assert(SS_NCH * SS_CH == SS_RB + 2 && SS_CH >= 3, "whole steps; the first step holds the row above the band and two of its rows");
struct SsRow { unsigned v[4]; unsigned l, r; };                               // eight samples; l: sample -1 in its HIGH half, r: sample 8 in its LOW half
struct SsRaw { uint4 v[SS_CH]; unsigned e[SS_CH]; };                          // a step's rows as loaded (+ the edge lane's neighbour sample)
struct SsLane
{
  const Pel* src; Pel* dst; int sstride, dstride;
  int gx, gxc, ex, y0, yEnd, h;                                               // first column (clamped for loads), edge sample column, band rows
  bool isL, isR, live;
  int x0, x1, cy0, cy1, avail;                                                // the lane's CTU
  unsigned lo32, hi32;                                                        // packed clip bounds + 32
};
__device__ __forceinline__ unsigned ss_pk_sign(unsigned a, unsigned b)
{
  unsigned r;
  asm("v_pk_sub_i16 %0, %1, %2\n\tv_pk_max_i16 %0, %0, -1 op_sel_hi:[1,0]\n\tv_pk_min_i16 %0, %0, 1 op_sel_hi:[1,0]" : "=&v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ unsigned ss_pk_add(unsigned a, unsigned b) { unsigned r; asm("v_pk_add_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ unsigned ss_pk_max(unsigned a, unsigned b) { unsigned r; asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ unsigned ss_pk_min(unsigned a, unsigned b) { unsigned r; asm("v_pk_min_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ unsigned ss_pk_sub32(unsigned a) { unsigned r; asm("v_pk_sub_i16 %0, %1, 32 op_sel_hi:[1,0]" : "=v"(r) : "v"(a)); return r; }
__device__ __forceinline__ unsigned ss_pk_add2(unsigned a) { unsigned r; asm("v_pk_add_i16 %0, %1, 2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a)); return r; }

__device__ __forceinline__ void ss_load(const SsLane& L, int q, SsRaw& raw)
{
#pragma unroll
  for (int i = 0; i < SS_CH; i++)
  {
    const Pel* row = L.src + (size_t)min(max(q + i, 0), L.h - 1) * L.sstride;
    raw.v[i] = *reinterpret_cast<const uint4*>(row + L.gxc);
    raw.e[i] = (unsigned)(unsigned short)row[L.ex];
  }
}
__device__ __forceinline__ void ss_row(const SsLane& L, const SsRaw& raw, int i, SsRow& r)
{
  r.v[0] = raw.v[i].x; r.v[1] = raw.v[i].y; r.v[2] = raw.v[i].z; r.v[3] = raw.v[i].w;
  // the neighbour lanes' samples: every lane takes part in the shifts (wave_shr:1 / wave_shl:1), the tile's edge lanes use what they loaded
  const unsigned fromL = (unsigned)__builtin_amdgcn_update_dpp(0, (int)r.v[3], 0x138, 0xF, 0xF, false);
  const unsigned fromR = (unsigned)__builtin_amdgcn_update_dpp(0, (int)r.v[0], 0x130, 0xF, 0xF, false);
  r.l = L.isL ? raw.e[i] << 16 : fromL;
  r.r = L.isR ? raw.e[i] : fromR;
}
__device__ __forceinline__ unsigned ss_left(const SsRow& r, int j) { return __builtin_amdgcn_alignbit(r.v[j], j == 0 ? r.l : r.v[j > 0 ? j - 1 : 0], 16); }     // samples (2j - 1, 2j)
__device__ __forceinline__ unsigned ss_right(const SsRow& r, int j) { return __builtin_amdgcn_alignbit(j == 3 ? r.r : r.v[j < 3 ? j + 1 : 3], r.v[j], 16); }    // samples (2j + 1, 2j + 2)
__device__ __forceinline__ void ss_store(const SsLane& L, int y, const unsigned (&o)[4])
{
  if (L.live && y >= L.y0 && y < L.yEnd) *reinterpret_cast<uint4*>(L.dst + (size_t)y * L.dstride + L.gx) = make_uint4(o[0], o[1], o[2], o[3]);
}
// bit k of `use` -> the 16-bit half of sample k
__device__ __forceinline__ void ss_expand(int use, unsigned (&m)[4])
{
#pragma unroll
  for (int j = 0; j < 4; j++) m[j] = (unsigned)__builtin_amdgcn_sbfe(use, 2 * j, 1) & 0xFFFFu | (unsigned)__builtin_amdgcn_sbfe(use, 2 * j + 1, 1) << 16;
}

// TYPE 0..3: edge offset with neighbours a = (x + DXA, y + DYA), b = (x - DXA, y - DYA); 4: band offset; -1: copy
struct SsEo { int aL, aR, bL, bR, aIn, bIn; unsigned tLo, tHi; unsigned mI[4]; bool fits; int off[5]; };
template <int DXA, int DYA>
__device__ __forceinline__ int ss_use(const SsLane& L, const SsEo& E, int y)
{
  const int avL = -(L.avail & 1), avR = -((L.avail >> 1) & 1), avA = -((L.avail >> 2) & 1), avB = -((L.avail >> 3) & 1);
  const int avAL = -((L.avail >> 4) & 1), avAR = -((L.avail >> 5) & 1), avBL = -((L.avail >> 6) & 1), avBR = -((L.avail >> 7) & 1);
  const bool topOut = (DYA != 0) && (y - 1 < L.cy0), botOut = (DYA != 0) && (y + 1 >= L.cy1);
  const int okA = topOut ? ((E.aIn & avA) | (E.aL & avAL) | (E.aR & avAR)) : (E.aIn | (E.aL & avL) | (E.aR & avR));
  const int okB = botOut ? ((E.bIn & avB) | (E.bL & avBL) | (E.bR & avBR)) : (E.bIn | (E.bL & avL) | (E.bR & avR));
  return okA & okB;
}
template <int DXA, int DYA>
__device__ __forceinline__ void ss_eo_setup(const SsLane& L, const vvcgpu_sao_ctu* prm, SsEo& E)
{
  E.aL = E.aR = E.bL = E.bR = 0;
#pragma unroll
  for (int k = 0; k < 8; k++)
  {
    const int x = L.gx + k;
    if (DXA != 0)
    {
      if (x + DXA < L.x0) E.aL |= 1 << k; else if (x + DXA >= L.x1) E.aR |= 1 << k;
      if (x - DXA < L.x0) E.bL |= 1 << k; else if (x - DXA >= L.x1) E.bR |= 1 << k;
    }
  }
  E.aIn = 0xFF & ~(E.aL | E.aR); E.bIn = 0xFF & ~(E.bL | E.bR);
#pragma unroll
  for (int c = 0; c < 5; c++) E.off[c] = prm->offset[c];
  E.fits = true;
#pragma unroll
  for (int c = 0; c < 5; c++) E.fits = E.fits && (unsigned)(E.off[c] + 32) < 64u;
  E.tLo = (unsigned)(E.off[0] + 32) | (unsigned)(E.off[1] + 32) << 8 | (unsigned)(E.off[2] + 32) << 16 | (unsigned)(E.off[3] + 32) << 24;
  E.tHi = (unsigned)(E.off[4] + 32);
  ss_expand(ss_use<DXA, DYA>(L, E, L.cy0 + 1), E.mI);                           // a row with both vertical neighbours inside the CTU (a CTU has >= 16 rows)
}
template <int DXA, int DYA, bool FITS>
__device__ __forceinline__ void ss_eo_row(const SsLane& L, const SsEo& E, const SsRow& up, const SsRow& mid, const SsRow& dn, int y, bool edgeRow)
{
  unsigned m[4];
  if (edgeRow) ss_expand(ss_use<DXA, DYA>(L, E, y), m);                        // the first / last row of a band may be the first / last row of its CTU
  else { m[0] = E.mI[0]; m[1] = E.mI[1]; m[2] = E.mI[2]; m[3] = E.mI[3]; }
  const SsRow& ra = DYA == 0 ? mid : up;                                        // a's row; b's row is the opposite one
  const SsRow& rb = DYA == 0 ? mid : dn;
  unsigned o[4];
#pragma unroll
  for (int j = 0; j < 4; j++)
  {
    const unsigned c = mid.v[j];
    const unsigned a = DXA < 0 ? ss_left(ra, j) : DXA > 0 ? ss_right(ra, j) : ra.v[j];
    const unsigned b = DXA < 0 ? ss_right(rb, j) : DXA > 0 ? ss_left(rb, j) : rb.v[j];
    const unsigned e = ss_pk_add(ss_pk_sign(c, a), ss_pk_sign(c, b));          // -2..2 per half
    unsigned t;
    if (FITS)
    {
      const unsigned tb = __builtin_amdgcn_perm(E.tHi, E.tLo, ss_pk_add2(e) | 0x0C000C00u);     // offset + 32 per half
      t = ss_pk_sub32(ss_pk_min(ss_pk_max(ss_pk_add(c, tb), L.lo32), L.hi32));
    }
    else
    {
      // offsets outside [-32, 31] (the ABI takes any int16; the reference does not produce them at 8 - 10 bits): a select chain per sample
      int r2[2];
#pragma unroll
      for (int hh = 0; hh < 2; hh++)
      {
        const int cc = (int)((c >> (16 * hh)) & 0xFFFFu), ee = (int)(short)(e >> (16 * hh));
        const int of = ee == -2 ? E.off[0] : ee == -1 ? E.off[1] : ee == 0 ? E.off[2] : ee == 1 ? E.off[3] : E.off[4];
        r2[hh] = min(max(cc + of, (int)(short)(L.lo32 & 0xFFFFu) - 32), (int)(short)(L.hi32 & 0xFFFFu) - 32);
      }
      t = (unsigned)r2[0] & 0xFFFFu | (unsigned)r2[1] << 16;
    }
    o[j] = (t & m[j]) | (c & ~m[j]);
  }
  ss_store(L, y, o);
}
// band offset: offset[sample >> boShift] per sample.  TAB: the offsets of the wave's CTU as a table in LDS (the uniform walk: a look-up in memory behind
// the previous row's store would wait for that store as well -- loads and stores share one counter), else read through the parameter record
template <bool TAB>
__device__ __forceinline__ void ss_bo_row(const SsLane& L, const int16_t* __restrict__ off, const unsigned short* tab, int boShift, const SsRow& mid, int y)
{
  unsigned o[4];
#pragma unroll
  for (int j = 0; j < 4; j++)
  {
    const unsigned c = mid.v[j];
    const unsigned b0 = (c & 0xFFFFu) >> boShift, b1 = c >> (16 + boShift);
    const unsigned tb = TAB ? (unsigned)tab[b0 & 31u] | (unsigned)tab[b1 & 31u] << 16 : (unsigned)(unsigned short)off[b0] | (unsigned)(unsigned short)off[b1] << 16;
    o[j] = ss_pk_sub32(ss_pk_min(ss_pk_max(ss_pk_add(ss_pk_add(c, tb), 0x00200020u), L.lo32), L.hi32));
  }
  ss_store(L, y, o);
}

// the walk of one lane's band; Body(up, mid, dn, y, edgeRow) produces row y.  ONE copy of the step's row bodies (the loop moves the prefetched rows
// into the registers of the current step): six type walks of unrolled double-buffered steps were ~40 KB of hot code on a CU whose waves run different types
template <typename Body>
__device__ __forceinline__ void ss_walk(const SsLane& L, SsRaw& A, Body body)   // A: the first step's rows, requested by the caller
{
  SsRaw B;
  SsRow up, mid, dn;
#pragma unroll 1
  for (int s = 0; s < SS_NCH; s++)
  {
    const int q = L.y0 - 1 + s * SS_CH;
    if (s + 1 < SS_NCH) ss_load(L, q + SS_CH, B);                              // the next step's rows travel while this step is computed
#pragma unroll
    for (int i = 0; i < SS_CH; i++)
    {
      up = mid; mid = dn; ss_row(L, A, i, dn);
      const int y = q + i - 1;                                                 // dn = row q + i: row q + i - 1 has both neighbours now
      if (s > 0 || i >= 2) body(up, mid, dn, y, (s == 0 && i == 2) || (s == SS_NCH - 1 && i == SS_CH - 1));
    }
#pragma unroll
    for (int i = 0; i < SS_CH; i++) { A.v[i] = B.v[i]; A.e[i] = B.e[i]; }
  }
}

struct SaoStripPlane { const Pel* src; Pel* dst; const vvcgpu_sao_ctu* params; int sstride, dstride, w, h, ctuW, ctuH, wCtu, tilesX, tileEnd, lxShift; };   // a tile: 8 << lxShift columns x (64 >> lxShift) SS_RB rows
struct SaoStrip3 { SaoStripPlane a[3]; int boShift, clpMin, clpMax, total, xcd; };
__global__ __launch_bounds__(64) void sao_apply_strip_kernel(SaoStrip3 p)
{
  const int b = vvc_xcd_index((int)blockIdx.x, p.total, p.xcd);
  if (b < 0) return;
  const int c = b < p.a[0].tileEnd ? 0 : b < p.a[1].tileEnd ? 1 : 2;
  const SaoStripPlane& a = c == 0 ? p.a[0] : c == 1 ? p.a[1] : p.a[2];
  const int tile = b - (c == 0 ? 0 : c == 1 ? p.a[0].tileEnd : p.a[1].tileEnd);
  const int ty = tile / a.tilesX, tx = tile - ty * a.tilesX;
  const int lane = threadIdx.x, lx = 1 << a.lxShift, l = lane & (lx - 1), g = lane >> a.lxShift;
  SsLane L;
  L.src = a.src; L.dst = a.dst; L.sstride = a.sstride; L.dstride = a.dstride; L.h = a.h;
  L.gx = (tx * lx + l) * 8; L.gxc = min(L.gx, a.w - 8);
  L.y0 = (ty * (64 >> a.lxShift) + g) * SS_RB; L.yEnd = min(L.y0 + SS_RB, a.h);
  L.live = L.gx < a.w && L.y0 < a.h;
  L.isL = l == 0; L.isR = l == lx - 1;
  L.ex = L.isL ? max(L.gx - 1, 0) : L.isR ? min(L.gx + 8, a.w - 1) : L.gxc;
  SsRaw A0;
  ss_load(L, L.y0 - 1, A0);                                                    // the first rows travel while the CTU's parameters are read
  const int cx = L.gxc / a.ctuW, cy = min(L.y0, a.h - 1) / a.ctuH;
  const vvcgpu_sao_ctu* prm = a.params + cy * a.wCtu + cx;
  L.x0 = cx * a.ctuW; L.x1 = min(L.x0 + a.ctuW, a.w); L.cy0 = cy * a.ctuH; L.cy1 = min(L.cy0 + a.ctuH, a.h);
  L.avail = prm->avail;
  L.lo32 = (unsigned)(p.clpMin + 32) * 0x10001u; L.hi32 = (unsigned)(p.clpMax + 32) * 0x10001u;
#ifdef SAO_FORCE_TYPE
  const int type = SAO_FORCE_TYPE;
#else
  const int type = prm->type;
#endif
#ifdef SAO_STRIP_COPYONLY
  ss_walk(L, A0, [&](const SsRow&, const SsRow& mid, const SsRow& dn, int y, bool) { unsigned o[4] = { mid.v[0], mid.v[1], mid.v[2], mid.v[3] ^ (dn.l & dn.r & 0u) }; ss_store(L, y, o); });
  return;
#endif
  // The tile of a plane is at most one CTU wide (host), so the type is wave-uniform wherever the CTU is at least as tall as the tile: one walk with
  // the type's row body.  A wave that sees several types, or offsets outside the byte table, takes the general walk: loads and lane shifts are common
  // code that every lane executes, only the row bodies branch on the lane's type (a shift inside a divergent branch would read lanes switched off there).
  SsEo E;
  E.fits = true; E.tLo = E.tHi = 0u; E.aL = E.aR = E.bL = E.bR = E.aIn = E.bIn = 0;
  E.mI[0] = E.mI[1] = E.mI[2] = E.mI[3] = 0u; E.off[0] = E.off[1] = E.off[2] = E.off[3] = E.off[4] = 0;
  switch (type)
  {
  case 0: ss_eo_setup<-1, 0>(L, prm, E); break;
  case 1: ss_eo_setup<0, -1>(L, prm, E); break;
  case 2: ss_eo_setup<-1, -1>(L, prm, E); break;
  case 3: ss_eo_setup<1, -1>(L, prm, E); break;
  default: break;
  }
  const int t0 = __builtin_amdgcn_readfirstlane(type);
  const bool fullBand = !L.live || L.yEnd == L.y0 + SS_RB;                     // (a band cut by the picture's last row ends its CTU on another row than its last: general walk)
  if (__all(type == t0 && E.fits && fullBand))
  {
    if (t0 == 0)      ss_walk(L, A0, [&](const SsRow& u, const SsRow& m, const SsRow& d, int y, bool er) { ss_eo_row<-1, 0, true>(L, E, u, m, d, y, er); });
    else if (t0 == 1) ss_walk(L, A0, [&](const SsRow& u, const SsRow& m, const SsRow& d, int y, bool er) { ss_eo_row<0, -1, true>(L, E, u, m, d, y, er); });
    else if (t0 == 2) ss_walk(L, A0, [&](const SsRow& u, const SsRow& m, const SsRow& d, int y, bool er) { ss_eo_row<-1, -1, true>(L, E, u, m, d, y, er); });
    else if (t0 == 3) ss_walk(L, A0, [&](const SsRow& u, const SsRow& m, const SsRow& d, int y, bool er) { ss_eo_row<1, -1, true>(L, E, u, m, d, y, er); });
    else if (t0 == 4)
    {
      __shared__ unsigned short boTab[32];
      if (lane < 32) boTab[lane] = (unsigned short)prm->offset[lane];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      ss_walk(L, A0, [&](const SsRow&, const SsRow& m, const SsRow&, int y, bool) { ss_bo_row<true>(L, prm->offset, boTab, p.boShift, m, y); });
    }
    else              ss_walk(L, A0, [&](const SsRow&, const SsRow& m, const SsRow&, int y, bool) { unsigned o[4] = { m.v[0], m.v[1], m.v[2], m.v[3] }; ss_store(L, y, o); });
    return;
  }
  ss_walk(L, A0, [&](const SsRow& u, const SsRow& m, const SsRow& d, int y, bool)
  {
    const bool er = y == L.cy0 || y == L.cy1 - 1;
    switch (type)
    {
    case 0: if (E.fits) ss_eo_row<-1, 0, true>(L, E, u, m, d, y, er); else ss_eo_row<-1, 0, false>(L, E, u, m, d, y, er); break;
    case 1: if (E.fits) ss_eo_row<0, -1, true>(L, E, u, m, d, y, er); else ss_eo_row<0, -1, false>(L, E, u, m, d, y, er); break;
    case 2: if (E.fits) ss_eo_row<-1, -1, true>(L, E, u, m, d, y, er); else ss_eo_row<-1, -1, false>(L, E, u, m, d, y, er); break;
    case 3: if (E.fits) ss_eo_row<1, -1, true>(L, E, u, m, d, y, er); else ss_eo_row<1, -1, false>(L, E, u, m, d, y, er); break;
    case 4: ss_bo_row<false>(L, prm->offset, nullptr, p.boShift, m, y); break;
    default: { unsigned o[4] = { m.v[0], m.v[1], m.v[2], m.v[3] }; ss_store(L, y, o); break; }
    }
  });
}


}  // namespace

extern "C" int vvcgpu_sao_apply(const vvc_pel* src, int src_stride, vvc_pel* dst, int dst_stride,
                                int width, int height, int ctu_w, int ctu_h, int bit_depth,
                                const vvcgpu_sao_ctu* params, int clp_min, int clp_max, void* stream)
{
  VVC_CHECK_ARG(src && dst && params, "sao_apply: null pointer");
  VVC_CHECK_ARG(src != dst, "sao_apply: src must not alias dst (reference reads the deblocked copy)");
  VVC_CHECK_ARG(width > 0 && height > 0, "sao_apply: bad size %dx%d", width, height);
  VVC_CHECK_ARG(src_stride >= width && dst_stride >= width, "sao_apply: stride < width");
  VVC_CHECK_ARG(ctu_w >= 8 && ctu_h >= 4 && (ctu_w & 7) == 0 && (ctu_h & 3) == 0,
                "sao_apply: CTU %dx%d must be a multiple of 8x4", ctu_w, ctu_h);
  VVC_CHECK_ARG(bit_depth >= 8 && bit_depth <= 10, "sao_apply: bit depth %d outside 8..10", bit_depth);
  int tpcShift = 6;                                                    // threads across a wave: one CTU width when that is a power of two
  if ((ctu_w & (ctu_w - 1)) == 0 && ctu_w <= 512) { tpcShift = 0; while ((8 << tpcShift) < ctu_w) tpcShift++; }
  const int nWaveCols = cdiv(width, 8 << tpcShift), nWaveRows = cdiv(height, (64 >> tpcShift) * SAO_ROWS);
  hipLaunchKernelGGL(sao_apply_kernel, dim3(cdiv(nWaveCols * nWaveRows, 4)), dim3(256), 0, (hipStream_t)stream, src, src_stride, dst,
                     dst_stride, width, height, ctu_w, ctu_h, cdiv(width, ctu_w), bit_depth - 5, params, clp_min, clp_max, tpcShift,
                     nWaveCols);
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}

extern "C" int vvcgpu_sao_apply_picture(const vvcgpu_planes* src, const vvcgpu_planes* dst, int width, int height, int ctu_size, int bit_depth,
                                        const vvcgpu_sao_ctu* params_y, const vvcgpu_sao_ctu* params_cb, const vvcgpu_sao_ctu* params_cr,
                                        int clp_min, int clp_max, void* stream)
{
  VVC_CHECK_ARG(src && dst && params_y && params_cb && params_cr, "sao_apply_picture: null pointer");
  VVC_CHECK_ARG(width > 0 && height > 0 && (width & 1) == 0 && (height & 1) == 0, "sao_apply_picture: bad size %dx%d", width, height);
  VVC_CHECK_ARG(ctu_size >= 16 && (ctu_size & 15) == 0, "sao_apply_picture: CTU size %d", ctu_size);
  VVC_CHECK_ARG(bit_depth >= 8 && bit_depth <= 10, "sao_apply_picture: bit depth %d outside 8..10", bit_depth);
  const vvcgpu_sao_ctu* prm[3] = { params_y, params_cb, params_cr };
  // the strip form: planes whose rows are whole 16-byte words at 16-byte aligned addresses, CTU rows that are whole bands of 16 rows
  bool strip = (width & 15) == 0 && (ctu_size & (2 * SS_RB - 1)) == 0;
  for (int c = 0; c < 3; c++)
    strip = strip && src->p[c] && dst->p[c] && (src->stride[c] & 7) == 0 && (dst->stride[c] & 7) == 0 && (reinterpret_cast<uintptr_t>(src->p[c]) & 15) == 0 &&
            (reinterpret_cast<uintptr_t>(dst->p[c]) & 15) == 0;
  if (strip)
  {
    SaoStrip3 q;
    int end = 0;
    for (int c = 0; c < 3; c++)
    {
      const int w = c ? width >> 1 : width, h = c ? height >> 1 : height, ctu = c ? ctu_size >> 1 : ctu_size;
      VVC_CHECK_ARG(src->p[c] != dst->p[c] && src->stride[c] >= w && dst->stride[c] >= w, "sao_apply_picture: plane %d", c);
      int lxShift = 4;                                                          // 16 lanes = 128 columns across, narrower where the CTU is (64: 8, 32 and below: 4)
      while (lxShift > 2 && (8 << lxShift) > ctu) lxShift--;
      const int tilesX = cdiv(w, 8 << lxShift), tilesY = cdiv(h, (64 >> lxShift) * SS_RB);
      end += tilesX * tilesY;
      q.a[c] = SaoStripPlane{ src->p[c], dst->p[c], prm[c], src->stride[c], dst->stride[c], w, h, ctu, ctu, cdiv(w, ctu), tilesX, end, lxShift };
    }
    q.boShift = bit_depth - 5; q.clpMin = clp_min; q.clpMax = clp_max; q.total = end; q.xcd = vvc_xcd_on();
    hipLaunchKernelGGL(sao_apply_strip_kernel, dim3(vvc_xcd_grid(end, q.xcd)), dim3(64), 0, (hipStream_t)stream, q);
    VVC_LAUNCH_CHECK();
    return VVCGPU_OK;
  }
  SaoApply3 p;
  int end = 0;
  for (int c = 0; c < 3; c++)
  {
    const int w = c ? width >> 1 : width, h = c ? height >> 1 : height, ctu = c ? ctu_size >> 1 : ctu_size;
    VVC_CHECK_ARG(src->p[c] && dst->p[c] && src->p[c] != dst->p[c] && src->stride[c] >= w && dst->stride[c] >= w, "sao_apply_picture: plane %d", c);
    int tpcShift = 6;
    if ((ctu & (ctu - 1)) == 0 && ctu <= 512) { tpcShift = 0; while ((8 << tpcShift) < ctu) tpcShift++; }
    const int nWaveCols = cdiv(w, 8 << tpcShift), nWaveRows = cdiv(h, (64 >> tpcShift) * SAO_ROWS);
    end += cdiv(nWaveCols * nWaveRows, 4);
    p.a[c] = SaoPlaneArgs{ src->p[c], dst->p[c], prm[c], src->stride[c], dst->stride[c], w, h, ctu, ctu, cdiv(w, ctu), tpcShift, nWaveCols, end };
  }
  p.boShift = bit_depth - 5; p.clpMin = clp_min; p.clpMax = clp_max;
  p.total = end; p.xcd = vvc_xcd_on();
  hipLaunchKernelGGL(sao_apply_picture_kernel, dim3(vvc_xcd_grid2(p.a[0].wgEnd, end, p.xcd)), dim3(256), 0, (hipStream_t)stream, p);
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}
