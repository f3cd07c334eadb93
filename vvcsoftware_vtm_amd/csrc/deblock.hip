// deblock.hip -- deblocking sample filters (L2) over host-derived edge/BS/QP maps (L1) for gfx950.
//
// Reference behaviour reproduced (bit-exact): LoopFilter::xEdgeFilterLuma / xEdgeFilterChroma /
// xPelFilterLuma / xPelFilterChroma / xUseStrongFiltering / xCalcDP / xCalcDQ
// (CommonLib/LoopFilter.cpp:543-980) in the pass order of loopFilterPic (:149-230): all vertical
// edges of the picture, then all horizontal edges.
//
// Design: ONE kernel, ONE pass over HBM (2 x picture bytes + maps) instead of the reference's two passes.
// Edges lie on an 8-sample grid and a filter touches at most 4 samples on each side, so a tile whose
// origin is shifted by (-4,-4) from the grid contains the complete footprint of every edge it owns, in both
// directions: the tile is staged in LDS, all vertical edges are filtered (LDS in place), barrier, all
// horizontal edges are filtered on the result, barrier, tile written back.  No halo, no inter-workgroup
// dependency, no second launch.  The same holds for chroma (8-sample chroma grid, 2 samples per side).
#include "common.h"

namespace {

__constant__ uint8_t c_tc[66] = {
  0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,1,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,5,5,6,6,7,8,9,10,11,13,14,16,18,20,22,24,
  26,28,30,32,34,36,38,40,42,44,46,48 };
__constant__ uint8_t c_beta[64] = {
  0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,6,7,8,9,10,11,12,13,14,15,16,17,18,20,22,24,26,28,30,32,34,36,38,40,42,44,46,48,50,52,
  54,56,58,60,62,64,66,68,70,72,74,76,78,80,82,84,86,88 };
__constant__ uint8_t c_chromaScale420[70] = {
  0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,29,30,31,32,33,33,34,34,35,35,36,36,
  37,37,38,39,40,41,42,43,44,45,46,47,48,49,50,51,52,53,54,55,56,57,58,59,60,61,62,63 };

constexpr int TS = 64;        // tile size (samples) in both dimensions
constexpr int TP = 72;        // LDS pitch (samples); 144-byte rows spread 4-row-apart segments over banks
constexpr int MAXQP = 63, TCOFF = 2, QPMAPSZ = 70;

struct Seg8 { int m[8]; };    // m[0..3] = P side (m0..m3), m[4..7] = Q side

__device__ __forceinline__ void filter_luma_line(Seg8& v, int tc, bool sw, bool noP, bool noQ, int thrCut,
                                                 bool fP, bool fQ, int cmin, int cmax)
{
  const int m0 = v.m[0], m1 = v.m[1], m2 = v.m[2], m3 = v.m[3], m4 = v.m[4], m5 = v.m[5], m6 = v.m[6], m7 = v.m[7];
  int n1 = m1, n2 = m2, n3 = m3, n4 = m4, n5 = m5, n6 = m6;
  if (sw)
  {
    n3 = clip3(m3 - 2 * tc, m3 + 2 * tc, (m1 + 2 * m2 + 2 * m3 + 2 * m4 + m5 + 4) >> 3);
    n4 = clip3(m4 - 2 * tc, m4 + 2 * tc, (m2 + 2 * m3 + 2 * m4 + 2 * m5 + m6 + 4) >> 3);
    n2 = clip3(m2 - 2 * tc, m2 + 2 * tc, (m1 + m2 + m3 + m4 + 2) >> 2);
    n5 = clip3(m5 - 2 * tc, m5 + 2 * tc, (m3 + m4 + m5 + m6 + 2) >> 2);
    n1 = clip3(m1 - 2 * tc, m1 + 2 * tc, (2 * m0 + 3 * m1 + m2 + m3 + m4 + 4) >> 3);
    n6 = clip3(m6 - 2 * tc, m6 + 2 * tc, (m3 + m4 + m5 + 3 * m6 + 2 * m7 + 4) >> 3);
    // the reference stores these into Pel without ClipPel (LoopFilter.cpp:871-876)
    n1 = (short)n1; n2 = (short)n2; n3 = (short)n3; n4 = (short)n4; n5 = (short)n5; n6 = (short)n6;
  }
  else
  {
    int delta = (9 * (m4 - m3) - 3 * (m5 - m2) + 8) >> 4;
    if (abs(delta) < thrCut)
    {
      delta = clip3(-tc, tc, delta);
      n3 = clip3(cmin, cmax, m3 + delta);
      n4 = clip3(cmin, cmax, m4 - delta);
      const int tc2 = tc >> 1;
      if (fP) n2 = clip3(cmin, cmax, m2 + clip3(-tc2, tc2, ((((m1 + m3 + 1) >> 1) - m2 + delta) >> 1)));
      if (fQ) n5 = clip3(cmin, cmax, m5 + clip3(-tc2, tc2, ((((m6 + m4 + 1) >> 1) - m5 - delta) >> 1)));
    }
  }
  if (noP) { n3 = m3; n2 = m2; n1 = m1; }
  if (noQ) { n4 = m4; n5 = m5; n6 = m6; }
  v.m[1] = n1; v.m[2] = n2; v.m[3] = n3; v.m[4] = n4; v.m[5] = n5; v.m[6] = n6;
}

__device__ __forceinline__ bool use_strong(const Seg8& v, int d, int beta, int tc)
{
  const int ds = abs(v.m[0] - v.m[3]) + abs(v.m[7] - v.m[4]);
  return (ds < (beta >> 3)) && (d < (beta >> 2)) && (abs(v.m[3] - v.m[4]) < ((tc * 5 + 1) >> 1));
}
__device__ __forceinline__ int calc_dp(const Seg8& v) { return abs(v.m[1] - 2 * v.m[2] + v.m[3]); }
__device__ __forceinline__ int calc_dq(const Seg8& v) { return abs(v.m[4] - 2 * v.m[5] + v.m[6]); }

// Decide + filter one 4-line luma segment held in registers.
// tab: the workgroup's LDS copy of c_tc | c_beta | c_chromaScale420 (DbTab offsets): a look-up in constant memory is a trip to the vector cache on the
// critical path qp -> tc / beta -> decision of every segment
constexpr int DB_TC = 0, DB_BETA = 66, DB_CS = 66 + 64, DB_TABN = 66 + 64 + 70;
__device__ __forceinline__ void luma_segment(Seg8 (&ln)[4], int bs, int qpP, int qpQ, bool noP, bool noQ,
                                             const vvcgpu_deblock_cfg& c, const uint8_t* tab)
{
  const int qp = (qpP + qpQ + 1) >> 1;
  const int scale = 1 << (c.bit_depth_luma - 8);
  const int tc = tab[DB_TC + clip3(0, MAXQP + TCOFF, qp + TCOFF * (bs - 1) + (c.tc_offset_div2 << 1))] * scale;
  const int beta = tab[DB_BETA + clip3(0, MAXQP, qp + (c.beta_offset_div2 << 1))] * scale;
  const int side = (beta + (beta >> 1)) >> 3, thrCut = tc * 10;
  const int dp0 = calc_dp(ln[0]), dq0 = calc_dq(ln[0]), dp3 = calc_dp(ln[3]), dq3 = calc_dq(ln[3]);
  const int d0 = dp0 + dq0, d3 = dp3 + dq3;
  if (d0 + d3 < beta)
  {
    const bool fP = (dp0 + dp3) < side, fQ = (dq0 + dq3) < side;
    const bool sw = use_strong(ln[0], 2 * d0, beta, tc) && use_strong(ln[3], 2 * d3, beta, tc);
#pragma unroll
    for (int i = 0; i < 4; i++) filter_luma_line(ln[i], tc, sw, noP, noQ, thrCut, fP, fQ, c.clp_min[0], c.clp_max[0]);
  }
}

// Tile loader/writer: tile origin (ox, oy) may be negative / overhang; outside samples read as 0 and are never written.
__device__ __forceinline__ void tile_load(short* lds, const Pel* plane, int stride, int w, int h, int ox, int oy,
                                          int tid, int nthreads)
{
  const bool vec_ok = ((stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(plane) & 7) == 0);
  for (int v = tid; v < TS * (TS / 4); v += nthreads)
  {
    const int r = v / (TS / 4), c = (v - r * (TS / 4)) * 4;
    const int y = oy + r, x = ox + c;
    pel4 val = { 0, 0, 0, 0 };
    if (y >= 0 && y < h)
    {
      const Pel* row = plane + (size_t)y * stride;
      if (vec_ok && x >= 0 && x + 3 < w) val = *reinterpret_cast<const pel4*>(row + x);
      else
      {
#pragma unroll
        for (int k = 0; k < 4; k++) if (x + k >= 0 && x + k < w) val[k] = row[x + k];
      }
    }
    *reinterpret_cast<pel4*>(lds + r * TP + c) = val;
  }
}
__device__ __forceinline__ void tile_store(const short* lds, Pel* plane, int stride, int w, int h, int ox, int oy,
                                           int tid, int nthreads)
{
  const bool vec_ok = ((stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(plane) & 7) == 0);
  for (int v = tid; v < TS * (TS / 4); v += nthreads)
  {
    const int r = v / (TS / 4), c = (v - r * (TS / 4)) * 4;
    const int y = oy + r, x = ox + c;
    if (y < 0 || y >= h) continue;
    const pel4 val = *reinterpret_cast<const pel4*>(lds + r * TP + c);
    Pel* row = plane + (size_t)y * stride;
    if (vec_ok && x >= 0 && x + 3 < w) *reinterpret_cast<pel4*>(row + x) = val;
    else
    {
#pragma unroll
      for (int k = 0; k < 4; k++) if (x + k >= 0 && x + k < w) row[x + k] = val[k];
    }
  }
}

__device__ __forceinline__ void deblock_luma_body(const dim3 bid, short* tile, const uint8_t* tab, Pel* __restrict__ Y, int stride, int w, int h,
                                                           const uint8_t* __restrict__ edgeV,
                                                           const uint8_t* __restrict__ edgeH,
                                                           const int8_t* __restrict__ qpm, vvcgpu_deblock_cfg cfg)
{
  const int tid = threadIdx.x;
  const int ox = bid.x * TS - 4, oy = bid.y * TS - 4;
  const int w4 = w >> 2;
  // the map bytes of this thread's two tasks travel WITH the tile (requested first: they are back before the tile's last rows); read inside the passes
  // they were two more trips to memory on the path load -> edge byte -> QP -> tc -> filter of every workgroup
  //   pass 1: vertical edges.   task = (edge k: x = ox+4+8k, segment s: rows oy+4s .. +3)
  //   pass 2: horizontal edges. task = (edge k: y = oy+4+8k, segment s: cols ox+4s .. +3)
  const int k1 = tid & 7, s1 = tid >> 3, x1 = ox + 4 + 8 * k1, y1 = oy + 4 * s1;
  const int s2 = tid & 15, k2 = tid >> 4, y2 = oy + 4 + 8 * k2, x2 = ox + 4 * s2;
  const bool in1 = x1 > 0 && x1 < w && y1 >= 0 && y1 < h, in2 = y2 > 0 && y2 < h && x2 >= 0 && x2 < w;
  const int u1 = in1 ? (y1 >> 2) * w4 + (x1 >> 2) : 1, u2 = in2 ? (y2 >> 2) * w4 + (x2 >> 2) : w4;
  const int e1 = in1 ? edgeV[u1] : 0, e2 = in2 ? edgeH[u2] : 0;
  const int qp1P = qpm[u1 - 1], qp1Q = qpm[u1], qp2P = qpm[u2 - w4], qp2Q = qpm[u2];
  tile_load(tile, Y, stride, w, h, ox, oy, tid, 128);
  __syncthreads();

  if (e1 & 3)
  {
    Seg8 ln[4];
    short* p = tile + (4 * s1) * TP + 8 * k1;       // sample x-4 of row y
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
      const pel8 v = *reinterpret_cast<const pel8*>(p + i * TP);
#pragma unroll
      for (int j = 0; j < 8; j++) ln[i].m[j] = v[j];
    }
    luma_segment(ln, e1 & 3, qp1P, qp1Q, (e1 >> 4) & 1, (e1 >> 5) & 1, cfg, tab);
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
      pel8 v;
#pragma unroll
      for (int j = 0; j < 8; j++) v[j] = (short)ln[i].m[j];
      *reinterpret_cast<pel8*>(p + i * TP) = v;
    }
  }
  __syncthreads();

  if (e2 & 3)
  {
    Seg8 ln[4];
    short* p = tile + (8 * k2) * TP + 4 * s2;       // row y-4, col x
#pragma unroll
    for (int j = 0; j < 8; j++)
    {
      const pel4 v = *reinterpret_cast<const pel4*>(p + j * TP);
#pragma unroll
      for (int i = 0; i < 4; i++) ln[i].m[j] = v[i];
    }
    luma_segment(ln, e2 & 3, qp2P, qp2Q, (e2 >> 4) & 1, (e2 >> 5) & 1, cfg, tab);
#pragma unroll
    for (int j = 1; j < 7; j++)
    {
      pel4 v;
#pragma unroll
      for (int i = 0; i < 4; i++) v[i] = (short)ln[i].m[j];
      *reinterpret_cast<pel4*>(p + j * TP) = v;
    }
  }
  __syncthreads();
  tile_store(tile, Y, stride, w, h, ox, oy, tid, 128);
}

__device__ __forceinline__ int chroma_tc(int qpP, int qpQ, int qpOff, const vvcgpu_deblock_cfg& c, const uint8_t* tab)
{
  int qp = ((qpP + qpQ + 1) >> 1) + qpOff;
  if (qp >= QPMAPSZ) qp -= 6;
  else if (qp >= 0) qp = tab[DB_CS + qp];
  return tab[DB_TC + clip3(0, MAXQP + TCOFF, qp + TCOFF + (c.tc_offset_div2 << 1))] * (1 << (c.bit_depth_chroma - 8));
}

// Chroma plane (w,h are CHROMA dimensions).  blockIdx.z selects Cb / Cr.
__device__ __forceinline__ void deblock_chroma_body(const dim3 bid, short* tile, const uint8_t* tab, Pel* __restrict__ Cb, Pel* __restrict__ Cr, int stride,
                                                             int w, int h, int w4 /* luma units per row */,
                                                             const uint8_t* __restrict__ edgeV,
                                                             const uint8_t* __restrict__ edgeH,
                                                             const int8_t* __restrict__ qpm, vvcgpu_deblock_cfg cfg)
{
  const int tid = threadIdx.x;
  const int comp = bid.z;                       // 0 = Cb, 1 = Cr
  Pel* plane = comp ? Cr : Cb;
  const int qpOff = comp ? cfg.cr_qp_offset : cfg.cb_qp_offset;
  const int cmin = cfg.clp_min[1 + comp], cmax = cfg.clp_max[1 + comp];
  const int ox = bid.x * TS - 4, oy = bid.y * TS - 4;
  // the map bytes of this thread's four tasks, requested with the tile (see deblock_luma_body)
  //   pass 1: vertical edges at chroma x = ox+4+8k; segments of 2 rows (one luma unit): 8 edges x 32 segments, two per thread
  //   pass 2: horizontal edges at chroma y = oy+4+8k; segments of 2 columns: 8 edges x 32 segments
  int e1[2], q1P[2], q1Q[2], e2[2], q2P[2], q2Q[2];
#pragma unroll
  for (int r = 0; r < 2; r++)
  {
    const int t = tid + 128 * r;
    const int x1 = ox + 4 + 8 * (t & 7), y1 = oy + 2 * (t >> 3);
    const bool in1 = x1 > 0 && x1 < w && y1 >= 0 && y1 < h;
    const int u1 = in1 ? (y1 >> 1) * w4 + (x1 >> 1) : 1;        // luma unit: (2y)/4, (2x)/4
    e1[r] = in1 ? edgeV[u1] : 0; q1P[r] = qpm[u1 - 1]; q1Q[r] = qpm[u1];
    const int y2 = oy + 4 + 8 * (t >> 5), x2 = ox + 2 * (t & 31);
    const bool in2 = y2 > 0 && y2 < h && x2 >= 0 && x2 < w;
    const int u2 = in2 ? (y2 >> 1) * w4 + (x2 >> 1) : w4;
    e2[r] = in2 ? edgeH[u2] : 0; q2P[r] = qpm[u2 - w4]; q2Q[r] = qpm[u2];
  }
  tile_load(tile, plane, stride, w, h, ox, oy, tid, 128);
  __syncthreads();

#pragma unroll
  for (int r = 0; r < 2; r++)
  {
    const int t = tid + 128 * r, k = t & 7, s = t >> 3, e = e1[r];
    if (((e >> 2) & 3) > 1)
    {
      const int tc = chroma_tc(q1P[r], q1Q[r], qpOff, cfg, tab);
      const bool noP = (e >> 4) & 1, noQ = (e >> 5) & 1;
      short* p = tile + (2 * s) * TP + 8 * k + 2;  // sample x-2 of row y
#pragma unroll
      for (int i = 0; i < 2; i++)
      {
        short* q = p + i * TP;
        const int m2 = q[0], m3 = q[1], m4 = q[2], m5 = q[3];
        const int delta = clip3(-tc, tc, ((((m4 - m3) << 2) + m2 - m5 + 4) >> 3));
        if (!noP) q[1] = (short)clip3(cmin, cmax, m3 + delta);
        if (!noQ) q[2] = (short)clip3(cmin, cmax, m4 - delta);
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 2; r++)
  {
    const int t = tid + 128 * r, s = t & 31, k = t >> 5, e = e2[r];
    if (((e >> 2) & 3) > 1)
    {
      const int tc = chroma_tc(q2P[r], q2Q[r], qpOff, cfg, tab);
      const bool noP = (e >> 4) & 1, noQ = (e >> 5) & 1;
      short* p = tile + (8 * k + 2) * TP + 2 * s;  // row y-2, col x
#pragma unroll
      for (int i = 0; i < 2; i++)
      {
        short* q = p + i;
        const int m2 = q[0], m3 = q[TP], m4 = q[2 * TP], m5 = q[3 * TP];
        const int delta = clip3(-tc, tc, ((((m4 - m3) << 2) + m2 - m5 + 4) >> 3));
        if (!noP) q[TP] = (short)clip3(cmin, cmax, m3 + delta);
        if (!noQ) q[2 * TP] = (short)clip3(cmin, cmax, m4 - delta);
      }
    }
  }
  __syncthreads();
  tile_store(tile, plane, stride, w, h, ox, oy, tid, 128);
}

// luma and both chroma planes in ONE launch: the chroma planes alone are ~1 workgroup per CU, a launch of their own costs its latency floor
__global__ __launch_bounds__(128) void deblock_picture_kernel(Pel* __restrict__ Y, int strideY, Pel* __restrict__ Cb, Pel* __restrict__ Cr, int strideC,
                                                              int w, int h, int glx, int nLuma, int gcx, int gcy,
                                                              const uint8_t* __restrict__ edgeV, const uint8_t* __restrict__ edgeH,
                                                              const int8_t* __restrict__ qpLuma, const int8_t* __restrict__ qpChroma,
                                                              vvcgpu_deblock_cfg cfg, int total, int xcd)
{
  __shared__ short tile[TS * TP];
  __shared__ uint8_t tab[DB_TABN + 8];
  const int b = vvc_xcd_index2((int)blockIdx.x, nLuma, total, xcd);
  if (b < 0) return;
  for (int i = threadIdx.x; i < DB_TABN; i += 128) tab[i] = i < DB_BETA ? c_tc[i] : i < DB_CS ? c_beta[i - DB_BETA] : c_chromaScale420[i - DB_CS];     // (visible behind the bodies' first barrier)
  if (b < nLuma) deblock_luma_body(dim3(b % glx, b / glx, 0), tile, tab, Y, strideY, w, h, edgeV, edgeH, qpLuma, cfg);
  else
  {
    const int c = b - nLuma, per = gcx * gcy, z = c / per, r = c - z * per;
    deblock_chroma_body(dim3(r % gcx, r / gcx, z), tile, tab, Cb, Cr, strideC, w >> 1, h >> 1, w >> 2, edgeV, edgeH, qpChroma, cfg);
  }
}


// ---------------------------------------------------------------------------------------------------
// Block form (round 6).  The footprint argument of the tile form holds for a tile of 8 x 8 samples as well: the block whose origin is shifted by
// (-4, -4) from the 8-sample grid contains ONE vertical edge (its middle column) and ONE horizontal edge (its middle row) with everything they
// read and write -- the vertical filter touches columns x - 3 .. x + 2, decisions read the block's rows only, and the horizontal pass of the
// block reads what the block's own vertical pass produced.  So a LANE takes a block: eight 16-byte row loads into registers, the two 4-row
// segments of its vertical edge, the two 4-column segments of its horizontal edge on the result (a register "transpose" is just indexing),
// eight row stores.  No LDS tile, no barriers, no phases that every wave of the chip runs in lockstep (the tile form: load, barrier, vertical pass,
// barrier, horizontal pass, barrier, store in 3060 workgroups; 22.6 us per 4K picture with the vector pipes 38 % busy).  A wave = 64 blocks
// side by side: a row request is 1 KB of consecutive bytes.  Chroma: the same blocks on the chroma grid, four 2-row / 2-column segments per edge.
// Planes whose rows are not 8-byte aligned keep the tile form (tests/test_gpu_deblock.py::test_deblock_unaligned_rows_tile_form).
// ---------------------------------------------------------------------------------------------------
struct DbBlock { int m[8][8]; };                                              // [row][column] of the shifted block

template <bool LUMA>
__device__ __forceinline__ void deblock_block(Pel* __restrict__ plane, int stride, int w, int h, int bx, int by, int w4, int comp,
                                              const uint8_t* __restrict__ edgeV, const uint8_t* __restrict__ edgeH, const int8_t* __restrict__ qpm,
                                              const vvcgpu_deblock_cfg& cfg, const uint8_t* tab)
{
  const int x0 = 8 * bx - 4, y0 = 8 * by - 4;                                  // block origin; the edges: x0 + 4 (vertical), y0 + 4 (horizontal)
  const bool colLo = bx > 0, colHi = x0 + 4 < w, rowLo = by > 0, rowHi = y0 + 4 < h;   // which halves lie inside the plane (w, h are multiples of 8)
  const bool hasV = colLo && colHi, hasH = rowLo && rowHi;
  // map bytes first (they are small and come back before the rows)
  constexpr int NS = LUMA ? 2 : 4;                                             // segments per edge: 4 rows / columns (luma), 2 (chroma: one luma unit)
  constexpr int SL = 8 / NS;
  int eV[NS], qVP[NS], qVQ[NS], eH[NS], qHP[NS], qHQ[NS];
#pragma unroll
  for (int t = 0; t < NS; t++)
  {
    const int yv = y0 + SL * t, xh = x0 + SL * t;
    const bool inV = hasV && yv >= 0 && yv < h, inH = hasH && xh >= 0 && xh < w;
    const int uV = inV ? (LUMA ? (yv >> 2) * w4 + ((x0 + 4) >> 2) : (yv >> 1) * w4 + ((x0 + 4) >> 1)) : 1;
    const int uH = inH ? (LUMA ? ((y0 + 4) >> 2) * w4 + (xh >> 2) : ((y0 + 4) >> 1) * w4 + (xh >> 1)) : w4;
    eV[t] = inV ? edgeV[uV] : 0; qVP[t] = qpm[uV - 1]; qVQ[t] = qpm[uV];
    eH[t] = inH ? edgeH[uH] : 0; qHP[t] = qpm[uH - w4]; qHQ[t] = qpm[uH];
  }
  // in place: a block none of whose segments is flagged keeps its samples -- nothing to compute and nothing to write back
  bool any = false;
#pragma unroll
  for (int t = 0; t < NS; t++) any = any || (LUMA ? ((eV[t] | eH[t]) & 3) != 0 : ((eV[t] >> 2) & 3) > 1 || ((eH[t] >> 2) & 3) > 1);
  DbBlock B;
  uint4 raw[8];
#pragma unroll
  for (int r = 0; r < 8; r++)
  {
    raw[r] = make_uint4(0u, 0u, 0u, 0u);
    const int y = y0 + r;
    if (r < 4 ? rowLo : rowHi)
    {
      const Pel* q = plane + (size_t)y * stride + x0;
      if (hasV) raw[r] = *reinterpret_cast<const uint4*>(q);
      else if (colLo) { const uint2 v = *reinterpret_cast<const uint2*>(q); raw[r].x = v.x; raw[r].y = v.y; }
      else            { const uint2 v = *reinterpret_cast<const uint2*>(q + 4); raw[r].z = v.x; raw[r].w = v.y; }
    }
  }
  if (!any) return;
#pragma unroll
  for (int r = 0; r < 8; r++)
  {
    const unsigned d[4] = { raw[r].x, raw[r].y, raw[r].z, raw[r].w };
#pragma unroll
    for (int c = 0; c < 8; c++) B.m[r][c] = (int)(short)(d[c >> 1] >> (16 * (c & 1)));
  }
#if defined(DB_COPYONLY) || defined(DB_NO_LUMA)
  if (LUMA) {} else if (false)
#else
  if (LUMA)
#endif
  {
#pragma unroll
    for (int t = 0; t < 2; t++)                                                // vertical edge: rows 4 t .. 4 t + 3
      if (eV[t] & 3)
      {
        Seg8 ln[4];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 0; j < 8; j++) ln[i].m[j] = B.m[4 * t + i][j];
        luma_segment(ln, eV[t] & 3, qVP[t], qVQ[t], (eV[t] >> 4) & 1, (eV[t] >> 5) & 1, cfg, tab);
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 1; j < 7; j++) B.m[4 * t + i][j] = (short)ln[i].m[j];
      }
#pragma unroll
    for (int t = 0; t < 2; t++)                                                // horizontal edge: columns 4 t .. 4 t + 3
      if (eH[t] & 3)
      {
        Seg8 ln[4];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 0; j < 8; j++) ln[i].m[j] = B.m[j][4 * t + i];
        luma_segment(ln, eH[t] & 3, qHP[t], qHQ[t], (eH[t] >> 4) & 1, (eH[t] >> 5) & 1, cfg, tab);
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 1; j < 7; j++) B.m[j][4 * t + i] = (short)ln[i].m[j];
      }
  }
#if defined(DB_COPYONLY) || defined(DB_NO_CHROMA)
  else if (false)
#else
  else
#endif
  {
    const int qpOff = comp ? cfg.cr_qp_offset : cfg.cb_qp_offset;
    const int cmin = cfg.clp_min[1 + comp], cmax = cfg.clp_max[1 + comp];
#pragma unroll
    for (int t = 0; t < 4; t++)                                                // vertical edge: rows 2 t, 2 t + 1; samples x - 2 .. x + 1 = columns 2 .. 5
      if (((eV[t] >> 2) & 3) > 1)
      {
        const int tc = chroma_tc(qVP[t], qVQ[t], qpOff, cfg, tab);
        const bool noP = (eV[t] >> 4) & 1, noQ = (eV[t] >> 5) & 1;
#pragma unroll
        for (int i = 0; i < 2; i++)
        {
          int (&q)[8] = B.m[2 * t + i];
          const int m2 = q[2], m3 = q[3], m4 = q[4], m5 = q[5];
          const int delta = clip3(-tc, tc, ((((m4 - m3) << 2) + m2 - m5 + 4) >> 3));
          if (!noP) q[3] = (short)clip3(cmin, cmax, m3 + delta);
          if (!noQ) q[4] = (short)clip3(cmin, cmax, m4 - delta);
        }
      }
#pragma unroll
    for (int t = 0; t < 4; t++)                                                // horizontal edge: columns 2 t, 2 t + 1; rows 2 .. 5
      if (((eH[t] >> 2) & 3) > 1)
      {
        const int tc = chroma_tc(qHP[t], qHQ[t], qpOff, cfg, tab);
        const bool noP = (eH[t] >> 4) & 1, noQ = (eH[t] >> 5) & 1;
#pragma unroll
        for (int i = 0; i < 2; i++)
        {
          const int c = 2 * t + i;
          const int m2 = B.m[2][c], m3 = B.m[3][c], m4 = B.m[4][c], m5 = B.m[5][c];
          const int delta = clip3(-tc, tc, ((((m4 - m3) << 2) + m2 - m5 + 4) >> 3));
          if (!noP) B.m[3][c] = (short)clip3(cmin, cmax, m3 + delta);
          if (!noQ) B.m[4][c] = (short)clip3(cmin, cmax, m4 - delta);
        }
      }
  }
#pragma unroll
  for (int r = 0; r < 8; r++)
  {
    if (!(r < 4 ? rowLo : rowHi)) continue;
    unsigned d[4];
#pragma unroll
    for (int c = 0; c < 4; c++) d[c] = ((unsigned)B.m[r][2 * c] & 0xFFFFu) | ((unsigned)B.m[r][2 * c + 1] << 16);
    Pel* q = plane + (size_t)(y0 + r) * stride + x0;
    if (hasV) *reinterpret_cast<uint4*>(q) = make_uint4(d[0], d[1], d[2], d[3]);
    else if (colLo) *reinterpret_cast<uint2*>(q) = make_uint2(d[0], d[1]);
    else            *reinterpret_cast<uint2*>(q + 4) = make_uint2(d[2], d[3]);
  }
}

__global__ __launch_bounds__(64) void deblock_block_kernel(Pel* __restrict__ Y, int strideY, Pel* __restrict__ Cb, Pel* __restrict__ Cr, int strideC,
                                                           int w, int h, int wavesRowL, int nLuma, int wavesRowC, int perC,
                                                           const uint8_t* __restrict__ edgeV, const uint8_t* __restrict__ edgeH,
                                                           const int8_t* __restrict__ qpLuma, const int8_t* __restrict__ qpChroma,
                                                           vvcgpu_deblock_cfg cfg, int total, int xcd)
{
  __shared__ uint8_t tab[DB_TABN + 8];
  const int b = vvc_xcd_index2((int)blockIdx.x, nLuma, total, xcd);
  if (b < 0) return;
  for (int i = threadIdx.x; i < DB_TABN; i += 64) tab[i] = i < DB_BETA ? c_tc[i] : i < DB_CS ? c_beta[i - DB_BETA] : c_chromaScale420[i - DB_CS];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const int lane = threadIdx.x;
  if (b < nLuma)
  {
    const int by = b / wavesRowL, bx = (b - by * wavesRowL) * 64 + lane;
    if (8 * bx - 4 < w) deblock_block<true>(Y, strideY, w, h, bx, by, w >> 2, 0, edgeV, edgeH, qpLuma, cfg, tab);
  }
  else
  {
    const int c = b - nLuma, z = c / perC, r = c - z * perC;
    const int by = r / wavesRowC, bx = (r - by * wavesRowC) * 64 + lane;
    if (8 * bx - 4 < (w >> 1)) deblock_block<false>(z ? Cr : Cb, strideC, w >> 1, h >> 1, bx, by, w >> 2, z, edgeV, edgeH, qpChroma, cfg, tab);
  }
}

}  // namespace

extern "C" int vvcgpu_deblock(vvc_pel* y, int stride_y, vvc_pel* cb, vvc_pel* cr, int stride_c, int width, int height,
                              const uint8_t* edge_ver, const uint8_t* edge_hor, const int8_t* qp_luma,
                              const int8_t* qp_chroma, const vvcgpu_deblock_cfg* cfg_host, void* stream)
{
  VVC_CHECK_ARG(y && edge_ver && edge_hor && qp_luma && cfg_host, "deblock: null pointer");
  VVC_CHECK_ARG(width > 0 && height > 0 && (width & 7) == 0 && (height & 7) == 0,
                "deblock: width/height must be positive multiples of 8 (got %dx%d)", width, height);
  VVC_CHECK_ARG(stride_y >= width, "deblock: luma stride %d < width %d", stride_y, width);
  VVC_CHECK_ARG((cb == nullptr) == (cr == nullptr), "deblock: Cb and Cr must both be given or both be NULL");
  VVC_CHECK_ARG(!cb || (qp_chroma && stride_c >= width / 2), "deblock: chroma needs qp_chroma and stride >= width/2");
  const vvcgpu_deblock_cfg cfg = *cfg_host;
  VVC_CHECK_ARG(cfg.bit_depth_luma >= 8 && cfg.bit_depth_luma <= 10 && cfg.bit_depth_chroma >= 8 &&
                cfg.bit_depth_chroma <= 10, "deblock: bit depths outside 8..10");
  hipStream_t st = (hipStream_t)stream;
  const int glx = cdiv(width + 4, TS), gly = cdiv(height + 4, TS);
  const int gcx = cdiv(width / 2 + 4, TS), gcy = cdiv(height / 2 + 4, TS);
  const int nLuma = glx * gly, nChroma = cb ? 2 * gcx * gcy : 0;
  const int xcd = vvc_xcd_on();
  // block form: rows that are whole 8-byte words at 8-byte aligned addresses, chroma sizes on the 8-sample grid as well
  const bool aligned = (stride_y & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 7) == 0 &&
                       (!cb || ((stride_c & 3) == 0 && (reinterpret_cast<uintptr_t>(cb) & 7) == 0 && (reinterpret_cast<uintptr_t>(cr) & 7) == 0));
  if (aligned)
  {
    // shifted blocks that hold a sample of the plane: origins -4, 4, .. < size (chroma sizes are multiples of 4: a half block is inside or outside as a whole)
    const int wavesRowL = cdiv(cdiv(width + 4, 8), 64), rowsL = cdiv(height + 4, 8), nL = wavesRowL * rowsL;
    const int wavesRowC = cdiv(cdiv(width / 2 + 4, 8), 64), rowsC = cdiv(height / 2 + 4, 8), perC = wavesRowC * rowsC, nC = cb ? 2 * perC : 0;
    hipLaunchKernelGGL(deblock_block_kernel, dim3(vvc_xcd_grid2(nL, nL + nC, xcd)), dim3(64), 0, st, y, stride_y, cb, cr, stride_c, width, height, wavesRowL, nL, wavesRowC, perC,
                       edge_ver, edge_hor, qp_luma, qp_chroma, cfg, nL + nC, xcd);
    VVC_LAUNCH_CHECK();
    return VVCGPU_OK;
  }
  hipLaunchKernelGGL(deblock_picture_kernel, dim3(vvc_xcd_grid2(nLuma, nLuma + nChroma, xcd)), dim3(128), 0, st, y, stride_y, cb, cr, stride_c, width, height, glx, nLuma, gcx, gcy,
                     edge_ver, edge_hor, qp_luma, qp_chroma, cfg, nLuma + nChroma, xcd);
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}
