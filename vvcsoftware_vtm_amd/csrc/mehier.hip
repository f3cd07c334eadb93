// mehier.hip -- hierarchical integer motion search for gfx950: the step-5 raster of xTZSearch AND the +-4 full search of xPatternSearch for every
// 16x16, 32x32 and 64x64 block of a block grid in ONE launch, every SAD computed once.
//
// Reference behaviour reproduced (bit-exact, per block identical to vvcgpu_sad_search on that block):
//   raster stage of InterSearch::xTZSearch                       EncoderLib/InterSearch.cpp:2159-2169 (visiting order y outer, x inner, strict '<')
//   InterSearch::xPatternSearch (bi-prediction refinement +-4)   EncoderLib/InterSearch.cpp:1886-1935
//   RdCost::xGetSAD* with row sub-sampling                       CommonLib/RdCost.cpp:450-1000, :258-283 (subShiftMode 2)
//   RdCost::getCostOfVectorWithPredictor                         CommonLib/RdCost.h:172-199
//
// Why one launch.  The three block sizes of a CU tree are searched over the same displacement grid around the same predictor, and the SAD of a
// 32x32 (64x64) block at a displacement is the exact sum of the SADs of its four (sixteen) 16x16 sub-blocks at that displacement -- same rows
// under 2:1 row sub-sampling (rows 0, 2, .. of the large block are rows 0, 2, .. of its sub-blocks), and the final `<< subShift` distributes over the
// sum.  Searching the sizes independently (three raster launches + three +-4 launches, round 3) issues the v_sad_u16 stream three times;
// here it is issued once and the larger blocks cost additions.
//
// Design.
//   * One 1024-thread workgroup owns a 64x64 SUPER-BLOCK: its search window (64 + 2 R)^2 samples (R = 95: 254 x 254, 150 KB at the conflict-free
//     row pitch of 148 dwords) is staged ONCE in LDS, biased by 0x8000 (v_sad_u16 is then exact for any int16).  One workgroup per CU.
//   * The SAD loop is the QUAD form of the strip kernels (raster_dev.h): a lane owns four consecutive raster columns (0 / 5 / 10 / 15 samples
//     into one span of LDS), the original rows are wave-uniform scalar operands from a packed copy (even + odd-shifted layout per chunk-row).
//   * A unit of work = (32x32 quadrant, 64 flattened (raster row, column quad) slots): the wave walks the quadrant's four 16x16 sub-blocks one
//     after the other with the SAME lane -> position map, so the 32x32 SAD of a lane's four positions is a register sum; the 64x64 SAD meets in
//     an LDS surface (one ds_add_u32 per position and quadrant).  39 x 10 quads = 390 slots = 7 slot waves x 4 quadrants = 28 units; the
//     +-4 grid (9 rows x 3 column quads, column step 1 instead of 5: same loop, other template argument) is 4 more units: 32 units = two
//     rounds of the workgroup's 16 waves.  Flattened slots stay bank-conflict free: with the pitch = 20 (mod 64) dwords the ds_read_b64 bank
//     of slot s is 10 s (mod 64), distinct for any 32 consecutive slots.
//   * Arg-min: per block and unit a 32-bit wave minimum of cost (DPP), then the first lane that holds it (lane order = visiting order) -> 64-bit
//     (cost << 24 | visiting index) LDS atomicMin per block; a super-block is finished by exactly ONE workgroup, which writes the final
//     vvcgpu_search_best records: no global atomics, no key memset, no decode launch.
//   * Grids whose last super-block row / column is partial (3840x2160: 2160 = 33 x 64 + 48) mask the missing sub-blocks; the window fill never
//     reads beyond what the per-size searches of the existing blocks read.
#include "common.h"
#include "raster_dev.h"

namespace {

constexpr int MH_PITCH = 148;                          // dwords per window row: = 20 (mod 64), >= (254 + 7 + 7) / 2
constexpr int MH_MAXN = 39;                            // raster positions per axis (search range 96, step 5)
constexpr int MH_MAXSLOTS = 448;                       // 7 slot waves
constexpr unsigned MH_INVALID = 0x30000000u;           // above every valid cost (SAD << 1 < 2^23, lambda * bits < 2^29), below 2^30

struct MhGeom
{
  int n16x, n16y;                                      // 16x16 blocks of the grid
  int nsbx, total;                                     // super-blocks per row, super-blocks in all
  int refX0, refY0;                                    // reference position of the zero vector of the grid origin
  int subShift, hs;                                    // row sub-sampling, sampled rows of a 16x16 block
  int nR, R;                                           // raster positions per axis, raster reach (5 (nR / 2))
  int nD, D;                                           // +-D grid: nD = 2 D + 1 positions per axis (0: none)
  int winBytes;                                        // LDS bytes of the window
};

// original rows packed per 16x16 block: [block][sampled row][even 8 dwords | odd 8 dwords], biased; layouts as r5c_pack_org_kernel (dist.hip)
__global__ __launch_bounds__(256) void mh_pack_org_kernel(const Pel* __restrict__ org, int os, int orgX0, int orgY0, int n16x, int nblocks, int hs, int subShift,
                                                          unsigned* __restrict__ packed)
{
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (size_t)nblocks * hs) return;
  const int b = (int)(gid / (unsigned)hs), row = (int)(gid - (size_t)b * hs);
  const int by = b / n16x, bx = b - by * n16x;
  const Pel* o = org + (size_t)(orgY0 + 16 * by + (row << subShift)) * os + orgX0 + 16 * bx;
  unsigned d[8];
  if ((reinterpret_cast<uintptr_t>(o) & 3) == 0)
  {
    const unsigned* q = reinterpret_cast<const unsigned*>(o);
#pragma unroll
    for (int k = 0; k < 8; k++) d[k] = q[k];
  }
  else
  {
#pragma unroll
    for (int k = 0; k < 8; k++) d[k] = (unsigned)(unsigned short)o[2 * k] | ((unsigned)(unsigned short)o[2 * k + 1] << 16);
  }
  uint4* dst = reinterpret_cast<uint4*>(packed + gid * 16);
  unsigned E[8], O[8];
#pragma unroll
  for (int k = 0; k < 8; k++)
  {
    E[k] = d[k] ^ 0x80008000u;
    O[k] = __builtin_amdgcn_alignbit(d[(k + 1) & 7], d[k], 16) ^ 0x80008000u;          // k < 7: samples (2k+1, 2k+2); k = 7: (15, 0)
  }
  dst[0] = make_uint4(E[0], E[1], E[2], E[3]); dst[1] = make_uint4(E[4], E[5], E[6], E[7]);
  dst[2] = make_uint4(O[0], O[1], O[2], O[3]); dst[3] = make_uint4(O[4], O[5], O[6], O[7]);
}

// arg-min of one block over the wave: kmin = (cost << 2 | candidate) per lane, idx0 = visiting index of the lane's candidate 0
__device__ __forceinline__ void mh_block_min(unsigned kmin, unsigned idx0, unsigned long long* key, int lane)
{
  const unsigned c = kmin >> 2;
  const unsigned km = wave_min_u32(c);
  const unsigned long long hit = __ballot(c == km);
  const int src = __builtin_ctzll(hit);
  const unsigned sel = (unsigned)__builtin_amdgcn_readlane((int)kmin, src) & 3u;
  const unsigned idx = (unsigned)__builtin_amdgcn_readlane((int)idx0, src) + sel;
  if (lane == 0 && km < MH_INVALID) atomicMin(key, ((unsigned long long)km << 24) | idx);
}

template <int STEP>
__device__ __forceinline__ void mh_positions(int OA, const unsigned* __restrict__ oq, unsigned base, int ldsStep, int hs, unsigned (&acc)[4])
{
  if (OA == 0)      r5q_positions<0, STEP>(oq, base, ldsStep, 1, 0, hs, acc);
  else if (OA == 1) r5q_positions<1, STEP>(oq, base, ldsStep, 1, 0, hs, acc);
  else if (OA == 2) r5q_positions<2, STEP>(oq, base, ldsStep, 1, 0, hs, acc);
  else              r5q_positions<3, STEP>(oq, base, ldsStep, 1, 0, hs, acc);
}

// One unit: the four 16x16 sub-blocks of quadrant q for the lane's four positions.  STEP 5: raster slot wave; STEP 1: the +-D grid.
//   base    LDS byte address of the lane's span for sub-block (0, 0) of the quadrant
//   cst[m]  (cost << 2 | m) of the lane's candidate m (MH_INVALID << 2 | m: no candidate)
//   idx0    visiting index of candidate 0
//   keys    LDS: 16 keys of the 16x16 blocks, 4 of the 32x32, 1 of the 64x64 (this grid)
//   surf    LDS: 64x64 partial sums, [slot][4]
template <int STEP>
__device__ __forceinline__ void mh_unit(const unsigned* __restrict__ orgPacked, const MhGeom& g, int OA, unsigned base, const unsigned (&cst)[4], unsigned idx0,
                                        int q, int sbx, int sby, int nsubx, int nsuby, unsigned long long* keys, unsigned* surf, int slot, bool live, int lane)
{
  const int qx = q & 1, qy = q >> 1;
  const int ldsStep = MH_PITCH << g.subShift;
  const int sh = g.subShift + 2;
  unsigned a32[4] = { 0u, 0u, 0u, 0u };
  int nsub = 0;
#pragma unroll 1
  for (int t = 0; t < 4; t++)
  {
    const int tx = 2 * qx + (t & 1), ty = 2 * qy + (t >> 1);
    if (tx >= nsubx || ty >= nsuby) continue;                                   // wave-uniform: sub-block outside the grid
    nsub++;
    const int b16 = (4 * sby + ty) * g.n16x + 4 * sbx + tx;
    const unsigned* oq = orgPacked + (size_t)b16 * 16u * (unsigned)g.hs;
    unsigned acc[4] = { 0u, 0u, 0u, 0u };
    mh_positions<STEP>(OA, oq, base + (unsigned)((t & 1) * 32 + (t >> 1) * 16 * MH_PITCH * 4), ldsStep, g.hs, acc);
    const unsigned k = min(min((acc[0] << sh) + cst[0], (acc[1] << sh) + cst[1]), min((acc[2] << sh) + cst[2], (acc[3] << sh) + cst[3]));
    mh_block_min(k, idx0, &keys[ty * 4 + tx], lane);
    a32[0] += acc[0]; a32[1] += acc[1]; a32[2] += acc[2]; a32[3] += acc[3];
  }
  if (nsub == 4)
  {
    const unsigned k = min(min((a32[0] << sh) + cst[0], (a32[1] << sh) + cst[1]), min((a32[2] << sh) + cst[2], (a32[3] << sh) + cst[3]));
    mh_block_min(k, idx0, &keys[16 + q], lane);
    if (nsubx == 4 && nsuby == 4 && live)
    {
#pragma unroll
      for (int m = 0; m < 4; m++) atomicAdd(&surf[slot * 4 + m], a32[m]);
    }
  }
}

__global__ __launch_bounds__(1024) void me_hier_kernel(const unsigned* __restrict__ orgPacked, const Pel* __restrict__ ref, int rs, MhGeom g, vvcgpu_mvcost mv,
                                                       vvcgpu_search_best* __restrict__ r16, vvcgpu_search_best* __restrict__ r32, vvcgpu_search_best* __restrict__ r64,
                                                       vvcgpu_search_best* __restrict__ d16, vvcgpu_search_best* __restrict__ d32, vvcgpu_search_best* __restrict__ d64)
{
  extern __shared__ __align__(16) unsigned refL[];
  __shared__ unsigned long long keys[42];                                       // raster: 16 + 4 + 1, then the same for the +-D grid
  __shared__ unsigned costTab[R5C_COST_N];                                      // lambda * bits, truncated (host: below 2^29)
  __shared__ unsigned char bitsRX[MH_MAXN + 1], bitsRY[MH_MAXN + 1], bitsDX[12], bitsDY[12];
  __shared__ unsigned surfD[64 * 4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int chunk = (g.total + 7) >> 3;                                          // XCD-aware order: every XCD gets a contiguous run of super-blocks
  const int item = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
  if (item >= g.total) return;
  const int sby = item / g.nsbx, sbx = item - sby * g.nsbx;
  const int nsubx = min(4, g.n16x - 4 * sbx), nsuby = min(4, g.n16y - 4 * sby);
  unsigned* surf = refL + (g.winBytes >> 2);                                     // [MH_MAXSLOTS][4]

  const int winCols = (g.nR - 1) * 5 + 16 * nsubx, winRows = (g.nR - 1) * 5 + 16 * nsuby - (1 << g.subShift) + 1;
  const ptrdiff_t winOff = (ptrdiff_t)(g.refY0 + 64 * sby - g.R) * rs + g.refX0 + 64 * sbx - g.R;
  const int off = (int)(winOff & 7);
  fill_window_cols<8>(refL, reinterpret_cast<const uint4*>(ref + (winOff - off)), rs >> 3, winRows, MH_PITCH, ((winCols - 1 + off) >> 3) + 1, tid, (int)blockDim.x);
  for (int n = tid; n < R5C_COST_N; n += (int)blockDim.x) costTab[n] = (unsigned)(unsigned long long)(mv.lambda * (double)n);
  if (tid < 42) keys[tid] = ~0ull;
  if (tid < 256) surfD[tid] = 0u;
  for (int n = tid; n < MH_MAXSLOTS * 4; n += (int)blockDim.x) surf[n] = 0u;
  if (tid < 2 * g.nR)
  {
    const int n = tid < g.nR ? tid : tid - g.nR;
    const int v = tid < g.nR ? (((-g.R + 5 * n) << mv.cost_scale) - mv.pred_hor) : (((-g.R + 5 * n) << mv.cost_scale) - mv.pred_ver);
    (tid < g.nR ? bitsRX : bitsRY)[n] = (unsigned char)expgolomb_bits(v >> mv.imv_shift);
  }
  else if (tid >= 128 && tid < 128 + 2 * g.nD)
  {
    const int t = tid - 128, n = t < g.nD ? t : t - g.nD;
    const int v = t < g.nD ? (((-g.D + n) << mv.cost_scale) - mv.pred_hor) : (((-g.D + n) << mv.cost_scale) - mv.pred_ver);
    (t < g.nD ? bitsDX : bitsDY)[n] = (unsigned char)expgolomb_bits(v >> mv.imv_shift);
  }
  __syncthreads();

  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = (int)(blockDim.x >> 6);
  const unsigned ldsBase = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)refL;
  const int nq = (g.nR + 3) >> 2, nslots = g.nR * nq, nsw = (nslots + 63) >> 6;
  const int nqd = (g.nD + 3) >> 2, nslotsD = g.nD * nqd;
  const int nunits = 4 * nsw + (g.nD ? 4 : 0);
  for (int u = wave; u < nunits; u += nwaves)
  {
    if (u < 4 * nsw)
    {
      const int sw = u >> 2, q = u & 3;
      const int s = sw * 64 + lane;
      const bool live = s < nslots;
      const int sc = live ? s : 0;                                                // dead lanes re-read a live lane's address (broadcast)
      const int jj = sc / nq, i0 = 4 * (sc - jj * nq);
      const int cx = 5 * i0 + off;
      const unsigned base = ldsBase + (unsigned)(2 * (cx >> 2) + jj * 5 * MH_PITCH) * 4u + (unsigned)((q & 1) * 64 + (q >> 1) * 32 * MH_PITCH * 4);
      const unsigned by = bitsRY[jj];
      unsigned cst[4];
#pragma unroll
      for (int m = 0; m < 4; m++)
      {
        const bool in = live && i0 + m < g.nR;
        cst[m] = ((in ? costTab[bitsRX[in ? i0 + m : 0] + by] : MH_INVALID) << 2) | (unsigned)m;
      }
      mh_unit<5>(orgPacked, g, off & 3, base, cst, (unsigned)(jj * g.nR + i0), q, sbx, sby, nsubx, nsuby, keys, surf, s, live, lane);
    }
    else
    {
      const int q = u - 4 * nsw;
      const bool live = lane < nslotsD;
      const int sc = live ? lane : 0;
      const int jj = sc / nqd, i0 = 4 * (sc - jj * nqd);
      const int cx = g.R - g.D + i0 + off;
      const unsigned base = ldsBase + (unsigned)(2 * (cx >> 2) + (g.R - g.D + jj) * MH_PITCH) * 4u + (unsigned)((q & 1) * 64 + (q >> 1) * 32 * MH_PITCH * 4);
      const unsigned by = bitsDY[jj];
      unsigned cst[4];
#pragma unroll
      for (int m = 0; m < 4; m++)
      {
        const bool in = live && i0 + m < g.nD;
        cst[m] = ((in ? costTab[bitsDX[in ? i0 + m : 0] + by] : MH_INVALID) << 2) | (unsigned)m;
      }
      mh_unit<1>(orgPacked, g, cx & 3, base, cst, (unsigned)(jj * g.nD + i0), q, sbx, sby, nsubx, nsuby, keys + 21, surfD, lane, live, lane);
    }
  }
  __syncthreads();

  // 64x64: cost + arg-min over the LDS surface (slot order = visiting order)
  if (nsubx == 4 && nsuby == 4)
  {
    const int sh = g.subShift + 2;
    if (wave < nsw)
    {
      const int s = tid;
      const bool live = s < nslots;
      const int sc = live ? s : 0;
      const int jj = sc / nq, i0 = 4 * (sc - jj * nq);
      const unsigned by = bitsRY[jj];
      unsigned k = 0xFFFFFFFFu;
#pragma unroll
      for (int m = 0; m < 4; m++)
      {
        const bool in = live && i0 + m < g.nR;
        const unsigned c = ((in ? costTab[bitsRX[in ? i0 + m : 0] + by] : MH_INVALID) << 2) | (unsigned)m;
        k = min(k, (surf[sc * 4 + m] << sh) + c);
      }
      mh_block_min(k, (unsigned)(jj * g.nR + i0), &keys[20], lane);
    }
    else if (wave == 8 && g.nD)
    {
      const bool live = lane < nslotsD;
      const int sc = live ? lane : 0;
      const int jj = sc / nqd, i0 = 4 * (sc - jj * nqd);
      const unsigned by = bitsDY[jj];
      unsigned k = 0xFFFFFFFFu;
#pragma unroll
      for (int m = 0; m < 4; m++)
      {
        const bool in = live && i0 + m < g.nD;
        const unsigned c = ((in ? costTab[bitsDX[in ? i0 + m : 0] + by] : MH_INVALID) << 2) | (unsigned)m;
        k = min(k, (surfD[sc * 4 + m] << sh) + c);
      }
      mh_block_min(k, (unsigned)(jj * g.nD + i0), &keys[41], lane);
    }
  }
  __syncthreads();

  // final records: thread t < 21: raster result of block t of the super-block, 21 <= t < 42: the +-D grid
  if (tid < 42)
  {
    const int grid = tid >= 21, t = tid - 21 * grid;
    if (grid && !g.nD) return;
    vvcgpu_search_best* out; int bidx; bool exists;
    if (t < 16)      { const int tx = t & 3, ty = t >> 2; exists = tx < nsubx && ty < nsuby; bidx = (4 * sby + ty) * g.n16x + 4 * sbx + tx; out = grid ? d16 : r16; }
    else if (t < 20) { const int qx = (t - 16) & 1, qy = (t - 16) >> 1; exists = 2 * qx + 2 <= nsubx && 2 * qy + 2 <= nsuby; bidx = (2 * sby + qy) * (g.n16x >> 1) + 2 * sbx + qx; out = grid ? d32 : r32; }
    else             { exists = nsubx == 4 && nsuby == 4; bidx = sby * (g.n16x >> 2) + sbx; out = grid ? d64 : r64; }
    if (!exists || !out) return;
    const unsigned long long key = keys[tid];
    const int idx = (int)(key & 0xFFFFFFu);
    const unsigned long long cost = key >> 24;
    const int n = grid ? g.nD : g.nR, step = grid ? 1 : 5, p0 = grid ? -g.D : -g.R;
    const int j = idx / n, i = idx - j * n;
    const int x = p0 + i * step, y = p0 + j * step;
    const unsigned bits = expgolomb_bits(((x << mv.cost_scale) - mv.pred_hor) >> mv.imv_shift) + expgolomb_bits(((y << mv.cost_scale) - mv.pred_ver) >> mv.imv_shift);
    vvcgpu_search_best r;
    r.x = x; r.y = y; r.cost = cost; r.sad = cost - (unsigned long long)(mv.lambda * (double)bits);
    out[bidx] = r;
  }
}

}  // namespace

extern "C" {

int vvcgpu_me_hier_search(const vvc_pel* org, int org_stride, const vvc_pel* ref, int ref_stride, const vvcgpu_me_hier_cfg* cfg_host,
                          const vvcgpu_mvcost* mvcost_host, vvcgpu_search_best* const* raster_best, vvcgpu_search_best* const* dense_best, void* stream)
{
  VVC_CHECK_ARG(org && ref && cfg_host && mvcost_host && raster_best, "me_hier_search: null pointer");
  const vvcgpu_me_hier_cfg c = *cfg_host;
  VVC_CHECK_ARG(c.n16x >= 1 && c.n16y >= 1 && c.n16x <= 4096 && c.n16y <= 4096, "me_hier_search: grid %d x %d", c.n16x, c.n16y);
  VVC_CHECK_ARG(c.raster_step == 5 && c.raster_range >= 5 && c.dense_range >= 0, "me_hier_search: raster step %d range %d, dense range %d", c.raster_step, c.raster_range, c.dense_range);
  VVC_CHECK_ARG(raster_best[0] && (c.n16x < 2 || c.n16y < 2 || raster_best[1]) && (c.n16x < 4 || c.n16y < 4 || raster_best[2]), "me_hier_search: raster result arrays");
  VVC_CHECK_ARG(c.dense_range == 0 || (dense_best && dense_best[0] && (c.n16x < 2 || c.n16y < 2 || dense_best[1]) && (c.n16x < 4 || c.n16y < 4 || dense_best[2])),
                "me_hier_search: dense result arrays");
  const int nR = 2 * (c.raster_range / 5) + 1, R = 5 * (c.raster_range / 5), nD = c.dense_range ? 2 * c.dense_range + 1 : 0;
  // outside the kernel's shape: the caller takes the per-size searches (vvcgpu_sad_search), which are the same results
  if (nR > MH_MAXN || nD > 9 || c.dense_range > R || c.sub_shift < 0 || c.sub_shift > 1 || (org_stride & 1) || (ref_stride & 7) || ((uintptr_t)org & 3) || ((uintptr_t)ref & 15) ||
      !(mvcost_host->lambda >= 0.0 && mvcost_host->lambda < 4.0e6))
  {
    vvcgpu_set_error("me_hier_search: shape outside the hierarchical kernel (raster +-%d, +-%d grid, sub_shift %d, strides %d / %d, lambda %g)",
                     c.raster_range, c.dense_range, c.sub_shift, org_stride, ref_stride, mvcost_host->lambda);
    return VVCGPU_E_UNSUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  MhGeom g;
  g.n16x = c.n16x; g.n16y = c.n16y; g.nsbx = cdiv(c.n16x, 4); g.total = g.nsbx * cdiv(c.n16y, 4);
  g.refX0 = c.ref_x; g.refY0 = c.ref_y; g.subShift = c.sub_shift; g.hs = 16 >> c.sub_shift;
  g.nR = nR; g.R = R; g.nD = nD; g.D = c.dense_range;
  const int winRowsMax = (nR - 1) * 5 + 64;
  g.winBytes = winRowsMax * MH_PITCH * 4;
  const size_t smem = (size_t)g.winBytes + MH_MAXSLOTS * 4 * sizeof(unsigned);
  const int nblocks = c.n16x * c.n16y;
  unsigned* packed = static_cast<unsigned*>(vvcgpu_scratch(st, (size_t)nblocks * g.hs * 16 * sizeof(unsigned)));
  if (!packed) return VVCGPU_E_DEVICE;
  hipLaunchKernelGGL(mh_pack_org_kernel, dim3((unsigned)(((size_t)nblocks * g.hs + 255) / 256)), dim3(256), 0, st, org, org_stride, c.org_x, c.org_y, c.n16x, nblocks, g.hs, c.sub_shift, packed);
  VVC_LAUNCH_CHECK();
  VVC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(me_hier_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL(me_hier_kernel, dim3(cdiv(g.total, 8) * 8), dim3(1024), smem, st, packed, ref, ref_stride, g, *mvcost_host,
                     raster_best[0], raster_best[1], raster_best[2], dense_best ? dense_best[0] : nullptr, dense_best ? dense_best[1] : nullptr, dense_best ? dense_best[2] : nullptr);
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}

}  // extern "C"
