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
//   * One 1024-thread workgroup owns a 64x64 SUPER-BLOCK at a time: its search window (64 + 2 R)^2 samples (R = 95: 254 x 254, 150 KB at the
//     conflict-free row pitch of 148 dwords) is staged ONCE in LDS, biased by 0x8000 (v_sad_u16 is then exact for any int16).  One workgroup per CU,
//     PERSISTENT (round 5): it walks a run of horizontally consecutive super-blocks of its XCD's chunk and SLIDES the window -- LDS is addressed
//     linearly, a step to the right moves every address on by 32 dwords, the 64 new columns of a row land behind its old end (row padding + the dead
//     first columns of the row below) and are requested one super-block ahead; tables and lane descriptors are set up once per workgroup.
//   * The SAD loop is the QUAD form of the strip kernels (raster_dev.h): a lane owns four consecutive raster columns (0 / 5 / 10 / 15 samples
//     into one span of LDS), the original rows are wave-uniform scalar operands from a packed copy (even + odd-shifted layout per chunk-row).
//   * A unit of work = (32x32 quadrant, 64 flattened (raster row, column quad) slots): the wave walks the quadrant's four 16x16 sub-blocks one
//     after the other with the SAME lane -> position map, so the 32x32 SAD of a lane's four positions is a register sum; the 64x64 SAD meets in
//     an LDS surface (one ds_add_u32 per position and quadrant).  39 x 10 quads = 390 slots = 7 slot waves x 4 quadrants = 28 units = two rounds of
//     the workgroup's 16 waves (twelve waves take two units of the same quadrant, four take one); the +-4 grid rides as 54 dense lanes in the idle
//     lanes of the last slot wave (see MhLane).  Issue priority follows a wave's progress inside its unit (s_setprio), so the waves of a SIMD finish together.  Flattened slots stay bank-conflict free: with the pitch = 20 (mod 64) dwords the ds_read_b64 bank
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
constexpr int MH_MAXSLIDE = 7;                         // slides of the window between two full fills (1 KB of slack behind the window)
constexpr int MH_MAXDL = 96;                           // lanes of the +-D grid (9 rows x at most 9 spans)
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
  // the spans of the +-D grid (host-built): column x is candidate m = (R + x) mod 4 of the span that starts (R + x - 5 m) / 4 quads into the window row
  int dlCount; short dlKey[12]; signed char dlX[48];   // per span: start in quads, grid column of each candidate (-128: none)
  unsigned magicNq, magicDl;                           // ceil(65536 / n): v / n == (v * magic) >> 16 for the slot indices (< 448, n <= 10)
};

// original rows packed per 16x16 block: [block][sampled row][even 8 dwords | odd 8 dwords], biased; layouts as r5c_pack_org_kernel (dist.hip).
// Packed by a launch of its own in front of the search (7.5 us per 4K picture).  Round 6 measured the search kernel packing the rows of its super-blocks
// ITSELF (first super-block of a run by every thread, the rest by the four waves without a second-round unit, read back by explicit scalar loads): exact,
// and 6 us faster when the entry is timed alone in a loop (178 against 185 us) -- but 220 against 192 + 7.5 us inside the picture's workload, where the rows come
// from HBM and the stores must be complete in front of every super-block's barrier; removed (docs/OPTIMISATION_LOG.md).
// id = quarter q (id & 3) of record (block of the super-block, sampled row): four lanes per record, a wave's store is 1 KB of consecutive bytes.
struct MhPackItem { unsigned e[5]; unsigned* dst; int q; };                     // the quarter's four dwords and the one behind them (wrapping: dword 8 = dword 0)
__device__ __forceinline__ void mh_pack_load(const Pel* __restrict__ org, int os, int orgX0, int orgY0, int n16x, int hs, int subShift, int sbx, int sby, int pnx,
                                             int id, unsigned* __restrict__ packed, MhPackItem& it)
{
  const int q = id & 3, rec = id >> 2;
  const int blk = rec / hs, row = rec - blk * hs;
  const int ty = blk / pnx, tx = blk - ty * pnx;
  const int by = 4 * sby + ty, bx = 4 * sbx + tx, b = by * n16x + bx;
  const Pel* o = org + (size_t)(orgY0 + 16 * by + (row << subShift)) * os + orgX0 + 16 * bx;
  // a lane reads the half row of its quarter (16 bytes) and one dword more (the odd-shifted quarters pair every dword with its successor), not the whole
  // row: half the load instructions of the first form
  const int h0 = 4 * (q & 1), hx = (h0 + 4) & 7;
  if ((reinterpret_cast<uintptr_t>(o) & 3) == 0)
  {
    const unsigned* p = reinterpret_cast<const unsigned*>(o);
#pragma unroll
    for (int k = 0; k < 4; k++) it.e[k] = p[h0 + k];
    it.e[4] = p[hx];
  }
  else
  {
#pragma unroll
    for (int k = 0; k < 4; k++) it.e[k] = (unsigned)(unsigned short)o[2 * (h0 + k)] | ((unsigned)(unsigned short)o[2 * (h0 + k) + 1] << 16);
    it.e[4] = (unsigned)(unsigned short)o[2 * hx] | ((unsigned)(unsigned short)o[2 * hx + 1] << 16);
  }
  it.q = q;
  it.dst = packed + ((size_t)b * hs + row) * 16 + 4 * q;
}
__device__ __forceinline__ void mh_pack_store(const MhPackItem& it)
{
  // quarter 0 / 1: even dwords 0..3 / 4..7; quarter 2 / 3: the odd-shifted dwords (samples (2k+1, 2k+2); k = 7: (15, 0))
  unsigned v[4];
#pragma unroll
  for (int j = 0; j < 4; j++) v[j] = ((it.q & 2) ? __builtin_amdgcn_alignbit(it.e[j + 1], it.e[j], 16) : it.e[j]) ^ 0x80008000u;
  *reinterpret_cast<uint4*>(it.dst) = make_uint4(v[0], v[1], v[2], v[3]);
}
__global__ __launch_bounds__(256) void mh_pack_org_kernel(const Pel* __restrict__ org, int os, int orgX0, int orgY0, int n16x, int nblocks, int hs, int subShift, unsigned* __restrict__ packed)
{
  const size_t gid4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if ((gid4 >> 2) >= (size_t)nblocks * hs) return;
  const int q = (int)(gid4 & 3), rec = (int)(gid4 >> 2), b = rec / hs, row = rec - b * hs, by = b / n16x, bx = b - by * n16x;
  MhPackItem it;
  mh_pack_load(org, os, orgX0, orgY0, n16x, hs, subShift, bx >> 2, by >> 2, 4, (((by & 3) * 4 + (bx & 3)) * hs + row) * 4 + q, packed, it);
  mh_pack_store(it);
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
  // (explicit ds_min_u64: behind atomicMin the compiler's atomic optimiser elects a lane of the one-lane mask again -- two v_mbcnt, a compare and a branch per fold)
  if (lane == 0 && km < MH_INVALID)
  {
    const unsigned long long v = ((unsigned long long)km << 24) | idx;
    asm volatile("ds_min_u64 %0, %1" :: "v"((unsigned)(size_t)(__attribute__((address_space(3))) unsigned long long*)key), "v"(v) : "memory");
  }
}

// What a lane of a slot wave works on: four candidates 0 / 5 / 10 / 15 samples into one LDS span (the quad loop of raster_dev.h).
//   raster lane : four consecutive columns of one raster row; visiting indices idx0 .. idx0 + 3 (lane order = visiting order)
//   dense lane  : a lane of the +-D grid.  Its columns are 1 apart, not 5, but the span of a lane may start at any multiple of 4 samples: column
//                 x of the grid is candidate m = (R + x) mod 4 of the span that starts (R + x - 5 m) / 4 quads into the row, and x + 5 (when it
//                 is still on the grid) is candidate m + 1 of the same span -- the 9 columns of the +-4 grid are 6 spans (dlKey / dlX, built once
//                 per workgroup), the 81 positions 54 lanes, which fit into the lanes the raster leaves idle in its last slot wave (390 of 448):
//                 the +-4 grid costs no unit of its own.  Candidates are not in visiting order: dense lanes carry explicit indices (7 bits each)
//                 and reduce 64-bit (cost << 24 | index) keys.
struct MhLane { unsigned base; unsigned cst[4]; unsigned idx; int kind; };        // kind 0 dead, 1 raster, 2 dense; idx: idx0 (raster) / 4 x 7 bits (dense)

struct MhTables
{
  const unsigned* costTab; const unsigned char* bitsRX; const unsigned char* bitsRY; const unsigned char* bitsDX; const unsigned char* bitsDY;
  const short* dlKey; const signed char* dlX; int dlCount;
};

__device__ __forceinline__ void mh_lane(int s, const MhGeom& g, const MhTables& T, int off, int nq, int nslots, int ndl, MhLane& L)
{
  L.kind = s < nslots ? 1 : (s - nslots < ndl ? 2 : 0);
  if (L.kind == 2)
  {
    const int e = s - nslots, j = (int)(((unsigned)e * g.magicDl) >> 16), l = e - j * T.dlCount;
    L.base = (unsigned)(2 * (int)T.dlKey[l] + 2 * (off >> 2) + (g.R - g.D + j) * MH_PITCH) * 4u;
    const unsigned by = T.bitsDY[j];
    L.idx = 0;
#pragma unroll
    for (int m = 0; m < 4; m++)
    {
      const int x = T.dlX[l * 4 + m];                                              // -128: the span has no grid column at candidate m
      const bool in = x != -128;
      L.cst[m] = ((in ? T.costTab[T.bitsDX[in ? x + g.D : 0] + by] : MH_INVALID) << 2) | (unsigned)m;
      L.idx |= (unsigned)(in ? j * g.nD + x + g.D : 0) << (7 * m);
    }
    return;
  }
  const int sc = L.kind ? s : 0;                                                   // dead lanes re-read a live lane's address (broadcast)
  const int jj = (int)(((unsigned)sc * g.magicNq) >> 16), i0 = 4 * (sc - jj * nq);
  const int cx = 5 * i0 + off;
  L.base = (unsigned)(2 * (cx >> 2) + jj * 5 * MH_PITCH) * 4u;
  const unsigned by = T.bitsRY[jj];
#pragma unroll
  for (int m = 0; m < 4; m++)
  {
    const bool in = L.kind && i0 + m < g.nR;
    L.cst[m] = ((in ? T.costTab[T.bitsRX[in ? i0 + m : 0] + by] : MH_INVALID) << 2) | (unsigned)m;
  }
  L.idx = (unsigned)(jj * g.nR + i0);
}

// the lane's packed minimum of four sums: raster lanes (cost << 2 | candidate), dense lanes a 64-bit (cost << 24 | visiting index) key
__device__ __forceinline__ unsigned mh_fold32(const unsigned (&a)[4], const MhLane& L, int sh)
{
  return min(min((a[0] << sh) + L.cst[0], (a[1] << sh) + L.cst[1]), min((a[2] << sh) + L.cst[2], (a[3] << sh) + L.cst[3]));
}
__device__ __forceinline__ unsigned long long mh_fold64(const unsigned (&a)[4], const MhLane& L, int sh)
{
  unsigned long long k = ~0ull;
#pragma unroll
  for (int m = 0; m < 4; m++)
  {
    const unsigned c = ((a[m] << sh) + L.cst[m]) >> 2;
    const unsigned long long km = c < MH_INVALID ? ((unsigned long long)c << 24) | ((L.idx >> (7 * m)) & 127u) : ~0ull;
    k = km < k ? km : k;
  }
  return k;
}
__device__ __forceinline__ void mh_block_min64(unsigned long long k, unsigned long long* key, int lane)
{
  const unsigned long long km = wave_min_u64(k);
  if (lane == 0 && km != ~0ull) asm volatile("ds_min_u64 %0, %1" :: "v"((unsigned)(size_t)(__attribute__((address_space(3))) unsigned long long*)key), "v"(km) : "memory");
}

// The sub-blocks of a quadrant that exist (wave-uniform), walked as ONE pipeline of unrolled stage bodies (raster_dev.h: r5q_positions_fixed; NST sampled
// rows per sub-block = 16 >> subShift, MH_PITCH << subShift dwords apart): the window reads and original rows of the next sub-block's first stage are
// requested before the last stage of the current one is summed, so the arg-min folds between two sub-blocks run with loads in flight.
struct MhWalk { const unsigned* orgPacked; unsigned base; int qx, qy, sbx, sby, nsubx, nsuby, n16x, sh; bool waveHasDense; int lane; };
template <int OA, int NST>
__device__ __forceinline__ int mh_walk(const MhWalk& w, const MhLane& L, unsigned long long* keys, unsigned (&a32)[4])
{
  constexpr int LSTEP = MH_PITCH * (16 / NST);
  auto exists = [&](int t) { return 2 * w.qx + (t & 1) < w.nsubx && 2 * w.qy + (t >> 1) < w.nsuby; };
  auto orgOf = [&](int t) { return w.orgPacked + (size_t)((4 * w.sby + 2 * w.qy + (t >> 1)) * w.n16x + 4 * w.sbx + 2 * w.qx + (t & 1)) * (unsigned)(16 * NST); };
  auto baseOf = [&](int t) { return w.base + (unsigned)((t & 1) * 32 + (t >> 1) * 16 * MH_PITCH * 4); };
  int t = 0, nsub = 0;
  while (t < 4 && !exists(t)) t++;
  R5qStageS A, B;
  if (t < 4) r5q_issue_at<OA, 0>(A, orgOf(t), baseOf(t));
#pragma unroll 1
  while (t < 4)
  {
    int tn = t + 1;
    while (tn < 4 && !exists(tn)) tn++;
    const int tx = 2 * w.qx + (t & 1), ty = 2 * w.qy + (t >> 1);
    // issue priority by progress: the arbiter takes the oldest wave first, so the waves of a SIMD finish one after the other and the last one runs
    // alone (per-unit stamps: 45 k, 49 k, 55 k cycles); a wave that is ahead in its unit yields to the ones behind
    if (nsub == 0) __builtin_amdgcn_s_setprio(3); else if (nsub == 1) __builtin_amdgcn_s_setprio(2); else if (nsub == 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
    nsub++;
    unsigned acc[4] = { 0u, 0u, 0u, 0u };
    const int tnc = tn < 4 ? tn : t;
    r5q_positions_fixed<OA, NST, LSTEP>(orgOf(t), baseOf(t), orgOf(tnc), baseOf(tnc), A, B, acc);
    mh_block_min(L.kind == 1 ? mh_fold32(acc, L, w.sh) : 0xFFFFFFFFu, L.idx, &keys[ty * 4 + tx], w.lane);
    if (w.waveHasDense) mh_block_min64(L.kind == 2 ? mh_fold64(acc, L, w.sh) : ~0ull, &keys[21 + ty * 4 + tx], w.lane);
    a32[0] += acc[0]; a32[1] += acc[1]; a32[2] += acc[2]; a32[3] += acc[3];
    t = tn;
  }
  return nsub;
}

// One unit: the four 16x16 sub-blocks of quadrant q for the lanes of one slot wave.
//   keys   LDS: raster keys of the 16x16 blocks (16), the 32x32 (4), the 64x64 (1), then the same 21 for the +-D grid
//   surf / surfD  LDS: 64x64 partial sums of the raster slots / the dense lanes, [slot][4]
__device__ __forceinline__ void mh_unit(const unsigned* orgPacked, const MhGeom& g, int OA, unsigned qbase, const MhLane& L, bool waveHasDense,
                                        int q, int sbx, int sby, int nsubx, int nsuby, unsigned long long* keys, unsigned* surf, unsigned* surfD, int* arrive, int s, int nslots, int lane)
{
  const int qx = q & 1, qy = q >> 1;
  const int sh = g.subShift + 2;
  const unsigned base = qbase + L.base;
  unsigned a32[4] = { 0u, 0u, 0u, 0u };
  int nsub = 0;
  {
    const int OAhs = OA | (g.hs == 8 ? 0 : 4);
    const MhWalk w = { orgPacked, base, qx, qy, sbx, sby, nsubx, nsuby, g.n16x, sh, waveHasDense, lane };
    if (OAhs == 0)      nsub = mh_walk<0, 8>(w, L, keys, a32);
    else if (OAhs == 1) nsub = mh_walk<1, 8>(w, L, keys, a32);
    else if (OAhs == 2) nsub = mh_walk<2, 8>(w, L, keys, a32);
    else if (OAhs == 3) nsub = mh_walk<3, 8>(w, L, keys, a32);
    else if (OAhs == 4) nsub = mh_walk<0, 16>(w, L, keys, a32);
    else if (OAhs == 5) nsub = mh_walk<1, 16>(w, L, keys, a32);
    else if (OAhs == 6) nsub = mh_walk<2, 16>(w, L, keys, a32);
    else                nsub = mh_walk<3, 16>(w, L, keys, a32);
  }
  __builtin_amdgcn_s_setprio(0);                                                  // (a unit that skipped its last sub-blocks would keep a raised priority through the record write and the next window slide)
  if (nsub == 4)
  {
    mh_block_min(L.kind == 1 ? mh_fold32(a32, L, sh) : 0xFFFFFFFFu, L.idx, &keys[16 + q], lane);
    if (waveHasDense) mh_block_min64(L.kind == 2 ? mh_fold64(a32, L, sh) : ~0ull, &keys[21 + 16 + q], lane);
    if (nsubx == 4 && nsuby == 4)
    {
      // 64x64: the four quadrants of a slot wave add their sums to the LDS surface; the one that arrives last folds the totals (no separate pass behind
      // a barrier).  A wave's LDS operations execute in program order, so a quadrant's adds are in the surface before its ticket: the wave that draws
      // ticket 3 reads all four contributions.
      unsigned* dst = L.kind == 2 ? surfD + (s - nslots) * 4 : surf + (L.kind ? s : 0) * 4;
      if (L.kind)
      {
#pragma unroll
        for (int m = 0; m < 4; m++) atomicAdd(&dst[m], a32[m]);
      }
      int ticket = 0;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");                       // (a wait for the adds above: the data is in LDS before the ticket)
      if (lane == 0) ticket = atomicAdd(arrive, 1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      ticket = __builtin_amdgcn_readfirstlane(ticket);
      if (ticket == 3)
      {
        unsigned a[4];
#pragma unroll
        for (int m = 0; m < 4; m++) a[m] = L.kind ? dst[m] : 0u;
        mh_block_min(L.kind == 1 ? mh_fold32(a, L, sh) : 0xFFFFFFFFu, L.idx, &keys[20], lane);
        if (waveHasDense) mh_block_min64(L.kind == 2 ? mh_fold64(a, L, sh) : ~0ull, &keys[41], lane);
      }
    }
  }
}

__global__ __launch_bounds__(1024) void me_hier_kernel(const unsigned* orgPacked, const Pel* __restrict__ ref, int rs, MhGeom g, vvcgpu_mvcost mv,
                                                       vvcgpu_search_best* __restrict__ r16, vvcgpu_search_best* __restrict__ r32, vvcgpu_search_best* __restrict__ r64,
                                                       vvcgpu_search_best* __restrict__ d16, vvcgpu_search_best* __restrict__ d32, vvcgpu_search_best* __restrict__ d64,
                                                       unsigned long long* __restrict__ diagArg)
{
#ifdef MH_DIAG
  unsigned long long* const diag = diagArg;
#endif
  extern __shared__ __align__(16) unsigned refL[];
  __shared__ unsigned long long keys[42];                                       // raster: 16 + 4 + 1, then the same for the +-D grid
  __shared__ unsigned costTab[R5C_COST_N];                                      // lambda * bits, truncated (host: below 2^29)
  __shared__ unsigned char bitsRX[MH_MAXN + 1], bitsRY[MH_MAXN + 1], bitsDX[12], bitsDY[12];
  __shared__ short dlKey[12];                                                  // dense spans of a grid row: start of the span in quads of the window row
  __shared__ signed char dlX[12 * 4];                                          // ... and the grid column of each of its four candidates (-128: none)
  __shared__ unsigned surfD[MH_MAXDL * 4];
  __shared__ int arrive[8];                                                    // per slot wave: quadrants that have added their 32x32 sums to the 64x64 surface
  // what only the 42 record writers of a super-block need lives in LDS, not in scalar registers across the search (the kernel holds its arguments
  // in ~80 scalar registers; inside the super-block loop they spilled)
  __shared__ vvcgpu_search_best* outPtr[6];
  __shared__ vvcgpu_mvcost mvL;
  const int tid = threadIdx.x, lane = tid & 63;
  const int chunk = (g.total + 7) >> 3;                                          // XCD-aware order: every XCD gets a contiguous run of super-blocks
  // what does not depend on the super-block, once per workgroup: the rate tables, the spans of the +-D grid, and (below) what every lane of a
  // slot wave works on -- the window of every super-block starts at the same offset from a 16-byte boundary (64 columns / 64 rows of an 8-sample
  // aligned stride apart)
  for (int n = tid; n < R5C_COST_N; n += (int)blockDim.x) costTab[n] = (unsigned)(unsigned long long)(mv.lambda * (double)n);
  if (tid == 700) { outPtr[0] = r16; outPtr[1] = r32; outPtr[2] = r64; outPtr[3] = d16; outPtr[4] = d32; outPtr[5] = d64; mvL = mv; }
  if (tid >= 512 && tid < 512 + 12) dlKey[tid - 512] = g.dlKey[tid - 512];
  if (tid >= 576 && tid < 576 + 48) dlX[tid - 576] = g.dlX[tid - 576];
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
  const int off = (int)(((ptrdiff_t)(g.refY0 - g.R) * rs + g.refX0 - g.R) & 7);
  const int nq = (g.nR + 3) >> 2, nslots = g.nR * nq, ndl = g.nD * g.dlCount, nsw = (nslots + ndl + 63) >> 6;
  // the units of this wave: u = wave, wave + 16 (28 units: 7 slot waves x 4 quadrants; a wave keeps its quadrant).  The last slot wave goes FIRST:
  // its dense lanes (rows one window row apart) meet LDS bank conflicts the raster lanes do not have, so its four units are the slowest; started
  // first they run beside twelve others instead of ending the workgroup alone
  MhLane LU[2];
  {
    MhTables T = { costTab, bitsRX, bitsRY, bitsDX, bitsDY, dlKey, dlX, g.dlCount };
#pragma unroll
    for (int r = 0; r < 2; r++)
    {
      const int u = wave + r * nwaves, sw = nsw - 1 - (u >> 2);
      LU[r].kind = 0; LU[r].base = 0; LU[r].idx = 0; LU[r].cst[0] = LU[r].cst[1] = LU[r].cst[2] = LU[r].cst[3] = MH_INVALID << 2;
      if (u < 4 * nsw) mh_lane(sw * 64 + lane, g, T, off, nq, nslots, ndl, LU[r]);
    }
  }
  // persistent: workgroup b walks a contiguous run of super-blocks of its XCD's chunk -- a workgroup owns a CU (150 KB of LDS), and a new workgroup per
  // super-block pays the dispatch of sixteen waves and the argument loads with nothing else running on the CU.  SLIDING WINDOW: the next super-block of
  // a run is the right-hand neighbour, whose window shares 190 of its 254 columns; LDS is addressed linearly, so the shared columns stay where they are
  // when every address moves on by 32 dwords (64 samples), and the 64 new columns of a row land behind its old end: in the row padding and in the
  // first -- now dead -- 64 columns of the row below (the last row runs into 1 KB of slack).  A slide loads 8 of 33 quads per row.
  const int perX = (int)(gridDim.x >> 3), runLen = (chunk + perX - 1) / perX, kk0 = (int)(blockIdx.x >> 3) * runLen;
  int slide = 0, pfQ0 = 0;
  bool pfHave = false;
  uint4 pf[2];
  // (row, column) of the run's super-blocks: one division per run, then a step to the right with wrap (the items of a run are consecutive)
  int sbyRun = ((int)(blockIdx.x & 7) * chunk + kk0) / g.nsbx, sbxRun = ((int)(blockIdx.x & 7) * chunk + kk0) - sbyRun * g.nsbx;
  for (int kk = kk0; kk < kk0 + runLen && kk < chunk; kk++)
  {
  const int item = (int)(blockIdx.x & 7) * chunk + kk;
  if (item >= g.total) break;
  const int sby = sbyRun, sbx = sbxRun;
  if (++sbxRun == g.nsbx) { sbxRun = 0; sbyRun++; }
  const int nsubx = min(4, g.n16x - 4 * sbx), nsuby = min(4, g.n16y - 4 * sby);
  // -DMH_DIAG + VVCGPU_MH_DIAG: core-clock stamps of one workgroup's phases (start, window staged, every unit's end, units done).  A build without the switch
  // holds no stamp code: the pointer, the two conditions and the item compares were scalar registers the super-block loop spilled (49 lane reloads per super-block)
#ifdef MH_DIAG
  const bool stamp = diag && item == (g.total >> 1) + 3;                            // (the fourth super-block of a run: a slide)
  const int stampK = item == (g.total >> 2) ? 0 : item == (g.total >> 3) ? 1 : item == 3 * (g.total >> 2) ? 2 : item == 5 * (g.total >> 3) ? 3 : -1;   // four more workgroups: phase ends only
#else
  constexpr bool stamp = false;
  constexpr int stampK = -1;
  unsigned long long* const diag = nullptr;
#endif
  if (diag && stampK >= 0 && tid == 0) diag[40 + 0 * 4 + stampK] = __builtin_amdgcn_s_memtime();
  if (stamp && tid == 0) diag[0] = __builtin_amdgcn_s_memtime();
  unsigned* surf = refL + (g.winBytes >> 2);                                     // [MH_MAXSLOTS][4]

  const int winCols = (g.nR - 1) * 5 + 16 * nsubx, winRows = (g.nR - 1) * 5 + 16 * nsuby - (1 << g.subShift) + 1;
  const ptrdiff_t winOff = (ptrdiff_t)(g.refY0 + 64 * sby - g.R) * rs + g.refX0 + 64 * sbx - g.R;
  const int nQuads = ((winCols - 1 + off) >> 3) + 1;
  const int rsQ = rs >> 3;
  // the quads (row, q) of a slide that this thread moves: items tid and tid + 1024 of nNew x winRows
  auto slideItem = [&](int it, int q0, int& row, int& q) { row = it >> 3; q = q0 + (it & 7); };     // (a slide is always 8 quads per row: both windows full width)
  if (pfHave)                                                                    // requested while the previous super-block was searched
  {
    slide++;
    unsigned* winL = refL + slide * 32;
#pragma unroll
    for (int u = 0; u < 2; u++)
    {
      const int it = tid + u * 1024;
      if (it < 8 * winRows)
      {
        int row, q;
        slideItem(it, pfQ0, row, q);
        uint2* d = reinterpret_cast<uint2*>(winL + row * MH_PITCH + 4 * q);
        d[0] = make_uint2(pf[u].x ^ 0x80008000u, pf[u].y ^ 0x80008000u);
        d[1] = make_uint2(pf[u].z ^ 0x80008000u, pf[u].w ^ 0x80008000u);
      }
    }
  }
  else
  {
    slide = 0;
    fill_window_cols<9>(refL, reinterpret_cast<const uint4*>(ref + (winOff - off)), rs >> 3, winRows, MH_PITCH, nQuads, tid, (int)blockDim.x);
  }
  // the right-hand neighbour's new columns, if it is the next super-block of this run: two 16-byte loads per thread, in flight during the search
  pfHave = false;
  if (kk + 1 < kk0 + runLen && kk + 1 < chunk && item + 1 < g.total && sbx + 1 < g.nsbx && slide < MH_MAXSLIDE)      // wave-uniform
  {
    const int nsubxN = min(4, g.n16x - 4 * (sbx + 1)), nQuadsN = (((g.nR - 1) * 5 + 16 * nsubxN - 1 + off) >> 3) + 1;
    pfQ0 = nQuads - 8;
    if (nQuadsN - pfQ0 == 8 && 8 * winRows <= 2048)
    {
      pfHave = true;
      const uint4* gsrc = reinterpret_cast<const uint4*>(ref + (winOff + 64 - off));
#pragma unroll
      for (int u = 0; u < 2; u++)
      {
        const int it = tid + u * 1024;
        pf[u] = make_uint4(0u, 0u, 0u, 0u);
        if (it < 8 * winRows)
        {
          int row, q;
          slideItem(it, pfQ0, row, q);
          pf[u] = gsrc[(size_t)row * rsQ + q];
        }
      }
    }
  }
  if (tid < 42) keys[tid] = ~0ull;
  if (tid >= 64 && tid < 72) arrive[tid - 64] = 0;
  if (tid < MH_MAXDL * 4) surfD[tid] = 0u;
  for (int n = tid; n < MH_MAXSLOTS * 4; n += (int)blockDim.x) surf[n] = 0u;
  // LDS-only barrier: __syncthreads() also waits for vmcnt(0), i.e. for the neighbour's columns that were requested just above.  The first super-block
  // of the run: its original rows are packed here by every thread (behind the window's requests), and the barrier is the full one (stores complete)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (stamp && tid == 0) diag[1] = __builtin_amdgcn_s_memtime();
  if (diag && stampK >= 0 && tid == 0) diag[40 + 1 * 4 + stampK] = __builtin_amdgcn_s_memtime();

  // one copy of the unit's code (the unrolled stage bodies are ~3 KB each): the lane descriptor of the round is picked field by field, so that
  // LU[] stays in registers (an array indexed by the loop counter would live in scratch memory)
#pragma unroll 1
  for (int r = 0; r < 2; r++)
  {
    const int u = wave + r * nwaves;
    if (u < 4 * nsw)
    {
      const int sw = nsw - 1 - (u >> 2), q = u & 3;
      const int s = sw * 64 + lane;
      const bool waveHasDense = ndl > 0 && sw * 64 + 63 >= nslots;                // wave-uniform
      MhLane L;
      L.base = r ? LU[1].base : LU[0].base; L.idx = r ? LU[1].idx : LU[0].idx; L.kind = r ? LU[1].kind : LU[0].kind;
#pragma unroll
      for (int m = 0; m < 4; m++) L.cst[m] = r ? LU[1].cst[m] : LU[0].cst[m];
      mh_unit(orgPacked, g, off & 3, ldsBase + (unsigned)(slide * 128 + (q & 1) * 64 + (q >> 1) * 32 * MH_PITCH * 4), L, waveHasDense, q, sbx, sby, nsubx, nsuby, keys, surf, surfD, &arrive[sw], s, nslots, lane);
      if (stamp && lane == 0) diag[8 + u] = __builtin_amdgcn_s_memtime();
    }
  }
  __syncthreads();
  if (stamp && tid == 0) diag[2] = diag[3] = __builtin_amdgcn_s_memtime();
  if (diag && stampK >= 0 && tid == 0) diag[40 + 2 * 4 + stampK] = __builtin_amdgcn_s_memtime();

  // final records: thread t < 21: raster result of block t of the super-block, 21 <= t < 42: the +-D grid
  if (tid < 42)
  {
    const int grid = tid >= 21, t = tid - 21 * grid;
    vvcgpu_search_best* out = nullptr; int bidx = 0; bool exists = false;
    if (t < 16)      { const int tx = t & 3, ty = t >> 2; exists = tx < nsubx && ty < nsuby; bidx = (4 * sby + ty) * g.n16x + 4 * sbx + tx; out = outPtr[grid * 3]; }
    else if (t < 20) { const int qx = (t - 16) & 1, qy = (t - 16) >> 1; exists = 2 * qx + 2 <= nsubx && 2 * qy + 2 <= nsuby; bidx = (2 * sby + qy) * (g.n16x >> 1) + 2 * sbx + qx; out = outPtr[grid * 3 + 1]; }
    else             { exists = nsubx == 4 && nsuby == 4; bidx = sby * (g.n16x >> 2) + sbx; out = outPtr[grid * 3 + 2]; }
    if (exists && out && !(grid && !g.nD))
    {
      const unsigned long long key = keys[tid];
      const int idx = (int)(key & 0xFFFFFFu);
      const unsigned long long cost = key >> 24;
      const int n = grid ? g.nD : g.nR, step = grid ? 1 : 5, p0 = grid ? -g.D : -g.R;
      const int j = (int)(((float)idx + 0.5f) * __frcp_rn((float)n)), i = idx - j * n;      // idx / n (idx < 1521, n <= 39: exact)
      const int x = p0 + i * step, y = p0 + j * step;
      // the bits of the position and lambda x bits: the tables the units use (bitsRX .. / costTab hold exactly these values) -- the record writers are one wave
      // that every other wave waits for; two exp-Golomb loops, a double multiply and an integer division per record were ~1.5 k of a super-block's 48 k cycles
      const unsigned bits = grid ? (unsigned)bitsDX[i] + bitsDY[j] : (unsigned)bitsRX[i] + bitsRY[j];
      vvcgpu_search_best r;
      r.x = x; r.y = y; r.cost = cost; r.sad = cost - (unsigned long long)costTab[bits];
      out[bidx] = r;
    }
  }
  // (no barrier here: the 42 record writers read keys[] above and are the threads that reset keys[] for the next super-block below, in program order; the window,
  // the surfaces and the arrival counters that the other waves rewrite meanwhile are not read by the writers; every wave meets again at the barrier behind the window step)
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
  if (nR > MH_MAXN || nD > 9 || (c.dense_range && R < c.dense_range + 15) || c.sub_shift < 0 || c.sub_shift > 1 || (org_stride & 1) || (ref_stride & 7) || ((uintptr_t)org & 3) || ((uintptr_t)ref & 15) ||
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
  g.dlCount = 0;
  g.magicNq = (65536u + (unsigned)((nR + 3) >> 2) - 1u) / (unsigned)((nR + 3) >> 2);
  memset(g.dlKey, 0, sizeof g.dlKey); memset(g.dlX, -128, sizeof g.dlX);
  for (int x = -c.dense_range; x <= c.dense_range && nD; x++)
  {
    const int m = (R + x) & 3, key = (R + x - 5 * m) >> 2;
    int l = 0;
    while (l < g.dlCount && g.dlKey[l] != key) l++;
    if (l == g.dlCount) g.dlKey[g.dlCount++] = (short)key;
    g.dlX[4 * l + m] = (signed char)x;
  }
  g.magicDl = g.dlCount ? (65536u + (unsigned)g.dlCount - 1u) / (unsigned)g.dlCount : 0u;
  const int winRowsMax = (nR - 1) * 5 + 64;
  g.winBytes = winRowsMax * MH_PITCH * 4 + MH_MAXSLIDE * 128 + 128;            // + the slack the sliding window runs into
  const size_t smem = (size_t)g.winBytes + MH_MAXSLOTS * 4 * sizeof(unsigned);
  const int nblocks = c.n16x * c.n16y;
  unsigned* packed = static_cast<unsigned*>(vvcgpu_scratch(st, (size_t)nblocks * g.hs * 16 * sizeof(unsigned)));
  if (!packed) return VVCGPU_E_DEVICE;
  hipLaunchKernelGGL(mh_pack_org_kernel, dim3((unsigned)(((size_t)nblocks * g.hs * 4 + 255) / 256)), dim3(256), 0, st, org, org_stride, c.org_x, c.org_y, c.n16x, nblocks, g.hs, c.sub_shift, packed);
  VVC_LAUNCH_CHECK();
  VVC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(me_hier_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  unsigned long long* diag = nullptr;
#ifdef MH_DIAG
  const bool wantDiag = getenv("VVCGPU_MH_DIAG") != nullptr;                 // measurement aid (tools/mehier_time.py, a -DMH_DIAG build): phase stamps of one workgroup
#else
  const bool wantDiag = false;
#endif
  if (wantDiag) { VVC_HIP(hipMalloc(&diag, 64 * sizeof(unsigned long long))); VVC_HIP(hipMemsetAsync(diag, 0, 64 * sizeof(unsigned long long), st)); }
  // VVCGPU_MH_WGS (read per call; tests / tuning): the number of persistent workgroups, so that small grids walk runs -- and slide their window -- too
  const char* wgsEnv = getenv("VVCGPU_MH_WGS");
  const int wgsMax = wgsEnv && atoi(wgsEnv) >= 8 ? (atoi(wgsEnv) / 8) * 8 : (vvcgpu_cu_count() / 8) * 8;
  const int gridWgs = min(cdiv(g.total, 8) * 8, wgsMax);
  hipLaunchKernelGGL(me_hier_kernel, dim3(gridWgs), dim3(1024), smem, st, packed, ref, ref_stride, g, *mvcost_host,
                     raster_best[0], raster_best[1], raster_best[2], dense_best ? dense_best[0] : nullptr, dense_best ? dense_best[1] : nullptr, dense_best ? dense_best[2] : nullptr, diag);
  VVC_LAUNCH_CHECK();
  if (wantDiag)
  {
    unsigned long long h[64];
    VVC_HIP(hipStreamSynchronize(st));
    VVC_HIP(hipMemcpy(h, diag, sizeof h, hipMemcpyDeviceToHost));
    (void)hipFree(diag);
    fprintf(stderr, "[vvcgpu me_hier diag] cycles: window staged %llu, units done %llu, 64x64 pass %llu; per unit end (since start):", h[1] - h[0], h[2] - h[0], h[3] - h[0]);
    for (int u = 0; u < 32; u++) if (h[8 + u]) fprintf(stderr, " %llu", h[8 + u] - h[0]);
    fprintf(stderr, "\n[vvcgpu me_hier diag] four more workgroups (window staged / all done):");
    for (int k = 0; k < 4; k++) fprintf(stderr, " %llu / %llu", h[44 + k] - h[40 + k], h[48 + k] - h[40 + k]);
    fprintf(stderr, "\n");
  }
  return VVCGPU_OK;
}

}  // extern "C"
