// interp.hip -- DCTIF interpolation (I1), motion compensation of prediction blocks (I3, incl. bi-pred average B1)
//               and the PelBuffer element-wise operations (B1-B4) for gfx950.
//
// Reference behaviour reproduced (bit-exact, bit depth <= 10):
//   InterpolationFilter::filter<N,isVertical,isFirst,isLast>   CommonLib/InterpolationFilter.cpp:290-379
//   InterpolationFilter::filterCopy<isFirst,isLast>            :205-264
//   InterPrediction::xPredInterBlk                             CommonLib/InterPrediction.cpp:480-547
//   AreaBuf<Pel>::addAvg / reconstruct / linearTransform       CommonLib/Buffer.cpp:114-151, 225-293 (+ cores :50-94)
//   AreaBuf::subtract / removeHighFreq / copyClip              CommonLib/Buffer.h:321-339, 389-416; Buffer.cpp:197-222
//   `Pel val = (sum + offset) >> shift` narrows to int16 BEFORE the clip (InterpolationFilter.cpp:368).
//
// Design: one 64-lane wave per descriptor.  MC works on 16x16 sub-tiles: the (16+7)^2 reference window is staged once
// in LDS, the horizontal pass writes the 14-bit intermediate to LDS (never to HBM -- the reference's m_filteredBlockTmp),
// the vertical pass and, for bi-prediction, the second list and the average stay in registers.
#include "common.h"
#include "dist_dev.h"
#include "mfma_tr.h"
#include <mutex>

// the generic MC scan loads bytes 32..47 of a descriptor as ONE uint4 (dst_stride | w, h | the four phases | is_luma, bi, reserved): the layout is part of
// the ABI, and the descriptor array must be 16-byte aligned (include/vvcgpu.h)
static_assert(offsetof(vvcgpu_mc_desc, dst_stride) == 32 && offsetof(vvcgpu_mc_desc, w) == 36 && offsetof(vvcgpu_mc_desc, frac_x0) == 40 &&
              offsetof(vvcgpu_mc_desc, is_luma) == 44 && sizeof(vvcgpu_mc_desc) == 48, "vvcgpu_mc_desc layout");

namespace {

__constant__ short c_lumaFilter[16][8] = {
  {  0, 0,   0, 64,  0,   0,  0,  0 }, {  0, 1,  -3, 63,  4,  -2,  1,  0 }, { -1, 2,  -5, 62,  8,  -3,  1,  0 },
  { -1, 3,  -8, 60, 13,  -4,  1,  0 }, { -1, 4, -10, 58, 17,  -5,  1,  0 }, { -1, 4, -11, 52, 26,  -8,  3, -1 },
  { -1, 3,  -9, 47, 31, -10,  4, -1 }, { -1, 4, -11, 45, 34, -10,  4, -1 }, { -1, 4, -11, 40, 40, -11,  4, -1 },
  { -1, 4, -10, 34, 45, -11,  4, -1 }, { -1, 4, -10, 31, 47,  -9,  3, -1 }, { -1, 3,  -8, 26, 52, -11,  4, -1 },
  {  0, 1,  -5, 17, 58, -10,  4, -1 }, {  0, 1,  -4, 13, 60,  -8,  3, -1 }, {  0, 1,  -3,  8, 62,  -5,  2, -1 },
  {  0, 1,  -2,  4, 63,  -3,  1,  0 } };
__constant__ short c_chromaFilter[32][4] = {
  {  0, 64,  0,  0 }, { -1, 63,  2,  0 }, { -2, 62,  4,  0 }, { -2, 60,  7, -1 }, { -2, 58, 10, -2 }, { -3, 57, 12, -2 },
  { -4, 56, 14, -2 }, { -4, 55, 15, -2 }, { -4, 54, 16, -2 }, { -5, 53, 18, -2 }, { -6, 52, 20, -2 }, { -6, 49, 24, -3 },
  { -6, 46, 28, -4 }, { -5, 44, 29, -4 }, { -4, 42, 30, -4 }, { -4, 39, 33, -4 }, { -4, 36, 36, -4 }, { -4, 33, 39, -4 },
  { -4, 30, 42, -4 }, { -4, 29, 44, -5 }, { -4, 28, 46, -6 }, { -3, 24, 49, -6 }, { -2, 20, 52, -6 }, { -2, 18, 53, -5 },
  { -2, 16, 54, -4 }, { -2, 15, 55, -4 }, { -2, 14, 56, -4 }, { -2, 12, 57, -3 }, { -2, 10, 58, -2 }, { -1,  7, 60, -2 },
  {  0,  4, 62, -2 }, {  0,  2, 63, -1 } };

constexpr int IF_INTERNAL_PREC = 14, IF_FILTER_PREC = 6, IF_INTERNAL_OFFS = 1 << 13;

struct IfMode { int shift, offset; };
__device__ __forceinline__ IfMode if_mode(bool isFirst, bool isLast, int bd)
{
  const int headRoom = max(2, IF_INTERNAL_PREC - bd);
  IfMode m;
  m.shift = IF_FILTER_PREC;
  if (isLast) { m.shift += isFirst ? 0 : headRoom; m.offset = (1 << (m.shift - 1)) + (isFirst ? 0 : IF_INTERNAL_OFFS << IF_FILTER_PREC); }
  else        { m.shift -= isFirst ? headRoom : 0; m.offset = isFirst ? -(IF_INTERNAL_OFFS << m.shift) : 0; }
  return m;
}
__device__ __forceinline__ int if_copy(int s, bool isFirst, bool isLast, int bd, int cmin, int cmax)
{
  const int shift = max(2, IF_INTERNAL_PREC - bd);
  if (isFirst == isLast) return s;
  if (isFirst) return (short)((short)(s << shift) - (short)IF_INTERNAL_OFFS);
  return clip3(cmin, cmax, (short)((s + IF_INTERNAL_OFFS + (1 << (shift - 1))) >> shift));
}

// ------------------------------------------------------------------------------------------------ I1
// four samples of a row in ONE 8-byte access whatever the address (blocks start at any sample: 2-byte alignment).  gfx950 under HSA runs with
// unaligned global access enabled and the compiler knows it: a load / store through a 2-byte-aligned type is one global_load / store_dwordx2.  (Until
// round 4 the access branched three ways on the address -- 8-, 4-, 2-byte aligned -- and a wave whose lanes disagreed walked all three.)
struct __attribute__((packed, aligned(2))) PelQuad { short v[4]; };
__device__ __forceinline__ void if_load4(const Pel* p, int (&v)[4])
{
  const PelQuad q = *reinterpret_cast<const PelQuad*>(p);
  v[0] = q.v[0]; v[1] = q.v[1]; v[2] = q.v[2]; v[3] = q.v[3];
}

// one descriptor by a group of G lanes (lane = index inside the group)
// rows [y0, y1) only (default: all of them): dist-like band splitting of heavy calls
template <int G>
__device__ __forceinline__ void if_one(const vvcgpu_if_desc& d, const Pel* __restrict__ srcBase, Pel* __restrict__ dstBase, int lane, bool act, int bd,
                                       int cmin, int cmax, int y0 = 0, int y1 = 1 << 20)
{
  const int hh = min((int)d.h, y1) - y0;
  const Pel* src = srcBase + d.src_off + (ptrdiff_t)y0 * d.src_stride;
  Pel* dst = dstBase + d.dst_off + (ptrdiff_t)y0 * d.dst_stride;
  const int N = d.taps, w = d.w, count = act ? w * hh : 0;
  if (N == 0)
  {
    for (int i = lane; i < count; i += G)
    {
      const int y = i / w, x = i - y * w;
      dst[(size_t)y * d.dst_stride + x] = (short)if_copy(src[(size_t)y * d.src_stride + x], d.is_first, d.is_last, bd, cmin, cmax);
    }
    return;
  }
  const int cStride = d.is_vertical ? d.src_stride : 1;
  src -= (N / 2 - 1) * cStride;
  const IfMode m = if_mode(d.is_first, d.is_last, bd);
  int c[8];
#pragma unroll
  for (int k = 0; k < 8; k++) c[k] = k < N ? d.coeff[k] : 0;
  {
    // four consecutive outputs of a row per lane: N + 3 samples (horizontal) or N loads of four samples (vertical) instead of 4 N two-byte loads.
    // Widths that are not a multiple of four (the W + 1 wide planes of xExtDIFUpSamplingH / Q: 5, 9, 17, 33, 65, 129 -- 13 % of the samples of a real
    // call mix and, with a sample-by-sample partial unit per row that every wave walked beside its full units, half of if_batch's time on it): the
    // LAST unit of a row starts at w - 4 and overlaps its neighbour, so it is a full unit too -- the overlapped outputs are written twice with the
    // same values, and it reads exactly up to the last sample the reference reads (x + 3 + N - 1 = w + N - 2), nothing beyond the reference's window.
    // Only rows narrower than four samples keep a partial unit (nv < 4, sample-by-sample loads).
    const int upr = (w + 3) >> 2, units = act ? upr * hh : 0;
    for (int u = lane; u < units; u += G)
    {
      const int y = u / upr, xu = (u - y * upr) << 2, x = w >= 4 ? min(xu, w - 4) : xu, nv = min(4, w - x);
      const Pel* s = src + (size_t)y * d.src_stride + x;
      int sum[4] = { 0, 0, 0, 0 };
      if (d.is_vertical)
      {
#pragma unroll
        for (int k = 0; k < 8; k++)
          if (k < N)
          {
            int v[4];
            if (nv == 4) if_load4(s + (size_t)k * d.src_stride, v);
            else
            {
#pragma unroll
              for (int j = 0; j < 4; j++) v[j] = j < nv ? (int)s[(size_t)k * d.src_stride + j] : 0;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) sum[j] += v[j] * c[k];
          }
      }
      else
      {
        int v[12];                                        // exactly the N + 3 samples the four outputs read: nothing beyond the reference's window
        if (nv == 4)
        {
          int q[4];
          if_load4(s, q); v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
          if (N == 8) { if_load4(s + 4, q); v[4] = q[0]; v[5] = q[1]; v[6] = q[2]; v[7] = q[3]; v[8] = s[8]; v[9] = s[9]; v[10] = s[10]; }
          else if (N == 4) { v[4] = s[4]; v[5] = s[5]; v[6] = s[6]; }
          else v[4] = s[4];
        }
        else
        {
#pragma unroll
          for (int j = 0; j < 11; j++) v[j] = j < nv + N - 1 ? (int)s[j] : 0;
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int k = 0; k < 8; k++) if (k < N) sum[j] += v[j + k] * c[k];
      }
      pel4 o;
#pragma unroll
      for (int j = 0; j < 4; j++)
      {
        int val = (short)((sum[j] + m.offset) >> m.shift);
        if (d.is_last) val = clip3(cmin, cmax, val);
        o[j] = (short)val;
      }
      Pel* dp = dst + (size_t)y * d.dst_stride + x;
      if (nv == 4) { PelQuad q; q.v[0] = o[0]; q.v[1] = o[1]; q.v[2] = o[2]; q.v[3] = o[3]; *reinterpret_cast<PelQuad*>(dp) = q; }
      else
      {
#pragma unroll
        for (int j = 0; j < 4; j++) if (j < nv) dp[j] = o[j];
      }
    }
  }
}

// A workgroup takes 64 consecutive descriptors and BINS them first (one ballot of its first wave): calls of at most 256 samples -- the reference
// encoder's table-slot calls are mostly 4 wide: tests/golden/trace_*.npz -- run four side by side in a wave, 16 lanes each; larger ones take a whole
// wave each; HEAVY calls (more than 512 samples) are listed for if_heavy_kernel, which splits them into bands of rows over many waves: one wave needs
// ~70 us for 128 x 135 samples, and that was the run time of the whole launch on a real call mix.  (Until round 4 a wave took four CONSECUTIVE
// descriptors and ran them side by side only when all four were small: on the real call mix three of four waves lost that form to one larger
// neighbour -- the mixed batch took 2.5 x the time of its parts, profiles/r04_entry_shapes.txt.)
constexpr int IF_HEAVY = 2048, IF_HEAVY_FILTER = 512;   // the filter does 8 multiply-adds per sample: its heavy threshold is lower than the element-wise ops'
constexpr int IF_WG_DESCS = 64;
// a heavy call is cut into at most 16 bands of rows: band height = 512 samples' worth, more when that would give more than 16
__device__ __forceinline__ int if_filter_band_rows(int w, int h) { return max(max(2, IF_HEAVY_FILTER / w), (h + 15) >> 4); }
__device__ __forceinline__ int if_band_rows(int w, int h) { return max(max(4, IF_HEAVY / w), (h + 15) >> 4); }
__global__ __launch_bounds__(256) void if_batch_kernel(const Pel* __restrict__ srcBase, Pel* __restrict__ dstBase,
                                                       const vvcgpu_if_desc* __restrict__ descs, int n, int perWg, int localHeavy, int bd,
                                                       int cmin, int cmax, int* __restrict__ heavyCount, int* __restrict__ heavyList, int* __restrict__ nextCounters)
{
  if (blockIdx.x == 0 && threadIdx.x < VVC_CTR_INTS) nextCounters[threadIdx.x] = 0;       // the counter set of the next call on this stream (vvcgpu_counters)
  __shared__ unsigned char tList[IF_WG_DESCS], sList[IF_WG_DESCS], mList[IF_WG_DESCS], hList[IF_WG_DESCS];
  __shared__ int cntT, cntS, cntM, cntH;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int base = blockIdx.x * perWg;
  if (wave == 0)
  {
    const int di = base + lane;
    int w = 0, h = 0;
    if (lane < perWg && di < n) { w = descs[di].w; h = descs[di].h; }
    const int sz = w * h;
    // tiny: at most four 4-output units (4 x 4 and smaller: half of the trace's calls) -- four lanes each, sixteen calls side by side
    const bool tiny = sz > 0 && ((w + 3) >> 2) * h <= 4, small = sz > 0 && !tiny && sz <= 256, heavy = sz > IF_HEAVY_FILTER, med = sz > 0 && !tiny && !small && !heavy;
    const unsigned long long below = (1ull << lane) - 1ull;
    const unsigned long long mt = __builtin_amdgcn_ballot_w64(tiny), ms = __builtin_amdgcn_ballot_w64(small), mm = __builtin_amdgcn_ballot_w64(med),
                             mh = __builtin_amdgcn_ballot_w64(heavy);
    if (tiny) tList[__popcll(mt & below)] = (unsigned char)lane;
    if (small) sList[__popcll(ms & below)] = (unsigned char)lane;
    if (med) mList[__popcll(mm & below)] = (unsigned char)lane;
    if (lane == 0) { cntT = (int)__popcll(mt); cntS = (int)__popcll(ms); cntM = (int)__popcll(mm); cntH = localHeavy ? (int)__popcll(mh) : 0; }
    // localHeavy (long lists: thousands of workgroups): the workgroup serves its own heavy calls, bands of rows dealt to its four waves -- a second
    // launch for them costs its own ~12 us of latency behind this one, a third of the call's time on a real encoder's call mix at the bench's batch size
    if (heavy && localHeavy) hList[__popcll(mh & below)] = (unsigned char)lane;
    if (mh != 0ull && !localHeavy)                      // ONE atomic per workgroup (same-address atomics retire at ~12 ns each)
    {
      int b = 0;
      if (lane == 0) b = atomicAdd(heavyCount, (int)__popcll(mh));
      b = __builtin_amdgcn_readfirstlane(b);
      // the entry carries the number of bands the call really has, so that the band walk of if_heavy_kernel skips the others without touching the descriptor
      if (heavy) heavyList[b + (int)__popcll(mh & below)] = di | (((h + if_filter_band_rows(w, h) - 1) / if_filter_band_rows(w, h) - 1) << 27);
    }
  }
  __syncthreads();
  const int nT = cntT, nS = cntS, nM = cntM;
  for (int g0 = wave * 16; g0 < nT; g0 += 64)
  {
    const int k = g0 + (lane >> 2);
    const bool act = k < nT;
    const vvcgpu_if_desc mine = descs[base + tList[act ? k : g0]];
    if_one<4>(mine, srcBase, dstBase, lane & 3, act, bd, cmin, cmax);
  }
  for (int g0 = wave * 4; g0 < nS; g0 += 16)
  {
    const int k = g0 + (lane >> 4);
    const bool act = k < nS;
    const vvcgpu_if_desc mine = descs[base + sList[act ? k : g0]];
    if_one<16>(mine, srcBase, dstBase, lane & 15, act, bd, cmin, cmax);
  }
  for (int k = wave; k < nM; k += 4)
  {
    const vvcgpu_if_desc d = descs[base + __builtin_amdgcn_readfirstlane((int)mList[k])];
    if_one<64>(d, srcBase, dstBase, lane, true, bd, cmin, cmax);
  }
  const int nH = cntH;
  for (int k = 0; k < nH; k++)
  {
    const vvcgpu_if_desc d = descs[base + __builtin_amdgcn_readfirstlane((int)hList[k])];
    const int br = if_filter_band_rows(d.w, d.h);
    for (int r0 = ((wave + k) & 3) * br; r0 < d.h; r0 += 4 * br) if_one<64>(d, srcBase, dstBase, lane, true, bd, cmin, cmax, r0, r0 + br);
  }
}
// one wave per (heavy call, band of rows)
__global__ __launch_bounds__(256) void if_heavy_kernel(const Pel* __restrict__ srcBase, Pel* __restrict__ dstBase, const vvcgpu_if_desc* __restrict__ descs,
                                                       int bd, int cmin, int cmax, const int* __restrict__ heavyCount, const int* __restrict__ heavyList)
{
  const int lane = threadIdx.x & 63;
  const int cnt = heavyCount[0], waves = gridDim.x * 4;
  // pair p = (band p / cnt, item p % cnt): BAND-major, so that the waves of one step hold neighbouring calls of the same band -- item-major
  // (p >> 4, p & 15) with a wave stride that is a multiple of 16 gave every wave the same band for the whole launch, and a call of two bands
  // (32 x 32: 3 % of the trace's samples) kept one wave in eight busy (0.131 ms per 4 M samples against 0.024 for 32 x 16)
  for (int p = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); p < cnt * 16; p += waves)
  {
    const int band = p / cnt, e = __builtin_amdgcn_readfirstlane(heavyList[p - band * cnt]);
    if (band > (e >> 27)) continue;
    const vvcgpu_if_desc d = descs[e & ((1 << 27) - 1)];
    const int br = if_filter_band_rows(d.w, d.h), r0 = band * br;
    if (r0 < d.h) if_one<64>(d, srcBase, dstBase, lane, true, bd, cmin, cmax, r0, r0 + br);
  }
}

// ------------------------------------------------------------------------------------------------ I3
constexpr int ST = 16;                 // sub-tile
constexpr int WP = ST + 8;             // window pitch (samples)
constexpr int WR = ST + 7;             // window rows

__device__ __forceinline__ bool mc_is_fast(int is_luma, int w, int h) { return is_luma ? (w == 16 && h == 16) : (w == 8 && h == 8); }

typedef short mc_s2 __attribute__((ext_vector_type(2)));
// ---------------------------------------------------------------------------------------------------
// Fast path, packed form.  G lanes serve one PU (64: a 16x16 luma PU per wave; 32: two 8x8 chroma PUs per wave, one per half).
//   * window staging: a lane loads EIGHT bytes at the 4-byte-aligned address of its dword and shifts by the window's sub-dword phase
//     (`v_alignbit`), so LDS holds the window sample-aligned, 2 samples per dword, row pitch WD dwords -- 5 load instructions per reference
//     for the 23 x 23 luma window instead of 9 two-byte ones, both references requested before either is used.  Rows / columns that the
//     reference's branch does not read (fy == 0: rows outside the block, fx == 0: columns outside) are not loaded: their taps are 0.
//     A staged row may start and end up to two samples beyond the columns the reference touches (in the same row).
//   * both passes are the same code: a lane takes FOUR consecutive outputs along the filter direction from 11 (7) consecutive samples =
//     three (two) aligned ds_read_b64, pairs D_m = (s[2m], s[2m+1]) are the dwords themselves, the odd pairs E_m one `v_alignbit` each, and
//     every output is N/2 `v_dot2_i32_i16` -- the first pass writes the 14-bit intermediate TRANSPOSED (tmpT[x][row]) so that the second
//     pass reads its column as a row.  The unit filter (frac 0) through the same code equals the reference's copy / single-pass branches
//     bit for bit ((64 t) >> 6 == t, and (2^h S - 2^19) >> 6 == (S - 2^(19-h)) >> (6-h)) except ONE: a rounded (bi == 0) horizontal-only
//     filter, whose first pass therefore takes the last-stage rounding and whose second pass copies.
//   * the block leaves through LDS as rows: one 8-byte store per lane.
template <int N, int S, int G> struct McStaged
{
  static constexpr int NR = S + N - 1, WD = (NR + 2) / 2 + ((((NR + 2) / 2) & 1) ? 1 : 0), NL = (NR * WD + G - 1) / G;
  uint2 ld[2][NL];
  unsigned phase[2];                                                      // bit u: load u starts on an odd sample (odd strides: per row)
  unsigned bad;                                                           // non-zero: a loaded sample lies outside the bit depth (this lane's loads)
};

// requests both windows of a PU (see mc_tile_dot2)
template <int N, int S, int G>
__device__ __forceinline__ void mc_stage(const vvcgpu_mc_desc& d, bool active, const Pel* __restrict__ ref0Base, const Pel* __restrict__ ref1Base, int gl,
                                         McStaged<N, S, G>& st, int bd)
{
  constexpr int half = N / 2 - 1, NR = S + N - 1, WD = McStaged<N, S, G>::WD, LOADS = NR * WD;
  const int nRef = d.bi == 1 ? 2 : 1;
  auto& ld = st.ld;
  auto& phase = st.phase;
  phase[0] = phase[1] = 0u;
  // The two-pass form narrows its first pass to 16 bits sample by sample ((s << headroom) - 8192); the reference's one-dimensional branches
  // narrow only the filtered value.  The two agree for samples inside the bit depth; a window that holds anything else is reported here and the
  // PU takes the sample-wise body of the generic kernel (which follows the reference branch by branch).  Whole dwords are tested, so a sample
  // beside the window may report a PU that did not need it: speed only.
  const unsigned outside = ~(((1u << bd) - 1u) * 0x10001u);
  unsigned bad = 0u;
#pragma unroll
  for (int r = 0; r < 2; r++)
  {
    const int rs = r ? d.ref1_stride : d.ref0_stride;
    const Pel* ref = (r ? ref1Base + d.ref1_off : ref0Base + d.ref0_off) - (ptrdiff_t)half * rs - half;       // window origin
    const int fx = r ? d.frac_x1 : d.frac_x0, fy = r ? d.frac_y1 : d.frac_y0;
#pragma unroll
    for (int u = 0; u < (LOADS + G - 1) / G; u++)
    {
      const int i = gl + G * u, rr = i / WD, dw = i - rr * WD;
      ld[r][u] = make_uint2(0u, 0u);
      // samples 2 dw, 2 dw + 1 of the window row; needed when the row and one of the two columns are
      const bool rowOk = fy ? rr < NR : (rr >= half && rr < half + S);
      const bool colOk = fx ? 2 * dw < NR : (2 * dw + 1 >= half && 2 * dw < half + S);
      if (active && r < nRef && i < LOADS && rowOk && colOk)
      {
        const unsigned char* a = reinterpret_cast<const unsigned char*>(ref + (ptrdiff_t)rr * rs) + 4 * dw;
        const unsigned* a4 = reinterpret_cast<const unsigned*>(reinterpret_cast<uintptr_t>(a) & ~(uintptr_t)3);
        // On an odd phase sample 2 dw is the high half of a4[0] and sample 2 dw + 1 the low half of a4[1]: both dwords are read whole, so at
        // the two ends of a row up to TWO samples beyond the needed columns are touched (include/vvcgpu.h states this).  Reading only the needed
        // halves (predicated loads, or an address select that re-reads the other dword) was measured: 12 more spilled VGPRs at the kernel's
        // 80-register budget and 0.090 instead of 0.076 ms for the MC launches of a 4K picture.
        if (reinterpret_cast<uintptr_t>(a) & 2)
        {
          ld[r][u].x = a4[0];
          ld[r][u].y = a4[1];
          phase[r] |= 1u << u;
        }
        else ld[r][u].x = a4[0];
        bad |= (ld[r][u].x | ld[r][u].y) & outside;
      }
    }
  }
  st.bad = bad;
}

template <int N, int S, int G>
__device__ __forceinline__ void mc_tile_dot2(const vvcgpu_mc_desc& d, bool active, const McStaged<N, S, G>& st, Pel* __restrict__ dstBase, int bd, int cmin, int cmax,
                                             int gl, unsigned* win, short* tmpT, short* outL)
{
  constexpr int half = N / 2 - 1, NR = S + N - 1, WD = McStaged<N, S, G>::WD;                              // dwords per window row (even: 8-byte reads)
  constexpr int TP = 2 * WD;                                             // tmpT pitch in samples
  constexpr int NG = S / 4, NP = N / 2, ND = NP + 2;                     // output groups per line, coefficient pairs, dwords read per lane
  constexpr int HITEMS = NR * NG, VITEMS = S * NG, LOADS = NR * WD;
  if (!active) return;                                                    // (the wave barriers below only order this wave's own LDS accesses)
  const int hr = max(2, IF_INTERNAL_PREC - bd);
  const bool rndRes = d.bi == 0;
  const int nRef = d.bi == 1 ? 2 : 1;
  const auto& ld = st.ld;
  const auto& phase = st.phase;
  int pred[2][4] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 } };
#pragma unroll
  for (int r = 0; r < 2; r++)
  {
    if (r >= nRef) break;                                                 // uniform per group; the other half of a chroma wave follows its own d
    const int fx = r ? d.frac_x1 : d.frac_x0, fy = r ? d.frac_y1 : d.frac_y0;
    const unsigned* cxp = reinterpret_cast<const unsigned*>(N == 8 ? c_lumaFilter[fx] : c_chromaFilter[fx]);
    const unsigned* cyp = reinterpret_cast<const unsigned*>(N == 8 ? c_lumaFilter[fy] : c_chromaFilter[fy]);
    unsigned cx[NP], cy[NP];
#pragma unroll
    for (int m = 0; m < NP; m++) { cx[m] = cxp[m]; cy[m] = cyp[m]; }
    const bool hOnly = rndRes && fy == 0 && fx != 0;                      // the one branch the two-pass form does not reproduce: see above
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < (LOADS + G - 1) / G; u++)
    {
      const int i = gl + G * u;
      if (i < LOADS) win[i] = (phase[r] >> u) & 1u ? __builtin_amdgcn_alignbit(ld[r][u].y, ld[r][u].x, 16) : ld[r][u].x;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    // four outputs from ND consecutive dwords
    auto four = [&](const unsigned (&D)[ND], const unsigned (&c)[NP], int (&o)[4])
    {
      unsigned E[ND - 1];
#pragma unroll
      for (int m = 0; m < ND - 1; m++) E[m] = __builtin_amdgcn_alignbit(D[m + 1], D[m], 16);
      o[0] = o[1] = o[2] = o[3] = 0;
#pragma unroll
      for (int m = 0; m < NP; m++)
      {
        const mc_s2 cm = __builtin_bit_cast(mc_s2, c[m]);
        o[0] = __builtin_amdgcn_sdot2(__builtin_bit_cast(mc_s2, D[m]), cm, o[0], false);
        o[1] = __builtin_amdgcn_sdot2(__builtin_bit_cast(mc_s2, E[m]), cm, o[1], false);
        o[2] = __builtin_amdgcn_sdot2(__builtin_bit_cast(mc_s2, D[m + 1]), cm, o[2], false);
        o[3] = __builtin_amdgcn_sdot2(__builtin_bit_cast(mc_s2, E[m + 1]), cm, o[3], false);
      }
    };
    {
      const int shift1 = hOnly ? IF_FILTER_PREC : IF_FILTER_PREC - hr;
      const int off1 = hOnly ? (1 << (IF_FILTER_PREC - 1)) : -(IF_INTERNAL_OFFS << shift1);
#pragma unroll
      for (int u = 0; u < (HITEMS + G - 1) / G; u++)
      {
        const int it = gl + G * u, rr = it / NG, g = it - rr * NG;
        if (it < HITEMS)
        {
          unsigned D[ND];
          const uint2* wp = reinterpret_cast<const uint2*>(win + rr * WD + 2 * g);
#pragma unroll
          for (int m = 0; m < ND / 2; m++) { const uint2 q = wp[m]; D[2 * m] = q.x; D[2 * m + 1] = q.y; }
          int o[4];
          four(D, cx, o);
#pragma unroll
          for (int j = 0; j < 4; j++)
          {
            int t = (short)((o[j] + off1) >> shift1);
            if (hOnly) t = clip3(cmin, cmax, t);
            tmpT[(4 * g + j) * TP + rr] = (short)t;
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    if (gl < VITEMS)
    {
      const int x = gl % S, yg = gl / S;
      if (hOnly)
      {
#pragma unroll
        for (int j = 0; j < 4; j++) pred[r][j] = tmpT[x * TP + half + 4 * yg + j];
      }
      else
      {
        unsigned D[ND];
        const uint2* tp = reinterpret_cast<const uint2*>(tmpT + x * TP + 4 * yg);
#pragma unroll
        for (int m = 0; m < ND / 2; m++) { const uint2 q = tp[m]; D[2 * m] = q.x; D[2 * m + 1] = q.y; }
        int o[4];
        four(D, cy, o);
        const int shift2 = rndRes ? IF_FILTER_PREC + hr : IF_FILTER_PREC;
        const int off2 = rndRes ? (1 << (shift2 - 1)) + (IF_INTERNAL_OFFS << IF_FILTER_PREC) : 0;
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
          int v = (short)((o[j] + off2) >> shift2);
          if (rndRes && (fx | fy) != 0) v = clip3(cmin, cmax, v);        // (a uni-predictive full-sample copy is not clipped: filterCopy with isFirst == isLast)
          pred[r][j] = v;
        }
      }
    }
  }
  // average, rows through LDS, 8-byte stores
  const int shiftNum = max(2, IF_INTERNAL_PREC - bd) + 1, offset = (1 << (shiftNum - 1)) + 2 * IF_INTERNAL_OFFS;
  if (gl < VITEMS)
  {
    const int x = gl % S, yg = gl / S;
#pragma unroll
    for (int j = 0; j < 4; j++)
    {
      int v = pred[0][j];
      if (d.bi == 1) v = clip3(cmin, cmax, (pred[0][j] + pred[1][j] + offset) >> shiftNum);
      outL[(4 * yg + j) * S + x] = (short)v;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
  if (active && gl < VITEMS)
  {
    const int row = gl / NG, seg = gl - row * NG;
    Pel* o = dstBase + d.dst_off + (ptrdiff_t)row * d.dst_stride + 4 * seg;
    const uint2 v = *reinterpret_cast<const uint2*>(outL + row * S + 4 * seg);
    if ((reinterpret_cast<uintptr_t>(o) & 7) == 0) *reinterpret_cast<uint2*>(o) = v;
    else { o[0] = (short)(v.x & 0xFFFF); o[1] = (short)(v.x >> 16); o[2] = (short)(v.y & 0xFFFF); o[3] = (short)(v.y >> 16); }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
}

constexpr int MC_LDS_DW = 23 * 12 + 16 * 12 + 128;                       // per wave: window, transposed intermediate, output rows (luma sizes)

// ---------------------------------------------------------------------------------------------------
// The fast shapes ON THE MATRIX CORES (round 5): 16x16 luma and 8x8 chroma PUs, uni- and bi-predictive, quarter- (chroma: eighth-) sample phases.
// The packed vector-pipe form (mc_tile_dot2) spends ~570 vector instructions per PU wave around 78 v_dot2 with the matrix pipe idle; here both filter passes are exact
// f16 products (the machinery of frac16m_kernel, fracsearch.hip): ~120 vector instructions per luma PU, ~100 per PAIR of chroma PUs.
//   window   rows through lanes, eight columns per lane as one 16-byte load (any alignment), samples as f16 bit patterns 0x6400 | v (= 1024 + v)
//   pass 1   W (A operand) x Toeplitz matrix of the taps (table, B operand); the constant that the sum has to start from travels in the product's
//            spare columns (operand constants 1.0 and 1024.0); one v_add_f32 with 2^23 S leaves u = plane + 16384 in the low mantissa bits (floor by
//            round-to-nearest at a chosen exponent); limbs lo = u & 127, hi = u >> 7 as 0x6400 | limb
//   pass 2   plane^T (A operand: the pass-1 result registers as they are) x Toeplitz matrix per limb (c and 128 c): lane = output row, registers = four
//            neighbouring columns -> one 8-byte store per lane; rounding shifts as fma + floor on exact f32 integers, the last one and the clip in
//            packed f16 (v_cvt_pkrtz in [1024, 2048) IS the floor; 0x6400 | v back to v by a mask)
//   chroma   two 8x8 PUs per product: block-diagonal Toeplitz matrices (pass 1: columns of PU 0 | PU 1 along k; pass 2: the table row a lane reads
//            belongs to ITS PU's phase), so every product is shared and half of the result lanes are real
//   the one branch of the reference that is not two passes -- a rounded, horizontal-only filter (uni, frac_y == 0, frac_x != 0) -- gets the last-stage
//   rounding and the clip in pass 1 (own constants) and a copy in pass 2.
// PUs it cannot take (bi == 2, other phases, reference samples outside the bit depth) are flagged for the generic kernel behind it.
typedef _Float16 mm_h2 __attribute__((ext_vector_type(2)));
constexpr int MM_TAL = 0;                                    // [4 phases][23 columns x + shift][4 lane groups]  luma pass 1
constexpr int MM_TBL = MM_TAL + 4 * 23 * 4;                  // [4 phases][2 row chunks][64 lanes]          luma pass 2
constexpr int MM_TAC = MM_TBL + 4 * 2 * 64;                  // [2 variants][8 phases][8 x][2] + a zero row   chroma pass 1
constexpr int MM_TBC = MM_TAC + 2 * 8 * 16 + 1;              // [8 phases][8 y][4 lane groups]               chroma pass 2
constexpr int MM_ENTRIES = MM_TBC + 8 * 32;                  // 16-byte entries (eight f16 each)

__global__ void mm_build_tables_kernel(_Float16* __restrict__ tab, int bd)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= MM_ENTRIES * 8) return;
  const int ent = i >> 3, e = i & 7;
  const int hr = max(2, IF_INTERNAL_PREC - bd);
  const float S = (float)(1 << (IF_FILTER_PREC - hr));
  // start value of a pass-1 sum, as b0 * 1.0 + b1 * 1024.0.  variant 0: 8192 S - 65536 - (S - 1) / 2 (removes the sample bias 1024 x 64, adds 16384 S
  // and the first-stage offset -8192 S, centres the floor); variant 1 (rounded horizontal-only): 32 + 16384 x 64 - 65536 - 31.5
  const float b0[2] = { -0.5f * (S - 1.f), 0.5f }, b1[2] = { 8.f * S - 64.f, 960.f };
  float v = 0.f;
  if (ent < MM_TBL)
  {
    // luma pass 1 reads whole ALIGNED 16-byte words: k = 8 g + e is the column of the aligned row segment, the window starts `shift` (0..7) columns into
    // it, and output x takes its taps from columns x + shift .. + 7: the table row depends on x + shift only (the start value of the sum travels in the
    // accumulator)
    const int q = ent / 92, xs = (ent % 92) >> 2, g = ent & 3;
    const int t = 8 * g + e - xs;
    if (t >= 0 && t <= 7) v = (float)c_lumaFilter[4 * q][t];
  }
  else if (ent < MM_TAC)
  {
    const int r = ent - MM_TBL, q = r / 128, ch = (r / 64) & 1, lane = r & 63, y = lane & 15, g = lane >> 4;
    const int row = 16 * ch + 4 * g + (e & 3), t = row - y;
    if (t >= 0 && t <= 7 && row <= 22) v = (float)c_lumaFilter[4 * q][t] * (e >= 4 ? 128.f : 1.f);
    if (ch == 1 && g == 2 && e < 2) v = e == 0 ? 0.f : -9280.f;     // x 1024: -(64 x 1024 x 129 + 16384 x 64), the limb biases and the 16384 of u
  }
  else if (ent < MM_TBC)
  {
    const int r = ent - MM_TAC;
    if (r < 2 * 8 * 16)
    {
      const int var = r / 128, q = (r / 16) & 7, x = (r >> 1) & 7, gl = r & 1;
      if (gl == 1 && e < 2) v = e == 0 ? b0[var] : b1[var];
      else
      {
        // phases != 0: lane group pair loads columns 0..7 | 3..10, of which 3, 4 give way to the constants and 5, 6, 7 are there already: 8, 9, 10 count; phase 0: 1..8
        const int col = q == 0 ? (gl == 0 ? 1 + e : -100) : (gl == 0 ? e : (e >= 5 ? 3 + e : -100));
        const int t = col - x;
        if (t >= 0 && t <= 3) v = (float)c_chromaFilter[4 * q][t];
      }
    }
  }
  else
  {
    const int r = ent - MM_TBC, q = r / 32, y = (r >> 2) & 7, g = r & 3;
    const int row = 4 * g + (e & 3), t = row - y;
    if (t >= 0 && t <= 3 && row <= 10) v = (float)c_chromaFilter[4 * q][t] * (e >= 4 ? 128.f : 1.f);
    if (g == 3 && e < 2) v = e == 0 ? 0.f : -9280.f;
  }
  tab[i] = (_Float16)v;
}

// which of the kernel's shapes a descriptor is: 1 luma 16x16, 2 chroma 8x8 (phases on the quarter / eighth grid, bi 0 or 1), 0 other; -1: a fast SHAPE
// that is left to the generic kernel
__device__ __forceinline__ int mm_kind(const vvcgpu_mc_desc& d)
{
  if (!mc_is_fast(d.is_luma, d.w, d.h)) return 0;
  const int m = (d.frac_x0 | d.frac_y0 | (d.bi == 1 ? d.frac_x1 | d.frac_y1 : 0));
  if ((m & 3) || m < 0 || m >= (d.is_luma ? 16 : 32) || d.bi < 0 || d.bi > 1) return -1;
  return d.is_luma ? 1 : 2;
}

struct MmK                                                   // lane constants of mc_mfma_kernel
{
  const _Float16* tabS;
  int lane, c16, g, hr;
  unsigned m7[3], m8[3], orX[3], orR[3];                     // limb masks / exponent patterns of a pass-1 result by row-chunk kind (see the kernel)
  unsigned rangeMask, uLo, uHi;
  float magicN, magicH, scBi1, scUni1, ofUni1, scBi2, ofBi2, cinN, cinH;
  mm_h2 pmin, pmax, pmin0, pmaxF;                            // packed clip bounds (+ 1024): the caller's range; the bit depth's range (a uni-predictive full-sample copy is NOT clipped: filterCopy, isFirst == isLast)
  int perm;                                                  // ds_bpermute address: lane (row & 15) + 16 chunk <- lane 4 (row & 15) + chunk
  int* genCount;                                             // PUs left to the generic kernel behind this one (it leaves at once when there are none)
};

// pass-1 result registers -> limb operand (kind: 0 every row real; 1 luma rows 16..31; 2 chroma rows 0..15); hclip (per lane): the rounded horizontal-only
// filter's clip of the last stage, on u = sample + 16384
template <bool ANYH>
__device__ __forceinline__ h8 mm_limbs(const MmK& K, const f4& acc, float magic, int kind, bool hclip)
{
  unsigned u[4];
#pragma unroll
  for (int j = 0; j < 4; j++) u[j] = __builtin_bit_cast(unsigned, acc[j] + magic);
  unsigned p01 = __builtin_amdgcn_perm(u[1], u[0], 0x05040100u), p23 = __builtin_amdgcn_perm(u[3], u[2], 0x05040100u);
  if (ANYH)
  {
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const unsigned lo = hclip ? K.uLo : 0u, hi = hclip ? K.uHi : 0xFFFFFFFFu;
    p01 = __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_elementwise_max(__builtin_bit_cast(us2, p01), __builtin_bit_cast(us2, lo)), __builtin_bit_cast(us2, hi)));
    p23 = __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_elementwise_max(__builtin_bit_cast(us2, p23), __builtin_bit_cast(us2, lo)), __builtin_bit_cast(us2, hi)));
  }
  uint4 o;
  o.x = (p01 & K.m7[kind]) | K.orX[kind];
  o.y = (p23 & K.m7[kind]) | K.orR[kind];
  o.z = ((p01 >> 7) & K.m8[kind]) | K.orR[kind];
  o.w = ((p23 >> 7) & K.m8[kind]) | K.orR[kind];
  return __builtin_bit_cast(h8, o);
}
// window registers as loaded -> A operand; cst: this lane's first two columns are the product's constants (chroma) / the whole lane is (luma)
__device__ __forceinline__ h8 mm_window(const MmK& K, uint4 u, bool cst, bool wholeLane, unsigned& bad)
{
  bad |= (u.x | u.y | u.z | u.w) & ~K.rangeMask;
  const unsigned andX = cst ? 0u : 0xFFFFFFFFu, orXw = cst ? 0x64003C00u : 0x64006400u;
  const unsigned andR = (cst && wholeLane) ? 0u : 0xFFFFFFFFu, orRw = (cst && wholeLane) ? 0u : 0x64006400u;
  u.x = (u.x & andX) | orXw; u.y = (u.y & andR) | orRw; u.z = (u.z & andR) | orRw; u.w = (u.w & andR) | orRw;
  return __builtin_bit_cast(h8, u);
}
// the tail of a PU: (f0 + f1 + offset) >> shiftNum for bi, f0 for uni (w1 = 0, sc2 = 1), + 1024 so that the truncating f16 conversion is the floor and the
// clip is packed; the four samples as two dwords.  copy: a uni-predictive full-sample PU -- InterpolationFilter::filterCopy with isFirst == isLast
// stores the sample as it is (HM_JEM_CLIP_PEL, InterpolationFilter.cpp:210-227); the samples are inside the bit depth here, so "no clip" = its range
__device__ __forceinline__ uint2 mm_tail(const MmK& K, const float (&f0)[4], const float (&f1)[4], float w1, float sc2, float of2, bool copy)
{
  float t[4];
#pragma unroll
  for (int j = 0; j < 4; j++) t[j] = __builtin_fmaf(__builtin_fmaf(f1[j], w1, f0[j]), sc2, of2);
  mm_h2 q0 = __builtin_bit_cast(mm_h2, __builtin_amdgcn_cvt_pkrtz(t[0], t[1])), q1 = __builtin_bit_cast(mm_h2, __builtin_amdgcn_cvt_pkrtz(t[2], t[3]));
  const mm_h2 lo = copy ? K.pmin0 : K.pmin, hi = copy ? K.pmaxF : K.pmax;
  q0 = __builtin_elementwise_min(__builtin_elementwise_max(q0, lo), hi);
  q1 = __builtin_elementwise_min(__builtin_elementwise_max(q1, lo), hi);
  uint2 o;
  o.x = __builtin_bit_cast(unsigned, q0) & 0x03FF03FFu;
  o.y = __builtin_bit_cast(unsigned, q1) & 0x03FF03FFu;
  return o;
}
struct __attribute__((packed, aligned(2))) MmQuad { uint2 v; };
struct MmRaw { uint4 w[2][2]; unsigned dstLo, dstHi, ds, frac, flg; };    // samples as loaded [reference][row chunk] (+ chroma: the output PU's descriptor fields)
struct MmWin { h8 w[2][2]; unsigned bad, dstLo, dstHi, ds, frac, flg; };   // a step's window operands; the loaded registers are free again

// ---- luma PU: everything about the descriptor is wave-uniform (scalar registers)
// window origin of a luma PU in reference rf as (16-byte aligned pointer, shift): the window's column 0 is `shift` samples into the aligned row segment
__device__ __forceinline__ const Pel* mm_luma_origin(const vvcgpu_mc_desc& d, int rf, const Pel* __restrict__ ref0Base, const Pel* __restrict__ ref1Base, int& shift)
{
  const Pel* p0 = (rf ? ref1Base + d.ref1_off : ref0Base + d.ref0_off) - (ptrdiff_t)3 * (rf ? d.ref1_stride : d.ref0_stride) - 3;
  shift = (int)((reinterpret_cast<uintptr_t>(p0) >> 1) & 7);
  return p0 - shift;
}
// Loads in ROW-MAJOR lane order -- lane 4 r + c takes aligned word c of row r: the four lanes of a row are one 64-byte access of the vector cache, 16
// accesses per instruction instead of the 64+ that a lane per (row, eight unaligned columns) costs (measured: the cache's access rate, not bandwidth or
// latency, bound the kernel at 65 us).  Requires strides that are multiples of 8 samples (the caller of this function checks); words that hold no
// column the reference reads are not touched (the lane repeats a neighbour's word).
__device__ __forceinline__ void mm_fetch_luma(const MmK& K, const vvcgpu_mc_desc& d, const Pel* __restrict__ ref0Base, const Pel* __restrict__ ref1Base, MmRaw& r)
{
  const int nRef = d.bi == 1 ? 2 : 1;
#pragma unroll
  for (int rf = 0; rf < 2; rf++)
  {
    if (rf >= nRef) { r.w[1][0] = r.w[1][1] = make_uint4(0u, 0u, 0u, 0u); break; }     // (constants: a copy of list 0's registers would wait for its loads right here)
    int shift;
    const Pel* pA = mm_luma_origin(d, rf, ref0Base, ref1Base, shift);
    const int rs = rf ? d.ref1_stride : d.ref0_stride, fx = rf ? d.frac_x1 : d.frac_x0, fy = rf ? d.frac_y1 : d.frac_y0;
    const int cLo = fx ? 0 : (3 + shift) >> 3, cHi = fx ? (22 + shift) >> 3 : (18 + shift) >> 3;
    const int word = min(max(K.lane & 3, cLo), cHi);
#pragma unroll
    for (int ch = 0; ch < 2; ch++)
    {
      const int row = fy ? min(16 * ch + (K.lane >> 2), 22) : min(max(16 * ch + (K.lane >> 2), 3), 18);
      r.w[rf][ch] = *reinterpret_cast<const uint4*>(pA + (ptrdiff_t)row * rs + 8 * word);
    }
  }
}
__device__ __forceinline__ void mm_luma_win(const MmK& K, const MmRaw& raw, MmWin& W)
{
  W.bad = 0;
#pragma unroll
  for (int rf = 0; rf < 2; rf++)
#pragma unroll
    for (int ch = 0; ch < 2; ch++)
    {
      uint4 u = raw.w[rf][ch];
      W.bad |= (u.x | u.y | u.z | u.w) & ~K.rangeMask;       // (over the whole aligned words: a neighbour outside the bit depth costs the PU the fast path, no more)
      // operand layout: lane (row & 15) + 16 word; every sample masked to the bit depth first: 0x6400 | v must stay a finite f16 whatever the word held
      u.x = (unsigned)__builtin_amdgcn_ds_bpermute(K.perm, (int)((u.x & K.rangeMask) | 0x64006400u));
      u.y = (unsigned)__builtin_amdgcn_ds_bpermute(K.perm, (int)((u.y & K.rangeMask) | 0x64006400u));
      u.z = (unsigned)__builtin_amdgcn_ds_bpermute(K.perm, (int)((u.z & K.rangeMask) | 0x64006400u));
      u.w = (unsigned)__builtin_amdgcn_ds_bpermute(K.perm, (int)((u.w & K.rangeMask) | 0x64006400u));
      W.w[rf][ch] = __builtin_bit_cast(h8, u);
    }
}
__device__ __forceinline__ void mm_luma(const MmK& K, const vvcgpu_mc_desc& d, const MmWin& W, const Pel* __restrict__ ref0Base, const Pel* __restrict__ ref1Base,
                                        Pel* __restrict__ dstBase, int* __restrict__ flags, int idx)
{
  const int nRef = d.bi == 1 ? 2 : 1;
  const bool hOnly = d.bi == 0 && d.frac_y0 == 0 && d.frac_x0 != 0;
  float fr[2][4] = { { 0.f, 0.f, 0.f, 0.f }, { 0.f, 0.f, 0.f, 0.f } };
#pragma unroll
  for (int rf = 0; rf < 2; rf++)
  {
    if (rf >= nRef) break;
    const int fx = (rf ? d.frac_x1 : d.frac_x0) >> 2, fy = (rf ? d.frac_y1 : d.frac_y0) >> 2;
    const int shift = (int)(((reinterpret_cast<uintptr_t>(rf ? ref1Base + d.ref1_off : ref0Base + d.ref0_off) >> 1) - 3 - 3 * (rf ? d.ref1_stride : d.ref0_stride)) & 7);
    const h8 ta = *reinterpret_cast<const h8*>(K.tabS + ((MM_TAL + fx * 92 + (K.c16 + shift) * 4 + K.g) * 8));
    const float cin = hOnly ? K.cinH : K.cinN;
    const f4 cin4 = { cin, cin, cin, cin };
    const h8 tb0 = *reinterpret_cast<const h8*>(K.tabS + ((MM_TBL + fy * 128 + K.lane) * 8));
    const h8 tb1 = *reinterpret_cast<const h8*>(K.tabS + ((MM_TBL + fy * 128 + 64 + K.lane) * 8));
    const f4 a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(W.w[rf][0], ta, cin4, 0, 0, 0);
    const f4 a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(W.w[rf][1], ta, cin4, 0, 0, 0);
    h8 p0, p1;
    if (hOnly) { p0 = mm_limbs<true>(K, a0, K.magicH, 0, true); p1 = mm_limbs<true>(K, a1, K.magicH, 1, true); }
    else       { p0 = mm_limbs<false>(K, a0, K.magicN, 0, false); p1 = mm_limbs<false>(K, a1, K.magicN, 1, false); }
    f4 acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(p0, tb0, f4{ 0.f, 0.f, 0.f, 0.f }, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(p1, tb1, acc, 0, 0, 0);
    // second-stage rounding of this list: bi acc >> 6; uni (acc + offset) >> (6 + headroom); rounded horizontal-only: the pass-2 copy, acc / 64
    const float sc = (d.bi == 1 || hOnly) ? K.scBi1 : K.scUni1, of = (d.bi == 1 || hOnly) ? 0.f : K.ofUni1;
#pragma unroll
    for (int j = 0; j < 4; j++) fr[rf][j] = floorf(__builtin_fmaf(acc[j], sc, of));
  }
  const bool isBad = __ballot(W.bad != 0) != 0ull;           // a reference sample outside the bit depth: the generic kernel takes the PU
  if (K.lane == 0) { flags[idx] = isBad; if (isBad) atomicAdd(K.genCount, 1); }
  if (isBad) return;
  const uint2 o = mm_tail(K, fr[0], fr[1], d.bi == 1 ? 1.f : 0.f, d.bi == 1 ? K.scBi2 : 1.f, d.bi == 1 ? K.ofBi2 : 1024.f, d.bi == 0 && (d.frac_x0 | d.frac_y0) == 0);
  Pel* dp = dstBase + d.dst_off + (ptrdiff_t)K.c16 * d.dst_stride + 4 * K.g;                 // lane (c16, g): row c16, columns 4 g .. 4 g + 3
  reinterpret_cast<MmQuad*>(dp)->v = o;
}

// ---- chroma: PUs iA (columns 0..7 of the products) and iB (8..15; < 0: absent).  A lane reads the descriptor fields it needs from ITS PU as vector loads:
// lane groups 2, 3 LOAD the samples of PU B; lanes with c16 >= 8 hold PU B's table rows and output
struct MmCDesc { uint4 q0, q1; unsigned frac, flg, dstLo, dstHi, ds; };    // per lane: bytes 0..31 + phases / flags of the PU it LOADS, output fields of the PU it WRITES
__device__ __forceinline__ void mm_chroma_desc(const MmK& K, const vvcgpu_mc_desc* __restrict__ descs, int iA, int iB, MmCDesc& c)
{
  const int iF = (K.g >= 2 && iB >= 0) ? iB : iA, iO = (K.c16 >= 8 && iB >= 0) ? iB : iA;
  const uint4* pF = reinterpret_cast<const uint4*>(descs + iF);
  const uint4* pO = reinterpret_cast<const uint4*>(descs + iO);
  c.q0 = pF[0]; c.q1 = pF[1];
  const uint4 q2 = pF[2], q1O = pO[1], q2O = pO[2];
  c.frac = q2.z; c.flg = q2.w;
  c.dstLo = q1O.x; c.dstHi = q1O.y; c.ds = q2O.x;
  // (the output PU's phases / flags ride in the same registers as the loaded PU's when they are the same PU; otherwise in the high halves below)
  c.q1.x = q2O.z; c.q1.y = q2O.w;                            // q1.x / q1.y (dst_off of the loaded PU) are not needed: reused for the OUTPUT PU's phases / flags
}
__device__ __forceinline__ void mm_fetch_chroma(const MmK& K, const MmCDesc& c, const Pel* __restrict__ ref0Base, const Pel* __restrict__ ref1Base, MmRaw& r)
{
  r.dstLo = c.dstLo; r.dstHi = c.dstHi; r.ds = c.ds; r.frac = c.q1.x; r.flg = c.q1.y;
  const bool biF = (int)(signed char)((c.flg >> 8) & 0xFFu) == 1;
#pragma unroll
  for (int rf = 0; rf < 2; rf++)
  {
    const bool second = rf == 1 && biF;
    const long long off = second ? (long long)(((unsigned long long)c.q0.w << 32) | c.q0.z) : (long long)(((unsigned long long)c.q0.y << 32) | c.q0.x);
    const int rs = second ? (int)c.q1.w : (int)c.q1.z;
    const int fx = (int)(signed char)((c.frac >> (second ? 16 : 0)) & 0xFFu), fy = (int)(signed char)((c.frac >> (second ? 24 : 8)) & 0xFFu);
    const Pel* base = (second ? ref1Base : ref0Base) + off;
    const int col = fx ? ((K.g & 1) ? 3 : 0) : 1, row = fy ? min(K.c16, 10) : min(max(K.c16, 1), 8);
    const Pel* q = base + (ptrdiff_t)(row - 1) * rs + (col - 1);
    pel8 v;
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = q[e];
    r.w[rf][0] = __builtin_bit_cast(uint4, v);
  }
}
__device__ __forceinline__ void mm_chroma_win(const MmK& K, const MmRaw& raw, MmWin& W)
{
  W.bad = 0;
  W.w[0][0] = mm_window(K, raw.w[0][0], (K.g & 1) == 1, false, W.bad);
  W.w[1][0] = mm_window(K, raw.w[1][0], (K.g & 1) == 1, false, W.bad);
  W.dstLo = raw.dstLo; W.dstHi = raw.dstHi; W.ds = raw.ds; W.frac = raw.frac; W.flg = raw.flg;
}
__device__ __forceinline__ void mm_chroma(const MmK& K, const MmWin& raw, int iA, int iB, Pel* __restrict__ dstBase, int* __restrict__ flags)
{
  const bool hasB = iB >= 0;
  // the lane's OUTPUT PU (by c16 >> 3)
  const int bi = (int)(signed char)((raw.flg >> 8) & 0xFFu);
  const int fx0 = (int)(signed char)(raw.frac & 0xFFu) >> 2, fy0 = (int)(signed char)((raw.frac >> 8) & 0xFFu) >> 2;
  const int fx1 = (int)(signed char)((raw.frac >> 16) & 0xFFu) >> 2, fy1 = (int)(signed char)((raw.frac >> 24) & 0xFFu) >> 2;
  const bool hOnly = bi == 0 && fy0 == 0 && fx0 != 0;
  const bool anyH = __ballot(hOnly) != 0ull;
  const int nRef = __ballot(bi == 1) != 0ull ? 2 : 1;
  const bool mineCols = (K.g >> 1) == (K.c16 >> 3) && (K.c16 < 8 || hasB);      // pass 1: this lane's table row is non-zero only in the k range of ITS PU
  float fr[2][4] = { { 0.f, 0.f, 0.f, 0.f }, { 0.f, 0.f, 0.f, 0.f } };
#pragma unroll
  for (int rf = 0; rf < 2; rf++)
  {
    if (rf >= nRef) break;
    const int fx = rf ? fx1 : fx0, fy = rf ? fy1 : fy0;
    const bool hO = rf == 0 && hOnly;
    const h8 w0 = raw.w[rf][0];
    const int rowA = mineCols ? ((hO ? 128 : 0) + fx * 16 + (K.c16 & 7) * 2 + (K.g & 1)) : 2 * 8 * 16;
    const h8 ta = *reinterpret_cast<const h8*>(K.tabS + ((MM_TAC + rowA) * 8));
    const h8 tb = *reinterpret_cast<const h8*>(K.tabS + ((MM_TBC + fy * 32 + (K.c16 & 7) * 4 + K.g) * 8));
    const f4 a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, ta, f4{ 0.f, 0.f, 0.f, 0.f }, 0, 0, 0);
    const float magic = hO ? K.magicH : K.magicN;
    const h8 p0 = anyH ? mm_limbs<true>(K, a0, magic, 2, hO) : mm_limbs<false>(K, a0, magic, 2, false);
    const f4 acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(p0, tb, f4{ 0.f, 0.f, 0.f, 0.f }, 0, 0, 0);
    const float sc = (bi == 1 || hO) ? K.scBi1 : K.scUni1, of = (bi == 1 || hO) ? 0.f : K.ofUni1;
#pragma unroll
    for (int j = 0; j < 4; j++) fr[rf][j] = floorf(__builtin_fmaf(acc[j], sc, of));
  }
  const unsigned long long badLanes = __ballot(raw.bad != 0);
  // a sample outside the bit depth is no f16 integer pattern (0x6400 | v reaches the exponent): in a product it poisons every output of its ROW, also the
  // other PU's, whose table entries for it are zero (0 x NaN) -- both PUs of the pair go to the generic kernel
  const bool badA = badLanes != 0ull, badB = hasB && badLanes != 0ull;
  if (K.lane == 0) { flags[iA] = badA; if (hasB) flags[iB] = badB; if (badA) atomicAdd(K.genCount, 1 + (badB ? 1 : 0)); }
  const uint2 o = mm_tail(K, fr[0], fr[1], bi == 1 ? 1.f : 0.f, bi == 1 ? K.scBi2 : 1.f, bi == 1 ? K.ofBi2 : 1024.f, bi == 0 && (fx0 | fy0) == 0);
  // lane (c16, g): row c16 & 7 of PU c16 >> 3, columns 4 (g & 1) ..; real when g >> 1 == c16 >> 3
  const bool outB = K.c16 >= 8;
  if ((K.g >> 1) == (K.c16 >> 3) && (outB ? (hasB && !badB) : !badA))
  {
    const long long dstOff = (long long)(((unsigned long long)raw.dstHi << 32) | raw.dstLo);
    Pel* dp = dstBase + dstOff + (ptrdiff_t)(K.c16 & 7) * (int)raw.ds + 4 * (K.g & 1);
    reinterpret_cast<MmQuad*>(dp)->v = o;
  }
}

// KIND_T 1: the luma PUs of the list, 2: the chroma PUs (two launches: A/B switch), 0: ONE launch whose workgroups alternate between the two shapes -- each
// workgroup with the LDS tables and the loop of one shape.  Persistent waves; wave w of W owns the descriptors w, w + W, ... (chroma: the descriptor PAIRS,
// so that two chroma PUs share the products): neighbouring PUs at the same time in neighbouring waves, and whatever runs of shapes the list has are dealt
// round the waves.  What bound the earlier forms of this kernel, in the order found (profiles/r05_mc_forms.txt): chunks of 64 descriptors per workgroup
// (a picture's list is rows of luma, then rows of chroma PUs: a third of the waves had all the luma work); a shared claim counter (same-address atomics:
// ~30 ns each); the scalar unit (one per CU: a descriptor-by-descriptor walk cost 400 scalar instructions per PU -- now a gather load and a ballot
// classify 64 of the wave's descriptors at once); the vector cache's access rate (a lane per row and eight unaligned columns: 64+ accesses per load).
__device__ __forceinline__ int mm_kind_of(const uint4& q)   // bytes 32..47 of a descriptor: dst_stride | w, h | phases | is_luma, bi
{
  vvcgpu_mc_desc d;
  d.w = (short)(q.y & 0xFFFFu); d.h = (short)(q.y >> 16);
  d.frac_x0 = (signed char)(q.z & 0xFFu); d.frac_y0 = (signed char)((q.z >> 8) & 0xFFu); d.frac_x1 = (signed char)((q.z >> 16) & 0xFFu); d.frac_y1 = (signed char)(q.z >> 24);
  d.is_luma = (signed char)(q.w & 0xFFu); d.bi = (signed char)((q.w >> 8) & 0xFFu);
  return mm_kind(d);
}
// the generic body for the PUs this kernel cannot take, served by the wave that found them BEHIND its walk (round 6, vvcgpu_mc_picture_batch: the launch of
// mc_batch_kernel behind this kernel cost 4.9 us per 4K picture for an empty list).  The second pass is a REAL call that takes everything it needs from
// a record in LDS: values kept alive for it across the walks cost the walks' loops spilled registers (190 instead of 49 us).  Defined below mc_generic_pu.
struct MmServe { const Pel* ref0Base; const Pel* ref1Base; Pel* dstBase; const vvcgpu_mc_desc* descs; const int* flags; int n, bd, cmin, cmax, w0, W, luma; };
__device__ __forceinline__ void mm_second_pass(const MmServe* sv, short* gen, unsigned* genT);
__device__ __noinline__ void mm_serve_one(const MmServe* sv, int li, short* gen, unsigned* genT);
constexpr int MM_GEN_SHORTS = 23 * 24 + 23 * 16;                               // WR x WP + WR x ST (declared below)

template <int KIND_T>
__global__ __launch_bounds__(256, 4) void mc_mfma_kernel(const Pel* __restrict__ ref0Base, const Pel* __restrict__ ref1Base, Pel* __restrict__ dstBase,
                                                                         const vvcgpu_mc_desc* __restrict__ descs, int n, int bd, int cmin, int cmax,
                                                                         const _Float16* __restrict__ image, int* __restrict__ flags, unsigned long long* __restrict__ diag,
                                                                         int* __restrict__ genCount, int* __restrict__ nextCounters, int serve)
{
  if (blockIdx.x == 0 && threadIdx.x < VVC_CTR_INTS) nextCounters[threadIdx.x] = 0;       // the counter set of the NEXT call on this stream (vvcgpu_counters)
  // KIND_T 0: ONE launch, workgroups alternate between the two shapes (both kinds of waves on every CU at the same time)
  const int KIND = KIND_T ? KIND_T : 1 + ((int)blockIdx.x & 1);
  const int T0 = KIND == 1 ? MM_TAL : MM_TAC, T1 = KIND == 1 ? MM_TAC : MM_ENTRIES;          // this kind's table entries
  __shared__ __align__(16) _Float16 tabL[(KIND_T == 2 ? MM_ENTRIES - MM_TAC : MM_TAC - MM_TAL) * 8];
  __shared__ __align__(16) short genS[4][MM_GEN_SHORTS];                     // the generic body's window / intermediate, per wave
  __shared__ __align__(16) unsigned genT[4][MC_LDS_DW];                      // ... and its packed-form tile
  __shared__ MmServe serveS;
  __shared__ int anyGenS[4];                                 // serve: per wave, what its walk left to the generic body (the count lands here instead of in genCount)
  for (int i = threadIdx.x; i < T1 - T0; i += 256) reinterpret_cast<uint4*>(tabL)[i] = reinterpret_cast<const uint4*>(image)[T0 + i];
  if (threadIdx.x < 4) anyGenS[threadIdx.x] = 0;
  MmK K;                                                     // (the barrier behind the table copy follows the lane constants and the serve record)
  K.genCount = serve ? &anyGenS[threadIdx.x >> 6] : genCount;
  K.tabS = tabL - T0 * 8;                                    // (indexed with the image's entry numbers)
  K.lane = threadIdx.x & 63; K.c16 = K.lane & 15; K.g = K.lane >> 4;
  const int g = K.g;
  const int hr = max(2, IF_INTERNAL_PREC - bd), S = 1 << (IF_FILTER_PREC - hr);
  K.hr = hr;
  K.rangeMask = (unsigned)((1 << bd) - 1) * 0x10001u;
  K.uLo = (unsigned)(16384 + cmin) * 0x10001u; K.uHi = (unsigned)(16384 + cmax) * 0x10001u;
  // limb masks / exponent patterns of a pass-1 result by row-chunk kind: 0 every row real; 1 luma rows 16..31 (lane group 2: the constants 1.0, 1024.0;
  // 3: nothing); 2 chroma rows 0..15 (lane group 3: the constants)
  K.m7[0] = 0x007F007Fu; K.m8[0] = 0x00FF00FFu; K.orX[0] = 0x64006400u;
  asm("" : "+v"(K.m7[0]), "+v"(K.m8[0]), "+v"(K.orX[0]));   // held in vector registers (v_and_or_b32 takes no literal)
  K.orR[0] = K.orX[0];
  K.m7[1] = g < 2 ? 0x007F007Fu : 0u; K.m8[1] = g < 2 ? 0x00FF00FFu : 0u; K.orX[1] = g < 2 ? 0x64006400u : g == 2 ? 0x64003C00u : 0u; K.orR[1] = g < 2 ? 0x64006400u : 0u;
  K.m7[2] = g < 3 ? 0x007F007Fu : 0u; K.m8[2] = g < 3 ? 0x00FF00FFu : 0u; K.orX[2] = g < 3 ? 0x64006400u : 0x64003C00u; K.orR[2] = g < 3 ? 0x64006400u : 0u;
  K.magicN = 8388608.f * (float)S; K.magicH = 536870912.f;
  K.cinN = 8192.f * (float)S - 65536.f - 0.5f * (float)(S - 1); K.cinH = 983040.5f;     // start values of a luma pass-1 sum (the chroma tables carry theirs)
  K.perm = (4 * K.c16 + K.g) * 4;
  K.pmin = mm_h2{ (_Float16)(short)(1024 + cmin), (_Float16)(short)(1024 + cmin) }; K.pmax = mm_h2{ (_Float16)(short)(1024 + cmax), (_Float16)(short)(1024 + cmax) };
  K.pmin0 = mm_h2{ (_Float16)1024.f, (_Float16)1024.f }; K.pmaxF = mm_h2{ (_Float16)(short)(1023 + (1 << bd)), (_Float16)(short)(1023 + (1 << bd)) };
  // second-stage rounding as fma + floor (exact: f32 integers below 2^24 times powers of two)
  K.scBi1 = 1.f / 64.f; K.scUni1 = 1.f / (float)(64 << hr); K.ofUni1 = (float)((1 << (5 + hr)) + (IF_INTERNAL_OFFS << 6)) * K.scUni1;
  K.scBi2 = 1.f / (float)(2 << hr); K.ofBi2 = (float)((1 << hr) + 2 * IF_INTERNAL_OFFS) * K.scBi2 + 1024.f;

  const int nb = KIND_T ? (int)gridDim.x : (KIND == 1 ? ((int)gridDim.x + 1) >> 1 : (int)gridDim.x >> 1), bi_ = KIND_T ? (int)blockIdx.x : (int)blockIdx.x >> 1;
  // XCD-aware walk (workgroups are dealt round-robin over the 8 XCDs, each with its own L2): at every step the waves of ONE XCD hold one contiguous
  // run of W / 8 descriptors, so the window lines that neighbouring PUs share are fetched from the fabric by one L2 (speed only)
  const int W = nb * 4, perX = W >> 3;
  const int w = (nb & 7) ? bi_ * 4 + __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6)
                         : (bi_ & 7) * perX + (bi_ >> 3) * 4 + __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  if (serve && threadIdx.x == 0) serveS = MmServe{ ref0Base, ref1Base, dstBase, descs, flags, n, bd, cmin, cmax, w, W, KIND == 1 ? 1 : 0 };     // (w of wave 0; read behind the walks)
  __syncthreads();
  // The wave's descriptors w + j W are classified 64 at a time, one per LANE (one gather load and a ballot; a descriptor-by-descriptor walk on the scalar
  // unit -- one per CU, shared by its 20 waves -- bound the kernel: 400 scalar instructions per PU); the walk over the set bits is a few scalar operations.
  // Three steps are in flight (in-kernel stamps, VVCGPU_MC_DIAG: with the descriptor read inside the step that requests the samples, 1400 of a step's
  // 5500 cycles were that read's latency): the descriptor of step k + 2 is being read, the samples of step k + 1 are requested, step k is computed.
  MmRaw raw;
  bool pend = false;
  if (KIND == 1)
  {
    vvcgpu_mc_desc dP = descs[0];
    int iP = 0, dstep = 0;
    for (int j0 = 0; w + (long long)j0 * W < n; j0 += 64)
    {
      const long long iL = w + (long long)(j0 + K.lane) * W;
      int k = 0;
      if (iL < n)
      {
        const uint4 q1 = reinterpret_cast<const uint4*>(descs + iL)[1], q2 = reinterpret_cast<const uint4*>(descs + iL)[2];
        k = mm_kind_of(q2);
        const int bi = (int)(signed char)((q2.w >> 8) & 0xFFu);
        if (k == 1 && ((q1.z | (bi == 1 ? q1.w : 0u)) & 7u)) k = -1;           // aligned 16-byte words need rows that keep their alignment (ref strides: bytes 24..31)
        if (k < 0) flags[iL] = 1;                            // a fast SHAPE these kernels do not take: the generic kernel's
      }
      {
        const unsigned long long gm = __ballot(iL < n && k <= 0);             // every other shape, and the fast shapes left above: the generic kernel's work
        if (gm != 0ull && K.lane == 0) atomicAdd(K.genCount, (int)__popcll(gm));
      }
      unsigned long long mine = __ballot(k == 1);
      auto nextIdx = [&]() -> int { if (mine == 0ull) return -1; const int j = (int)__builtin_ctzll(mine); mine &= mine - 1ull; return w + (j0 + j) * W; };
      // A step's descriptor is wave-uniform, but it is read with VECTOR loads (every lane the same address) a step ahead and moved to scalar registers
      // when its step begins: scalar loads return out of order, so with one in flight every LDS wait of the step (operand permutes, table reads) is an
      // lgkmcnt(0) that also waits for the descriptor -- ~2000 of a step's 5000 cycles (VVCGPU_MC_DIAG stamps).
      auto descLoad = [&](int i, uint4 (&v)[3]) { const uint4* q = reinterpret_cast<const uint4*>(descs + i); v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; };
      auto descScalar = [&](const uint4 (&v)[3]) -> vvcgpu_mc_desc
      {
        unsigned u[12] = { v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, v[1].z, v[1].w, v[2].x, v[2].y, v[2].z, v[2].w };
#pragma unroll
        for (int t = 0; t < 12; t++) u[t] = (unsigned)__builtin_amdgcn_readfirstlane((int)u[t]);
        vvcgpu_mc_desc d;
        d.ref0_off = (long long)(((unsigned long long)u[1] << 32) | u[0]); d.ref1_off = (long long)(((unsigned long long)u[3] << 32) | u[2]);
        d.dst_off = (long long)(((unsigned long long)u[5] << 32) | u[4]);
        d.ref0_stride = (int)u[6]; d.ref1_stride = (int)u[7]; d.dst_stride = (int)u[8];
        d.w = (short)(u[9] & 0xFFFFu); d.h = (short)(u[9] >> 16);
        d.frac_x0 = (signed char)(u[10] & 0xFFu); d.frac_y0 = (signed char)((u[10] >> 8) & 0xFFu); d.frac_x1 = (signed char)((u[10] >> 16) & 0xFFu); d.frac_y1 = (signed char)(u[10] >> 24);
        d.is_luma = (signed char)(u[11] & 0xFFu); d.bi = (signed char)((u[11] >> 8) & 0xFFu); d.reserved = 0;
        return d;
      };
      int iA = nextIdx();
      if (iA < 0) continue;
      uint4 dv[3];
      descLoad(iA, dv);
      while (iA >= 0)
      {
        const vvcgpu_mc_desc dA = descScalar(dv);
        const int iB = nextIdx();
        descLoad(iB >= 0 ? iB : iA, dv);                     // the next step's descriptor: a whole step ahead
        if (pend)
        {
          // VVCGPU_MC_DIAG (measurement aid): core-clock stamps of one wave's steps: step start, samples arrived + operands, next samples requested, done
          const bool st = diag && bi_ == (nb >> 1) && (threadIdx.x >> 6) == 0 && dstep < 12;
          if (st && K.lane == 0) diag[dstep * 4 + 0] = __builtin_amdgcn_s_memtime();
          MmWin Wn;
          mm_luma_win(K, raw, Wn);                           // the previous PU's samples have arrived: operands; the loaded registers are free
          if (st && K.lane == 0) { asm volatile("" :: "v"(Wn.w[0][0]), "v"(Wn.w[1][1])); diag[dstep * 4 + 1] = __builtin_amdgcn_s_memtime(); }
          mm_fetch_luma(K, dA, ref0Base, ref1Base, raw);     // this PU's samples travel behind the previous PU's products
          if (st && K.lane == 0) diag[dstep * 4 + 2] = __builtin_amdgcn_s_memtime();
          mm_luma(K, dP, Wn, ref0Base, ref1Base, dstBase, flags, iP);
          if (st && K.lane == 0) diag[dstep * 4 + 3] = __builtin_amdgcn_s_memtime();
          dstep++;
        }
        else mm_fetch_luma(K, dA, ref0Base, ref1Base, raw);
        dP = dA; iP = iA; pend = true;
        iA = iB;
      }
    }
    if (pend) { MmWin Wn; mm_luma_win(K, raw, Wn); mm_luma(K, dP, Wn, ref0Base, ref1Base, dstBase, flags, iP); }
    if (serve && anyGenS[(int)threadIdx.x >> 6] != 0) mm_second_pass(&serveS, genS[(int)threadIdx.x >> 6], genT[(int)threadIdx.x >> 6]);
  }
  else
  {
    const int units = (n + 1) >> 1;
    int iAP = -1, iBP = -1;
    for (int j0 = 0; w + (long long)j0 * W < units; j0 += 64)
    {
      const long long uL = w + (long long)(j0 + K.lane) * W;
      int iAv = -1, iBv = -1;
      if (uL < units)
      {
        const int k0 = mm_kind_of(reinterpret_cast<const uint4*>(descs + 2 * uL)[2]);
        const int k1 = 2 * uL + 1 < n ? mm_kind_of(reinterpret_cast<const uint4*>(descs + 2 * uL + 1)[2]) : 0;
        iAv = k0 == 2 ? (int)(2 * uL) : k1 == 2 ? (int)(2 * uL + 1) : -1;
        iBv = (k0 == 2 && k1 == 2) ? (int)(2 * uL + 1) : -1;
      }
      unsigned long long mine = __ballot(iAv >= 0);
      auto nextJ = [&]() -> int { if (mine == 0ull) return -1; const int j = (int)__builtin_ctzll(mine); mine &= mine - 1ull; return j; };
      int jA = nextJ();
      if (jA < 0) continue;
      int iA = __builtin_amdgcn_readlane(iAv, jA), iB = __builtin_amdgcn_readlane(iBv, jA);
      MmCDesc cA;
      mm_chroma_desc(K, descs, iA, iB, cA);
      while (jA >= 0)
      {
        const int jN = nextJ();
        const int iAN = __builtin_amdgcn_readlane(iAv, jN >= 0 ? jN : jA), iBN = __builtin_amdgcn_readlane(iBv, jN >= 0 ? jN : jA);
        MmCDesc cN;
        mm_chroma_desc(K, descs, iAN, iBN, cN);              // the descriptor fields of the step after this one (vector loads, consumed next iteration)
        if (pend)
        {
          MmWin Wn;
          mm_chroma_win(K, raw, Wn);
          mm_fetch_chroma(K, cA, ref0Base, ref1Base, raw);
          mm_chroma(K, Wn, iAP, iBP, dstBase, flags);
        }
        else mm_fetch_chroma(K, cA, ref0Base, ref1Base, raw);
        iAP = iA; iBP = iB; pend = true;
        jA = jN; iA = iAN; iB = iBN; cA = cN;
      }
    }
    if (pend) { MmWin Wn; mm_chroma_win(K, raw, Wn); mm_chroma(K, Wn, iAP, iBP, dstBase, flags); }
    if (serve && anyGenS[(int)threadIdx.x >> 6] != 0) mm_second_pass(&serveS, genS[(int)threadIdx.x >> 6], genT[(int)threadIdx.x >> 6]);
  }
}

// the table image of mc_mfma_kernel per device and bit depth
static const _Float16* mm_image(int bd)
{
  static std::mutex mtx;
  static _Float16* images[64][3] = { { nullptr } };
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { vvcgpu_set_error("mc image: device index"); return nullptr; }
  std::lock_guard<std::mutex> lock(mtx);
  _Float16*& slot = images[dev][bd - 8];
  if (!slot)
  {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, MM_ENTRIES * 16);
    if (e != hipSuccess) { (void)hipGetLastError(); vvcgpu_set_error("mc image: hipMalloc failed: %s", hipGetErrorString(e)); return nullptr; }
    hipLaunchKernelGGL(mm_build_tables_kernel, dim3(cdiv(MM_ENTRIES * 8, 256)), dim3(256), 0, (hipStream_t)0, static_cast<_Float16*>(p), bd);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) { (void)hipFree(p); vvcgpu_set_error("building the motion-compensation table image failed: %s", hipGetErrorString(e)); return nullptr; }
    slot = static_cast<_Float16*>(p);
  }
  return slot;
}

// four 4x4 luma PUs side by side, sixteen lanes each, through the packed code of the fast kernel (N = 8 taps, tile 4, 16 lanes: 11 window rows of
// 6 dwords, 11 first-pass items, 4 second-pass items per PU).  Affine prediction is made of these: a whole wave per sub-block through the
// sample-wise body was 0.6 ms for the 518 k sub-blocks of a 4K picture.  Only in the SUB44 variant of the generic kernel (the affine entry points
// launch it): with this path the kernel needs 214 VGPRs instead of 127 -- inline or as a real call -- which costs every other PU size a wave per SIMD.
constexpr int MC44_DW = 11 * 6 + 4 * 6 + 8;                               // per group: window, transposed intermediate (4 x 12 shorts), output (16 shorts)
// Returns the PUs (bits of todo44) whose windows hold a sample outside the bit depth: the packed two-pass form is not the reference's for those (mc_stage),
// the caller serves them with the sample-wise body.
__device__ __forceinline__ unsigned long long mc_luma4x4_chunk(unsigned long long todo44, const vvcgpu_mc_desc* __restrict__ descs, const Pel* __restrict__ ref0Base,
                                                 const Pel* __restrict__ ref1Base, Pel* __restrict__ dstBase, int bd, int cmin, int cmax, int lane, unsigned* tileL)
{
  unsigned* L = tileL + (lane >> 4) * MC44_DW;
  const int gl = lane & 15;
  // the next four PUs' descriptors and windows are requested before the current four are computed
  auto pick = [&](int& sel)
  {
    const int firstSel = (int)__builtin_ctzll(todo44);
    sel = -1;
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (todo44) { const int b = (int)__builtin_ctzll(todo44); todo44 &= todo44 - 1ull; if (k == (lane >> 4)) sel = b; }
    return descs[sel >= 0 ? sel : firstSel];
  };
  int selC, selN = -1;
  vvcgpu_mc_desc dC = pick(selC), dN = dC;
  McStaged<8, 4, 16> sC, sN;
  mc_stage<8, 4, 16>(dC, selC >= 0, ref0Base, ref1Base, gl, sC, bd);
  unsigned redoLo = 0u, redoHi = 0u;
  for (;;)
  {
    const bool more = todo44 != 0ull;
    if (more) { dN = pick(selN); mc_stage<8, 4, 16>(dN, selN >= 0, ref0Base, ref1Base, gl, sN, bd); }
    const unsigned long long bm = __builtin_amdgcn_ballot_w64(selC >= 0 && sC.bad != 0u);
    const bool grpBad = ((bm >> (lane & 48)) & 0xFFFFull) != 0ull;           // this lane group's PU
    if (grpBad && selC >= 0) { if (selC < 32) redoLo |= 1u << selC; else redoHi |= 1u << (selC - 32); }
    mc_tile_dot2<8, 4, 16>(dC, selC >= 0 && !grpBad, sC, dstBase, bd, cmin, cmax, gl, L, reinterpret_cast<short*>(L + 11 * 6), reinterpret_cast<short*>(L + 11 * 6 + 4 * 6));
    if (!more) break;
    dC = dN; selC = selN; sC = sN;
  }
  unsigned lo = 0u, hi = 0u;
#pragma unroll
  for (int g4 = 0; g4 < 64; g4 += 16) { lo |= (unsigned)__builtin_amdgcn_readlane((int)redoLo, g4); hi |= (unsigned)__builtin_amdgcn_readlane((int)redoHi, g4); }
  return ((unsigned long long)hi << 32) | lo;
}

// One PU through the generic body: tiles of the packed form where the PU is a grid of them, else (or when a sample leaves the bit depth) sample by sample.
// ONE wave; win / tmp / tileL: that wave's LDS (WR x WP, WR x ST shorts, MC_LDS_DW dwords); DIST: the prediction goes to predT (pitch d.w) instead of dst.
// TAG separates the copies by caller: a function that is not inlined takes the loosest register budget of the kernels that call it.
#define MC_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)
template <bool DIST, int TAG>
__device__ __forceinline__ void mc_generic_pu(const vvcgpu_mc_desc& d, const Pel* __restrict__ ref0Base, const Pel* __restrict__ ref1Base, Pel* __restrict__ dstBase,
                                              int bd, int cmin, int cmax, int lane, short* win, short* tmp, unsigned* tileL, short* predT)
{
  // descriptors live in device memory, the host cannot validate them: a shape outside the contract (the prediction tile of the fused form is
  // 128 x 128, bi is 0 or 1 there) is skipped with the sentinel ~0 as its distortion instead of overrunning LDS (wave-uniform)
  // a PU whose sides are multiples of the packed path's tile (16 luma / 8 chroma samples) is a grid of tiles with the same fractional phase: the
  // wave walks them with the packed code of the fast kernel (here, not there: inlined into the fast kernel the loop cost it its 80-VGPR budget and
  // the MC stage of the canonical workload went from 0.075 to 0.18 ms)
  if (!DIST && (d.w % (d.is_luma ? 16 : 8)) == 0 && (d.h % (d.is_luma ? 16 : 8)) == 0)
  {
    const int T = d.is_luma ? 16 : 8, tx = d.w / T, nt = tx * (d.h / T);
    auto tile = [&](int t) { vvcgpu_mc_desc q = d; const int y = (t / tx) * T, x = (t - (t / tx) * tx) * T; q.w = q.h = (short)T;
                             q.ref0_off += (int64_t)y * d.ref0_stride + x; q.ref1_off += (int64_t)y * d.ref1_stride + x; q.dst_off += (int64_t)y * d.dst_stride + x; return q; };
    unsigned* L = tileL;
    bool outsideDepth = false;                               // a reference sample outside the bit depth (mc_stage): the whole PU again, sample-wise, below
    if (d.is_luma)
    {
      for (int t = 0; t < nt; t++)
      {
        const vvcgpu_mc_desc q = tile(t);
        McStaged<8, 16, 64> st;
        mc_stage<8, 16, 64>(q, true, ref0Base, ref1Base, lane, st, bd);
        if (__builtin_amdgcn_ballot_w64(st.bad != 0u) != 0ull) { outsideDepth = true; break; }
        mc_tile_dot2<8, 16, 64>(q, true, st, dstBase, bd, cmin, cmax, lane, L, reinterpret_cast<short*>(L + 23 * 12), reinterpret_cast<short*>(L + 23 * 12 + 16 * 12));
      }
    }
    else
    {
      const bool hi = lane >= 32;                            // two chroma tiles side by side in the wave's halves
      unsigned* Lh = L + (hi ? MC_LDS_DW / 2 : 0);
      for (int t = 0; t < nt; t += 2)
      {
        const bool on = t + (hi ? 1 : 0) < nt;
        const vvcgpu_mc_desc q = tile(on ? t + (hi ? 1 : 0) : t);
        McStaged<4, 8, 32> st;
        mc_stage<4, 8, 32>(q, on, ref0Base, ref1Base, lane & 31, st, bd);
        if (__builtin_amdgcn_ballot_w64(st.bad != 0u) != 0ull) { outsideDepth = true; break; }
        mc_tile_dot2<4, 8, 32>(q, on, st, dstBase, bd, cmin, cmax, lane & 31, Lh, reinterpret_cast<short*>(Lh + 11 * 6), reinterpret_cast<short*>(Lh + 11 * 6 + 8 * 6));
      }
    }
    if (!outsideDepth) return;
  }
  const int N = d.is_luma ? 8 : 4, half = N / 2 - 1;
  const bool rndRes = d.bi == 0;
  const int nRef = d.bi == 1 ? 2 : 1;

  for (int sy = 0; sy < d.h; sy += ST)
    for (int sx = 0; sx < d.w; sx += ST)
    {
      const int tw = min(ST, d.w - sx), th = min(ST, d.h - sy);
      const int npx = tw * th;
      int pred[2][4];
#pragma unroll
      for (int r = 0; r < 2; r++)
      {
        if (r >= nRef) break;
        const Pel* ref = (r ? ref1Base + d.ref1_off : ref0Base + d.ref0_off) + (size_t)sy * (r ? d.ref1_stride : d.ref0_stride) + sx;
        const int rs = r ? d.ref1_stride : d.ref0_stride;
        const int fx = r ? d.frac_x1 : d.frac_x0, fy = r ? d.frac_y1 : d.frac_y0;
        const short* cx = d.is_luma ? c_lumaFilter[fx] : c_chromaFilter[fx];
        const short* cy = d.is_luma ? c_lumaFilter[fy] : c_chromaFilter[fy];
        // stage only what the branch needs: rows [-half, th+N-1-half) when fy != 0, cols likewise when fx != 0
        const int r0 = fy ? -half : 0, nr = fy ? th + N - 1 : th;
        const int c0 = fx ? -half : 0, nc = fx ? tw + N - 1 : tw;
        MC_WAVE_SYNC();                               // previous users of win/tmp are done
        for (int i = lane; i < nr * nc; i += 64)
        {
          const int rr = i / nc, cc = i - rr * nc;
          win[rr * WP + cc] = ref[(ptrdiff_t)(r0 + rr) * rs + c0 + cc];
        }
        MC_WAVE_SYNC();
        if (fx && fy)
        {
          const IfMode mh = if_mode(true, false, bd);
          for (int i = lane; i < nr * tw; i += 64)
          {
            const int rr = i / tw, x = i - rr * tw;
            int sum = 0;
            for (int k = 0; k < N; k++) sum += win[rr * WP + x + k] * cx[k];
            tmp[rr * ST + x] = (short)((sum + mh.offset) >> mh.shift);
          }
          MC_WAVE_SYNC();
          const IfMode mv = if_mode(false, rndRes, bd);
#pragma unroll
          for (int j = 0; j < 4; j++)
          {
            const int p = lane + 64 * j;
            if (p < npx)
            {
              const int y = p / tw, x = p - y * tw;
              int sum = 0;
              for (int k = 0; k < N; k++) sum += tmp[(y + k) * ST + x] * cy[k];
              int v = (short)((sum + mv.offset) >> mv.shift);
              if (rndRes) v = clip3(cmin, cmax, v);
              pred[r][j] = v;
            }
          }
        }
        else
        {
          const IfMode m1 = if_mode(true, rndRes, bd);
#pragma unroll
          for (int j = 0; j < 4; j++)
          {
            const int p = lane + 64 * j;
            if (p < npx)
            {
              const int y = p / tw, x = p - y * tw;
              int v;
              if (!fx && !fy) v = if_copy(win[y * WP + x], true, rndRes, bd, cmin, cmax);
              else
              {
                int sum = 0;
                if (fx) { for (int k = 0; k < N; k++) sum += win[y * WP + x + k] * cx[k]; }
                else    { for (int k = 0; k < N; k++) sum += win[(y + k) * WP + x] * cy[k]; }
                v = (short)((sum + m1.offset) >> m1.shift);
                if (rndRes) v = clip3(cmin, cmax, v);
              }
              pred[r][j] = v;
            }
          }
        }
      }
      Pel* dst = DIST ? predT + sy * d.w + sx : dstBase + d.dst_off + (size_t)sy * d.dst_stride + sx;
      const int dstStride = DIST ? (int)d.w : d.dst_stride;
      const int shiftNum = max(2, IF_INTERNAL_PREC - bd) + 1, offset = (1 << (shiftNum - 1)) + 2 * IF_INTERNAL_OFFS;
#pragma unroll
      for (int j = 0; j < 4; j++)
      {
        const int p = lane + 64 * j;
        if (p < npx)
        {
          const int y = p / tw, x = p - y * tw;
          int v = pred[0][j];
          if (d.bi == 1) v = clip3(cmin, cmax, (pred[0][j] + pred[1][j] + offset) >> shiftNum);
          dst[(size_t)y * dstStride + x] = (short)v;
        }
      }
    }
}

// (a real call: it runs for the rare PU only.  The scan below is inline in the kernel -- as a call of its own it saved and restored ~90 callee-saved
// registers through scratch memory in EVERY wave: 190 MB of traffic, 30 us)
__device__ __noinline__ void mm_serve_one(const MmServe* sv, int li, short* gen, unsigned* genT)
{
  static_assert(MM_GEN_SHORTS == WR * WP + WR * ST, "LDS of the generic body inside mc_mfma_kernel");
  const MmServe S = *sv;
  const vvcgpu_mc_desc d = S.descs[li];
  mc_generic_pu<false, 1>(d, S.ref0Base, S.ref1Base, S.dstBase, S.bd, S.cmin, S.cmax, (int)threadIdx.x & 63, gen, gen + WR * WP, genT, nullptr);
}
__device__ __forceinline__ void mm_second_pass(const MmServe* sv, short* gen, unsigned* genT)
{
  // flags[] was written by this wave in its walk: its stores are complete behind the wait (vector stores write through to the L2), and the loads below are
  // agent-scope atomic loads (served by the L2).  NOT __threadfence(): on this chip that is an L2 write-back + invalidate per wave -- 4096 of them took
  // the kernel from 49 to 245 us
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  const MmServe S = *sv;
  const int lane = (int)threadIdx.x & 63, w = S.w0 + ((int)threadIdx.x >> 6);
  auto serve = [&](int li) { mm_serve_one(sv, li, gen, genT); };
  if (S.luma)
  {
    // what this wave's descriptors leave to the generic body: every shape that is not a fast one (luma or chroma), the fast shapes the walk rejected
    // (phases, strides) and the luma PUs whose samples left the bit depth
    for (int j0 = 0; w + (long long)j0 * S.W < S.n; j0 += 64)
    {
      const long long iL = w + (long long)(j0 + lane) * S.W;
      bool gen1 = false;
      if (iL < S.n)
      {
        const uint4 q1 = reinterpret_cast<const uint4*>(S.descs + iL)[1], q2 = reinterpret_cast<const uint4*>(S.descs + iL)[2];
        int k = mm_kind_of(q2);
        const int bi = (int)(signed char)((q2.w >> 8) & 0xFFu);
        if (k == 1 && ((q1.z | (bi == 1 ? q1.w : 0u)) & 7u)) k = -1;
        gen1 = k <= 0 || (k == 1 && __hip_atomic_load(S.flags + iL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0);
      }
      unsigned long long todo = __ballot(gen1);
      while (todo)
      {
        const int j = (int)__builtin_ctzll(todo);
        todo &= todo - 1ull;
        serve(w + (j0 + j) * S.W);
      }
    }
  }
  else
  {
    // the chroma PUs of this wave's pairs whose samples left the bit depth
    const int units = (S.n + 1) >> 1;
    for (int j0 = 0; w + (long long)j0 * S.W < units; j0 += 64)
    {
      const long long uL = w + (long long)(j0 + lane) * S.W;
      int g0 = -1, g1 = -1;
      if (uL < units)
      {
        const int k0 = mm_kind_of(reinterpret_cast<const uint4*>(S.descs + 2 * uL)[2]);
        const int k1 = 2 * uL + 1 < S.n ? mm_kind_of(reinterpret_cast<const uint4*>(S.descs + 2 * uL + 1)[2]) : 0;
        if (k0 == 2 && __hip_atomic_load(S.flags + 2 * uL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) g0 = (int)(2 * uL);
        if (k1 == 2 && __hip_atomic_load(S.flags + 2 * uL + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) g1 = (int)(2 * uL + 1);
      }
      unsigned long long todo = __ballot(g0 >= 0 || g1 >= 0);
      while (todo)
      {
        const int j = (int)__builtin_ctzll(todo);
        todo &= todo - 1ull;
        const int a0 = __builtin_amdgcn_readlane(g0, j), a1 = __builtin_amdgcn_readlane(g1, j);
        if (a0 >= 0) serve(a0);
        if (a1 >= 0) serve(a1);
      }
    }
  }
}

// generic kernel: any size, one wave per PU, persistent over the list of PUs the fast kernel left.
// DIST (vvcgpu_mc_dist_batch): the prediction goes into an LDS tile instead of dst, and the wave returns its distortion against the original
// (descriptor field dst_off / dst_stride = the original block, reserved = row sub-sampling shift of the SAD); list == nullptr: every descriptor.
template <bool DIST, bool SUB44 = false>
__global__ __launch_bounds__(64) void mc_batch_kernel(const Pel* __restrict__ ref0Base, const Pel* __restrict__ ref1Base,
                                                      Pel* __restrict__ dstBase, const vvcgpu_mc_desc* __restrict__ descs,
                                                      int bd, int cmin, int cmax,
                                                      int nDirect, int distKind, const Pel* __restrict__ orgBase, unsigned long long* __restrict__ out, int chunk,
                                                      const int* __restrict__ flags, int takeFast, const int* __restrict__ genCount = nullptr)
{
  // behind the matrix-core kernel: nothing left for this one (a picture of conforming 16x16 / 8x8 PUs) -- every wave leaves at once
  if (!DIST && genCount && __builtin_amdgcn_readfirstlane(*genCount) == 0) return;
  __shared__ short win[WR * WP];
  __shared__ short tmp[WR * ST];
  __shared__ __align__(16) short predT[DIST ? 128 * 128 : 8];
  __shared__ __align__(16) unsigned tileL[DIST ? 4 : MC_LDS_DW];
  const int lane = threadIdx.x;
  // !DIST: a wave looks at `chunk` consecutive descriptors at a time, one per lane, and serves those the fast kernel leaves.  (A list of them filled
  // by the fast kernel cost one same-address atomic per PU: 6 of the 7 ms of an affine prediction of 518 k 4x4 sub-blocks.)  chunk (host): 64 for
  // long lists, down to 1 for short ones -- a wave serves its PUs one after the other, and a short list of large PUs needs every wave it can get
  // (1947 PUs of 64x64 in chunks of 64: 31 busy waves, 2.2 ms instead of 0.08).
  const int STEP = DIST ? 1 : chunk;
  for (int base0 = (int)blockIdx.x * STEP; base0 < nDirect; base0 += (int)gridDim.x * STEP)
  {
  unsigned long long todo = 1ull;
  if (!DIST)
  {
    bool mine = false, sub44 = false;
    if (lane < chunk && base0 + lane < nDirect)
    {
      const uint4 q = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(descs + base0 + lane) + 32);     // dst_stride | w, h | phases | is_luma, bi
      const int isLuma = (int)(signed char)(q.w & 0xFFu), qw = (int)(short)(q.y & 0xFFFFu), qh = (int)(short)(q.y >> 16);
      mine = takeFast || !mc_is_fast(isLuma, qw, qh) || (flags && flags[base0 + lane] != 0);       // (flags: the fast shapes the matrix-core kernel in front has left; takeFast: there is none)
      sub44 = SUB44 && isLuma && qw == 4 && qh == 4;
    }
    todo = __builtin_amdgcn_ballot_w64(mine);
    unsigned long long todo44 = __builtin_amdgcn_ballot_w64(sub44);        // 4x4 luma (affine sub-blocks): four at a time, see mc_luma4x4_quad
    todo &= ~todo44;
    if (SUB44 && todo44) todo |= mc_luma4x4_chunk(todo44, descs + base0, ref0Base, ref1Base, dstBase, bd, cmin, cmax, lane, tileL);
  }
  while (todo)
  {
  const int li = base0 + (DIST ? 0 : (int)__builtin_ctzll(todo));
  todo &= todo - 1ull;
  const vvcgpu_mc_desc d = descs[li];
  // descriptors live in device memory, the host cannot validate them: a shape outside the contract (the prediction tile of the fused form is
  // 128 x 128, bi is 0 or 1 there) is skipped with the sentinel ~0 as its distortion instead of overrunning LDS (wave-uniform)
  if (DIST && (d.w < 1 || d.h < 1 || d.w > 128 || d.h > 128 || d.bi < 0 || d.bi > 1))
  {
    if (lane == 0) out[li] = ~0ull;
    continue;
  }
  mc_generic_pu<DIST, 0>(d, ref0Base, ref1Base, dstBase, bd, cmin, cmax, lane, win, tmp, tileL, predT);
  if (DIST)
  {
    __syncthreads();                                   // the tile is complete
    const Pel* org = orgBase + d.dst_off;
    const int os = d.dst_stride, w = d.w, h = d.h;
    unsigned long long res;
    typedef const __attribute__((address_space(3))) short* LdsPel;
    if (distKind == 1) res = satd_block<64, LdsPel>(org, os, (LdsPel)predT, w, w, h, lane);
    else
    {
      const int ss = distKind == 0 ? d.reserved : 0, rows = h >> ss;
      unsigned long long acc = 0;
      for (int i = lane; i < rows * w; i += 64)
      {
        const int r = i / w, x = i - r * w;
        const int df = (int)org[(size_t)(r << ss) * os + x] - (int)predT[(r << ss) * w + x];
        acc += distKind == 2 ? (unsigned)(df * df) : (unsigned)abs(df);
      }
      res = wave_sum_u64(acc) << ss;
    }
    if (lane == 0) out[li] = res;
    __syncthreads();                                   // before the next descriptor overwrites the tile
  }
  }
  }
}

// ------------------------------------------------------------------------------------------------ B1-B4
__device__ __forceinline__ int pelop_apply(int op, int a, int b, const vvcgpu_pelop_cfg& c)
{
  switch (op)
  {
  case 0: return clip3(c.clp_min, c.clp_max, (a + b + c.offset) >> c.shift);
  case 1: return clip3(c.clp_min, c.clp_max, a + b);
  case 2: { const int t = (c.shift >= 0 ? (c.scale * a) >> c.shift : (c.scale * a) << -c.shift) + c.offset;
            return c.clip ? clip3(c.clp_min, c.clp_max, t) : t; }
  case 3: return a - b;
  case 4: return c.clip ? clip3(c.clp_min, c.clp_max, 2 * a - b) : 2 * a - b;
  default: return clip3(c.clp_min, c.clp_max, a);
  }
}

// rows [y0, y1) of one descriptor by `nthr` threads (t = index among them); each lane moves 8 samples (16 bytes) per access when the three
// operands allow it
__device__ __forceinline__ void pelop_one(int op, const vvcgpu_pelop_desc& d, const Pel* __restrict__ s0Base, const Pel* __restrict__ s1Base, Pel* dstBase,
                                          const vvcgpu_pelop_cfg& c, int t, int nthr, int y0, int y1)
{
  const Pel* s0 = s0Base + d.src0_off;
  const Pel* s1 = s1Base ? s1Base + d.src1_off : nullptr;
  Pel* dst = dstBase + d.dst_off;
  const bool vec = ((d.w & 7) == 0) && ((d.src0_stride & 7) == 0) && ((d.dst_stride & 7) == 0) &&
                   ((reinterpret_cast<uintptr_t>(s0) & 15) == 0) && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) &&
                   (!s1 || (((d.src1_stride & 7) == 0) && ((reinterpret_cast<uintptr_t>(s1) & 15) == 0)));
  if (vec)
  {
    const int wv = d.w >> 3;
    for (int i = t + y0 * wv; i < wv * y1; i += nthr)
    {
      const int y = i / wv, x = (i - y * wv) << 3;
      const pel8 a = *reinterpret_cast<const pel8*>(s0 + (size_t)y * d.src0_stride + x);
      pel8 b = { 0, 0, 0, 0, 0, 0, 0, 0 };
      if (s1) b = *reinterpret_cast<const pel8*>(s1 + (size_t)y * d.src1_stride + x);
      pel8 r;
#pragma unroll
      for (int k = 0; k < 8; k++) r[k] = (short)pelop_apply(op, a[k], b[k], c);
      *reinterpret_cast<pel8*>(dst + (size_t)y * d.dst_stride + x) = r;
    }
    return;
  }
  for (int i = t + y0 * d.w; i < d.w * y1; i += nthr)
  {
    const int y = i / d.w, x = i - y * d.w;
    const int a = s0[(size_t)y * d.src0_stride + x];
    const int b = s1 ? s1[(size_t)y * d.src1_stride + x] : 0;
    dst[(size_t)y * d.dst_stride + x] = (short)pelop_apply(op, a, b, c);
  }
}

// perWg == 1: one workgroup per descriptor (the reference calls these per CU: up to 128x128 samples; few descriptors = plane-sized bands: gridDim.y
// workgroups share a descriptor, each takes a contiguous share of the rows).  perWg > 1 (long lists: blocks): a workgroup takes perWg consecutive
// descriptors, one per wave at a time -- a workgroup per 8 x 4 block is bound by the dispatcher.
__global__ __launch_bounds__(256) void pelop_batch_kernel(int op, const Pel* __restrict__ s0Base, const Pel* __restrict__ s1Base,
                                                          Pel* dstBase, const vvcgpu_pelop_desc* __restrict__ descs, int n,
                                                          vvcgpu_pelop_cfg c, int perWg, int* __restrict__ heavyCount, int* __restrict__ heavyList,
                                                          int* __restrict__ nextCounters)
{
  const int tid = threadIdx.x;
  if (nextCounters && blockIdx.x == 0 && blockIdx.y == 0 && tid < VVC_CTR_INTS) nextCounters[tid] = 0;   // the counter set of the next call on this stream
  if (perWg > 1)                                          // the workgroup bins its perWg (<= 64) consecutive descriptors first, as if_batch_kernel does:
  {                                                       // at most 256 samples: four side by side in a wave; larger: a wave each; heavy: listed
    __shared__ unsigned char sList[64], mList[64];
    __shared__ int cntS, cntM;
    const int lane = tid & 63, wave = tid >> 6;
    const int base = blockIdx.x * perWg;
    if (wave == 0)
    {
      const int di = base + lane;
      int sz = 0;
      if (lane < perWg && di < n) sz = (int)descs[di].w * (int)descs[di].h;
      const bool small = sz > 0 && sz <= 256, heavy = sz > IF_HEAVY, med = sz > 0 && !small && !heavy;
      const unsigned long long below = (1ull << lane) - 1ull;
      const unsigned long long ms = __builtin_amdgcn_ballot_w64(small), mm = __builtin_amdgcn_ballot_w64(med), mh = __builtin_amdgcn_ballot_w64(heavy);
      if (small) sList[__popcll(ms & below)] = (unsigned char)lane;
      if (med) mList[__popcll(mm & below)] = (unsigned char)lane;
      if (lane == 0) { cntS = (int)__popcll(ms); cntM = (int)__popcll(mm); }
      if (mh != 0ull)                                     // bands over many waves (pelop_heavy_kernel): one wave per 128 x 128 block was the launch's run time
      {
        int b = 0;
        if (lane == 0) b = atomicAdd(heavyCount, (int)__popcll(mh));
        b = __builtin_amdgcn_readfirstlane(b);
        if (heavy) heavyList[b + (int)__popcll(mh & below)] = di;
      }
    }
    __syncthreads();
    const int nS = cntS, nM = cntM;
    for (int g0 = wave * 4; g0 < nS; g0 += 16)
    {
      const int k = g0 + (lane >> 4);
      if (k < nS) { const vvcgpu_pelop_desc mine = descs[base + sList[k]]; pelop_one(op, mine, s0Base, s1Base, dstBase, c, lane & 15, 16, 0, mine.h); }
    }
    for (int k = wave; k < nM; k += 4)
    {
      const vvcgpu_pelop_desc d = descs[base + mList[k]];
      pelop_one(op, d, s0Base, s1Base, dstBase, c, lane, 64, 0, d.h);
    }
    return;
  }
  const vvcgpu_pelop_desc d = descs[blockIdx.x];
  const int rows = (d.h + (int)gridDim.y - 1) / (int)gridDim.y, y0 = (int)blockIdx.y * rows, y1 = min(d.h, y0 + rows);
  if (y0 >= y1) return;
  pelop_one(op, d, s0Base, s1Base, dstBase, c, tid, 256, y0, y1);
}

// one wave per (heavy block, band of rows)
__global__ __launch_bounds__(256) void pelop_heavy_kernel(int op, const Pel* __restrict__ s0Base, const Pel* __restrict__ s1Base, Pel* dstBase,
                                                          const vvcgpu_pelop_desc* __restrict__ descs, vvcgpu_pelop_cfg c,
                                                          const int* __restrict__ heavyCount, const int* __restrict__ heavyList)
{
  const int lane = threadIdx.x & 63;
  const int cnt = heavyCount[0], waves = gridDim.x * 4;
  for (int p = blockIdx.x * 4 + (threadIdx.x >> 6); p < cnt * 16; p += waves)       // band-major pairs (see if_heavy_kernel)
  {
    const int band = p / cnt;
    const vvcgpu_pelop_desc d = descs[heavyList[p - band * cnt]];
    const int br = if_band_rows(d.w, d.h), r0 = band * br;
    if (r0 < d.h) pelop_one(op, d, s0Base, s1Base, dstBase, c, lane, 64, r0, min((int)d.h, r0 + br));
  }
}

}  // namespace

__attribute__((visibility("hidden"))) int vvcgpu_mc_image_build(int bit_depth) { return mm_image(bit_depth) ? VVCGPU_OK : VVCGPU_E_DEVICE; }

extern "C" {

int vvcgpu_if_batch(const vvc_pel* src_base, vvc_pel* dst_base, const vvcgpu_if_desc* descs, int n,
                    int bit_depth, int clp_min, int clp_max, void* stream)
{
  VVC_CHECK_ARG(n >= 0, "if_batch: n %d", n);
  if (n == 0) return VVCGPU_OK;
  VVC_CHECK_ARG(src_base && dst_base && descs, "if_batch: null pointer");
  if (bit_depth > 10 || bit_depth < 8) { vvcgpu_set_error("if_batch: bit depth %d outside 8..10", bit_depth); return VVCGPU_E_UNSUPPORTED; }
  hipStream_t st = (hipStream_t)stream;
  int* heavyList = static_cast<int*>(vvcgpu_scratch(st, sizeof(int) * (size_t)n));
  if (!heavyList) return VVCGPU_E_DEVICE;
  int cur = 0;
  int* counters = vvcgpu_counters(st, &cur);
  if (!counters) return VVCGPU_E_DEVICE;
  VVC_CHECK_ARG(n < (1 << 27), "if_batch: n %d", n);
  int perWg = IF_WG_DESCS;                                // fewer descriptors per workgroup when 64 would leave compute units without one
  while (perWg > 16 && cdiv(n, perWg) < 4096) perWg >>= 1;
  // few calls: a heavy one may be most of the work of the whole launch -> bands over the machine; very many calls: the second launch is amortised and its
  // machine-wide bands beat a workgroup's four waves (real call mix: 0.042 -> 0.038 ms at 30 k calls, 0.093 -> 0.110 at 121 k)
  const int localHeavy = n >= 8192 && n < 65536;
  hipLaunchKernelGGL(if_batch_kernel, dim3(cdiv(n, perWg)), dim3(256), 0, st, src_base, dst_base, descs, n, perWg, localHeavy,
                     bit_depth, clp_min, clp_max, counters + VVC_CTR_INTS * cur, heavyList, counters + VVC_CTR_INTS * (cur ^ 1));
  if (!localHeavy)
    hipLaunchKernelGGL(if_heavy_kernel, dim3(1024), dim3(256), 0, st, src_base, dst_base, descs, bit_depth, clp_min, clp_max, counters + VVC_CTR_INTS * cur, heavyList);
  VVC_LAUNCH_CHECK_COUNTERS(st);
  return VVCGPU_OK;
}

// skip_fast: the caller knows that no descriptor is one of the fast kernel's shapes (affine sub-blocks): its launch is left out -- 65 k workgroups
// that only look at their descriptors and leave cost 80 us for the 518 k sub-blocks of a 4K picture
__attribute__((visibility("hidden"))) int vvcgpu_mc_batch_impl(const vvc_pel* ref0_base, const vvc_pel* ref1_base, vvc_pel* dst_base,
                         const vvcgpu_mc_desc* descs, int n, int bit_depth, int clp_min, int clp_max, void* stream, bool skip_fast, bool sub44, bool serve_in_kernel = false);

int vvcgpu_mc_batch(const vvc_pel* ref0_base, const vvc_pel* ref1_base, vvc_pel* dst_base,
                    const vvcgpu_mc_desc* descs, int n, int bit_depth, int clp_min, int clp_max, void* stream)
{
  return vvcgpu_mc_batch_impl(ref0_base, ref1_base, dst_base, descs, n, bit_depth, clp_min, clp_max, stream, false, false);
}
// a picture's PU list as an encoder builds it for the common partition (16x16 luma / 8x8 chroma PUs): ONE launch -- what the matrix-core kernel cannot
// take (other shapes, phases, strides, samples outside the bit depth) is served by the wave that found it behind its walk, through the generic body
int vvcgpu_mc_picture_batch(const vvc_pel* ref0_base, const vvc_pel* ref1_base, vvc_pel* dst_base,
                            const vvcgpu_mc_desc* descs, int n, int bit_depth, int clp_min, int clp_max, void* stream)
{
  return vvcgpu_mc_batch_impl(ref0_base, ref1_base, dst_base, descs, n, bit_depth, clp_min, clp_max, stream, false, false, true);
}

int vvcgpu_mc_batch_impl(const vvc_pel* ref0_base, const vvc_pel* ref1_base, vvc_pel* dst_base,
                         const vvcgpu_mc_desc* descs, int n, int bit_depth, int clp_min, int clp_max, void* stream, bool skip_fast, bool sub44, bool serve_in_kernel)
{
  VVC_CHECK_ARG(n >= 0, "mc_batch: n %d", n);
  if (n == 0) return VVCGPU_OK;
  VVC_CHECK_ARG(ref0_base && dst_base && descs, "mc_batch: null pointer");
  VVC_CHECK_ARG(((uintptr_t)descs & 15) == 0, "mc_batch: descriptor array must be 16-byte aligned");
  if (bit_depth > 10 || bit_depth < 8) { vvcgpu_set_error("mc_batch: bit depth %d outside 8..10", bit_depth); return VVCGPU_E_UNSUPPORTED; }
  hipStream_t st = (hipStream_t)stream;
  const int mfmaOff = vvcgpu_no_mfma();                                   // VVCGPU_NO_MFMA (common.h): every PU through the generic kernel (its packed vector-pipe form takes the fast shapes)
  const int* flags = nullptr;
  int* genCount = nullptr;
  if (!skip_fast && !mfmaOff)
  {
    const _Float16* image = mm_image(bit_depth);
    if (!image) return VVCGPU_E_DEVICE;
    int* fl = static_cast<int*>(vvcgpu_scratch_region(st, VVC_SCRATCH_HELPER, (size_t)n * sizeof(int)));
    if (!fl) return VVCGPU_E_DEVICE;
    const int wgL = cdiv(n, 4) < 256 * 4 ? cdiv(n, 4) : 256 * 4;   // persistent: four workgroups per CU
    unsigned long long* diag = nullptr;
    const bool wantDiag = getenv("VVCGPU_MC_DIAG") != nullptr;           // measurement aid: step stamps of one luma wave
    if (wantDiag) { VVC_HIP(hipMalloc(&diag, 64 * sizeof(unsigned long long))); VVC_HIP(hipMemsetAsync(diag, 0, 64 * sizeof(unsigned long long), st)); }
    int cur = 0;
    int* counters = vvcgpu_counters(st, &cur);
    if (!counters) return VVCGPU_E_DEVICE;
    genCount = counters + VVC_CTR_INTS * cur;
    hipLaunchKernelGGL(mc_mfma_kernel<0>, dim3(wgL < 2 ? 2 : (wgL & ~1)), dim3(256), 0, st, ref0_base, ref1_base ? ref1_base : ref0_base, dst_base, descs, n, bit_depth, clp_min, clp_max, image, fl, diag,
                       genCount, counters + VVC_CTR_INTS * (cur ^ 1), serve_in_kernel ? 1 : 0);
    if (wantDiag)
    {
      unsigned long long h[64];
      VVC_HIP(hipStreamSynchronize(st));
      VVC_HIP(hipMemcpy(h, diag, sizeof h, hipMemcpyDeviceToHost));
      (void)hipFree(diag);
      fprintf(stderr, "[vvcgpu mc diag] luma steps of one wave, cycles (wait + operands / request next / products + store | step):");
      for (int k = 0; k < 12; k++) if (h[4 * k + 3]) fprintf(stderr, " %llu/%llu/%llu|%llu", h[4 * k + 1] - h[4 * k], h[4 * k + 2] - h[4 * k + 1], h[4 * k + 3] - h[4 * k + 2], k ? h[4 * k] - h[4 * k - 4] : 0ull);
      fprintf(stderr, "\n");
    }
    flags = fl;
  }
  if (flags && serve_in_kernel)                                          // the matrix-core kernel has served what it could not take itself
  {
    VVC_LAUNCH_CHECK_COUNTERS(st);
    VVC_LAUNCH_CHECK();
    return VVCGPU_OK;
  }
  const int chunk = n >= 64 * 8192 ? 64 : (n + 8191) / 8192;            // ~8192 waves: 32 per CU
  // behind the matrix-core kernel the generic one usually finds nothing (or a few PUs) to do: a grid of 1024 waves that walk the list leaves sooner than 8192
  const int gridMax = flags ? 1024 : 8192;
  if (sub44)
    hipLaunchKernelGGL((mc_batch_kernel<false, true>), dim3(cdiv(n, chunk) < gridMax ? cdiv(n, chunk) : gridMax), dim3(64), 0, st, ref0_base,
                       ref1_base ? ref1_base : ref0_base, dst_base, descs, bit_depth, clp_min, clp_max, n, 0, nullptr, nullptr, chunk, flags, mfmaOff ? 1 : 0, genCount);
  else
    hipLaunchKernelGGL((mc_batch_kernel<false, false>), dim3(cdiv(n, chunk) < gridMax ? cdiv(n, chunk) : gridMax), dim3(64), 0, st, ref0_base,
                       ref1_base ? ref1_base : ref0_base, dst_base, descs, bit_depth, clp_min, clp_max, n, 0, nullptr, nullptr, chunk, flags, mfmaOff ? 1 : 0, genCount);
  if (genCount) VVC_LAUNCH_CHECK_COUNTERS(st);
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}

int vvcgpu_mc_dist_batch(int kind, const vvc_pel* ref0_base, const vvc_pel* ref1_base, const vvc_pel* org_base, const vvcgpu_mc_desc* descs, int n,
                         int bit_depth, int clp_min, int clp_max, uint64_t* out, void* stream)
{
  VVC_CHECK_ARG(kind >= 0 && kind <= 2, "mc_dist_batch: kind %d (0 SAD, 1 Hadamard, 2 SSE)", kind);
  VVC_CHECK_ARG(n >= 0, "mc_dist_batch: n %d", n);
  if (n == 0) return VVCGPU_OK;
  VVC_CHECK_ARG(ref0_base && org_base && descs && out, "mc_dist_batch: null pointer");
  if (bit_depth > 10 || bit_depth < 8) { vvcgpu_set_error("mc_dist_batch: bit depth %d outside 8..10", bit_depth); return VVCGPU_E_UNSUPPORTED; }
  hipLaunchKernelGGL((mc_batch_kernel<true, false>), dim3(n < 4096 ? n : 4096), dim3(64), 0, (hipStream_t)stream, ref0_base, ref1_base ? ref1_base : ref0_base,
                     nullptr, descs, bit_depth, clp_min, clp_max, n, kind, org_base, reinterpret_cast<unsigned long long*>(out), 1, nullptr, 0);
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}

int vvcgpu_pelop_batch(int op, const vvc_pel* src0_base, const vvc_pel* src1_base, vvc_pel* dst_base,
                       const vvcgpu_pelop_desc* descs, int n, const vvcgpu_pelop_cfg* cfg_host, void* stream)
{
  VVC_CHECK_ARG(op >= 0 && op <= 5, "pelop_batch: op %d", op);
  VVC_CHECK_ARG(n >= 0, "pelop_batch: n %d", n);
  if (n == 0) return VVCGPU_OK;
  VVC_CHECK_ARG(src0_base && dst_base && descs && cfg_host, "pelop_batch: null pointer");
  VVC_CHECK_ARG(src1_base || op == 2 || op == 5, "pelop_batch: op %d needs src1", op);
  int perWg = n < 8192 ? 1 : 64;                          // long lists: up to 64 descriptors per workgroup, fewer when that would leave compute units without one
  while (perWg > 16 && cdiv(n, perWg) < 4096) perWg >>= 1;
  hipStream_t st = (hipStream_t)stream;
  int* heavyList = nullptr; int* counters = nullptr; int cur = 0;
  if (perWg > 1)                                          // long lists: heavy blocks go to a list and a second launch
  {
    heavyList = static_cast<int*>(vvcgpu_scratch(st, sizeof(int) * (size_t)n));
    if (!heavyList) return VVCGPU_E_DEVICE;
    counters = vvcgpu_counters(st, &cur);
    if (!counters) return VVCGPU_E_DEVICE;
  }
  hipLaunchKernelGGL(pelop_batch_kernel, dim3(cdiv(n, perWg), n < 2048 ? 8 : 1), dim3(256), 0, st, op, src0_base, src1_base,
                     dst_base, descs, n, *cfg_host, perWg, counters ? counters + VVC_CTR_INTS * cur : nullptr, heavyList, counters ? counters + VVC_CTR_INTS * (cur ^ 1) : nullptr);
  if (perWg > 1)
  {
    hipLaunchKernelGGL(pelop_heavy_kernel, dim3(1024), dim3(256), 0, st, op, src0_base, src1_base, dst_base, descs, *cfg_host, counters + VVC_CTR_INTS * cur, heavyList);
    VVC_LAUNCH_CHECK_COUNTERS(st);
    return VVCGPU_OK;
  }
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}

}  // extern "C"
