// intra.hip -- intra sample predictors (next row N4) for gfx950: planar, DC, 65 angular modes with wide-angle mapping, the
// simplified PDPC, and the [1 2 1] reference sample filter.
//
// Reference behaviour reproduced (bit-exact):
//   IntraPrediction::predIntraAng (mode switch + PDPC, JVET_K0063)   CommonLib/IntraPrediction.cpp:251-347
//   xGetPredValDc :173-211, getWideAngle / setReferenceArrayLengths :213-249, xPredIntraPlanar :424-477,
//   xPredIntraAng :540-773 (linear interpolation iff deltaFract != 0, angular PDPC), xFilterReferenceSamples :1071-1104
//
// Design: the reference walks rows with running sums and early `break`s; every prediction sample is in fact a closed form of
// (x, y) and a handful of reference samples, so one wavefront takes one PU, stages the <= 257 reference samples (optionally
// filtered) and the projected main reference in LDS, and every lane computes samples independently, iterating in destination
// order so that stores coalesce for horizontal modes too.  Four PUs per workgroup, no workgroup barrier.
#include "common.h"
#include "dist_dev.h"

namespace {

__constant__ short kAng[27]    = { 0, 1, 2, 3, 5, 7, 9, 11, 13, 15, 17, 19, 21, 23, 26, 29, 32, 35, 39, 45, 49, 54, 60, 68, 79, 93, 114 };
__constant__ short kInvAng[27] = { 0, 8192, 4096, 2731, 1638, 1170, 910, 745, 630, 546, 482, 431, 390, 356, 315, 282, 256, 234, 210, 182, 167, 152, 137, 120,
                                   104, 88, 72 };
enum { PLANAR = 0, DC = 1, HOR = 18, DIA = 34, VER = 50, VDIA = 66 };

constexpr int REF_MAX = 160;            // >= longest side reference + 2 (2 * 64 + 22 + 1)
constexpr int NEG_MAX = 64;             // projected samples left of the main reference

__device__ __forceinline__ int ilog2(int v) { return 31 - __clz(v); }
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

// One prediction block by one wave: predIntraAng into `dst` (row pitch ds; global memory or the wave's LDS tile).  top / left / tmp / mainBuf:
// the wave's LDS work arrays (REF_MAX, REF_MAX, 2 REF_MAX, NEG_MAX + REF_MAX shorts).
// Rows [y0, y1) only (default: the whole block); dst addresses row y0.
__device__ __forceinline__ void intra_pred_block(const vvcgpu_intra_desc& d, const Pel* __restrict__ refsBase, Pel* dst, int ds, int clpMin, int clpMax, int lane,
                                                 short* top, short* left, short* tmpBuf, short* mainBuf, int y0 = 0, int y1 = 1 << 30)
{
  const int w = d.w, h = d.h, mode = d.mode;
  const int log2W = ilog2(w), log2H = ilog2(h);
  int T = w << 1, L = h << 1;                               // setReferenceArrayLengths :233-249
  {
    const int ratio = min(2, abs(log2W - log2H));
    if (w > h) L += (w >> ratio) - h + ((w + 31) >> 5);
    else if (h > w) T += (h >> ratio) - w + ((h + 31) >> 5);
  }
  // top[0] = top-left, top[1 + x]; left[0] = top-left, left[1 + y]
  const Pel* refs = refsBase + d.ref_off;
  if (!d.filter_refs)
  {
    for (int i = lane; i <= T; i += 64) top[i] = refs[i];
    for (int i = lane; i <= L; i += 64) left[i] = i == 0 ? refs[0] : refs[T + i];
  }
  else
  {
    // regular reference sample filter along the chain  left[L] .. left[1], top-left, top[1] .. top[T]; the two ends stay
    short* c = tmpBuf;                                      // chain position L + i for top[i], L - i for left[i]
    for (int i = lane; i <= T + L; i += 64) c[i] = i <= L ? (i == L ? refs[0] : refs[T + (L - i)]) : refs[i - L];
    wave_sync();
    for (int i = lane; i <= T + L; i += 64)
    {
      const int v = (i == 0 || i == T + L) ? c[i] : (c[i - 1] + 2 * c[i] + c[i + 1] + 2) >> 2;
      if (i <= L) left[L - i] = (short)v;
      if (i >= L) top[i - L] = (short)v;
    }
  }
  wave_sync();

  const int count = w * h;
  const int scale = (log2W - 2 + log2H - 2 + 2) >> 2;
  const int topLeft = top[0];

  if (mode == PLANAR || mode == DC)
  {
    int dc = 0;
    if (mode == DC)                                         // :173-211
    {
      int sum = 0;
      if (w >= h) for (int i = lane; i < w; i += 64) sum += top[1 + i];
      if (w <= h) for (int i = lane; i < h; i += 64) sum += left[1 + i];
#pragma unroll
      for (int m = 1; m < 64; m <<= 1) sum += __shfl_xor(sum, m);
      const int denom = (w == h) ? (w << 1) : max(w, h);
      dc = (short)((sum + (denom >> 1)) >> ilog2(denom));
    }
    const int bottomLeft = left[h + 1], topRight = top[w + 1];
    for (int i = y0 * w + lane; i < min(y1, h) * w; i += 64)
    {
      const int y = i >> log2W, x = i & (w - 1);
      const int l = left[y + 1], t = top[x + 1];
      const int wT = 32 >> min(31, (y << 1) >> scale), wL = 32 >> min(31, (x << 1) >> scale);
      int v;
      if (mode == PLANAR)                                   // :424-477 + PDPC :293-307
      {
        const int horPred = (l << log2W) + (x + 1) * (topRight - l), vertPred = (t << log2H) + (y + 1) * (bottomLeft - t);
        const int p = (short)(((horPred << log2H) + (vertPred << log2W) + count) >> (1 + log2W + log2H));
        v = (wL * l + wT * t + (64 - wL - wT) * p + 32) >> 6;
      }
      else                                                  // PDPC :308-323
      {
        const int wTL = (wL >> 4) + (wT >> 4);
        v = (wL * l + wT * t - wTL * topLeft + (64 - wL - wT + wTL) * dc + 32) >> 6;
      }
      dst[(ptrdiff_t)(y - y0) * ds + x] = (Pel)min(max(v, clpMin), clpMax);
    }
    return;
  }

  // angular :545-773
  int predMode = mode;                                      // getWideAngle :213-231
  {
    const int modeShift = (min(2, abs(log2W - log2H)) << 2) + 2;
    if (w > h && predMode < 2 + modeShift) predMode += VDIA - 1;
    else if (h > w && predMode > VDIA - modeShift) predMode -= VDIA - 1;
  }
  const bool isVer = predMode >= DIA;
  const int angMode = isVer ? predMode - VER : -(predMode - HOR);
  const int absAngMode = abs(angMode);
  const int invAngle = kInvAng[absAngMode], angle = (angMode < 0 ? -1 : 1) * (int)kAng[absAngMode];
  const short* mainR = isVer ? top : left;                  // index 0 = top-left
  const short* sideR = isVer ? left : top;
  const int H = isVer ? h : w, log2Wm = isVer ? log2W : log2H;   // block in the orientation of the main reference
  const int sideLen = isVer ? L : T;
  short* mainX = mainBuf + NEG_MAX;                         // main reference with the projected extension to the left (:588-609)
  if (angle < 0)
  {
    const int mainLen = (isVer ? w : h) + 1;
    for (int i = lane; i <= mainLen; i += 64) mainX[i] = mainR[i];
    const int lowest = (H * angle) >> 5;                    // k runs -1 ... > lowest
    for (int k = -1 - lane; k > lowest; k -= 64) mainX[k] = sideR[(128 + (-k) * invAngle) >> 8];
    wave_sync();
    mainR = mainX;
  }
  const bool pdpcCorner = predMode == 2 || predMode == VDIA;
  const bool pdpcNear = !pdpcCorner && ((predMode >= VDIA - 8) || (predMode <= 2 + 8));
  const bool pdpcHV = angle == 0;                           // HOR / VER: PDPC of predIntraAng :324-346
  for (int i = y0 * w + lane; i < min(y1, h) * w; i += 64)
  {
    const int dy = i >> log2W, dx = i & (w - 1);
    const int x = isVer ? dx : dy, y = isVer ? dy : dx;     // coordinates in the orientation of the main reference
    const int deltaPos = (y + 1) * angle, deltaInt = deltaPos >> 5, deltaFract = deltaPos & 31;
    int v;
    if (deltaFract) v = (short)(((32 - deltaFract) * mainR[x + deltaInt + 1] + deltaFract * mainR[x + deltaInt + 2] + 16) >> 5);
    else v = mainR[x + deltaInt + 1];
    if (pdpcHV)
    {
      // in destination coordinates: HOR uses the row above, VER the column to the left
      if (mode == HOR) { const int wT = 32 >> min(31, (dy << 1) >> scale); v = (wT * top[dx + 1] - wT * topLeft + 64 * v + 32) >> 6; }
      else             { const int wL = 32 >> min(31, (dx << 1) >> scale); v = (wL * left[dy + 1] - wL * topLeft + 64 * v + 32) >> 6; }
      v = min(max(v, clpMin), clpMax);
    }
    else if (pdpcCorner)                                    // :697-711
    {
      const int wT = 16 >> min(31, (y << 1) >> scale), wL = 16 >> min(31, (x << 1) >> scale);
      if (wT + wL != 0)
      {
        const int c = x + y + 1;
        const int l = wL != 0 ? sideR[c + 1] : 0, t = wT != 0 ? mainR[c + 1] : 0;
        v = min(max((wL * l + wT * t + (64 - wL - wT) * v + 32) >> 6, clpMin), clpMax);
      }
    }
    else if (pdpcNear)                                      // :717-743
    {
      const int deltaPos0 = (2 + (x + 1) * invAngle) >> 2, deltaFrac0 = deltaPos0 & 63, deltay = y + (deltaPos0 >> 6) + 1;
      const int wL = 32 >> min(31, (x << 1) >> scale);
      if (deltay <= sideLen - 1 && wL != 0)
      {
        const int l = (short)(((64 - deltaFrac0) * sideR[deltay] + deltaFrac0 * sideR[deltay + 1] + 32) >> 6);
        v = min(max((wL * l + (64 - wL) * v + 32) >> 6, clpMin), clpMax);
      }
    }
    (void)log2Wm;
    dst[(ptrdiff_t)(dy - y0) * ds + dx] = (Pel)v;
  }
}

__global__ __launch_bounds__(256) void intra_pred_kernel(const Pel* __restrict__ refsBase, Pel* __restrict__ dstBase,
                                                         const vvcgpu_intra_desc* __restrict__ descs, int n, int clpMin, int clpMax)
{
  __shared__ short topS[4][REF_MAX], leftS[4][REF_MAX], tmpS[4][2 * REF_MAX], mainS[4][NEG_MAX + REF_MAX];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x * 4 + wave;
  if (b >= n) return;                                       // no workgroup barrier below
  const vvcgpu_intra_desc d = descs[b];
  intra_pred_block(d, refsBase, dstBase + d.dst_off, d.dst_stride, clpMin, clpMax, lane, topS[wave], leftS[wave], tmpS[wave], mainS[wave]);
}

// ---- intra mode pre-selection (IntraSearch::estIntraPredLumaQT, EncoderLib/IntraSearch.cpp:397-480): predIntraAng of one candidate mode into
// the wave's LDS tile, then the Hadamard distortion against the original (distParam.distFunc = xGetHADs) -- the prediction never reaches HBM.
// Measured at 4K (31 k blocks of 16 x 16 x 67 modes, tools/intra_search_time.py): 2.55 ms against 1.17 + 1.13 ms for prediction and distortion as two
// launches -- the fused form saves 2 GB of prediction traffic but its waves (prediction AND Hadamard code: 100+ VGPRs, a 32 KB tile per
// workgroup) are latency-bound at a quarter of the separate kernels' occupancy.  Variants tried: a 1024-sample band tile with the Hadamard code
// inlined (187 VGPRs, 5.1 ms), the same with the distortion as a real call and a 128-VGPR cap (3.0 ms), 80-VGPR cap (spills, 9.8 ms).
__global__ __launch_bounds__(256) void intra_satd_kernel(const Pel* __restrict__ refsBase, const Pel* __restrict__ orgBase,
                                                         const vvcgpu_intra_satd_desc* __restrict__ descs, int n, int clpMin, int clpMax,
                                                         unsigned long long* __restrict__ out)
{
  __shared__ short topS[4][REF_MAX], leftS[4][REF_MAX], tmpS[4][2 * REF_MAX], mainS[4][NEG_MAX + REF_MAX];
  __shared__ __align__(16) short predS[4][64 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x * 4 + wave;
  if (b >= n) return;                                       // no workgroup barrier below
  const vvcgpu_intra_satd_desc s = descs[b];
  if (s.w < 1 || s.h < 1 || s.w > 64 || s.h > 64) { if (lane == 0) out[b] = ~0ull; return; }   // outside the 64 x 64 LDS tile (wave-uniform): sentinel, no overrun
  vvcgpu_intra_desc d;
  d.ref_off = s.ref_off; d.dst_off = 0; d.dst_stride = s.w; d.w = s.w; d.h = s.h; d.mode = s.mode; d.filter_refs = s.filter_refs;
  intra_pred_block(d, refsBase, predS[wave], s.w, clpMin, clpMax, lane, topS[wave], leftS[wave], tmpS[wave], mainS[wave]);
  wave_sync();
  typedef const __attribute__((address_space(3))) short* LdsPel;
  const unsigned long long res = satd_block<64, LdsPel>(orgBase + s.org_off, s.org_stride, (LdsPel)predS[wave], s.w, s.w, s.h, lane);
  if (lane == 0) out[b] = res;
}


// ---- reference sample gathering: xFillReferenceSamples :807-1004 ---------------------------------------------------------------
// One wavefront per block.  The reference pads through a line buffer (an unavailable unit repeats the last sample of the unit
// before it, a leading unavailable run repeats the first sample of the first available unit); in closed form every sample of an
// unavailable unit is one fixed sample of the nearest available unit, so each lane resolves its output samples independently.
__global__ __launch_bounds__(256) void intra_fill_refs_kernel(const Pel* __restrict__ recBase, const unsigned char* __restrict__ flagsBase,
                                                              Pel* __restrict__ refsBase, const vvcgpu_intra_fill_desc* __restrict__ descs, int n,
                                                              int bitDepth)
{
  __shared__ short srcUnit[4][192];                      // per unit: the unit that provides its samples (itself when available), or -1
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x * 4 + wave;
  if (b >= n) return;
  const vvcgpu_intra_fill_desc d = descs[b];
  const int w = d.w, h = d.h, uw = d.unit_w, uh = d.unit_h, rs = d.rec_stride;
  int T = w << 1, L = h << 1;
  {
    const int lw = ilog2(w), lh = ilog2(h), ratio = min(2, abs(lw - lh));
    if (w > h) L += (w >> ratio) - h + ((w + 31) >> 5);
    else if (h > w) T += (h >> ratio) - w + ((h + 31) >> 5);
  }
  const int aboveUnits = (T + uw - 1) / uw, leftUnits = (L + uh - 1) / uh, total = aboveUnits + leftUnits + 1;
  const unsigned char* flags = flagsBase + d.flags_off;
  short* su = srcUnit[wave];
  int firstLocal = 0x7fff;
  for (int u = lane; u < total; u += 64)
  {
    int p = u;
    while (p >= 0 && !flags[p]) p--;
    su[u] = (short)p;                                     // nearest available unit at or before u, -1 if none
    if (flags[u]) firstLocal = min(firstLocal, u);
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) firstLocal = min(firstLocal, __shfl_xor(firstLocal, m));
  wave_sync();
  const int first = firstLocal;                           // 0x7fff: nothing available
  const Pel* rec = recBase + d.rec_off;
  Pel* refs = refsBase + d.ref_off;
  const int dc = 1 << (bitDepth - 1);
  for (int i = lane; i <= T + L; i += 64)
  {
    int u, o;
    if (i <= T) { const int q = uw - 1 + i; u = leftUnits + q / uw; o = q % uw; }
    else { const int k = leftUnits * uh - (i - T); u = k / uh; o = k % uh; }
    int v = dc;
    if (first != 0x7fff)
    {
      int s = su[u];
      if (s != u) { if (s >= 0) o = (s < leftUnits ? uh : uw) - 1; else { s = first; o = 0; } }
      if (s < leftUnits) v = rec[(ptrdiff_t)((leftUnits - s) * uh - 1 - o) * rs - 1];
      else if (s == leftUnits) v = rec[-rs - 1];
      else v = rec[-rs + (s - leftUnits - 1) * uw + o];
    }
    refs[i] = (Pel)v;
  }
}

// ---- CCLM (JVET_K0190): xGetLumaRecPixels :1283-1581 + xGetLMParameters :1597-1857 + predIntraChromaLM :390-403 -------------
// One wavefront per chroma block.  The down-sampled luma is never stored: neighbours are produced for the parameter sums and the
// inner samples on the fly for the final linear map (6 luma reads per chroma sample, all L1/L2 hits of one small region).
__device__ __forceinline__ int floor_log2(unsigned x) { return 31 - __clz((int)x); }     // x > 0

__global__ __launch_bounds__(256) void cclm_pred_kernel(const Pel* __restrict__ lumaBase, const Pel* __restrict__ nbBase, Pel* __restrict__ dstBase,
                                                        const vvcgpu_cclm_desc* __restrict__ descs, int n, int bdLuma, int bdChroma, int clpMin,
                                                        int clpMax)
{
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= n) return;
  const vvcgpu_cclm_desc d = descs[b];
  const int w = d.w, h = d.h, rs = d.luma_stride, rs2 = rs * 2;
  const bool aboveAvail = d.above_avail != 0, leftAvail = d.left_avail != 0;
  const Pel* luma = lumaBase + d.luma_off;
  const Pel* nbAbove = nbBase + d.nb_off;
  const Pel* nbLeft = nbAbove + w;
  auto six = [&](const Pel* p) { return (p[0] * 2 + p[-1] + p[1] + p[rs] * 2 + p[rs - 1] + p[rs + 1] + 4) >> 3; };
  auto two = [&](const Pel* p) { return (p[0] + p[rs] + 1) >> 1; };

  int a = 0, bb = 1 << (bdChroma - 1), shift = 0;
  if (aboveAvail || leftAvail)
  {
    int x = 0, y = 0, xx = 0, xy = 0, countShift = 0;
    const int minDim = (leftAvail && aboveAvail) ? min(w, h) : (leftAvail ? h : w);
    const int lgMin = floor_log2((unsigned)minDim);
    if (aboveAvail)
    {
      for (int j = lane; j < minDim; j += 64)
      {
        const int idx = (j * w) >> lgMin;
        const Pel* p = luma - rs2 + 2 * idx;
        const int s = (idx == 0 && !leftAvail) ? two(p) : six(p), c = nbAbove[idx];
        x += s; y += c; xx += s * s; xy += s * c;
      }
      countShift = lgMin;
    }
    if (leftAvail)
    {
      for (int i = lane; i < minDim; i += 64)
      {
        const int idx = (i * h) >> lgMin;
        const int s = six(luma + (ptrdiff_t)idx * rs2 - 2), c = nbLeft[idx];
        x += s; y += c; xx += s * s; xy += s * c;
      }
      countShift += aboveAvail ? 1 : lgMin;
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { x += __shfl_xor(x, m); y += __shfl_xor(y, m); xx += __shfl_xor(xx, m); xy += __shfl_xor(xy, m); }
    const int tempShift = bdChroma + countShift - 15;
    if (tempShift > 0)
    {
      const int r = 1 << (tempShift - 1);
      x = (x + r) >> tempShift; y = (y + r) >> tempShift; xx = (xx + r) >> tempShift; xy = (xy + r) >> tempShift;
      countShift -= tempShift;
    }
    const int avgX = x >> countShift, avgY = y >> countShift;
    const int rErrX = x & ((1 << countShift) - 1), rErrY = y & ((1 << countShift) - 1);
    const int iB = 7;
    shift = 13 - iB;
    if (countShift == 0) { a = 0; bb = 1 << (bdChroma - 1); shift = 0; }
    else
    {
      const int a1 = xy - ((avgX * avgY) << countShift) - avgX * rErrY - avgY * rErrX;
      const int a2 = xx - ((avgX * avgX) << countShift) - 2 * avgX * rErrX;
      int sA1 = a1 == 0 ? 0 : floor_log2((unsigned)abs(a1)) - (bdChroma - 2);
      int sA2 = a2 == 0 ? 0 : floor_log2((unsigned)abs(a2)) - 5;
      sA1 = max(sA1, 0); sA2 = max(sA2, 0);
      const int sA = sA2 + (bdChroma + 4) - shift - sA1;
      const int a2s = a2 >> sA2, a1s = a1 >> sA1;
      if (a2s >= 32) a = (int)((unsigned)a1s * (unsigned)(((1 << (bdLuma + 4)) + a2s / 2) / a2s));     // m_auShiftLM[a2s - 32]
      else a = 0;
      if (sA < 0) a = (int)((unsigned)a << -sA); else a = a >> sA;
      a = min(max(a, -(1 << (15 - iB))), (1 << (15 - iB)) - 1);
      a = a * (1 << iB);
      int nn = 0;
      if (a != 0) nn = (short)(floor_log2((unsigned)(abs(a) + ((a < 0 ? -1 : 1) - 1) / 2)) - 5);
      shift = (shift + iB) - nn;
      a = a >> nn;
      bb = avgY - ((a * avgX) >> shift);
    }
  }
  Pel* dst = dstBase + d.dst_off;
  const int lgW = floor_log2((unsigned)w);
  for (int i = lane; i < w * h; i += 64)
  {
    const int yy = i >> lgW, xq = i & (w - 1);
    const Pel* p = luma + (ptrdiff_t)yy * rs2 + 2 * xq;
    const int s = (short)((xq == 0 && !leftAvail) ? two(p) : six(p));
    dst[(ptrdiff_t)yy * d.dst_stride + xq] = (Pel)min(max(((a * s) >> shift) + bb, clpMin), clpMax);
  }
}

}  // namespace

extern "C" int vvcgpu_intra_ref_lengths(int w, int h, int* top_len, int* left_len)
{
  VVC_CHECK_ARG(top_len && left_len && w >= 4 && h >= 4 && w <= 64 && h <= 64 && !(w & (w - 1)) && !(h & (h - 1)), "intra_ref_lengths: %dx%d", w, h);
  int lw = 0, lh = 0; while ((1 << (lw + 1)) <= w) lw++; while ((1 << (lh + 1)) <= h) lh++;
  const int ratio = abs(lw - lh) < 2 ? abs(lw - lh) : 2;
  *left_len = h << 1; *top_len = w << 1;
  if (w > h) *left_len += (w >> ratio) - h + ((w + 31) >> 5);
  else if (h > w) *top_len += (h >> ratio) - w + ((h + 31) >> 5);
  return VVCGPU_OK;
}

extern "C" int vvcgpu_intra_pred_batch(const vvc_pel* refs_base, vvc_pel* dst_base, const vvcgpu_intra_desc* descs, int n, int clp_min,
                                       int clp_max, void* stream)
{
  VVC_CHECK_ARG(n >= 0, "intra_pred_batch: n %d", n);
  if (n == 0) return VVCGPU_OK;
  VVC_CHECK_ARG(refs_base && dst_base && descs, "intra_pred_batch: null pointer");
  VVC_CHECK_ARG(clp_min <= clp_max && clp_min >= -32768 && clp_max <= 32767, "intra_pred_batch: clip range %d..%d", clp_min, clp_max);
  hipLaunchKernelGGL(intra_pred_kernel, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, refs_base, dst_base, descs, n, clp_min, clp_max);
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}

extern "C" int vvcgpu_intra_satd_batch(const vvc_pel* refs_base, const vvc_pel* org_base, const vvcgpu_intra_satd_desc* descs, int n, int clp_min,
                                       int clp_max, uint64_t* out, void* stream)
{
  VVC_CHECK_ARG(n >= 0, "intra_satd_batch: n %d", n);
  if (n == 0) return VVCGPU_OK;
  VVC_CHECK_ARG(refs_base && org_base && descs && out, "intra_satd_batch: null pointer");
  VVC_CHECK_ARG(clp_min <= clp_max && clp_min >= -32768 && clp_max <= 32767, "intra_satd_batch: clip range %d..%d", clp_min, clp_max);
  hipLaunchKernelGGL(intra_satd_kernel, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, refs_base, org_base, descs, n, clp_min, clp_max,
                     reinterpret_cast<unsigned long long*>(out));
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}

extern "C" int vvcgpu_cclm_pred_batch(const vvc_pel* luma_base, const vvc_pel* nb_base, vvc_pel* dst_base, const vvcgpu_cclm_desc* descs, int n,
                                      int bit_depth_luma, int bit_depth_chroma, int clp_min, int clp_max, void* stream)
{
  VVC_CHECK_ARG(n >= 0, "cclm_pred_batch: n %d", n);
  if (n == 0) return VVCGPU_OK;
  VVC_CHECK_ARG(luma_base && nb_base && dst_base && descs, "cclm_pred_batch: null pointer");
  VVC_CHECK_ARG(bit_depth_luma >= 8 && bit_depth_luma <= 12 && bit_depth_chroma >= 8 && bit_depth_chroma <= 12, "cclm_pred_batch: bit depths %d %d",
                bit_depth_luma, bit_depth_chroma);
  VVC_CHECK_ARG(clp_min <= clp_max, "cclm_pred_batch: clip range");
  hipLaunchKernelGGL(cclm_pred_kernel, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, luma_base, nb_base, dst_base, descs, n, bit_depth_luma,
                     bit_depth_chroma, clp_min, clp_max);
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}

extern "C" int vvcgpu_intra_fill_refs_batch(const vvc_pel* rec_base, const uint8_t* flags_base, vvc_pel* refs_base, const vvcgpu_intra_fill_desc* descs,
                                            int n, int bit_depth, void* stream)
{
  VVC_CHECK_ARG(n >= 0, "intra_fill_refs_batch: n %d", n);
  if (n == 0) return VVCGPU_OK;
  VVC_CHECK_ARG(rec_base && flags_base && refs_base && descs, "intra_fill_refs_batch: null pointer");
  VVC_CHECK_ARG(bit_depth >= 8 && bit_depth <= 12, "intra_fill_refs_batch: bit depth %d", bit_depth);
  hipLaunchKernelGGL(intra_fill_refs_kernel, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, rec_base, flags_base, refs_base, descs, n, bit_depth);
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}
