// alf.hip -- ALF classification (A1) and diamond filtering (A2) for gfx950.
//
// Reference behaviour reproduced (bit-exact):
//   AdaptiveLoopFilter::deriveClassificationBlk  CommonLib/AdaptiveLoopFilter.cpp:292-463
//   AdaptiveLoopFilter::filterBlk<5|7>           CommonLib/AdaptiveLoopFilter.cpp:465-650
//   ALFProcess border handling (copy + extendBorderPel(3))  :87-92  -> done here by clamping the
//   tile loader's coordinates, so no temp copy and no border pass touch HBM.
//
// Design (HBM-bound stencils, guide App. B "element-wise"/G13):
//   * one workgroup = one 64x64 (classify) / 64x32 (filter) luma tile staged ONCE in LDS with its halo,
//     16-byte global loads where the row segment is inside the picture;
//   * classification: 17x17 "quad" sums Q (4x4 pixel Laplacian sums on the grid shifted by -2) are
//     computed once and each 4x4 block adds its four quads -- every pixel Laplacian is evaluated once
//     instead of four times;
//   * filtering: one thread per 4x4 block (the granularity at which coefficients change), 10 input rows
//     streamed through registers into 16 accumulators; the class's coefficients are permuted once per block.
#include "common.h"
#include <cstddef>

namespace {

constexpr int CT = 64;            // classify tile (pixels)
constexpr int CP = 72;            // LDS pitch (samples): x origin = tile_x - 4, 72 = 64 + 8
constexpr int CR = CT + 6;        // rows: y origin = tile_y - 3
constexpr int QN = CT / 4 + 1;    // 17 quads per dimension

// Loads rows [y0, y0+rows) x cols [x0, x0+pitch) of the plane into LDS (int16), replicating the
// picture border (== extendBorderPel).  x0 is a multiple of 4 samples, pitch a multiple of 4.
template <int PITCH>
__device__ __forceinline__ void load_tile_clamped(short* __restrict__ lds, const Pel* __restrict__ src,
                                                  int stride, int w, int h, int x0, int y0, int rows,
                                                  int tid, int nthreads)
{
  constexpr int VPR = PITCH / 4;                      // 8-byte vectors per row
  const int nvec = rows * VPR;
  const bool vec_ok = ((stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(src) & 7) == 0);
  for (int v = tid; v < nvec; v += nthreads)
  {
    const int r = v / VPR, c = (v - r * VPR) * 4;
    int y = y0 + r;
    y = y < 0 ? 0 : (y >= h ? h - 1 : y);
    const int x = x0 + c;
    const Pel* row = src + (size_t)y * stride;
    pel4 val;
    if (vec_ok && x >= 0 && x + 3 < w)
      val = *reinterpret_cast<const pel4*>(row + x);
    else
    {
#pragma unroll
      for (int k = 0; k < 4; k++)
      {
        int xx = x + k;
        xx = xx < 0 ? 0 : (xx >= w ? w - 1 : xx);
        val[k] = row[xx];
      }
    }
    *reinterpret_cast<pel4*>(lds + r * PITCH + c) = val;
  }
}

__global__ __launch_bounds__(320) void alf_classify_kernel(const Pel* __restrict__ src, int stride, int w, int h,
                                                           int shift, uint16_t* __restrict__ cls, int gx, int total, int xcd)
{
  __shared__ short tile[CR * CP];
  __shared__ int quad[QN * QN * 4];
  const int tid = threadIdx.x;
  const int b = vvc_xcd_index((int)blockIdx.x, total, xcd);                 // tiles in raster order, one contiguous run per XCD
  if (b < 0) return;
  const int by = b / gx, bx = b - by * gx;
  const int tx0 = bx * CT, ty0 = by * CT;
  load_tile_clamped<CP>(tile, src, stride, w, h, tx0 - 4, ty0 - 3, CR, tid, 320);
  __syncthreads();

  if (tid < QN * QN)
  {
    const int qi = tid / QN, qj = tid - qi * QN;
    // quad region: picture rows ty0 + 4qi - 2 .. +3, cols tx0 + 4qj - 2 .. +3  ->  LDS row 4qi+1, col 4qj+2
    // Two samples per instruction: a row of the 6x6 neighbourhood (columns 4qj+1 .. 4qj+6) is four aligned dwords, its five sample pairs
    // P0..P4 = (s0,s1) .. (s4,s5) are two of the dwords and three v_alignbit; |2c - a - b| of a pair of samples is ONE v_sad_u16 of the
    // doubled centre pair against the packed sum of the two neighbour pairs (samples are non-negative, <= 15 bits: nothing wraps), which also
    // accumulates.  18 instructions per row of four samples instead of 64.
    const unsigned* base = reinterpret_cast<const unsigned*>(tile + (4 * qi) * CP + 4 * qj);
    unsigned sv = 0, sh = 0, sd0 = 0, sd1 = 0;
    unsigned A[5], B[5], C[5];                        // pairs of the rows above / at / below the centre row
    auto loadRow = [&](int r, unsigned (&P)[5])
    {
      const uint2 lo = *reinterpret_cast<const uint2*>(base + r * (CP / 2)), hi = *reinterpret_cast<const uint2*>(base + r * (CP / 2) + 2);
      P[0] = __builtin_amdgcn_alignbit(lo.y, lo.x, 16); P[1] = lo.y; P[2] = __builtin_amdgcn_alignbit(hi.x, lo.y, 16); P[3] = hi.x;
      P[4] = __builtin_amdgcn_alignbit(hi.y, hi.x, 16);
    };
    typedef unsigned short us2v __attribute__((ext_vector_type(2)));
    auto padd = [](unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, __builtin_bit_cast(us2v, a) + __builtin_bit_cast(us2v, b)); };
    loadRow(0, A); loadRow(1, B);
#pragma unroll
    for (int y = 0; y < 4; y++)
    {
      loadRow(y + 2, C);
      // centre row = B: centres (s1,s2) = B[1], (s3,s4) = B[3]
      const unsigned c01 = padd(B[1], B[1]), c23 = padd(B[3], B[3]);
      sv  = __builtin_amdgcn_sad_u16(c01, padd(A[1], C[1]), sv);   sv  = __builtin_amdgcn_sad_u16(c23, padd(A[3], C[3]), sv);
      sh  = __builtin_amdgcn_sad_u16(c01, padd(B[0], B[2]), sh);   sh  = __builtin_amdgcn_sad_u16(c23, padd(B[2], B[4]), sh);
      sd0 = __builtin_amdgcn_sad_u16(c01, padd(A[0], C[2]), sd0);  sd0 = __builtin_amdgcn_sad_u16(c23, padd(A[2], C[4]), sd0);
      sd1 = __builtin_amdgcn_sad_u16(c01, padd(C[0], A[2]), sd1);  sd1 = __builtin_amdgcn_sad_u16(c23, padd(C[2], A[4]), sd1);
#pragma unroll
      for (int k = 0; k < 5; k++) { A[k] = B[k]; B[k] = C[k]; }
    }
    int* q = quad + tid * 4;
    q[0] = (int)sv; q[1] = (int)sh; q[2] = (int)sd0; q[3] = (int)sd1;
  }
  __syncthreads();

  if (tid < 256)
  {
    const int bi = tid >> 4, bj = tid & 15;
    const int by = ty0 + 4 * bi, bx = tx0 + 4 * bj;
    if (by < h && bx < w)
    {
      const int* q00 = quad + (bi * QN + bj) * 4;
      const int* q01 = q00 + 4;
      const int* q10 = q00 + QN * 4;
      const int* q11 = q10 + 4;
      const int sumV = q00[0] + q01[0] + q10[0] + q11[0];
      const int sumH = q00[1] + q01[1] + q10[1] + q11[1];
      const int sumD0 = q00[2] + q01[2] + q10[2] + q11[2];
      const int sumD1 = q00[3] + q01[3] + q10[3] + q11[3];
      // th[] of AdaptiveLoopFilter.cpp:294 packed 4 bits per entry
      const unsigned long long th = 0x4333333332222210ull;
      const int activity = (short)clip3(0, 15, ((sumV + sumH) * 32) >> shift);
      int classIdx = (int)((th >> (4 * activity)) & 15);
      int hv1, hv0, d1, d0, dirHV, dirD;
      if (sumV > sumH) { hv1 = sumV; hv0 = sumH; dirHV = 1; } else { hv1 = sumH; hv0 = sumV; dirHV = 3; }
      if (sumD0 > sumD1) { d1 = sumD0; d0 = sumD1; dirD = 0; } else { d1 = sumD1; d0 = sumD0; dirD = 2; }
      int hvd1, hvd0, mainDir, secDir;
      // products wrap modulo 2^32 exactly like the reference's int arithmetic
      if ((int)((unsigned)d1 * (unsigned)hv0) > (int)((unsigned)hv1 * (unsigned)d0))
      { hvd1 = d1; hvd0 = d0; mainDir = dirD; secDir = dirHV; }
      else
      { hvd1 = hv1; hvd0 = hv0; mainDir = dirHV; secDir = dirD; }
      int strength = 0;
      if (hvd1 > 2 * hvd0) strength = 1;
      if (hvd1 * 2 > 9 * hvd0) strength = 2;
      if (strength) classIdx += (((mainDir & 1) << 1) + strength) * 5;
      // transposeTable {0,1,0,2,2,3,1,3} packed 2 bits per entry (:447)
      const int transposeIdx = (0xDE84u >> (2 * (mainDir * 2 + (secDir >> 1)))) & 3;
      cls[(size_t)(by >> 2) * (w >> 2) + (bx >> 2)] = (uint16_t)(classIdx | (transposeIdx << 8));
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Filtering.  Tile FW x FH output pixels, halo 3 rows, 4 columns (keeps 8-byte alignment).
constexpr int FW = 64, FH = 32;
constexpr int FP = FW + 8;          // LDS pitch, x origin = tile_x - 4
constexpr int FR = FH + 6;          // rows, y origin = tile_y - 3

struct AlfCoeffs { int16_t c[25 * 13]; };   // passed by value in the kernel argument block

// coefficient index K(dy,dx) of the point-symmetric diamonds (AdaptiveLoopFilter.cpp:600-636)
template <bool IS7>
__device__ __forceinline__ constexpr int tapIndex(int dy, int dx)
{
  if (dy < 0 || (dy == 0 && dx < 0)) { dy = -dy; dx = -dx; }
  if (IS7)
  {
    if (dy == 3) return dx == 0 ? 0 : -1;
    if (dy == 2) return dx == 1 ? 1 : dx == 0 ? 2 : dx == -1 ? 3 : -1;
    if (dy == 1) return (dx >= -2 && dx <= 2) ? 6 - dx : -1;
    return dx <= 3 ? 12 - dx : -1;
  }
  else
  {
    if (dy == 2) return dx == 0 ? 0 : -1;
    if (dy == 1) return (dx >= -1 && dx <= 1) ? 2 - dx : -1;
    if (dy == 0) return dx <= 2 ? 6 - dx : -1;
    return -1;
  }
}

template <bool IS7, bool LUMA>
__device__ __forceinline__ void alf_filter_body(const int bidx, const int bidy, short* tile, short* scoef, const Pel* __restrict__ src, int sstride,
                                                         Pel* __restrict__ dst, int dstride, int w, int h,
                                                         int ctu, int wCtu, const uint16_t* __restrict__ cls,
                                                         const int16_t* __restrict__ coeffs, const uint8_t* __restrict__ ctuEnable,
                                                         int clpMin, int clpMax)
{
  const int tid = threadIdx.x;
  const int tx0 = bidx * FW, ty0 = bidy * FH;
  load_tile_clamped<FP>(tile, src, sstride, w, h, tx0 - 4, ty0 - 3, FR, tid, 128);
  // `coeffs` points INTO the kernel-argument segment (alf_kernarg): indexing the by-value argument with the thread index made every lane copy the
  // whole 650-byte struct to scratch memory first (656 bytes of private segment per lane; no measurable time at 4K, but no reason to keep it)
  for (int i = tid; i < (LUMA ? 25 * 13 : 7); i += 128) scoef[i] = coeffs[i];
  __syncthreads();

  // 16 x 8 blocks of 4x4 per tile, one per thread
  const int bj = tid & 15, bi = tid >> 4;
  const int bx = tx0 + 4 * bj, by = ty0 + 4 * bi;
  if (bx >= w || by >= h) return;
  constexpr int R = IS7 ? 3 : 2;
  constexpr int NC = IS7 ? 13 : 7;
  const bool enabled = !ctuEnable || ctuEnable[(by / ctu) * wCtu + bx / ctu];
  const short* p = tile + (4 * bi + 3 - R) * FP + 4 * bj;      // row by-R, col bx-4

  if (!enabled)
  {
    // dst receives the unfiltered samples so that dst is a complete picture (no temp copy in HBM)
#pragma unroll
    for (int y = 0; y < 4; y++)
      *reinterpret_cast<pel4*>(dst + (size_t)(by + y) * dstride + bx) =
          *reinterpret_cast<const pel4*>(p + (R + y) * FP + 4);
    return;
  }

  int f[NC];
  if (LUMA)
  {
    const uint16_t c = cls[(size_t)(by >> 2) * (w >> 2) + (bx >> 2)];
    const short* cf = scoef + (c & 0xff) * 13;
    const int t = c >> 8;
    // permutations of AdaptiveLoopFilter.cpp:545-580, packed 4 bits per entry
    if (IS7)
    {
      const unsigned long long perm = t == 0 ? 0xCBA9876543210ull : t == 1 ? 0xC62037B518A49ull
                                    : t == 2 ? 0xCBA9456781230ull : 0xC62015B734A89ull;
#pragma unroll
      for (int i = 0; i < 13; i++) f[i] = cf[(perm >> (4 * i)) & 15];
    }
    else
    {
      const unsigned perm = t == 0 ? 0x6543210u : t == 1 ? 0x6203514u : t == 2 ? 0x6541230u : 0x6201534u;
#pragma unroll
      for (int i = 0; i < 7; i++) f[i] = cf[(perm >> (4 * i)) & 15];
    }
  }
  else
  {
#pragma unroll
    for (int i = 0; i < NC; i++) f[i] = scoef[i];
  }

  int acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = 0;

  // Two taps per instruction: the taps of one diamond row (dy fixed, dx = -wid .. wid) are taken in pairs (dx, dx + 1) -- the last one with a zero
  // partner -- as packed 16-bit coefficient pairs, built once per block; a tile row arrives as six dwords = the even sample pairs E[m] = (s[2m],
  // s[2m+1]), the odd pairs O[m] = (s[2m+1], s[2m+2]) cost one v_alignbit each, and a pair of taps on a pair of samples is one v_dot2_i32_i16:
  // 16 of them per output sample of the 7x7 diamond instead of 25 multiply-adds, and no unpacking of the row into 32-bit registers.
  typedef short s2v __attribute__((ext_vector_type(2)));
  unsigned cpk[2 * R + 1][R + 1];                       // [dy + R][pair]
#pragma unroll
  for (int dy = -R; dy <= R; dy++)
  {
    const int wid = R - (dy < 0 ? -dy : dy);
#pragma unroll
    for (int q = 0; q <= R; q++)
    {
      const int dx0 = -wid + 2 * q;
      unsigned v = 0;
      if (dx0 <= wid)
      {
        const int ka = tapIndex<IS7>(dy, dx0);
        v = (unsigned)f[ka] & 0xFFFFu;
        if (dx0 + 1 <= wid) v |= (unsigned)f[tapIndex<IS7>(dy, dx0 + 1)] << 16;
      }
      cpk[dy + R][q] = v;
    }
  }
#pragma unroll
  for (int r = 0; r < 4 + 2 * R; r++)
  {
    // input row by - R + r, columns bx-4 .. bx+7
    unsigned E[6], O[5];
    const uint2 v0 = *reinterpret_cast<const uint2*>(p + r * FP), v1 = *reinterpret_cast<const uint2*>(p + r * FP + 4), v2 = *reinterpret_cast<const uint2*>(p + r * FP + 8);
    E[0] = v0.x; E[1] = v0.y; E[2] = v1.x; E[3] = v1.y; E[4] = v2.x; E[5] = v2.y;
#pragma unroll
    for (int m = 0; m < 5; m++) O[m] = __builtin_amdgcn_alignbit(E[m + 1], E[m], 16);
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
      const int dy = r - R - i;
      if (dy < -R || dy > R) continue;
      const int wid = R - (dy < 0 ? -dy : dy);
#pragma unroll
      for (int q = 0; q <= R; q++)
      {
        const int dx0 = -wid + 2 * q;
        if (dx0 > wid) continue;
        const s2v cp = __builtin_bit_cast(s2v, cpk[dy + R][q]);
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
          const int idx = 4 + j + dx0;                  // first sample of the pair
          const unsigned sp = (idx & 1) ? O[(idx - 1) >> 1] : E[idx >> 1];
          acc[i][j] = __builtin_amdgcn_sdot2(__builtin_bit_cast(s2v, sp), cp, acc[i][j], false);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; i++)
  {
    pel4 o;
#pragma unroll
    for (int j = 0; j < 4; j++) o[j] = (short)clip3(clpMin, clpMax, (acc[i][j] + 256) >> 9);
    *reinterpret_cast<pel4*>(dst + (size_t)(by + i) * dstride + bx) = o;
  }
}

// address of a by-value kernel argument inside the kernel-argument segment (byte offset as laid out by the C struct rules)
__device__ __forceinline__ const int16_t* alf_kernarg(size_t off)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return reinterpret_cast<const int16_t*>((const char*)__builtin_amdgcn_kernarg_segment_ptr() + off);
#else
  (void)off; return nullptr;
#endif
}
// the kernel's ONE argument: the coefficients are read straight from the kernel-argument segment at offsetof(AlfFilterArgs, coeffs) (indexing a
// by-value argument puts a copy of it into every lane's scratch memory) -- the offset follows the struct, not a hand-kept mirror of a parameter list
struct AlfFilterArgs { const Pel* src; int sstride; Pel* dst; int dstride, w, h, ctu, wCtu; const uint16_t* cls; AlfCoeffs coeffs; const uint8_t* ctuEnable; int clpMin, clpMax; };

template <bool IS7, bool LUMA>
__global__ __launch_bounds__(128) void alf_filter_kernel(AlfFilterArgs a)
{
  __shared__ short tile[FR * FP];
  __shared__ short scoef[25 * 13 + 3];
  alf_filter_body<IS7, LUMA>((int)blockIdx.x, (int)blockIdx.y, tile, scoef, a.src, a.sstride, a.dst, a.dstride, a.w, a.h, a.ctu, a.wCtu, a.cls,
                             alf_kernarg(offsetof(AlfFilterArgs, coeffs)), a.ctuEnable, a.clpMin, a.clpMax);
}

// luma (classifier-driven 7x7 or 5x5) and both chroma planes (5x5, one filter) of a picture in one launch
struct AlfPlane { const Pel* src; Pel* dst; const uint8_t* enable; int sstride, dstride; };
struct AlfFilter3 { AlfPlane a[3]; const uint16_t* cls; int w, h, ctu, glx, nLuma, gcx, gcy, clpMin, clpMax, total, xcd; AlfCoeffs luma; short chroma[8]; };
template <bool IS7>
__global__ __launch_bounds__(128) void alf_filter_picture_kernel(AlfFilter3 p)
{
  __shared__ short tile[FR * FP];
  __shared__ short scoef[25 * 13 + 3];
  const int b = vvc_xcd_index2((int)blockIdx.x, p.nLuma, p.total, p.xcd);
  if (b < 0) return;
  if (b < p.nLuma)
    alf_filter_body<IS7, true>(b % p.glx, b / p.glx, tile, scoef, p.a[0].src, p.a[0].sstride, p.a[0].dst, p.a[0].dstride, p.w, p.h, p.ctu, (p.w + p.ctu - 1) / p.ctu,
                               p.cls, alf_kernarg(offsetof(AlfFilter3, luma)), p.a[0].enable, p.clpMin, p.clpMax);
  else
  {
    const int c = b - p.nLuma, per = p.gcx * p.gcy, z = c / per, r = c - z * per;
    const AlfPlane& a = z ? p.a[2] : p.a[1];
    const int ctuC = p.ctu >> 1, wc = p.w >> 1;
    alf_filter_body<false, false>(r % p.gcx, r / p.gcx, tile, scoef, a.src, a.sstride, a.dst, a.dstride, wc, p.h >> 1, ctuC, (wc + ctuC - 1) / ctuC,
                                  nullptr, alf_kernarg(offsetof(AlfFilter3, chroma)), a.enable, p.clpMin, p.clpMax);
  }
}


}  // namespace

extern "C" {

int vvcgpu_alf_classify(const vvc_pel* src, int src_stride, int width, int height, int bit_depth,
                        uint16_t* cls, void* stream)
{
  VVC_CHECK_ARG(src && cls, "alf_classify: null pointer");
  VVC_CHECK_ARG(width > 0 && height > 0 && (width & 3) == 0 && (height & 3) == 0,
                "alf_classify: width/height must be positive multiples of 4 (got %dx%d)", width, height);
  VVC_CHECK_ARG(src_stride >= width, "alf_classify: stride %d < width %d", src_stride, width);
  VVC_CHECK_ARG(bit_depth >= 8 && bit_depth <= 10, "alf_classify: bit depth %d outside 8..10", bit_depth);
  const int gx = cdiv(width, CT), total = gx * cdiv(height, CT), xcd = vvc_xcd_on();
  hipLaunchKernelGGL(alf_classify_kernel, dim3(vvc_xcd_grid(total, xcd)), dim3(320), 0, (hipStream_t)stream, src, src_stride, width, height,
                     bit_depth + 4, cls, gx, total, xcd);
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}

static int alf_filter_common(bool luma, const vvc_pel* src, int src_stride, vvc_pel* dst, int dst_stride,
                             int width, int height, int ctu_size, const uint16_t* cls, int filter_type,
                             const int16_t* coeff_host, const uint8_t* ctu_enable, int clp_min, int clp_max,
                             void* stream)
{
  VVC_CHECK_ARG(src && dst && coeff_host, "alf_filter: null pointer");
  VVC_CHECK_ARG(src != dst, "alf_filter: src must not alias dst");
  VVC_CHECK_ARG(!luma || cls, "alf_filter_luma: null classifier");
  VVC_CHECK_ARG(width > 0 && height > 0 && (width & 3) == 0 && (height & 3) == 0,
                "alf_filter: width/height must be positive multiples of 4 (got %dx%d)", width, height);
  VVC_CHECK_ARG(src_stride >= width && dst_stride >= width && (dst_stride & 3) == 0,
                "alf_filter: bad strides %d/%d (dst stride must be a multiple of 4)", src_stride, dst_stride);
  VVC_CHECK_ARG(((uintptr_t)dst & 7) == 0, "alf_filter: dst must be 8-byte aligned");
  VVC_CHECK_ARG(ctu_size >= 4 && (ctu_size & 3) == 0, "alf_filter: bad ctu size %d", ctu_size);
  VVC_CHECK_ARG(filter_type == 0 || filter_type == 1, "alf_filter: filter_type %d", filter_type);
  AlfCoeffs cf;
  memset(&cf, 0, sizeof cf);
  memcpy(cf.c, coeff_host, sizeof(int16_t) * (luma ? 25 * 13 : 7));
  dim3 grid(cdiv(width, FW), cdiv(height, FH));
  const int wCtu = cdiv(width, ctu_size);
  hipStream_t st = (hipStream_t)stream;
  AlfFilterArgs ka{ src, src_stride, dst, dst_stride, width, height, ctu_size, wCtu, cls, cf, ctu_enable, clp_min, clp_max };
  if (luma && filter_type == 1)
    hipLaunchKernelGGL((alf_filter_kernel<true, true>), grid, dim3(128), 0, st, ka);
  else if (luma)
    hipLaunchKernelGGL((alf_filter_kernel<false, true>), grid, dim3(128), 0, st, ka);
  else
    hipLaunchKernelGGL((alf_filter_kernel<false, false>), grid, dim3(128), 0, st, ka);
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}

int vvcgpu_alf_filter_luma(const vvc_pel* src, int src_stride, vvc_pel* dst, int dst_stride,
                           int width, int height, int ctu_size, const uint16_t* cls,
                           int filter_type, const int16_t* coeff_host, const uint8_t* ctu_enable,
                           int clp_min, int clp_max, void* stream)
{
  return alf_filter_common(true, src, src_stride, dst, dst_stride, width, height, ctu_size, cls, filter_type,
                           coeff_host, ctu_enable, clp_min, clp_max, stream);
}

int vvcgpu_alf_filter_chroma(const vvc_pel* src, int src_stride, vvc_pel* dst, int dst_stride,
                             int width, int height, int ctu_size, const int16_t* coeff_host,
                             const uint8_t* ctu_enable, int clp_min, int clp_max, void* stream)
{
  return alf_filter_common(false, src, src_stride, dst, dst_stride, width, height, ctu_size, nullptr, 0,
                           coeff_host, ctu_enable, clp_min, clp_max, stream);
}

int vvcgpu_alf_filter_picture(const vvcgpu_planes* src, const vvcgpu_planes* dst, int width, int height, int ctu_size, const uint16_t* cls,
                              int filter_type, const int16_t* luma_coeff_host, const int16_t* chroma_coeff_host, const uint8_t* enable_y,
                              const uint8_t* enable_cb, const uint8_t* enable_cr, int clp_min, int clp_max, void* stream)
{
  VVC_CHECK_ARG(src && dst && cls && luma_coeff_host && chroma_coeff_host, "alf_filter_picture: null pointer");
  VVC_CHECK_ARG(width > 0 && height > 0 && (width & 7) == 0 && (height & 7) == 0, "alf_filter_picture: width/height must be multiples of 8 (got %dx%d)", width, height);
  VVC_CHECK_ARG(ctu_size >= 8 && (ctu_size & 7) == 0, "alf_filter_picture: bad ctu size %d", ctu_size);
  VVC_CHECK_ARG(filter_type == 0 || filter_type == 1, "alf_filter_picture: filter_type %d", filter_type);
  AlfFilter3 p;
  memset(&p, 0, sizeof p);
  const uint8_t* en[3] = { enable_y, enable_cb, enable_cr };
  for (int c = 0; c < 3; c++)
  {
    const int w = c ? width >> 1 : width;
    VVC_CHECK_ARG(src->p[c] && dst->p[c] && src->p[c] != dst->p[c] && src->stride[c] >= w && dst->stride[c] >= w && (dst->stride[c] & 3) == 0 &&
                  ((uintptr_t)dst->p[c] & 7) == 0, "alf_filter_picture: plane %d (dst needs stride %% 4 == 0 and 8-byte alignment)", c);
    p.a[c] = AlfPlane{ src->p[c], dst->p[c], en[c], src->stride[c], dst->stride[c] };
  }
  memcpy(p.luma.c, luma_coeff_host, sizeof(int16_t) * 25 * 13);
  memcpy(p.chroma, chroma_coeff_host, sizeof(int16_t) * 7);
  p.cls = cls; p.w = width; p.h = height; p.ctu = ctu_size; p.clpMin = clp_min; p.clpMax = clp_max;
  p.glx = cdiv(width, FW); p.nLuma = p.glx * cdiv(height, FH);
  p.gcx = cdiv(width >> 1, FW); p.gcy = cdiv(height >> 1, FH);
  const int total = p.nLuma + 2 * p.gcx * p.gcy;
  p.total = total; p.xcd = vvc_xcd_on();
  if (filter_type) hipLaunchKernelGGL(alf_filter_picture_kernel<true>, dim3(vvc_xcd_grid2(p.nLuma, total, p.xcd)), dim3(128), 0, (hipStream_t)stream, p);
  else             hipLaunchKernelGGL(alf_filter_picture_kernel<false>, dim3(vvc_xcd_grid2(p.nLuma, total, p.xcd)), dim3(128), 0, (hipStream_t)stream, p);
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}

}  // extern "C"
