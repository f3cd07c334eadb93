// transform.hip -- 2-D separable integer transforms (T1 forward, T2 inverse, T3 transform skip) for gfx950.
//
// Reference behaviour reproduced (bit-exact): xTrMxN_EMT / xITrMxN_EMT (CommonLib/TrQuant.cpp:138-310) with the 1-D
// stages of TrQuant_EMT.cpp expressed as integer matrix products with the reference's own tables (tr_tables.inc, dumped
// from initROM(); tests/golden/gen_tr_tables.py proves fast transform == table for every slot), intermediate rounding
// `(sum + rnd) >> shift` between the stages (:214-215), inverse stages clipped to [-2^15, 2^15-1] (:253-256), zero-out of
// columns/rows >= 32 (:157-162, :755-759); xTransformSkip / xITransformSkip (:795-847, :1112-1163).
//
// Design: one workgroup per TU, both 1-D stages inside the workgroup with the intermediate in LDS (the reference's
// alloca'd `tmp`, never in HBM).  Products fit 24 x 24 bits (|coef| <= 362, data < 2^16), accumulation is exact int32.
// One wave handles one output ROW/COLUMN index per iteration so the matrix row is wave-uniform (scalar loads) while
// the data operand streams from LDS with an odd dword pitch (conflict-free).
#include "common.h"
#include "tr_tables.inc"

namespace {

// int32 copies of the golden matrices: d_tr32[type][size] row-major (T[k][n]) and d_tr32t (transposed, T[n][k]) so that the
// row a wave needs is always contiguous -> scalar (SGPR) loads.  Layout as VVC_TR_TABLES: size N at (N*N-4)/3.
__device__ int d_tr32[3 * 5460];
__device__ int d_tr32t[3 * 5460];
__device__ short d_trTables[3 * 5460];

__device__ __forceinline__ const int* tr32(int type, int n)  { return d_tr32  + type * 5460 + (n * n - 4) / 3; }
__device__ __forceinline__ const int* tr32t(int type, int n) { return d_tr32t + type * 5460 + (n * n - 4) / 3; }
__device__ __forceinline__ const short* tr16(int type, int n) { return d_trTables + type * 5460 + (n * n - 4) / 3; }
__device__ __forceinline__ int ilog2(int v) { return 31 - __clz(v); }

constexpr int MAXN = 64;
typedef short short2v __attribute__((ext_vector_type(2)));

// ---- forward, stage 1: lane = row i; the row lives in registers as packed int16 pairs; T row j is wave-uniform.
template <int W>
__device__ __forceinline__ void fwd_stage1(const Pel* __restrict__ resi, int stride, int h, int lane, int wj, int s1,
                                           const short* __restrict__ Th, int* __restrict__ tmpL, int ph)
{
  if (lane >= h) return;
  const Pel* row = resi + (size_t)lane * stride;
  int x[W];
#pragma unroll
  for (int k = 0; k < W; k++) x[k] = row[k];
  const int rnd = 1 << (s1 - 1);
  for (int j = 0; j < wj; j++)
  {
    const short* t = Th + j * W;                       // uniform address -> scalar loads
    int sum = 0;
#pragma unroll
    for (int k = 0; k < W; k++) sum += __mul24(x[k], (int)t[k]);
    tmpL[j * ph + lane] = (sum + rnd) >> s1;
  }
}
// ---- forward, stage 2: lane = horizontal frequency i (< wj); its tmp row comes from LDS (odd pitch: conflict free)
template <int H>
__device__ __forceinline__ void fwd_stage2(const int* __restrict__ tmpL, int ph, int w, int wj, int hj, int lane, int s2,
                                           const int* __restrict__ Tv, TCoeff* __restrict__ coeff)
{
  if (lane >= w) return;
  if (lane >= wj) { for (int j = 0; j < H; j++) coeff[j * w + lane] = 0; return; }
  int t[H];
#pragma unroll
  for (int k = 0; k < H; k++) t[k] = tmpL[lane * ph + k];
  const int rnd = 1 << (s2 - 1);
  for (int j = 0; j < hj; j++)
  {
    const int* tv = Tv + j * H;                        // uniform
    int sum = 0;
#pragma unroll
    for (int k = 0; k < H; k++) sum += __mul24(t[k], tv[k]);
    coeff[j * w + lane] = (sum + rnd) >> s2;
  }
  for (int j = hj; j < H; j++) coeff[j * w + lane] = 0;
}

template <int W>
__device__ __forceinline__ void fwd_dispatch_h(int h, const int* tmpL, int ph, int wj, int hj, int lane, int s2, int trVer, TCoeff* coeff)
{
  switch (h)
  {
  case 2:  fwd_stage2<2>(tmpL, ph, W, wj, hj, lane, s2, tr32(trVer, 2), coeff); break;
  case 4:  fwd_stage2<4>(tmpL, ph, W, wj, hj, lane, s2, tr32(trVer, 4), coeff); break;
  case 8:  fwd_stage2<8>(tmpL, ph, W, wj, hj, lane, s2, tr32(trVer, 8), coeff); break;
  case 16: fwd_stage2<16>(tmpL, ph, W, wj, hj, lane, s2, tr32(trVer, 16), coeff); break;
  case 32: fwd_stage2<32>(tmpL, ph, W, wj, hj, lane, s2, tr32(trVer, 32), coeff); break;
  default: fwd_stage2<64>(tmpL, ph, W, wj, hj, lane, s2, tr32(trVer, 64), coeff); break;
  }
}

template <int W>
__device__ __forceinline__ void fwd_tu(const vvcgpu_tr_desc& d, const Pel* resi, TCoeff* coeff, int bd, int lane, int* tmpL)
{
  const int h = d.h, lw = ilog2(W), lh = ilog2(h);
  const int s1 = lw + bd + 6 - 15 + 2, s2 = lh + 6 + 2;
  const int wj = W > 32 ? 32 : W, hj = h > 32 ? 32 : h;
  const int ph = h + 1;
  fwd_stage1<W>(resi, d.resi_stride, h, lane, wj, s1, tr16(d.tr_hor, W), tmpL, ph);
  __syncthreads();
  fwd_dispatch_h<W>(h, tmpL, ph, wj, hj, lane, s2, d.tr_ver, coeff);
}

// One wave per TU.
__global__ __launch_bounds__(64) void tr_fwd_kernel(const Pel* __restrict__ resiBase, TCoeff* __restrict__ coeffBase,
                                                    const vvcgpu_tr_desc* __restrict__ descs, int bd)
{
  __shared__ int tmpL[32 * (MAXN + 1)];            // tmp[j][i], j < 32 kept columns, pitch h+1
  const int lane = threadIdx.x;
  const vvcgpu_tr_desc d = descs[blockIdx.x];
  const int w = d.w, h = d.h, lw = ilog2(w), lh = ilog2(h);
  const Pel* resi = resiBase + d.resi_off;
  TCoeff* coeff = coeffBase + d.coeff_off;
  if (d.tr_hor == 3)
  {
    int shift = 15 - bd - ((lw + lh) >> 1), scale = 1;
    if ((lw + lh) & 1) { shift -= 8; scale = 181; }
    for (int i = lane; i < w * h; i += 64)
    {
      const int y = i >> lw, x = i & (w - 1);
      const int v = resi[(size_t)y * d.resi_stride + x] * scale;
      coeff[i] = shift >= 0 ? v << shift : (v + (1 << (-shift - 1))) >> -shift;
    }
    return;
  }
  switch (w)
  {
  case 2:  fwd_tu<2>(d, resi, coeff, bd, lane, tmpL); break;
  case 4:  fwd_tu<4>(d, resi, coeff, bd, lane, tmpL); break;
  case 8:  fwd_tu<8>(d, resi, coeff, bd, lane, tmpL); break;
  case 16: fwd_tu<16>(d, resi, coeff, bd, lane, tmpL); break;
  case 32: fwd_tu<32>(d, resi, coeff, bd, lane, tmpL); break;
  default: fwd_tu<64>(d, resi, coeff, bd, lane, tmpL); break;
  }
}

// ---- inverse, stage 1 (vertical): lane = kept column i; its coefficient column is read coalesced across lanes.
template <int H>
__device__ __forceinline__ void inv_stage1(const TCoeff* __restrict__ coeff, int w, int wj, int lane, const int* __restrict__ TvT,
                                           int* __restrict__ tmpL, int ph)
{
  constexpr int HJ = H > 32 ? 32 : H;
  if (lane >= wj) return;
  int c[HJ];
#pragma unroll
  for (int k = 0; k < HJ; k++) c[k] = coeff[k * w + lane];
  for (int j = 0; j < H; j++)
  {
    const int* t = TvT + j * H;                        // TvT[j][k] = Tv[k][j], uniform
    int sum = 0;
#pragma unroll
    for (int k = 0; k < HJ; k++) sum += c[k] * t[k];
    tmpL[lane * ph + j] = clip3(-(1 << 15), (1 << 15) - 1, (sum + 256) >> 9);
  }
}
// ---- inverse, stage 2 (horizontal): lane = row i
template <int W>
__device__ __forceinline__ void inv_stage2(const int* __restrict__ tmpL, int ph, int h, int lane, int s2, const int* __restrict__ ThT,
                                           Pel* __restrict__ resi, int stride)
{
  constexpr int WJ = W > 32 ? 32 : W;
  if (lane >= h) return;
  int t[WJ];
#pragma unroll
  for (int k = 0; k < WJ; k++) t[k] = tmpL[k * ph + lane];
  const int rnd = 1 << (s2 - 1);
  Pel* row = resi + (size_t)lane * stride;
  for (int j = 0; j < W; j++)
  {
    const int* th = ThT + j * W;                       // ThT[j][k] = Th[k][j], uniform
    int sum = 0;
#pragma unroll
    for (int k = 0; k < WJ; k++) sum += t[k] * th[k];
    row[j] = (short)clip3(-(1 << 15), (1 << 15) - 1, (sum + rnd) >> s2);
  }
}

template <int W>
__device__ __forceinline__ void inv_tu(const vvcgpu_tr_desc& d, const TCoeff* coeff, Pel* resi, int bd, int lane, int* tmpL)
{
  const int h = d.h;
  const int s2 = (6 + 15 - 1) - bd + 2;
  const int wj = W > 32 ? 32 : W;
  const int ph = h + 1;
  switch (h)
  {
  case 2:  inv_stage1<2>(coeff, W, wj, lane, tr32t(d.tr_ver, 2), tmpL, ph); break;
  case 4:  inv_stage1<4>(coeff, W, wj, lane, tr32t(d.tr_ver, 4), tmpL, ph); break;
  case 8:  inv_stage1<8>(coeff, W, wj, lane, tr32t(d.tr_ver, 8), tmpL, ph); break;
  case 16: inv_stage1<16>(coeff, W, wj, lane, tr32t(d.tr_ver, 16), tmpL, ph); break;
  case 32: inv_stage1<32>(coeff, W, wj, lane, tr32t(d.tr_ver, 32), tmpL, ph); break;
  default: inv_stage1<64>(coeff, W, wj, lane, tr32t(d.tr_ver, 64), tmpL, ph); break;
  }
  __syncthreads();
  inv_stage2<W>(tmpL, ph, h, lane, s2, tr32t(d.tr_hor, W), resi, d.resi_stride);
}

__global__ __launch_bounds__(64) void tr_inv_kernel(const TCoeff* __restrict__ coeffBase, Pel* __restrict__ resiBase,
                                                    const vvcgpu_tr_desc* __restrict__ descs, int bd)
{
  __shared__ int tmpL[32 * (MAXN + 1)];            // tmp[i][j], i < 32 kept columns, pitch h+1
  const int lane = threadIdx.x;
  const vvcgpu_tr_desc d = descs[blockIdx.x];
  const int w = d.w, h = d.h, lw = ilog2(w), lh = ilog2(h);
  const TCoeff* coeff = coeffBase + d.coeff_off;
  Pel* resi = resiBase + d.resi_off;
  if (d.tr_hor == 3)
  {
    int shift = 15 - bd - ((lw + lh) >> 1), scale = 1;
    if ((lw + lh) & 1) { shift += 7; scale = 181; }
    for (int i = lane; i < w * h; i += 64)
    {
      const int y = i >> lw, x = i & (w - 1);
      const int c = coeff[i] * scale;
      resi[(size_t)y * d.resi_stride + x] = (short)(shift >= 0 ? (c + (shift ? 1 << (shift - 1) : 0)) >> shift : c << -shift);
    }
    return;
  }
  switch (w)
  {
  case 2:  inv_tu<2>(d, coeff, resi, bd, lane, tmpL); break;
  case 4:  inv_tu<4>(d, coeff, resi, bd, lane, tmpL); break;
  case 8:  inv_tu<8>(d, coeff, resi, bd, lane, tmpL); break;
  case 16: inv_tu<16>(d, coeff, resi, bd, lane, tmpL); break;
  case 32: inv_tu<32>(d, coeff, resi, bd, lane, tmpL); break;
  default: inv_tu<64>(d, coeff, resi, bd, lane, tmpL); break;
  }
}

static bool g_tablesUploaded[64] = { false };

static int ensure_tables()
{
  int dev = 0;
  VVC_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) { vvcgpu_set_error("device index %d out of range", dev); return VVCGPU_E_DEVICE; }
  if (!g_tablesUploaded[dev])
  {
    VVC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(d_trTables), VVC_TR_TABLES, sizeof(VVC_TR_TABLES)));
    static int t32[3 * 5460], t32t[3 * 5460];
    for (int t = 0; t < 3; t++)
      for (int n = 2; n <= 64; n <<= 1)
      {
        const int o = t * 5460 + (n * n - 4) / 3;
        for (int k = 0; k < n; k++)
          for (int j = 0; j < n; j++) { t32[o + k * n + j] = VVC_TR_TABLES[o + k * n + j]; t32t[o + j * n + k] = VVC_TR_TABLES[o + k * n + j]; }
      }
    VVC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(d_tr32), t32, sizeof(t32)));
    VVC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(d_tr32t), t32t, sizeof(t32t)));
    g_tablesUploaded[dev] = true;
  }
  return VVCGPU_OK;
}

static int check_descs_args(const void* a, const void* b, const void* d, int n, int bd, const char* who)
{
  VVC_CHECK_ARG(n >= 0, "%s: n %d", who, n);
  if (n == 0) return 1;
  VVC_CHECK_ARG(a && b && d, "%s: null pointer", who);
  if (bd < 8 || bd > 10) { vvcgpu_set_error("%s: bit depth %d outside 8..10", who, bd); return VVCGPU_E_UNSUPPORTED; }
  return VVCGPU_OK;
}

}  // namespace

extern "C" {

int vvcgpu_tr_fwd_batch(const vvc_pel* resi_base, vvc_coef* coeff_base, const vvcgpu_tr_desc* descs, int n,
                        int bit_depth, void* stream)
{
  const int rc = check_descs_args(resi_base, coeff_base, descs, n, bit_depth, "tr_fwd_batch");
  if (rc) return rc > 0 ? VVCGPU_OK : rc;
  const int rt = ensure_tables();
  if (rt) return rt;
  hipLaunchKernelGGL(tr_fwd_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, resi_base, coeff_base, descs, bit_depth);
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}

int vvcgpu_tr_inv_batch(const vvc_coef* coeff_base, vvc_pel* resi_base, const vvcgpu_tr_desc* descs, int n,
                        int bit_depth, void* stream)
{
  const int rc = check_descs_args(coeff_base, resi_base, descs, n, bit_depth, "tr_inv_batch");
  if (rc) return rc > 0 ? VVCGPU_OK : rc;
  const int rt = ensure_tables();
  if (rt) return rt;
  hipLaunchKernelGGL(tr_inv_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, coeff_base, resi_base, descs, bit_depth);
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}

const int16_t* vvcgpu_tr_matrix_host(int type, int n)
{
  if (type < 0 || type > 2 || n < 2 || n > 64 || (n & (n - 1))) return nullptr;
  return VVC_TR_TABLES + type * 5460 + (n * n - 4) / 3;
}

}  // extern "C"
