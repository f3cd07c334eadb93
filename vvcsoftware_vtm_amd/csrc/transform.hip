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

__device__ short d_trTables[3 * 5460];

__device__ __forceinline__ const short* tr_matrix(int type, int n) { return d_trTables + type * 5460 + (n * n - 4) / 3; }
__device__ __forceinline__ int ilog2(int v) { return 31 - __clz(v); }

constexpr int MAXN = 64;

__global__ __launch_bounds__(256) void tr_fwd_kernel(const Pel* __restrict__ resiBase, TCoeff* __restrict__ coeffBase,
                                                     const vvcgpu_tr_desc* __restrict__ descs, int bd)
{
  __shared__ int blk[MAXN * (MAXN + 1)];          // residual, pitch w+1
  __shared__ int tmp[MAXN * (MAXN + 1)];          // tmp[j][i], pitch h+1
  const int tid = threadIdx.x;
  const vvcgpu_tr_desc d = descs[blockIdx.x];
  const int w = d.w, h = d.h, lw = ilog2(w), lh = ilog2(h);
  const Pel* resi = resiBase + d.resi_off;
  TCoeff* coeff = coeffBase + d.coeff_off;
  if (d.tr_hor == 3)
  {
    int shift = 15 - bd - ((lw + lh) >> 1), scale = 1;
    if ((lw + lh) & 1) { shift -= 8; scale = 181; }
    for (int i = tid; i < w * h; i += 256)
    {
      const int y = i >> lw, x = i & (w - 1);
      const int v = resi[(size_t)y * d.resi_stride + x] * scale;
      coeff[i] = shift >= 0 ? v << shift : (v + (1 << (-shift - 1))) >> -shift;
    }
    return;
  }
  const int s1 = lw + bd + 6 - 15 + 2, s2 = lh + 6 + 2;
  const int wj = w > 32 ? 32 : w, hj = h > 32 ? 32 : h;          // kept columns / rows (zero-out threshold 32)
  const short* Th = tr_matrix(d.tr_hor, w);
  const short* Tv = tr_matrix(d.tr_ver, h);
  const int pw = w + 1, ph = h + 1;
  for (int i = tid; i < w * h; i += 256)
  {
    const int y = i >> lw, x = i & (w - 1);
    blk[y * pw + x] = resi[(size_t)y * d.resi_stride + x];
  }
  __syncthreads();
  // stage 1: tmp[j][i] = rnd(sum_k blk[i][k] * Th[j][k]); lanes run over i (rows), j is uniform per iteration
  for (int o = tid; o < wj * h; o += 256)
  {
    const int j = o >> lh, i = o & (h - 1);
    const short* t = Th + j * w;
    const int* b = blk + i * pw;
    int sum = 0;
    for (int k = 0; k < w; k++) sum += __mul24(b[k], (int)t[k]);
    tmp[j * ph + i] = (sum + (1 << (s1 - 1))) >> s1;
  }
  __syncthreads();
  // stage 2: coeff[j][i] = rnd(sum_k tmp[i][k] * Tv[j][k]); lanes run over i (horizontal frequency) -> coalesced stores
  for (int o = tid; o < w * h; o += 256)
  {
    const int j = o >> lw, i = o & (w - 1);
    int v = 0;
    if (i < wj && j < hj)
    {
      const short* t = Tv + j * h;
      const int* b = tmp + i * ph;
      int sum = 0;
      for (int k = 0; k < h; k++) sum += __mul24(b[k], (int)t[k]);
      v = (sum + (1 << (s2 - 1))) >> s2;
    }
    coeff[o] = v;
  }
}

__global__ __launch_bounds__(256) void tr_inv_kernel(const TCoeff* __restrict__ coeffBase, Pel* __restrict__ resiBase,
                                                     const vvcgpu_tr_desc* __restrict__ descs, int bd)
{
  __shared__ int cf[MAXN * (MAXN + 1)];           // coefficients cf[k][i], pitch w+1
  __shared__ int tmp[MAXN * (MAXN + 1)];          // tmp[i][j] (column i, row j), pitch h+1
  const int tid = threadIdx.x;
  const vvcgpu_tr_desc d = descs[blockIdx.x];
  const int w = d.w, h = d.h, lw = ilog2(w), lh = ilog2(h);
  const TCoeff* coeff = coeffBase + d.coeff_off;
  Pel* resi = resiBase + d.resi_off;
  if (d.tr_hor == 3)
  {
    int shift = 15 - bd - ((lw + lh) >> 1), scale = 1;
    if ((lw + lh) & 1) { shift += 7; scale = 181; }
    for (int i = tid; i < w * h; i += 256)
    {
      const int y = i >> lw, x = i & (w - 1);
      const int c = coeff[i] * scale;
      resi[(size_t)y * d.resi_stride + x] = (short)(shift >= 0 ? (c + (shift ? 1 << (shift - 1) : 0)) >> shift : c << -shift);
    }
    return;
  }
  const int s1 = 6 + 1 + 2, s2 = (6 + 15 - 1) - bd + 2;
  const int cmin = -(1 << 15), cmax = (1 << 15) - 1;
  const int wj = w > 32 ? 32 : w, hj = h > 32 ? 32 : h;
  const short* Th = tr_matrix(d.tr_hor, w);
  const short* Tv = tr_matrix(d.tr_ver, h);
  const int pw = w + 1, ph = h + 1;
  for (int i = tid; i < w * hj; i += 256)         // only the kept rows are read (:755-759)
  {
    const int y = i >> lw, x = i & (w - 1);
    cf[y * pw + x] = coeff[i];
  }
  __syncthreads();
  // vertical stage: tmp[i][j] = clip(rnd(sum_{k<hj} cf[k][i] * Tv[k][j])) for kept columns i
  for (int o = tid; o < wj * h; o += 256)
  {
    const int j = o / wj, i = o - j * wj;         // lanes run over i -> cf[k][i] conflict-free, Tv[k][j] uniform
    int sum = 0;
    for (int k = 0; k < hj; k++) sum += cf[k * pw + i] * (int)Tv[k * h + j];
    tmp[i * ph + j] = clip3(cmin, cmax, (sum + (1 << (s1 - 1))) >> s1);
  }
  __syncthreads();
  // horizontal stage: resi[i][j] = clip(rnd(sum_{k<wj} tmp[k][i] * Th[k][j])); lanes run over j -> coalesced stores
  for (int o = tid; o < w * h; o += 256)
  {
    const int i = o >> lw, j = o & (w - 1);
    int sum = 0;
    for (int k = 0; k < wj; k++) sum += tmp[k * ph + i] * (int)Th[k * w + j];
    resi[(size_t)i * d.resi_stride + j] = (short)clip3(cmin, cmax, (sum + (1 << (s2 - 1))) >> s2);
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
  hipLaunchKernelGGL(tr_fwd_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, resi_base, coeff_base, descs, bit_depth);
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
  hipLaunchKernelGGL(tr_inv_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, coeff_base, resi_base, descs, bit_depth);
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}

const int16_t* vvcgpu_tr_matrix_host(int type, int n)
{
  if (type < 0 || type > 2 || n < 2 || n > 64 || (n & (n - 1))) return nullptr;
  return VVC_TR_TABLES + type * 5460 + (n * n - 4) / 3;
}

}  // extern "C"
