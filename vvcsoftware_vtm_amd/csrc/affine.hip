// affine.hip -- affine gradient search kernels (next row N3) for gfx950.
//
// Reference behaviour reproduced (bit-exact): AffineGradientSearch::xHorizontalSobelFilter / xVerticalSobelFilter /
// xEqualCoeffComputer (CommonLib/AffineGradientSearch.cpp:66-174; SIMD twins x86/AffineGradientSearchX86.h:72-312), the three
// table slots of AffineGradientSearch.h:50-54.
//
// Design: one workgroup per PU.  Sobel: every output sample is the 3x3 response at the nearest INTERIOR position (that is what
// the reference's ring-copy rules amount to), so the ring needs no second pass.  Equal coefficients: every lane walks its
// share of the block with the <= 6 x 7 sums in 64-bit registers, then wave shuffles and a small LDS stage combine them.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void affine_sobel_kernel(int vertical, const Pel* __restrict__ predBase, int* __restrict__ derivBase,
                                                           const vvcgpu_afg_desc* __restrict__ descs)
{
  const vvcgpu_afg_desc d = descs[blockIdx.x];
  const Pel* pred = predBase + d.pred_off;
  int* deriv = derivBase + d.deriv_off;
  const int w = d.w, h = d.h, ps = d.pred_stride;
  for (int i = threadIdx.x; i < w * h; i += 256)
  {
    const int j = i / w, k = i - j * w;
    const int y = min(max(j, 1), h - 2), x = min(max(k, 1), w - 2);
    const Pel* c = pred + (ptrdiff_t)y * ps + x;
    int v;
    if (!vertical) v = c[1 - ps] - c[-1 - ps] + (c[1] << 1) - (c[-1] << 1) + c[1 + ps] - c[-1 + ps];
    else           v = c[ps - 1] - c[-ps - 1] + (c[ps] << 1) - (c[-ps] << 1) + c[ps + 1] - c[-ps + 1];
    deriv[(ptrdiff_t)j * d.deriv_stride + k] = v;
  }
}

template <int P>
__device__ __forceinline__ void eq_accumulate(const Pel* __restrict__ resi, const int* __restrict__ gx, const int* __restrict__ gy, int stride,
                                              int w, int h, int tid, long long (&acc)[P][P + 1])
{
  for (int i = tid; i < w * h; i += 256)
  {
    const int j = i / w, k = i - j * w;
    const int idx = j * stride + k;
    const int x = gx[idx], y = gy[idx];
    int iC[P];
    if (P == 4) { iC[0] = x; iC[1] = k * x + j * y; iC[2] = y; iC[3] = j * x - k * y; }
    else        { iC[0] = x; iC[1] = k * x; iC[2] = y; iC[3] = k * y; iC[4] = j * x; iC[5] = j * y; }
    const long long r = (long long)resi[idx];
#pragma unroll
    for (int col = 0; col < P; col++)
    {
#pragma unroll
      for (int row = 0; row < P; row++) acc[col][row] += (long long)iC[col] * iC[row];
      acc[col][P] += ((long long)iC[col] * r) << 3;
    }
  }
}

template <int P>
__device__ __forceinline__ void eq_block(const vvcgpu_afe_desc& d, const Pel* resiBase, const int* gxBase, const int* gyBase, long long* out,
                                         long long (*part)[48])
{
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  long long acc[P][P + 1];
#pragma unroll
  for (int c = 0; c < P; c++)
#pragma unroll
    for (int r = 0; r <= P; r++) acc[c][r] = 0;
  eq_accumulate<P>(resiBase + d.resi_off, gxBase + d.deriv_off, gyBase + d.deriv_off, d.deriv_stride, d.w, d.h, tid, acc);
#pragma unroll
  for (int c = 0; c < P; c++)
#pragma unroll
    for (int r = 0; r <= P; r++)
    {
      long long v = acc[c][r];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0) part[wave][c * 7 + r] = v;
    }
  __syncthreads();
  if (tid < 49)
  {
    const int row7 = tid / 7, col7 = tid - row7 * 7;       // out[row7][col7]; rows 1..P hold the equations
    long long v = 0;
    if (row7 >= 1 && row7 <= P && col7 <= P)
      for (int k = 0; k < 4; k++) v += part[k][(row7 - 1) * 7 + col7];
    out[tid] = v;
  }
}

__global__ __launch_bounds__(256) void affine_equal_coeff_kernel(const Pel* __restrict__ resiBase, const int* __restrict__ gxBase,
                                                                 const int* __restrict__ gyBase, const vvcgpu_afe_desc* __restrict__ descs,
                                                                 long long* __restrict__ out)
{
  __shared__ long long part[4][48];
  const vvcgpu_afe_desc d = descs[blockIdx.x];
  long long* o = out + (size_t)blockIdx.x * 49;
  if (d.six_param) eq_block<6>(d, resiBase, gxBase, gyBase, o, part);
  else             eq_block<4>(d, resiBase, gxBase, gyBase, o, part);
}

}  // namespace

extern "C" {

int vvcgpu_affine_sobel_batch(int vertical, const vvc_pel* pred_base, int32_t* deriv_base, const vvcgpu_afg_desc* descs, int n, void* stream)
{
  VVC_CHECK_ARG(n >= 0, "affine_sobel_batch: n %d", n);
  if (n == 0) return VVCGPU_OK;
  VVC_CHECK_ARG(pred_base && deriv_base && descs, "affine_sobel_batch: null pointer");
  VVC_CHECK_ARG(vertical == 0 || vertical == 1, "affine_sobel_batch: vertical %d", vertical);
  hipLaunchKernelGGL(affine_sobel_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, vertical, pred_base, deriv_base, descs);
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}

int vvcgpu_affine_equal_coeff_batch(const vvc_pel* resi_base, const int32_t* derivx_base, const int32_t* derivy_base,
                                    const vvcgpu_afe_desc* descs, int n, int64_t* out, void* stream)
{
  VVC_CHECK_ARG(n >= 0, "affine_equal_coeff_batch: n %d", n);
  if (n == 0) return VVCGPU_OK;
  VVC_CHECK_ARG(resi_base && derivx_base && derivy_base && descs && out, "affine_equal_coeff_batch: null pointer");
  hipLaunchKernelGGL(affine_equal_coeff_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, resi_base, derivx_base, derivy_base, descs,
                     reinterpret_cast<long long*>(out));
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}

}  // extern "C"
