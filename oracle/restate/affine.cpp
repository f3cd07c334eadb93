// oracle/restate/affine.cpp -- TEST INFRASTRUCTURE: scalar restatement of the affine gradient search kernels (next row N3).
//   AffineGradientSearch::xHorizontalSobelFilter / xVerticalSobelFilter / xEqualCoeffComputer   CommonLib/AffineGradientSearch.cpp:66-174
// Pinned against the compiled reference (scalar bodies and the SIMD table slots) by tests/golden/affine.npz.
#include "orc_common.h"

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// every output equals the Sobel response at the nearest interior position: that is what the copy rules of :83-96 / :116-129 amount to
ORC_API int orc_affine_sobel(int vertical, const Pel* pred, int predStride, int32_t* deriv, int derivStride, int w, int h)
{
  for (int j = 0; j < h; j++)
    for (int k = 0; k < w; k++)
    {
      const int y = clampi(j, 1, h - 2), x = clampi(k, 1, w - 2);
      const Pel* c = pred + y * predStride + x;
      int v;
      if (!vertical)
        v = c[1 - predStride] - c[-1 - predStride] + (c[1] << 1) - (c[-1] << 1) + c[1 + predStride] - c[-1 + predStride];
      else
        v = c[predStride - 1] - c[-predStride - 1] + (c[predStride] << 1) - (c[-predStride] << 1) + c[predStride + 1] - c[-predStride + 1];
      deriv[j * derivStride + k] = v;
    }
  return 0;
}

ORC_API int orc_affine_equal_coeff(const Pel* resi, const int32_t* gx, const int32_t* gy, int derivStride, int w, int h, int sixParam, int64_t* out)
{
  const int P = sixParam ? 6 : 4;
  for (int i = 0; i < 49; i++) out[i] = 0;
  for (int j = 0; j < h; j++)
    for (int k = 0; k < w; k++)
    {
      const int idx = j * derivStride + k;
      int iC[6];
      if (!sixParam) { iC[0] = gx[idx]; iC[1] = k * gx[idx] + j * gy[idx]; iC[2] = gy[idx]; iC[3] = j * gx[idx] - k * gy[idx]; }
      else { iC[0] = gx[idx]; iC[1] = k * gx[idx]; iC[2] = gy[idx]; iC[3] = k * gy[idx]; iC[4] = j * gx[idx]; iC[5] = j * gy[idx]; }
      for (int col = 0; col < P; col++)
      {
        for (int row = 0; row < P; row++) out[(col + 1) * 7 + row] += (int64_t)iC[col] * iC[row];
        out[(col + 1) * 7 + P] += ((int64_t)iC[col] * resi[idx]) << 3;
      }
    }
  return 0;
}

ORC_API int orc_affine_sobel_batch(int vertical, const Pel* predBase, int32_t* derivBase, const vvcgpu_afg_desc* d, int n)
{
  for (int i = 0; i < n; i++) orc_affine_sobel(vertical, predBase + d[i].pred_off, d[i].pred_stride, derivBase + d[i].deriv_off, d[i].deriv_stride, d[i].w, d[i].h);
  return 0;
}
ORC_API int orc_affine_equal_coeff_batch(const Pel* resiBase, const int32_t* gxBase, const int32_t* gyBase, const vvcgpu_afe_desc* d, int n, int64_t* out)
{
  for (int i = 0; i < n; i++)
    orc_affine_equal_coeff(resiBase + d[i].resi_off, gxBase + d[i].deriv_off, gyBase + d[i].deriv_off, d[i].deriv_stride, d[i].w, d[i].h, d[i].six_param, out + (size_t)i * 49);
  return 0;
}
