// oracle/restate/transform.cpp -- TEST INFRASTRUCTURE: CPU restatement of the 2-D integer transforms.
// Matrix generation follows initROM (CommonLib/Rom.cpp:245-299); the 2-D drivers follow xTrMxN_EMT / xITrMxN_EMT
// (CommonLib/TrQuant.cpp:138-310) with the 1-D stages written as plain matrix products -- tests/golden/gen_tr_tables.py
// shows (against the compiled reference) that every fast transform of TrQuant_EMT.cpp equals its ROM matrix exactly.
// Transform skip follows xTransformSkip / xITransformSkip (TrQuant.cpp:795-847, 1112-1163), rotation off.
#include "orc_common.h"
#include <cmath>
#include <vector>

static int16_t g_tr[3][7][64 * 64];
static bool g_trInit = false;
static int ilog2(int v) { int l = 0; while ((1 << l) < v) l++; return l; }

static void initTables()
{
  if (g_trInit) return;
  const double PI = 3.14159265358979323846;
  for (int lg = 1; lg <= 6; lg++)
  {
    const int c = 1 << lg;
    const double s = sqrt((double)c) * (64 << 2);                                      // COM16_C806_TRANS_PREC = 2
    for (int k = 0; k < c; k++)
      for (int n = 0; n < c; n++)
      {
        double w0 = k == 0 ? sqrt(0.5) : 1;
        double v = cos(PI * (n + 0.5) * k / c) * w0 * sqrt(2.0 / c);
        g_tr[0][lg][k * c + n] = (int16_t)(s * v + (v > 0 ? 0.5 : -0.5));
        v = cos(PI * (k + 0.5) * (n + 0.5) / (c + 0.5)) * sqrt(2.0 / (c + 0.5));
        g_tr[1][lg][k * c + n] = (int16_t)(s * v + (v > 0 ? 0.5 : -0.5));
        v = sin(PI * (k + 0.5) * (n + 1) / (c + 0.5)) * sqrt(2.0 / (c + 0.5));
        g_tr[2][lg][k * c + n] = (int16_t)(s * v + (v > 0 ? 0.5 : -0.5));
      }
  }
  g_trInit = true;
}
ORC_API const int16_t* orc_tr_matrix(int type, int n) { initTables(); return g_tr[type][ilog2(n)]; }

ORC_API int orc_tr_fwd(const Pel* resi, int stride, TCoeff* coeff, int w, int h, int trHor, int trVer, int bd)
{
  initTables();
  const int lw = ilog2(w), lh = ilog2(h);
  if (trHor == 3)                                                    // transform skip
  {
    int shift = 15 - bd - ((lw + lh) >> 1), scale = 1;
    if ((lw + lh) & 1) { shift -= 8; scale = 181; }                  // ADJ_DEQUANT_SHIFT
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++)
    {
      const int v = resi[y * stride + x] * scale;
      coeff[y * w + x] = shift >= 0 ? v << shift : (v + (1 << (-shift - 1))) >> -shift;
    }
    return 0;
  }
  const int s1 = lw + bd + 6 - 15 + 2, s2 = lh + 6 + 2;             // :151-152
  const int skipW = w > 32 ? w - 32 : 0, skipH = h > 32 ? h - 32 : 0;   // useQTBT (:157-162)
  const int16_t* Th = g_tr[trHor][lw]; const int16_t* Tv = g_tr[trVer][lh];
  std::vector<int> tmp((size_t)w * h, 0);
  for (int i = 0; i < h; i++)                                        // 1st: rows, output transposed tmp[j*h + i]
    for (int j = 0; j < w - skipW; j++)
    {
      int sum = 0;
      for (int k = 0; k < w; k++) sum += resi[i * stride + k] * Th[j * w + k];
      tmp[j * h + i] = (sum + (1 << (s1 - 1))) >> s1;
    }
  for (int i = 0; i < w * h; i++) coeff[i] = 0;
  for (int i = 0; i < w - skipW; i++)                                // 2nd: columns, coeff[j*w + i]
    for (int j = 0; j < h - skipH; j++)
    {
      int sum = 0;
      for (int k = 0; k < h; k++) sum += tmp[i * h + k] * Tv[j * h + k];
      coeff[j * w + i] = (sum + (1 << (s2 - 1))) >> s2;
    }
  return 0;
}

ORC_API int orc_tr_inv(const TCoeff* coeff, Pel* resi, int stride, int w, int h, int trHor, int trVer, int bd)
{
  initTables();
  const int lw = ilog2(w), lh = ilog2(h);
  if (trHor == 3)
  {
    int shift = 15 - bd - ((lw + lh) >> 1), scale = 1;
    if ((lw + lh) & 1) { shift += 7; scale = 181; }                  // ADJ_QUANT_SHIFT
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++)
    {
      const int c = coeff[y * w + x] * scale;
      resi[y * stride + x] = (Pel)(shift >= 0 ? (c + (shift ? 1 << (shift - 1) : 0)) >> shift : c << -shift);
    }
    return 0;
  }
  const int s1 = 6 + 1 + 2, s2 = (6 + 15 - 1) - bd + 2;             // :253-254
  const int cmin = -(1 << 15), cmax = (1 << 15) - 1;
  const int skipW = w > 32 ? w - 32 : 0, skipH = h > 32 ? h - 32 : 0;
  const int16_t* Th = g_tr[trHor][lw]; const int16_t* Tv = g_tr[trVer][lh];
  std::vector<int> tmp((size_t)w * h, 0);
  for (int i = 0; i < w - skipW; i++)                                // vertical first: tmp[i*h + j]
    for (int j = 0; j < h; j++)
    {
      int sum = 0;
      for (int k = 0; k < h - skipH; k++) sum += coeff[k * w + i] * Tv[k * h + j];
      tmp[i * h + j] = clip3i(cmin, cmax, (sum + (1 << (s1 - 1))) >> s1);
    }
  for (int i = 0; i < h; i++)                                        // horizontal: block[i*w + j]
    for (int j = 0; j < w; j++)
    {
      int sum = 0;
      for (int k = 0; k < w - skipW; k++) sum += tmp[k * h + i] * Th[k * w + j];
      resi[i * stride + j] = (Pel)clip3i(cmin, cmax, (sum + (1 << (s2 - 1))) >> s2);
    }
  return 0;
}

ORC_API int orc_tr_fwd_batch(const Pel* resiBase, TCoeff* coeffBase, const vvcgpu_tr_desc* d, int n, int bd)
{
  for (int i = 0; i < n; i++) orc_tr_fwd(resiBase + d[i].resi_off, d[i].resi_stride, coeffBase + d[i].coeff_off, d[i].w, d[i].h, d[i].tr_hor, d[i].tr_ver, bd);
  return 0;
}
ORC_API int orc_tr_inv_batch(const TCoeff* coeffBase, Pel* resiBase, const vvcgpu_tr_desc* d, int n, int bd)
{
  for (int i = 0; i < n; i++) orc_tr_inv(coeffBase + d[i].coeff_off, resiBase + d[i].resi_off, d[i].resi_stride, d[i].w, d[i].h, d[i].tr_hor, d[i].tr_ver, bd);
  return 0;
}
