// mfma_tr.h -- the four 1-D transform stages of a W x H TU (W, H in {16, 32, 64}) on the matrix cores, one wave per TU.
//
// Shared by the fused residual chain (resichain.hip) and the standalone transform entries (transform.hip).  Reference behaviour: xTrMxN_EMT /
// xITrMxN_EMT (CommonLib/TrQuant.cpp:138-310) as integer matrix products with the reference's tables.
//
// v_mfma_f32_16x16x32_f16 accumulates in f32, which is exact for integers below 2^24: the matrix entries (|c| <= 362... 90 for DCT-II, all exact
// f16 values) times operands of at most 11 bits over 64 terms stay below 2^24; 16-bit operands are split into two signed 8-bit limbs
// (t = 256 hi + lo), one MFMA chain per limb, recombined in int32 with the reference's rounding shift and clipping.  The result tile of one
// stage is the operand of the next WITHOUT leaving the lane: a 16x16 result has its column on the lane and four consecutive rows in registers,
// the next product sums over that row index, and the k order of an MFMA is free as long as both operands agree -- so the matrix operand is read
// from LDS in the k order the result registers already have.
//   forward:  M1 = X Th^T  (H x WJ),  C = Tv M1 (HJ x WJ)         inverse:  Y1^T = Cq^T Tv (WJ x H),  R^T = Th^T Y1^T (W x H)
//   WJ = min(W, 32), HJ = min(H, 32): the zero-out of the frequencies >= 32 (TrQuant.cpp:157-162, :755-759)
#pragma once
#include "common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

// ---- f16 copies of the matrices in LDS.  Per type t and size n in {16, 32}: T (row-major T[j][k]) and its transpose; for 64 DCT-II only.
// Rows are padded by 16 bytes, which spreads the 16 rows read by one ds_read_b64 / b128 over all banks.
constexpr int RC_S16 = 16 * 24, RC_S32 = 32 * 40, RC_S64 = 64 * 72;                  // halves per matrix copy (row pitch n + 8)
constexpr int RC_TYPE = 2 * RC_S16 + 2 * RC_S32;
// behind them the 4- and 8-point matrices (per type: T4, T4^T, T8, T8^T, rows unpadded) of the packed-tile form (resichain.hip)
constexpr int RC_SMALL_OFF = 3 * RC_TYPE + 2 * RC_S64, RC_SMALL_TYPE = 2 * 16 + 2 * 64;
constexpr int RC_TAB_HALVES = RC_SMALL_OFF + 3 * RC_SMALL_TYPE;
static_assert(RC_TAB_HALVES % 8 == 0 && RC_SMALL_OFF % 8 == 0, "the image is copied with 16-byte loads");
__device__ __forceinline__ int rc_small_off(int type, int n, int transposed) { return RC_SMALL_OFF + type * RC_SMALL_TYPE + (n == 4 ? transposed * 16 : 32 + transposed * 64); }
__device__ __forceinline__ int rc_tab_off(int type, int n, int transposed)
{
  if (n == 64) return 3 * RC_TYPE + transposed * RC_S64;
  return type * RC_TYPE + (n == 16 ? transposed * RC_S16 : 2 * RC_S16 + transposed * RC_S32);
}
// the image (built once per device in global memory, resichain.hip) as a device pointer; nullptr + error text on failure
const _Float16* vvcgpu_mfma_image(const VvcTrTables& tb);

// copies the matrices a TU size needs: sizes 16 / 32: the T and T^T copies of that size for the three types; 64: the DCT-II pair
template <int N>
__device__ __forceinline__ void rc_load_tables(_Float16* tab, const _Float16* __restrict__ image, int tid)
{
  constexpr int SZ = N == 16 ? 2 * RC_S16 : N == 32 ? 2 * RC_S32 : 2 * RC_S64;  // halves per type (T and T^T are adjacent)
  constexpr int NT = N == 64 ? 1 : 3, NV = SZ / 8;
#pragma unroll
  for (int t = 0; t < NT; t++)
  {
    const int off = rc_tab_off(t, N, 0);
    const uint4* src = reinterpret_cast<const uint4*>(image + off);
    uint4* dst = reinterpret_cast<uint4*>(tab + off);
    uint4 v[(NV + 255) / 256];
#pragma unroll
    for (int u = 0; u < (NV + 255) / 256; u++) if (tid + 256 * u < NV) v[u] = src[tid + 256 * u];
#pragma unroll
    for (int u = 0; u < (NV + 255) / 256; u++) if (tid + 256 * u < NV) dst[tid + 256 * u] = v[u];
  }
}
__device__ __forceinline__ void rc_load_small_tables(_Float16* tab, const _Float16* __restrict__ image, int tid)
{
  constexpr int NV = 3 * RC_SMALL_TYPE / 8;
  if (tid < NV) reinterpret_cast<uint4*>(tab + RC_SMALL_OFF)[tid] = reinterpret_cast<const uint4*>(image + RC_SMALL_OFF)[tid];
}
__device__ __forceinline__ void rc_load_all_tables(_Float16* tab, const _Float16* __restrict__ image, int tid)
{
  rc_load_tables<16>(tab, image, tid);
  rc_load_tables<32>(tab, image, tid);
  rc_load_tables<64>(tab, image, tid);
}

// matrix operand of a product whose OTHER operand is a result tile: row `row` of the LDS matrix, the eight k values of k-step s in result-tile
// order -- k = 32 s + 4 g + j (j < 4, tile 2 s) and 32 s + 16 + 4 g + (j - 4) (tile 2 s + 1)
__device__ __forceinline__ h8 rc_mat_frag32(const _Float16* mat, int pitch, int row, int s, int g)
{
  const h4 a = *reinterpret_cast<const h4*>(mat + row * pitch + 32 * s + 4 * g);
  const h4 b = *reinterpret_cast<const h4*>(mat + row * pitch + 32 * s + 16 + 4 * g);
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ h4 rc_mat_frag16(const _Float16* mat, int pitch, int row, int g)
{
  return *reinterpret_cast<const h4*>(mat + row * pitch + 4 * g);
}

// 16-bit signed integer -> two signed 8-bit limbs as f16 (v = 256 hi + lo, lo in [-128, 127], hi in [-128, 128])
__device__ __forceinline__ void rc_limbs(int v, _Float16& hi, _Float16& lo)
{
  const int l = (int)(signed char)v;
  lo = (_Float16)(short)l;
  hi = (_Float16)(short)((v - l) >> 8);
}

// K-step bookkeeping of a product with inner dimension KD: one 16x16x16 step for KD = 16, KD / 32 steps of 16x16x32 otherwise
template <int KD> struct RcK { static constexpr int STEPS = KD == 16 ? 1 : KD / 32; };

// D += A B for one 16x16 tile over all k-steps; operands as fragment arrays per k-step (h8) or one h4 when the inner dimension is 16
template <int KD>
__device__ __forceinline__ f4 rc_mma(const h8 (&a)[RcK<KD>::STEPS], const h8 (&b)[RcK<KD>::STEPS], f4 acc)
{
  if (KD == 16)
  {
    const h4 a4 = __builtin_shufflevector(a[0], a[0], 0, 1, 2, 3), b4 = __builtin_shufflevector(b[0], b[0], 0, 1, 2, 3);
    return __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc, 0, 0, 0);
  }
#pragma unroll
  for (int s = 0; s < RcK<KD>::STEPS; s++) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[s], b[s], acc, 0, 0, 0);
  return acc;
}

// fragments (per k-step) of row `row` of an LDS matrix with row pitch `pitch`, the first KD values of the row, in result-tile k order
template <int KD>
__device__ __forceinline__ void rc_mat_frags(h8 (&f)[RcK<KD>::STEPS], const _Float16* mat, int pitch, int row, int g)
{
  if (KD == 16)
  {
    const h4 a = rc_mat_frag16(mat, pitch, row, g);
    f[0] = __builtin_shufflevector(a, a, 0, 1, 2, 3, 0, 1, 2, 3);
  }
  else
  {
#pragma unroll
    for (int s = 0; s < RcK<KD>::STEPS; s++) f[s] = rc_mat_frag32(mat, pitch, row, s, g);
  }
}

// fragments of a RESULT-derived operand: tiles t[0 .. KD/16) (four registers each: rows 4 g .. 4 g + 3 of tile), one limb (hi or lo) of each
template <int KD>
__device__ __forceinline__ void rc_tile_frags(h8 (&fh)[RcK<KD>::STEPS], h8 (&fl)[RcK<KD>::STEPS], const int (*t)[4])
{
  if (KD == 16)
  {
    _Float16 h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; j++) rc_limbs(t[0][j], h[j], l[j]);
    fh[0] = h8{ h[0], h[1], h[2], h[3], h[0], h[1], h[2], h[3] };
    fl[0] = h8{ l[0], l[1], l[2], l[3], l[0], l[1], l[2], l[3] };
  }
  else
  {
#pragma unroll
    for (int s = 0; s < RcK<KD>::STEPS; s++)
    {
      _Float16 h[8], l[8];
#pragma unroll
      for (int j = 0; j < 4; j++) { rc_limbs(t[2 * s][j], h[j], l[j]); rc_limbs(t[2 * s + 1][j], h[4 + j], l[4 + j]); }
      fh[s] = h8{ h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7] };
      fl[s] = h8{ l[0], l[1], l[2], l[3], l[4], l[5], l[6], l[7] };
    }
  }
}

// tile counts of a W x H TU
template <int W, int H> struct MtShape
{
  static constexpr int WJ = W > 32 ? 32 : W, HJ = H > 32 ? 32 : H;   // kept frequencies per dimension
  static constexpr int RT = H / 16, CT = W / 16;                     // tiles along the sample rows / columns
  static constexpr int JT = WJ / 16, IT = HJ / 16;                   // tiles along the kept horizontal / vertical frequencies
  static constexpr int XS = W == 16 ? 1 : W / 32;                    // k-steps of the first forward stage
};

// ---- forward stage 1 (horizontal): M1[r][j1] = sum_k X[r][k] Th[j1][k], rounded: t1[jt][rt][reg] = row 16 rt + 4 g + reg, frequency 16 jt + c.
// x[rt][s]: the lane's residual samples of row 16 rt + c: columns 32 s + 8 g .. + 7 (W >= 32) or 4 g .. 4 g + 3 in the low half (W = 16);
// |x| <= 1023 is the caller's business (exact f16, row sums below 2^24).
template <int W, int H>
__device__ __forceinline__ void mt_fwd1(int (&t1)[(MtShape<W, H>::JT)][(MtShape<W, H>::RT)][4], const h8 (&x)[(MtShape<W, H>::RT)][(MtShape<W, H>::XS)],
                                        const _Float16* Th, int s1, int c, int g)
{
  typedef MtShape<W, H> S;
  f4 m1[S::RT][S::JT];
#pragma unroll
  for (int rt = 0; rt < S::RT; rt++)
#pragma unroll
    for (int jt = 0; jt < S::JT; jt++) m1[rt][jt] = f4{ 0.f, 0.f, 0.f, 0.f };
  if (W == 16)
  {
    const h4 b = *reinterpret_cast<const h4*>(Th + c * 24 + 4 * g);
#pragma unroll
    for (int rt = 0; rt < S::RT; rt++)
      m1[rt][0] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_shufflevector(x[rt][0], x[rt][0], 0, 1, 2, 3), b, m1[rt][0], 0, 0, 0);
  }
  else
  {
#pragma unroll
    for (int s = 0; s < S::XS; s++)
    {
      h8 b[S::JT];
#pragma unroll
      for (int jt = 0; jt < S::JT; jt++) b[jt] = *reinterpret_cast<const h8*>(Th + (16 * jt + c) * (W + 8) + 32 * s + 8 * g);
#pragma unroll
      for (int rt = 0; rt < S::RT; rt++)
#pragma unroll
        for (int jt = 0; jt < S::JT; jt++) m1[rt][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x[rt][s], b[jt], m1[rt][jt], 0, 0, 0);
    }
  }
#pragma unroll
  for (int rt = 0; rt < S::RT; rt++)
#pragma unroll
    for (int jt = 0; jt < S::JT; jt++)
#pragma unroll
      for (int r = 0; r < 4; r++) t1[jt][rt][r] = ((int)m1[rt][jt][r] + (1 << (s1 - 1))) >> s1;
}

// ---- forward stage 2 (vertical): C[j2][j1] = sum_r Tv[j2][r] M1[r][j1], rounded: cf[it][jt][reg] = vertical frequency 16 it + 4 g + reg,
// horizontal frequency 16 jt + c   (A = Tv rows from LDS in result-tile k order, B = M1 limbs)
template <int W, int H>
__device__ __forceinline__ void mt_fwd2(int (&cf)[(MtShape<W, H>::IT)][(MtShape<W, H>::JT)][4], const int (&t1)[(MtShape<W, H>::JT)][(MtShape<W, H>::RT)][4],
                                        const _Float16* Tv, int s2, int c, int g)
{
  typedef MtShape<W, H> S;
  h8 bh[S::JT][RcK<H>::STEPS], bl[S::JT][RcK<H>::STEPS];
#pragma unroll
  for (int jt = 0; jt < S::JT; jt++) rc_tile_frags<H>(bh[jt], bl[jt], t1[jt]);
#pragma unroll
  for (int it = 0; it < S::IT; it++)
  {
    h8 a[RcK<H>::STEPS];
    rc_mat_frags<H>(a, Tv, H + 8, 16 * it + c, g);
#pragma unroll
    for (int jt = 0; jt < S::JT; jt++)
    {
      const f4 hi = rc_mma<H>(a, bh[jt], f4{ 0.f, 0.f, 0.f, 0.f }), lo = rc_mma<H>(a, bl[jt], f4{ 0.f, 0.f, 0.f, 0.f });
#pragma unroll
      for (int r = 0; r < 4; r++) cf[it][jt][r] = ((((int)hi[r]) << 8) + (int)lo[r] + (1 << (s2 - 1))) >> s2;
    }
  }
}

// ---- inverse stage 1 (vertical): Y1T[i][r] = sum_k Cq[k][i] Tv[k][r], k < HJ, clipped to 16 bits: y1[rt][jt][reg] = horizontal frequency
// 16 jt + 4 g + reg, sample row 16 rt + c.  cq[jt][it][reg] = coefficient (16-bit) of vertical frequency 16 it + 4 g + reg, horizontal frequency
// 16 jt + c -- the forward result tile read as X^T (A operand: row = its column c, k = its rows); B = rows of Tv^T from LDS.
template <int W, int H>
__device__ __forceinline__ void mt_inv1(int (&y1)[(MtShape<W, H>::RT)][(MtShape<W, H>::JT)][4], const int (&cq)[(MtShape<W, H>::JT)][(MtShape<W, H>::IT)][4],
                                        const _Float16* TvT, int c, int g)
{
  typedef MtShape<W, H> S;
  h8 ah[S::JT][RcK<S::HJ>::STEPS], al[S::JT][RcK<S::HJ>::STEPS];
#pragma unroll
  for (int jt = 0; jt < S::JT; jt++) rc_tile_frags<S::HJ>(ah[jt], al[jt], cq[jt]);
#pragma unroll
  for (int rt = 0; rt < S::RT; rt++)
  {
    h8 b[RcK<S::HJ>::STEPS];
    rc_mat_frags<S::HJ>(b, TvT, H + 8, 16 * rt + c, g);                 // rows of Tv^T have pitch H + 8; only k < HJ is read
#pragma unroll
    for (int jt = 0; jt < S::JT; jt++)
    {
      const f4 hi = rc_mma<S::HJ>(ah[jt], b, f4{ 0.f, 0.f, 0.f, 0.f }), lo = rc_mma<S::HJ>(al[jt], b, f4{ 0.f, 0.f, 0.f, 0.f });
#pragma unroll
      for (int r = 0; r < 4; r++) y1[rt][jt][r] = clip3(-(1 << 15), (1 << 15) - 1, ((((int)hi[r]) << 8) + (int)lo[r] + 256) >> 9);
    }
  }
}

// ---- inverse stage 2 (horizontal): RT[x][r] = sum_i Th[i][x] Y1T[i][r], i < WJ (A = rows of Th^T from LDS, B = Y1T limbs).  The result tile
// (xt, rt) holds the residual of row 16 rt + c, columns 16 xt + 4 g .. + 3 (four consecutive samples of one row per lane), rounded by s2 and
// clipped to 16 bits; `emit(rt, xt, resi[4])` receives it.
template <int W, int H, class Emit>
__device__ __forceinline__ void mt_inv2(const int (&y1)[(MtShape<W, H>::RT)][(MtShape<W, H>::JT)][4], const _Float16* ThT, int s2, int c, int g, Emit emit)
{
  typedef MtShape<W, H> S;
  h8 bh[S::RT][RcK<S::WJ>::STEPS], bl[S::RT][RcK<S::WJ>::STEPS];
#pragma unroll
  for (int rt = 0; rt < S::RT; rt++) rc_tile_frags<S::WJ>(bh[rt], bl[rt], y1[rt]);
#pragma unroll
  for (int xt = 0; xt < S::CT; xt++)
  {
    h8 a[RcK<S::WJ>::STEPS];
    rc_mat_frags<S::WJ>(a, ThT, W + 8, 16 * xt + c, g);
#pragma unroll
    for (int rt = 0; rt < S::RT; rt++)
    {
      const f4 hi = rc_mma<S::WJ>(a, bh[rt], f4{ 0.f, 0.f, 0.f, 0.f }), lo = rc_mma<S::WJ>(a, bl[rt], f4{ 0.f, 0.f, 0.f, 0.f });
      int resi[4];
#pragma unroll
      for (int r = 0; r < 4; r++) resi[r] = clip3(-(1 << 15), (1 << 15) - 1, ((((int)hi[r]) << 8) + (int)lo[r] + (1 << (s2 - 1))) >> s2);
      emit(rt, xt, resi);
    }
  }
}
