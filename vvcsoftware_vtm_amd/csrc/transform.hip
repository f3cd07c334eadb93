// transform.hip -- 2-D separable integer transforms (T1 forward, T2 inverse, T3 transform skip) for gfx950.
//
// Reference behaviour reproduced (bit-exact): xTrMxN_EMT / xITrMxN_EMT (CommonLib/TrQuant.cpp:138-310) with the 1-D
// stages of TrQuant_EMT.cpp expressed as integer matrix products with the reference's own tables (tr_tables.inc, dumped
// from initROM(); tests/golden/gen_tr_tables.py proves fast transform == table for every slot), intermediate rounding
// `(sum + rnd) >> shift` between the stages (:214-215), inverse stages clipped to [-2^15, 2^15-1] (:253-256), zero-out of
// columns/rows >= 32 (:157-162, :755-759); xTransformSkip / xITransformSkip (:795-847, :1112-1163).
//
// Design: both 1-D stages of a TU run inside one wave with the intermediate in LDS (the reference's alloca'd `tmp`,
// never in HBM); accumulation is exact int32.  A batch is served by two launches:
//   * LARGE TUs (a side of 32 or 64): one wave per TU, lane = row / column, the matrix row of the current output index
//     is wave-uniform (scalar loads); persistent grid-stride waves that skip the other descriptors.
//   * SMALL TUs (both sides <= 16) and transform skip: a 4x4 TU would leave 60 of 64 lanes idle and, worse, a launch
//     of one workgroup per TU is bound by the dispatcher (~1 workgroup/ns chip-wide: 0.5 ms for a 4K picture of 4x4
//     TUs).  Here a 256-thread workgroup takes 64 consecutive descriptors, bins them by max(w, h) in LDS and gives
//     every TU a group of 4 / 8 / 16 lanes (16 / 8 / 4 TUs per wave); the matrices of sizes <= 16 sit in LDS as int32,
//     each lane group reading the rows of its own TU's transform types (same-address reads broadcast).
#include "common.h"
#include "mfma_tr.h"
#include "tr_tables.inc"
#include <mutex>

// resichain.hip: the chain's bodies as plain transforms (mode 1 forward, 2 inverse): ONE launch with packed matrix-core tiles for long calls
int vvcgpu_tr_chain_launch(int mode, const vvc_pel* resi_in, vvc_pel* resi_out, vvc_coef* coeff, const vvcgpu_tr_desc* descs, int n, int bit_depth, void* stream);

namespace {

// int32 copies of the golden matrices: d_tr32[type][size] row-major (T[k][n]) and d_tr32t (transposed, T[n][k]) so that the
// row a wave needs is always contiguous -> scalar (SGPR) loads.  Layout as VVC_TR_TABLES: size N at (N*N-4)/3.
__device__ int d_tr32[3 * 5460];
__device__ int d_tr32t[3 * 5460];

__device__ __forceinline__ const int* tr32(int type, int n)  { return d_tr32  + type * 5460 + (n * n - 4) / 3; }
__device__ __forceinline__ const int* tr32t(int type, int n) { return d_tr32t + type * 5460 + (n * n - 4) / 3; }
__device__ __forceinline__ int ilog2(int v) { return 31 - __clz(v); }

constexpr int MAXN = 64;
// address-space-qualified views: behind a real call a plain pointer is a FLAT pointer -- every LDS access becomes a flat_load / flat_store that the
// compiler can neither batch nor reorder against the global loads beside it (measured: the staging loop alone ran one memory round trip per element)
typedef __attribute__((address_space(3))) int LdsInt;
typedef const __attribute__((address_space(1))) int GlbCInt;
typedef __attribute__((address_space(1))) int GlbInt;
typedef __attribute__((address_space(1))) short GlbPel;

typedef short short2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ bool is_large_tu(const vvcgpu_tr_desc& d) { return d.tr_hor != 3 && (d.w > 16 || d.h > 16); }

// ---------------------------------------------------------------------------------------------------
// Small TUs (w, h <= 16) and transform skip: 64 descriptors per 256-thread workgroup, binned in LDS.
constexpr int SM_DESCS = 64, SM_TAB = 376;          // per type: size 4 at 0, 8 at 16, 16 at 80, 2 at 336; padded for over-reads
__device__ __forceinline__ int small_off(int n) { return n == 4 ? 0 : n == 8 ? 16 : n == 16 ? 80 : 336; }
struct SmallShared
{
  vvcgpu_tr_desc d[SM_DESCS];
  int tab[3][SM_TAB];                               // T[k][n]   (forward: row = output index)
  int tabT[3][SM_TAB];                              // T^T       (inverse: row = output index)
  int tmp[4][4 * 16 * 17];                          // per wave: P TUs x S x (S+1)
  int cnt[4];                                       // bins: 0 transform skip, 1 S <= 4, 2 S = 8, 3 S = 16
  unsigned char list[4][SM_DESCS];
};

__device__ __forceinline__ void small_tables(SmallShared& sh, int tid)
{
  for (int i = tid; i < 3 * SM_TAB; i += 256)
  {
    const int t = i / SM_TAB, e = i - t * SM_TAB;
    int nsz = 0, o = 0;
    if (e < 16) { nsz = 4; o = e; } else if (e < 80) { nsz = 8; o = e - 16; } else if (e < 336) { nsz = 16; o = e - 80; } else if (e < 340) { nsz = 2; o = e - 336; }
    sh.tab[t][e] = nsz ? tr32(t, nsz)[o] : 0;
    sh.tabT[t][e] = nsz ? tr32t(t, nsz)[o] : 0;
  }
}
// smCnt / smLists (may be null): the small TUs of the call binned on the device (tr_collect_large_kernel): the batch is then a run of the
// concatenated bin lists (lane groups of 16 first), so that a workgroup's TUs share a bin and its lane groups are full -- in the caller's order a real
// encoder's call mix leaves most groups part-filled (profiles/r04_entry_shapes.txt: the mixed batch took 1.5 x the time of its parts)
__device__ __forceinline__ int small_total(const int* __restrict__ smCnt, int n) { return smCnt ? smCnt[0] + smCnt[1] + smCnt[2] + smCnt[3] : n; }
__device__ __forceinline__ void small_setup(SmallShared& sh, const vvcgpu_tr_desc* __restrict__ descs, int n, int batch, int tid, int useMfma,
                                            const int* __restrict__ smCnt = nullptr, const int* __restrict__ smLists = nullptr)
{
  __syncthreads();                                  // tables ready / previous batch done with d, list, cnt
  if (tid < 4) sh.cnt[tid] = 0;
  __syncthreads();
  const int base = batch * SM_DESCS, total = small_total(smCnt, n);
  if (tid < SM_DESCS && base + tid < total)
  {
    int di = base + tid;
    if (smCnt)
    {
      int v = di, b = 3;
#pragma unroll
      for (int k = 3; k > 0; k--) { const int c = smCnt[k]; if (b == k && v >= c) { v -= c; b = k - 1; } }
      di = smLists[(size_t)b * n + v];
    }
    const vvcgpu_tr_desc d = descs[di];
    sh.d[tid] = d;
    const int S = max((int)d.w, (int)d.h);
    int bin = d.tr_hor == 3 ? 0 : S <= 4 ? 1 : S == 8 ? 2 : S == 16 ? 3 : -1;           // -1: large, the other launches
    if (useMfma && bin == 3 && d.w == 16 && d.h == 16) bin = -1;                           // 16 x 16 runs on the matrix cores
    if (bin >= 0) sh.list[bin][atomicAdd(&sh.cnt[bin], 1)] = (unsigned char)tid;
  }
  __syncthreads();
}

#define TR_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

template <int S>
__device__ __forceinline__ void fwd_small_group(SmallShared& sh, int bin, int grp, int lane, int wave, int bd,
                                                const Pel* __restrict__ resiBase, TCoeff* __restrict__ coeffBase)
{
  constexpr int P = 64 / S;
  const int g = lane / S, r = lane % S, li = grp * P + g;
  const bool act = li < sh.cnt[bin];
  const vvcgpu_tr_desc& d = sh.d[sh.list[bin][act ? li : 0]];
  const int w = d.w, h = d.h, lw = ilog2(w), lh = ilog2(h);
  const int s1 = lw + bd + 6 - 15 + 2, s2 = lh + 6 + 2;
  int* tmp = sh.tmp[wave] + g * (S * (S + 1));
  if (act && r < h)                                 // stage 1 (horizontal): lane = row r
  {
    const Pel* row = resiBase + d.resi_off + (size_t)r * d.resi_stride;
    int x[S];
#pragma unroll
    for (int k = 0; k < S; k++) x[k] = k < w ? (int)row[k] : 0;
    const int* T = sh.tab[d.tr_hor] + small_off(w);
    const int rnd = 1 << (s1 - 1);
    for (int j = 0; j < w; j++)
    {
      int sum = 0;
#pragma unroll
      for (int k = 0; k < S; k++) sum += __mul24(x[k], T[j * w + k]);   // k >= w: x[k] = 0, T reads stay inside the padded table
      tmp[j * (S + 1) + r] = (sum + rnd) >> s1;
    }
  }
  TR_WAVE_SYNC();
  if (act && r < w)                                 // stage 2 (vertical): lane = horizontal frequency r
  {
    int t[S];
#pragma unroll
    for (int k = 0; k < S; k++) t[k] = k < h ? tmp[r * (S + 1) + k] : 0;
    const int* T = sh.tab[d.tr_ver] + small_off(h);
    const int rnd = 1 << (s2 - 1);
    TCoeff* coeff = coeffBase + d.coeff_off;
    for (int j = 0; j < h; j++)
    {
      int sum = 0;
#pragma unroll
      for (int k = 0; k < S; k++) sum += __mul24(t[k], T[j * h + k]);
      coeff[j * w + r] = (sum + rnd) >> s2;
    }
  }
  TR_WAVE_SYNC();
}

__global__ __launch_bounds__(256) void tr_fwd_small_kernel(const Pel* __restrict__ resiBase, TCoeff* __restrict__ coeffBase,
                                                           const vvcgpu_tr_desc* __restrict__ descs, int n, int bd, int useMfma,
                                                           const int* __restrict__ smCnt, const int* __restrict__ smLists)
{
  __shared__ SmallShared sh;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  small_tables(sh, tid);
  const int total = small_total(smCnt, n);
  for (int batch = blockIdx.x; batch * SM_DESCS < total; batch += gridDim.x)
  {
  small_setup(sh, descs, n, batch, tid, useMfma, smCnt, smLists);
  for (int q = 0; q < sh.cnt[0]; q++)                // transform skip: element-wise, whole workgroup
  {
    const vvcgpu_tr_desc& d = sh.d[sh.list[0][q]];
    const int w = d.w, h = d.h, lw = ilog2(w), lh = ilog2(h);
    int shift = 15 - bd - ((lw + lh) >> 1), scale = 1;
    if ((lw + lh) & 1) { shift -= 8; scale = 181; }
    const Pel* resi = resiBase + d.resi_off;
    TCoeff* coeff = coeffBase + d.coeff_off;
    for (int i = tid; i < w * h; i += 256)
    {
      const int y = i >> lw, x = i & (w - 1);
      const int v = resi[(size_t)y * d.resi_stride + x] * scale;
      coeff[i] = shift >= 0 ? v << shift : (v + (1 << (-shift - 1))) >> -shift;
    }
  }
  for (int g = wave; g * 16 < sh.cnt[1]; g += 4) fwd_small_group<4>(sh, 1, g, lane, wave, bd, resiBase, coeffBase);
  for (int g = wave; g * 8 < sh.cnt[2]; g += 4)  fwd_small_group<8>(sh, 2, g, lane, wave, bd, resiBase, coeffBase);
  for (int g = wave; g * 4 < sh.cnt[3]; g += 4)  fwd_small_group<16>(sh, 3, g, lane, wave, bd, resiBase, coeffBase);
  }
}

template <int S, bool M24>
__device__ __forceinline__ int inv_dot(const int (&c)[S], const int* T)
{
  int sum = 0;
#pragma unroll
  for (int k = 0; k < S; k++) sum += M24 ? __mul24(c[k], T[k]) : c[k] * T[k];
  return sum;
}

template <int S>
__device__ __forceinline__ void inv_small_group(SmallShared& sh, int bin, int grp, int lane, int wave, int bd,
                                                const TCoeff* __restrict__ coeffBase, Pel* __restrict__ resiBase, const int* ldsCoef = nullptr)
{
  constexpr int P = 64 / S;
  const int g = lane / S, r = lane % S, li = grp * P + g;
  const bool act = li < sh.cnt[bin];
  const vvcgpu_tr_desc& d = sh.d[sh.list[bin][act ? li : 0]];
  const int w = d.w, h = d.h;
  const int s2 = (6 + 15 - 1) - bd + 2;
  int* tmp = sh.tmp[wave] + g * (S * (S + 1));
  {                                                 // stage 1 (vertical): lane = column r
    const bool on = act && r < w;
    int c[S];
    const TCoeff* coeff = ldsCoef ? ldsCoef + g * (S * S) : coeffBase + d.coeff_off;     // fused de-quantiser: the group's TUs sit in LDS, S x S ints each
    bool fits = true;
#pragma unroll
    for (int k = 0; k < S; k++) { c[k] = (on && k < h) ? coeff[k * w + r] : 0; fits = fits && (c[k] >= -(1 << 23)) && (c[k] < (1 << 23)); }
    const int* T = sh.tabT[d.tr_ver] + small_off(h);
    // 24-bit multiplies when every coefficient of the wave allows it (always, for quantiser output); exact 32-bit otherwise
    if (__builtin_amdgcn_ballot_w64(!fits) == 0ull)
    {
      if (on) for (int j = 0; j < h; j++) tmp[r * (S + 1) + j] = clip3(-(1 << 15), (1 << 15) - 1, (inv_dot<S, true>(c, T + j * h) + 256) >> 9);
    }
    else
    {
      if (on) for (int j = 0; j < h; j++) tmp[r * (S + 1) + j] = clip3(-(1 << 15), (1 << 15) - 1, (inv_dot<S, false>(c, T + j * h) + 256) >> 9);
    }
  }
  TR_WAVE_SYNC();
  if (act && r < h)                                 // stage 2 (horizontal): lane = row r; tmp is clipped to 16 bits -> 24-bit multiplies
  {
    int t[S];
#pragma unroll
    for (int k = 0; k < S; k++) t[k] = k < w ? tmp[k * (S + 1) + r] : 0;
    const int* T = sh.tabT[d.tr_hor] + small_off(w);
    const int rnd = 1 << (s2 - 1);
    Pel* row = resiBase + d.resi_off + (size_t)r * d.resi_stride;
    for (int j = 0; j < w; j++) row[j] = (short)clip3(-(1 << 15), (1 << 15) - 1, (inv_dot<S, true>(t, T + j * w) + rnd) >> s2);
  }
  TR_WAVE_SYNC();
}

__global__ __launch_bounds__(256) void tr_inv_small_kernel(const TCoeff* __restrict__ coeffBase, Pel* __restrict__ resiBase,
                                                           const vvcgpu_tr_desc* __restrict__ descs, int n, int bd, int useMfma,
                                                           const int* __restrict__ smCnt, const int* __restrict__ smLists)
{
  __shared__ SmallShared sh;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  small_tables(sh, tid);
  const int total = small_total(smCnt, n);
  for (int batch = blockIdx.x; batch * SM_DESCS < total; batch += gridDim.x)
  {
  small_setup(sh, descs, n, batch, tid, useMfma, smCnt, smLists);
  for (int q = 0; q < sh.cnt[0]; q++)
  {
    const vvcgpu_tr_desc& d = sh.d[sh.list[0][q]];
    const int w = d.w, h = d.h, lw = ilog2(w), lh = ilog2(h);
    int shift = 15 - bd - ((lw + lh) >> 1), scale = 1;
    if ((lw + lh) & 1) { shift += 7; scale = 181; }
    const TCoeff* coeff = coeffBase + d.coeff_off;
    Pel* resi = resiBase + d.resi_off;
    for (int i = tid; i < w * h; i += 256)
    {
      const int y = i >> lw, x = i & (w - 1);
      const int c = coeff[i] * scale;
      resi[(size_t)y * d.resi_stride + x] = (short)(shift >= 0 ? (c + (shift ? 1 << (shift - 1) : 0)) >> shift : c << -shift);
    }
  }
  for (int g = wave; g * 16 < sh.cnt[1]; g += 4) inv_small_group<4>(sh, 1, g, lane, wave, bd, coeffBase, resiBase);
  for (int g = wave; g * 8 < sh.cnt[2]; g += 4)  inv_small_group<8>(sh, 2, g, lane, wave, bd, coeffBase, resiBase);
  for (int g = wave; g * 4 < sh.cnt[3]; g += 4)  inv_small_group<16>(sh, 3, g, lane, wave, bd, coeffBase, resiBase);
  }
}

// ---------------------------------------------------------------------------------------------------
// Large TUs, fast path: 256-thread persistent workgroups, one wave per TU; every matrix (int16, 16-byte aligned rows)
// resident in LDS, 16-bit operands packed in pairs and multiplied with v_dot2_i32_i16 (two exact MACs per instruction),
// the matrix row of the current output index read with broadcast 16-byte LDS loads.  When a dimension has fewer than
// 64 rows / columns the idle lanes take a share of the output indices.  Operands that do not fit 16 bits (never the
// case for residuals / quantiser output) and odd row addresses fall back to the scalar-load stages above.
constexpr int LG_TYPE = 1368;                       // shorts per type: size 2 at 0, 4 at 8, 8 at 24, 16 at 88, 32 at 344
constexpr int LG_TAB = 3 * LG_TYPE + 4096;          // + DCT-II 64 at 3 * LG_TYPE
__device__ __forceinline__ int lg_off(int type, int n)
{
  return n == 64 ? 3 * LG_TYPE : type * LG_TYPE + (n == 2 ? 0 : n == 4 ? 8 : n == 8 ? 24 : n == 16 ? 88 : 344);
}
__device__ __forceinline__ void lg_load(short* tab, const int* __restrict__ src32, int tid)
{
  for (int t = 0; t < 3; t++)
    for (int n = 2; n <= 32; n <<= 1)
      for (int e = tid; e < n * n; e += 256) tab[lg_off(t, n) + e] = (short)src32[t * 5460 + (n * n - 4) / 3 + e];
  for (int e = tid; e < 4096; e += 256) tab[3 * LG_TYPE + e] = (short)src32[1364 + e];
}

// sum_k a[k] * T[k], k < N, a packed in pairs, T = LDS row of int16 (wave-uniform or per lane group)
template <int N>
__device__ __forceinline__ int dot_row(const unsigned (&ap)[(N + 1) / 2], const short* T)
{
  int sum = 0;
  if (N >= 8)
  {
    const uint4* tr = reinterpret_cast<const uint4*>(T);
#pragma unroll
    for (int q = 0; q < N / 8; q++)
    {
      const uint4 t = tr[q];
      sum = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, ap[4 * q]), __builtin_bit_cast(short2v, t.x), sum, false);
      sum = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, ap[4 * q + 1]), __builtin_bit_cast(short2v, t.y), sum, false);
      sum = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, ap[4 * q + 2]), __builtin_bit_cast(short2v, t.z), sum, false);
      sum = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, ap[4 * q + 3]), __builtin_bit_cast(short2v, t.w), sum, false);
    }
  }
  else
  {
    const unsigned* tr = reinterpret_cast<const unsigned*>(T);
#pragma unroll
    for (int q = 0; q < N / 2; q++) sum = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, ap[q]), __builtin_bit_cast(short2v, tr[q]), sum, false);
  }
  return sum;
}
__device__ __forceinline__ unsigned pack16(int lo, int hi) { return ((unsigned)lo & 0xFFFFu) | ((unsigned)hi << 16); }
__device__ __forceinline__ bool fits16(int v) { return v == (int)(short)v; }

// forward stage 1: lane = row; returns (wave-uniform) whether every intermediate fits 16 bits
template <int W>
__device__ __forceinline__ bool fwd_stage1_fast(const Pel* __restrict__ resi, int stride, int h, int lane, int s1, const short* T,
                                                int* __restrict__ tmpL, int ph)
{
  constexpr int WJ = W > 32 ? 32 : W;
  bool ok = true;
  if (lane < h)
  {
    const unsigned* row = reinterpret_cast<const unsigned*>(resi + (size_t)lane * stride);
    unsigned xp[W / 2];
#pragma unroll
    for (int m = 0; m < W / 2; m++) xp[m] = row[m];
    const int rnd = 1 << (s1 - 1);
#pragma unroll 2
    for (int j = 0; j < WJ; j++)
    {
      const int v = (dot_row<W>(xp, T + j * W) + rnd) >> s1;
      tmpL[j * ph + lane] = v;
      ok = ok && fits16(v);
    }
  }
  return __builtin_amdgcn_ballot_w64(!ok) == 0ull;
}
// forward stage 2: lane = (horizontal frequency i < wj, share g of the output rows)
template <int H>
__device__ __forceinline__ void fwd_stage2_fast(const int* __restrict__ tmpL, int ph, int w, int wj, int lane, int s2, const short* T,
                                                TCoeff* __restrict__ coeff)
{
  constexpr int HJ = H > 32 ? 32 : H;
  const int i = lane & (wj - 1), g = lane / wj, G = 64 / wj, jPer = HJ / G;     // wj < 32 only with h >= 32: HJ = 32 >= G
  unsigned tp[H / 2];
#pragma unroll
  for (int m = 0; m < H / 2; m++) tp[m] = pack16(tmpL[i * ph + 2 * m], tmpL[i * ph + 2 * m + 1]);
  const int rnd = 1 << (s2 - 1);
#pragma unroll 2
  for (int jj = 0; jj < jPer; jj++)
  {
    const int j = g * jPer + jj;
    coeff[j * w + i] = (dot_row<H>(tp, T + j * H) + rnd) >> s2;
  }
}
// Slow generic stages for the cases the packed 16-bit path cannot take (rows at odd addresses, operands beyond 16 bits:
// neither occurs for encoder residuals / quantiser output).  Plain loops, exact 32-bit arithmetic, no register arrays.
__device__ __noinline__ void fwd_stage1_slow(const Pel* __restrict__ resi, int stride, int w, int h, int lane, int wj, int s1,
                                             const short* T, int* __restrict__ tmpL, int ph)
{
  if (lane >= h) return;
  const Pel* row = resi + (size_t)lane * stride;
  const int rnd = 1 << (s1 - 1);
#pragma unroll 1
  for (int j = 0; j < wj; j++)
  {
    int sum = 0;
#pragma unroll 1
    for (int k = 0; k < w; k++) sum += (int)row[k] * (int)T[j * w + k];
    tmpL[j * ph + lane] = (sum + rnd) >> s1;
  }
}
__device__ __noinline__ void fwd_stage2_slow(const int* __restrict__ tmpL, int ph, int w, int h, int wj, int hj, int lane, int s2,
                                             const short* T, TCoeff* __restrict__ coeff)
{
  if (lane >= w) return;
  const int rnd = 1 << (s2 - 1);
#pragma unroll 1
  for (int j = 0; j < h; j++)
  {
    int v = 0;
    if (lane < wj && j < hj)
    {
      int sum = 0;
#pragma unroll 1
      for (int k = 0; k < h; k++) sum += tmpL[lane * ph + k] * (int)T[j * h + k];
      v = (sum + rnd) >> s2;
    }
    coeff[j * w + lane] = v;
  }
}
__device__ __noinline__ void inv_stage1_slow(const TCoeff* __restrict__ coeff, int w, int h, int wj, int hj, int lane, const short* TT,
                                             int* __restrict__ tmpL, int ph)
{
  if (lane >= wj) return;
#pragma unroll 1
  for (int j = 0; j < h; j++)
  {
    int sum = 0;
#pragma unroll 1
    for (int k = 0; k < hj; k++) sum += coeff[k * w + lane] * (int)TT[j * h + k];
    tmpL[lane * ph + j] = clip3(-(1 << 15), (1 << 15) - 1, (sum + 256) >> 9);
  }
}

template <int W>
__device__ __forceinline__ void fwd_tu_large(const vvcgpu_tr_desc& d, const Pel* resi, TCoeff* coeff, int bd, int lane, int* tmpL,
                                             const short* tab)
{
  const int h = d.h, lw = ilog2(W), lh = ilog2(h);
  const int s1 = lw + bd + 6 - 15 + 2, s2 = lh + 6 + 2;
  const int wj = W > 32 ? 32 : W, hj = h > 32 ? 32 : h;
  const int ph = h + 1;
  const bool aligned = (((uintptr_t)resi | (uintptr_t)(d.resi_stride * 2)) & 3) == 0;
  bool fast = false;
  if (aligned) fast = fwd_stage1_fast<W>(resi, d.resi_stride, h, lane, s1, tab + lg_off(d.tr_hor, W), tmpL, ph);
  else fwd_stage1_slow(resi, d.resi_stride, W, h, lane, wj, s1, tab + lg_off(d.tr_hor, W), tmpL, ph);
  TR_WAVE_SYNC();
  if (fast)
  {
    const short* Tv = tab + lg_off(d.tr_ver, h);
    switch (h)
    {
    case 2:  fwd_stage2_fast<2>(tmpL, ph, W, wj, lane, s2, Tv, coeff); break;
    case 4:  fwd_stage2_fast<4>(tmpL, ph, W, wj, lane, s2, Tv, coeff); break;
    case 8:  fwd_stage2_fast<8>(tmpL, ph, W, wj, lane, s2, Tv, coeff); break;
    case 16: fwd_stage2_fast<16>(tmpL, ph, W, wj, lane, s2, Tv, coeff); break;
    case 32: fwd_stage2_fast<32>(tmpL, ph, W, wj, lane, s2, Tv, coeff); break;
    default: fwd_stage2_fast<64>(tmpL, ph, W, wj, lane, s2, Tv, coeff); break;
    }
    // zero-out: rows >= hj (contiguous) and columns >= wj of the kept rows
    for (int e = hj * W + lane; e < h * W; e += 64) coeff[e] = 0;
    if (W > 32) for (int e = lane; e < hj * 32; e += 64) coeff[(e >> 5) * W + 32 + (e & 31)] = 0;
  }
  else fwd_stage2_slow(tmpL, ph, W, h, wj, hj, lane, s2, tab + lg_off(d.tr_ver, h), coeff);
  TR_WAVE_SYNC();
}

__global__ __launch_bounds__(256, 3) void tr_fwd_large_kernel(const Pel* __restrict__ resiBase, TCoeff* __restrict__ coeffBase,
                                                           const vvcgpu_tr_desc* __restrict__ descs, const int* __restrict__ list, int bd)
{
  __shared__ __align__(16) short tab[LG_TAB];
  __shared__ int tmpAll[4][32 * (MAXN + 1)];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cnt = list[0];
  if ((int)blockIdx.x * 4 >= cnt) return;
  lg_load(tab, d_tr32, tid);
  __syncthreads();
  int* tmpL = tmpAll[wave];
  for (int k = blockIdx.x * 4 + wave; k < cnt; k += gridDim.x * 4)
  {
    const vvcgpu_tr_desc d = descs[__builtin_amdgcn_readfirstlane(list[1 + k])];
    const Pel* resi = resiBase + d.resi_off;
    TCoeff* coeff = coeffBase + d.coeff_off;
    switch (d.w)
    {
    case 2:  fwd_tu_large<2>(d, resi, coeff, bd, lane, tmpL, tab); break;
    case 4:  fwd_tu_large<4>(d, resi, coeff, bd, lane, tmpL, tab); break;
    case 8:  fwd_tu_large<8>(d, resi, coeff, bd, lane, tmpL, tab); break;
    case 16: fwd_tu_large<16>(d, resi, coeff, bd, lane, tmpL, tab); break;
    case 32: fwd_tu_large<32>(d, resi, coeff, bd, lane, tmpL, tab); break;
    default: fwd_tu_large<64>(d, resi, coeff, bd, lane, tmpL, tab); break;
    }
  }
}

// inverse stage 1 (vertical): lane = (kept column i < wj, share g of the output rows); false when a coefficient needs > 16 bits
template <int H>
__device__ __forceinline__ bool inv_stage1_fast(const TCoeff* __restrict__ coeff, int w, int wj, int lane, const short* TT,
                                                int* __restrict__ tmpL, int ph)
{
  constexpr int HJ = H > 32 ? 32 : H;
  const int i = lane & (wj - 1), g = lane / wj, G = 64 / wj, jPer = H / G;
  int c[HJ];
  bool ok = true;
#pragma unroll
  for (int k = 0; k < HJ; k++) { c[k] = coeff[k * w + i]; ok = ok && fits16(c[k]); }
  if (__builtin_amdgcn_ballot_w64(!ok) != 0ull) return false;
  unsigned cp[(HJ + 1) / 2];
#pragma unroll
  for (int m = 0; m < HJ / 2; m++) cp[m] = pack16(c[2 * m], c[2 * m + 1]);
#pragma unroll 2
  for (int jj = 0; jj < jPer; jj++)
  {
    const int j = g * jPer + jj;
    tmpL[i * ph + j] = clip3(-(1 << 15), (1 << 15) - 1, (dot_row<HJ>(cp, TT + j * H) + 256) >> 9);
  }
  return true;
}
// inverse stage 2 (horizontal): lane = (row r < h, share g of the output columns); tmp is clipped to 16 bits by stage 1
template <int W>
__device__ __forceinline__ void inv_stage2_fast(const int* __restrict__ tmpL, int ph, int h, int lane, int s2, const short* TT,
                                                Pel* __restrict__ resi, int stride)
{
  constexpr int WJ = W > 32 ? 32 : W;
  const int r = lane & (h - 1), g = lane / h, G = 64 / h, jPer = W / G;
  unsigned tp[(WJ + 1) / 2];
#pragma unroll
  for (int m = 0; m < WJ / 2; m++) tp[m] = pack16(tmpL[(2 * m) * ph + r], tmpL[(2 * m + 1) * ph + r]);
  const int rnd = 1 << (s2 - 1);
  Pel* row = resi + (size_t)r * stride;
#pragma unroll 2
  for (int jj = 0; jj < jPer; jj++)
  {
    const int j = g * jPer + jj;
    row[j] = (short)clip3(-(1 << 15), (1 << 15) - 1, (dot_row<WJ>(tp, TT + j * W) + rnd) >> s2);
  }
}
template <int W>
__device__ __forceinline__ void inv_tu_large(const vvcgpu_tr_desc& d, const TCoeff* coeff, Pel* resi, int bd, int lane, int* tmpL,
                                             const short* tabT, int pitch = W)            // pitch: row pitch of `coeff` (the fused de-quantiser keeps wj)
{
  const int h = d.h;
  const int s2 = (6 + 15 - 1) - bd + 2;
  const int wj = W > 32 ? 32 : W;
  const int ph = h + 1;
  const short* TvT = tabT + lg_off(d.tr_ver, h);
  bool fast;
  switch (h)
  {
  case 2:  fast = inv_stage1_fast<2>(coeff, pitch, wj, lane, TvT, tmpL, ph); break;
  case 4:  fast = inv_stage1_fast<4>(coeff, pitch, wj, lane, TvT, tmpL, ph); break;
  case 8:  fast = inv_stage1_fast<8>(coeff, pitch, wj, lane, TvT, tmpL, ph); break;
  case 16: fast = inv_stage1_fast<16>(coeff, pitch, wj, lane, TvT, tmpL, ph); break;
  case 32: fast = inv_stage1_fast<32>(coeff, pitch, wj, lane, TvT, tmpL, ph); break;
  default: fast = inv_stage1_fast<64>(coeff, pitch, wj, lane, TvT, tmpL, ph); break;
  }
  if (!fast) inv_stage1_slow(coeff, pitch, h, wj, h > 32 ? 32 : h, lane, TvT, tmpL, ph);
  TR_WAVE_SYNC();
  inv_stage2_fast<W>(tmpL, ph, h, lane, s2, tabT + lg_off(d.tr_hor, W), resi, d.resi_stride);
  TR_WAVE_SYNC();
}

__global__ __launch_bounds__(256, 3) void tr_inv_large_kernel(const TCoeff* __restrict__ coeffBase, Pel* __restrict__ resiBase,
                                                           const vvcgpu_tr_desc* __restrict__ descs, const int* __restrict__ list, int bd)
{
  __shared__ __align__(16) short tabT[LG_TAB];
  __shared__ int tmpAll[4][32 * (MAXN + 1)];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cnt = list[0];
  if ((int)blockIdx.x * 4 >= cnt) return;
  lg_load(tabT, d_tr32t, tid);
  __syncthreads();
  int* tmpL = tmpAll[wave];
  for (int k = blockIdx.x * 4 + wave; k < cnt; k += gridDim.x * 4)
  {
    const vvcgpu_tr_desc d = descs[__builtin_amdgcn_readfirstlane(list[1 + k])];
    const TCoeff* coeff = coeffBase + d.coeff_off;
    Pel* resi = resiBase + d.resi_off;
    switch (d.w)
    {
    case 2:  inv_tu_large<2>(d, coeff, resi, bd, lane, tmpL, tabT); break;
    case 4:  inv_tu_large<4>(d, coeff, resi, bd, lane, tmpL, tabT); break;
    case 8:  inv_tu_large<8>(d, coeff, resi, bd, lane, tmpL, tabT); break;
    case 16: inv_tu_large<16>(d, coeff, resi, bd, lane, tmpL, tabT); break;
    case 32: inv_tu_large<32>(d, coeff, resi, bd, lane, tmpL, tabT); break;
    default: inv_tu_large<64>(d, coeff, resi, bd, lane, tmpL, tabT); break;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Large TUs whose sides are both 16, 32 or 64: the stages of mfma_tr.h on the matrix cores, one wave per TU.  A TU the matrix-core form
// cannot take (residual outside +-1023 / a coefficient beyond 16 bits / rows not 16-byte aligned) is appended to the list of the dot2 kernels,
// which run behind this one.
template <int W, int H>
__device__ __forceinline__ bool fwd_tu_mfma(const Pel* __restrict__ resi, int stride, int trHor, int trVer, TCoeff* __restrict__ coeff, int bd, int lane,
                                            const _Float16* tab)
{
  typedef MtShape<W, H> S;
  constexpr int LW = W == 16 ? 4 : W == 32 ? 5 : 6, LH = H == 16 ? 4 : H == 32 ? 5 : 6;
  const int c = lane & 15, g = lane >> 4;
  if ((((uintptr_t)resi | (uintptr_t)((size_t)stride * 2)) & (W == 16 ? 7u : 15u)) != 0) return false;
  h8 x[S::RT][S::XS];
  bool inRange = true;
  if (W == 16)
  {
#pragma unroll
    for (int rt = 0; rt < S::RT; rt++)
    {
      const pel4 v = *reinterpret_cast<const pel4*>(resi + (size_t)(16 * rt + c) * stride + 4 * g);
      _Float16 a[4];
#pragma unroll
      for (int j = 0; j < 4; j++) { inRange = inRange && v[j] >= -1023 && v[j] <= 1023; a[j] = (_Float16)v[j]; }
      x[rt][0] = h8{ a[0], a[1], a[2], a[3], a[0], a[1], a[2], a[3] };
    }
  }
  else
  {
#pragma unroll
    for (int rt = 0; rt < S::RT; rt++)
#pragma unroll
      for (int s = 0; s < S::XS; s++)
      {
        const pel8 v = *reinterpret_cast<const pel8*>(resi + (size_t)(16 * rt + c) * stride + 32 * s + 8 * g);
#pragma unroll
        for (int j = 0; j < 8; j++) { inRange = inRange && v[j] >= -1023 && v[j] <= 1023; x[rt][s][j] = (_Float16)v[j]; }
      }
  }
  if (__builtin_amdgcn_ballot_w64(!inRange) != 0ull) return false;
  const _Float16* Th = tab + rc_tab_off(trHor, W, 0);
  const _Float16* Tv = tab + rc_tab_off(trVer, H, 0);
  const int s1 = LW + bd + 6 - 15 + 2, s2 = LH + 6 + 2;
  int t1[S::JT][S::RT][4];
  mt_fwd1<W, H>(t1, x, Th, s1, c, g);
  int cf[S::IT][S::JT][4];
  mt_fwd2<W, H>(cf, t1, Tv, s2, c, g);
#pragma unroll
  for (int it = 0; it < S::IT; it++)
#pragma unroll
    for (int jt = 0; jt < S::JT; jt++)
#pragma unroll
      for (int r = 0; r < 4; r++) coeff[(16 * it + 4 * g + r) * W + 16 * jt + c] = cf[it][jt][r];
  // zero-out: columns >= 32 of the kept rows, then the rows >= 32 (TrQuant.cpp:157-162)
  const int4v z = { 0, 0, 0, 0 };
  if (W == 64) for (int e = lane; e < S::HJ * 8; e += 64) *reinterpret_cast<int4v*>(coeff + (e >> 3) * 64 + 32 + 4 * (e & 7)) = z;
  if (H == 64) for (int e = lane; e < 32 * W / 4; e += 64) *reinterpret_cast<int4v*>(coeff + 32 * W + 4 * e) = z;
  return true;
}

// coeff: global or LDS, row pitch `pitch`; only the kept region (columns < WJ, rows < HJ) is read
template <int W, int H, class CP>
__device__ __forceinline__ bool inv_tu_mfma(CP coeff, int pitch, Pel* __restrict__ resi, int stride, int trHor, int trVer, int bd, int lane,
                                            const _Float16* tab)
{
  typedef MtShape<W, H> S;
  const int c = lane & 15, g = lane >> 4;
  int cq[S::JT][S::IT][4];
  bool ok = true;
#pragma unroll
  for (int jt = 0; jt < S::JT; jt++)
#pragma unroll
    for (int it = 0; it < S::IT; it++)
#pragma unroll
      for (int r = 0; r < 4; r++) { const int v = coeff[(16 * it + 4 * g + r) * pitch + 16 * jt + c]; ok = ok && fits16(v); cq[jt][it][r] = v; }
  if (__builtin_amdgcn_ballot_w64(!ok) != 0ull) return false;
  const _Float16* ThT = tab + rc_tab_off(trHor, W, 1);
  const _Float16* TvT = tab + rc_tab_off(trVer, H, 1);
  int y1[S::RT][S::JT][4];
  mt_inv1<W, H>(y1, cq, TvT, c, g);
  const int s2 = (6 + 15 - 1) - bd + 2;
  const bool aligned = (((uintptr_t)resi | (uintptr_t)((size_t)stride * 2)) & 7u) == 0;
  mt_inv2<W, H>(y1, ThT, s2, c, g, [&](int rt, int xt, const int (&v)[4])
  {
    Pel* dst = resi + (size_t)(16 * rt + c) * stride + 16 * xt + 4 * g;
    if (aligned) *reinterpret_cast<pel4*>(dst) = pel4{ (short)v[0], (short)v[1], (short)v[2], (short)v[3] };
    else
    {
#pragma unroll
      for (int r = 0; r < 4; r++) dst[r] = (short)v[r];
    }
  });
  return true;
}

// shapes of the matrix-core forms: both sides in {16, 32, 64}
__device__ __forceinline__ bool is_mfma_shape(int w, int h) { return (w == 16 || w == 32 || w == 64) && (h == 16 || h == 32 || h == 64); }

#define TR_MFMA_SHAPES(X) X(64, 64) X(64, 32) X(32, 64) X(32, 32) X(64, 16) X(16, 64) X(32, 16) X(16, 32) X(16, 16)

// lists: mfmaCount[0] TUs at mfmaList[0 ..]; failures are appended to large[1 + large[0]++]
__global__ __launch_bounds__(256, 2) void tr_fwd_mfma_kernel(const Pel* __restrict__ resiBase, TCoeff* __restrict__ coeffBase,
                                                             const vvcgpu_tr_desc* __restrict__ descs, const int* __restrict__ mfmaCount,
                                                             const int* __restrict__ mfmaList, int* __restrict__ large, int bd,
                                                             const _Float16* __restrict__ image)
{
  __shared__ __align__(16) _Float16 tab[RC_TAB_HALVES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int cnt = mfmaCount[0];
  if ((int)blockIdx.x * 4 >= cnt) return;
  rc_load_all_tables(tab, image, tid);
  __syncthreads();
  for (int k = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(tid >> 6); k < cnt; k += gridDim.x * 4)   // wave-uniform for the compiler: list entry and descriptor through the scalar cache
  {
    const int ti = __builtin_amdgcn_readfirstlane(mfmaList[k]);
    const vvcgpu_tr_desc d = descs[ti];
    const Pel* resi = resiBase + d.resi_off;
    TCoeff* coeff = coeffBase + d.coeff_off;
    bool done = false;
#define X(W_, H_) if (d.w == W_ && d.h == H_) done = fwd_tu_mfma<W_, H_>(resi, d.resi_stride, d.tr_hor, d.tr_ver, coeff, bd, lane, tab);
    TR_MFMA_SHAPES(X)
#undef X
    if (!done && lane == 0) large[1 + atomicAdd(&large[0], 1)] = ti;
  }
}
__global__ __launch_bounds__(256, 2) void tr_inv_mfma_kernel(const TCoeff* __restrict__ coeffBase, Pel* __restrict__ resiBase,
                                                             const vvcgpu_tr_desc* __restrict__ descs, const int* __restrict__ mfmaCount,
                                                             const int* __restrict__ mfmaList, int* __restrict__ large, int bd,
                                                             const _Float16* __restrict__ image)
{
  __shared__ __align__(16) _Float16 tab[RC_TAB_HALVES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int cnt = mfmaCount[0];
  if ((int)blockIdx.x * 4 >= cnt) return;
  rc_load_all_tables(tab, image, tid);
  __syncthreads();
  for (int k = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(tid >> 6); k < cnt; k += gridDim.x * 4)   // wave-uniform for the compiler: list entry and descriptor through the scalar cache
  {
    const int ti = __builtin_amdgcn_readfirstlane(mfmaList[k]);
    const vvcgpu_tr_desc d = descs[ti];
    const TCoeff* coeff = coeffBase + d.coeff_off;
    Pel* resi = resiBase + d.resi_off;
    bool done = false;
#define X(W_, H_) if (d.w == W_ && d.h == H_) done = inv_tu_mfma<W_, H_>((GlbCInt*)coeff, W_, resi, d.resi_stride, d.tr_hor, d.tr_ver, bd, lane, tab);
    TR_MFMA_SHAPES(X)
#undef X
    if (!done && lane == 0) large[1 + atomicAdd(&large[0], 1)] = ti;
  }
}

// indices of the large TUs of a batch, in two lists (order irrelevant: TUs are independent): ws[0] = count of the matrix-core list (entries at
// ws[2 + n ..]), ws[1] = count of the dot2 list (entries at ws[2 ..]: `large` = ws + 1 is a count followed by its entries)
// (same-address device atomics retire at ~12 ns each: one per wave made this kernel 35 us for 137 k descriptors; here a 1024-thread workgroup
// aggregates its 16 waves in LDS and reserves its range of each list with ONE atomic)
// smCnt / smLists (long calls only, else null): the small TUs as well, one list per bin of the small kernels (0 transform skip, 1 S <= 4, 2 S = 8,
// 3 S = 16); smCnt is a counter set of vvcgpu_counters (nextCnt: the other set, cleared here for the next call on the stream)
__global__ __launch_bounds__(1024) void tr_collect_large_kernel(const vvcgpu_tr_desc* __restrict__ descs, int n, int* __restrict__ ws, int useMfma,
                                                                int* __restrict__ smCnt, int* __restrict__ smLists, int* __restrict__ nextCnt)
{
  if (nextCnt && blockIdx.x == 0 && threadIdx.x < VVC_CTR_INTS) nextCnt[threadIdx.x] = 0;
  __shared__ int wcnt[6][16], gbase[6];
  const int ti = blockIdx.x * 1024 + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int cat = -1;                                                             // 0 dot2 list, 1 matrix-core list, 2 + bin: small
  if (ti < n)
  {
    const int* f = reinterpret_cast<const int*>(descs + ti) + 5;            // bytes 20..27: w, h, tr_hor, tr_ver
    const int wh = f[0], tt = f[1];
    const int w = (short)(wh & 0xFFFF), h = wh >> 16, S = max(w, h);
    const bool tr = (signed char)(tt & 0xFF) != 3;
    if (tr && useMfma && is_mfma_shape(w, h)) cat = 1;
    else if (tr && (w > 16 || h > 16)) cat = 0;
    else if (smCnt) cat = 2 + (!tr ? 0 : S <= 4 ? 1 : S == 8 ? 2 : 3);
  }
  unsigned long long m[6];
#pragma unroll
  for (int k = 0; k < 6; k++) { m[k] = __builtin_amdgcn_ballot_w64(cat == k); if (lane == 0) wcnt[k][wave] = (int)__popcll(m[k]); }
  __syncthreads();
  if (threadIdx.x < 6)
  {
    const int k = threadIdx.x;
    int tot = 0;
    for (int q = 0; q < 16; q++) { const int c = wcnt[k][q]; wcnt[k][q] = tot; tot += c; }
    gbase[k] = !tot ? 0 : k < 2 ? atomicAdd(&ws[1 - k], tot) : atomicAdd(&smCnt[k - 2], tot);   // ws[1]: dot2 list, ws[0]: matrix-core list
  }
  __syncthreads();
  if (cat < 0) return;
  const int pos = gbase[cat] + wcnt[cat][wave] + (int)__popcll(m[cat] & ((1ull << lane) - 1ull));
  if (cat == 0) ws[2 + pos] = ti;
  else if (cat == 1) ws[2 + n + pos] = ti;
  else smLists[(size_t)(cat - 2) * n + pos] = ti;
}

// ---------------------------------------------------------------------------------------------------
// N1: de-quantisation in front of the inverse transform (vvcgpu_dequant_tr_inv_batch).
//   Quant::dequant (Quant.cpp:277-428, flat scaling): an element-wise map.
//   Dependent quantisation (DepQuant.cpp:708-785): the reconstruction level of a coefficient depends on a 4-state
//   machine driven by the parities of the levels before it in (reverse) scan order.  The transition of one level is a
//   map {0..3} -> {0..3} (8 bits); maps compose associatively, so a wave walks the scan in 64 contiguous chunks: every lane
//   composes the maps of its chunk, an inclusive wave scan of the composed maps gives each lane its entry state, and a
//   second walk reconstructs the levels.  (Zero levels above the last significant one keep state 0, so the walk can
//   start at the end of the scan instead of searching the last level as the reference does.)
__device__ unsigned short d_scan[15876];            // diagonal 4x4-grouped scans of all W x H in 2..64, [log2 w - 1][log2 h - 1]
__device__ int d_scanOff[36];

__device__ __forceinline__ unsigned dq_compose(unsigned first, unsigned then)     // map applied first, then the second one
{
  unsigned r = 0;
#pragma unroll
  for (int s = 0; s < 4; s++) r |= ((then >> (2 * ((first >> (2 * s)) & 3))) & 3) << (2 * s);
  return r;
}

// ---------------------------------------------------------------------------------------------------
// N1 in ONE launch: de-quantiser and inverse transform of a TU in the same wave, the de-quantised coefficients in LDS (the separate form above
// wrote them to a workspace in HBM and read them back in a second and third launch).  A workgroup takes `per` consecutive descriptors and serves
// them in phases: transform skip and the lane-group sizes (<= 16), then the matrix-core shapes, then the remaining large shapes; the tables of
// the phases share one LDS region and are reloaded only when a workgroup's phase changes (homogeneous batches: never).
struct DqP
{
  int dep, rightShift, shift;
  long long scale, inMin, inMax, invQScale, add;
};
__device__ __forceinline__ DqP dq_params(const vvcgpu_dqtr_desc& d, int bd, int lw, int lh)
{
  DqP q;
  const int transformShift = 15 - bd - ((lw + lh) >> 1);
  const bool sqrt2 = ((lw + lh) & 1) != 0;
  q.dep = d.dep_quant;
  {
    const int per = d.qp / 6, rem = d.qp - 6 * per;
    q.rightShift = (sqrt2 ? 8 : 0) + (6 - (transformShift + per));
    const int invq = rem == 0 ? 40 : rem == 1 ? 45 : rem == 2 ? 51 : rem == 3 ? 57 : rem == 4 ? 64 : 72;
    q.scale = (long long)invq * (sqrt2 ? 181 : 1);
    const int targetBits = min(16, 32 + q.rightShift - 7);
    q.inMin = -(1ll << (targetBits - 1)); q.inMax = (1ll << (targetBits - 1)) - 1;
  }
  {
    const int qpDQ = d.qp + 1, qpPer = qpDQ / 6, qpRem = qpDQ - 6 * qpPer;
    int shift = 6 + 1 - qpPer - transformShift + (sqrt2 ? 8 : 0);
    const int invq = qpRem == 0 ? 40 : qpRem == 1 ? 45 : qpRem == 2 ? 51 : qpRem == 3 ? 57 : qpRem == 4 ? 64 : 72;
    long long s = (long long)invq * (sqrt2 ? 181 : 1);
    if (shift < 0) { s <<= -shift; shift = 0; }
    q.invQScale = s; q.shift = shift; q.add = (1ll << shift) >> 1;
  }
  return q;
}
__device__ __forceinline__ int dq_scalar(const DqP& q, int lv)
{
  const long long c = min(max((long long)lv, q.inMin), q.inMax);
  const long long v = q.rightShift > 0 ? (c * q.scale + (1ll << (q.rightShift - 1))) >> q.rightShift : (c * q.scale) << -q.rightShift;
  return (int)min(max(v, -(1ll << 15)), (1ll << 15) - 1);
}

// De-quantises one TU with a group of L lanes (L = 4, 8, 16: L TUs ... 64 / L TUs side by side in the wave; L = 64: the whole wave).  lig = lane
// index inside the group; act = the group has a TU (all lanes of the wave must call: the scan uses shuffles).  The levels of the kept region
// (x < wj, y < hj) are first staged in `stage` (LDS, row pitch wj); levels outside it (64-wide / 64-high TUs only) are read from memory.
// `sink(pos, x, y, v)` receives every de-quantised coefficient (pos = raster index y w + x), each exactly once.
// position of a scan index without the table (host_scan_order below is the definition): coefficient groups of g x g (g = 4, or 2 when a side is 2)
// in up-right diagonal order over the gw x gh grid, the same order inside a group.  dq_cg: group index -> (gy << 8 | gx).
__device__ __forceinline__ int dq_cg(int c, int gw, int gh)
{
  int D = 0, rem = c;
  for (;;)
  {
    const int len = min(D, gh - 1) - max(0, D - gw + 1) + 1;
    if (rem < len) break;
    rem -= len; D++;
  }
  const int gy = min(D, gh - 1) - rem;
  return (gy << 8) | (D - gy);
}
struct DqScan
{
  int lg, gw, gh, cur, ox, oy;                               // cur: the group (ox, oy) belongs to
  __device__ __forceinline__ void init(int w, int h) { lg = ((w | h) & 3) ? 1 : 2; gw = w >> lg; gh = h >> lg; cur = -1; ox = oy = 0; }
  __device__ __forceinline__ void pos(int s, int& x, int& y)
  {
    const int c = s >> (2 * lg), k = s & ((1 << (2 * lg)) - 1);
    if (c != cur) { const int o = dq_cg(c, gw, gh); cur = c; ox = (o & 255) << lg; oy = (o >> 8) << lg; }
    // in-group offsets of scan position k: 4 x 4: x 0010120123123233, y 0102103210321323; 2 x 2: x 0011, y 0101 (two bits each, k = 0 lowest)
    const unsigned kx = lg == 2 ? 0xFB9E4910u : 0x50u, ky = lg == 2 ? 0xEDB1B184u : 0x44u;
    x = ox + (int)((kx >> (2 * k)) & 3); y = oy + (int)((ky >> (2 * k)) & 3);
  }
};

// De-quantises one TU with a group of L lanes (L = 4, 8, 16: 64 / L TUs side by side in the wave; L = 64: the whole wave).  lig = lane
// index inside the group; act = the group has a TU (all lanes of the wave must call: the scan uses shuffles).  The levels of the kept region
// (x < wj, y < hj) are first staged in `stage` (LDS, row pitch wj); levels outside it (64-wide / 64-high TUs only) are read from memory.
// `sink(pos, x, y, v)` receives every de-quantised coefficient (pos = raster index y w + x), each exactly once.
template <int L, class Sink>
__device__ __forceinline__ void dq_group(const vvcgpu_dqtr_desc& d, const TCoeff* __restrict__ levelFlat, int bd, int lig, bool act, int* stageFlat, Sink sink)
{
  GlbCInt* level = (GlbCInt*)levelFlat;
  LdsInt* stage = (LdsInt*)stageFlat;
  const int w = act ? d.w : 2, h = act ? d.h : 2, lw = ilog2(w), lh = ilog2(h);
  const int wj = w > 32 ? 32 : w, hj = h > 32 ? 32 : h, lwj = ilog2(wj);
  const int cnt = act ? w * h : 0, kept = act ? wj * hj : 0;
  for (int e0 = lig; e0 < kept; e0 += 16 * L)                // all loads of a pass in flight before the first LDS store
  {
    int v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) { const int e = e0 + u * L; v[u] = e < kept ? level[((e >> lwj) << lw) + (e & (wj - 1))] : 0; }
#pragma unroll
    for (int u = 0; u < 16; u++) { const int e = e0 + u * L; if (e < kept) stage[e] = v[u]; }
  }
  TR_WAVE_SYNC();
  const DqP q = dq_params(d, bd, lw, lh);
  if (!act || !q.dep)
  {
    for (int pos = lig; pos < cnt; pos += L)
    {
      const int y = pos >> lw, x = pos & (w - 1);
      const bool in = x < wj && y < hj;
      const int lv = in ? stage[y * wj + x] : level[pos];
      sink(pos, x, y, dq_scalar(q, lv));
    }
  }
  const bool dep = act && q.dep;
  // ---- dependent quantisation.  The 4-state machine (transitions 32040, DepQuant.cpp:782) is LINEAR over GF(2): with the state written as
  // (hi, lo), step t maps it to (parity_t ^ lo, hi).  Entering step t, hi = xor of the parities of the earlier steps of the OTHER step parity
  // (t - 1, t - 3, ...) and lo = xor of those of the same parity (t - 2, t - 4, ...), and only hi enters the reconstruction (state >> 1).  So a
  // lane needs two prefix xors: every lane folds the parities of its own steps by step parity, a ballot + popcount gives the prefix over the
  // lanes of the group, and the second pass reconstructs.  Steps run from the END of the scan (step t = scan index cnt - 1 - t; zero levels above
  // the last significant one leave state 0 untouched, so no search for the last level is needed).
  const bool cg4 = ((w | h) & 3) == 0;
  const int laneId = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const unsigned long long below = ((1ull << lig) - 1ull) << (laneId - lig);      // the lanes of my group in front of me
  const bool fast32 = q.invQScale < (1ll << 14) && q.shift < 30;                     // |2 lv +- 1| < 2^17: the product stays inside 31 bits
  const int scale32 = (int)q.invQScale, add32 = (int)q.add;
  auto recon = [&](int lv, int hi) -> int
  {
    if (lv == 0) return 0;
    const int qIdx = 2 * lv + (lv > 0 ? -hi : hi);
    if (fast32) return min(max((qIdx * scale32 + add32) >> q.shift, -(1 << 15)), (1 << 15) - 1);
    return (int)min(max(((long long)qIdx * q.invQScale + q.add) >> q.shift, -(1ll << 15)), (1ll << 15) - 1);
  };
  if (__builtin_amdgcn_ballot_w64(dep && !cg4) == 0ull)
  {
    // coefficient groups of 4 x 4 (every shape without a side of 2): a lane takes whole groups -- 16 levels in registers, the in-group scan unrolled
    constexpr int rasterOf[16] = { 0, 4, 1, 8, 5, 2, 12, 9, 6, 3, 13, 10, 7, 14, 11, 15 };      // scan position k -> y * 4 + x inside the group
    const int nCg = dep ? cnt >> 4 : 0, m = nCg > L ? nCg / L : 1;                              // groups per lane (power of two)
    const int gw = w >> 2, gh = h >> 2;
    auto load16 = [&](int T, int (&r)[16], int& ox, int& oy)
    {
      const int o = dq_cg(nCg - 1 - T, gw, gh);
      ox = (o & 255) << 2; oy = (o >> 8) << 2;
      if (ox < wj && oy < hj)
      {
#pragma unroll
        for (int y = 0; y < 4; y++)
        {
          const int4v v = *reinterpret_cast<const __attribute__((address_space(3))) int4v*>(stage + (oy + y) * wj + ox);
          r[4 * y] = v.x; r[4 * y + 1] = v.y; r[4 * y + 2] = v.z; r[4 * y + 3] = v.w;
        }
      }
      else
      {
#pragma unroll
        for (int i = 0; i < 16; i++) r[i] = level[((oy + (i >> 2)) << lw) + ox + (i & 3)];
      }
    };
    int r[16], ox = 0, oy = 0;
    int a0 = 0, a1 = 0;                                      // xor of the levels at even / odd steps (bit 0 is the parity)
    for (int j = 0; j < m; j++)
    {
      const int T = lig * m + j;
      if (T < nCg)
      {
        load16(T, r, ox, oy);
#pragma unroll
        for (int k = 0; k < 16; k++) { if (k & 1) a0 ^= r[rasterOf[k]]; else a1 ^= r[rasterOf[k]]; }      // step t = 16 T + 15 - k: even for odd k
      }
    }
    const unsigned long long B0 = __builtin_amdgcn_ballot_w64((a0 & 1) != 0), B1 = __builtin_amdgcn_ballot_w64((a1 & 1) != 0);
    int x0 = (int)__popcll(B0 & below) & 1, x1 = (int)__popcll(B1 & below) & 1;
    for (int j = 0; j < m; j++)
    {
      const int T = lig * m + j;
      if (T < nCg)
      {
        if (m > 1) load16(T, r, ox, oy);
#pragma unroll
        for (int k = 15; k >= 0; k--)
        {
          const int i = rasterOf[k], lv = r[i], x = ox + (i & 3), y = oy + (i >> 2);
          const int hi = (k & 1) ? x1 : x0;                 // even step (odd k): xor over the odd steps before it
          sink((y << lw) + x, x, y, recon(lv, hi));
          if (k & 1) x0 ^= lv & 1; else x1 ^= lv & 1;
        }
      }
    }
  }
  else
  {
    // some TU of the wave has a side of 2 (groups of 2 x 2): every TU of the wave goes step by step through the analytic scan
    const bool on = dep;
    const int C = on ? (cnt + L - 1) / L : 0;
    const int t0 = lig * C, t1 = min(t0 + C, on ? cnt : 0);
    DqScan sc;
    sc.init(w, h);
    int a0 = 0, a1 = 0;
    for (int t = t0; t < t1; t++)
    {
      int x, y;
      sc.pos(cnt - 1 - t, x, y);
      const int lv = (x < wj && y < hj) ? stage[y * wj + x] : level[(y << lw) + x];
      if (t & 1) a1 ^= lv; else a0 ^= lv;
    }
    const unsigned long long B0 = __builtin_amdgcn_ballot_w64((a0 & 1) != 0), B1 = __builtin_amdgcn_ballot_w64((a1 & 1) != 0);
    int x0 = (int)__popcll(B0 & below) & 1, x1 = (int)__popcll(B1 & below) & 1;
    for (int t = t0; t < t1; t++)
    {
      int x, y;
      sc.pos(cnt - 1 - t, x, y);
      const int lv = (x < wj && y < hj) ? stage[y * wj + x] : level[(y << lw) + x];
      sink((y << lw) + x, x, y, recon(lv, (t & 1) ? x0 : x1));
      if (t & 1) x1 ^= lv & 1; else x0 ^= lv & 1;
    }
  }
  TR_WAVE_SYNC();
}

constexpr int DQ_UNI = ((LG_TAB * 2 + 15) & ~15) + 4 * 32 * (MAXN + 1) * 4;       // the largest phase: int16 matrices + four wave buffers of the dot2 form
static_assert(DQ_UNI >= (int)sizeof(SmallShared) && DQ_UNI >= RC_TAB_HALVES * 2, "phase region");

template <int S>
__device__ __noinline__ void dq_small_group(SmallShared& sh, int bin, int grp, int lane, int wave, int bd, const TCoeff* __restrict__ levelBase,
                                               Pel* __restrict__ resiBase, TCoeff* __restrict__ coeffOut, int* coefW)
{
  constexpr int P = 64 / S, L = S;
  const int g = lane / S, lig = lane % S, li = grp * P + g;
  const bool act = li < sh.cnt[bin];
  const vvcgpu_dqtr_desc& d = reinterpret_cast<const vvcgpu_dqtr_desc&>(sh.d[sh.list[bin][act ? li : 0]]);
  int* stageF = coefW + g * (S * S);
  LdsInt* stage = (LdsInt*)stageF;
  const int w = d.w;
  GlbInt* out = coeffOut ? (GlbInt*)(coeffOut + d.level_off) : nullptr;
  dq_group<L>(d, levelBase + d.level_off, bd, lig, act, stageF, [&](int pos, int x, int y, int v)
  {
    stage[y * w + x] = v;                                    // w <= 16: the kept region is the TU, pitch w
    if (out) out[pos] = v;
  });
  inv_small_group<S>(sh, bin, grp, lane, wave, bd, nullptr, resiBase, coefW);
}

// shape dispatch of the fused kernel as real calls (inlined into one kernel the eight matrix-core bodies and the six dot2 bodies crash hipcc's simplifycfg)
__device__ __noinline__ bool dq_inv_mfma(int w, int h, const TCoeff* coef, Pel* resi, int stride, int trHor, int trVer, int bd, int lane, const _Float16* ftab)
{
  bool done = false;
#define X(W_, H_) if (w == W_ && h == H_) done = inv_tu_mfma<W_, H_>((const LdsInt*)coef, W_ > 32 ? 32 : W_, resi, stride, trHor, trVer, bd, lane, ftab);
  TR_MFMA_SHAPES(X)
#undef X
  return done;
}
__device__ __noinline__ void dq_inv_large(const vvcgpu_tr_desc& d, const TCoeff* coef, Pel* resi, int bd, int lane, int* tmpL, const short* tabT, int wj)
{
  switch (d.w)
  {
  case 2:  inv_tu_large<2>(d, coef, resi, bd, lane, tmpL, tabT, wj); break;
  case 4:  inv_tu_large<4>(d, coef, resi, bd, lane, tmpL, tabT, wj); break;
  case 8:  inv_tu_large<8>(d, coef, resi, bd, lane, tmpL, tabT, wj); break;
  case 16: inv_tu_large<16>(d, coef, resi, bd, lane, tmpL, tabT, wj); break;
  case 32: inv_tu_large<32>(d, coef, resi, bd, lane, tmpL, tabT, wj); break;
  default: inv_tu_large<64>(d, coef, resi, bd, lane, tmpL, tabT, wj); break;
  }
}

// one TU of each phase as a real call (see above)
__device__ __noinline__ void dq_ts_tu(const vvcgpu_dqtr_desc& d, const TCoeff* __restrict__ levelBase, Pel* __restrict__ resiBase, TCoeff* __restrict__ coeffOut,
                                      int bd, int lane, int* coefW)
{
  const int lw = ilog2(d.w), lh = ilog2(d.h);
  int shift = 15 - bd - ((lw + lh) >> 1), scale = 1;
  if ((lw + lh) & 1) { shift += 7; scale = 181; }
  GlbPel* resi = (GlbPel*)(resiBase + d.resi_off);
  GlbInt* out = coeffOut ? (GlbInt*)(coeffOut + d.level_off) : nullptr;
  const int stride = d.resi_stride;
  dq_group<64>(d, levelBase + d.level_off, bd, lane, true, coefW, [&](int pos, int x, int y, int v)
  {
    if (out) out[pos] = v;
    const int c = v * scale;
    resi[(size_t)y * stride + x] = (short)(shift >= 0 ? (c + (shift ? 1 << (shift - 1) : 0)) >> shift : c << -shift);
  });
}
// de-quantises into coefW (kept region, pitch wj) and, if asked, into coeffOut
__device__ __noinline__ void dq_large_stage(const vvcgpu_dqtr_desc& d, const TCoeff* __restrict__ levelBase, TCoeff* __restrict__ coeffOut, int bd, int lane,
                                            int* coefW)
{
  const int wj = d.w > 32 ? 32 : d.w;
  GlbInt* out = coeffOut ? (GlbInt*)(coeffOut + d.level_off) : nullptr;
  LdsInt* cw = (LdsInt*)coefW;
  dq_group<64>(d, levelBase + d.level_off, bd, lane, true, coefW, [&](int pos, int x, int y, int v)
  {
    if (x < wj && y < 32) cw[y * wj + x] = v;
    if (out) out[pos] = v;
  });
}

// ---- device-side order of a batch for the fused kernel.  A workgroup of dqtr_fused_kernel serves `per` descriptors in up to three phases, each
// with its own tables in the phase region (lane-group tables, f16 matrices, int16 matrices).  In the caller's order -- a real encoder's call mix:
// 85 % of the TUs at most 8 wide, a few 32 / 64 wide ones in every run of 64 -- every workgroup walks ALL phases for a handful of TUs each and
// reloads ~26 KB of matrices per 64 descriptors (more bytes than the TUs themselves): the mixed batch took 2.4 x the time of its parts
// (profiles/r04_chain_shapes.txt).  Here the descriptor INDICES are binned by phase class first (the two-pass scheme of rc_classify_kernel,
// resichain.hip: per-workgroup counts in LDS, one global atomic per class and workgroup); the fused kernel then walks the concatenated class lists,
// heaviest class first, so that a workgroup's descriptors share a phase, lane groups are full and the tables stay.
constexpr int DQC_NCLS = 6, DQC_WGS = 128;                 // classes in list order: 0 dot2 form (large), 1 matrix cores, 2 lane groups of 16, 3 of 8, 4 of 4, 5 transform skip
__device__ __forceinline__ int dqc_class(int w, int h, int trHor, int useMfma)
{
  const int S = max(w, h);
  if (trHor == 3) return 5;
  if (S <= 4) return 4;
  if (S == 8) return 3;
  if (S == 16) return 2;
  return useMfma && is_mfma_shape(w, h) ? 1 : 0;
}
__global__ __launch_bounds__(1024) void dqtr_classify_kernel(const vvcgpu_dqtr_desc* __restrict__ descs, int n, int* __restrict__ hdr, int* __restrict__ lists,
                                                             int* __restrict__ nextHdr, int useMfma)
{
  if (blockIdx.x == 0 && threadIdx.x < VVC_CTR_INTS) nextHdr[threadIdx.x] = 0;         // the header of the NEXT call on this stream (vvcgpu_counters)
  __shared__ int cnt[DQC_NCLS], base[DQC_NCLS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int per = (n + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * per, hi = min(n, lo + per);
  if (tid < DQC_NCLS) cnt[tid] = 0;
  __syncthreads();
  for (int pass = 0; pass < 2; pass++)
  {
    for (int t0 = lo; t0 < hi; t0 += 1024)
    {
      const int ti = t0 + tid;
      int cls = -1;
      if (ti < hi)
      {
        const vvcgpu_tr_desc& d = reinterpret_cast<const vvcgpu_tr_desc*>(descs)[ti];
        cls = dqc_class(d.w, d.h, d.tr_hor, useMfma);
      }
#pragma unroll
      for (int k = 0; k < DQC_NCLS; k++)
      {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(cls == k);
        if (m == 0ull) continue;
        int b = 0;
        if (lane == 0) b = atomicAdd(&cnt[k], (int)__popcll(m));
        b = __builtin_amdgcn_readfirstlane(b);
        if (pass == 1 && cls == k) lists[(size_t)k * n + base[k] + b + (int)__popcll(m & ((1ull << lane) - 1ull))] = ti;
      }
    }
    __syncthreads();
    if (pass == 0 && tid < DQC_NCLS) { base[tid] = cnt[tid] ? atomicAdd(&hdr[tid], cnt[tid]) : 0; cnt[tid] = 0; }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256, 2) void dqtr_fused_kernel(const TCoeff* __restrict__ levelBase, Pel* __restrict__ resiBase,
                                                            const vvcgpu_dqtr_desc* __restrict__ descs, int n, int per, int bd,
                                                            TCoeff* __restrict__ coeffOut, const _Float16* __restrict__ image, int useMfma,
                                                            const int* __restrict__ hdr, const int* __restrict__ lists)
{
  __shared__ int idxOf[SM_DESCS];                            // descriptor index of the batch's t-th entry (the caller's order when lists == nullptr)
  __shared__ __align__(16) unsigned char uni[DQ_UNI];
  __shared__ __align__(16) int coef[4][1024];                // per wave: the kept region of a large TU / the TUs of a lane-group item
  __shared__ int cntM, cntL, cntS[4];
  __shared__ unsigned char listM[SM_DESCS], listL[SM_DESCS];
  __shared__ unsigned short binOf[SM_DESCS];                 // bin * 64 + position in the bin's list (up to 255: a batch of 64 TUs of bin 3), 0xFFFF: not a lane-group TU
  SmallShared& sh = *reinterpret_cast<SmallShared*>(uni);
  _Float16* ftab = reinterpret_cast<_Float16*>(uni);
  short* tabT = reinterpret_cast<short*>(uni);
  int* tmpAll = reinterpret_cast<int*>(uni + ((LG_TAB * 2 + 15) & ~15));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int* coefW = coef[wave];
  int curTab = 0;                                            // 0 nothing, 1 lane-group tables, 2 f16 matrices, 3 int16 matrices (workgroup-uniform)
  // ordered form: a batch is a run of ONE class list, short for the classes whose TUs are long serial jobs of a wave (a workgroup that drew 64
  // large TUs would be the tail of the launch), long for the lane-group classes (16 / 8 / 4 TUs side by side in a wave)
  // (and every class shorter when the whole call would otherwise be fewer than ~1024 batches: idle compute units cost more than part-filled waves)
  int bEnd[DQC_NCLS], cCnt[DQC_NCLS], totalBatches = 0, shr = 0;
#pragma unroll
  for (int c = 0; c < DQC_NCLS; c++) cCnt[c] = lists ? hdr[c] : 0;
  auto pkOf = [&](int c) { return max(4, (c <= 1 ? 8 : c == 5 ? 16 : SM_DESCS) >> shr); };
  for (;; shr++)
  {
    totalBatches = 0;
#pragma unroll
    for (int c = 0; c < DQC_NCLS; c++) { const int pk = pkOf(c); totalBatches += (cCnt[c] + pk - 1) / pk; bEnd[c] = totalBatches; }
    if (totalBatches >= 1024 || shr == 4) break;
  }
  if (!lists) totalBatches = (n + per - 1) / per;
  for (int batch = blockIdx.x; batch < totalBatches; batch += gridDim.x)
  {
    int base = batch * per, count = min(per, n - base), cls = 0;
    if (lists)
    {
      int b0 = 0;
#pragma unroll
      for (int c = 0; c < DQC_NCLS - 1; c++) if (batch >= bEnd[c]) { cls = c + 1; b0 = bEnd[c]; }
      const int pk = pkOf(cls);
      base = (batch - b0) * pk;
      count = min(pk, cCnt[cls] - base);
    }
    __syncthreads();                                         // the previous batch is done with the lists and the phase region
    if (tid < 4) cntS[tid] = 0;
    if (tid == 0) { cntM = 0; cntL = 0; }
    __syncthreads();
    // the descriptor copies and lists of the lane-group phase live in the phase region: only a batch that has such TUs touches it, so a run of
    // large-TU batches keeps its matrices
    if (tid < count)
    {
      const int di = lists ? lists[(size_t)cls * n + base + tid] : base + tid;
      idxOf[tid] = di;
      const vvcgpu_tr_desc d = reinterpret_cast<const vvcgpu_tr_desc*>(descs)[di];
      const int S = max((int)d.w, (int)d.h);
      // 16 x 16 stays with the lane groups here: one tile per wave behind a serial de-quantiser was slower (0.122 vs 0.083 ms at 4K)
      const int bin = d.tr_hor == 3 ? 0 : S <= 4 ? 1 : S == 8 ? 2 : S == 16 ? 3 : -1;
      if (bin >= 0) { const int k = atomicAdd(&cntS[bin], 1); binOf[tid] = (unsigned short)(bin * 64 + k); }
      else
      {
        binOf[tid] = 0xFFFFu;
        if (useMfma && is_mfma_shape(d.w, d.h)) listM[atomicAdd(&cntM, 1)] = (unsigned char)tid;
        else listL[atomicAdd(&cntL, 1)] = (unsigned char)tid;
      }
    }
    __syncthreads();
    const int anySmall = cntS[0] + cntS[1] + cntS[2] + cntS[3];
    if (anySmall)
    {
      if (curTab > 1) { curTab = 0; }                        // the matrices are about to be overwritten
      if (tid < 4) sh.cnt[tid] = cntS[tid];
      if (tid < count && binOf[tid] != 0xFFFFu)
      {
        sh.d[tid] = reinterpret_cast<const vvcgpu_tr_desc*>(descs)[idxOf[tid]];
        sh.list[binOf[tid] >> 6][binOf[tid] & 63] = (unsigned char)tid;
      }
      __syncthreads();
    }
    if (anySmall)
    {
      if (curTab != 1) { small_tables(sh, tid); curTab = 1; __syncthreads(); }
      for (int q = wave; q < sh.cnt[0]; q += 4)              // transform skip: the de-quantised value goes straight through the element-wise inverse
        dq_ts_tu(reinterpret_cast<const vvcgpu_dqtr_desc&>(sh.d[sh.list[0][q]]), levelBase, resiBase, coeffOut, bd, lane, coefW);
      for (int g = wave; g * 16 < sh.cnt[1]; g += 4) dq_small_group<4>(sh, 1, g, lane, wave, bd, levelBase, resiBase, coeffOut, coefW);
      for (int g = wave; g * 8 < sh.cnt[2]; g += 4)  dq_small_group<8>(sh, 2, g, lane, wave, bd, levelBase, resiBase, coeffOut, coefW);
      for (int g = wave; g * 4 < sh.cnt[3]; g += 4)  dq_small_group<16>(sh, 3, g, lane, wave, bd, levelBase, resiBase, coeffOut, coefW);
    }
    if (cntM)
    {
      if (curTab != 2)
      {
        __syncthreads();                                     // the lane-group phase is done with the region
        rc_load_all_tables(ftab, image, tid); curTab = 2;
        __syncthreads();
      }
      for (int q = wave; q < cntM; q += 4)
      {
        const vvcgpu_dqtr_desc d = descs[idxOf[listM[q]]];
        dq_large_stage(d, levelBase, coeffOut, bd, lane, coefW);
        const bool done = dq_inv_mfma(d.w, d.h, coefW, resiBase + d.resi_off, d.resi_stride, d.tr_hor, d.tr_ver, bd, lane, ftab);
        if (!done && lane == 0) listL[atomicAdd(&cntL, 1)] = listM[q];          // a coefficient beyond 16 bits: the dot2 form's exact 32-bit stage takes the TU
        TR_WAVE_SYNC();
      }
    }
    __syncthreads();
    if (cntL)
    {
      if (curTab != 3) { lg_load(tabT, d_tr32t, tid); curTab = 3; __syncthreads(); }
      int* tmpL = tmpAll + wave * (32 * (MAXN + 1));
      for (int q = wave; q < cntL; q += 4)
      {
        const vvcgpu_dqtr_desc dq = descs[idxOf[listL[q]]];
        const vvcgpu_tr_desc& d = reinterpret_cast<const vvcgpu_tr_desc&>(dq);
        dq_large_stage(dq, levelBase, coeffOut, bd, lane, coefW);
        dq_inv_large(d, coefW, resiBase + d.resi_off, bd, lane, tmpL, tabT, d.w > 32 ? 32 : d.w);
      }
    }
  }
}

// ---- forward scalar quantisation without RDOQ: Quant::quant (Quant.cpp:721-834) + xSignBitHidingHDQ (:142-273) ---------------
// Without sign hiding the map is element-wise.  With it, the coefficient groups (16 coefficients in scan order) are independent
// once the TU's abs-sum and its last group with a level are known: pass 1 finds both, pass 2 decides every group's adjustment.
struct QuantParams { int qBits, qBits8, scale, whScale; long long add; };

__device__ __forceinline__ int quant_one(const QuantParams& q, int c, int& deltaU)
{
  const long long tmp = (long long)abs(c) * q.scale * q.whScale;
  const int mag = (int)((tmp + q.add) >> q.qBits);
  deltaU = (int)((tmp - ((long long)mag << q.qBits)) >> q.qBits8);
  return mag;
}

// TUs of up to 256 coefficients are handled by 16 lanes each, four side by side in a wavefront (a step = one coefficient group of
// each; most of a picture's TUs are small, so this fills the lanes and shares the set-up); larger TUs take the whole wavefront
// (four coefficient groups per step).
template <bool LARGE>
__device__ __forceinline__ void quant_tu(const TCoeff* __restrict__ coeffBase, TCoeff* __restrict__ levelBase, const vvcgpu_quant_desc& d, bool live,
                                         int ti, int bd, unsigned* __restrict__ absSumOut, int lane)
{
  constexpr int LPT = LARGE ? 64 : 16;
  const int k = lane & 15, slot = lane >> 4, tl = lane & (LPT - 1);
  const int w = d.w, h = d.h, cnt = live ? w * h : 0, lw = ilog2(w), lh = ilog2(h);
  const TCoeff* coef = coeffBase + d.coeff_off;
  TCoeff* level = levelBase + d.level_off;
  QuantParams q;
  {
    const int per = d.qp / 6, rem = d.qp - 6 * per;
    int transformShift = 15 - bd - ((lw + lh) >> 1);
    q.whScale = 1;
    if ((lw + lh) & 1) { transformShift += 7; q.whScale = 181; }
    q.qBits = 14 + per + transformShift; q.qBits8 = q.qBits - 8;
    q.scale = rem == 0 ? 26214 : rem == 1 ? 23302 : rem == 2 ? 20560 : rem == 3 ? 18396 : rem == 4 ? 16384 : 14564;
    q.add = (long long)(d.intra_slice ? 171 : 85) << (q.qBits - 9);
  }
  const bool sbh = d.sign_hiding && w >= 4 && h >= 4;
  const unsigned short* scan = d_scan + d_scanOff[(lw - 1) * 6 + (lh - 1)];
  int maxCnt = cnt;                                         // the wavefront runs as many steps as its largest TU needs
#pragma unroll
  for (int m = LPT; m < 64; m <<= 1) maxCnt = max(maxCnt, __shfl_xor(maxCnt, m));
  if (maxCnt == 0) return;
  int sum = 0;
  if (!sbh || !live)
  {
    // element-wise (sign hiding off): levels are final
    for (int s0 = 0; s0 < maxCnt; s0 += LPT)
    {
      const int si = s0 + tl;
      if (si < cnt && !sbh)
      {
        int du; const int c = coef[si];
        const int mag = quant_one(q, c, du);
        sum += mag;
        level[si] = min(max(c < 0 ? -mag : mag, -32768), 32767);
      }
    }
  }
  // Sign hiding, ONE pass from the end of the scan: the first coefficient group met with a level is the reference's "last" group
  // (:183-186), every other group searches all 16 positions.  The reference hides only if uiAbsSum >= 2; a group that qualifies
  // (last - first >= 4) has two levels, so the test can only fail when the 32-bit sum wrapped -- handled after the loop.
  // 16 lanes per coefficient group; the sequential search for the cheapest parity fix (:196-262, the highest scan position wins
  // ties) is a 16-lane min over (cost, -position).
  bool foundLast = false, fixedAny = false;
  const int steps = (maxCnt + LPT - 1) / LPT;
  for (int st = steps - 1; st >= 0; st--)
  {
    const int si = st * LPT + tl;
    const bool in = sbh && live && si < cnt;
    const int pos = in ? scan[si] : 0;
    const int c = in ? coef[pos] : 0;
    int du;
    const int mag = quant_one(q, c, du);
    sum += in ? mag : 0;
    int lv = min(max(c < 0 ? -mag : mag, -32768), 32767);
    const unsigned long long nzAll = __ballot(lv != 0);
    const unsigned nz = (unsigned)((nzAll >> (16 * slot)) & 0xFFFFull);
    // is this lane's group the last one with a level?  no earlier-met (= later in scan order) group of this TU had one
    bool isLast;
    if (LARGE)
    {
      const unsigned long long higher = slot == 3 ? 0ull : (nzAll >> (16 * (slot + 1)));
      isLast = !foundLast && nz != 0 && higher == 0ull;
      foundLast = foundLast || nzAll != 0ull;
    }
    else { isLast = !foundLast && nz != 0; foundLast = foundLast || nz != 0; }
    const int first = nz ? __ffs((int)nz) - 1 : 16, last = nz ? 31 - __clz((int)nz) : -1;
    int ssum = lv;
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) ssum += __shfl_xor(ssum, m);
    const int firstLv = __shfl(lv, (lane & 48) + (first & 15));
    const unsigned signbit = firstLv > 0 ? 0u : 1u;
    const bool fix = last - first >= 4 && signbit != (unsigned)(ssum & 1);
    const int start = isLast ? last : 15;
    const int TMAX = 0x7fffffff;
    int cost = TMAX, change = 0;
    if (k <= start)
    {
      if (lv != 0)
      {
        if (du > 0) { cost = -du; change = 1; }
        else if (!(k == first && abs(lv) == 1)) { cost = du; change = -1; }
      }
      else if (k < first) { if ((c >= 0 ? 0u : 1u) == signbit) { cost = -du; change = 1; } }
      else { cost = -du; change = 1; }
    }
    long long key = ((long long)cost << 5) + (15 - k);
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) { const long long o = __shfl_xor(key, m); key = min(key, o); }
    if (fix && k == 15 - (int)(key & 31))
    {
      if (lv == 32767 || lv == -32768) change = -1;
      lv += c >= 0 ? change : -change;
    }
    fixedAny = fixedAny || (fix && in);
    if (in) level[pos] = lv;
  }
#pragma unroll
  for (int m = 1; m < LPT; m <<= 1) sum += __shfl_xor(sum, m);
  if (live && tl == 0) absSumOut[ti] = (unsigned)sum;
  // uiAbsSum is a 32-bit int in the reference: if it wrapped below 2 no hiding happened there -- rewrite the plain levels
  if (sbh && live && sum < 2 && __ballot(fixedAny) != 0ull)
    for (int s0 = 0; s0 < cnt; s0 += LPT)
    {
      const int si = s0 + tl;
      if (si < cnt) { int du; const int c = coef[si]; const int mag = quant_one(q, c, du); level[si] = min(max(c < 0 ? -mag : mag, -32768), 32767); }
    }
}

// two launches: the first takes the small TUs (four per wavefront) and lists the large ones; the second walks that list, one large
// TU per wavefront at a time (a persistent grid: no empty workgroups for the many small TUs of a picture)
__global__ __launch_bounds__(256) void quant_small_kernel(const TCoeff* __restrict__ coeffBase, TCoeff* __restrict__ levelBase,
                                                          const vvcgpu_quant_desc* __restrict__ descs, int n, int bd, unsigned* __restrict__ absSumOut,
                                                          int* __restrict__ largeList)
{
  const int lane = threadIdx.x & 63;
  const int ti = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
  const vvcgpu_quant_desc d = descs[ti < n ? ti : n - 1];
  const bool large = (int)d.w * d.h > 256;
  if (ti < n && large && (lane & 15) == 0) largeList[1 + atomicAdd(&largeList[0], 1)] = ti;
  const bool live = ti < n && !large;
  if (__ballot(live) == 0ull) return;
  quant_tu<false>(coeffBase, levelBase, d, live, ti, bd, absSumOut, lane);
}

__global__ __launch_bounds__(256) void quant_large_kernel(const TCoeff* __restrict__ coeffBase, TCoeff* __restrict__ levelBase,
                                                          const vvcgpu_quant_desc* __restrict__ descs, int bd, unsigned* __restrict__ absSumOut,
                                                          const int* __restrict__ largeList)
{
  const int lane = threadIdx.x & 63;
  const int count = largeList[0], waves = gridDim.x * 4;
  for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < count; i += waves)
  {
    const int ti = largeList[1 + i];
    const vvcgpu_quant_desc d = descs[ti];
    quant_tu<true>(coeffBase, levelBase, d, true, ti, bd, absSumOut, lane);
  }
}

// ---- dependent-quantisation trellis: DQIntern::DepQuant::quant (DepQuant.cpp:1323-1391) ---------------------------------------
// The trellis is a sequential walk down the scan with four states; TUs are independent.  FOUR LANES own one TU, lane k carries
// trellis state k (its previous-position state and its skip state live in the lane's registers), sixteen TUs share a wavefront.
// Per scan position: every lane prices the transitions LEAVING its state (two quantisation candidates + zero), the three 64-bit
// costs entering decision k are gathered with quad shuffles in the reference's comparison order (:1222-1249, strict '<'), the
// winner's template context (16 abs levels + 16 context seeds = 12 dwords) is pulled from the source lane, and the new rates are
// looked up in the caller's rate tables.  Decisions (absLevel << 4 | prevId + 2) go to the workspace for the back-trace; the
// per-state sub-block memory of CommonCtx (:828-858) lives in the workspace as well and is touched only at sub-block ends.
__device__ unsigned short d_dqInv[15876];           // raster position -> scan id, same layout as d_scan
// What a position record holds that depends on the TU's SHAPE only (built on the host with the scan tables), per scan id si, for the
// position AFTER si in the walk (scan id max(si - 1, 0)): the byte selectors of its five template neighbours inside the sub-block
// (DqRec below) and the word (neighbour positions 5 x 4 bits | sigOff << 20 | gtxOff << 24) for luma (.x) and chroma (.y).
__device__ uint4 d_dqPosSel[15876];
__device__ uint2 d_dqPosMisc[15876];

// Small per-lane tables are ext-vector VALUES, not arrays: element selects then stay register selects (with arrays LLVM rewrites a
// select of loads into a load through a selected address, which pins the whole state struct in scratch memory).
typedef unsigned dq_u4 __attribute__((ext_vector_type(4)));
typedef unsigned dq_u8 __attribute__((ext_vector_type(8)));
typedef int dq_i8 __attribute__((ext_vector_type(8)));
typedef long long dq_l4 __attribute__((ext_vector_type(4)));
typedef int dq_i4 __attribute__((ext_vector_type(4)));

struct DqState
{
  long long rdCost;
  dq_u4 lev;                          // 16 abs levels of the current sub-block (bytes)
  dq_u4 aux;                          // per level min(4 - (v & 1), v) | (v != 0) << 5: what it adds to sumAbs1 and sumNum of a template
  int numSigSbb, refSbbCtxId;         // refSbbCtxId also names the LDS slot with the 16 template-context seeds of the sub-block (-1: all zero)
  int sbb0, sbb1, sig0, sig1;
  int gc;                             // row of the greater-than-x rate table (coefficient bit sums [0..6])
  int goRice;
};

// member-wise copy: a whole-struct assignment also copies the padding through scratch memory
__device__ __forceinline__ void dq_copy(DqState& d, const DqState& s)
{
  d.rdCost = s.rdCost; d.lev = s.lev; d.aux = s.aux; d.numSigSbb = s.numSigSbb; d.refSbbCtxId = s.refSbbCtxId;
  d.sbb0 = s.sbb0; d.sbb1 = s.sbb1; d.sig0 = s.sig0; d.sig1 = s.sig1; d.gc = s.gc; d.goRice = s.goRice;
}
__device__ __forceinline__ unsigned dq_get_byte(const dq_u4 a, int j)
{
  const int d = j >> 2;
  const unsigned v = d == 0 ? a[0] : d == 1 ? a[1] : d == 2 ? a[2] : a[3];
  return (v >> ((j & 3) * 8)) & 0xFFu;
}
__device__ __forceinline__ void dq_set_byte(dq_u4& a, int j, unsigned val)
{
  const int d = j >> 2, sh = (j & 3) * 8;
#pragma unroll
  for (int i = 0; i < 4; i++) { const unsigned m = i == d ? 0xFFu << sh : 0u; a[i] = (a[i] & ~m) | ((val << sh) & m); }   // no conditional store: keeps the array in registers
}
__device__ __forceinline__ unsigned dq_get_u16(const dq_u8 c, int j)
{
  const int d = j >> 1;
  unsigned v = c[0];
#pragma unroll
  for (int i = 1; i < 8; i++) v = d == i ? c[i] : v;
  return (v >> ((j & 1) * 16)) & 0xFFFFu;
}
typedef const __attribute__((address_space(3))) vvcgpu_dq_rates* DqLdsRates;
__device__ __forceinline__ int dq_level_bits(DqLdsRates rt, int gc, int goRice, unsigned level)       // State::getLevelBits :909-931
{
  const unsigned idx = level < 5 ? level : 5 + ((level - 5) & 1);
  const int bits = rt->gtx[gc][idx];
  if (level < 5) return bits;
  // escape part; the prefix loop of :924-929 ends at length = floor(log2(value - thres + 2^goRice))
  const unsigned value = (level - 5) >> 1;
  const unsigned range = goRice == 0 ? 6u : goRice == 1 ? 5u : goRice == 2 ? 6u : 3u;              // g_auiGoRiceRange
  const unsigned thres = range << goRice;
  const unsigned length = 31u - (unsigned)__clz((int)(value - thres + (1u << goRice)));
  const unsigned esc = value < thres ? (value >> goRice) + 1 + goRice : range + 1 + (length << 1) - goRice;
  return bits + (int)(esc << 15);
}
__device__ __forceinline__ long long dq_shfl64(long long v, int src) { return __shfl(v, src); }
// quad permutation with a compile-time pattern (v_mov_b32 dpp quad_perm): no LDS round trip on the cost chain
template <int CTRL>
__device__ __forceinline__ long long dq_quad64(long long v)
{
  const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)v, CTRL, 0xF, 0xF, true);
  const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)((unsigned long long)v >> 32), CTRL, 0xF, 0xF, true);
  return (long long)(((unsigned long long)hi << 32) | lo);
}

template <int CTRL>
__device__ __forceinline__ unsigned dq_quad32(unsigned v) { return (unsigned)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true); }

// the diagonal scan inside a 4x4 sub-block: scan index of in-block position y * 4 + x, and its inverse (4 bits each)
constexpr unsigned long long dq_pack_scan4(bool inverse)
{
  unsigned long long kof = 0, posof = 0; int k = 0;
  for (int d = 0; d < 7; d++)
    for (int y = (d < 3 ? d : 3); y >= 0; y--)
    {
      const int x = d - y;
      if (x > 3) continue;
      kof |= (unsigned long long)k << (4 * (y * 4 + x)); posof |= (unsigned long long)(y * 4 + x) << (4 * k); k++;
    }
  return inverse ? posof : kof;
}
constexpr unsigned long long DQ_KOFPOS = dq_pack_scan4(false), DQ_POSOFK = dq_pack_scan4(true);

// What a trellis step needs that does not depend on the trellis state, per scan position: the four quantisation candidates of
// Quantizer::preQuantCoeff (:786-808), the two "start here" costs (checkRdCostStart :1196-1213: candidate 0 / 2 + last-position bits +
// level bits in the start context), and for the position AFTER it the in-sub-block template neighbours (:139-168) and its context offsets.
// The quad fills the sixteen records of a sub-block when the walk enters it (four positions per lane) instead of every lane repeating
// the same arithmetic at every step: ~250 of a step's ~735 instructions were this.
struct DqRec
{
  long long dist[4];                  // pqData.deltaDist by slot (qIdx & 3)
  unsigned short ab[4];               // pqData.absLevel by slot
  long long start[2];                 // decision 0 / decision 2
  unsigned misc;                      // neighbour positions 5 x 4 bits | sigOff << 20 | gtxOff << 24
  // v_perm_b32 selectors that pick the five neighbours out of the sixteen level bytes: group A = neighbours 0..3, group B = neighbour 4;
  // Lo reads bytes 0..7, Hi bytes 8..15, selector 12 (= constant zero) where the neighbour is in the other half or does not exist
  unsigned selLoA, selHiA, selLoB, selHiB, pad;
};
static_assert(sizeof(DqRec) == 80, "DqRec");
constexpr int DQ_REC_N = 8;                                               // positions filled at a time (half a sub-block)
constexpr int DQ_SEED_BYTES = 5 * 32;                                     // per TU: the seeds of context slots 0..3 + an all-zero slot
constexpr int DQ_LDS_BYTES = 64 * (DQ_REC_N * (int)sizeof(DqRec) + DQ_SEED_BYTES);   // 64 quads per workgroup
constexpr int DQ_RT_SLOTS = 16;

__global__ __launch_bounds__(256) void depquant_kernel(const TCoeff* __restrict__ coeffBase, TCoeff* __restrict__ levelBase,
                                                       const vvcgpu_depquant_desc* __restrict__ descs, int n,
                                                       const vvcgpu_dq_rates* __restrict__ ratesBase, int bd, unsigned* __restrict__ absSumOut,
                                                       unsigned* __restrict__ wsDec, unsigned char* __restrict__ wsCtx)
{
  const int lane = threadIdx.x & 63, k = lane & 3, qbase = lane & ~3;
  const int ti = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (lane >> 2);
  const bool live = ti < n;
  const vvcgpu_depquant_desc d = descs[live ? ti : n - 1];
  const int w = d.w, h = d.h, N = w * h, lw = ilog2(w), lh = ilog2(h);
  const int widthInSbb = w >> 2, heightInSbb = h >> 2, numSbb = N >> 4;
  const bool luma = d.luma != 0;
  const TCoeff* coef = coeffBase + d.coeff_off;
  TCoeff* level = levelBase + d.level_off;
  const int tabOff = d_scanOff[(lw - 1) * 6 + (lh - 1)];
  const unsigned short* scan = d_scan + tabOff;
  const unsigned short* inv = d_dqInv + tabOff;
  const uint4* posSel = d_dqPosSel + tabOff;
  const uint2* posMisc = d_dqPosMisc + tabOff;
  unsigned* dec = wsDec + (size_t)d.coeff_off * 4;                         // [scanIdx][4]
  // CommonCtx's per-state sub-block memory (:828-858) as a pool of 16-byte blocks [sub-block in scan order][context slot]: the levels of a
  // sub-block as the state that took slot k at its end left them (the first 4 N of the 8 N bytes a TU has in the workspace)
  unsigned char* ctxMem = wsCtx + (size_t)d.coeff_off * 8;

  // Quantizer::initQuantBlock :647-706 (the same IEEE double arithmetic)
  int qShift, maxQIdx, thresLast, distShift;
  long long qAdd, qScale, distAdd, distStepAdd, distOrgFact;
  {
    const int qpDQ = d.qp + 1, qpPer = qpDQ / 6, qpRem = qpDQ - 6 * qpPer;
    const bool sqrt2 = ((lw + lh) & 1) != 0;
    const int transformShift = 15 - bd - ((lw + lh) >> 1);
    const int qs = qpRem == 0 ? 26214 : qpRem == 1 ? 23302 : qpRem == 2 ? 20560 : qpRem == 3 ? 18396 : qpRem == 4 ? 16384 : 14564;
    qShift = 14 - 1 + qpPer + transformShift;
    qAdd = -((3ll << qShift) >> 1);
    const int invShift = 6 + 1 - qpPer - transformShift + (sqrt2 ? 8 : 0);
    qScale = sqrt2 ? (qs * 181) >> 7 : qs;
    const unsigned qIdxBD = min(16u, (unsigned)(32 + invShift - 6 - 1));
    maxQIdx = (1 << (qIdxBD - 1)) - 4;
    thresLast = (int)((3ll << qShift) / (4 * qScale));
    const int nomDShift = 15 - 2 * transformShift + qShift;
    const double qScale2 = (double)((long long)qs * qs);
    const double nomDistFactor = nomDShift < 0 ? 1.0 / ((double)(1ll << (-nomDShift)) * qScale2 * d.lambda) : (double)(1ll << nomDShift) / (qScale2 * d.lambda);
    const long long pow2dfShift = (long long)(nomDistFactor * qScale2) + 1;
    int dfShift = 0;
    while ((1ull << dfShift) < (unsigned long long)pow2dfShift && dfShift < 63) dfShift++;
    distShift = 62 + qShift - 2 * 15 - dfShift;
    distAdd = (1ll << distShift) >> 1;
    distStepAdd = (long long)(nomDistFactor * (double)(1ll << (distShift + qShift)) + .5);
    distOrgFact = (long long)(nomDistFactor * (double)(1ll << (distShift + 1)) + .5);
  }

  // first tested position :1337-1349 (four lanes split the search), levels start as zero
  int first = -1;
  if (live)
  {
    for (int i = k; i < N; i += 4) level[i] = 0;
    // (sixteen positions a round per quad, the loads of a round independent of each other: the search of a 64x64 TU with a zeroed-out
    // high-frequency region walks ~2000 positions whose coefficients come from memory)
    for (int i = N - 1 - k; i >= 0 && first < 0; i -= 16)
    {
      int a[4];
#pragma unroll
      for (int u = 0; u < 4; u++) a[u] = abs(coef[scan[max(i - 4 * u, 0)]]);
#pragma unroll
      for (int u = 3; u >= 0; u--) if (i - 4 * u >= 0 && a[u] > thresLast) first = i - 4 * u;
    }
  }
  first = max(first, __shfl_xor(first, 1));
  first = max(first, __shfl_xor(first, 2));
  if (live && first < 0 && k == 0) absSumOut[ti] = 0;

  // LDS: per quad the position records and the template seeds of its four context slots; per workgroup the rate tables.  The tables
  // are looked up on the critical path of every step, so the walk only ever reads them from LDS: up to DQ_RT_SLOTS distinct tables of
  // the workgroup's 64 TUs are staged per pass, TUs whose table found no slot walk in the next pass (one pass unless a caller
  // mixes more than sixteen tables inside 64 consecutive TUs).
  extern __shared__ __align__(16) unsigned char dqSmem[];
  DqRec* const recTu = reinterpret_cast<DqRec*>(dqSmem) + (threadIdx.x >> 2) * DQ_REC_N;
  unsigned* const seedTu = reinterpret_cast<unsigned*>(dqSmem + 64 * DQ_REC_N * sizeof(DqRec)) + (threadIdx.x >> 2) * (DQ_SEED_BYTES / 4);
  __shared__ vvcgpu_dq_rates rtCache[DQ_RT_SLOTS];
  __shared__ int rtSlot[DQ_RT_SLOTS];
  __shared__ int rtPending;
#pragma unroll
  for (int i = 0; i < DQ_SEED_BYTES / 4 / 4; i++) seedTu[k * (DQ_SEED_BYTES / 4 / 4) + i] = 0u;

  auto walk = [&](const bool run, DqLdsRates rt)
  {
  int maxFirst = run ? first : -1;
#pragma unroll
  for (int m = 4; m < 64; m <<= 1) maxFirst = max(maxFirst, __shfl_xor(maxFirst, m));
  maxFirst = __builtin_amdgcn_readfirstlane(maxFirst);                    // the walk's position is the same in every lane: keep it (and what
  if (maxFirst < 0) return;                                               // derives from it) in scalar registers

  const int sigSet = max(k - 1, 0);                                       // RateEstimator::sigFlagBits(stateId) :282-285
  DqState P, S;                                                           // previous-position state k, skip state k
  {
    P.rdCost = 0x7FFFFFFFFFFFFFFFll >> 1; P.numSigSbb = 0; P.refSbbCtxId = -1; P.goRice = 0; P.sbb0 = P.sbb1 = 0;
    P.sig0 = rt->sig[sigSet][0][0]; P.sig1 = rt->sig[sigSet][0][1];
    P.lev = dq_u4{ 0, 0, 0, 0 }; P.aux = dq_u4{ 0, 0, 0, 0 }; P.gc = 0;
    dq_copy(S, P);
  }
  DqState P0; dq_copy(P0, P);
  // The level history of a state (CommonCtx::update copies `setCpSize` bytes of it from the parent state at every sub-block end, :1104-1130)
  // is never copied here: a block of the pool is written once, and a context slot carries the ANCESTRY of its path -- which slot its
  // ancestor took at the end of each of the last 32 sub-blocks, two bits each, youngest in the low bits, and how many of them exist
  // (a path that starts inside a sub-block has none: the reference zeroes its history).  The farthest block a template reads lies 30
  // sub-blocks back (64x64).  Like the flags below, the pair lives in the lane whose number is the slot's.  The copy was 30 x 64 lines
  // of 16 bytes per wavefront and sub-block end for 64x64 TUs, a twentieth of their walk.
  unsigned long long ancCur = 0; int ancLen = 0;
  dq_u8 Fcur;                                                             // coded-sub-block flags (bit per sub-block) of context slot k, current half
#pragma unroll
  for (int i = 0; i < 8; i++) Fcur[i] = 0;
  long long finalCost = 0;

  auto fillRec = [&](int si, int p, int coefAbs, const uint4 sel, const unsigned misc)
  {
    const int x = p & (w - 1), y = p >> lw;
    // Quantizer::preQuantCoeff :786-808
    dq_l4 pqDist = { 0, 0, 0, 0 }; dq_i4 pqAbs = { 0, 0, 0, 0 };
    {
      const long long scaledOrg = (long long)coefAbs * qScale;
      int qIdx = max(1, min(maxQIdx, (int)((scaledOrg + qAdd) >> qShift)));
      long long scaledAdd = qIdx * distStepAdd - scaledOrg * distOrgFact;
#pragma unroll
      for (int i = 0; i < 4; i++)
      {
        const int slot = qIdx & 3;
        const long long dd = (scaledAdd * qIdx + distAdd) >> distShift;
        const int al = (++qIdx) >> 1;
#pragma unroll
        for (int t = 0; t < 4; t++) { pqDist[t] = t == slot ? dd : pqDist[t]; pqAbs[t] = t == slot ? al : pqAbs[t]; }
        scaledAdd += distStepAdd;
      }
    }
    const int lastOffset = rt->last_x[x] + rt->last_y[y];
    DqRec* r = recTu + (si & (DQ_REC_N - 1));
    r->selLoA = sel.x; r->selHiA = sel.y; r->selLoB = sel.z; r->selHiB = sel.w;       // the shape-only part: straight from the table
#pragma unroll
    for (int t = 0; t < 4; t++) r->dist[t] = pqDist[t];
    *reinterpret_cast<uint2*>(r->ab) = make_uint2((unsigned)pqAbs[0] | (unsigned)pqAbs[1] << 16, (unsigned)pqAbs[2] | (unsigned)pqAbs[3] << 16);
    r->start[0] = pqDist[0] + lastOffset + dq_level_bits(rt, 0, 0, (unsigned)pqAbs[0]);
    r->start[1] = pqDist[2] + lastOffset + dq_level_bits(rt, 0, 0, (unsigned)pqAbs[2]);
    r->misc = misc;
  };
  // transitions leaving state k: state 0: pq0 -> dec0, pq2 -> dec2; state 1: pq2 -> dec0, pq0 -> dec2; state 2: pq3 -> dec1, pq1 -> dec3;
  // state 3: pq1 -> dec1, pq3 -> dec3; the zero transition goes to dec0 / dec2 / dec1 / dec3  (:1229-1240)
  const int lowIdx = k == 0 ? 0 : k == 1 ? 2 : k == 2 ? 3 : 1, highIdx = lowIdx ^ 2;
  struct DqRecRegs { long long dl, dh, start; uint2 ab; unsigned misc; uint4 sel; };
  auto loadRec = [&](int inside)
  {
    const DqRec* r = recTu + inside;
    DqRecRegs v;
    v.dl = r->dist[lowIdx]; v.dh = r->dist[highIdx]; v.start = r->start[k >> 1];
    v.ab = *reinterpret_cast<const uint2*>(r->ab); v.misc = r->misc;
    v.sel = make_uint4(r->selLoA, r->selHiA, r->selLoB, r->selHiB);
    return v;
  };
  auto abOf = [](uint2 ab, int t) { return (int)(((t < 2 ? ab.x : ab.y) >> ((t & 1) * 16)) & 0xFFFFu); };
  DqRecRegs R, Rn;
  int pfPos[2] = { 0, 0 }, pfAbs[2] = { 0, 0 };
  uint4 pfSel[2] = { make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0) }; unsigned pfMisc[2] = { 0, 0 };
  auto prefetch = [&](int beg)
  {
#pragma unroll
    for (int j = 0; j < 2; j++)
    {
      const int si = beg + k + 4 * j;
      pfPos[j] = scan[si]; const int c = coef[pfPos[j]]; pfAbs[j] = abs(c);
      pfSel[j] = posSel[si]; const uint2 m = posMisc[si]; pfMisc[j] = (luma ? m.x : m.y) | (c < 0 ? 0x80000000u : 0u);   // bit 31: the coefficient's sign
    }
  };
  Rn.dl = Rn.dh = Rn.start = 0; Rn.ab = make_uint2(0, 0); Rn.misc = 0; Rn.sel = make_uint4(0, 0, 0, 0);

  for (int scanIdx = maxFirst; scanIdx >= 0; scanIdx--)
  {
    const bool act = run && scanIdx <= first;                             // quad-uniform
    const int sIdx = scanIdx;                                             // inactive quads compute on valid indices and discard
    const int insidePos = sIdx & 15;
    const bool eosbb = insidePos == 0, sosbb = insidePos == 15;
    const bool socsbb = sosbb && sIdx > 16 && sIdx < N - 1;
    const bool eocsbb = eosbb && sIdx > 0 && sIdx < N - 16;
    const int spt = socsbb ? 1 : (eocsbb ? 2 : 0);
    const int nxt = max(sIdx - 1, 0);
    // a quad that is not active yet (scanIdx > first) computes along and its state is whatever that leaves: it starts from the
    // initial state at its first tested position (instead of guarding every state copy of every step)
    if (scanIdx == first)
    {
      dq_copy(P, P0); dq_copy(S, P0); ancCur = 0; ancLen = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) Fcur[i] = 0;
    }
    const int recPos = sIdx & (DQ_REC_N - 1);
    if (recPos == DQ_REC_N - 1 || scanIdx == maxFirst)                    // wave-uniform: the walk enters a group of positions
    {
      const int beg = sIdx & ~(DQ_REC_N - 1);
      static_assert(DQ_REC_N == 8, "two records per lane");
      if (scanIdx == maxFirst) prefetch(beg);
#pragma unroll
      for (int j = 0; j < 2; j++) fillRec(beg + k + 4 * j, pfPos[j], pfAbs[j], pfSel[j], pfMisc[j]);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
      R = loadRec(recPos);
      // the coefficients of the NEXT group are the one stream of a TU that comes from HBM: requested here (with the shape part of its
      // records), used eight steps later
      prefetch(max(beg - DQ_REC_N, 0));
    }
    else R = Rn;
    if (recPos != 0) Rn = loadRec(recPos - 1);                            // the next step's record is in flight during this one

    const long long INF = 0x7FFFFFFFFFFFFFFFll;
    long long cLow, cHigh, cZero = INF;
    {
      // checkRdCostNonZero / checkRdCostZero by scan-position type (:1133-1177) as selects (the three-way branch diverges inside a wavefront):
      // the significance bits count unless the sub-block's coded flag is inferred (its end with nothing significant so far: no zero either),
      // the coded-flag bits at its start
      const bool zeroOk = !(spt == 2 && P.numSigSbb == 0);
      const int sb = spt == 1 ? P.sbb1 : 0;
      const int extra1 = (zeroOk ? P.sig1 : 0) + sb, extra0 = (zeroOk ? P.sig0 : 0) + sb;
      cLow = P.rdCost + R.dl + dq_level_bits(rt, P.gc, P.goRice, (unsigned)abOf(R.ab, lowIdx)) + extra1;
      cHigh = P.rdCost + R.dh + dq_level_bits(rt, P.gc, P.goRice, (unsigned)abOf(R.ab, highIdx)) + extra1;
      if (zeroOk) cZero = P.rdCost + extra0;
    }
    // decision k: sources a = 0 / 2, b = a + 1; k < 2 takes their "low" transitions, k >= 2 the "high" ones.  The source lanes are a fixed
    // pattern of the quad: lanes (0, 1, 2, 3) read a = (0, 2, 0, 2) and b = (1, 3, 1, 3)
    const int a = (k & 1) * 2, b = a + 1;
    const long long aLow = dq_quad64<0x88>(cLow), aHigh = dq_quad64<0x88>(cHigh), aZero = dq_quad64<0x88>(cZero);
    const long long bLow = dq_quad64<0xDD>(cLow), bHigh = dq_quad64<0xDD>(cHigh), bZero = dq_quad64<0xDD>(cZero);
    long long dCost = INF >> 2; int dAbs = -1, dPrev = -2;
    {
      // pq index of the transition a -> k and b -> k
      const int ia = k == 0 ? 0 : k == 2 ? 2 : k == 1 ? 3 : 1, ib = ia ^ 2;
      const int absA = abOf(R.ab, ia), absB = abOf(R.ab, ib);
      // comparison order (strict '<'): k < 2: a, a's zero, b; k >= 2: a, b, b's zero -- as selects, the two orders share one code path
      const bool lo = k < 2;
      const long long cA = lo ? aLow : aHigh, cB = lo ? bLow : bHigh;
      const long long c2 = lo ? aZero : cB, c3 = lo ? cB : bZero;
      const int abs2 = lo ? 0 : absB, prev2 = lo ? a : b, abs3 = lo ? absB : 0;
      if (cA < dCost) { dCost = cA; dAbs = absA; dPrev = a; }
      if (c2 < dCost) { dCost = c2; dAbs = abs2; dPrev = prev2; }
      if (c3 < dCost) { dCost = c3; dAbs = abs3; dPrev = b; }
      if (spt == 2) { const long long c = S.rdCost + S.sbb0; if (c < dCost) { dCost = c; dAbs = 0; dPrev = 4 + k; } }          // checkRdCostSkipSbb
      if ((k & 1) == 0 && R.start < dCost) { dCost = R.start; dAbs = abOf(R.ab, k); dPrev = -1; }                     // checkRdCostStart (decisions 0, 2)
    }
    // (with the coefficient's sign in bit 31: the back-trace then reads nothing but the decisions -- the coefficients of a picture do not stay in L2)
    if (act) dec[(size_t)sIdx * 4 + k] = ((unsigned)max(dAbs, 0) << 4) | (unsigned)(dPrev + 2) | (R.misc & 0x80000000u);
    if (scanIdx == 0) finalCost = dCost;

    // ---- state update (:1259-1318); every lane pulls its winner's context from the source lane
    DqState C; dq_copy(C, P);                                              // becomes the new previous state
    if (sIdx > 0)
    {
      const int sigOff = (int)((R.misc >> 20) & 15u), gtxOff = (int)((R.misc >> 24) & 31u);
      const int nextInside = nxt & 15;
      // source of the copied context: lane dPrev (0..3), own skip state (4 + k) or nothing
      const int srcLane = qbase + (dPrev >= 0 && dPrev < 4 ? dPrev : k);
      dq_u4 lv = { 0, 0, 0, 0 }, ax = { 0, 0, 0, 0 }; int sNum, sRef, sSbb0, sSbb1;
#pragma unroll
      for (int i = 0; i < 4; i++) lv[i] = (unsigned)__shfl((int)P.lev[i], srcLane);
      if (!eosbb)
#pragma unroll
        for (int i = 0; i < 4; i++) ax[i] = (unsigned)__shfl((int)P.aux[i], srcLane);
      sNum = __shfl(P.numSigSbb, srcLane); sRef = __shfl(P.refSbbCtxId, srcLane);
      sSbb0 = __shfl(P.sbb0, srcLane); sSbb1 = __shfl(P.sbb1, srcLane);
      if (dPrev >= 4) { sNum = S.numSigSbb; sRef = S.refSbbCtxId;
#pragma unroll
        for (int i = 0; i < 4; i++) lv[i] = S.lev[i]; }
      // sub-block flags of the inherited context slot: a register pull from the lane that owns the slot (slot id = lane in the quad);
      // done by the whole quad (the branch below diverges inside a quad)
      dq_u8 nf = { 0, 0, 0, 0, 0, 0, 0, 0 };
      unsigned long long nAnc = 0; int nLen = 0;
      if (eosbb)
      {
        const int pr = dPrev >= 0 ? sRef : -1;
#pragma unroll
        for (int i = 0; i < 8; i++) { const unsigned v = (unsigned)__shfl((int)Fcur[i], qbase + max(pr, 0)); nf[i] = pr >= 0 ? v : 0u; }
        const unsigned long long pa = (unsigned long long)dq_shfl64((long long)ancCur, qbase + max(pr, 0));
        const int pl = __shfl(ancLen, qbase + max(pr, 0));
        nAnc = pr >= 0 ? (pa << 2) | (unsigned long long)pr : 0ull; nLen = pr >= 0 ? min(pl + 1, 32) : 0;
      }
      C.rdCost = dCost;
      if (dPrev > -2)
      {
        int sumAbs, sumAbs1, sumNum;
        if (!eosbb)                                                        // State::updateState :1004-1068
        {
          if (dPrev >= 0) { C.numSigSbb = sNum + (dAbs != 0); C.refSbbCtxId = sRef; C.sbb0 = sSbb0; C.sbb1 = sSbb1;
#pragma unroll
            for (int i = 0; i < 4; i++) { C.lev[i] = lv[i]; C.aux[i] = ax[i]; } }
          else { C.numSigSbb = 1; C.refSbbCtxId = -1;
#pragma unroll
            for (int i = 0; i < 4; i++) { C.lev[i] = 0; C.aux[i] = 0; } }
          const unsigned lvNew = (unsigned)min(255, dAbs);
          dq_set_byte(C.lev, insidePos, lvNew);
          dq_set_byte(C.aux, insidePos, min(4u - (lvNew & 1u), lvNew) | (lvNew != 0u ? 32u : 0u));
          // the seeds of a sub-block belong to the context slot that was current when the walk entered it; every state that descends
          // from it reads the same sixteen values (slot 4 = zeros: a path that started inside the sub-block)
          const unsigned tinit = reinterpret_cast<const unsigned short*>(seedTu)[(C.refSbbCtxId < 0 ? 4 : C.refSbbCtxId) * 16 + nextInside];
          sumAbs = (int)(tinit >> 8); sumAbs1 = (int)((tinit >> 3) & 31); sumNum = (int)(tinit & 7);
          // the five template neighbours inside the sub-block: four byte permutes pick them out of the sixteen levels (and out of their
          // sumAbs1 / sumNum contributions), v_sad_u8 against zero adds the picked bytes up
          {
            const unsigned nA = __builtin_amdgcn_perm(C.lev[1], C.lev[0], R.sel.x) | __builtin_amdgcn_perm(C.lev[3], C.lev[2], R.sel.y);
            const unsigned nB = __builtin_amdgcn_perm(C.lev[1], C.lev[0], R.sel.z) | __builtin_amdgcn_perm(C.lev[3], C.lev[2], R.sel.w);
            const unsigned xA = __builtin_amdgcn_perm(C.aux[1], C.aux[0], R.sel.x) | __builtin_amdgcn_perm(C.aux[3], C.aux[2], R.sel.y);
            const unsigned xB = __builtin_amdgcn_perm(C.aux[1], C.aux[0], R.sel.z) | __builtin_amdgcn_perm(C.aux[3], C.aux[2], R.sel.w);
            sumAbs = (int)__builtin_amdgcn_sad_u8(nB, 0u, __builtin_amdgcn_sad_u8(nA, 0u, (unsigned)sumAbs));
            const unsigned sx = __builtin_amdgcn_sad_u8(xB, 0u, __builtin_amdgcn_sad_u8(xA, 0u, 0u));
            sumAbs1 += (int)(sx & 31u); sumNum += (int)(sx >> 5);
          }
        }
        else                                                               // State::updateStateEOS :1071-1102 + CommonCtx::update :1104-1164
        {
          if (dPrev >= 0) { C.numSigSbb = sNum + (dAbs != 0);
#pragma unroll
            for (int i = 0; i < 4; i++) C.lev[i] = lv[i]; }
          else { C.numSigSbb = 1;
#pragma unroll
            for (int i = 0; i < 4; i++) C.lev[i] = 0; }
          dq_set_byte(C.lev, insidePos, (unsigned)min(255, dAbs));
          const int sbbId = sIdx >> 4;
          if (act) *reinterpret_cast<uint4*>(ctxMem + (size_t)(sbbId * 4 + k) * 16) = make_uint4(C.lev[0], C.lev[1], C.lev[2], C.lev[3]);
          const int pos = scan[sIdx], px = pos & (w - 1), py = pos >> lw, nxtPos = scan[nxt], nx = nxtPos & (w - 1), ny = nxtPos >> lw;
          {
            const int sbbPos = (py >> 2) * widthInSbb + (px >> 2);
#pragma unroll
            for (int i = 0; i < 8; i++) { const unsigned m = i == (sbbPos >> 5) ? 1u << (sbbPos & 31) : 0u; nf[i] = (nf[i] & ~m) | (C.numSigSbb != 0 ? m : 0u); }
          }
          const int nsx = nx >> 2, nsy = ny >> 2, nsp = nsy * widthInSbb + nsx;
          const int right = nsx < widthInSbb - 1 ? nsp + 1 : 0, below = nsy < heightInSbb - 1 ? nsp + widthInSbb : 0;
          unsigned fr = nf[0], fb = nf[0];
#pragma unroll
          for (int i = 1; i < 8; i++) { fr = (right >> 5) == i ? nf[i] : fr; fb = (below >> 5) == i ? nf[i] : fb; }
          const int sigNSbb = ((right && ((fr >> (right & 31)) & 1u)) || (below && ((fb >> (below & 31)) & 1u))) ? 1 : 0;
#pragma unroll
          for (int i = 0; i < 8; i++) Fcur[i] = nf[i];
          ancCur = nAnc; ancLen = nLen;
          C.numSigSbb = 0; C.refSbbCtxId = k;
          C.sbb0 = rt->sig_sbb[sigNSbb][0]; C.sbb1 = rt->sig_sbb[sigNSbb][1];
          // template seeds of the sixteen positions of the next sub-block from the levels outside it (:1131-1160).  Every template
          // neighbour outside a 4x4 sub-block lies in the sub-block to its right, below it or below-right of it, whose sixteen levels
          // are sixteen consecutive bytes of the history (scan order): three 16-byte loads and compile-time byte picks replace eighty
          // dependent byte loads behind eighty table look-ups (13.6 us per sub-block end, a fifth of the kernel).
          dq_u8 cti = { 0, 0, 0, 0, 0, 0, 0, 0 };
          if (act)
          {
            const int bx = nsx * 4, by = nsy * 4;
            const bool hasR = nsx + 1 < widthInSbb, hasB = nsy + 1 < heightInSbb;
            // read from the blocks of this state's ancestors (the sub-block that just ended: its levels are still in C.lev)
            const uint4 own = make_uint4(C.lev[0], C.lev[1], C.lev[2], C.lev[3]), zero4 = make_uint4(0, 0, 0, 0);
            // (loads from addresses that are always valid, the choice made on the VALUES: a choice between a loaded value and `own` / zero
            // becomes a load through a selected address, i.e. `own` goes to scratch memory and the three loads wait for each other)
            auto sbbLevels = [&](bool exists, int rasterPos)
            {
              const int j = inv[exists ? rasterPos : 0] >> 4, back = j - sbbId - 1;      // back = 0: the parent's sub-block
              const unsigned slot = (unsigned)(nAnc >> (2 * min(max(back, 0), 31))) & 3u;
              const uint4 hv = *reinterpret_cast<const uint4*>(ctxMem + (size_t)(j * 4 + (int)slot) * 16);
              const bool fromHist = exists && back >= 0 && back < nLen, fromOwn = exists && j == sbbId;
              uint4 r;
              r.x = fromHist ? hv.x : fromOwn ? own.x : 0u; r.y = fromHist ? hv.y : fromOwn ? own.y : 0u;
              r.z = fromHist ? hv.z : fromOwn ? own.z : 0u; r.w = fromHist ? hv.w : fromOwn ? own.w : 0u;
              return r;
            };
            const uint4 LR = sbbLevels(hasR, by * w + bx + 4), LB = sbbLevels(hasB, (by + 4) * w + bx), LD = sbbLevels(hasR && hasB, (by + 4) * w + bx + 4);
            auto pick = [](const uint4& v, int j) { const unsigned q = j < 4 ? v.x : j < 8 ? v.y : j < 12 ? v.z : v.w; return (q >> ((j & 3) * 8)) & 0xFFu; };
            // contribution of one neighbour level to (sumNum | sumAbs1 << 3 | sumAbs << 8): at most five are added, the fields do not carry
            auto cv = [](unsigned v) { return (v != 0u ? 1u : 0u) + (min(4u - (v & 1u), v) << 3) + (v << 8); };
            unsigned cR[4][2], cB[2][4];
#pragma unroll
            for (int y = 0; y < 4; y++)
#pragma unroll
              for (int x = 0; x < 2; x++) cR[y][x] = cv(pick(LR, (int)((DQ_KOFPOS >> (4 * (y * 4 + x))) & 15)));
#pragma unroll
            for (int y = 0; y < 2; y++)
#pragma unroll
              for (int x = 0; x < 4; x++) cB[y][x] = cv(pick(LB, (int)((DQ_KOFPOS >> (4 * (y * 4 + x))) & 15)));
            const unsigned cD = cv(pick(LD, (int)(DQ_KOFPOS & 15)));
#pragma unroll
            for (int i = 0; i < 16; i++)
            {
              const int pi = (int)((DQ_POSOFK >> (4 * i)) & 15), x = pi & 3, y = pi >> 2;
              const int dx[5] = { 1, 2, 1, 0, 0 }, dy[5] = { 0, 0, 1, 1, 2 };
              unsigned sum = 0;
#pragma unroll
              for (int t = 0; t < 5; t++)
              {
                const int X = x + dx[t], Y = y + dy[t];
                if (X > 3 && Y > 3) sum += cD; else if (X > 3) sum += cR[Y][X - 4]; else if (Y > 3) sum += cB[Y - 4][X];
              }
              const unsigned seed = (sum & 0xFFu) | (min(127u, sum >> 8) << 8);
              cti[i >> 1] |= seed << ((i & 1) * 16);
            }
            *reinterpret_cast<uint4*>(seedTu + k * 8) = make_uint4(cti[0], cti[1], cti[2], cti[3]);
            *reinterpret_cast<uint4*>(seedTu + k * 8 + 4) = make_uint4(cti[4], cti[5], cti[6], cti[7]);
          }
#pragma unroll
          for (int i = 0; i < 4; i++) { C.lev[i] = 0; C.aux[i] = 0; }
          const unsigned tinit = dq_get_u16(cti, nextInside);
          sumNum = (int)(tinit & 7); sumAbs1 = (int)((tinit >> 3) & 31); sumAbs = (int)(tinit >> 8);
        }
        const int sumGt1 = sumAbs1 - sumNum;
        sumAbs -= sumNum;
        const int sc = sigOff + min(sumAbs1, 5), gc = gtxOff + min(sumGt1, 4);
        C.sig0 = rt->sig[sigSet][sc][0]; C.sig1 = rt->sig[sigSet][sc][1];
        C.gc = gc;
        const int ga = min(sumAbs, 31);
        C.goRice = ga < 12 ? 0 : ga < 25 ? 1 : 2;                          // g_auiGoRicePars
      }
      if (eosbb) { __threadfence_block(); }
    }
    if (socsbb) dq_copy(S, P);                                             // swap( m_prevStates, m_skipStates ) :1314-1317
    dq_copy(P, C);
  }

  // ---- best final state and back-trace :1368-1390.  Decisions 4..7 are implicit: at a sub-block end they are a copy of decisions
  // 0..3 (:1269), elsewhere { level 0, same skip id } (startDec :1218).  The chain through the decisions is serial, the loads are not: the
  // quad takes eight scan positions a round, lane j loads the four decisions of positions base + j and base + 4 + j (16 bytes each, the coefficient's sign in bit 31) and
  // their raster positions -- one round ahead --, the chain then runs over quad broadcasts in registers (every lane alike) and lane j writes the
  // level of its position.  (With lane 0 alone every position was a dependent load from memory: ~0.5 ms of a 64x64 TU's 3.1 ms.)
  long long c1 = dq_shfl64(finalCost, qbase + 1), c2 = dq_shfl64(finalCost, qbase + 2), c3 = dq_shfl64(finalCost, qbase + 3);
  const long long c0 = dq_shfl64(finalCost, qbase);
  if (!run) return;
  int prevId = -2; long long minCost = 0;
  if (c0 < minCost) { prevId = 0; minCost = c0; }
  if (c1 < minCost) { prevId = 1; minCost = c1; }
  if (c2 < minCost) { prevId = 2; minCost = c2; }
  if (c3 < minCost) { prevId = 3; minCost = c3; }
  unsigned absSum = 0;
  __threadfence_block();                                                   // the decisions were stored by the four lanes
  const uint4* dec4 = reinterpret_cast<const uint4*>(dec);
  // (eight positions a round, two per lane: the chain over eight positions takes about as long as the loads of the next eight)
  uint4 dv[2]; int pos[2];
#pragma unroll
  for (int u = 0; u < 2; u++) { const int i = min(4 * u + k, N - 1); dv[u] = dec4[i]; pos[u] = scan[i]; }
  for (int base = 0; prevId >= 0; base += 8)
  {
    uint4 dn[2]; int posn[2];
#pragma unroll
    for (int u = 0; u < 2; u++) { const int i = min(base + 8 + 4 * u + k, N - 1); dn[u] = dec4[i]; posn[u] = scan[i]; }
#pragma unroll
    for (int u = 0; u < 2; u++)
    {
      int myAl = 0; bool mine = false, myNeg = false;
      // a link of the chain: every lane picks the decision of the current state out of ITS position's four (two selects on the bits of
      // the state: nested conditionals became branches), lane j's pick is the one that counts
      auto pick = [&]()
      {
        const bool b0 = (prevId & 1) != 0, b1 = (prevId & 2) != 0;
        const unsigned lo = b0 ? dv[u].y : dv[u].x, hi = b0 ? dv[u].w : dv[u].z;
        return b1 ? hi : lo;
      };
      auto link = [&](int j, unsigned v)
      {
        const bool on = prevId >= 0, keep = prevId >= 4 && ((base + 4 * u + j) & 15) != 0;
        const int al = keep ? 0 : (int)((v >> 4) & 0x7FFFFFFu), nextPrev = keep ? prevId : (int)(v & 15) - 2;
        if (on && j == k) { myAl = al; mine = true; myNeg = (v >> 31) != 0u; }
        absSum += on ? (unsigned)al : 0u; prevId = on ? nextPrev : prevId;
      };
      link(0, dq_quad32<0x00>(pick()));                                    // quad_perm [j, j, j, j]: lane j of the quad to all four
      link(1, dq_quad32<0x55>(pick()));
      link(2, dq_quad32<0xAA>(pick()));
      link(3, dq_quad32<0xFF>(pick()));
      if (mine) level[pos[u]] = myNeg ? -myAl : myAl;
    }
#pragma unroll
    for (int u = 0; u < 2; u++) { dv[u] = dn[u]; pos[u] = posn[u]; }
  }
  if (k != 0) return;
  absSumOut[ti] = absSum;
  };

  bool done = !(live && first >= 0);
  for (;;)
  {
    if (threadIdx.x < DQ_RT_SLOTS) rtSlot[threadIdx.x] = -1;
    if (threadIdx.x == 0) rtPending = 0;
    __syncthreads();
    int mySlot = -1;
    if (!done && k == 0)
    {
      for (int t = 0; t < DQ_RT_SLOTS && mySlot < 0; t++)
      {
        const int sl = (d.rates_idx + t) & (DQ_RT_SLOTS - 1);
        const int old = atomicCAS(&rtSlot[sl], -1, d.rates_idx);
        if (old == -1 || old == d.rates_idx) mySlot = sl;
      }
      if (mySlot < 0) rtPending = 1;
    }
    mySlot = __shfl(mySlot, qbase);
    __syncthreads();
    const bool more = rtPending != 0;
    for (int sl = 0; sl < DQ_RT_SLOTS; sl++)
      if (rtSlot[sl] >= 0)
      {
        const int* src = reinterpret_cast<const int*>(ratesBase + rtSlot[sl]);
        int* dst = reinterpret_cast<int*>(&rtCache[sl]);
        for (int i = threadIdx.x; i < (int)(sizeof(vvcgpu_dq_rates) / 4); i += 256) dst[i] = src[i];
      }
    __syncthreads();
    const bool run = !done && mySlot >= 0;
    walk(run, (DqLdsRates)&rtCache[max(mySlot, 0)]);
    done = done || run;
    if (!more) break;
    __syncthreads();
  }
}

// ---- N1: rate-distortion optimised quantiser (QuantRDOQ::xRateDistOptQuant, QuantRDOQ.cpp:694-1409) -------------------------
// Sixteen lanes per TU, lane k = position k of the current 4x4 coefficient group.  What the reference does one coefficient at a
// time splits into: per-coefficient quantities (parallel), the level decisions (each reads the five template neighbours: the
// anti-diagonals of a group are independent, so seven steps decide sixteen levels; neighbours inside the group travel by lane
// shuffles, those in earlier groups are read back from the level buffer), and the running cost sums, which are IEEE double
// additions in scan order and therefore stay a serial chain (evaluated by every lane of the team alike, operands by shuffle).
// All double arithmetic is written in the reference's order with contraction off.
#pragma clang fp contract(off)

constexpr unsigned long long rdoq_pack_kofpos()
{
  // lane (scan index inside a 4x4 group) of in-group position y * 4 + x, from the diagonal scan
  unsigned long long v = 0; int k = 0;
  for (int d = 0; d < 7; d++)
    for (int y = (d < 3 ? d : 3); y >= 0; y--)
    {
      const int x = d - y;
      if (x > 3) continue;
      v |= (unsigned long long)k << (4 * (y * 4 + x)); k++;
    }
  return v;
}
constexpr unsigned long long RDOQ_KOFPOS = rdoq_pack_kofpos();

struct RdoqBits { int par0, par1, gt10, gt11, gt20, gt21; };

__device__ __forceinline__ int rdoq_ic_rate(unsigned a, const RdoqBits& b, int rice)              // xGetICRate :235-313
{
  if (a == 0) return 0;
  int rate = 32768;
  if (a >= 5)
  {
    unsigned symbol = (a - 5) >> 1;
    const int thr = rice == 1 ? 5 : 6;                                                              // g_auiGoRiceRange[0..2]
    if (symbol < (unsigned)(thr << rice)) rate += (int)((symbol >> rice) + 1 + rice) << 15;
    else
    {
      // the escape loop (:279-286) ends at length = floor(log2(symbol' + 2^rice)), symbol' = symbol - (thr << rice)
      const int length = 31 - __clz((int)(symbol - (unsigned)(thr << rice) + (1u << rice)));
      rate += (thr + length + 1 - rice + length) << 15;
    }
    rate += (((a - 1) & 1) ? b.par1 : b.par0) + b.gt11 + b.gt21;
  }
  else if (a == 1) rate += b.par0 + b.gt10;
  else if (a == 2) rate += b.par1 + b.gt10;
  else if (a == 3) rate += b.par0 + b.gt11 + b.gt20;
  else rate += b.par1 + b.gt11 + b.gt20;
  return rate;
}

__device__ __forceinline__ double rdoq_shfl(double v, int src) { return __shfl(v, src); }

__global__ __launch_bounds__(256) void rdoq_kernel(const TCoeff* __restrict__ coeffBase, TCoeff* __restrict__ levelBase,
                                                   const vvcgpu_rdoq_desc* __restrict__ descs, int n, const vvcgpu_rdoq_rates* __restrict__ rates,
                                                   int bd, unsigned* __restrict__ absSumOut, double* __restrict__ wsD, int* __restrict__ wsI,
                                                   double* __restrict__ wsCG, unsigned char* __restrict__ wsSG, size_t c)
{
  const int ti = (int)((blockIdx.x * 256u + threadIdx.x) >> 4), k = threadIdx.x & 15, tb = threadIdx.x & 48;
  if (ti >= n) return;                                                                              // whole teams leave
  const vvcgpu_rdoq_desc d = descs[ti];
  const int w = d.w, h = d.h, lw = ilog2(w), lh = ilog2(h), numCG = (w * h) >> 4, wig = w >> 2, hig = h >> 2;
  const unsigned short* scan = d_scan + d_scanOff[(lw - 1) * 6 + (lh - 1)];
  const TCoeff* src = coeffBase + d.coeff_off;
  TCoeff* dst = levelBase + d.level_off;
  const vvcgpu_rdoq_rates* rt = rates + d.rates_idx;
  const double lambda = d.lambda;
  const bool luma = d.luma != 0;
  const int per = d.qp / 6, rem = d.qp - 6 * per;
  const int transformShift = 15 - bd - ((lw + lh) >> 1);
  const bool sqrt2 = ((lw + lh) & 1) != 0;
  const int qBits = 14 + per + transformShift;
  const int qs = rem == 0 ? 26214 : rem == 1 ? 23302 : rem == 2 ? 20560 : rem == 3 ? 18396 : rem == 4 ? 16384 : 14564;     // g_quantScales
  const int quantCoef = sqrt2 ? (qs * 181) >> 7 : qs;
  const double errScale = ldexp(1.0, 15 - 2 * transformShift + (sqrt2 ? 1 : 0)) / quantCoef / quantCoef;                  // xGetErrScaleCoeff :482-506
  const int half = 1 << (qBits - 1);
  // workspace, indexed by coeff_off + scan position (coeff_off is a multiple of 16: the sixteen lanes write one line)
  double* wCoeff = wsD + d.coeff_off; double* wSig = wsD + c + d.coeff_off; double* wCoeff0 = wsD + 2 * c + d.coeff_off;
  int* wUp = wsI + d.coeff_off; int* wDown = wsI + c + d.coeff_off; int* wSigDelta = wsI + 2 * c + d.coeff_off; int* wDeltaU = wsI + 3 * c + d.coeff_off;
  double* wCG = wsCG + (d.coeff_off >> 4);
  unsigned char* wSG = wsSG + (d.coeff_off >> 4);

  double blockUncoded = 0, baseCost = 0;
  int cgLastScanPos = -1, lastScanPos = -1;
  for (int subSet = numCG - 1; subSet >= 0; subSet--)
  {
    const int sp = (subSet << 4) + k, pos = scan[sp], x = pos & (w - 1), y = pos >> lw, x4 = x & 3, y4 = y & 3, diag4 = x4 + y4;
    const int cgX = x >> 2, cgY = y >> 2, cgPos = cgY * wig + cgX;
    const int sigRight = cgX + 1 < wig ? wSG[cgPos + 1] : 0, sigLower = cgY + 1 < hig ? wSG[cgPos + wig] : 0;
    const int sg0 = rt->sig_group[sigRight | sigLower][0], sg1 = rt->sig_group[sigRight | sigLower][1];
    // ---- per coefficient :830-843
    const long long tmpLevel = (long long)abs(src[pos]) * quantCoef;
    const int levelDouble = (int)min(tmpLevel, (long long)0x7FFFFFFF - half);
    const unsigned maxAbs = min(32767u, (unsigned)((levelDouble + half) >> qBits));
    const double err0 = (double)levelDouble;
    const double cost0 = err0 * err0 * errScale;
    if (lastScanPos < 0)
    {
      const unsigned m = (unsigned)(__ballot(maxAbs > 0) >> tb) & 0xFFFFu;
      if (m) { lastScanPos = (subSet << 4) + 31 - __clz((int)m); cgLastScanPos = subSet; }
    }
    const bool inRange = lastScanPos >= 0 && sp <= lastScanPos, isLast = sp == lastScanPos;
    // ---- template neighbours (x+1,y) (x+2,y) (x+1,y+1) (x,y+1) (x,y+2): validity as nested at ContextModelling.h:144-164
    const bool v0 = x < w - 1, v1 = x < w - 2, v2 = v0 && y < h - 1, v3 = y < h - 1, v4 = y < h - 2;
    const bool in0 = x4 < 3, in1 = x4 < 2, in2 = x4 < 3 && y4 < 3, in3 = y4 < 3, in4 = y4 < 2;
    int nb0 = (v0 && !in0) ? dst[pos + 1] : 0, nb1 = (v1 && !in1) ? dst[pos + 2] : 0, nb2 = (v2 && !in2) ? dst[pos + w + 1] : 0,
        nb3 = (v3 && !in3) ? dst[pos + w] : 0, nb4 = (v4 && !in4) ? dst[pos + 2 * w] : 0;
    const int p4 = y4 * 4 + x4;
    const int l0 = tb + (int)((RDOQ_KOFPOS >> (4 * ((p4 + 1) & 15))) & 15), l1 = tb + (int)((RDOQ_KOFPOS >> (4 * ((p4 + 2) & 15))) & 15),
              l2 = tb + (int)((RDOQ_KOFPOS >> (4 * ((p4 + 5) & 15))) & 15), l3 = tb + (int)((RDOQ_KOFPOS >> (4 * ((p4 + 4) & 15))) & 15),
              l4 = tb + (int)((RDOQ_KOFPOS >> (4 * ((p4 + 8) & 15))) & 15);
    int level = 0, incUp = 0, incDown = 0, sigDelta = 0, deltaU = 0;
    double costCoeff = 0, costSig = 0;
    for (int dg = 6; dg >= 0; dg--)
    {
      const int t0 = __shfl(level, l0), t1 = __shfl(level, l1), t2 = __shfl(level, l2), t3 = __shfl(level, l3), t4 = __shfl(level, l4);
      if (diag4 == dg && inRange)
      {
        int sumAbs = 0, numPos = 0, sumGo = 0;
        auto upd = [&](bool valid, int a) { if (valid) { sumAbs += min(4 - (a & 1), a); numPos += a != 0; sumGo += a - (a != 0); } };
        upd(v0, in0 ? t0 : nb0); upd(v1, in1 ? t1 : nb1); upd(v2, in2 ? t2 : nb2); upd(v3, in3 ? t3 : nb3); upd(v4, in4 ? t4 : nb4);
        int ctxSig = 0, ofs = 0;
        if (!isLast)
        {
          const int diag = x + y;
          ctxSig = min(sumAbs, 5) + (diag < 2 ? 6 : 0) + ((luma && diag < 5) ? 6 : 0);
          ofs = min(sumAbs - numPos, 4) + 1 + (diag == 0 ? (luma ? 15 : 5) : (luma ? (diag < 3 ? 10 : (diag < 10 ? 5 : 0)) : 0));
        }
        const int sm = min(sumGo, 31), rice = sm < 12 ? 0 : sm < 25 ? 1 : 2;                       // g_auiGoRicePars
        const RdoqBits b = { rt->par[ofs][0], rt->par[ofs][1], rt->gt1[ofs][0], rt->gt1[ofs][1], rt->gt2[ofs][0], rt->gt2[ofs][1] };
        const int sig0 = rt->sig[ctxSig][0], sig1 = rt->sig[ctxSig][1];
        // xGetCodedLevel :107-162
        double codedCost; unsigned best = 0; bool done = false;
        if (!isLast && maxAbs < 3)
        {
          costSig = lambda * sig0;
          codedCost = cost0 + costSig;
          done = maxAbs == 0;
        }
        else codedCost = 1.7976931348623157e308;
        if (!done)
        {
          const double currSig = isLast ? 0.0 : lambda * sig1;
          const int minAbs = maxAbs > 1 ? (int)maxAbs - 1 : 1;
          for (int a = (int)maxAbs; a >= minAbs; a--)
          {
            const double err = (double)(levelDouble - (int)((unsigned)a << qBits));
            double cost = err * err * errScale + lambda * rdoq_ic_rate((unsigned)a, b, rice);
            cost += currSig;
            if (cost < codedCost) { best = (unsigned)a; codedCost = cost; costSig = currSig; }
          }
        }
        costCoeff = codedCost;
        if (!isLast) sigDelta = sig1 - sig0;
        deltaU = (levelDouble - (int)(best << qBits)) >> (qBits - 8);
        if (best > 0)
        {
          const int now = rdoq_ic_rate(best, b, rice);
          incUp = rdoq_ic_rate(best + 1, b, rice) - now;
          incDown = rdoq_ic_rate(best - 1, b, rice) - now;
        }
        else incUp = b.par0 + b.gt10;
        level = (int)best;
      }
    }
    // ---- the running sums of the group, scan order fifteen down to zero :1023-1041
    double sigCost = 0, sigCost0 = 0, codedLevelAndDist = 0, uncodedDist = 0; int nnzBeforePos0 = 0;
    const unsigned nzMask = (unsigned)(__ballot(level != 0) >> tb) & 0xFFFFu;
    for (int kk = 15; kk >= 0; kk--)
    {
      const double cc = rdoq_shfl(costCoeff, tb + kk), c0 = rdoq_shfl(cost0, tb + kk), cs = rdoq_shfl(costSig, tb + kk);
      blockUncoded += c0;
      baseCost += (lastScanPos >= 0 && (subSet << 4) + kk <= lastScanPos) ? cc : c0;
      sigCost += cs;
      if (kk == 0) sigCost0 = cs;
      if ((nzMask >> kk) & 1u) { codedLevelAndDist += cc - cs; uncodedDist += c0; if (kk != 0) nnzBeforePos0++; }
    }
    bool sigGroup = nzMask != 0;
    double cgSig = 0;
    if (cgLastScanPos >= 0)
    {
      if (subSet)
      {
        if (!sigGroup)
        {
          baseCost += lambda * sg0 - sigCost;
          cgSig = lambda * sg0;
        }
        else if (subSet < cgLastScanPos)
        {
          if (nnzBeforePos0 == 0) { baseCost -= sigCost0; sigCost -= sigCost0; }
          double costZeroCG = baseCost;
          baseCost += lambda * sg1;
          costZeroCG += lambda * sg0;
          cgSig = lambda * sg1;
          costZeroCG += uncodedDist;
          costZeroCG -= codedLevelAndDist;
          costZeroCG -= sigCost;
          if (costZeroCG < baseCost)
          {
            sigGroup = false;
            baseCost = costZeroCG;
            cgSig = lambda * sg0;
            if (level) { level = 0; costCoeff = cost0; costSig = 0; }
          }
        }
      }
      else sigGroup = true;
    }
    dst[pos] = level;
    wCoeff[sp] = costCoeff; wSig[sp] = costSig; wCoeff0[sp] = cost0;
    wUp[sp] = incUp; wDown[sp] = incDown; wSigDelta[sp] = sigDelta; wDeltaU[sp] = deltaU;
    if (k == 0) { wSG[cgPos] = sigGroup ? 1 : 0; wCG[subSet] = cgSig; }
    __threadfence_block();                                                 // levels and group flags are read by other lanes of the team later
  }
  if (lastScanPos < 0) { if (k == 0) absSumOut[ti] = 0; return; }

  // ---- last position :1127-1262 (serial chain on the base cost; ends at the first level above one)
  double bestCost = blockUncoded + lambda * rt->cbf[0];
  baseCost += lambda * rt->cbf[1];
  int bestLastIdxP1 = 0;
  bool foundLast = false;
  for (int cg = cgLastScanPos; cg >= 0 && !foundLast; cg--)
  {
    baseCost -= wCG[cg];
    const int sp = (cg << 4) + k, pos = scan[sp], px = pos & (w - 1), py = pos >> lw;
    if (!wSG[(py >> 2) * wig + (px >> 2)]) continue;
    const int lvl = dst[pos];
    const double cc = wCoeff[sp], cs = wSig[sp], c0 = wCoeff0[sp];
    const int gx = px < 4 ? px : (2 * (31 - __clz(px))) + ((px >> (30 - __clz(px))) & 1), gy = py < 4 ? py : (2 * (31 - __clz(py))) + ((py >> (30 - __clz(py))) & 1);   // g_uiGroupIdx
    double rl = rt->last_x[gx] + rt->last_y[gy];                                    // xGetRateLast :407-421
    if (gx > 3) rl += 32768.0 * ((gx - 2) >> 1);
    if (gy > 3) rl += 32768.0 * ((gy - 2) >> 1);
    const double costLast = lambda * rl;
    for (int kk = 15; kk >= 0; kk--)
    {
      const int l = __shfl(lvl, tb + kk);
      const double cck = rdoq_shfl(cc, tb + kk), csk = rdoq_shfl(cs, tb + kk), c0k = rdoq_shfl(c0, tb + kk), clk = rdoq_shfl(costLast, tb + kk);
      if ((cg << 4) + kk > lastScanPos) continue;
      if (l)
      {
        const double total = baseCost + clk - csk;
        if (total < bestCost) { bestLastIdxP1 = (cg << 4) + kk + 1; bestCost = total; }
        if (l > 1) { foundLast = true; break; }
        baseCost -= cck;
        baseCost += c0k;
      }
      else baseCost -= csk;
    }
  }

  // ---- signs, the positions beyond the chosen last one, the sum of levels :1263-1276
  unsigned absSum = 0;
  for (int cg = 0; cg <= cgLastScanPos; cg++)
  {
    const int sp = (cg << 4) + k, pos = scan[sp];
    const int lvl = sp < bestLastIdxP1 ? dst[pos] : 0;
    absSum += (unsigned)lvl;
    dst[pos] = src[pos] < 0 ? -lvl : lvl;
  }
  for (int m = 1; m < 16; m <<= 1) absSum += (unsigned)__shfl_xor((int)absSum, m);
  if (k == 0) absSumOut[ti] = absSum;

  // ---- sign bit hiding :1278-1406: every group on its own; lane = candidate position
  if (!d.sign_hiding || (int)absSum < 2) return;
  const double inv = rem == 0 ? 40.0 : rem == 1 ? 45.0 : rem == 2 ? 51.0 : rem == 3 ? 57.0 : rem == 4 ? 64.0 : 72.0;               // g_invQuantScales
  const long long rdFactor = (long long)(inv * inv * (1 << (2 * per)) / lambda / 16 / 1 + 0.5);
  int lastCG = -1;
  for (int subSet = cgLastScanPos; subSet >= 0; subSet--)
  {
    const int sp = (subSet << 4) + k, pos = scan[sp];
    const int lvl = dst[pos];
    const unsigned m = (unsigned)(__ballot(lvl != 0) >> tb) & 0xFFFFu;
    const int lastNZ = m ? 31 - __clz((int)m) : -1, firstNZ = m ? __ffs((int)m) - 1 : 16;
    int sum = lvl;
    for (int s = 1; s < 16; s <<= 1) sum += __shfl_xor(sum, s);
    if (lastNZ >= 0 && lastCG == -1) lastCG = 1;
    if (lastNZ - firstNZ >= 4)
    {
      const unsigned signbit = __shfl(lvl, tb + firstNZ) > 0 ? 0u : 1u;
      if (signbit != (unsigned)(sum & 1))
      {
        const long long MAXC = 0x7FFFFFFFFFFFFFFFll;
        long long curCost = MAXC; int curChange = 0;
        if (k <= (lastCG == 1 ? lastNZ : 15))
        {
          const int dU = wDeltaU[sp], up = wUp[sp], down = wDown[sp], sd = wSigDelta[sp];
          if (lvl != 0)
          {
            const long long costUp = rdFactor * (-dU) + up;
            long long costDown = rdFactor * dU + down - (abs(lvl) == 1 ? sd : 0);
            if (lastCG == 1 && lastNZ == k && abs(lvl) == 1) costDown -= 4 << 15;
            if (costUp < costDown) { curCost = costUp; curChange = 1; }
            else { curChange = -1; curCost = (k == firstNZ && abs(lvl) == 1) ? MAXC : costDown; }
          }
          else
          {
            curCost = rdFactor * (-(long long)abs(dU)) + (1 << 15) + up + sd;
            curChange = 1;
            if (k < firstNZ && (src[pos] >= 0 ? 0u : 1u) != signbit) curCost = MAXC;
          }
        }
        // minimum cost; among equals the position visited first (the highest) stays
        long long bc = curCost; int bk = k;
        for (int s = 1; s < 16; s <<= 1)
        {
          const long long oc = __shfl_xor(bc, s); const int ok = __shfl_xor(bk, s);
          if (oc < bc || (oc == bc && ok > bk)) { bc = oc; bk = ok; }
        }
        if (bk == k && bc != MAXC)
        {
          int change = curChange;
          if (lvl == 32767 || lvl == -32768) change = -1;
          dst[pos] = src[pos] >= 0 ? lvl + change : lvl - change;
        }
      }
    }
    if (lastCG == 1) lastCG = 0;
  }
}

static bool g_tablesUploaded[64] = { false };
static std::mutex g_tablesMutex;            // the C ABI may be entered from several host threads: one uploads, the others wait
constexpr int g_smallGrid = 1280;                  // workgroups of the small-TU kernels (swept in round 2)

// diagonal 4x4-grouped coefficient scan (Rom.cpp:357-405): groups of 4x4 (2x2 when a side is 2) visited along the diagonals
// x + y = d from the bottom-left end upwards, the positions inside a group likewise
static void host_scan_order(int w, int h, uint16_t* out)
{
  const int lg = ((w & 3) + (h & 3)) > 0 ? 1 : 2, g = 1 << lg, gwN = w >> lg, ghN = h >> lg;
  int n = 0;
  for (int D = 0; D < gwN + ghN - 1; D++)
    for (int gy = (D < ghN - 1 ? D : ghN - 1); gy >= 0; gy--)
    {
      const int gx = D - gy;
      if (gx >= gwN) continue;
      for (int dd = 0; dd < 2 * g - 1; dd++)
        for (int y = (dd < g - 1 ? dd : g - 1); y >= 0; y--)
        {
          const int x = dd - y;
          if (x < g) out[n++] = (uint16_t)((gy * g + y) * w + gx * g + x);
        }
    }
}

static int ensure_tables()
{
  int dev = 0;
  VVC_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) { vvcgpu_set_error("device index %d out of range", dev); return VVCGPU_E_DEVICE; }
  std::lock_guard<std::mutex> lock(g_tablesMutex);
  if (!g_tablesUploaded[dev])
  {
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
    static uint16_t scan[15876];
    int off[36], o = 0;
    for (int a = 0; a < 6; a++)
      for (int b = 0; b < 6; b++) { off[a * 6 + b] = o; host_scan_order(2 << a, 2 << b, scan + o); o += (2 << a) * (2 << b); }
    VVC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(d_scan), scan, sizeof(scan)));
    VVC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(d_scanOff), off, sizeof(off)));
    // dependent quantisation: raster -> scan id
    static uint16_t invs[15876];
    for (int a = 0; a < 6; a++)
      for (int b = 0; b < 6; b++)
      {
        const int N = (2 << a) * (2 << b), o0 = off[a * 6 + b];
        for (int i = 0; i < N; i++) invs[o0 + scan[o0 + i]] = (uint16_t)i;
      }
    // depquant_kernel keeps the ancestry of a trellis path as 32 two-bit entries: the sub-block a template read goes to (right of / below / below-right of the
    // sub-block about to be walked) must lie at most 32 sub-blocks back in the scan from the one that just ended.  True for every shape up to 64x64 (30 for 64x64);
    // checked here so that a larger transform size cannot pass silently.
    for (int a = 1; a < 6; a++)
      for (int b = 1; b < 6; b++)
      {
        const int W = 2 << a, H = 2 << b, o0 = off[a * 6 + b], wS = W >> 2, hS = H >> 2;
        for (int n = 0; n + 1 < wS * hS; n++)                     // n: the sub-block about to be walked, n + 1 the one that just ended
        {
          const int p = scan[o0 + 16 * n], sx = (p % W) >> 2, sy = (p / W) >> 2;
          const int cand[3][2] = { { sx + 1, sy }, { sx, sy + 1 }, { sx + 1, sy + 1 } };
          for (auto& c : cand)
            if (c[0] < wS && c[1] < hS)
            {
              const int j = invs[o0 + (4 * c[1]) * W + 4 * c[0]] >> 4;
              if (j - (n + 1) - 1 > 31) { vvcgpu_set_error("depquant tables: a template reaches %d sub-blocks back in a %dx%d TU", j - n - 2, W, H); return VVCGPU_E_UNSUPPORTED; }
            }
        }
      }
    // the shape-only part of the trellis' position records (depquant_kernel, fillRec)
    static uint4 psel[15876];
    static uint2 pmisc[15876];
    for (int a = 0; a < 6; a++)
      for (int b = 0; b < 6; b++)
      {
        const int W = 2 << a, H = 2 << b, N = W * H, o0 = off[a * 6 + b];
        for (int si = 0; si < N; si++)
        {
          const int sn = si > 0 ? si - 1 : 0, p2 = scan[o0 + sn], x2 = p2 % W, y2 = p2 / W, beg = sn & ~15;
          const int cx[5] = { x2 + 1, x2 + 2, x2 + 1, x2, x2 }, cy[5] = { y2, y2, y2 + 1, y2 + 1, y2 + 2 };
          unsigned nb = 0, selLo[2] = { 0x0C0C0C0Cu, 0x0C0C0C0Cu }, selHi[2] = { 0x0C0C0C0Cu, 0x0C0C0C0Cu };
          for (int t = 0; t < 5; t++)
          {
            const int r = (cx[t] < W && cy[t] < H) ? (int)invs[o0 + cy[t] * W + cx[t]] - beg : 0;
            const unsigned rel = (r > 0 && r < 16) ? (unsigned)r : 0u, sh = (unsigned)(t & 3) * 8u, m = 0xFFu << sh;
            nb |= rel << (4 * t);
            if (rel != 0u && rel < 8u) selLo[t >> 2] = (selLo[t >> 2] & ~m) | (rel << sh);
            if (rel >= 8u) selHi[t >> 2] = (selHi[t >> 2] & ~m) | ((rel - 8u) << sh);
          }
          const int diag = x2 + y2;
          const unsigned sigL = diag < 2 ? 12 : diag < 5 ? 6 : 0, sigC = diag < 2 ? 6 : 0;
          const unsigned gtxL = diag < 1 ? 16 : diag < 3 ? 11 : diag < 10 ? 6 : 1, gtxC = diag < 1 ? 6 : 1;
          psel[o0 + si] = make_uint4(selLo[0], selHi[0], selLo[1], selHi[1]);
          pmisc[o0 + si] = make_uint2(nb | sigL << 20 | gtxL << 24, nb | sigC << 20 | gtxC << 24);
        }
      }
    VVC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(d_dqPosSel), psel, sizeof(psel)));
    VVC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(d_dqPosMisc), pmisc, sizeof(pmisc)));
    VVC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(d_dqInv), invs, sizeof(invs)));
    g_tablesUploaded[dev] = true;
  }
  return VVCGPU_OK;
}

// VVCGPU_NO_MFMA=1 (common.h) keeps every large TU on the dot2 kernels
static int tr_use_mfma() { return vvcgpu_no_mfma() ? 0 : 1; }

static int check_descs_args(const void* a, const void* b, const void* d, int n, int bd, const char* who)
{
  VVC_CHECK_ARG(n >= 0, "%s: n %d", who, n);
  if (n == 0) return 1;
  VVC_CHECK_ARG(a && b && d, "%s: null pointer", who);
  if (bd < 8 || bd > 10) { vvcgpu_set_error("%s: bit depth %d outside 8..10", who, bd); return VVCGPU_E_UNSUPPORTED; }
  return VVCGPU_OK;
}

}  // namespace

// device addresses of the golden tables on the current device, for the kernels of other translation units (resichain.hip)
int vvcgpu_tr_tables(VvcTrTables* out)
{
  const int rt = ensure_tables();
  if (rt) return rt;
  void* p = nullptr;
  VVC_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(d_tr32)));      out->tr32 = static_cast<const int*>(p);
  VVC_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(d_tr32t)));     out->tr32t = static_cast<const int*>(p);
  VVC_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(d_dqInv)));     out->dqInv = static_cast<const unsigned short*>(p);
  VVC_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(d_scanOff)));   out->scanOff = static_cast<const int*>(p);
  return VVCGPU_OK;
}

extern "C" {

int vvcgpu_tr_fwd_batch(const vvc_pel* resi_base, vvc_coef* coeff_base, const vvcgpu_tr_desc* descs, int n,
                        int bit_depth, void* stream)
{
  const int rc = check_descs_args(resi_base, coeff_base, descs, n, bit_depth, "tr_fwd_batch");
  if (rc) return rc > 0 ? VVCGPU_OK : rc;
  const int rt = ensure_tables();
  if (rt) return rt;
  hipStream_t st = (hipStream_t)stream;
  VvcTrTables tb;
  const int rtb = vvcgpu_tr_tables(&tb);
  if (rtb) return rtb;
  const _Float16* image = vvcgpu_mfma_image(tb);
  if (!image) return VVCGPU_E_DEVICE;
  // long calls: ONE launch of the residual chain's bodies in forward-only mode (packed matrix-core tiles for TUs with a 4- / 8-point side, lane groups
  // for 8x8 and smaller) instead of the small / matrix-core / dot2 kernels in a row -- on a real encoder's call mix those three were each bound by
  // their own per-wave latency (profiles/r04_shape_mix.txt)
  if (n >= 16384 && tr_use_mfma()) return vvcgpu_tr_chain_launch(1, resi_base, nullptr, coeff_base, descs, n, bit_depth, stream);
  // long calls: the small TUs are binned on the device as well and the small kernel walks the bin lists (see small_setup)
  const bool ordered = n >= 16384;
  int* ws = static_cast<int*>(vvcgpu_scratch(st, sizeof(int) * ((ordered ? 6 : 2) * (size_t)n + 2)));   // cached per-stream scratch: the lists of large (and small) TUs
  if (!ws) return VVCGPU_E_DEVICE;
  VVC_HIP(hipMemsetAsync(ws, 0, 2 * sizeof(int), st));
  int* smCnt = nullptr; int* smLists = nullptr; int* nextCnt = nullptr;
  if (ordered)
  {
    int cur = 0;
    int* counters = vvcgpu_counters(st, &cur);
    if (!counters) return VVCGPU_E_DEVICE;
    smCnt = counters + VVC_CTR_INTS * cur; nextCnt = counters + VVC_CTR_INTS * (cur ^ 1); smLists = ws + 2 + 2 * (size_t)n;
  }
  const int nb = cdiv(n, SM_DESCS), nl = cdiv(n, 4);
  hipLaunchKernelGGL(tr_collect_large_kernel, dim3(cdiv(n, 1024)), dim3(1024), 0, st, descs, n, ws, tr_use_mfma(), smCnt, smLists, nextCnt);
  hipLaunchKernelGGL(tr_fwd_small_kernel, dim3(nb < g_smallGrid ? nb : g_smallGrid), dim3(256), 0, st, resi_base, coeff_base, descs, n, bit_depth, tr_use_mfma(), smCnt, smLists);
  hipLaunchKernelGGL(tr_fwd_mfma_kernel, dim3(nl < 768 ? nl : 768), dim3(256), 0, st, resi_base, coeff_base, descs, ws, ws + 2 + n, ws + 1, bit_depth, image);
  hipLaunchKernelGGL(tr_fwd_large_kernel, dim3(nl < 768 ? nl : 768), dim3(256), 0, st, resi_base, coeff_base, descs, ws + 1, bit_depth);
  if (ordered) VVC_LAUNCH_CHECK_COUNTERS(st);
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
  hipStream_t st = (hipStream_t)stream;
  VvcTrTables tb;
  const int rtb = vvcgpu_tr_tables(&tb);
  if (rtb) return rtb;
  const _Float16* image = vvcgpu_mfma_image(tb);
  if (!image) return VVCGPU_E_DEVICE;
  // long calls: the residual chain's bodies in inverse-only mode (see vvcgpu_tr_fwd_batch)
  // (measured on the real call mix: 0.048 vs 0.054 ms at 35 k TUs, 0.126 vs 0.089 at 141 k -- there the three kernels' own latencies are amortised and their
  // lane-group forms run at five waves per SIMD against the chain kernel's three)
  if (n >= 16384 && n < 65536 && tr_use_mfma()) return vvcgpu_tr_chain_launch(2, nullptr, resi_base, const_cast<vvc_coef*>(coeff_base), descs, n, bit_depth, stream);
  // long calls: the small TUs are binned on the device as well and the small kernel walks the bin lists (see small_setup)
  const bool ordered = n >= 16384;
  int* ws = static_cast<int*>(vvcgpu_scratch(st, sizeof(int) * ((ordered ? 6 : 2) * (size_t)n + 2)));   // cached per-stream scratch: the lists of large (and small) TUs
  if (!ws) return VVCGPU_E_DEVICE;
  VVC_HIP(hipMemsetAsync(ws, 0, 2 * sizeof(int), st));
  int* smCnt = nullptr; int* smLists = nullptr; int* nextCnt = nullptr;
  if (ordered)
  {
    int cur = 0;
    int* counters = vvcgpu_counters(st, &cur);
    if (!counters) return VVCGPU_E_DEVICE;
    smCnt = counters + VVC_CTR_INTS * cur; nextCnt = counters + VVC_CTR_INTS * (cur ^ 1); smLists = ws + 2 + 2 * (size_t)n;
  }
  const int nb = cdiv(n, SM_DESCS), nl = cdiv(n, 4);
  hipLaunchKernelGGL(tr_collect_large_kernel, dim3(cdiv(n, 1024)), dim3(1024), 0, st, descs, n, ws, tr_use_mfma(), smCnt, smLists, nextCnt);
  hipLaunchKernelGGL(tr_inv_small_kernel, dim3(nb < g_smallGrid ? nb : g_smallGrid), dim3(256), 0, st, coeff_base, resi_base, descs, n, bit_depth, tr_use_mfma(), smCnt, smLists);
  hipLaunchKernelGGL(tr_inv_mfma_kernel, dim3(nl < 768 ? nl : 768), dim3(256), 0, st, coeff_base, resi_base, descs, ws, ws + 2 + n, ws + 1, bit_depth, image);
  hipLaunchKernelGGL(tr_inv_large_kernel, dim3(nl < 768 ? nl : 768), dim3(256), 0, st, coeff_base, resi_base, descs, ws + 1, bit_depth);
  if (ordered) VVC_LAUNCH_CHECK_COUNTERS(st);
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}

int vvcgpu_dequant_tr_inv_batch(const vvc_coef* level_base, vvc_pel* resi_base, const vvcgpu_dqtr_desc* descs, int n,
                                int bit_depth, vvc_coef* coeff_out, void* stream)
{
  static_assert(sizeof(vvcgpu_dqtr_desc) == sizeof(vvcgpu_tr_desc), "descriptor layouts must stay interchangeable");
  const int rc = check_descs_args(level_base, resi_base, descs, n, bit_depth, "dequant_tr_inv_batch");
  if (rc) return rc > 0 ? VVCGPU_OK : rc;
  VVC_CHECK_ARG(coeff_out != level_base, "dequant_tr_inv_batch: coeff_out must not alias the levels");
  const int rt = ensure_tables();
  if (rt) return rt;
  hipStream_t st = (hipStream_t)stream;
  VvcTrTables tb;
  const int rtb = vvcgpu_tr_tables(&tb);
  if (rtb) return rtb;
  const _Float16* image = vvcgpu_mfma_image(tb);
  if (!image) return VVCGPU_E_DEVICE;
  // descriptors per workgroup: 64 when the batch is long (lane-group TUs need many per wave), fewer when that would leave compute units idle
  int per = SM_DESCS;
  while (per > 4 && cdiv(n, per) < 1024) per >>= 1;
  const int nb = cdiv(n, per);
  // calls long enough to fill workgroups with one phase each are put into class order on the device first (dqtr_classify_kernel)
  const bool ordered = n >= 16384;                                           // shorter calls: the extra launch (~15 us) costs more than the order gains
  if (ordered)
  {
    int* lists = static_cast<int*>(vvcgpu_scratch(st, (size_t)DQC_NCLS * n * sizeof(int)));
    if (!lists) return VVCGPU_E_DEVICE;
    int cur = 0;
    int* counters = vvcgpu_counters(st, &cur);
    if (!counters) return VVCGPU_E_DEVICE;
    int* hdr = counters + VVC_CTR_INTS * cur;
    hipLaunchKernelGGL(dqtr_classify_kernel, dim3(n < 1024 * DQC_WGS ? cdiv(n, 1024) : DQC_WGS), dim3(1024), 0, st, descs, n, hdr, lists,
                       counters + VVC_CTR_INTS * (cur ^ 1), tr_use_mfma());
    VVC_LAUNCH_CHECK_COUNTERS(st);
    hipLaunchKernelGGL(dqtr_fused_kernel, dim3(nb < 512 ? nb : 512), dim3(256), 0, st, level_base, resi_base, descs, n, per, bit_depth, coeff_out, image,
                       tr_use_mfma(), hdr, lists);
    VVC_LAUNCH_CHECK_COUNTERS(st);
    return VVCGPU_OK;
  }
  hipLaunchKernelGGL(dqtr_fused_kernel, dim3(nb < 512 ? nb : 512), dim3(256), 0, st, level_base, resi_base, descs, n, per, bit_depth, coeff_out, image,
                     tr_use_mfma(), nullptr, nullptr);
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}

int vvcgpu_quant_batch(const vvc_coef* coeff_base, vvc_coef* level_base, const vvcgpu_quant_desc* descs, int n, int bit_depth, uint32_t* abs_sum,
                       void* stream)
{
  VVC_CHECK_ARG(n >= 0, "quant_batch: n %d", n);
  if (n == 0) return VVCGPU_OK;
  VVC_CHECK_ARG(coeff_base && level_base && descs && abs_sum, "quant_batch: null pointer");
  VVC_CHECK_ARG(bit_depth >= 8 && bit_depth <= 10, "quant_batch: bit depth %d outside 8..10", bit_depth);
  const int rt = ensure_tables();
  if (rt) return rt;
  hipStream_t st = (hipStream_t)stream;
  int* list = static_cast<int*>(vvcgpu_scratch(st, ((size_t)n + 1) * sizeof(int)));
  if (!list) return VVCGPU_E_DEVICE;
  VVC_HIP(hipMemsetAsync(list, 0, sizeof(int), st));
  hipLaunchKernelGGL(quant_small_kernel, dim3(cdiv(n, 16)), dim3(256), 0, st, coeff_base, level_base, descs, n, bit_depth, abs_sum, list);
  const int nl = cdiv(n, 4);
  hipLaunchKernelGGL(quant_large_kernel, dim3(nl < 512 ? nl : 512), dim3(256), 0, st, coeff_base, level_base, descs, bit_depth, abs_sum, list);
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}

size_t vvcgpu_depquant_workspace_bytes(size_t total_coeffs, int n)
{
  (void)n;
  const size_t c = (total_coeffs + 15) & ~(size_t)15;
  return c * 16 + c * 8 + 256;                             // decisions (4 x u32 per position) + 8 level histories per TU
}

int vvcgpu_depquant_batch(const vvc_coef* coeff_base, vvc_coef* level_base, const vvcgpu_depquant_desc* descs, int n,
                          const vvcgpu_dq_rates* rates, int bit_depth, uint32_t* abs_sum, size_t total_coeffs, void* ws, size_t ws_bytes,
                          void* stream)
{
  VVC_CHECK_ARG(n >= 0, "depquant_batch: n %d", n);
  if (n == 0) return VVCGPU_OK;
  VVC_CHECK_ARG(coeff_base && level_base && descs && rates && abs_sum && ws, "depquant_batch: null pointer");
  VVC_CHECK_ARG(bit_depth >= 8 && bit_depth <= 10, "depquant_batch: bit depth %d outside 8..10", bit_depth);
  VVC_CHECK_ARG(total_coeffs >= 16 && ws_bytes >= vvcgpu_depquant_workspace_bytes(total_coeffs, n) && (reinterpret_cast<uintptr_t>(ws) & 15) == 0,
                "depquant_batch: workspace of %zu bytes for %zu coefficients is too small (need %zu) or unaligned", ws_bytes, total_coeffs,
                vvcgpu_depquant_workspace_bytes(total_coeffs, n));
  const int rt = ensure_tables();
  if (rt) return rt;
  // the workspace is split as vvcgpu_depquant_workspace_bytes lays it out: c * 16 bytes of decisions, then the level histories
  const size_t c = (total_coeffs + 15) & ~(size_t)15;
  unsigned* dec = static_cast<unsigned*>(ws);
  unsigned char* ctx = static_cast<unsigned char*>(ws) + c * 16;
  VVC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(depquant_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, DQ_LDS_BYTES));
  hipLaunchKernelGGL(depquant_kernel, dim3(cdiv(n, 64)), dim3(256), DQ_LDS_BYTES, (hipStream_t)stream, coeff_base, level_base, descs, n, rates, bit_depth,
                     abs_sum, dec, ctx);
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}

size_t vvcgpu_rdoq_workspace_bytes(size_t total_coeffs, int n)
{
  (void)n;
  const size_t c = (total_coeffs + 15) & ~(size_t)15;
  return c * 24 + c * 16 + (c >> 4) * 8 + (c >> 4) + 256;  // three cost arrays, four rate-delta arrays, per group: flag cost + flag
}

int vvcgpu_rdoq_batch(const vvc_coef* coeff_base, vvc_coef* level_base, const vvcgpu_rdoq_desc* descs, int n,
                      const vvcgpu_rdoq_rates* rates, int bit_depth, uint32_t* abs_sum, size_t total_coeffs, void* ws, size_t ws_bytes,
                      void* stream)
{
  VVC_CHECK_ARG(n >= 0, "rdoq_batch: n %d", n);
  if (n == 0) return VVCGPU_OK;
  VVC_CHECK_ARG(coeff_base && level_base && descs && rates && abs_sum && ws, "rdoq_batch: null pointer");
  VVC_CHECK_ARG(bit_depth >= 8 && bit_depth <= 10, "rdoq_batch: bit depth %d outside 8..10", bit_depth);
  VVC_CHECK_ARG(total_coeffs >= 16 && ws_bytes >= vvcgpu_rdoq_workspace_bytes(total_coeffs, n) && (reinterpret_cast<uintptr_t>(ws) & 15) == 0,
                "rdoq_batch: workspace of %zu bytes for %zu coefficients is too small (need %zu) or unaligned", ws_bytes, total_coeffs,
                vvcgpu_rdoq_workspace_bytes(total_coeffs, n));
  const int rt = ensure_tables();
  if (rt) return rt;
  const size_t c = (total_coeffs + 15) & ~(size_t)15;
  unsigned char* base = static_cast<unsigned char*>(ws);
  double* wsD = reinterpret_cast<double*>(base);
  int* wsI = reinterpret_cast<int*>(base + c * 24);
  double* wsCG = reinterpret_cast<double*>(base + c * 40);
  unsigned char* wsSG = base + c * 40 + (c >> 4) * 8;
  // (a group flag is always written before a left / upper neighbour group reads it: the workspace needs no clearing)
  hipLaunchKernelGGL(rdoq_kernel, dim3(cdiv(n, 16)), dim3(256), 0, (hipStream_t)stream, coeff_base, level_base, descs, n, rates, bit_depth,
                     abs_sum, wsD, wsI, wsCG, wsSG, c);
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}

int vvcgpu_scan_order_host(int w, int h, uint16_t* out)
{
  if (!out || w < 2 || w > 64 || h < 2 || h > 64 || (w & (w - 1)) || (h & (h - 1))) return VVCGPU_E_ARG;
  host_scan_order(w, h, out);
  return VVCGPU_OK;
}

const int16_t* vvcgpu_tr_matrix_host(int type, int n)
{
  if (type < 0 || type > 2 || n < 2 || n > 64 || (n & (n - 1))) return nullptr;
  return VVC_TR_TABLES + type * 5460 + (n * n - 4) / 3;
}

}  // extern "C"
