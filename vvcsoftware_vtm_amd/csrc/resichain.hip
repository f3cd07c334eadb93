// resichain.hip -- the residual chain of a TU in one pass (vvcgpu_resi_chain_batch):
//   residual = org - pred -> forward transform (xTrMxN_EMT, TrQuant.cpp:138-220) -> Quant::quant with sign bit hiding (Quant.cpp:721-834,
//   :142-273) -> Quant::dequant (Quant.cpp:277-428) -> inverse transform (xITrMxN_EMT, TrQuant.cpp:238-310) -> clip(pred + resi')
// i.e. what InterSearch::xEstimateInterResidualQT does per TU between :4409 (transformNxN) and :4504 (distortion of the reconstruction).
// The separate entry points (vvcgpu_pelop_batch / tr_fwd / quant / dequant_tr_inv / pelop) move the residual, the coefficients and the
// de-quantised coefficients through HBM five times; here they stay in registers / LDS, the levels and the reconstruction are written once.
//
// One chain launch (+ a generic launch for what it cannot take) behind the entry point, fed by a device-side classification of the descriptor list:
//   * both sides in 16 / 32 / 64: ONE WAVE PER TU, all four 1-D stages on the matrix cores.  v_mfma_f32_16x16x32_f16 accumulates in f32, which is
//     exact for integers below 2^24: the matrix entries (|c| <= 362) and the residual (|x| <= 1023) are exact f16 values and a 64-term row sum
//     stays below 2^24; the 16-bit intermediates of the later stages are split into two signed 8-bit limbs (t = 256 hi + lo), one MFMA chain per
//     limb, recombined in int32 with the reference's rounding shift and clipping.  The result tile of one stage is the operand of the next
//     WITHOUT leaving the lane: a 16x16 result has its column on the lane and four consecutive rows in registers, the next product sums over
//     that row index, and the k order of an MFMA is free as long as both operands agree -- so the matrix operand is read from LDS in the
//     k order the result registers already have.  (forward: M1 = X Th^T, C = Tv M1; inverse: Y1^T = Cq^T Tv, R^T = Th^T Y1^T: every product
//     sums over the row index of the previous result.)
//   * a 16- / 32- / 64-point side with an 8- or 4-point one (round 4): PACKED tiles -- two or four TUs share one matrix-core multi-tile, the short
//     stages as block-diagonal products (rc_tile_packed, rc_tile_packed_wl / _hl below).
//   * 8 x 8, 8 x 4, 4 x 8, 4 x 4: lane groups of 8 / 4 lanes per TU (8 / 16 TUs per wave), integer multiply-adds, the two transposes through wave-private LDS.
//   * what is left (2-wide chroma TUs; TUs whose residual leaves +-1023): generic wave-per-TU path through LDS buffers (correct for every W x H in 2..64, slow).
// Every body takes a compile-time MODE: the chain, or its forward / inverse half alone -- vvcgpu_tr_fwd_batch / vvcgpu_tr_inv_batch run long calls through
// the same kernel (vvcgpu_tr_chain_launch).
// The quantiser works in the layout all of them produce -- a lane holds four vertically consecutive coefficients of one column, an aligned quad
// of lanes holds a 4x4 coefficient group: sign bit hiding is decided per quad with DPP quad permutes.
#include "common.h"
#include "mfma_tr.h"
#include <mutex>

namespace {
// Pointer arguments of functions that are NOT inlined arrive as generic pointers: every access through them is a FLAT instruction (both wait counters, no
// immediate offsets, LDS through the aperture).  This assumption lets the compiler infer the LDS address space again inside the callee (the pattern for global pointers,
// !is_shared & !is_private, does not survive the optimiser's De Morgan rewrite; global data through flat loads costs little).
#if defined(__HIP_DEVICE_COMPILE__)
#define RC_IS_LDS(p) __builtin_assume(__builtin_amdgcn_is_shared((const __attribute__((address_space(0))) void*)(p)))
#else
#define RC_IS_LDS(p) ((void)0)
#endif


typedef vvcgpu_resi_chain_desc RcDesc;

__device__ __forceinline__ int ilog2(int v) { return 31 - __clz(v); }
#define RC_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

// ---- work lists (device): hdr[0 .. RC_NCLS) = counts of the classes, hdr[RC_FB] = count of the fall-back list
constexpr int RC_HDR = VVC_CTR_INTS;
enum { RC_C64 = 0, RC_C32, RC_C16, RC_C8, RC_C4, RC_CGEN, RC_R6432, RC_R3264, RC_R6416, RC_R1664, RC_R3216, RC_R1632, RC_R84, RC_R48,
       RC_P168, RC_P816, RC_P164, RC_P416, RC_P328, RC_P832, RC_P324, RC_P432, RC_P648, RC_P864, RC_P644, RC_P464, RC_NCLS };   // RC_P*: packed tiles (rc_tile_packed*)
constexpr int RC_FB = RC_NCLS;
static_assert(RC_FB < RC_HDR, "the header is one counter set of vvcgpu_counters");

__host__ __device__ __forceinline__ int rc_class(const RcDesc& d, bool packed)
{
  const int w = d.w, h = d.h;
  if (w == h)
  {
    if (w == 64) return RC_C64;
    if (w == 32) return RC_C32;
    if (w == 16) return RC_C16;
    if (w == 8) return RC_C8;
    if (w == 4) return RC_C4;
    return RC_CGEN;
  }
  if (w == 64) return h == 32 ? RC_R6432 : h == 16 ? RC_R6416 : h == 8 ? (packed ? RC_P648 : RC_CGEN) : h == 4 ? (packed ? RC_P644 : RC_CGEN) : RC_CGEN;
  if (w == 32) return h == 64 ? RC_R3264 : h == 16 ? RC_R3216 : h == 8 ? (packed ? RC_P328 : RC_CGEN) : h == 4 ? (packed ? RC_P324 : RC_CGEN) : RC_CGEN;
  if (w == 16) return h == 64 ? RC_R1664 : h == 32 ? RC_R1632 : h == 8 ? (packed ? RC_P168 : RC_CGEN) : h == 4 ? (packed ? RC_P164 : RC_CGEN) : RC_CGEN;
  if (w == 8 && h == 4) return RC_R84;
  if (w == 4 && h == 8) return RC_R48;
  if (h == 16 && packed) return w == 8 ? RC_P816 : w == 4 ? RC_P416 : RC_CGEN;
  if (h == 32 && packed) return w == 8 ? RC_P832 : w == 4 ? RC_P432 : RC_CGEN;
  if (h == 64 && packed) return w == 8 ? RC_P864 : w == 4 ? RC_P464 : RC_CGEN;
  return RC_CGEN;
}

// Same-address device-scope atomics retire at ~12 ns each (MI355X_MICROARCH.md, row 'fanin'): one atomic per wave and class made this kernel
// 50 us for a 4K picture's 138 k TUs.  Here a workgroup of 1024 threads walks a contiguous slice of the list twice: pass 1 counts per class in
// LDS, ONE global atomic per class reserves the slice's range of every list, pass 2 writes the indices (LDS counters give the positions).
constexpr int RC_CLS_WGS = 128;
// TR: the descriptors are vvcgpu_tr_desc (the plain transform entries through the chain's bodies, vvcgpu_tr_chain_launch): pass 0 also writes each one as
// a chain descriptor into conv (residual plane = "original" and "reconstruction", coefficients = "levels"); transform skip goes to the generic class
template <bool TR>
__global__ __launch_bounds__(1024) void rc_classify_kernel(const void* __restrict__ descsRaw, int n, int* __restrict__ hdr, int* __restrict__ lists,
                                                           unsigned* __restrict__ absSum, int* __restrict__ nextHdr, bool packed, RcDesc* __restrict__ conv)
{
  const RcDesc* descs = static_cast<const RcDesc*>(descsRaw);
  if (blockIdx.x == 0 && threadIdx.x < VVC_CTR_INTS) nextHdr[threadIdx.x] = 0;         // the header of the NEXT call on this stream (vvcgpu_counters)
  __shared__ int cnt[RC_NCLS], base[RC_NCLS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int per = (n + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * per, hi = min(n, lo + per);
  if (tid < RC_NCLS) cnt[tid] = 0;
  __syncthreads();
  // the class of descriptor ti (-1: not served), with the side effects of the first visit (TR: the chain descriptor; chain: the "not served" mark)
  auto classOf = [&](int ti, bool first) -> int
  {
    int cls = -1;
    if (TR)
    {
      const vvcgpu_tr_desc t = static_cast<const vvcgpu_tr_desc*>(descsRaw)[ti];
      RcDesc d;
      d.org_off = t.resi_off; d.pred_off = 0; d.rec_off = t.resi_off; d.level_off = t.coeff_off;
      d.org_stride = t.resi_stride; d.pred_stride = 0; d.rec_stride = t.resi_stride;
      d.w = t.w; d.h = t.h; d.tr_hor = t.tr_hor; d.tr_ver = t.tr_ver; d.intra_slice = 0; d.sign_hiding = 0; d.qp = 0; d.reserved[0] = 0; d.reserved[1] = 0;
      if (first) conv[ti] = d;
      const bool shape = d.w >= 2 && d.w <= 64 && d.h >= 2 && d.h <= 64 && (d.w & (d.w - 1)) == 0 && (d.h & (d.h - 1)) == 0;
      const bool ok = shape && d.tr_hor >= 0 && d.tr_hor <= 2 && d.tr_ver >= 0 && d.tr_ver <= 2 && (d.w <= 32 || d.tr_hor == 0) && (d.h <= 32 || d.tr_ver == 0);
      if (shape && d.tr_hor == 3) cls = RC_CGEN;                              // transform skip: element-wise, in the generic kernel
      else if (ok) cls = rc_class(d, packed);
    }
    else
    {
      const int* f = reinterpret_cast<const int*>(descs + ti) + 11;          // bytes 44..51: w, h, tr_hor, tr_ver, intra_slice, sign_hiding
      const int wh = f[0], tt = f[1];
      RcDesc d;
      d.w = (short)(wh & 0xFFFF); d.h = (short)(wh >> 16); d.tr_hor = (signed char)(tt & 0xFF); d.tr_ver = (signed char)((tt >> 8) & 0xFF);
      const bool ok = d.tr_hor >= 0 && d.tr_hor <= 2 && d.tr_ver >= 0 && d.tr_ver <= 2 && d.w >= 2 && d.w <= 64 && d.h >= 2 && d.h <= 64 &&
                      (d.w & (d.w - 1)) == 0 && (d.h & (d.h - 1)) == 0 && (d.w <= 32 || d.tr_hor == 0) && (d.h <= 32 || d.tr_ver == 0);
      if (ok) cls = rc_class(d, packed);
      else if (first) absSum[ti] = 0xFFFFFFFFu;                              // precondition violated: TU not served, marked
    }
    return cls;
  };
  // position of every lane's descriptor inside the workgroup's part of its class list: one LDS atomic per wave and class present
  auto rankOf = [&](int cls) -> int
  {
    int pos = 0;
#pragma unroll
    for (int k = 0; k < RC_NCLS; k++)
    {
      const unsigned long long m = __builtin_amdgcn_ballot_w64(cls == k);
      if (m == 0ull) continue;
      int b = 0;
      if (lane == 0) b = atomicAdd(&cnt[k], (int)__popcll(m));
      b = __builtin_amdgcn_readfirstlane(b);
      if (cls == k) pos = b + (int)__popcll(m & ((1ull << lane) - 1ull));
    }
    return pos;
  };
  if (per <= 2 * 1024)
  {
    // ONE visit per descriptor (a slice of at most 2048): class and position stay in registers across the reservation of the list ranges
    // (the two-pass form below read and ranked every descriptor twice: 13.3 us for the 138 k TUs of a 4K picture)
    int clsR[2], posR[2];
#pragma unroll
    for (int it = 0; it < 2; it++)
    {
      const int ti = lo + it * 1024 + tid;
      clsR[it] = ti < hi ? classOf(ti, true) : -1;
      posR[it] = rankOf(clsR[it]);
    }
    __syncthreads();
    if (tid < RC_NCLS) base[tid] = cnt[tid] ? atomicAdd(&hdr[tid], cnt[tid]) : 0;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; it++)
      if (clsR[it] >= 0) lists[(size_t)clsR[it] * n + base[clsR[it]] + posR[it]] = lo + it * 1024 + tid;
    return;
  }
  for (int pass = 0; pass < 2; pass++)
  {
    for (int t0 = lo; t0 < hi; t0 += 1024)
    {
      const int ti = t0 + tid;
      const int cls = ti < hi ? classOf(ti, pass == 0) : -1;
      const int pos = rankOf(cls);
      if (pass == 1 && cls >= 0) lists[(size_t)cls * n + base[cls] + pos] = ti;
    }
    __syncthreads();
    if (pass == 0 && tid < RC_NCLS) { base[tid] = cnt[tid] ? atomicAdd(&hdr[tid], cnt[tid]) : 0; cnt[tid] = 0; }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------
// Quantiser / de-quantiser of one TU (flat scaling lists)
struct RcQ
{
  unsigned mul;                       // quantiser scale * (181 for 2:1 shapes)
  int qBits, qBits8;
  long long add;
  int dqScale, rightShift, inMin, inMax;
  bool sbh;
};
__device__ __forceinline__ RcQ rc_qparams(int w, int h, int qp, int bd, int intraSlice, int signHiding)
{
  RcQ q;
  const int lw = ilog2(w), lh = ilog2(h), per = qp / 6, rem = qp - 6 * per;
  const int transformShift = 15 - bd - ((lw + lh) >> 1);
  const bool sqrt2 = ((lw + lh) & 1) != 0;
  const int scale = rem == 0 ? 26214 : rem == 1 ? 23302 : rem == 2 ? 20560 : rem == 3 ? 18396 : rem == 4 ? 16384 : 14564;
  q.mul = (unsigned)scale * (sqrt2 ? 181u : 1u);
  q.qBits = 14 + per + transformShift + (sqrt2 ? 7 : 0);
  q.qBits8 = q.qBits - 8;
  q.add = (long long)(intraSlice ? 171 : 85) << (q.qBits - 9);
  const int invq = rem == 0 ? 40 : rem == 1 ? 45 : rem == 2 ? 51 : rem == 3 ? 57 : rem == 4 ? 64 : 72;
  q.dqScale = invq * (sqrt2 ? 181 : 1);
  q.rightShift = (sqrt2 ? 8 : 0) + (6 - (transformShift + per));
  const int targetBits = min(16, 32 + q.rightShift - 7);
  q.inMin = -(1 << (targetBits - 1)); q.inMax = (1 << (targetBits - 1)) - 1;
  q.sbh = signHiding && w >= 4 && h >= 4;
  return q;
}
// (64-bit arithmetic as the reference's: a 32-bit form of both -- 24-bit multiplies, a funnel shift for (tmp + add) >> qBits, exact for
// |c| < 2^24 and 16 <= qBits <= 31 -- was built and measured in round 6: 71.3 against 68.5 us for the 4K chain, the branch around the 64-bit
// form costs more than the quarter-rate multiplies it saves; docs/OPTIMISATION_LOG.md)
__device__ __forceinline__ int rc_quant_one(const RcQ& q, int c, int& deltaU, int& mag)
{
  const unsigned long long tmp = (unsigned long long)(unsigned)abs(c) * q.mul;
  mag = (int)((tmp + (unsigned long long)q.add) >> q.qBits);
  deltaU = (int)((long long)(tmp - ((unsigned long long)(unsigned)mag << q.qBits)) >> q.qBits8);
  return min(max(c < 0 ? -mag : mag, -32768), 32767);
}
__device__ __forceinline__ int rc_dequant_one(const RcQ& q, int lv)
{
  const long long c = min(max(lv, q.inMin), q.inMax);
  const long long v = q.rightShift > 0 ? (c * q.dqScale + (1ll << (q.rightShift - 1))) >> q.rightShift : (c * q.dqScale) << -q.rightShift;
  return (int)min(max(v, -32768ll), 32767ll);
}

__device__ __forceinline__ unsigned quad_or(unsigned v)
{
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);      // quad_perm [1,0,3,2]
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);      // quad_perm [2,3,0,1]
  return v;
}
__device__ __forceinline__ int quad_min(int v)
{
  v = min(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false));
  v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false));
  return v;
}
__device__ __forceinline__ int wave_max_i32(int v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ int wave_sum_i32(int v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// scan position (diagonal 4x4 scan: 0 4 1 8 5 2 12 9 6 3 13 10 7 14 11 15 in raster terms) of (row r, column x) inside a coefficient group
__device__ __forceinline__ int rc_kpos(int r, int x)
{
  const unsigned tab = r == 0 ? 0x9520u : r == 1 ? 0xC841u : r == 2 ? 0xEB73u : 0xFDA6u;
  return (int)((tab >> (4 * x)) & 15u);
}

// any level of the quad's coefficient group non-zero?
__device__ __forceinline__ bool rc_cg_nonzero(const int (&lv)[4])
{
  return quad_or((unsigned)((lv[0] | lv[1] | lv[2] | lv[3]) != 0)) != 0u;
}

// Sign bit hiding of one coefficient group (xSignBitHidingHDQ, Quant.cpp:142-273; logic as quant_tu in transform.hip, which is pinned against
// the reference): the lane holds rows 0..3 of column x = lane & 3 of the group; cf = coefficients, du = the quantiser's deltaU.
__device__ __forceinline__ void rc_sbh_quad(int (&lv)[4], const int (&du)[4], const int (&cf)[4], bool isLast, int lane)
{
  const int x = lane & 3;
  unsigned mA = 0, mB = 0;                                    // mA: non-zero (low half) | odd (high half) per scan position; mB: negative
#pragma unroll
  for (int r = 0; r < 4; r++)
  {
    const int k = rc_kpos(r, x);
    if (lv[r] != 0) mA |= 1u << k;
    if (lv[r] & 1) mA |= 0x10000u << k;
    if (lv[r] < 0) mB |= 1u << k;
  }
  mA = quad_or(mA); mB = quad_or(mB);
  const unsigned nz = mA & 0xFFFFu;
  const int first = nz ? __ffs((int)nz) - 1 : 16, last = nz ? 31 - __clz((int)nz) : -1;
  const unsigned parity = (unsigned)__popc(mA >> 16) & 1u;
  const unsigned signbit = nz ? ((mB >> first) & 1u) : 1u;
  const bool fix = last - first >= 4 && signbit != parity;
  const int start = isLast ? last : 15;
  constexpr int BIG = 1 << 20;
  int bestKey = BIG * 32, bestChange = 0, bestR = 0;
#pragma unroll
  for (int r = 0; r < 4; r++)
  {
    const int k = rc_kpos(r, x);
    int cost = BIG, change = 0;
    if (k <= start)
    {
      if (lv[r] != 0)
      {
        if (du[r] > 0) { cost = -du[r]; change = 1; }
        else if (!(k == first && abs(lv[r]) == 1)) { cost = du[r]; change = -1; }
      }
      else if (k < first) { if ((cf[r] >= 0 ? 0u : 1u) == signbit) { cost = -du[r]; change = 1; } }
      else { cost = -du[r]; change = 1; }
    }
    const int key = cost * 32 + (15 - k);
    if (key < bestKey) { bestKey = key; bestChange = change; bestR = r; }
  }
  const int qmin = quad_min(bestKey);
  if (fix && bestKey == qmin)                                 // keys are distinct inside a group (the scan position is part of the key)
  {
#pragma unroll
    for (int r = 0; r < 4; r++)
      if (r == bestR)
      {
        int change = bestChange;
        if (lv[r] == 32767 || lv[r] == -32768) change = -1;
        lv[r] += cf[r] >= 0 ? change : -change;
      }
  }
}

// ---------------------------------------------------------------------------------------------------
// Matrix-core path (stages and the LDS matrix image: mfma_tr.h).  The image is built ONCE per device in global memory (rc_build_tables_kernel)
// and copied by every workgroup with 16-byte loads.  Behind the f16 matrices (RC_TAB_HALVES halves) the image carries the 4- / 8-point matrices
// as int32 for the lane-group bodies (RcSmallTab), so that a workgroup's whole table set is ONE run of 16-byte loads.
struct RcSmallTab { int t[3][16 + 64]; int tt[3][16 + 64]; };       // per type: size 4 at 0, size 8 at 16; tt = transposes
static_assert(sizeof(RcSmallTab) % 16 == 0, "copied with 16-byte loads");
constexpr int RC_IMG_U4 = RC_TAB_HALVES / 8 + (int)sizeof(RcSmallTab) / 16;       // 16-byte words of the image
__global__ __launch_bounds__(256) void rc_build_tables_kernel(_Float16* __restrict__ tab, const int* __restrict__ tr32, const int* __restrict__ tr32t)
{
  const int tid = blockIdx.x * 256 + threadIdx.x, nthreads = gridDim.x * 256;
  RcSmallTab* st = reinterpret_cast<RcSmallTab*>(tab + RC_TAB_HALVES);
  for (int e = tid; e < 3 * 80; e += nthreads)
  {
    const int t = e / 80, o = e - t * 80, nsz = o < 16 ? 4 : 8, oo = o < 16 ? o : o - 16;
    st->t[t][o] = tr32[t * 5460 + (nsz * nsz - 4) / 3 + oo];
    st->tt[t][o] = tr32t[t * 5460 + (nsz * nsz - 4) / 3 + oo];
  }
  for (int t = 0; t < 3; t++)
    for (int n = 16; n <= 32; n <<= 1)
      for (int e = tid; e < n * n; e += nthreads)
      {
        const int r = e / n, k = e - r * n, src = t * 5460 + (n * n - 4) / 3 + e;
        tab[rc_tab_off(t, n, 0) + r * (n + 8) + k] = (_Float16)tr32[src];
        tab[rc_tab_off(t, n, 1) + r * (n + 8) + k] = (_Float16)tr32t[src];
      }
  for (int e = tid; e < 4096; e += nthreads)
  {
    const int r = e >> 6, k = e & 63;
    tab[rc_tab_off(0, 64, 0) + r * 72 + k] = (_Float16)tr32[1364 + e];
    tab[rc_tab_off(0, 64, 1) + r * 72 + k] = (_Float16)tr32t[1364 + e];
  }
  for (int t = 0; t < 3; t++)
    for (int n = 4; n <= 8; n <<= 1)
      for (int e = tid; e < n * n; e += nthreads)
      {
        const int src = t * 5460 + (n * n - 4) / 3 + e;
        tab[rc_small_off(t, n, 0) + e] = (_Float16)tr32[src];
        tab[rc_small_off(t, n, 1) + e] = (_Float16)tr32t[src];
      }
}

// One TU of W x H (W, H in {16, 32, 64}) on the matrix cores.  Returns false (nothing written) when a residual sample lies outside +-1023
// (precondition violated: the caller's generic path takes the TU).
// mode (workgroup-uniform; the plain transform entries run through the same bodies, vvcgpu_tr_chain_launch below): 0 = the chain; 1 = FORWARD
// transform only: the "original" plane holds the residual, no prediction is read, the coefficients (unquantised, int32) go where the chain writes
// levels, incl. the zero-out region; 2 = INVERSE transform only: the coefficients are read where the chain writes levels (16-bit values: a TU with
// a larger one returns false / goes to the fall-back list), the residual goes where the chain writes the reconstruction, no prediction, no pixel clip.
enum { RC_CHAIN = 0, RC_FWD = 1, RC_INV = 2 };
template <int W, int H, int MODE>
__device__ __forceinline__ bool rc_tu_mfma(const RcDesc& d, const Pel* __restrict__ orgBase, const Pel* __restrict__ predBase, Pel* __restrict__ recBase,
                                           TCoeff* __restrict__ levelBase, unsigned* __restrict__ absSumOut, int ti, int bd, int clpMin, int clpMax,
                                           const _Float16* tab, const unsigned short* __restrict__ dqInv, const int* __restrict__ scanOff, int lane)
{
  constexpr int mode = MODE;

  typedef MtShape<W, H> S;
  constexpr int LW = W == 16 ? 4 : W == 32 ? 5 : 6, LH = H == 16 ? 4 : H == 32 ? 5 : 6;
  const int c = lane & 15, g = lane >> 4;
  const Pel* org = orgBase + d.org_off;
  const Pel* pred = predBase + d.pred_off;
  const _Float16* Th = tab + rc_tab_off(d.tr_hor, W, 0);
  const _Float16* ThT = tab + rc_tab_off(d.tr_hor, W, 1);
  const _Float16* Tv = tab + rc_tab_off(d.tr_ver, H, 0);
  const _Float16* TvT = tab + rc_tab_off(d.tr_ver, H, 1);
  TCoeff* level = levelBase + d.level_off;
  const int4v z = { 0, 0, 0, 0 };
  int cq[S::JT][S::IT][4];                             // [column tile (i)][row tile (k)]: the inverse stages' input

  if (mode != RC_INV)
  {
    // ---- residual fragments of stage F1 (A = X from memory, natural k order)
    h8 x[S::RT][S::XS];
    bool inRange = true;
    if (W == 16)
    {
#pragma unroll
      for (int rt = 0; rt < S::RT; rt++)
      {
        const pel4 o = *reinterpret_cast<const pel4*>(org + (size_t)(16 * rt + c) * d.org_stride + 4 * g);
        pel4 p = { 0, 0, 0, 0 };
        if (mode == RC_CHAIN) p = *reinterpret_cast<const pel4*>(pred + (size_t)(16 * rt + c) * d.pred_stride + 4 * g);
        _Float16 a[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { const int v = (int)o[j] - (int)p[j]; inRange = inRange && v >= -1023 && v <= 1023; a[j] = (_Float16)(short)v; }
        x[rt][0] = h8{ a[0], a[1], a[2], a[3], a[0], a[1], a[2], a[3] };
      }
    }
    else
    {
      pel8 o[S::RT][S::XS], p[S::RT][S::XS];
#pragma unroll
      for (int rt = 0; rt < S::RT; rt++)
#pragma unroll
        for (int s = 0; s < S::XS; s++)
        {
          o[rt][s] = *reinterpret_cast<const pel8*>(org + (size_t)(16 * rt + c) * d.org_stride + 32 * s + 8 * g);
          p[rt][s] = pel8{ 0, 0, 0, 0, 0, 0, 0, 0 };
          if (mode == RC_CHAIN) p[rt][s] = *reinterpret_cast<const pel8*>(pred + (size_t)(16 * rt + c) * d.pred_stride + 32 * s + 8 * g);
        }
#pragma unroll
      for (int rt = 0; rt < S::RT; rt++)
#pragma unroll
        for (int s = 0; s < S::XS; s++)
#pragma unroll
          for (int j = 0; j < 8; j++)
          {
            const int v = (int)o[rt][s][j] - (int)p[rt][s][j];
            inRange = inRange && v >= -1023 && v <= 1023;
            x[rt][s][j] = (_Float16)(short)v;
          }
    }
    if (__builtin_amdgcn_ballot_w64(!inRange) != 0ull) return false;

    // rounding shifts of the forward stages (TrQuant.cpp:151-152, 214)
    const int s1 = LW + bd + 6 - 15 + 2, s2 = LH + 6 + 2;
    int t1[S::JT][S::RT][4];                             // [column tile][row tile]: the row tiles are the k dimension of the next product
    mt_fwd1<W, H>(t1, x, Th, s1, c, g);
    int cf[S::IT][S::JT][4];                             // [row tile of C (vertical frequency)][column tile (horizontal frequency)]
    mt_fwd2<W, H>(cf, t1, Tv, s2, c, g);

    // ---- quantiser: tile (it, jt) holds rows 16 it + 4 g + reg, column 16 jt + c; the quad c >> 2 of row group g is one coefficient group
    if (mode == RC_CHAIN)
    {
      const RcQ q = rc_qparams(W, H, d.qp, bd, d.intra_slice, d.sign_hiding);
      const unsigned short* inv = dqInv + scanOff[(LW - 1) * 6 + (LH - 1)];
      int lv[S::IT][S::JT][4], du[S::IT][S::JT][4];
      int sum = 0, lastCg = -1, cgIdx[S::IT][S::JT];
#pragma unroll
      for (int it = 0; it < S::IT; it++)
#pragma unroll
        for (int jt = 0; jt < S::JT; jt++)
        {
#pragma unroll
          for (int r = 0; r < 4; r++) { int mag; lv[it][jt][r] = rc_quant_one(q, cf[it][jt][r], du[it][jt][r], mag); sum += mag; }
          cgIdx[it][jt] = (int)inv[(16 * it + 4 * g) * W + 16 * jt + (c & ~3)] >> 4;
          if (rc_cg_nonzero(lv[it][jt])) lastCg = max(lastCg, cgIdx[it][jt]);
        }
      lastCg = wave_max_i32(lastCg);
      sum = wave_sum_i32(sum);
      if (lane == 0) absSumOut[ti] = (unsigned)sum;
#pragma unroll
      for (int it = 0; it < S::IT; it++)
#pragma unroll
        for (int jt = 0; jt < S::JT; jt++)
        {
          if (q.sbh) rc_sbh_quad(lv[it][jt], du[it][jt], cf[it][jt], cgIdx[it][jt] == lastCg, lane);
#pragma unroll
          for (int r = 0; r < 4; r++) { level[(16 * it + 4 * g + r) * W + 16 * jt + c] = lv[it][jt][r]; cq[jt][it][r] = rc_dequant_one(q, lv[it][jt][r]); }
        }
    }
    else
    {
#pragma unroll
      for (int it = 0; it < S::IT; it++)
#pragma unroll
        for (int jt = 0; jt < S::JT; jt++)
#pragma unroll
          for (int r = 0; r < 4; r++) level[(16 * it + 4 * g + r) * W + 16 * jt + c] = cf[it][jt][r];
    }
    // zero-out region of the level array: columns >= 32 of the kept rows, then the rows >= 32
    if (W == 64) for (int e = lane; e < S::HJ * 8; e += 64) *reinterpret_cast<int4v*>(level + (e >> 3) * 64 + 32 + 4 * (e & 7)) = z;
    if (H == 64) for (int e = lane; e < 32 * W / 4; e += 64) *reinterpret_cast<int4v*>(level + 32 * W + 4 * e) = z;
    if (mode == RC_FWD) return true;
  }
  else
  {
    bool fits = true;
#pragma unroll
    for (int it = 0; it < S::IT; it++)
#pragma unroll
      for (int jt = 0; jt < S::JT; jt++)
#pragma unroll
        for (int r = 0; r < 4; r++)
        {
          const int v = level[(16 * it + 4 * g + r) * W + 16 * jt + c];
          fits = fits && v >= -32768 && v <= 32767;
          cq[jt][it][r] = v;
        }
    if (__builtin_amdgcn_ballot_w64(!fits) != 0ull) return false;
  }

  // ---- inverse stages, reconstruction
  int y1[S::RT][S::JT][4];                             // [row tile (r)][frequency tile (i)]: the frequency tiles are the k dimension of the last product
  mt_inv1<W, H>(y1, cq, TvT, c, g);
  const int s2i = (6 + 15 - 1) - bd + 2;
  Pel* rec = recBase + d.rec_off;
  mt_inv2<W, H>(y1, ThT, s2i, c, g, [&](int rt, int xt, const int (&resi)[4])
  {
    pel4 out;
    if (mode == RC_CHAIN)
    {
      const pel4 pv = *reinterpret_cast<const pel4*>(pred + (size_t)(16 * rt + c) * d.pred_stride + 16 * xt + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; r++) out[r] = (short)clip3(clpMin, clpMax, (int)pv[r] + (int)(short)resi[r]);
    }
    else
    {
#pragma unroll
      for (int r = 0; r < 4; r++) out[r] = (short)resi[r];
    }
    *reinterpret_cast<pel4*>(rec + (size_t)(16 * rt + c) * d.rec_stride + 16 * xt + 4 * g) = out;
  });
  return true;
}

// the single-wave matrix-core bodies of the chain kernel's rarer classes (64x16 .. 16x32) as real calls: inline, the kernel's register need is the
// sum of what the compiler keeps live across its switch (docs/OPTIMISATION_LOG.md, round 6)
template <int W, int H, int MODE>
__device__ __noinline__ bool rc_tu_mfma_call(const RcDesc* __restrict__ descs, int ti, const Pel* __restrict__ orgBase, const Pel* __restrict__ predBase, Pel* __restrict__ recBase,
                                             TCoeff* __restrict__ levelBase, unsigned* __restrict__ absSumOut, int bd, int clpMin, int clpMax,
                                             const _Float16* tab, const unsigned short* __restrict__ dqInv, const int* __restrict__ scanOff, int lane)
{
  return rc_tu_mfma<W, H, MODE>(descs[ti], orgBase, predBase, recBase, levelBase, absSumOut, ti, bd, clpMin, clpMax, tab, dqInv, scanOff, lane);
}

// ---------------------------------------------------------------------------------------------------
// One TU of W x H, both sides in {32, 64}, by the FOUR waves of a workgroup together (round 6).  One wave per TU held a
// whole 64 x 64 chain -- 80 products, 198 vector registers, two waves per SIMD for the WHOLE kernel, and 20 us of one wave's latency per TU.
// Here a wave owns a strip of every stage and no wave holds more than a quarter of a tile set:
//   F1  M1 = X Th^T      row tile rt = wave (H = 64) or wave >> 1 with ONE frequency tile (H = 32)     -> limbs of M1, transposed, to LDS (E1[j1][r])
//   F2  C = Tv M1        ONE 16 x 16 tile (it, jt) = (wave >> 1, wave & 1); B operand = E1 rows (16-byte reads) -> quantiser on that tile
//                        (abs sum and last group meet in LDS) -> levels to memory, limbs of the de-quantised tile, transposed, to LDS (E2[i][k])
//   I1  Y1^T = Cq^T Tv   sample-row tile rt = wave (H = 64) or wave >> 1; A operand = E2 rows; both frequency tiles (the next stage sums over them)
//   I2  R^T = Th^T Y1^T  the wave's row tile, all column tiles (H = 64) or every second one (H = 32: two waves share a row tile) -> reconstruction
// 20 products per wave for 64 x 64, three barriers between the stages and one behind the TU.  E1 / E2 are f16 limb planes with row pitch H + 8 / 40
// halves: pitch / 2 = 4 (mod 16) dwords spreads the 16 rows that one ds_write_b64 / ds_read_b128 pass touches over all 64 banks.  E2 lies over E1
// (dead once every wave has its F2 operands: the barrier behind the quantiser's partial sums), and both lie over the per-wave scratch of the other
// bodies: the co-operative classes are the FIRST slots of a workgroup and the barrier behind a TU is the fence between the two uses.
// `red`: { abs sum, last coefficient group, "outside the matrix-core range" } -- zero / -1 / zero on entry, left so on exit.
constexpr int RC_EX_HALVES = 2 * 32 * 72;                  // E1 of a 64-row TU: two limb planes of 32 rows x 72 halves = 9216 bytes
template <int W, int H, int MODE>
__device__ __forceinline__ bool rc_tu_coop(const RcDesc& d, const Pel* __restrict__ orgBase, const Pel* __restrict__ predBase, Pel* __restrict__ recBase,
                                           TCoeff* __restrict__ levelBase, unsigned* __restrict__ absSumOut, int ti, int bd, int clpMin, int clpMax,
                                           const _Float16* tab, const unsigned short* __restrict__ dqInv, const int* __restrict__ scanOff,
                                           _Float16* ex, int* red, int wv, int lane)
{
  static_assert((W == 32 || W == 64) && (H == 32 || H == 64), "co-operative body: both sides 32 or 64");
  typedef MtShape<W, H> S;
  static_assert(S::JT == 2 && S::IT == 2, "32 kept frequencies per dimension");
  constexpr int LW = W == 32 ? 5 : 6, LH = H == 32 ? 5 : 6;
  constexpr int P1 = H + 8, P2 = 40;
  constexpr int XS = W / 32, HS = H / 32;
  const int c = lane & 15, g = lane >> 4;
  const Pel* org = orgBase + d.org_off;
  const Pel* pred = predBase + d.pred_off;
  const _Float16* Th = tab + rc_tab_off(d.tr_hor, W, 0);
  const _Float16* ThT = tab + rc_tab_off(d.tr_hor, W, 1);
  const _Float16* Tv = tab + rc_tab_off(d.tr_ver, H, 0);
  const _Float16* TvT = tab + rc_tab_off(d.tr_ver, H, 1);
  TCoeff* level = levelBase + d.level_off;
  _Float16* e1h = ex;
  _Float16* e1l = ex + 32 * P1;
  _Float16* e2h = ex;
  _Float16* e2l = ex + 32 * P2;
  const int it = wv >> 1, jt = wv & 1;                      // the wave's coefficient tile
  const int rtW = S::RT == 4 ? wv : (wv >> 1);              // the wave's sample-row tile
  const int4v z = { 0, 0, 0, 0 };
  auto leave = [&]() -> bool                                // a TU outside the matrix-core range: every wave takes this exit (the flag is read behind a barrier)
  {
    __syncthreads();
    if (wv == 0 && lane == 0) red[2] = 0;
    __syncthreads();
    return false;
  };

  if (MODE != RC_INV)
  {
    // ---- F1: the wave's 16 rows
    h8 x[XS];
    bool inRange = true;
    {
      pel8 o[XS], pv[XS];
#pragma unroll
      for (int s = 0; s < XS; s++)
      {
        o[s] = *reinterpret_cast<const pel8*>(org + (size_t)(16 * rtW + c) * d.org_stride + 32 * s + 8 * g);
        pv[s] = pel8{ 0, 0, 0, 0, 0, 0, 0, 0 };
        if (MODE == RC_CHAIN) pv[s] = *reinterpret_cast<const pel8*>(pred + (size_t)(16 * rtW + c) * d.pred_stride + 32 * s + 8 * g);
      }
#pragma unroll
      for (int s = 0; s < XS; s++)
#pragma unroll
        for (int j = 0; j < 8; j++)
        {
          const int v = (int)o[s][j] - (int)pv[s][j];
          inRange = inRange && v >= -1023 && v <= 1023;
          x[s][j] = (_Float16)(short)v;
        }
    }
    if (__builtin_amdgcn_ballot_w64(!inRange) != 0ull && lane == 0) red[2] = 1;
    const int s1 = LW + bd + 6 - 15 + 2, s2 = LH + 6 + 2;
    constexpr int NJ = S::RT == 4 ? 2 : 1;                  // frequency tiles of this wave in F1
#pragma unroll
    for (int j = 0; j < NJ; j++)
    {
      const int jf = S::RT == 4 ? j : jt;
      f4 m1 = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
      for (int s = 0; s < XS; s++)
        m1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x[s], *reinterpret_cast<const h8*>(Th + (16 * jf + c) * (W + 8) + 32 * s + 8 * g), m1, 0, 0, 0);
      _Float16 hi[4], lo[4];
#pragma unroll
      for (int r = 0; r < 4; r++) rc_limbs(((int)m1[r] + (1 << (s1 - 1))) >> s1, hi[r], lo[r]);
      *reinterpret_cast<h4*>(e1h + (16 * jf + c) * P1 + 16 * rtW + 4 * g) = h4{ hi[0], hi[1], hi[2], hi[3] };
      *reinterpret_cast<h4*>(e1l + (16 * jf + c) * P1 + 16 * rtW + 4 * g) = h4{ lo[0], lo[1], lo[2], lo[3] };
    }
    __syncthreads();
    if (red[2] != 0) return leave();

    // ---- F2: tile (it, jt) of the coefficients
    int cf[4];
    {
      f4 hi = { 0.f, 0.f, 0.f, 0.f }, lo = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
      for (int s = 0; s < HS; s++)
      {
        const h8 a = *reinterpret_cast<const h8*>(Tv + (16 * it + c) * (H + 8) + 32 * s + 8 * g);
        hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, *reinterpret_cast<const h8*>(e1h + (16 * jt + c) * P1 + 32 * s + 8 * g), hi, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, *reinterpret_cast<const h8*>(e1l + (16 * jt + c) * P1 + 32 * s + 8 * g), lo, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; r++) cf[r] = ((((int)hi[r]) << 8) + (int)lo[r] + (1 << (s2 - 1))) >> s2;
    }
    const int tid = wv * 64 + lane;
    if (MODE == RC_CHAIN)
    {
      const RcQ q = rc_qparams(W, H, d.qp, bd, d.intra_slice, d.sign_hiding);
      const unsigned short* inv = dqInv + scanOff[(LW - 1) * 6 + (LH - 1)];
      int lv[4], du[4], sum = 0;
#pragma unroll
      for (int r = 0; r < 4; r++) { int mag; lv[r] = rc_quant_one(q, cf[r], du[r], mag); sum += mag; }
      const int cgIdx = (int)inv[(16 * it + 4 * g) * W + 16 * jt + (c & ~3)] >> 4;
      int lastCg = rc_cg_nonzero(lv) ? cgIdx : -1;
      lastCg = wave_max_i32(lastCg);
      sum = wave_sum_i32(sum);
      if (lane == 0) { atomicAdd(&red[0], sum); atomicMax(&red[1], lastCg); }
      __syncthreads();                                      // (every wave has read its E1 operands: E2 may be written)
      lastCg = red[1];
      if (tid == 0) absSumOut[ti] = (unsigned)red[0];
      if (q.sbh) rc_sbh_quad(lv, du, cf, cgIdx == lastCg, lane);
      _Float16 hi[4], lo[4];
#pragma unroll
      for (int r = 0; r < 4; r++)
      {
        level[(16 * it + 4 * g + r) * W + 16 * jt + c] = lv[r];
        rc_limbs(rc_dequant_one(q, lv[r]), hi[r], lo[r]);
      }
      *reinterpret_cast<h4*>(e2h + (16 * jt + c) * P2 + 16 * it + 4 * g) = h4{ hi[0], hi[1], hi[2], hi[3] };
      *reinterpret_cast<h4*>(e2l + (16 * jt + c) * P2 + 16 * it + 4 * g) = h4{ lo[0], lo[1], lo[2], lo[3] };
    }
    else
    {
#pragma unroll
      for (int r = 0; r < 4; r++) level[(16 * it + 4 * g + r) * W + 16 * jt + c] = cf[r];
    }
    // zero-out region of the level array: columns >= 32 of the kept rows, then the rows >= 32
    if (W == 64) for (int e = tid; e < 32 * 8; e += 256) *reinterpret_cast<int4v*>(level + (e >> 3) * 64 + 32 + 4 * (e & 7)) = z;
    if (H == 64) for (int e = tid; e < 32 * W / 4; e += 256) *reinterpret_cast<int4v*>(level + 32 * W + 4 * e) = z;
    if (MODE == RC_FWD)
    {
      __syncthreads();                                      // E1 is the next slot's scratch
      return true;
    }
  }
  else
  {
    bool fits = true;
    _Float16 hi[4], lo[4];
#pragma unroll
    for (int r = 0; r < 4; r++)
    {
      const int v = level[(16 * it + 4 * g + r) * W + 16 * jt + c];
      fits = fits && v >= -32768 && v <= 32767;
      rc_limbs(v, hi[r], lo[r]);
    }
    if (__builtin_amdgcn_ballot_w64(!fits) != 0ull && lane == 0) red[2] = 1;
    *reinterpret_cast<h4*>(e2h + (16 * jt + c) * P2 + 16 * it + 4 * g) = h4{ hi[0], hi[1], hi[2], hi[3] };
    *reinterpret_cast<h4*>(e2l + (16 * jt + c) * P2 + 16 * it + 4 * g) = h4{ lo[0], lo[1], lo[2], lo[3] };
  }
  __syncthreads();
  if (MODE == RC_INV && red[2] != 0) return leave();

  // ---- I1: Y1^T tiles (jt', rtW), both frequency tiles
  int y1[2][4];
  {
    const h8 b = *reinterpret_cast<const h8*>(TvT + (16 * rtW + c) * (H + 8) + 8 * g);
#pragma unroll
    for (int j = 0; j < 2; j++)
    {
      const f4 hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const h8*>(e2h + (16 * j + c) * P2 + 8 * g), b, f4{ 0.f, 0.f, 0.f, 0.f }, 0, 0, 0);
      const f4 lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const h8*>(e2l + (16 * j + c) * P2 + 8 * g), b, f4{ 0.f, 0.f, 0.f, 0.f }, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; r++) y1[j][r] = clip3(-(1 << 15), (1 << 15) - 1, ((((int)hi[r]) << 8) + (int)lo[r] + 256) >> 9);
    }
  }
  // ---- I2 + reconstruction: the wave's row tile, its column tiles
  {
    const int s2i = (6 + 15 - 1) - bd + 2;
    Pel* rec = recBase + d.rec_off;
    h8 bh[1], bl[1];
    rc_tile_frags<32>(bh, bl, y1);
    constexpr int XSTEP = S::RT == 4 ? 1 : 2;
#pragma unroll
    for (int xi = 0; xi < S::CT / XSTEP; xi++)
    {
      const int xt = S::RT == 4 ? xi : 2 * xi + (wv & 1);
      h8 a[1];
      rc_mat_frags<32>(a, ThT, W + 8, 16 * xt + c, g);
      const f4 hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], bh[0], f4{ 0.f, 0.f, 0.f, 0.f }, 0, 0, 0);
      const f4 lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], bl[0], f4{ 0.f, 0.f, 0.f, 0.f }, 0, 0, 0);
      pel4 out;
      pel4 pv = { 0, 0, 0, 0 };
      if (MODE == RC_CHAIN) pv = *reinterpret_cast<const pel4*>(pred + (size_t)(16 * rtW + c) * d.pred_stride + 16 * xt + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; r++)
      {
        const int resi = clip3(-(1 << 15), (1 << 15) - 1, ((((int)hi[r]) << 8) + (int)lo[r] + (1 << (s2i - 1))) >> s2i);
        out[r] = MODE == RC_CHAIN ? (short)clip3(clpMin, clpMax, (int)pv[r] + (int)(short)resi) : (short)resi;
      }
      *reinterpret_cast<pel4*>(rec + (size_t)(16 * rtW + c) * d.rec_stride + 16 * xt + 4 * g) = out;
    }
  }
  if (MODE == RC_CHAIN && wv == 0 && lane == 0) { red[0] = 0; red[1] = -1; }     // (read by every wave in front of the barrier above)
  __syncthreads();                                          // E2 is the next slot's scratch
  return true;
}

// A TU that a matrix-core / lane-group body cannot take (its residual leaves +-1023, or inverse-only coefficients beyond 16 bits) is served ON THE SPOT by the wave
// that found it, through the generic body with two 4096-int buffers in global scratch (`fbScr`: per wave, rc_chain_launch) -- rare, slow, exact.  Rounds 3 - 5 listed such
// TUs for a launch behind the chain (4.7 us per 4K picture for an empty list).
template <int MODE>
__device__ __noinline__ void rc_fallback_call(const RcDesc* __restrict__ descs, int ti, const Pel* __restrict__ orgBase, const Pel* __restrict__ predBase, Pel* __restrict__ recBase,
                                              TCoeff* __restrict__ levelBase, unsigned* __restrict__ absSumOut, int bd, int clpMin, int clpMax,
                                              const int* __restrict__ tr32, const unsigned short* __restrict__ dqInv, const int* __restrict__ scanOff, int* fbScr, int lane);
// the lanes whose bit is set in `mask` each hold a TU index in tiLane: one after the other
template <int MODE>
__device__ __forceinline__ void rc_fallback_lanes(unsigned long long mask, int tiLane, const RcDesc* __restrict__ descs, const Pel* __restrict__ orgBase, const Pel* __restrict__ predBase,
                                                  Pel* __restrict__ recBase, TCoeff* __restrict__ levelBase, unsigned* __restrict__ absSumOut, int bd, int clpMin, int clpMax,
                                                  const int* __restrict__ tr32, const unsigned short* __restrict__ dqInv, const int* __restrict__ scanOff, int* fbScr, int lane)
{
  while (mask)
  {
    const int l = (int)__builtin_ctzll(mask);
    mask &= mask - 1ull;
    rc_fallback_call<MODE>(descs, __builtin_amdgcn_readlane(tiLane, l), orgBase, predBase, recBase, levelBase, absSumOut, bd, clpMin, clpMax, tr32, dqInv, scanOff, fbScr, lane);
  }
}

// ---------------------------------------------------------------------------------------------------
// Packed tiles: TUs with a 4- or 8-point side (a real encode's residual is mostly these: tests/golden/trace_*.npz) on the matrix cores --
// G = (16 / W) (16 / H) TUs of W x H side by side in ONE 16x16 tile, sub-TU (sx, sy) at tile columns W sx .., rows H sy ..  The 1-D stages become
// products with BLOCK-DIAGONAL matrices (the W-point matrix repeated 16 / W times along the diagonal); a lane builds its fragment of such a matrix
// from the 4 / 8 / 16-point table of the transform type of the sub-TU its row belongs to, or zero off the diagonal blocks.  The sub-TUs choose their
// types independently: a product D = A B shares B between all rows of A, so where the type belongs to the OTHER operand's index the stage runs
// 16 / H (or 16 / W) passes into the same accumulator, each with that operand masked to one row (column) of sub-TUs -- 7 .. 28 small MFMAs per
// tile instead of 7, on a pipe that is otherwise idle.  Everything else (limbs, rounding, quantiser on the lane's 4 x 1 column of a 4x4 group,
// sign-bit hiding per quad) is the 16x16 form with per-lane TU parameters.  A TU with a residual outside +-1023 goes to the fall-back list.
__device__ __forceinline__ h4 rc_frag4(const _Float16* tab, int type, int n, int transposed, int row, int k0)
{
  const int off = n == 16 ? rc_tab_off(type, 16, transposed) + row * 24 + k0 : rc_small_off(type, n, transposed) + row * n + k0;
  return *reinterpret_cast<const h4*>(tab + off);
}
__device__ __forceinline__ h4 rc_limb_h4(const int (&v)[4], bool high)
{
  _Float16 h[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; j++) rc_limbs(v[j], h[j], l[j]);
  return high ? h4{ h[0], h[1], h[2], h[3] } : h4{ l[0], l[1], l[2], l[3] };
}

template <int W, int H, int MODE>
__device__ __noinline__ void rc_tile_packed(const RcDesc* __restrict__ descs, const int* __restrict__ list, int cnt, int item,
                                               const Pel* __restrict__ orgBase, const Pel* __restrict__ predBase, Pel* __restrict__ recBase,
                                               TCoeff* __restrict__ levelBase, unsigned* __restrict__ absSumOut, int bd, int clpMin, int clpMax,
                                               const _Float16* tab, const unsigned short* __restrict__ dqInv, const int* __restrict__ scanOff,
                                               const int* __restrict__ tr32, int* __restrict__ fbScr, int* info, int lane)
{
  unsigned long long fbMask = 0ull;                                             // sub-TUs this tile cannot take (lane j < G: TU j of the tile): served behind it
  int fbTi = 0;
  constexpr int mode = MODE;

  constexpr int NX = 16 / W, NY = 16 / H, G = NX * NY;
  constexpr int LW = W == 4 ? 2 : W == 8 ? 3 : 4, LH = H == 4 ? 2 : H == 8 ? 3 : 4;
  const int c = lane & 15, g = lane >> 4;
  const h4 zero4 = { (_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0 };
  // the wave's TUs: sub-TU s = sy NX + sx is entry item G + s of the class list; info[s] = descriptor index (-1: none), info[16 + s] = types
  {
    int ti = -1, types = 0;
    if (lane < G && item * G + lane < cnt) { ti = list[item * G + lane]; const RcDesc& d = descs[ti]; types = (int)d.tr_hor | ((int)d.tr_ver << 2); }
    if (lane < G) { info[lane] = ti; info[16 + lane] = types; }
  }
  RC_WAVE_SYNC();
  // two views of the tile: L = sample layout (the lane's four samples: tile row c, columns 4 g ..), Q = coefficient layout (rows 4 g .., column c)
  const int sxL = (4 * g) >> LW, syL = c >> LH, sxQ = c >> LW, syQ = (4 * g) >> LH;
  const int tiL = info[syL * NX + sxL], tiQ = info[syQ * NX + sxQ], ti0 = info[0];
  auto laneMask = [&](int sx, int sy) -> unsigned long long               // lanes of view L that hold samples of sub-TU (sx, sy)
  {
    const unsigned long long cm = H == 16 ? 0xFFFFull : (((1ull << H) - 1ull) << (sy * H));
    unsigned long long m = 0;
#pragma unroll
    for (int gg = 0; gg < W / 4; gg++) m |= cm << (16 * (sx * (W / 4) + gg));
    return m;
  };
  auto laneMaskQ = [&](int sx, int sy) -> unsigned long long              // lanes of view Q that hold coefficients of sub-TU (sx, sy)
  {
    const unsigned long long cm = W == 16 ? 0xFFFFull : (((1ull << W) - 1ull) << (sx * W));
    unsigned long long m = 0;
#pragma unroll
    for (int gg = 0; gg < H / 4; gg++) m |= cm << (16 * (sy * (H / 4) + gg));
    return m;
  };
  const RcDesc& dL = descs[tiL >= 0 ? tiL : ti0];
  const RcDesc& dQ = descs[tiQ >= 0 ? tiQ : ti0];
  const int rowL = c & (H - 1), colL = (4 * g) & (W - 1);
  const int y0 = (4 * g) & (H - 1), xq = c & (W - 1);
  TCoeff* level = levelBase + dQ.level_off;
  pel4 pv = { 0, 0, 0, 0 };
  bool okL, okQ;
  int cq[4];
  if (mode != RC_INV)
  {
  int x[4] = { 0, 0, 0, 0 };
  bool inRange = true;
  if (tiL >= 0)
  {
    const pel4 o = *reinterpret_cast<const pel4*>(orgBase + dL.org_off + (size_t)rowL * dL.org_stride + colL);
    if (mode == RC_CHAIN) pv = *reinterpret_cast<const pel4*>(predBase + dL.pred_off + (size_t)rowL * dL.pred_stride + colL);
#pragma unroll
    for (int j = 0; j < 4; j++) { x[j] = (int)o[j] - (int)pv[j]; inRange = inRange && x[j] >= -1023 && x[j] <= 1023; }
  }
  const unsigned long long badLanes = __builtin_amdgcn_ballot_w64(!inRange);
  okL = tiL >= 0 && (badLanes & laneMask(sxL, syL)) == 0ull; okQ = tiQ >= 0 && (badLanes & laneMask(sxQ, syQ)) == 0ull;
  { const bool fb_ = badLanes != 0ull && lane < G && info[lane] >= 0 && (badLanes & laneMask(lane % NX, lane / NX)) != 0ull; if (fb_) fbTi = info[lane]; fbMask |= __builtin_amdgcn_ballot_w64(fb_); }
  h4 xa = zero4;
  if (okL) xa = h4{ (_Float16)(short)x[0], (_Float16)(short)x[1], (_Float16)(short)x[2], (_Float16)(short)x[3] };

  // ---- F1: M1[r][j] = sum_k X[r][k] Th(sub-TU of r, j)[j][k]: the type belongs to the row of X -> one pass per row of sub-TUs
  const int s1 = LW + bd + 6 - 15 + 2, s2 = LH + 6 + 2;
  int t1[4];
  {
    f4 m1 = { 0.f, 0.f, 0.f, 0.f };
    const bool diag = sxQ == sxL;                                         // the lane's fragment lies in a diagonal block (j = c, k = 4 g ..)
#pragma unroll
    for (int p = 0; p < NY; p++)
    {
      const h4 a = (NY == 1 || syL == p) ? xa : zero4;
      const h4 b = diag ? rc_frag4(tab, info[16 + p * NX + sxQ] & 3, W, 0, c & (W - 1), colL) : zero4;
      m1 = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, m1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) t1[r] = ((int)m1[r] + (1 << (s1 - 1))) >> s1;     // M1[row 4 g + r][frequency c]
  }
  // ---- F2: C[j2][j1] = sum_r Tv(sub-TU of j2, j1)[j2][r] M1[r][j1]: the type also belongs to the column of M1 -> one pass per column of sub-TUs
  int cf[4];
  {
    const h4 bh = rc_limb_h4(t1, true), bl = rc_limb_h4(t1, false);
    f4 hi = { 0.f, 0.f, 0.f, 0.f }, lo = { 0.f, 0.f, 0.f, 0.f };
    const bool diag = syL == syQ;                                         // A: row j2 = c, k = r = 4 g ..
#pragma unroll
    for (int q = 0; q < NX; q++)
    {
      const bool mine = NX == 1 || sxQ == q;                              // B / D column c belongs to sub-TU column q
      const h4 a = diag ? rc_frag4(tab, (info[16 + syL * NX + q] >> 2) & 3, H, 0, rowL, (4 * g) & (H - 1)) : zero4;
      hi = __builtin_amdgcn_mfma_f32_16x16x16f16(a, mine ? bh : zero4, hi, 0, 0, 0);
      lo = __builtin_amdgcn_mfma_f32_16x16x16f16(a, mine ? bl : zero4, lo, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) cf[r] = ((((int)hi[r]) << 8) + (int)lo[r] + (1 << (s2 - 1))) >> s2;   // C[vertical frequency 4 g + r][horizontal frequency c]
  }
  if (mode == RC_FWD)                                                     // forward transform only: the coefficients are the result
  {
    if (okQ)
    {
#pragma unroll
      for (int r = 0; r < 4; r++) level[(y0 + r) * W + xq] = cf[r];
    }
    RC_WAVE_SYNC();
    rc_fallback_lanes<MODE>(fbMask, fbTi, descs, orgBase, predBase, recBase, levelBase, absSumOut, bd, clpMin, clpMax, tr32, dqInv, scanOff, fbScr, lane);
    return;
  }
  // ---- quantiser in view Q: the lane's four coefficients are rows y0 .. y0 + 3 of column xq of its TU; its quad is one coefficient group
  const RcQ q = rc_qparams(W, H, dQ.qp, bd, dQ.intra_slice, dQ.sign_hiding);
  int lv[4], du[4], sum = 0;
#pragma unroll
  for (int r = 0; r < 4; r++) { int mag; lv[r] = rc_quant_one(q, cf[r], du[r], mag); sum += mag; }
  const int cgIdx = (int)(dqInv + scanOff[(LW - 1) * 6 + (LH - 1)])[y0 * W + (xq & ~3)] >> 4;
  int lastCg = rc_cg_nonzero(lv) ? cgIdx : -1;
  // reductions over the lanes of one TU in view Q: column bits below W, row-group bits below H / 4
#pragma unroll
  for (int m = 1; m < W; m <<= 1) { sum += __shfl_xor(sum, m); if (m >= 4) lastCg = max(lastCg, __shfl_xor(lastCg, m)); }
#pragma unroll
  for (int m = 16; m < 4 * H; m <<= 1) { sum += __shfl_xor(sum, m); lastCg = max(lastCg, __shfl_xor(lastCg, m)); }
  if (okQ && xq == 0 && y0 == 0) absSumOut[tiQ] = (unsigned)sum;
  if (q.sbh) rc_sbh_quad(lv, du, cf, cgIdx == lastCg, lane);
  if (okQ)
  {
#pragma unroll
    for (int r = 0; r < 4; r++) level[(y0 + r) * W + xq] = lv[r];
  }
#pragma unroll
  for (int r = 0; r < 4; r++) cq[r] = rc_dequant_one(q, lv[r]);
  }
  else                                                                    // inverse transform only: the coefficients come from memory, in view Q
  {
    bool fits = true;
#pragma unroll
    for (int r = 0; r < 4; r++) { cq[r] = tiQ >= 0 ? level[(y0 + r) * W + xq] : 0; fits = fits && cq[r] >= -32768 && cq[r] <= 32767; }
    const unsigned long long badLanes = __builtin_amdgcn_ballot_w64(!fits);
    okQ = tiQ >= 0 && (badLanes & laneMaskQ(sxQ, syQ)) == 0ull; okL = tiL >= 0 && (badLanes & laneMaskQ(sxL, syL)) == 0ull;
    { const bool fb_ = badLanes != 0ull && lane < G && info[lane] >= 0 && (badLanes & laneMaskQ(lane % NX, lane / NX)) != 0ull; if (fb_) fbTi = info[lane]; fbMask |= __builtin_amdgcn_ballot_w64(fb_); }
    if (!okQ) { cq[0] = 0; cq[1] = 0; cq[2] = 0; cq[3] = 0; }
  }
  // ---- I1: Y1T[i][r] = sum_k Cq[k][i] Tv(sub-TU of i, r)[k][r]: A = Cq^T (row i = c), B = rows of Tv^T; the type also belongs to A's row -> passes
  int y1[4];
  {
    const h4 ah = rc_limb_h4(cq, true), al = rc_limb_h4(cq, false);
    f4 hi = { 0.f, 0.f, 0.f, 0.f }, lo = { 0.f, 0.f, 0.f, 0.f };
    const bool diag = syL == syQ;                                         // B: column r = c, k = 4 g ..
#pragma unroll
    for (int qq = 0; qq < NX; qq++)
    {
      const bool mine = NX == 1 || sxQ == qq;                             // A row i = c belongs to sub-TU column qq
      const h4 b = diag ? rc_frag4(tab, (info[16 + syL * NX + qq] >> 2) & 3, H, 1, rowL, (4 * g) & (H - 1)) : zero4;
      hi = __builtin_amdgcn_mfma_f32_16x16x16f16(mine ? ah : zero4, b, hi, 0, 0, 0);
      lo = __builtin_amdgcn_mfma_f32_16x16x16f16(mine ? al : zero4, b, lo, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) y1[r] = clip3(-(1 << 15), (1 << 15) - 1, ((((int)hi[r]) << 8) + (int)lo[r] + 256) >> 9);   // Y1T[horizontal frequency 4 g + r][sample row c]
  }
  // ---- I2: RT[x][r] = sum_i Th(sub-TU of x, r)[i][x] Y1T[i][r]: A = rows of Th^T (row x = c), B = Y1T; the type also belongs to B's column -> passes
  {
    const h4 bh = rc_limb_h4(y1, true), bl = rc_limb_h4(y1, false);
    f4 hi = { 0.f, 0.f, 0.f, 0.f }, lo = { 0.f, 0.f, 0.f, 0.f };
    const bool diag = sxQ == sxL;
#pragma unroll
    for (int p = 0; p < NY; p++)
    {
      const bool mine = NY == 1 || syL == p;                              // B / D column r = c belongs to sub-TU row p
      const h4 a = diag ? rc_frag4(tab, info[16 + p * NX + sxQ] & 3, W, 1, c & (W - 1), colL) : zero4;
      hi = __builtin_amdgcn_mfma_f32_16x16x16f16(a, mine ? bh : zero4, hi, 0, 0, 0);
      lo = __builtin_amdgcn_mfma_f32_16x16x16f16(a, mine ? bl : zero4, lo, 0, 0, 0);
    }
    const int s2i = (6 + 15 - 1) - bd + 2;
    if (okL)                                                              // residual of tile row c, columns 4 g ..: view L again
    {
      pel4 out;
#pragma unroll
      for (int r = 0; r < 4; r++)
      {
        const int resi = clip3(-(1 << 15), (1 << 15) - 1, ((((int)hi[r]) << 8) + (int)lo[r] + (1 << (s2i - 1))) >> s2i);
        out[r] = mode == RC_CHAIN ? (short)clip3(clpMin, clpMax, (int)pv[r] + (int)(short)resi) : (short)resi;
      }
      *reinterpret_cast<pel4*>(recBase + dL.rec_off + (size_t)rowL * dL.rec_stride + colL) = out;
    }
  }
  RC_WAVE_SYNC();                                                         // info is rewritten by the wave's next item
  rc_fallback_lanes<MODE>(fbMask, fbTi, descs, orgBase, predBase, recBase, levelBase, absSumOut, bd, clpMin, clpMax, tr32, dqInv, scanOff, fbScr, lane);
}

// Packed tiles with a 32- or 64-point side: 32 x 8 / 32 x 4 / 64 x 8 / 64 x 4 (16 / H TUs one above the other in a 16-row x W-column multi-tile) and 8 x 32 / 4 x 32 /
// 8 x 64 / 4 x 64 (16 / W TUs side by side in an H-row x 16-column multi-tile); a 64-point side keeps its 32 low frequencies (zero-out) and is always DCT-II.  The 32-point stages are the 16x16x32 products of mfma_tr.h (operand of a
// result tile in its register k order), the short side is block-diagonal as above; only the 32-point stages depend on the other operand's
// sub-TU and run in passes.
template <int W, int H, int MODE>
__device__ __noinline__ void rc_tile_packed_wl(const RcDesc* __restrict__ descs, const int* __restrict__ list, int cnt, int item,
                                                const Pel* __restrict__ orgBase, const Pel* __restrict__ predBase, Pel* __restrict__ recBase,
                                                TCoeff* __restrict__ levelBase, unsigned* __restrict__ absSumOut, int bd, int clpMin, int clpMax,
                                                const _Float16* tab, const unsigned short* __restrict__ dqInv, const int* __restrict__ scanOff,
                                                const int* __restrict__ tr32, int* __restrict__ fbScr, int* info, int lane)
{
  unsigned long long fbMask = 0ull;                                             // sub-TUs this tile cannot take (lane j < G: TU j of the tile): served behind it
  int fbTi = 0;
  constexpr int mode = MODE;

  constexpr int NY = 16 / H, G = NY, LH = H == 4 ? 2 : 3, LW = W == 32 ? 5 : 6, XS = W / 32, CT = W / 16, PITCH = W + 8;
  constexpr int NP = W == 64 ? 1 : NY;                 // passes of the long stages: a 64-point side is DCT-II for every TU (no type to tell apart)
  const int c = lane & 15, g = lane >> 4;
  const h4 zero4 = { (_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0 };
  const h8 zero8 = { (_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0 };
  {
    int ti = -1, types = 0;
    if (lane < G && item * G + lane < cnt) { ti = list[item * G + lane]; const RcDesc& d = descs[ti]; types = (int)d.tr_hor | ((int)d.tr_ver << 2); }
    if (lane < G) { info[lane] = ti; info[16 + lane] = types; }
  }
  RC_WAVE_SYNC();
  const int syL = c >> LH, syQ = (4 * g) >> LH;                          // sample views: tile row c; coefficient view: tile rows 4 g ..
  const int tiL = info[syL], tiQ = info[syQ], ti0 = info[0];
  auto laneMask = [&](int sy) -> unsigned long long { const unsigned long long cm = ((1ull << H) - 1ull) << (sy * H); return cm | (cm << 16) | (cm << 32) | (cm << 48); };
  auto laneMaskQ = [&](int sy) -> unsigned long long                     // lanes of the coefficient view that hold TU sy: row groups g of its rows, every column
  {
    unsigned long long m = 0;
#pragma unroll
    for (int gg = 0; gg < H / 4; gg++) m |= 0xFFFFull << (16 * (sy * (H / 4) + gg));
    return m;
  };
  const RcDesc& dL = descs[tiL >= 0 ? tiL : ti0];
  const RcDesc& dQ = descs[tiQ >= 0 ? tiQ : ti0];
  const int rowL = c & (H - 1), y0 = (4 * g) & (H - 1);
  TCoeff* level = levelBase + dQ.level_off;
  const bool diagV = syL == syQ;
  bool okL, okQ;
  int cqA[2][4];                                                          // the inverse stages' input
  if (mode != RC_INV)
  {
  h8 xa[XS];
#pragma unroll
  for (int sk = 0; sk < XS; sk++) xa[sk] = zero8;
  bool inRange = true;
  if (tiL >= 0)
  {
#pragma unroll
    for (int sk = 0; sk < XS; sk++)
    {
      const pel8 o = *reinterpret_cast<const pel8*>(orgBase + dL.org_off + (size_t)rowL * dL.org_stride + 32 * sk + 8 * g);
      pel8 pp = { 0, 0, 0, 0, 0, 0, 0, 0 };
      if (mode == RC_CHAIN) pp = *reinterpret_cast<const pel8*>(predBase + dL.pred_off + (size_t)rowL * dL.pred_stride + 32 * sk + 8 * g);
#pragma unroll
      for (int j = 0; j < 8; j++) { const int v = (int)o[j] - (int)pp[j]; inRange = inRange && v >= -1023 && v <= 1023; xa[sk][j] = (_Float16)(short)v; }
    }
  }
  const unsigned long long badLanes = __builtin_amdgcn_ballot_w64(!inRange);
  okL = tiL >= 0 && (badLanes & laneMask(syL)) == 0ull; okQ = tiQ >= 0 && (badLanes & laneMask(syQ)) == 0ull;
  { const bool fb_ = badLanes != 0ull && lane < G && info[lane] >= 0 && (badLanes & laneMask(lane)) != 0ull; if (fb_) fbTi = info[lane]; fbMask |= __builtin_amdgcn_ballot_w64(fb_); }
  if (!okL) {
#pragma unroll
    for (int sk = 0; sk < XS; sk++) xa[sk] = zero8;
  }
  const int s1 = LW + bd + 6 - 15 + 2, s2 = LH + 6 + 2;
  // ---- F1 (32-point, one pass per TU: the type belongs to the row of X)
  int t1[2][4];
  {
    f4 m1[2] = { { 0.f, 0.f, 0.f, 0.f }, { 0.f, 0.f, 0.f, 0.f } };
#pragma unroll
    for (int p = 0; p < NP; p++)
    {
      const _Float16* Th = tab + rc_tab_off(W == 64 ? 0 : info[16 + p] & 3, W, 0);
#pragma unroll
      for (int sk = 0; sk < XS; sk++)
      {
        const h8 a = (NP == 1 || syL == p) ? xa[sk] : zero8;
#pragma unroll
        for (int jt = 0; jt < 2; jt++)
          m1[jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, *reinterpret_cast<const h8*>(Th + (16 * jt + c) * PITCH + 32 * sk + 8 * g), m1[jt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int jt = 0; jt < 2; jt++)
#pragma unroll
      for (int r = 0; r < 4; r++) t1[jt][r] = ((int)m1[jt][r] + (1 << (s1 - 1))) >> s1;       // M1[row 4 g + r][frequency 16 jt + c]
  }
  // ---- F2 (block-diagonal H-point; the type belongs to the row of the matrix)
  int cf[2][4];
  {
    const h4 a = diagV ? rc_frag4(tab, (info[16 + syL] >> 2) & 3, H, 0, rowL, (4 * g) & (H - 1)) : zero4;
#pragma unroll
    for (int jt = 0; jt < 2; jt++)
    {
      const f4 hi = __builtin_amdgcn_mfma_f32_16x16x16f16(a, rc_limb_h4(t1[jt], true), f4{ 0.f, 0.f, 0.f, 0.f }, 0, 0, 0);
      const f4 lo = __builtin_amdgcn_mfma_f32_16x16x16f16(a, rc_limb_h4(t1[jt], false), f4{ 0.f, 0.f, 0.f, 0.f }, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; r++) cf[jt][r] = ((((int)hi[r]) << 8) + (int)lo[r] + (1 << (s2 - 1))) >> s2;   // C[vertical frequency 4 g + r][horizontal 16 jt + c]
    }
  }
  if (mode == RC_FWD)                                                     // forward transform only: the coefficients are the result
  {
    if (okQ)
    {
#pragma unroll
      for (int jt = 0; jt < 2; jt++)
#pragma unroll
        for (int r = 0; r < 4; r++)
        {
          level[(y0 + r) * W + 16 * jt + c] = cf[jt][r];
          if (W == 64) level[(y0 + r) * W + 32 + 16 * jt + c] = 0;
        }
    }
    RC_WAVE_SYNC();
    rc_fallback_lanes<MODE>(fbMask, fbTi, descs, orgBase, predBase, recBase, levelBase, absSumOut, bd, clpMin, clpMax, tr32, dqInv, scanOff, fbScr, lane);
    return;
  }
  // ---- quantiser: the lane's coefficients are rows y0 .. y0 + 3 of columns c and 16 + c of TU syQ
  const RcQ q = rc_qparams(W, H, dQ.qp, bd, dQ.intra_slice, dQ.sign_hiding);
  const unsigned short* inv = dqInv + scanOff[(LW - 1) * 6 + (LH - 1)];
  int lv[2][4], du[2][4], sum = 0, lastCg = -1, cgIdx[2];
#pragma unroll
  for (int jt = 0; jt < 2; jt++)
  {
#pragma unroll
    for (int r = 0; r < 4; r++) { int mag; lv[jt][r] = rc_quant_one(q, cf[jt][r], du[jt][r], mag); sum += mag; }
    cgIdx[jt] = (int)inv[y0 * W + 16 * jt + (c & ~3)] >> 4;
    if (rc_cg_nonzero(lv[jt])) lastCg = max(lastCg, cgIdx[jt]);
  }
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) { sum += __shfl_xor(sum, m); if (m >= 4) lastCg = max(lastCg, __shfl_xor(lastCg, m)); }
#pragma unroll
  for (int m = 16; m < 4 * H; m <<= 1) { sum += __shfl_xor(sum, m); lastCg = max(lastCg, __shfl_xor(lastCg, m)); }
  if (okQ && c == 0 && y0 == 0) absSumOut[tiQ] = (unsigned)sum;
#pragma unroll
  for (int jt = 0; jt < 2; jt++)
  {
    if (q.sbh) rc_sbh_quad(lv[jt], du[jt], cf[jt], cgIdx[jt] == lastCg, lane);
    if (okQ)
    {
#pragma unroll
      for (int r = 0; r < 4; r++) level[(y0 + r) * W + 16 * jt + c] = lv[jt][r];
      if (W == 64)                                                        // zero-out region: horizontal frequencies 32 .. 63 of the lane's four rows
      {
#pragma unroll
        for (int r = 0; r < 4; r++) level[(y0 + r) * W + 32 + 16 * jt + c] = 0;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) cqA[jt][r] = rc_dequant_one(q, lv[jt][r]);
  }
  }
  else                                                                    // inverse transform only: the coefficients come from memory
  {
    bool fits = true;
#pragma unroll
    for (int jt = 0; jt < 2; jt++)
#pragma unroll
      for (int r = 0; r < 4; r++) { cqA[jt][r] = tiQ >= 0 ? level[(y0 + r) * W + 16 * jt + c] : 0; fits = fits && cqA[jt][r] >= -32768 && cqA[jt][r] <= 32767; }
    const unsigned long long badLanes = __builtin_amdgcn_ballot_w64(!fits);
    okQ = tiQ >= 0 && (badLanes & laneMaskQ(syQ)) == 0ull; okL = tiL >= 0 && (badLanes & laneMaskQ(syL)) == 0ull;
    { const bool fb_ = badLanes != 0ull && lane < G && info[lane] >= 0 && (badLanes & laneMaskQ(lane)) != 0ull; if (fb_) fbTi = info[lane]; fbMask |= __builtin_amdgcn_ballot_w64(fb_); }
    if (!okQ)
    {
#pragma unroll
      for (int jt = 0; jt < 2; jt++)
#pragma unroll
        for (int r = 0; r < 4; r++) cqA[jt][r] = 0;
    }
  }
  // ---- I1 (block-diagonal H-point; the type belongs to the column r = c of the matrix operand)
  int y1[2][4];
  {
    const h4 b = diagV ? rc_frag4(tab, (info[16 + syL] >> 2) & 3, H, 1, rowL, (4 * g) & (H - 1)) : zero4;
#pragma unroll
    for (int jt = 0; jt < 2; jt++)
    {
      const f4 hi = __builtin_amdgcn_mfma_f32_16x16x16f16(rc_limb_h4(cqA[jt], true), b, f4{ 0.f, 0.f, 0.f, 0.f }, 0, 0, 0);
      const f4 lo = __builtin_amdgcn_mfma_f32_16x16x16f16(rc_limb_h4(cqA[jt], false), b, f4{ 0.f, 0.f, 0.f, 0.f }, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; r++) y1[jt][r] = clip3(-(1 << 15), (1 << 15) - 1, ((((int)hi[r]) << 8) + (int)lo[r] + 256) >> 9);   // Y1T[frequency 16 jt + 4 g + r][row c]
    }
  }
  // ---- I2 (32-point over the frequency i, operand in result-tile k order; the type belongs to the column r = c of Y1T: one pass per TU)
  {
    h8 bh[1], bl[1];
    rc_tile_frags<32>(bh, bl, y1);
    f4 hi[CT], lo[CT];
#pragma unroll
    for (int xt = 0; xt < CT; xt++) { hi[xt] = f4{ 0.f, 0.f, 0.f, 0.f }; lo[xt] = f4{ 0.f, 0.f, 0.f, 0.f }; }
#pragma unroll
    for (int p = 0; p < NP; p++)
    {
      const bool mine = NP == 1 || syL == p;
      const _Float16* ThT = tab + rc_tab_off(W == 64 ? 0 : info[16 + p] & 3, W, 1);
#pragma unroll
      for (int xt = 0; xt < CT; xt++)
      {
        const h8 a = rc_mat_frag32(ThT, PITCH, 16 * xt + c, 0, g);           // the kept frequencies i < 32 of row x
        hi[xt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, mine ? bh[0] : zero8, hi[xt], 0, 0, 0);
        lo[xt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, mine ? bl[0] : zero8, lo[xt], 0, 0, 0);
      }
    }
    const int s2i = (6 + 15 - 1) - bd + 2;
    if (okL)
    {
      const Pel* pr = predBase + dL.pred_off + (size_t)rowL * dL.pred_stride;
      Pel* rec = recBase + dL.rec_off + (size_t)rowL * dL.rec_stride;
#pragma unroll
      for (int xt = 0; xt < CT; xt++)                                     // residual of tile row c, columns 16 xt + 4 g ..
      {
        pel4 pv = { 0, 0, 0, 0 };
        if (mode == RC_CHAIN) pv = *reinterpret_cast<const pel4*>(pr + 16 * xt + 4 * g);
        pel4 out;
#pragma unroll
        for (int r = 0; r < 4; r++)
        {
          const int resi = clip3(-(1 << 15), (1 << 15) - 1, ((((int)hi[xt][r]) << 8) + (int)lo[xt][r] + (1 << (s2i - 1))) >> s2i);
          out[r] = mode == RC_CHAIN ? (short)clip3(clpMin, clpMax, (int)pv[r] + (int)(short)resi) : (short)resi;
        }
        *reinterpret_cast<pel4*>(rec + 16 * xt + 4 * g) = out;
      }
    }
  }
  RC_WAVE_SYNC();
  rc_fallback_lanes<MODE>(fbMask, fbTi, descs, orgBase, predBase, recBase, levelBase, absSumOut, bd, clpMin, clpMax, tr32, dqInv, scanOff, fbScr, lane);
}

template <int W, int H, int MODE>
__device__ __noinline__ void rc_tile_packed_hl(const RcDesc* __restrict__ descs, const int* __restrict__ list, int cnt, int item,
                                                const Pel* __restrict__ orgBase, const Pel* __restrict__ predBase, Pel* __restrict__ recBase,
                                                TCoeff* __restrict__ levelBase, unsigned* __restrict__ absSumOut, int bd, int clpMin, int clpMax,
                                                const _Float16* tab, const unsigned short* __restrict__ dqInv, const int* __restrict__ scanOff,
                                                const int* __restrict__ tr32, int* __restrict__ fbScr, int* info, int lane)
{
  unsigned long long fbMask = 0ull;                                             // sub-TUs this tile cannot take (lane j < G: TU j of the tile): served behind it
  int fbTi = 0;
  constexpr int mode = MODE;

  constexpr int NX = 16 / W, G = NX, LW = W == 4 ? 2 : 3, LH = H == 32 ? 5 : 6, RT = H / 16, KS = H / 32, PITCH = H + 8;
  constexpr int NP = H == 64 ? 1 : NX;                 // passes of the long stages: a 64-point side is DCT-II for every TU
  const int c = lane & 15, g = lane >> 4;
  const h4 zero4 = { (_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0 };
  const h8 zero8 = { (_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0 };
  {
    int ti = -1, types = 0;
    if (lane < G && item * G + lane < cnt) { ti = list[item * G + lane]; const RcDesc& d = descs[ti]; types = (int)d.tr_hor | ((int)d.tr_ver << 2); }
    if (lane < G) { info[lane] = ti; info[16 + lane] = types; }
  }
  RC_WAVE_SYNC();
  const int sxL = (4 * g) >> LW, sxQ = c >> LW;                          // sample view: tile columns 4 g ..; coefficient view: tile column c
  const int tiL = info[sxL], tiQ = info[sxQ], ti0 = info[0];
  auto laneMask = [&](int sx) -> unsigned long long
  {
    unsigned long long m = 0;
#pragma unroll
    for (int gg = 0; gg < W / 4; gg++) m |= 0xFFFFull << (16 * (sx * (W / 4) + gg));
    return m;
  };
  auto laneMaskQ = [&](int sx) -> unsigned long long                     // lanes of the coefficient view that hold TU sx: its columns c, every row group
  {
    const unsigned long long cm = ((1ull << W) - 1ull) << (sx * W);
    return cm | (cm << 16) | (cm << 32) | (cm << 48);
  };
  const RcDesc& dL = descs[tiL >= 0 ? tiL : ti0];
  const RcDesc& dQ = descs[tiQ >= 0 ? tiQ : ti0];
  const int colL = (4 * g) & (W - 1), xq = c & (W - 1);
  TCoeff* level = levelBase + dQ.level_off;
  const bool diagH = sxQ == sxL;
  pel4 pv[RT];
#pragma unroll
  for (int rt = 0; rt < RT; rt++) pv[rt] = pel4{ 0, 0, 0, 0 };
  bool okL, okQ;
  int cqA[2][4];                                                          // the inverse stages' input
  if (mode != RC_INV)
  {
  h4 xa[RT];
#pragma unroll
  for (int rt = 0; rt < RT; rt++) xa[rt] = zero4;
  bool inRange = true;
  if (tiL >= 0)
  {
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
    {
      const pel4 o = *reinterpret_cast<const pel4*>(orgBase + dL.org_off + (size_t)(16 * rt + c) * dL.org_stride + colL);
      if (mode == RC_CHAIN) pv[rt] = *reinterpret_cast<const pel4*>(predBase + dL.pred_off + (size_t)(16 * rt + c) * dL.pred_stride + colL);
#pragma unroll
      for (int j = 0; j < 4; j++) { const int v = (int)o[j] - (int)pv[rt][j]; inRange = inRange && v >= -1023 && v <= 1023; xa[rt][j] = (_Float16)(short)v; }
    }
  }
  const unsigned long long badLanes = __builtin_amdgcn_ballot_w64(!inRange);
  okL = tiL >= 0 && (badLanes & laneMask(sxL)) == 0ull; okQ = tiQ >= 0 && (badLanes & laneMask(sxQ)) == 0ull;
  { const bool fb_ = badLanes != 0ull && lane < G && info[lane] >= 0 && (badLanes & laneMask(lane)) != 0ull; if (fb_) fbTi = info[lane]; fbMask |= __builtin_amdgcn_ballot_w64(fb_); }
  if (!okL) {
#pragma unroll
    for (int rt = 0; rt < RT; rt++) xa[rt] = zero4;
  }
  const int s1 = LW + bd + 6 - 15 + 2, s2 = LH + 6 + 2;
  // ---- F1 (block-diagonal W-point; the type belongs to the column j = c of the matrix operand)
  int t1[RT][4];
  {
    const h4 b = diagH ? rc_frag4(tab, info[16 + sxQ] & 3, W, 0, c & (W - 1), colL) : zero4;
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
    {
      const f4 m1 = __builtin_amdgcn_mfma_f32_16x16x16f16(xa[rt], b, f4{ 0.f, 0.f, 0.f, 0.f }, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; r++) t1[rt][r] = ((int)m1[r] + (1 << (s1 - 1))) >> s1;         // M1[row 16 rt + 4 g + r][frequency c]
    }
  }
  // ---- F2 (32-point over the rows, operand in result-tile k order; the type belongs to the column j1 = c of M1: one pass per TU)
  int cf[2][4];
  {
    h8 bh[KS], bl[KS];
    rc_tile_frags<H>(bh, bl, t1);
    f4 hi[2] = { { 0.f, 0.f, 0.f, 0.f }, { 0.f, 0.f, 0.f, 0.f } }, lo[2] = { { 0.f, 0.f, 0.f, 0.f }, { 0.f, 0.f, 0.f, 0.f } };
#pragma unroll
    for (int qq = 0; qq < NP; qq++)
    {
      const bool mine = NP == 1 || sxQ == qq;
      const _Float16* Tv = tab + rc_tab_off(H == 64 ? 0 : (info[16 + qq] >> 2) & 3, H, 0);
#pragma unroll
      for (int it = 0; it < 2; it++)                                      // the kept vertical frequencies: 32
#pragma unroll
        for (int sk = 0; sk < KS; sk++)
        {
          const h8 a = rc_mat_frag32(Tv, PITCH, 16 * it + c, sk, g);
          hi[it] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, mine ? bh[sk] : zero8, hi[it], 0, 0, 0);
          lo[it] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, mine ? bl[sk] : zero8, lo[it], 0, 0, 0);
        }
    }
#pragma unroll
    for (int it = 0; it < 2; it++)
#pragma unroll
      for (int r = 0; r < 4; r++) cf[it][r] = ((((int)hi[it][r]) << 8) + (int)lo[it][r] + (1 << (s2 - 1))) >> s2;   // C[vertical frequency 16 it + 4 g + r][horizontal c]
  }
  if (mode == RC_FWD)                                                     // forward transform only: the coefficients are the result
  {
    if (okQ)
    {
#pragma unroll
      for (int it = 0; it < 2; it++)
#pragma unroll
        for (int r = 0; r < 4; r++)
        {
          level[(16 * it + 4 * g + r) * W + xq] = cf[it][r];
          if (H == 64) level[(32 + 16 * it + 4 * g + r) * W + xq] = 0;
        }
    }
    RC_WAVE_SYNC();
    rc_fallback_lanes<MODE>(fbMask, fbTi, descs, orgBase, predBase, recBase, levelBase, absSumOut, bd, clpMin, clpMax, tr32, dqInv, scanOff, fbScr, lane);
    return;
  }
  // ---- quantiser: the lane's coefficients are rows 16 it + 4 g .. of column xq of TU sxQ
  const RcQ q = rc_qparams(W, H, dQ.qp, bd, dQ.intra_slice, dQ.sign_hiding);
  const unsigned short* inv = dqInv + scanOff[(LW - 1) * 6 + (LH - 1)];
  int lv[2][4], du[2][4], sum = 0, lastCg = -1, cgIdx[2];
#pragma unroll
  for (int it = 0; it < 2; it++)
  {
#pragma unroll
    for (int r = 0; r < 4; r++) { int mag; lv[it][r] = rc_quant_one(q, cf[it][r], du[it][r], mag); sum += mag; }
    cgIdx[it] = (int)inv[(16 * it + 4 * g) * W + (xq & ~3)] >> 4;
    if (rc_cg_nonzero(lv[it])) lastCg = max(lastCg, cgIdx[it]);
  }
#pragma unroll
  for (int m = 1; m < W; m <<= 1) { sum += __shfl_xor(sum, m); if (m >= 4) lastCg = max(lastCg, __shfl_xor(lastCg, m)); }
#pragma unroll
  for (int m = 16; m < 64; m <<= 1) { sum += __shfl_xor(sum, m); lastCg = max(lastCg, __shfl_xor(lastCg, m)); }
  if (okQ && xq == 0 && g == 0) absSumOut[tiQ] = (unsigned)sum;
#pragma unroll
  for (int it = 0; it < 2; it++)
  {
    if (q.sbh) rc_sbh_quad(lv[it], du[it], cf[it], cgIdx[it] == lastCg, lane);
    if (okQ)
    {
#pragma unroll
      for (int r = 0; r < 4; r++) level[(16 * it + 4 * g + r) * W + xq] = lv[it][r];
      if (H == 64)                                                        // zero-out region: vertical frequencies 32 .. 63 of the lane's column
      {
#pragma unroll
        for (int r = 0; r < 4; r++) level[(32 + 16 * it + 4 * g + r) * W + xq] = 0;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) cqA[it][r] = rc_dequant_one(q, lv[it][r]);
  }
  }
  else                                                                    // inverse transform only: the coefficients come from memory
  {
    bool fits = true;
#pragma unroll
    for (int it = 0; it < 2; it++)
#pragma unroll
      for (int r = 0; r < 4; r++) { cqA[it][r] = tiQ >= 0 ? level[(16 * it + 4 * g + r) * W + xq] : 0; fits = fits && cqA[it][r] >= -32768 && cqA[it][r] <= 32767; }
    const unsigned long long badLanes = __builtin_amdgcn_ballot_w64(!fits);
    okQ = tiQ >= 0 && (badLanes & laneMaskQ(sxQ)) == 0ull; okL = tiL >= 0 && (badLanes & laneMaskQ(sxL)) == 0ull;
    { const bool fb_ = badLanes != 0ull && lane < G && info[lane] >= 0 && (badLanes & laneMaskQ(lane)) != 0ull; if (fb_) fbTi = info[lane]; fbMask |= __builtin_amdgcn_ballot_w64(fb_); }
    if (!okQ)
    {
#pragma unroll
      for (int it = 0; it < 2; it++)
#pragma unroll
        for (int r = 0; r < 4; r++) cqA[it][r] = 0;
    }
  }
  // ---- I1 (32-point over the vertical frequency, A = Cq^T in result-tile k order; the type belongs to A's row i = c: one pass per TU)
  int y1[RT][4];
  {
    h8 ah[1], al[1];
    rc_tile_frags<32>(ah, al, cqA);
    f4 hi[RT], lo[RT];
#pragma unroll
    for (int rt = 0; rt < RT; rt++) { hi[rt] = f4{ 0.f, 0.f, 0.f, 0.f }; lo[rt] = f4{ 0.f, 0.f, 0.f, 0.f }; }
#pragma unroll
    for (int qq = 0; qq < NP; qq++)
    {
      const bool mine = NP == 1 || sxQ == qq;
      const _Float16* TvT = tab + rc_tab_off(H == 64 ? 0 : (info[16 + qq] >> 2) & 3, H, 1);
#pragma unroll
      for (int rt = 0; rt < RT; rt++)
      {
        const h8 b = rc_mat_frag32(TvT, PITCH, 16 * rt + c, 0, g);          // the kept frequencies k < 32 of row r
        hi[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(mine ? ah[0] : zero8, b, hi[rt], 0, 0, 0);
        lo[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(mine ? al[0] : zero8, b, lo[rt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
      for (int r = 0; r < 4; r++) y1[rt][r] = clip3(-(1 << 15), (1 << 15) - 1, ((((int)hi[rt][r]) << 8) + (int)lo[rt][r] + 256) >> 9);   // Y1T[frequency 4 g + r][row 16 rt + c]
  }
  // ---- I2 (block-diagonal W-point; the type belongs to the row x = c of the matrix)
  {
    const h4 a = diagH ? rc_frag4(tab, info[16 + sxQ] & 3, W, 1, c & (W - 1), colL) : zero4;
    const int s2i = (6 + 15 - 1) - bd + 2;
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
    {
      const f4 hi = __builtin_amdgcn_mfma_f32_16x16x16f16(a, rc_limb_h4(y1[rt], true), f4{ 0.f, 0.f, 0.f, 0.f }, 0, 0, 0);
      const f4 lo = __builtin_amdgcn_mfma_f32_16x16x16f16(a, rc_limb_h4(y1[rt], false), f4{ 0.f, 0.f, 0.f, 0.f }, 0, 0, 0);
      if (okL)                                                            // residual of tile row 16 rt + c, columns 4 g ..: the sample view again
      {
        pel4 out;
#pragma unroll
        for (int r = 0; r < 4; r++)
        {
          const int resi = clip3(-(1 << 15), (1 << 15) - 1, ((((int)hi[r]) << 8) + (int)lo[r] + (1 << (s2i - 1))) >> s2i);
          out[r] = mode == RC_CHAIN ? (short)clip3(clpMin, clpMax, (int)pv[rt][r] + (int)(short)resi) : (short)resi;
        }
        *reinterpret_cast<pel4*>(recBase + dL.rec_off + (size_t)(16 * rt + c) * dL.rec_stride + colL) = out;
      }
    }
  }
  RC_WAVE_SYNC();
  rc_fallback_lanes<MODE>(fbMask, fbTi, descs, orgBase, predBase, recBase, levelBase, absSumOut, bd, clpMin, clpMax, tr32, dqInv, scanOff, fbScr, lane);
}

// ---------------------------------------------------------------------------------------------------
// Generic path: one wave per TU, any W x H in 2..64; the residual / intermediates / coefficients live in two LDS buffers of the wave.
// Matrices come from global memory (int32 tables).  Exact 32-bit arithmetic as the reference's `int` loops.
__device__ __forceinline__ const int* rc_t32(const int* tr32, int type, int n) { return tr32 + type * 5460 + (n * n - 4) / 3; }
// LDS copy of the matrices the generic path reads (rc_generic_kernel<512, 4>): per type the sizes 2 .. 32 (1364 ints), then DCT-II 64
constexpr int RC_GT_TYPE = 1364, RC_GT_INTS = 3 * RC_GT_TYPE + 4096;
__device__ __forceinline__ const int* rc_t32_lds(const int* tabL, int type, int n) { return n == 64 ? tabL + 3 * RC_GT_TYPE : tabL + type * RC_GT_TYPE + (n * n - 4) / 3; }

template <int MODE = RC_CHAIN>
__device__ void rc_tu_generic(const RcDesc& d, const Pel* __restrict__ orgBase, const Pel* __restrict__ predBase, Pel* __restrict__ recBase,
                              TCoeff* __restrict__ levelBase, unsigned* __restrict__ absSumOut, int ti, int bd, int clpMin, int clpMax,
                              const int* __restrict__ tr32, const unsigned short* __restrict__ dqInv, const int* __restrict__ scanOff,
                              int* bufA, int* bufB, int lane, const int* tabL = nullptr)
{
  constexpr int mode = MODE;
  const int w = d.w, h = d.h, lw = ilog2(w), lh = ilog2(h);
  const int wj = w > 32 ? 32 : w, hj = h > 32 ? 32 : h;
  const Pel* org = orgBase + d.org_off;
  const Pel* pred = predBase + d.pred_off;
  const int* Th = tabL ? rc_t32_lds(tabL, d.tr_hor, w) : rc_t32(tr32, d.tr_hor, w);      // the matrix entries sit on the inner loops: LDS when the kernel has a copy
  const int* Tv = tabL ? rc_t32_lds(tabL, d.tr_ver, h) : rc_t32(tr32, d.tr_ver, h);
  TCoeff* level = levelBase + d.level_off;
  Pel* rec = recBase + d.rec_off;
  if (mode != RC_CHAIN && d.tr_hor == 3)                 // transform skip of the plain transform entries (TrQuant.cpp:795-847): element-wise
  {
    int shift = 15 - bd - ((lw + lh) >> 1), scale = 1;
    if ((lw + lh) & 1) { shift += mode == RC_FWD ? -8 : 7; scale = 181; }
    for (int e = lane; e < w * h; e += 64)
    {
      const int r = e >> lw, k = e & (w - 1);
      if (mode == RC_FWD)
      {
        const int v = (int)org[(size_t)r * d.org_stride + k] * scale;
        level[e] = shift >= 0 ? v << shift : (v + (1 << (-shift - 1))) >> -shift;
      }
      else
      {
        const int cc = level[e] * scale;
        rec[(size_t)r * d.rec_stride + k] = (short)(shift >= 0 ? (cc + (shift ? 1 << (shift - 1) : 0)) >> shift : cc << -shift);
      }
    }
    return;
  }
  const int s1 = lw + bd + 6 - 15 + 2, s2 = lh + 6 + 2;
  if (mode != RC_INV)
  {
  for (int e = lane; e < w * h; e += 64)
  {
    const int r = e >> lw, k = e & (w - 1);
    bufA[e] = (int)org[(size_t)r * d.org_stride + k] - (mode == RC_CHAIN ? (int)pred[(size_t)r * d.pred_stride + k] : 0);
  }
  RC_WAVE_SYNC();
  for (int e = lane; e < wj * h; e += 64)                // F1: bufB[j * h + r]
  {
    const int j = e >> lh, r = e & (h - 1);
    int sum = 0;
#pragma unroll 4
    for (int k = 0; k < w; k++) sum += bufA[r * w + k] * Th[j * w + k];
    bufB[e] = (sum + (1 << (s1 - 1))) >> s1;
  }
  RC_WAVE_SYNC();
  for (int e = lane; e < w * h; e += 64)                 // F2: coefficients, raster, zero outside the kept region
  {
    const int j2 = e >> lw, j1 = e & (w - 1);
    int v = 0;
    if (j2 < hj && j1 < wj)
    {
      int sum = 0;
#pragma unroll 4
      for (int r = 0; r < h; r++) sum += bufB[j1 * h + r] * Tv[j2 * h + r];
      v = (sum + (1 << (s2 - 1))) >> s2;
    }
    bufA[e] = v;
  }
  RC_WAVE_SYNC();
  if (mode == RC_FWD)                                    // forward transform only: the coefficients (zero outside the kept region) are the result
  {
    for (int e = lane; e < w * h; e += 64) level[e] = bufA[e];
    RC_WAVE_SYNC();
    return;
  }
  // quantiser: lane = column (chunks of 64 columns never occur: w <= 64), four rows per pass
  const RcQ q = rc_qparams(w, h, d.qp, bd, d.intra_slice, d.sign_hiding);
  int sum = 0;
  if (!q.sbh)
  {
    for (int e = lane; e < w * h; e += 64)
    {
      int duu, mag;
      const int l = rc_quant_one(q, bufA[e], duu, mag);
      sum += mag;
      level[e] = l;
      bufA[e] = rc_dequant_one(q, l);
    }
  }
  else
  {
    const unsigned short* inv = dqInv + scanOff[(lw - 1) * 6 + (lh - 1)];
    const int col = lane & (w - 1), rowsPerPass = 4 * (64 >> lw);          // narrow TUs: several groups of 4 rows side by side in the wave
    const int rsub = (lane >> lw) * 4;
    int lastCg = -1;
    for (int r0 = 0; r0 < h; r0 += rowsPerPass)
    {
      const int row = r0 + rsub;
      int lv[4], duu[4];
      const bool on = row < h;
#pragma unroll
      for (int r = 0; r < 4; r++) { int mag; lv[r] = rc_quant_one(q, on ? bufA[(row + r) * w + col] : 0, duu[r], mag); sum += on ? mag : 0; }
      const bool nz = rc_cg_nonzero(lv);
      if (on && nz) lastCg = max(lastCg, (int)inv[row * w + (col & ~3)] >> 4);
    }
    lastCg = wave_max_i32(lastCg);
    for (int r0 = 0; r0 < h; r0 += rowsPerPass)
    {
      const int row = r0 + rsub;
      const bool on = row < h;
      int lv[4], duu[4], cfv[4];
#pragma unroll
      for (int r = 0; r < 4; r++) { int mag; cfv[r] = on ? bufA[(row + r) * w + col] : 0; lv[r] = rc_quant_one(q, cfv[r], duu[r], mag); }
      const int cg = on ? (int)inv[row * w + (col & ~3)] >> 4 : -2;
      rc_sbh_quad(lv, duu, cfv, cg == lastCg, lane);
      if (on)
      {
#pragma unroll
        for (int r = 0; r < 4; r++) { level[(row + r) * w + col] = lv[r]; bufA[(row + r) * w + col] = rc_dequant_one(q, lv[r]); }
      }
    }
  }
  sum = wave_sum_i32(sum);
  if (lane == 0) absSumOut[ti] = (unsigned)sum;
  }
  else
  {
    for (int e = lane; e < w * h; e += 64) bufA[e] = level[e];       // inverse transform only: the coefficients come from memory
  }
  RC_WAVE_SYNC();
  for (int e = lane; e < wj * h; e += 64)                // I1 (vertical): bufB[i * h + r] = clip(sum_k Cq[k][i] Tv[k][r])
  {
    const int i = e >> lh, r = e & (h - 1);
    int acc = 0;
#pragma unroll 4
    for (int k = 0; k < hj; k++) acc += bufA[k * w + i] * Tv[k * h + r];
    bufB[e] = clip3(-(1 << 15), (1 << 15) - 1, (acc + 256) >> 9);
  }
  RC_WAVE_SYNC();
  const int s2i = (6 + 15 - 1) - bd + 2;
  for (int e = lane; e < w * h; e += 64)                 // I2 (horizontal) + reconstruction
  {
    const int r = e >> lw, x = e & (w - 1);
    int acc = 0;
#pragma unroll 4
    for (int i = 0; i < wj; i++) acc += bufB[i * h + r] * Th[i * w + x];
    const int resi = (short)clip3(-(1 << 15), (1 << 15) - 1, (acc + (1 << (s2i - 1))) >> s2i);
    rec[(size_t)r * d.rec_stride + x] = mode == RC_CHAIN ? (short)clip3(clpMin, clpMax, (int)pred[(size_t)r * d.pred_stride + x] + resi) : (short)resi;
  }
  RC_WAVE_SYNC();
}

// ---------------------------------------------------------------------------------------------------
// generic kernel: the class-`generic` list, then the fall-back list of the matrix-core kernels (TUs whose residual left +-1023)
// The class-`generic` list holds TUs with a side of at most 8 (both sides >= 16 are matrix-core classes), i.e. at most 64 x 8 samples: 512-int
// buffers, four waves per workgroup (eight times the waves per compute unit of a 4096-int single-wave form).  The fall-back list of the
// matrix-core classes (a residual outside +-1023) holds TUs of up to 64 x 64.
template <int MODE>
__device__ __noinline__ void rc_fallback_call(const RcDesc* __restrict__ descs, int ti, const Pel* __restrict__ orgBase, const Pel* __restrict__ predBase, Pel* __restrict__ recBase,
                                              TCoeff* __restrict__ levelBase, unsigned* __restrict__ absSumOut, int bd, int clpMin, int clpMax,
                                              const int* __restrict__ tr32, const unsigned short* __restrict__ dqInv, const int* __restrict__ scanOff, int* fbScr, int lane)
{
  rc_tu_generic<MODE>(descs[ti], orgBase, predBase, recBase, levelBase, absSumOut, ti, bd, clpMin, clpMax, tr32, dqInv, scanOff, fbScr, fbScr + 4096, lane, nullptr);
}

// The class-`generic` list (TUs with a side of 2: chroma of 4-wide luma TUs): four waves per workgroup with 512-int buffers each and the matrices in LDS.
// (Until round 6 this launch also served the fall-back list of the matrix-core bodies; those TUs are now served in place, rc_fallback_call.)
// (launch bounds as the chain kernel's: the generic body is ONE function for both kernels, compiled for the looser of its callers' register budgets)
template <int MODE>
__global__ __launch_bounds__(256, 3) void rc_generic_kernel(const Pel* __restrict__ orgBase, const Pel* __restrict__ predBase, Pel* __restrict__ recBase,
                                                        TCoeff* __restrict__ levelBase, const RcDesc* __restrict__ descs,
                                                        const int* __restrict__ countA, const int* __restrict__ listA,
                                                        unsigned* __restrict__ absSumOut, int bd, int clpMin, int clpMax, VvcTrTables tb)
{
  __shared__ int lds[2 * 4 * 512 + RC_GT_INTS];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ca = countA[0];
  if ((int)blockIdx.x * 4 >= ca) return;
  int* tabL = lds + 2 * 4 * 512;
  for (int e = threadIdx.x; e < RC_GT_INTS; e += 256)
    tabL[e] = e < 3 * RC_GT_TYPE ? tb.tr32[(e / RC_GT_TYPE) * 5460 + e % RC_GT_TYPE] : tb.tr32[1364 + e - 3 * RC_GT_TYPE];
  __syncthreads();
  for (int k = blockIdx.x * 4 + wave; k < ca; k += (int)gridDim.x * 4)
  {
    const int ti = listA[k];
    rc_tu_generic<MODE>(descs[ti], orgBase, predBase, recBase, levelBase, absSumOut, ti, bd, clpMin, clpMax, tb.tr32, tb.dqInv, tb.scanOff, lds + wave * 1024,
                        lds + wave * 1024 + 512, lane, tabL);
  }
}

// ---------------------------------------------------------------------------------------------------
// (multiplies: 24-bit, full rate -- operands are at most 18-bit intermediates and 8-bit matrix entries; the 32-bit v_mul_lo_u32 is a quarter-rate instruction)
// 4 x 4 and 8 x 8: lane groups of S lanes per TU, G = 64 / S TUs per wave.  Forward: lane = row (stage 1), transposed through LDS, lane = column
// (stage 2, quantiser, de-quantiser, vertical inverse stage), transposed back, lane = row (horizontal inverse stage + reconstruction: the
// prediction row is still in the lane's registers).
template <int S, int MODE>
__device__ __noinline__ void rc_small_group(const RcDesc* __restrict__ descs, const int* __restrict__ list, int cnt, int item,
                                               const Pel* __restrict__ orgBase, const Pel* __restrict__ predBase, Pel* __restrict__ recBase,
                                               TCoeff* __restrict__ levelBase, unsigned* __restrict__ absSumOut, int bd, int clpMin, int clpMax,
                                               const RcSmallTab& tabs, int* tmpL, int lane)
{
  constexpr int mode = MODE;
  RC_IS_LDS(tmpL);

  constexpr int G = 64 / S, LS = S == 4 ? 2 : 3, TO = S == 4 ? 0 : 16;
  typedef short pelS __attribute__((ext_vector_type(S)));
  const int tg = lane / S, li = lane % S;
  const int k = item * G + tg;
  const bool act = k < cnt;
  const int ti = list[act ? k : 0];
  const RcDesc d = descs[ti];
  int* tt = tmpL + tg * (S * (S + 1));
  const int* Th = tabs.t[d.tr_hor] + TO;
  const int* Tv = tabs.t[d.tr_ver] + TO;
  const int* ThT = tabs.tt[d.tr_hor] + TO;
  const int* TvT = tabs.tt[d.tr_ver] + TO;
  // stage F1: lane = row li
  pelS p;
#pragma unroll
  for (int jj = 0; jj < S; jj++) p[jj] = 0;
  const int s1 = LS + bd + 6 - 15 + 2, s2 = LS + 6 + 2;
  int cq[S];                                                              // the inverse stages' input: column li
  TCoeff* level = levelBase + d.level_off;
  if (mode != RC_INV)
  {
  const pelS o = *reinterpret_cast<const pelS*>(orgBase + d.org_off + (size_t)li * d.org_stride);
  if (mode == RC_CHAIN) p = *reinterpret_cast<const pelS*>(predBase + d.pred_off + (size_t)li * d.pred_stride);
  {
    int x[S];
#pragma unroll
    for (int j = 0; j < S; j++) x[j] = (int)o[j] - (int)p[j];
#pragma unroll
    for (int j = 0; j < S; j++)
    {
      int sum = 0;
#pragma unroll
      for (int kk = 0; kk < S; kk++) sum += __mul24(x[kk], Th[j * S + kk]);
      tt[j * (S + 1) + li] = (sum + (1 << (s1 - 1))) >> s1;
    }
  }
  RC_WAVE_SYNC();
  // stage F2: lane = column li (horizontal frequency), registers = rows
  int cf[S];
  {
    int t[S];
#pragma unroll
    for (int r = 0; r < S; r++) t[r] = tt[li * (S + 1) + r];
#pragma unroll
    for (int j = 0; j < S; j++)
    {
      int sum = 0;
#pragma unroll
      for (int r = 0; r < S; r++) sum += __mul24(t[r], Tv[j * S + r]);
      cf[j] = (sum + (1 << (s2 - 1))) >> s2;
    }
  }
  RC_WAVE_SYNC();
  if (mode == RC_FWD)                                                     // forward transform only: the coefficients are the result
  {
    if (act)
    {
#pragma unroll
      for (int j = 0; j < S; j++) level[j * S + li] = cf[j];
    }
    return;
  }
  // quantiser (coefficient groups: rows 4 R .. 4 R + 3 x the lane's aligned quad)
  const RcQ q = rc_qparams(S, S, d.qp, bd, d.intra_slice, d.sign_hiding);
  int lv[S], du[S], sum = 0;
#pragma unroll
  for (int j = 0; j < S; j++) { int mag; lv[j] = rc_quant_one(q, cf[j], du[j], mag); sum += mag; }
#pragma unroll
  for (int m = 1; m < S; m <<= 1) sum += __shfl_xor(sum, m);
  if (act && li == 0) absSumOut[ti] = (unsigned)sum;
  if (q.sbh)
  {
    // coefficient-group scan index inside the TU: 4x4: one group; 8x8: (cx, cy) -> 2 cx + cy (diagonal scan of the 2 x 2 group grid)
    int lastCg = -1;
#pragma unroll
    for (int R = 0; R < S / 4; R++)
    {
      int l4[4] = { lv[4 * R], lv[4 * R + 1], lv[4 * R + 2], lv[4 * R + 3] };
      if (rc_cg_nonzero(l4)) lastCg = max(lastCg, 2 * (li >> 2) + R);
    }
    if (S == 8) lastCg = max(lastCg, __shfl_xor(lastCg, 4));
#pragma unroll
    for (int R = 0; R < S / 4; R++)
    {
      int l4[4] = { lv[4 * R], lv[4 * R + 1], lv[4 * R + 2], lv[4 * R + 3] };
      const int d4[4] = { du[4 * R], du[4 * R + 1], du[4 * R + 2], du[4 * R + 3] };
      const int c4[4] = { cf[4 * R], cf[4 * R + 1], cf[4 * R + 2], cf[4 * R + 3] };
      rc_sbh_quad(l4, d4, c4, 2 * (li >> 2) + R == lastCg, lane);
#pragma unroll
      for (int r = 0; r < 4; r++) lv[4 * R + r] = l4[r];
    }
  }
  if (act)
  {
#pragma unroll
    for (int j = 0; j < S; j++) level[j * S + li] = lv[j];
  }
#pragma unroll
  for (int j = 0; j < S; j++) cq[j] = rc_dequant_one(q, lv[j]);
  }
  else
  {
#pragma unroll
    for (int j = 0; j < S; j++) cq[j] = act ? level[j * S + li] : 0;      // any 32-bit coefficient: exact 32-bit multiplies below
  }
  // stage I1 (vertical): y[r] = clip(sum_k Cq[k] Tv[k][r]); written transposed: tt[r][column li]
  {
    bool fits24 = true;                                                   // 24-bit multiplies when every coefficient of the wave allows it (always, for de-quantiser output)
    if (mode != RC_CHAIN)
    {
#pragma unroll
      for (int kk = 0; kk < S; kk++) fits24 = fits24 && cq[kk] >= -(1 << 23) && cq[kk] < (1 << 23);
    }
    const bool narrow = mode == RC_CHAIN || __builtin_amdgcn_ballot_w64(!fits24) == 0ull;
#pragma unroll
    for (int r = 0; r < S; r++)
    {
      int acc = 0;
      if (narrow)
      {
#pragma unroll
        for (int kk = 0; kk < S; kk++) acc += __mul24(cq[kk], TvT[r * S + kk]);
      }
      else
      {
#pragma unroll
        for (int kk = 0; kk < S; kk++) acc += cq[kk] * TvT[r * S + kk];
      }
      tt[r * (S + 1) + li] = clip3(-(1 << 15), (1 << 15) - 1, (acc + 256) >> 9);
    }
  }
  RC_WAVE_SYNC();
  // stage I2 (horizontal): lane = row li
  const int s2i = (6 + 15 - 1) - bd + 2;
  {
    int y[S];
#pragma unroll
    for (int i = 0; i < S; i++) y[i] = tt[li * (S + 1) + i];
    pelS out;
#pragma unroll
    for (int x = 0; x < S; x++)
    {
      int acc = 0;
#pragma unroll
      for (int i = 0; i < S; i++) acc += __mul24(y[i], ThT[x * S + i]);
      const int resi = (short)clip3(-(1 << 15), (1 << 15) - 1, (acc + (1 << (s2i - 1))) >> s2i);
      out[x] = mode == RC_CHAIN ? (short)clip3(clpMin, clpMax, (int)p[x] + resi) : (short)resi;
    }
    if (act) *reinterpret_cast<pelS*>(recBase + d.rec_off + (size_t)li * d.rec_stride) = out;
  }
  RC_WAVE_SYNC();
}

// 8 x 4 and 4 x 8 (the most frequent rectangles of a real encode: tests/golden/trace_*.npz): the same scheme with 8 lanes per TU -- lane = row
// in the horizontal stages (H rows), lane = column in the vertical stages and the quantiser (W columns; the other lanes of the group idle there).
// The TU has two coefficient groups side by side (8 x 4) or one above the other (4 x 8): the group's scan index is its position.
template <int W, int H, int MODE>
__device__ __noinline__ void rc_rect_group(const RcDesc* __restrict__ descs, const int* __restrict__ list, int cnt, int item,
                                              const Pel* __restrict__ orgBase, const Pel* __restrict__ predBase, Pel* __restrict__ recBase,
                                              TCoeff* __restrict__ levelBase, unsigned* __restrict__ absSumOut, int bd, int clpMin, int clpMax,
                                              const RcSmallTab& tabs, int* tmpL, int lane)
{
  constexpr int mode = MODE;

  constexpr int L = 8, G = 64 / L, LW = W == 4 ? 2 : 3, LH = H == 4 ? 2 : 3, TOW = W == 4 ? 0 : 16, TOH = H == 4 ? 0 : 16;
  typedef short pelW __attribute__((ext_vector_type(W)));
  const int tg = lane / L, li = lane % L;
  const int k = item * G + tg;
  const bool act = k < cnt;
  const int ti = list[act ? k : 0];
  const RcDesc d = descs[ti];
  int* tt = tmpL + tg * (L * (L + 1));
  const int* Th = tabs.t[d.tr_hor] + TOW;
  const int* Tv = tabs.t[d.tr_ver] + TOH;
  const int* ThT = tabs.tt[d.tr_hor] + TOW;
  const int* TvT = tabs.tt[d.tr_ver] + TOH;
  const bool isRow = li < H, isCol = li < W;
  const int rr = isRow ? li : 0;
  // stage F1: lane = row
  pelW p;
#pragma unroll
  for (int jj = 0; jj < W; jj++) p[jj] = 0;
  const int s1 = LW + bd + 6 - 15 + 2, s2 = LH + 6 + 2;
  int cq[H];                                                              // the inverse stages' input: column li
  TCoeff* level = levelBase + d.level_off;
  if (mode != RC_INV)
  {
  const pelW o = *reinterpret_cast<const pelW*>(orgBase + d.org_off + (size_t)rr * d.org_stride);
  if (mode == RC_CHAIN) p = *reinterpret_cast<const pelW*>(predBase + d.pred_off + (size_t)rr * d.pred_stride);
  if (isRow)
  {
    int x[W];
#pragma unroll
    for (int j = 0; j < W; j++) x[j] = (int)o[j] - (int)p[j];
#pragma unroll
    for (int j = 0; j < W; j++)
    {
      int sum = 0;
#pragma unroll
      for (int kk = 0; kk < W; kk++) sum += __mul24(x[kk], Th[j * W + kk]);
      tt[j * (L + 1) + li] = (sum + (1 << (s1 - 1))) >> s1;
    }
  }
  RC_WAVE_SYNC();
  // stage F2: lane = column (horizontal frequency), registers = rows
  int cf[H];
  {
    int t[H];
#pragma unroll
    for (int r = 0; r < H; r++) t[r] = isCol ? tt[li * (L + 1) + r] : 0;
#pragma unroll
    for (int j = 0; j < H; j++)
    {
      int sum = 0;
#pragma unroll
      for (int r = 0; r < H; r++) sum += __mul24(t[r], Tv[j * H + r]);
      cf[j] = (sum + (1 << (s2 - 1))) >> s2;
    }
  }
  RC_WAVE_SYNC();
  if (mode == RC_FWD)                                                     // forward transform only: the coefficients are the result
  {
    if (act && isCol)
    {
#pragma unroll
      for (int j = 0; j < H; j++) level[j * W + li] = cf[j];
    }
    return;
  }
  // quantiser (coefficient groups: rows 4 R .. 4 R + 3 x the lane's aligned quad); idle lanes carry zeros
  const RcQ q = rc_qparams(W, H, d.qp, bd, d.intra_slice, d.sign_hiding);
  int lv[H], du[H], sum = 0;
#pragma unroll
  for (int j = 0; j < H; j++) { int mag; lv[j] = rc_quant_one(q, cf[j], du[j], mag); sum += mag; }
#pragma unroll
  for (int m = 1; m < L; m <<= 1) sum += __shfl_xor(sum, m);
  if (act && li == 0) absSumOut[ti] = (unsigned)sum;
  if (q.sbh)
  {
    int lastCg = -1;
#pragma unroll
    for (int R = 0; R < H / 4; R++)
    {
      int l4[4] = { lv[4 * R], lv[4 * R + 1], lv[4 * R + 2], lv[4 * R + 3] };
      if (rc_cg_nonzero(l4)) lastCg = max(lastCg, (W == 8 ? (li >> 2) : 0) + R);
    }
    lastCg = max(lastCg, __shfl_xor(lastCg, 4));
#pragma unroll
    for (int R = 0; R < H / 4; R++)
    {
      int l4[4] = { lv[4 * R], lv[4 * R + 1], lv[4 * R + 2], lv[4 * R + 3] };
      const int d4[4] = { du[4 * R], du[4 * R + 1], du[4 * R + 2], du[4 * R + 3] };
      const int c4[4] = { cf[4 * R], cf[4 * R + 1], cf[4 * R + 2], cf[4 * R + 3] };
      rc_sbh_quad(l4, d4, c4, (W == 8 ? (li >> 2) : 0) + R == lastCg, lane);
#pragma unroll
      for (int r = 0; r < 4; r++) lv[4 * R + r] = l4[r];
    }
  }
  if (act && isCol)
  {
#pragma unroll
    for (int j = 0; j < H; j++) level[j * W + li] = lv[j];
  }
#pragma unroll
  for (int j = 0; j < H; j++) cq[j] = rc_dequant_one(q, lv[j]);
  }
  else
  {
#pragma unroll
    for (int j = 0; j < H; j++) cq[j] = (act && isCol) ? level[j * W + li] : 0;   // any 32-bit coefficient: exact 32-bit multiplies below
  }
  // stage I1 (vertical), written transposed: tt[r][column]
  bool fits24 = true;
  if (mode != RC_CHAIN)
  {
#pragma unroll
    for (int kk = 0; kk < H; kk++) fits24 = fits24 && cq[kk] >= -(1 << 23) && cq[kk] < (1 << 23);
  }
  const bool narrow = mode == RC_CHAIN || __builtin_amdgcn_ballot_w64(!fits24) == 0ull;
  if (isCol)
  {
#pragma unroll
    for (int r = 0; r < H; r++)
    {
      int acc = 0;
      if (narrow)
      {
#pragma unroll
        for (int kk = 0; kk < H; kk++) acc += __mul24(cq[kk], TvT[r * H + kk]);
      }
      else
      {
#pragma unroll
        for (int kk = 0; kk < H; kk++) acc += cq[kk] * TvT[r * H + kk];
      }
      tt[r * (L + 1) + li] = clip3(-(1 << 15), (1 << 15) - 1, (acc + 256) >> 9);
    }
  }
  RC_WAVE_SYNC();
  // stage I2 (horizontal): lane = row
  const int s2i = (6 + 15 - 1) - bd + 2;
  if (isRow)
  {
    int y[W];
#pragma unroll
    for (int i = 0; i < W; i++) y[i] = tt[li * (L + 1) + i];
    pelW out;
#pragma unroll
    for (int x = 0; x < W; x++)
    {
      int acc = 0;
#pragma unroll
      for (int i = 0; i < W; i++) acc += __mul24(y[i], ThT[x * W + i]);
      const int resi = (short)clip3(-(1 << 15), (1 << 15) - 1, (acc + (1 << (s2i - 1))) >> s2i);
      out[x] = mode == RC_CHAIN ? (short)clip3(clpMin, clpMax, (int)p[x] + resi) : (short)resi;
    }
    if (act) *reinterpret_cast<pelW*>(recBase + d.rec_off + (size_t)li * d.rec_stride) = out;
  }
  RC_WAVE_SYNC();
}

// All five size classes in ONE launch.  Each class alone is bound by the latency of a TU, not by throughput (405 64x64 TUs are 405 waves:
// 20 us; 6480 16x16 TUs: 14 us; ...), and kernels on one stream run one after the other (82 us for the five launches at 4K).  Here a workgroup
// walks "slots" (four wave items of one class), longest classes first, so short items fill the machine while the long ones run.
constexpr int RC_NORD = 25, RC_NCOOP = 4;                    // the first RC_NCOOP entries: both sides >= 32, the workgroup's four waves per TU (rc_tu_coop)
__device__ __forceinline__ int rc_ord_g(int k)               // TUs per wave item: lane groups 64 / S, packed tiles: 16 / the short side
{
  return k == 6 || k == 15 || k == 16 ? 8 : k == 14 ? 16 : (k == 12 || k == 13 || k == 19 || k == 20 || k >= 23) ? 4 : k >= 10 ? 2 : 1;
}
// Class lists of a launch.  Classified (vvcgpu_resi_chain_batch): rc_classify_kernel's lists, class c at lists + c n, counts in hdr.  Runs
// (vvcgpu_resi_chain_runs_batch): the caller's descriptors are grouped by shape, so class c is the index range [base[c], base[c] + cnt[c]) -- `lists` is
// then an identity array (lib.hip: vvcgpu_iota) and every body reads its TU indices as before.
struct RcBins { int use; int base[RC_NCLS]; int cnt[RC_NCLS]; };
// class / range of a slot: lane j holds end[j]; the ends ascend, so the number of lanes whose end is at or below the slot is the class ordinal
__device__ __forceinline__ void rc_slot_class(const int* sCnt, const int* sEnd, int slot, int lane, int& k, int& start, int& cntK, int& endK)
{
  const int eL = sEnd[lane < RC_NORD ? lane : RC_NORD - 1], cL = sCnt[lane < RC_NORD ? lane : RC_NORD - 1];
  k = (int)__popcll(__builtin_amdgcn_ballot_w64(lane < RC_NORD - 1 && slot >= eL));
  start = k ? __builtin_amdgcn_readlane(eL, k - 1) : 0;
  cntK = __builtin_amdgcn_readlane(cL, k); endK = __builtin_amdgcn_readlane(eL, k);
}
template <int MODE>
__global__ __launch_bounds__(256, 3) void rc_chain_kernel(const Pel* __restrict__ orgBase, const Pel* __restrict__ predBase, Pel* __restrict__ recBase,
                                                          TCoeff* __restrict__ levelBase, const RcDesc* __restrict__ descs, int n,
                                                          const int* __restrict__ hdr, const int* __restrict__ lists, int* __restrict__ fbScratch,
                                                          unsigned* __restrict__ absSumOut, int bd, int clpMin, int clpMax,
                                                          const _Float16* __restrict__ image, VvcTrTables tb, RcBins bins, int* __restrict__ nextHdr)
{
  if (bins.use && blockIdx.x == 0 && threadIdx.x < VVC_CTR_INTS) nextHdr[threadIdx.x] = 0;   // (no classifier in front: this launch clears the counter set of the NEXT call, vvcgpu_counters)
  __shared__ __align__(16) _Float16 tab[RC_TAB_HALVES];
  __shared__ __align__(16) RcSmallTab tabs;
  __shared__ __align__(16) int tmpAll[4][8 * 8 * 9];          // per wave: transposes of the lane-group forms / the TU list of a packed tile; all of it: the limb planes of a co-operative TU
  __shared__ int red[4];                                      // co-operative TUs: abs sum, last coefficient group, range flag (rc_tu_coop)
  static_assert(sizeof(tmpAll) >= RC_EX_HALVES * sizeof(_Float16), "limb planes of rc_tu_coop");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform for the compiler too: list entries and descriptors of single-TU items arrive through the scalar cache
  int* const fbScr = fbScratch + ((size_t)blockIdx.x * 4 + wave) * 8192;     // this wave's two 4096-int buffers for a TU its body cannot take (rc_fallback_call)
  // slots (four wave items of one entry) in the order below: the longest items first, so that the short ones fill the machine while they run.
  // 8x8 / 8x4 / 4x8 / 4x4 stay with the lane groups: as packed tiles (exact as well) they are slower (8M samples: 8x8 0.164 vs 0.143 ms, 4x4 0.175
  // vs 0.142) -- a lane of a tile touches four 8-byte row pieces of its TU, a lane of a group one whole row
  constexpr int NORD = RC_NORD, NCOOP = RC_NCOOP;
  constexpr int ordCls[NORD] = { RC_C64, RC_R6432, RC_R3264, RC_C32, RC_R6416, RC_R1664, RC_C8, RC_R3216, RC_R1632, RC_C16,
                                 RC_P168, RC_P816, RC_P164, RC_P416, RC_C4, RC_R84, RC_R48, RC_P328, RC_P832, RC_P324, RC_P432,
                                 RC_P648, RC_P864, RC_P644, RC_P464 };
  constexpr int ordG[NORD] = { 1, 1, 1, 1, 1, 1, 8, 1, 1, 1, 2, 2, 4, 4, 16, 8, 8, 2, 2, 4, 4, 2, 2, 4, 4 };   // TUs per wave item (= rc_ord_g)
  // class counts and slot ranges live in LDS: as 75 scalars they were 143 spilled scalar registers around every body (round 6)
  __shared__ int sCnt[NORD], sEnd[NORD];
  __shared__ long long sOff[NORD];                            // where class k's TU indices start in `lists`
  // Prologue items: the table image is ~40 KB per workgroup, three workgroups per CU ask for it at the same time and a CU takes ~11 bytes per cycle in such a
  // burst -- 8.6 us of the launch with every wave waiting (a second copy of the image costs exactly that).  When the list holds enough 4x4 TUs, ONE wave of the
  // workgroup copies the f16 matrices while the other three each run an item of that class (sixteen TUs; it needs the small int32 tables only): the class's
  // LAST 3 x #workgroups items (twice as many when the class has them: the copy takes about as long as two items, 62.6 -> 60.6 us) are taken out of the
  // slot schedule for that; a list short of 4x4 TUs gives items of the 8x8 lane-group class (eight TUs, about two 4x4 items long) instead.
  constexpr int KPRE4 = 14, KPRE8 = 6;
  static_assert(ordCls[KPRE4] == RC_C4 && ordG[KPRE4] == 16 && ordCls[KPRE8] == RC_C8 && ordG[KPRE8] == 8, "the lane-group classes");
  const int G3 = 3 * (int)gridDim.x;
  const int c4 = bins.use ? bins.cnt[RC_C4] : hdr[RC_C4], items4 = (c4 + 15) >> 4, c8 = bins.use ? bins.cnt[RC_C8] : hdr[RC_C8], items8 = (c8 + 7) >> 3;
  const int kPre = items4 >= G3 ? KPRE4 : items8 >= G3 ? KPRE8 : -1;
  const bool w3 = kPre == KPRE4 && items4 >= 7 * (int)gridDim.x;                  // the copying wave takes a 4x4 item behind its copy as well (60.1 -> 58.4 us)
  const int perWave = kPre == KPRE4 && items4 >= 2 * G3 ? 2 : 1, nPre = perWave * G3 + (w3 ? (int)gridDim.x : 0);
  int totalB = 0;                                             // slots of the schedule without the prologue items
#pragma unroll
  for (int k = 0; k < NORD; k++)
  {
    const int ck = bins.use ? bins.cnt[ordCls[k]] : hdr[ordCls[k]], ik = (ck + ordG[k] - 1) / ordG[k];
    totalB += k < NCOOP ? ik : ((k == kPre ? ik - nPre : ik) + 3) >> 2;     // co-operative classes: one TU per slot (the four waves together)
  }
  const bool usePre = kPre >= 0 && totalB >= (int)gridDim.x;
  const int cPre = kPre == KPRE4 ? c4 : c8, itemsPre = kPre == KPRE4 ? items4 : items8;
  int total = 0;
#pragma unroll
  for (int k = 0; k < NORD; k++)
  {
    int ck = bins.use ? bins.cnt[ordCls[k]] : hdr[ordCls[k]];
    if (k == kPre && usePre) ck = (itemsPre - nPre) * ordG[k];   // (whole items: the class's partial last item is a prologue item)
    const int ik = (ck + ordG[k] - 1) / ordG[k];
    total += k < NCOOP ? ik : (ik + 3) >> 2;
    if (tid == 0) { sCnt[k] = ck; sEnd[k] = total; sOff[k] = bins.use ? (long long)bins.base[ordCls[k]] : (long long)ordCls[k] * n; }
  }
  if ((int)blockIdx.x >= total) return;
#ifdef RC_DIAG
  unsigned long long dg[24]; int dgk[12]; int dgn = 0;
  const bool dgOn = (blockIdx.x == 37 || blockIdx.x == 500 || blockIdx.x == 767) && tid == 0;
  if (dgOn) dg[0] = __builtin_amdgcn_s_memtime();
#endif
  if (tid == 0) { red[0] = 0; red[1] = -1; red[2] = 0; }
  // Runs mode: nothing has read the descriptors yet (the classifier of the other mode leaves them in the L2 on its way: without it the chain kernel was
  // 72.3 instead of 67.6 us) -- a wave touches the descriptor lines of its items of the workgroup's first slots now, behind the table copy they arrive.
  int pfD[2] = { 0, 0 };
  if (bins.use)
  {
    __syncthreads();                                          // sEnd / sCnt / sOff
#pragma unroll
    for (int r = 0; r < 2; r++)
    {
      const int pair = lane + 64 * r, j = pair >> 3, line = pair & 7;          // (slot j of this workgroup, 128-byte line of the item's descriptors)
      const int slot = (int)blockIdx.x + j * (int)gridDim.x;
      if (slot < total)
      {
        int k = 0, start = 0;
        for (int q = 0; q < NORD - 1; q++) { const int e = sEnd[q]; if (slot >= e) { k = q + 1; start = e; } }
        const int gK = rc_ord_g(k), cntK = sCnt[k];
        const int first = k < NCOOP ? slot - start : ((slot - start) * 4 + wave) * gK, tus = k < NCOOP ? 1 : gK;
        const int t = first + 2 * line;
        if (2 * line < tus && t < cntK) pfD[r] = *reinterpret_cast<const int*>(descs + (int)sOff[k] + t);
      }
    }
  }
  if (usePre)
  {
    const uint4* src = reinterpret_cast<const uint4*>(image);
    if (tid < (int)sizeof(RcSmallTab) / 16) reinterpret_cast<uint4*>(&tabs)[tid] = src[RC_TAB_HALVES / 8 + tid];
    __syncthreads();
    if (wave == 3)
    {
      // memory -> LDS without registers in between (1 KB per instruction, the image is copied as it lies: lane-linear), all 39 in flight at once: through
      // registers the one wave had three rounds of 13 loads, each a trip through the burst (25 - 29 k cycles, the items of the others 17 k)
      constexpr int NW4 = RC_TAB_HALVES / 8;
      typedef __attribute__((address_space(1))) const void* GPtr;
      typedef __attribute__((address_space(3))) void* LPtr;
#pragma unroll
      for (int u = 0; u < (NW4 + 63) / 64; u++)
        if (lane + 64 * u < NW4)
          __builtin_amdgcn_global_load_lds((GPtr)(src + lane + 64 * u), (LPtr)(reinterpret_cast<uint4*>(tab) + 64 * u), 16, 0, 0);
      if (w3)
      {
        const int* const list4 = lists + (bins.use ? (long long)bins.base[RC_C4] : (long long)RC_C4 * n);
        rc_small_group<4, MODE>(descs, list4, cPre, itemsPre - (int)gridDim.x + (int)blockIdx.x, orgBase, predBase, recBase, levelBase, absSumOut, bd, clpMin, clpMax, tabs, tmpAll[wave], lane);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    else
    {
      const int clsPre = kPre == KPRE4 ? (int)RC_C4 : (int)RC_C8;
      const int* const listPre = lists + (bins.use ? (long long)bins.base[clsPre] : (long long)clsPre * n);
      const int first = itemsPre - nPre + ((int)blockIdx.x * 3 + wave) * perWave;
      if (kPre == KPRE8)
        rc_small_group<8, MODE>(descs, listPre, cPre, first, orgBase, predBase, recBase, levelBase, absSumOut, bd, clpMin, clpMax, tabs, tmpAll[wave], lane);
      else for (int r = 0; r < perWave; r++)
        rc_small_group<4, MODE>(descs, listPre, cPre, first + r, orgBase, predBase, recBase, levelBase, absSumOut, bd, clpMin, clpMax, tabs, tmpAll[wave], lane);
    }
  }
  else
  {
    // the table image (f16 matrices + the int32 4- / 8-point matrices) as ONE run of 16-byte loads, all in flight before the first store.  In-kernel stamps
    // (RC_DIAG): 16 - 24 k cycles until the barrier below with three workgroups per CU -- every workgroup of an XCD asks its L2 for the same 41 KB at the
    // same time; starting at staggered offsets was slower (70.6 vs 67.3 us), the phase-by-phase copy of round 5 the same (67.9 vs 68.5)
    constexpr int PER = (RC_IMG_U4 + 255) / 256;
    uint4 v[PER];
    const uint4* src = reinterpret_cast<const uint4*>(image);
#pragma unroll
    for (int u = 0; u < PER; u++) if (tid + 256 * u < RC_IMG_U4) v[u] = src[tid + 256 * u];
#pragma unroll
    for (int u = 0; u < PER; u++)
    {
      const int i = tid + 256 * u;
      if (i < RC_TAB_HALVES / 8) reinterpret_cast<uint4*>(tab)[i] = v[u];
      else if (i < RC_IMG_U4) reinterpret_cast<uint4*>(&tabs)[i - RC_TAB_HALVES / 8] = v[u];
    }
  }
  __syncthreads();
#ifdef RC_DIAG
  if (dgOn) dg[1] = __builtin_amdgcn_s_memtime();
#endif
  for (int slot = blockIdx.x; slot < total; slot += gridDim.x)
  {
#ifdef RC_DIAG
    if (dgOn && dgn < 11) { dg[2 + dgn] = __builtin_amdgcn_s_memtime(); }
#endif
    int k, start, cntK, endK;
    rc_slot_class(sCnt, sEnd, slot, lane, k, start, cntK, endK);
    const int gK = rc_ord_g(k);
    const long long offV = sOff[k];
    const int* const listK = lists + (((long long)__builtin_amdgcn_readfirstlane((int)(offV >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)offV));   // this slot's class list
    const int itemsK = (cntK + gK - 1) / gK;
    const int item = k < NCOOP ? slot - start : (slot - start) * 4 + wave;
    bool done = true;
    int ti = 0;
#define RC_MF(K, W_, H_)                                                                                                                      \
    case K: if (item < cntK) { ti = __builtin_amdgcn_readfirstlane(listK[item]);                                    \
        done = rc_tu_mfma_call<W_, H_, MODE>(descs, ti, orgBase, predBase, recBase, levelBase, absSumOut, bd, clpMin, clpMax, tab, tb.dqInv, tb.scanOff, lane); } break;
#define RC_PK(K, W_, H_)                                                                                                                      \
    case K: if (item < itemsK) rc_tile_packed<W_, H_, MODE>(descs, listK, cntK, item, orgBase, predBase, recBase, levelBase, absSumOut, bd, clpMin, \
                                                        clpMax, tab, tb.dqInv, tb.scanOff, tb.tr32, fbScr, tmpAll[wave], lane); break;
#define RC_CO(K, W_, H_)                                                                                                                      \
    case K: { ti = __builtin_amdgcn_readfirstlane(listK[item]);                                                       \
        done = rc_tu_coop<W_, H_, MODE>(descs[ti], orgBase, predBase, recBase, levelBase, absSumOut, ti, bd, clpMin, clpMax, tab, tb.dqInv, tb.scanOff, \
                                        reinterpret_cast<_Float16*>(&tmpAll[0][0]), red, wave, lane);                                         \
        if (wave != 0) done = true; } break;                 /* (one fall-back entry per TU) */
    switch (k)
    {
    RC_CO(0, 64, 64) RC_CO(1, 64, 32) RC_CO(2, 32, 64) RC_CO(3, 32, 32) RC_MF(4, 64, 16) RC_MF(5, 16, 64) RC_MF(7, 32, 16) RC_MF(8, 16, 32)
    case 9:
    {
      // 16x16, the most numerous matrix-core class: the workgroup's consecutive slots of this class are walked HERE, with the list entry of the
      // item after next and the descriptor of the next item requested (vector loads: no scalar load in flight during the stages) while the current
      // TU runs -- list entry -> descriptor -> samples are three memory latencies in a row otherwise.  The registers of the pipeline live only
      // inside this loop (across the 64-point bodies they were spills: docs/OPTIMISATION_LOG.md).
      int vz = 0;
      asm volatile("" : "+v"(vz));                           // a zero the compiler takes for lane-dependent: the loads below stay vector loads
      const int* lst = listK;
      const int G4 = 4 * (int)gridDim.x;
      int it = item;
      if (it >= cntK) break;
      int nAside = 0;                                        // TUs outside the matrix-core range (at most one per slot of the walk: far below the 576 ints of the wave's scratch)
      int ti0 = __builtin_amdgcn_readfirstlane(lst[it]);
      int tiv1 = it + G4 < cntK ? lst[it + G4 + vz] : 0;
      uint4 dq = reinterpret_cast<const uint4*>(descs + ti0)[(lane & 3) + vz];
      for (;;)
      {
        RcDesc dCur;
        {
          unsigned* w = reinterpret_cast<unsigned*>(&dCur);
#pragma unroll
          for (int q = 0; q < 4; q++)
          {
            w[4 * q + 0] = __builtin_amdgcn_readlane(dq.x, q); w[4 * q + 1] = __builtin_amdgcn_readlane(dq.y, q);
            w[4 * q + 2] = __builtin_amdgcn_readlane(dq.z, q); w[4 * q + 3] = __builtin_amdgcn_readlane(dq.w, q);
          }
        }
        const bool more = it + G4 < cntK && slot + (int)gridDim.x < endK;          // the next slot of this workgroup is of this class too
        const int ti1 = __builtin_amdgcn_readfirstlane(tiv1);
        if (more)
        {
          dq = reinterpret_cast<const uint4*>(descs + ti1)[(lane & 3) + vz];
          tiv1 = it + 2 * G4 < cntK ? lst[it + 2 * G4 + vz] : 0;
        }
        const bool ok = rc_tu_mfma<16, 16, MODE>(dCur, orgBase, predBase, recBase, levelBase, absSumOut, ti0, bd, clpMin, clpMax, tab, tb.dqInv, tb.scanOff, lane);
        if (!ok) { if (lane == 0) tmpAll[wave][nAside] = ti0; nAside++; }     // served behind the walk: a call inside this loop keeps the loop's registers in scratch memory
        if (!more) break;
        it += G4; ti0 = ti1; slot += (int)gridDim.x;
      }
      RC_WAVE_SYNC();
      for (int i = 0; i < nAside; i++)
        rc_fallback_call<MODE>(descs, __builtin_amdgcn_readfirstlane(tmpAll[wave][i]), orgBase, predBase, recBase, levelBase, absSumOut, bd, clpMin, clpMax, tb.tr32, tb.dqInv, tb.scanOff, fbScr, lane);
      break;
    }
    RC_PK(10, 16, 8) RC_PK(11, 8, 16) RC_PK(12, 16, 4) RC_PK(13, 4, 16)
    case 6: if (item < itemsK) rc_small_group<8, MODE>(descs, listK, cntK, item, orgBase, predBase, recBase, levelBase, absSumOut, bd, clpMin, clpMax, tabs, tmpAll[wave], lane); break;
    case 14: if (item < itemsK) rc_small_group<4, MODE>(descs, listK, cntK, item, orgBase, predBase, recBase, levelBase, absSumOut, bd, clpMin, clpMax, tabs, tmpAll[wave], lane); break;
    case 15: if (item < itemsK) rc_rect_group<8, 4, MODE>(descs, listK, cntK, item, orgBase, predBase, recBase, levelBase, absSumOut, bd, clpMin, clpMax, tabs, tmpAll[wave], lane); break;
    case 16: if (item < itemsK) rc_rect_group<4, 8, MODE>(descs, listK, cntK, item, orgBase, predBase, recBase, levelBase, absSumOut, bd, clpMin, clpMax, tabs, tmpAll[wave], lane); break;
#define RC_PK32(K, F)                                                                                                                         \
    case K: if (item < itemsK) F(descs, listK, cntK, item, orgBase, predBase, recBase, levelBase, absSumOut, bd, clpMin, clpMax, tab,     \
                                   tb.dqInv, tb.scanOff, tb.tr32, fbScr, tmpAll[wave], lane); break;
    RC_PK32(17, (rc_tile_packed_wl<32, 8, MODE>)) RC_PK32(18, (rc_tile_packed_hl<8, 32, MODE>)) RC_PK32(19, (rc_tile_packed_wl<32, 4, MODE>)) RC_PK32(20, (rc_tile_packed_hl<4, 32, MODE>))
    RC_PK32(21, (rc_tile_packed_wl<64, 8, MODE>)) RC_PK32(22, (rc_tile_packed_hl<8, 64, MODE>)) RC_PK32(23, (rc_tile_packed_wl<64, 4, MODE>))
    default: if (item < itemsK) rc_tile_packed_hl<4, 64, MODE>(descs, listK, cntK, item, orgBase, predBase, recBase, levelBase, absSumOut, bd, clpMin, clpMax, tab,
                                                            tb.dqInv, tb.scanOff, tb.tr32, fbScr, tmpAll[wave], lane); break;
#undef RC_PK32
    }
#undef RC_MF
#undef RC_CO
#undef RC_PK
    if (!done) rc_fallback_call<MODE>(descs, ti, orgBase, predBase, recBase, levelBase, absSumOut, bd, clpMin, clpMax, tb.tr32, tb.dqInv, tb.scanOff, fbScr, lane);   // residual outside +-1023 (co-operative TU: wave 0, behind the TU's last barrier)
    asm volatile("" :: "v"(pfD[0]), "v"(pfD[1]));            // (the touched descriptor words: their registers stay theirs until the loads have landed)
#ifdef RC_DIAG
    if (dgOn && dgn < 11) { dgk[dgn] = k; dgn++; }
#endif
  }
#ifdef RC_DIAG
  if (dgOn)
  {
    dg[2 + dgn] = __builtin_amdgcn_s_memtime();
    printf("[rc diag wg %d wave %d] start->tables %llu cycles; slots:", (int)blockIdx.x, tid >> 6, dg[1] - dg[0]);
    for (int i = 0; i < dgn; i++) printf(" k%d:%llu", dgk[i], dg[3 + i] - dg[2 + i]);
    printf(" | total %llu\n", dg[2 + dgn] - dg[0]);
  }
#endif
}

}  // namespace

// f16 LDS image of the matrices (mfma_tr.h): built once per device
const _Float16* vvcgpu_mfma_image(const VvcTrTables& tb)
{
  static std::mutex imageMutex;
  static _Float16* images[64] = { nullptr };
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { vvcgpu_set_error("mfma image: device index"); return nullptr; }
  std::lock_guard<std::mutex> lock(imageMutex);
  if (!images[dev])
  {
    // built on the null stream with blocking calls (not on the caller's stream: that would serialise every other thread's first call behind
    // a stream of unknown length); the buffer is released again if any step fails
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, (size_t)RC_IMG_U4 * 16);
    if (e != hipSuccess) { vvcgpu_set_error("mfma image: hipMalloc failed: %s", hipGetErrorString(e)); return nullptr; }
    e = hipMemset(p, 0, (size_t)RC_IMG_U4 * 16);                    // row padding
    if (e == hipSuccess)
    {
      hipLaunchKernelGGL(rc_build_tables_kernel, dim3(16), dim3(256), 0, (hipStream_t)0, static_cast<_Float16*>(p), tb.tr32, tb.tr32t);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();     // other streams may use the image right after this call returns
    if (e != hipSuccess)
    {
      (void)hipFree(p);
      vvcgpu_set_error("building the f16 table image failed: %s", hipGetErrorString(e));
      return nullptr;
    }
    images[dev] = static_cast<_Float16*>(p);
  }
  return images[dev];
}

__attribute__((visibility("hidden"))) int vvcgpu_tr_image_build(void)
{
  VvcTrTables tb;
  const int rt = vvcgpu_tr_tables(&tb);
  if (rt) return rt;
  return vvcgpu_mfma_image(tb) ? VVCGPU_OK : VVCGPU_E_DEVICE;
}

// the classify / chain / generic launches behind vvcgpu_resi_chain_batch (mode RC_CHAIN) and, for long calls, behind vvcgpu_tr_fwd_batch /
// vvcgpu_tr_inv_batch (RC_FWD / RC_INV; transform.hip): ONE chain launch with packed tiles instead of the small / matrix-core / dot2 kernels in a row
static int rc_chain_launch(int mode, const vvc_pel* org_base, const vvc_pel* pred_base, vvc_pel* rec_base, vvc_coef* level_base, const void* descs_raw, int n,
                           int bit_depth, int clp_min, int clp_max, uint32_t* abs_sum, hipStream_t st, const RcBins* runs = nullptr)
{
  VvcTrTables tb;
  const int rt = vvcgpu_tr_tables(&tb);
  if (rt) return rt;
  const _Float16* image = vvcgpu_mfma_image(tb);
  if (!image) return VVCGPU_E_DEVICE;
  // scratch: the class lists, (plain transforms) the descriptors as chain descriptors, and per wave of the chain launch two 4096-int buffers for a TU its
  // body cannot take (rc_fallback_call: touched only then)
  constexpr int CHAIN_WGS = 768;
  const size_t ints = (size_t)RC_NCLS * n;
  const size_t convOff = (ints * sizeof(int) + 63) & ~(size_t)63;
  const size_t fbOff = (convOff + (mode == RC_CHAIN ? 0 : (size_t)n * sizeof(RcDesc)) + 63) & ~(size_t)63;
  int* ws = static_cast<int*>(vvcgpu_scratch(st, fbOff + (size_t)CHAIN_WGS * 4 * 8192 * sizeof(int)));
  if (!ws) return VVCGPU_E_DEVICE;
  RcDesc* conv = mode == RC_CHAIN ? nullptr : reinterpret_cast<RcDesc*>(reinterpret_cast<unsigned char*>(ws) + convOff);
  int* fbScratch = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(ws) + fbOff);
  const RcDesc* descs = mode == RC_CHAIN ? static_cast<const RcDesc*>(descs_raw) : conv;
  // the header lives in the stream's persistent zeroed counters: this call's set is clean, the classifier clears the other set for the next
  // call (no fill launch in front of the chain)
  int cur = 0;
  int* counters = vvcgpu_counters(st, &cur);
  if (!counters) return VVCGPU_E_DEVICE;
  int* hdr = counters + VVC_CTR_INTS * cur;
  int* lists = ws;
  RcBins bins;
  memset(&bins, 0, sizeof bins);
  if (runs)
  {
    // the caller's descriptors are grouped by shape (vvcgpu_resi_chain_runs_batch): no classifier launch, the class lists are ranges of an identity array
    bins = *runs;
    bins.use = 1;
    lists = vvcgpu_iota(st, n);
    if (!lists) return VVCGPU_E_DEVICE;
  }
  const dim3 cg(n < 1024 * RC_CLS_WGS ? cdiv(n, 1024) : RC_CLS_WGS);
  if (runs) { }
  else if (mode == RC_CHAIN)
    hipLaunchKernelGGL(rc_classify_kernel<false>, cg, dim3(1024), 0, st, descs_raw, n, hdr, lists, abs_sum, counters + VVC_CTR_INTS * (cur ^ 1), true, conv);
  else
    hipLaunchKernelGGL(rc_classify_kernel<true>, cg, dim3(1024), 0, st, descs_raw, n, hdr, lists, abs_sum, counters + VVC_CTR_INTS * (cur ^ 1), true, conv);
  VVC_LAUNCH_CHECK_COUNTERS(st);
  // (measured: forking the size classes onto library-owned side streams and joining them with events is SLOWER than launching them back to
  // back on the caller's stream, 0.158 vs 0.115 ms at 4K -- a cross-stream event costs more than these 20 us kernels gain)
  {
#define RC_CHAIN_LAUNCH(M) hipLaunchKernelGGL(rc_chain_kernel<M>, dim3(CHAIN_WGS), dim3(256), 0, st, org_base, pred_base, rec_base, level_base, descs, n, hdr, lists, fbScratch, \
                                             abs_sum, bit_depth, clp_min, clp_max, image, tb, bins, counters + VVC_CTR_INTS * (cur ^ 1))
    if (mode == RC_CHAIN) RC_CHAIN_LAUNCH(RC_CHAIN); else if (mode == RC_FWD) RC_CHAIN_LAUNCH(RC_FWD); else RC_CHAIN_LAUNCH(RC_INV);
#undef RC_CHAIN_LAUNCH
    VVC_LAUNCH_CHECK_COUNTERS(st);
  }
  // the class-`generic` list (2-wide TUs); runs mode: such shapes take the classified path, so there is nothing to launch
  if (!runs)
  {
    const int wgS = cdiv(n, 4) < 256 ? cdiv(n, 4) : 256;
#define RC_GEN_LAUNCH(M) hipLaunchKernelGGL(rc_generic_kernel<M>, dim3(wgS), dim3(256), 0, st, org_base, pred_base, rec_base, level_base, descs, hdr + RC_CGEN, \
                                           lists + (size_t)RC_CGEN * n, abs_sum, bit_depth, clp_min, clp_max, tb)
    if (mode == RC_CHAIN) RC_GEN_LAUNCH(RC_CHAIN); else if (mode == RC_FWD) RC_GEN_LAUNCH(RC_FWD); else RC_GEN_LAUNCH(RC_INV);
#undef RC_GEN_LAUNCH
    VVC_LAUNCH_CHECK_COUNTERS(st);
  }
  return VVCGPU_OK;
}

// the plain transform entries of transform.hip, long calls (mode 1: forward, 2: inverse)
__attribute__((visibility("hidden"))) int vvcgpu_tr_chain_launch(int mode, const vvc_pel* resi_in, vvc_pel* resi_out, vvc_coef* coeff, const vvcgpu_tr_desc* descs, int n,
                                                                   int bit_depth, void* stream)
{
  return rc_chain_launch(mode, resi_in, resi_in, resi_out, coeff, descs, n, bit_depth, 0, 0, nullptr, (hipStream_t)stream);
}

extern "C" {

int vvcgpu_resi_chain_batch(const vvc_pel* org_base, const vvc_pel* pred_base, vvc_pel* rec_base, vvc_coef* level_base,
                            const vvcgpu_resi_chain_desc* descs, int n, int bit_depth, int clp_min, int clp_max, uint32_t* abs_sum, void* stream)
{
  VVC_CHECK_ARG(n >= 0, "resi_chain_batch: n %d", n);
  if (n == 0) return VVCGPU_OK;
  VVC_CHECK_ARG(org_base && pred_base && rec_base && level_base && descs && abs_sum, "resi_chain_batch: null pointer");
  VVC_CHECK_ARG(bit_depth >= 8 && bit_depth <= 10, "resi_chain_batch: bit depth %d outside 8..10", bit_depth);
  VVC_CHECK_ARG(clp_min <= clp_max, "resi_chain_batch: clipping range");
  return rc_chain_launch(RC_CHAIN, org_base, pred_base, rec_base, level_base, descs, n, bit_depth, clp_min, clp_max, abs_sum, (hipStream_t)stream);
}

int vvcgpu_resi_chain_runs_batch(const vvc_pel* org_base, const vvc_pel* pred_base, vvc_pel* rec_base, vvc_coef* level_base,
                                 const vvcgpu_resi_chain_desc* descs, int n, const int32_t* runs_host, int n_runs,
                                 int bit_depth, int clp_min, int clp_max, uint32_t* abs_sum, void* stream)
{
  VVC_CHECK_ARG(n >= 0 && n_runs >= 0, "resi_chain_runs_batch: n %d, n_runs %d", n, n_runs);
  if (n == 0) return VVCGPU_OK;
  VVC_CHECK_ARG(org_base && pred_base && rec_base && level_base && descs && abs_sum && runs_host, "resi_chain_runs_batch: null pointer");
  VVC_CHECK_ARG(bit_depth >= 8 && bit_depth <= 10, "resi_chain_runs_batch: bit depth %d outside 8..10", bit_depth);
  VVC_CHECK_ARG(clp_min <= clp_max, "resi_chain_runs_batch: clipping range");
  RcBins bins;
  memset(&bins, 0, sizeof bins);
  long long at = 0;
  bool classified = false;                                   // a shape outside the chain's own bodies: the classified path takes the whole call
  for (int r = 0; r < n_runs; r++)
  {
    const int w = runs_host[3 * r], h = runs_host[3 * r + 1], cnt = runs_host[3 * r + 2];
    VVC_CHECK_ARG(cnt >= 0 && w >= 2 && w <= 64 && h >= 2 && h <= 64 && (w & (w - 1)) == 0 && (h & (h - 1)) == 0, "resi_chain_runs_batch: run %d (%d x %d, %d TUs)", r, w, h, cnt);
    if (cnt == 0) continue;
    RcDesc d;
    d.w = (short)w; d.h = (short)h;
    const int cls = rc_class(d, true);
    if (cls == RC_CGEN) classified = true;
    else
    {
      VVC_CHECK_ARG(bins.cnt[cls] == 0, "resi_chain_runs_batch: shape %d x %d appears in two runs", w, h);
      bins.base[cls] = (int)at; bins.cnt[cls] = cnt;
    }
    at += cnt;
  }
  VVC_CHECK_ARG(at == n, "resi_chain_runs_batch: the runs hold %lld TUs, n = %d", at, n);
  return rc_chain_launch(RC_CHAIN, org_base, pred_base, rec_base, level_base, descs, n, bit_depth, clp_min, clp_max, abs_sum, (hipStream_t)stream, classified ? nullptr : &bins);
}

}  // extern "C"
