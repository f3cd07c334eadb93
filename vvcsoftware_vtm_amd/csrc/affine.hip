// affine.hip -- affine gradient search kernels (next row N3) for gfx950.
//
// Reference behaviour reproduced (bit-exact): AffineGradientSearch::xHorizontalSobelFilter / xVerticalSobelFilter /
// xEqualCoeffComputer (CommonLib/AffineGradientSearch.cpp:66-174; SIMD twins x86/AffineGradientSearchX86.h:72-312), the three
// table slots of AffineGradientSearch.h:50-54.
//
// Design: one wavefront per PU.  Sobel: every output sample is the 3x3 response at the nearest INTERIOR position (that is what
// the reference's ring-copy rules amount to), so the ring needs no second pass.  Equal coefficients: every lane walks its
// share of the block with the <= 6 x 7 sums in 64-bit registers, then one transposed wave reduction (63 shuffles for all sums).
#include "common.h"
#include "dist_dev.h"

namespace {

__global__ __launch_bounds__(256) void affine_sobel_kernel(int vertical, const Pel* __restrict__ predBase, int* __restrict__ derivBase,
                                                           const vvcgpu_afg_desc* __restrict__ descs, int n)
{
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);      // one wavefront per PU
  if (b >= n) return;
  const vvcgpu_afg_desc d = descs[b];
  const Pel* pred = predBase + d.pred_off;
  int* deriv = derivBase + d.deriv_off;
  const int w = d.w, h = d.h, ps = d.pred_stride;
  for (int i = threadIdx.x & 63; i < w * h; i += 64)
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
  for (int i = tid; i < w * h; i += 64)
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

// Sum M = 64 (or 32) per-lane values over the wavefront with a halving butterfly: at every step a lane keeps one half of its
// values and hands the other half to its partner, so 63 (31 + 1) shuffles replace 6 per value; lane L ends with the total of value L.
template <int M>
__device__ __forceinline__ long long wave_transpose_sum(long long (&v)[M], int lane)
{
  static_assert(M == 64 || M == 32, "M");
#pragma unroll
  for (int s = M / 2, len = M; s > 0; s >>= 1, len >>= 1)
  {
    const bool up = (lane & s) != 0;
#pragma unroll
    for (int i = 0; i < len / 2; i++)
    {
      const long long keep = up ? v[i + len / 2] : v[i], send = up ? v[i] : v[i + len / 2];
      v[i] = keep + __shfl_xor(send, s);
    }
  }
  if (M == 32) v[0] += __shfl_xor(v[0], 32);
  return v[0];
}

// one wavefront per PU: every lane accumulates its samples, one transposed reduction, lanes 0 .. P (P + 1) - 1 hold the sums
template <int P>
__device__ __forceinline__ void eq_block(const vvcgpu_afe_desc& d, const Pel* resiBase, const int* gxBase, const int* gyBase, long long* out, int lane)
{
  long long acc[P][P + 1];
#pragma unroll
  for (int c = 0; c < P; c++)
#pragma unroll
    for (int r = 0; r <= P; r++) acc[c][r] = 0;
  eq_accumulate<P>(resiBase + d.resi_off, gxBase + d.deriv_off, gyBase + d.deriv_off, d.deriv_stride, d.w, d.h, lane, acc);
  constexpr int M = P == 6 ? 64 : 32;
  long long v[M];
#pragma unroll
  for (int i = 0; i < M; i++) v[i] = i < P * (P + 1) ? acc[i / (P + 1)][i % (P + 1)] : 0;
  const long long total = wave_transpose_sum<M>(v, lane);       // lane L: sum of value L = acc[L / (P + 1)][L % (P + 1)]
  // out[row7][col7]; rows 1..P hold the equations, everything else is zero
  if (lane < 49)
  {
    const int row7 = lane / 7, col7 = lane - row7 * 7;
    const bool used = row7 >= 1 && row7 <= P && col7 <= P;
    const int src = used ? (row7 - 1) * (P + 1) + col7 : 0;
    long long val = __shfl(total, src);
    out[lane] = used ? val : 0;
  }
  else (void)__shfl(total, 0);
}

__global__ __launch_bounds__(256) void affine_equal_coeff_kernel(const Pel* __restrict__ resiBase, const int* __restrict__ gxBase,
                                                                 const int* __restrict__ gyBase, const vvcgpu_afe_desc* __restrict__ descs, int n,
                                                                 long long* __restrict__ out)
{
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= n) return;
  const vvcgpu_afe_desc d = descs[b];
  long long* o = out + (size_t)b * 49;
  if (d.six_param) eq_block<6>(d, resiBase, gxBase, gyBase, o, lane);
  else             eq_block<4>(d, resiBase, gxBase, gyBase, o, lane);
}

// ---- affine sub-block motion vectors: the derivation loop of InterPrediction::xPredAffineBlk (InterPrediction.cpp:618-701) per sub-block, both lists
__global__ __launch_bounds__(256) void affine_subblock_descs_kernel(const vvcgpu_affine_pu* __restrict__ pus, int puBytes, int n, int comp, int picW, int picH,
                                                                    int maxCuW, int maxCuH, int orgX, int orgY, int rs0, int rs1,
                                                                    vvcgpu_mc_desc* __restrict__ out)
{
  const int pi = blockIdx.x;
  if (pi >= n) return;
  const vvcgpu_affine_pu pu = *reinterpret_cast<const vvcgpu_affine_pu*>(reinterpret_cast<const char*>(pus) + (size_t)pi * puBytes);   // puBytes: array stride
  const int sc = comp ? 1 : 0;                                          // 4:2:0 component scale
  const int bw = 4 >> sc, bh = 4 >> sc;                                  // AFFINE_MIN_BLOCK_SIZE, scaled (:585-586, :619-620)
  const int cxW = pu.w >> sc, cxH = pu.h >> sc;
  const int nbx = cxW / bw, nby = cxH / bh;
  const int iBit = 7, shift = iBit - 4 + 2 + 2;                          // MAX_CU_DEPTH; :658
  const int lgW = 31 - __clz(cxW), lgH = 31 - __clz(cxH);
  const int horMax = (picW + 8 - pu.pos_x - 1) << 4, horMin = (-maxCuW - 8 - pu.pos_x + 1) << 4;
  const int verMax = (picH + 8 - pu.pos_y - 1) << 4, verMin = (-maxCuH - 8 - pu.pos_y + 1) << 4;
  for (int sb = threadIdx.x; sb < nbx * nby; sb += blockDim.x)
  {
    const int bxI = sb % nbx, byI = sb / nbx, wq = bxI * bw, hq = byI * bh;
    vvcgpu_mc_desc d;
    d.w = (int16_t)bw; d.h = (int16_t)bh; d.is_luma = comp ? 0 : 1; d.bi = pu.bi ? 1 : 0; d.reserved = 0;
    d.dst_off = pu.dst_off + (int64_t)hq * pu.dst_stride + wq; d.dst_stride = pu.dst_stride;
    d.ref0_stride = rs0; d.ref1_stride = rs1; d.ref1_off = 0; d.frac_x1 = 0; d.frac_y1 = 0;
#pragma unroll
    for (int l = 0; l < 2; l++)
    {
      if (l == 1 && !pu.bi) break;
      const int ltx = pu.mv[l][0][0], lty = pu.mv[l][0][1];
      const int dHorX = (pu.mv[l][1][0] - ltx) << (iBit - lgW), dHorY = (pu.mv[l][1][1] - lty) << (iBit - lgW);
      int dVerX, dVerY;
      if (pu.six_param) { dVerX = (pu.mv[l][2][0] - ltx) << (iBit - lgH); dVerY = (pu.mv[l][2][1] - lty) << (iBit - lgH); }
      else { dVerX = -dHorY; dVerY = dHorX; }
      int mh = (ltx << iBit) + dHorX * ((bw >> 1) + wq) + dVerX * ((bh >> 1) + hq);
      int mvv = (lty << iBit) + dHorY * ((bw >> 1) + wq) + dVerY * ((bh >> 1) + hq);
      const int off = 1 << (shift - 1);                                  // roundAffineMv
      mh = mh >= 0 ? (mh + off) >> shift : -((-mh + off) >> shift);
      mvv = mvv >= 0 ? (mvv + off) >> shift : -((-mvv + off) >> shift);
      mh = min(horMax, max(horMin, mh));
      mvv = min(verMax, max(verMin, mvv));
      const int xInt = mh >> (4 + sc), xFrac = mh & (sc ? 31 : 15), yInt = mvv >> (4 + sc), yFrac = mvv & (sc ? 31 : 15);
      const int64_t refOff = (int64_t)((pu.pos_y >> sc) + hq + yInt + orgY) * (l ? rs1 : rs0) + (pu.pos_x >> sc) + wq + xInt + orgX;
      if (l == 0) { d.ref0_off = refOff; d.frac_x0 = (int8_t)xFrac; d.frac_y0 = (int8_t)yFrac; }
      else        { d.ref1_off = refOff; d.frac_x1 = (int8_t)xFrac; d.frac_y1 = (int8_t)yFrac; }
    }
    out[pu.first_desc + sb] = d;
  }
}

// ---- one iteration of the affine gradient search (InterSearch::xAffineMotionEstimation, InterSearch.cpp:3456-3534) behind its prediction: error,
// both Sobel planes and the normal-equation sums in ONE pass over the PU, plus the distortion of the prediction the next cost check needs.
// One workgroup per PU: the prediction is staged in LDS once; a sample's two derivatives come from the same eight neighbours (nearest interior
// position, as affine_sobel_kernel), no derivative plane and no residue plane is written.
constexpr int AFI_MAX = 128, AFI_WAVE_MAX = 1024;          // PUs of up to AFI_WAVE_MAX samples are served by one wavefront each, larger ones by a workgroup
typedef const __attribute__((address_space(3))) Pel* AfiLdsPel;

// NT = 64: the wavefront owns the PU; NT = 256: the four wavefronts of the workgroup share it and their sums meet in `red`
template <int P, int NT>
__device__ __forceinline__ void afi_equations(const vvcgpu_affine_iter& d, const Pel* __restrict__ org, const Pel* predL, int w, int h, long long* out,
                                              long long (*red)[64], int tid)
{
  const int lane = tid & 63, wave = tid >> 6;
  long long acc[P][P + 1];
#pragma unroll
  for (int c = 0; c < P; c++)
#pragma unroll
    for (int r = 0; r <= P; r++) acc[c][r] = 0;
  for (int i = tid; i < w * h; i += NT)
  {
    const int j = i / w, k = i - j * w;
    const int yy = min(max(j, 1), h - 2), xx = min(max(k, 1), w - 2);
    const Pel* c = predL + yy * w + xx;
    const int x = c[1 - w] - c[-1 - w] + (c[1] << 1) - (c[-1] << 1) + c[1 + w] - c[-1 + w];
    const int y = c[w - 1] - c[-w - 1] + (c[w] << 1) - (c[-w] << 1) + c[w + 1] - c[-w + 1];
    int iC[P];
    if (P == 4) { iC[0] = x; iC[1] = k * x + j * y; iC[2] = y; iC[3] = j * x - k * y; }
    else        { iC[0] = x; iC[1] = k * x; iC[2] = y; iC[3] = k * y; iC[4] = j * x; iC[5] = j * y; }
    const long long r = (long long)(Pel)((int)org[(size_t)j * d.org_stride + k] - (int)predL[j * w + k]);      // the error block is a Pel block
#pragma unroll
    for (int col = 0; col < P; col++)
    {
#pragma unroll
      for (int row = 0; row < P; row++) acc[col][row] += (long long)iC[col] * iC[row];
      acc[col][P] += ((long long)iC[col] * r) << 3;
    }
  }
  constexpr int M = P == 6 ? 64 : 32;
  long long v[M];
#pragma unroll
  for (int i = 0; i < M; i++) v[i] = i < P * (P + 1) ? acc[i / (P + 1)][i % (P + 1)] : 0;
  const long long total = wave_transpose_sum<M>(v, lane);       // lane L: this wave's sum of value L
  const int row7 = lane / 7, col7 = lane - row7 * 7;
  const bool used = row7 >= 1 && row7 <= P && col7 <= P;
  const int src = used ? (row7 - 1) * (P + 1) + col7 : 0;
  if (NT == 64)
  {
    const long long val = __shfl(total, src & 63);
    if (lane < 49) out[lane] = used ? val : 0;
  }
  else
  {
    red[wave][lane] = total;
    __syncthreads();
    if (tid < 49) out[tid] = used ? red[0][src] + red[1][src] + red[2][src] + red[3][src] : 0;
  }
}

// distortion of rows [r0, r1) x 16 of the PU against the prediction in LDS, by one wavefront
__device__ __forceinline__ unsigned long long afi_dist(const vvcgpu_affine_iter& d, const Pel* __restrict__ org, const Pel* predL, int w, int h,
                                                       int distKind, int band0, int bandStep, int lane)
{
  // bands of sixteen rows: every Hadamard tile of an affine PU (both sides >= 16) is at most sixteen rows high, the tile shape is the whole PU's
  unsigned long long sum = 0;
  for (int b = band0; b * 16 < h; b += bandStep)
  {
    const Pel* o = org + (size_t)b * 16 * d.org_stride;
    AfiLdsPel c = (AfiLdsPel)predL + b * 16 * w;
    if (distKind == 1) sum += satd_block<64, AfiLdsPel>(o, d.org_stride, c, w, w, 16, lane, 0, h);
    else
    {
      unsigned s = 0;
      for (int i = lane; i < 16 * w; i += 64) { const int j = i / w, k = i - j * w; s += (unsigned)abs((int)o[(size_t)j * d.org_stride + k] - (int)c[j * w + k]); }
      sum += wave_sum_u64(s);
    }
  }
  return sum;
}

// PUs of more than AFI_WAVE_MAX samples: one workgroup each (the others leave at once)
__global__ __launch_bounds__(256) void affine_iter_kernel(const Pel* __restrict__ orgBase, const Pel* __restrict__ predBase,
                                                          const vvcgpu_affine_iter* __restrict__ items, int n, int distKind,
                                                          long long* __restrict__ coeffOut, unsigned long long* __restrict__ distOut)
{
  __shared__ __align__(16) Pel predL[AFI_MAX * AFI_MAX];
  __shared__ long long red[4][64];
  __shared__ unsigned long long distW[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const vvcgpu_affine_iter d = items[blockIdx.x];
  const int w = d.pu.w, h = d.pu.h;
  if (w < 1 || h < 1 || w > AFI_MAX || h > AFI_MAX)                 // outside the LDS tile (uniform over the workgroup, before any barrier): sentinel, no overrun
  {
    if (tid == 0 && distOut) distOut[blockIdx.x] = ~0ull;
    return;
  }
  if (w * h <= AFI_WAVE_MAX) return;
  const Pel* org = orgBase + d.org_off;
  const Pel* pred = predBase + d.pu.dst_off;
  for (int i = tid; i < w * h; i += 256) { const int j = i / w, k = i - j * w; predL[i] = pred[(size_t)j * d.pu.dst_stride + k]; }
  __syncthreads();
  if (distOut)
  {
    const unsigned long long sum = afi_dist(d, org, predL, w, h, distKind, wave, 4, lane);
    if (lane == 0) distW[wave] = sum;
  }
  long long* out = coeffOut + (size_t)blockIdx.x * 49;
  if (d.pu.six_param) afi_equations<6, 256>(d, org, predL, w, h, out, red, tid);
  else                afi_equations<4, 256>(d, org, predL, w, h, out, red, tid);
  if (distOut && tid == 0) distOut[blockIdx.x] = distW[0] + distW[1] + distW[2] + distW[3];       // behind the barrier of afi_equations
}

// the small PUs: one wavefront each, four per workgroup
__global__ __launch_bounds__(256) void affine_iter_small_kernel(const Pel* __restrict__ orgBase, const Pel* __restrict__ predBase,
                                                                const vvcgpu_affine_iter* __restrict__ items, int n, int distKind,
                                                                long long* __restrict__ coeffOut, unsigned long long* __restrict__ distOut)
{
  __shared__ __align__(16) Pel predAll[4][AFI_WAVE_MAX];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x * 4 + wave;
  if (b >= n) return;
  const vvcgpu_affine_iter d = items[b];
  const int w = d.pu.w, h = d.pu.h;
  if (w < 1 || h < 1 || w > AFI_MAX || h > AFI_MAX || w * h > AFI_WAVE_MAX) return;   // large or out-of-contract PUs: affine_iter_kernel answers (sentinel there)
  Pel* predL = predAll[wave];
  const Pel* org = orgBase + d.org_off;
  const Pel* pred = predBase + d.pu.dst_off;
  for (int i = lane; i < w * h; i += 64) { const int j = i / w, k = i - j * w; predL[i] = pred[(size_t)j * d.pu.dst_stride + k]; }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
  if (distOut)
  {
    const unsigned long long sum = afi_dist(d, org, predL, w, h, distKind, 0, 1, lane);
    if (lane == 0) distOut[b] = sum;
  }
  long long* out = coeffOut + (size_t)b * 49;
  if (d.pu.six_param) afi_equations<6, 64>(d, org, predL, w, h, out, nullptr, lane);
  else                afi_equations<4, 64>(d, org, predL, w, h, out, nullptr, lane);
}

}  // namespace

extern "C" __attribute__((visibility("hidden"))) int vvcgpu_mc_batch_impl(const vvc_pel* ref0_base, const vvc_pel* ref1_base, vvc_pel* dst_base, const vvcgpu_mc_desc* descs, int n, int bit_depth,
                                    int clp_min, int clp_max, void* stream, bool skip_fast, bool sub44);       // interp.hip (not part of the ABI)

extern "C" {

int vvcgpu_affine_sobel_batch(int vertical, const vvc_pel* pred_base, int32_t* deriv_base, const vvcgpu_afg_desc* descs, int n, void* stream)
{
  VVC_CHECK_ARG(n >= 0, "affine_sobel_batch: n %d", n);
  if (n == 0) return VVCGPU_OK;
  VVC_CHECK_ARG(pred_base && deriv_base && descs, "affine_sobel_batch: null pointer");
  VVC_CHECK_ARG(vertical == 0 || vertical == 1, "affine_sobel_batch: vertical %d", vertical);
  hipLaunchKernelGGL(affine_sobel_kernel, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, vertical, pred_base, deriv_base, descs, n);
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}

int vvcgpu_affine_equal_coeff_batch(const vvc_pel* resi_base, const int32_t* derivx_base, const int32_t* derivy_base,
                                    const vvcgpu_afe_desc* descs, int n, int64_t* out, void* stream)
{
  VVC_CHECK_ARG(n >= 0, "affine_equal_coeff_batch: n %d", n);
  if (n == 0) return VVCGPU_OK;
  VVC_CHECK_ARG(resi_base && derivx_base && derivy_base && descs && out, "affine_equal_coeff_batch: null pointer");
  hipLaunchKernelGGL(affine_equal_coeff_kernel, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, resi_base, derivx_base, derivy_base, descs, n,
                     reinterpret_cast<long long*>(out));
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}

int vvcgpu_affine_subblock_descs(const vvcgpu_affine_pu* pus, int n, int comp, int pic_w, int pic_h, int max_cu_w, int max_cu_h,
                                 int ref_origin_x, int ref_origin_y, int ref0_stride, int ref1_stride, vvcgpu_mc_desc* out, void* stream)
{
  VVC_CHECK_ARG(n >= 0, "affine_subblock_descs: n %d", n);
  if (n == 0) return VVCGPU_OK;
  VVC_CHECK_ARG(pus && out, "affine_subblock_descs: null pointer");
  VVC_CHECK_ARG(comp == 0 || comp == 1, "affine_subblock_descs: comp %d", comp);
  VVC_CHECK_ARG(pic_w > 0 && pic_h > 0 && max_cu_w > 0 && max_cu_h > 0 && ref0_stride > 0 && ref1_stride > 0, "affine_subblock_descs: geometry");
  hipLaunchKernelGGL(affine_subblock_descs_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, pus, (int)sizeof(vvcgpu_affine_pu), n, comp, pic_w, pic_h, max_cu_w, max_cu_h,
                     ref_origin_x, ref_origin_y, ref0_stride, ref1_stride, out);
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}

int vvcgpu_affine_pred_batch(const vvc_pel* ref0_base, const vvc_pel* ref1_base, vvc_pel* dst_base, const vvcgpu_affine_pu* pus, int n, int n_subblocks,
                             vvcgpu_mc_desc* subblock_ws, int comp, int pic_w, int pic_h, int max_cu_w, int max_cu_h, int ref_origin_x, int ref_origin_y,
                             int ref0_stride, int ref1_stride, int bit_depth, int clp_min, int clp_max, void* stream)
{
  VVC_CHECK_ARG(n >= 0, "affine_pred_batch: n %d", n);
  if (n == 0) return VVCGPU_OK;
  VVC_CHECK_ARG(ref0_base && dst_base && pus && subblock_ws, "affine_pred_batch: null pointer");
  VVC_CHECK_ARG(n_subblocks >= n, "affine_pred_batch: n_subblocks %d for %d PUs", n_subblocks, n);
  const int rc = vvcgpu_affine_subblock_descs(pus, n, comp, pic_w, pic_h, max_cu_w, max_cu_h, ref_origin_x, ref_origin_y, ref0_stride, ref1_stride, subblock_ws,
                                              stream);
  if (rc != VVCGPU_OK) return rc;
  // luma: every descriptor is a 4x4 block -- no fast-kernel launch, the packed 4x4 variant of the generic kernel; chroma (2x2): the plain one, no fast-kernel launch either
  return vvcgpu_mc_batch_impl(ref0_base, ref1_base ? ref1_base : ref0_base, dst_base, subblock_ws, n_subblocks, bit_depth, clp_min, clp_max, stream, true,
                              comp == 0);
}

int vvcgpu_affine_me_iter_batch(const vvc_pel* org_base, const vvc_pel* ref_base, vvc_pel* pred_base, const vvcgpu_affine_iter* items, int n,
                                int n_subblocks, vvcgpu_mc_desc* subblock_ws, int dist_kind, int pic_w, int pic_h, int max_cu_w, int max_cu_h,
                                int ref_origin_x, int ref_origin_y, int ref_stride, int bit_depth, int clp_min, int clp_max, int64_t* coeff_out,
                                uint64_t* dist_out, void* stream)
{
  VVC_CHECK_ARG(n >= 0, "affine_me_iter_batch: n %d", n);
  if (n == 0) return VVCGPU_OK;
  VVC_CHECK_ARG(org_base && ref_base && pred_base && items && subblock_ws && coeff_out, "affine_me_iter_batch: null pointer");
  VVC_CHECK_ARG(n_subblocks >= n, "affine_me_iter_batch: n_subblocks %d for %d PUs", n_subblocks, n);
  VVC_CHECK_ARG(dist_kind == 0 || dist_kind == 1, "affine_me_iter_batch: dist_kind %d (0 SAD, 1 Hadamard)", dist_kind);
  VVC_CHECK_ARG(pic_w > 0 && pic_h > 0 && max_cu_w > 0 && max_cu_h > 0 && ref_stride > 0, "affine_me_iter_batch: geometry");
  hipStream_t st = (hipStream_t)stream;
  // sub-block vectors -> sub-block prediction (the entry points a caller would chain itself) -> everything that reads the prediction, fused
  hipLaunchKernelGGL(affine_subblock_descs_kernel, dim3(n), dim3(256), 0, st, reinterpret_cast<const vvcgpu_affine_pu*>(items), (int)sizeof(vvcgpu_affine_iter),
                     n, 0, pic_w, pic_h, max_cu_w, max_cu_h, ref_origin_x, ref_origin_y, ref_stride, ref_stride, subblock_ws);
  VVC_LAUNCH_CHECK();
  const int rc = vvcgpu_mc_batch_impl(ref_base, ref_base, pred_base, subblock_ws, n_subblocks, bit_depth, clp_min, clp_max, stream, true, true);   // 4x4 luma only
  if (rc != VVCGPU_OK) return rc;
  // every PU is served by exactly one of the two: by size, which only the device knows
  hipLaunchKernelGGL(affine_iter_small_kernel, dim3(cdiv(n, 4)), dim3(256), 0, st, org_base, pred_base, items, n, dist_kind,
                     reinterpret_cast<long long*>(coeff_out), reinterpret_cast<unsigned long long*>(dist_out));
  hipLaunchKernelGGL(affine_iter_kernel, dim3(n), dim3(256), 0, st, org_base, pred_base, items, n, dist_kind, reinterpret_cast<long long*>(coeff_out),
                     reinterpret_cast<unsigned long long*>(dist_out));
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}

}  // extern "C"
