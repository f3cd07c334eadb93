// rdpcm.hip -- residual DPCM of transform-skipped / lossless TUs (row T3 of SURVEY 8(a); a range-extension tool, off in the shipped cfgs).
//   vvcgpu_rdpcm_fwd_batch = TrQuant::applyForwardRDPCM (TrQuant.cpp:991-1045) with Quant::transformSkipQuantOneSample /
//                            invTrSkipDeQuantOneSample (Quant.cpp:911-1090, flat scaling lists)
//   vvcgpu_rdpcm_inv_batch = TrQuant::invRdpcmNxN (TrQuant.cpp:632-688)
// The forward form is a closed loop: every sample's delta is taken against the RECONSTRUCTED running sum, so a line (a column for vertical, a row
// for horizontal DPCM) is a serial chain; the lines of a TU and the TUs of a batch are independent: one wave per TU, one lane per line.
#include "common.h"

namespace {

__device__ __forceinline__ int ilog2(int v) { return 31 - __clz(v); }

struct TsQ { int transformShift, qBits, scale, invScale, rightShift, inMin, inMax; long long add; };

__device__ __forceinline__ TsQ ts_params(int w, int h, int qp, int bd, bool halfRound, bool intraSlice)
{
  TsQ q;
  const int per = qp / 6, rem = qp - 6 * per;
  q.transformShift = 15 - bd - ((ilog2(w) + ilog2(h)) >> 1);                // getTransformShift, maxLog2TrDynamicRange 15 (ChromaFormat.h:117-120)
  q.qBits = 14 + per + q.transformShift;                                    // QUANT_SHIFT + per + shift (:942)
  q.scale = rem == 0 ? 26214 : rem == 1 ? 23302 : rem == 2 ? 20560 : rem == 3 ? 18396 : rem == 4 ? 16384 : 14564;
  q.add = (long long)(halfRound ? 256 : (intraSlice ? 171 : 85)) << (q.qBits - 9);       // (:945; evaluated in 64 bits, stored as int there: fits)
  q.invScale = rem == 0 ? 40 : rem == 1 ? 45 : rem == 2 ? 51 : rem == 3 ? 57 : rem == 4 ? 64 : 72;
  q.rightShift = 6 - (q.transformShift + per);                              // IQUANT_SHIFT - (shift + per) (:1003)
  const int targetBits = min(16, 32 + q.rightShift - 7);                    // (:1050)
  q.inMin = -(1 << (targetBits - 1)); q.inMax = (1 << (targetBits - 1)) - 1;
  return q;
}
// transformSkipQuantOneSample (:947-975)
__device__ __forceinline__ int ts_quant_one(const TsQ& q, int resiDiff)
{
  const int tc = q.transformShift >= 0 ? resiDiff << q.transformShift : (resiDiff + (1 << (-q.transformShift - 1))) >> -q.transformShift;
  const int sign = tc < 0 ? -1 : 1;
  const long long tmp = (long long)abs(tc) * q.scale;
  const int lv = (int)((tmp + (int)q.add) >> q.qBits) * sign;
  return min(max(lv, -32768), 32767);
}
// invTrSkipDeQuantOneSample (:1046-1090): Intermediate_Int is `int` (TypeDef.h:374): the products wrap in 32 bits like the reference's
__device__ __forceinline__ short ts_dequant_one(const TsQ& q, int level)
{
  const int c = min(max(level, q.inMin), q.inMax);
  int v;
  if (q.rightShift > 0) v = (int)((unsigned)(c * q.invScale) + (1u << (q.rightShift - 1))) >> q.rightShift;
  else v = (int)((unsigned)(c * q.invScale) << -q.rightShift);
  v = min(max(v, -32768), 32767);
  if (q.transformShift >= 0) return (short)((v + (q.transformShift == 0 ? 0 : 1 << (q.transformShift - 1))) >> q.transformShift);
  return (short)(v << -q.transformShift);
}

__global__ __launch_bounds__(256) void rdpcm_fwd_kernel(const Pel* __restrict__ resiBase, TCoeff* __restrict__ coeffBase,
                                                        const vvcgpu_rdpcm_desc* __restrict__ descs, int n, int bd, unsigned* __restrict__ absSum)
{
  const int lane = threadIdx.x & 63, ti = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ti >= n) return;
  const vvcgpu_rdpcm_desc d = descs[ti];
  const int w = d.w, h = d.h, mode = d.mode;
  const Pel* resi = resiBase + d.resi_off;
  TCoeff* coeff = coeffBase + d.coeff_off;
  const TsQ q = ts_params(w, h, d.qp, bd, mode != 0, d.intra_slice != 0);
  const int sizeM1 = w * h - 1;
  // major axis = the lines (x for vertical DPCM, y otherwise), minor axis = the walk along a line (:1004-1007)
  const int nMajor = mode == 2 ? w : h, nMinor = mode == 2 ? h : w;
  unsigned sum = 0;
  for (int major = lane; major < nMajor; major += 64)
  {
    int acc = 0;
    for (int minor = 0; minor < nMinor; minor++)
    {
      const int x = mode == 2 ? major : minor, y = mode == 2 ? minor : major;
      const int sampleIndex = y * w + x, ci = d.rotate ? sizeM1 - sampleIndex : sampleIndex;
      const int delta = (int)resi[(size_t)y * d.resi_stride + x] - acc;
      int lv; short rec;
      if (d.lossless) { lv = delta; rec = (short)delta; }
      else { lv = ts_quant_one(q, delta); rec = ts_dequant_one(q, lv); }
      coeff[ci] = lv;
      sum += (unsigned)abs(lv);
      if (mode != 0) acc += rec;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  if (lane == 0) absSum[ti] = sum;
}

__global__ __launch_bounds__(256) void rdpcm_inv_kernel(Pel* __restrict__ resiBase, const vvcgpu_rdpcm_desc* __restrict__ descs, int n)
{
  const int lane = threadIdx.x & 63, ti = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ti >= n) return;
  const vvcgpu_rdpcm_desc d = descs[ti];
  if (d.mode == 0) return;
  Pel* resi = resiBase + d.resi_off;
  const int nMajor = d.mode == 2 ? d.w : d.h, nMinor = d.mode == 2 ? d.h : d.w;
  const ptrdiff_t stepMinor = d.mode == 2 ? d.resi_stride : 1, stepMajor = d.mode == 2 ? 1 : d.resi_stride;
  for (int major = lane; major < nMajor; major += 64)
  {
    Pel* p = resi + major * stepMajor;
    int acc = p[0];                                                          // 32-bit accumulator, the first sample is left as it is (:665-672)
    for (int minor = 1; minor < nMinor; minor++)
    {
      acc += p[minor * stepMinor];
      p[minor * stepMinor] = (Pel)min(max(acc, -32768), 32767);
    }
  }
}

}  // namespace

extern "C" {

int vvcgpu_rdpcm_fwd_batch(const vvc_pel* resi_base, vvc_coef* coeff_base, const vvcgpu_rdpcm_desc* descs, int n, int bit_depth, uint32_t* abs_sum,
                           void* stream)
{
  VVC_CHECK_ARG(n >= 0, "rdpcm_fwd_batch: n %d", n);
  if (n == 0) return VVCGPU_OK;
  VVC_CHECK_ARG(resi_base && coeff_base && descs && abs_sum, "rdpcm_fwd_batch: null pointer");
  VVC_CHECK_ARG(bit_depth >= 8 && bit_depth <= 10, "rdpcm_fwd_batch: bit depth %d outside 8..10", bit_depth);
  hipLaunchKernelGGL(rdpcm_fwd_kernel, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, resi_base, coeff_base, descs, n, bit_depth, abs_sum);
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}

int vvcgpu_rdpcm_inv_batch(vvc_pel* resi_base, const vvcgpu_rdpcm_desc* descs, int n, void* stream)
{
  VVC_CHECK_ARG(n >= 0, "rdpcm_inv_batch: n %d", n);
  if (n == 0) return VVCGPU_OK;
  VVC_CHECK_ARG(resi_base && descs, "rdpcm_inv_batch: null pointer");
  hipLaunchKernelGGL(rdpcm_inv_kernel, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, resi_base, descs, n);
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}

}  // extern "C"
