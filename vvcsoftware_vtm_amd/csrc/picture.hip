// picture.hip -- picture-level passes of the "next" row N4 for gfx950: border extension and picture hash (CRC / checksum).
//
// Reference behaviour reproduced (bit-exact):
//   Picture::extendPicBorder          CommonLib/Picture.cpp:996-1041
//   compCRC / calcCRC                 CommonLib/PicYuvMD5.cpp:83-141
//   compChecksum / calcChecksum       CommonLib/PicYuvMD5.cpp:143-181
//
// Both are single passes over a plane (HBM-bound by construction: P x 2 B read, margin x 2 B written).  The CRC register of
// the reference is  s <- (s * x + bit) mod P  with P = x^16 + x^12 + x^5 + 1, so after the whole message (and the 16 appended
// zero bits) it is  0xFFFF * x^(8N+16) + sum_i byte_i * x^(8 (N-1-i) + 16)  mod P: a GF(2)-linear sum.  Every lane reduces the
// 8 samples it loaded, multiplies by the power of x that moves them to their place inside the wavefront's 512-sample block,
// the wavefront XOR-reduces, blocks of one wavefront are chained with a constant multiplier, and every wavefront XORs its
// share into the result with the power of x of what follows it.
#include "common.h"

namespace {

constexpr unsigned CRC_POLY = 0x1021u;

__host__ __device__ inline unsigned gf_mul(unsigned a, unsigned b)       // a * b mod P, 16-bit operands
{
  unsigned r = 0;
#pragma unroll
  for (int i = 15; i >= 0; i--)
  {
    r = ((r << 1) & 0xFFFFu) ^ ((r & 0x8000u) ? CRC_POLY : 0u);
    if ((b >> i) & 1u) r ^= a;
  }
  return r;
}
__host__ __device__ inline unsigned gf_pow(unsigned base, unsigned long long n)   // base^n mod P
{
  unsigned r = 1, b = base;
  while (n)
  {
    if (n & 1ull) r = gf_mul(r, b);
    b = gf_mul(b, b);
    n >>= 1;
  }
  return r;
}

constexpr int CRC_SPL = 8;                    // samples per lane and step
constexpr int CRC_BLOCK = 64 * CRC_SPL;       // samples per wavefront and step

struct CrcConsts { unsigned laneBase, blockShift, waveBase, initTerm; };   // x^(bits of 8 samples), x^(bits of a block), x^(bits of a wavefront's blocks), 0xffff * x^(8N+16)

// The message is padded IN FRONT with zero samples up to nWaves * blocksPerWave blocks (a zero register stays zero on zero
// input), so every wavefront owns exactly blocksPerWave blocks and its share is shifted by a power of one constant.
__global__ __launch_bounds__(256) void crc_kernel(const Pel* __restrict__ plane, int stride, int w, int h, int bps, int blocksPerWave, int nWaves,
                                                  CrcConsts k, int vec, unsigned* __restrict__ out)
{
  __shared__ unsigned short tab[256];                     // tab[v] = v * x^16 mod P
  {
    unsigned v = threadIdx.x << 8;
    for (int i = 0; i < 8; i++) v = ((v << 1) & 0xFFFFu) ^ ((v & 0x8000u) ? CRC_POLY : 0u);
    tab[threadIdx.x] = (unsigned short)v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= nWaves) return;
  const long long N = (long long)w * h;
  const long long pad = (long long)nWaves * blocksPerWave * CRC_BLOCK - N;
  const unsigned laneShift = gf_pow(k.laneBase, (unsigned)(63 - lane));      // moves the lane's bytes to their place in the block
  unsigned acc = 0;                                       // register of this wavefront's blocks so far (lane-uniform)
  for (int bi = 0; bi < blocksPerWave; bi++)
  {
    const long long blockEnd = ((long long)wave * blocksPerWave + bi + 1) * CRC_BLOCK - pad;
    if (blockEnd <= 0) continue;                          // wholly inside the padding: contributes zero to a zero register
    const long long v0 = blockEnd - CRC_BLOCK + (long long)lane * CRC_SPL;     // first sample index of the lane (may be < 0)
    unsigned pel[CRC_SPL];
    if (vec)                                              // w, stride multiples of 8, 16-byte aligned plane: the 8 samples are one aligned uint4
    {
      uint4 q = make_uint4(0, 0, 0, 0);
      if (v0 >= 0)
      {
        const int y = (int)(v0 / w), x = (int)(v0 - (long long)y * w);
        q = *reinterpret_cast<const uint4*>(plane + (size_t)y * stride + x);
      }
      pel[0] = q.x & 0xFFFFu; pel[1] = q.x >> 16; pel[2] = q.y & 0xFFFFu; pel[3] = q.y >> 16;
      pel[4] = q.z & 0xFFFFu; pel[5] = q.z >> 16; pel[6] = q.w & 0xFFFFu; pel[7] = q.w >> 16;
    }
    else
    {
      long long i = v0;
      int y = 0, x = 0;
      if (i >= 0) { y = (int)(i / w); x = (int)(i - (long long)y * w); }
#pragma unroll
      for (int j = 0; j < CRC_SPL; j++, i++)
      {
        pel[j] = 0;
        if (i >= 0)
        {
          pel[j] = (unsigned short)plane[(size_t)y * stride + x];
          if (++x == w) { x = 0; y++; }
        }
      }
    }
    unsigned s = 0;
#pragma unroll
    for (int j = 0; j < CRC_SPL; j++)
    {
      s = (((s << 8) | (pel[j] & 0xFFu)) ^ tab[s >> 8]) & 0xFFFFu;
      if (bps == 2) s = (((s << 8) | (pel[j] >> 8)) ^ tab[s >> 8]) & 0xFFFFu;
    }
    s = gf_mul(s, laneShift);
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) s ^= (unsigned)__shfl_xor((int)s, m);
    acc = gf_mul(acc, k.blockShift) ^ s;
  }
  if (lane == 0)
  {
    unsigned r = gf_mul(gf_mul(acc, gf_pow(k.waveBase, (unsigned)(nWaves - 1 - wave))), CRC_POLY);    // * x^16: the appended zero bits
    if (wave == 0) r ^= k.initTerm;                       // the 0xffff start value, shifted through the whole message
    atomicXor(out, r);
  }
}

__global__ __launch_bounds__(256) void checksum_kernel(const Pel* __restrict__ plane, int stride, int w, int h, int twoBytes,
                                                       unsigned* __restrict__ out)
{
  __shared__ unsigned part[4];
  unsigned sum = 0;
  for (int y = blockIdx.x; y < h; y += gridDim.x)
  {
    const Pel* row = plane + (size_t)y * stride;
    const unsigned ym = (unsigned)(y & 0xff) ^ (unsigned)(y >> 8);
    for (int x = threadIdx.x; x < w; x += 256)
    {
      const unsigned mask = ((unsigned)(x & 0xff) ^ (unsigned)(x >> 8) ^ ym) & 0xFFu;     // uint8_t xor_mask (:150)
      const int pel = row[x];
      sum += (unsigned)((pel & 0xff) ^ (int)mask);
      if (twoBytes) sum += (unsigned)((pel >> 8) ^ (int)mask);
    }
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) sum += (unsigned)__shfl_xor((int)sum, m);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

// every margin sample = nearest picture sample; one thread per margin sample, rows of the padded plane
__global__ __launch_bounds__(256) void extend_border_kernel(Pel* __restrict__ plane, int stride, int w, int h, int mx, int my)
{
  const int y = (int)blockIdx.y - my;                        // padded row
  const int pw = w + 2 * mx;
  const bool inner = y >= 0 && y < h;
  const int n = inner ? 2 * mx : pw;                         // margin samples in this row
  const Pel* src = plane + (ptrdiff_t)min(max(y, 0), h - 1) * stride;
  Pel* dst = plane + (ptrdiff_t)y * stride;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
  {
    const int x = inner ? (i < mx ? i - mx : w + i - mx) : i - mx;
    dst[x] = src[min(max(x, 0), w - 1)];
  }
}

}  // namespace

extern "C" {

int vvcgpu_extend_border(vvc_pel* plane, int stride, int w, int h, int margin_x, int margin_y, void* stream)
{
  VVC_CHECK_ARG(plane, "extend_border: null pointer");
  VVC_CHECK_ARG(w > 0 && h > 0 && margin_x >= 0 && margin_y >= 0 && stride >= w + 2 * margin_x, "extend_border: %dx%d margins %d,%d stride %d",
                w, h, margin_x, margin_y, stride);
  if (margin_x == 0 && margin_y == 0) return VVCGPU_OK;
  VVC_CHECK_ARG(h + 2 * margin_y <= 65535, "extend_border: too many rows");
  const int gx = cdiv(margin_y > 0 ? w + 2 * margin_x : 2 * margin_x, 256);
  hipLaunchKernelGGL(extend_border_kernel, dim3(gx > 0 ? gx : 1, h + 2 * margin_y), dim3(256), 0, (hipStream_t)stream, plane, stride, w, h, margin_x,
                     margin_y);
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}

int vvcgpu_picture_hash(int method, const vvc_pel* plane, int stride, int w, int h, int bit_depth, uint32_t* out, void* stream)
{
  if (method == 0) { vvcgpu_set_error("picture_hash: MD5 is a serial chain per plane and is not offered on the device"); return VVCGPU_E_UNSUPPORTED; }
  VVC_CHECK_ARG(method == 1 || method == 2, "picture_hash: method %d", method);
  VVC_CHECK_ARG(plane && out, "picture_hash: null pointer");
  VVC_CHECK_ARG(w > 0 && h > 0 && stride >= w && bit_depth >= 1 && bit_depth <= 16, "picture_hash: %dx%d stride %d bit depth %d", w, h, stride, bit_depth);
  hipStream_t st = (hipStream_t)stream;
  VVC_HIP(hipMemsetAsync(out, 0, sizeof(uint32_t), st));
  if (method == 1)
  {
    const int bps = bit_depth > 8 ? 2 : 1;
    const long long N = (long long)w * h, nBlocks = (N + CRC_BLOCK - 1) / CRC_BLOCK;
    // ~1024 wavefronts (one per SIMD); fewer for small planes
    long long waves = nBlocks < 1024 ? nBlocks : 1024;
    const int bpw = (int)((nBlocks + waves - 1) / waves);
    waves = (nBlocks + bpw - 1) / bpw;
    CrcConsts k;
    k.laneBase = gf_pow(2u, 8ull * bps * CRC_SPL);
    k.blockShift = gf_pow(2u, 8ull * bps * CRC_BLOCK);
    k.waveBase = gf_pow(k.blockShift, (unsigned long long)bpw);
    k.initTerm = gf_mul(0xFFFFu, gf_pow(2u, 8ull * (unsigned long long)N * bps + 16ull));
    const int vec = (w % 8 == 0 && stride % 8 == 0 && (reinterpret_cast<uintptr_t>(plane) & 15) == 0) ? 1 : 0;
    hipLaunchKernelGGL(crc_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, plane, stride, w, h, bps, bpw, (int)waves, k, vec, out);
  }
  else
  {
    hipLaunchKernelGGL(checksum_kernel, dim3(h < 1024 ? h : 1024), dim3(256), 0, st, plane, stride, w, h, bit_depth > 8 ? 1 : 0, out);
  }
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}

}  // extern "C"
