// stats.hip -- encoder-side statistics kernels for gfx950: SAO class statistics (S2) and ALF covariance (A3).
//
// Reference behaviour reproduced (bit-exact integers):
//   EncSampleAdaptiveOffset::getStatistics/getBlkStats  EncoderLib/EncSampleAdaptiveOffset.cpp:278-330,1122-1490
//   EncAdaptiveLoopFilter::deriveStatsForFiltering/getBlkStats/calcCovariance  EncoderLib/EncAdaptiveLoopFilter.cpp:1317-1515
// The reference accumulates the ALF products into doubles; every partial sum is an integer < 2^53, so the
// int64 sums produced here convert to exactly the same doubles in any summation order.
#include "common.h"

namespace {

// =====================================================================================================
// S2: SAO statistics.  One workgroup per CTU; the deblocked tile (+1 halo) is staged in LDS, each
// thread walks one column segment with the five classifiers evaluated together, EO class counters live in
// registers (statically indexed), BO bands in a packed 64-bit LDS histogram.  Two bodies with the same results:
// the scalar one (sao_stats_body: any CTU shape) and the packed one (sao_stats_body_pk, round 5: CTUs at least 64
// wide, 256-thread workgroups over strips of 16 rows per thread, two signs per packed instruction, categories as
// v_perm_b32 selectors, v_dot4_u32_u8 accumulation).
// =====================================================================================================
constexpr int SAO_MAX_CTU = 128;

// wave reduction of the packed accumulators (count << 21 | sum of (d + 1024)) into the workgroup's eo[4][5][2]
__device__ __forceinline__ void sao_stats_reduce(const unsigned (&acc)[4][5], int* eo)
{
  const int tid = threadIdx.x;
  // halving butterfly: at every step a lane keeps one half of its values and sends the
  // other half to its partner (lane ^ 32, 16, 8, 4, 2), so 10 + 5 + 3 + 2 + 1 + 1 = 22 exchanges do what 20 full 6-step reductions of
  // count and sum (240 exchanges) did; afterwards lane l holds the wave total of accumulator 10 b5 + 5 b4 + 3 b3 + 2 b2 + b1.
  {
    const int lane = tid & 63;
    const unsigned* v = &acc[0][0];
    unsigned a10[10], a5[5], a3[3], a2[2];
    { const bool hi = lane & 32;
#pragma unroll
      for (int i = 0; i < 10; i++) a10[i] = (hi ? v[i + 10] : v[i]) + (unsigned)__shfl_xor((int)(hi ? v[i] : v[i + 10]), 32); }
    { const bool hi = lane & 16;
#pragma unroll
      for (int i = 0; i < 5; i++) a5[i] = (hi ? a10[i + 5] : a10[i]) + (unsigned)__shfl_xor((int)(hi ? a10[i] : a10[i + 5]), 16); }
    { const bool hi = lane & 8;
#pragma unroll
      for (int i = 0; i < 3; i++) { const unsigned lo_v = a5[i], hi_v = i + 3 < 5 ? a5[i + 3] : 0u; a3[i] = (hi ? hi_v : lo_v) + (unsigned)__shfl_xor((int)(hi ? lo_v : hi_v), 8); } }
    { const bool hi = lane & 4;
#pragma unroll
      for (int i = 0; i < 2; i++) { const unsigned lo_v = a3[i], hi_v = i + 2 < 3 ? a3[i + 2] : 0u; a2[i] = (hi ? hi_v : lo_v) + (unsigned)__shfl_xor((int)(hi ? lo_v : hi_v), 4); } }
    const bool hi1 = lane & 2;
    unsigned a1 = (hi1 ? a2[1] : a2[0]) + (unsigned)__shfl_xor((int)(hi1 ? a2[0] : a2[1]), 2);
    a1 += (unsigned)__shfl_xor((int)a1, 1);
    const int sub3 = ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);              // index inside the group of three (3 = padding)
    const int k5 = ((lane >> 3) & 1) * 3 + sub3;                              // index inside the group of five (>= 5 = padding)
    if (!(lane & 1) && sub3 < 3 && k5 < 5)
    {
      const int idx = ((lane >> 5) & 1) * 10 + ((lane >> 4) & 1) * 5 + k5;   // == t * 5 + k
      const int c = (int)(a1 >> 21), d = (int)(a1 & 0x1FFFFFu) - 1024 * c;
      atomicAdd(&eo[idx * 2], d); atomicAdd(&eo[idx * 2 + 1], c);
    }
  }
}
// the CTU's output record (behind a barrier after the last reduce)
__device__ __forceinline__ void sao_stats_output(const int* eo, const unsigned long long* bo, long long* __restrict__ out, int cx, int cy, int wCtu, int nthreads)
{
  const int tid = threadIdx.x;
  long long* o = out + (size_t)(cy * wCtu + cx) * 320;
  for (int i = tid; i < 320; i += nthreads)
  {
    const int t = i >> 6, isCount = (i >> 5) & 1, k = i & 31;
    long long v = 0;
    if (t < 4) { if (k < 5) v = eo[(t * 5 + k) * 2 + isCount]; }
    else
    {
      unsigned long long pk = 0ull;
      for (int r = 0; r < 16; r++) pk += bo[r * 32 + k];
      const long long c = (long long)(pk >> 32);
      v = isCount ? c : (long long)(pk & 0xffffffffull) - 1024 * c;
    }
    o[i] = v;
  }
}

__device__ __forceinline__ void sao_stats_body(const int cx, const int cy, const int nthreads, const Pel* __restrict__ org, int ostride,
                                                        const Pel* __restrict__ rec, int rstride, int w, int h,
                                                        int ctuW, int ctuH, int wCtu, int boShift,
                                                        const uint8_t* __restrict__ availMap, int skipR, int skipB,
                                                        long long* __restrict__ out)
{
  extern __shared__ __align__(16) unsigned char smem[];
  const int pitch = ctuW + 2;
  short* tile = reinterpret_cast<short*>(smem);                                   // (ctuH+2) x pitch
  const int tileBytes = ((ctuH + 2) * pitch * 2 + 15) & ~15;
  unsigned long long* bo = reinterpret_cast<unsigned long long*>(smem + tileBytes);   // 16 replicas x 32 packed bands (same-band atomics serialise)
  int* eo = reinterpret_cast<int*>(smem + tileBytes + 16 * 32 * 8);                   // [4][5][2]

  const int tid = threadIdx.x;
  const int x0 = cx * ctuW, y0 = cy * ctuH;
  const int width = min(ctuW, w - x0), height = min(ctuH, h - y0);

  // staging by rows: a wave takes whole tile rows, its lanes the columns (no division by the run-time pitch per element)
  for (int r = tid >> 6; r < ctuH + 2; r += nthreads >> 6)
  {
    const Pel* row = rec + (size_t)min(max(y0 + r - 1, 0), h - 1) * rstride;
    for (int c = tid & 63; c < pitch; c += 64) tile[r * pitch + c] = row[min(max(x0 + c - 1, 0), w - 1)];
  }
  for (int i = tid; i < 16 * 32; i += nthreads) bo[i] = 0ull;
  if (tid < 40) eo[tid] = 0;
  __syncthreads();

  const int a = availMap ? availMap[cy * wCtu + cx] : ((cx > 0 ? 1 : 0) | (cy > 0 ? 4 : 0) | (cx > 0 && cy > 0 ? 16 : 0));
  const bool left = a & 1, above = (a >> 2) & 1, aboveLeft = (a >> 4) & 1;
  const bool right = x0 + ctuW < w, below = y0 + ctuH < h;
  const int startXe = left ? 0 : 1, endXe = right ? width - skipR : width - 1;
  const int endX90 = right ? width - skipR : width;              // also BO
  const int endY0 = below ? height - skipB : height;             // EO_0 and BO
  const int endYd = below ? height - skipB : height - 1;         // EO_90/135/45
  const int startY90 = above ? 0 : 1;

  const int groups = nthreads / ctuW;
  const int rpg = ctuH / groups;
  const int x = tid % ctuW, g = tid / ctuW;
  // per (EO class, category) one packed accumulator: count in bits 21.., sum of (d + 1024) below -- one compare, one select, one add
  // per category instead of two selects and two adds.  A thread walks <= 16 rows (host), so even the sum over the 64 lanes of a wave
  // stays inside the fields (sum <= 1024 * 2047 < 2^21, count <= 1024 < 2^11): the wave reduction below works on the packed words.
  unsigned acc[4][5];
#pragma unroll
  for (int t = 0; t < 4; t++)
#pragma unroll
    for (int e = 0; e < 5; e++) acc[t][e] = 0u;

  if (x < width)
  {
    const bool inXe = x >= startXe && x < endXe;
    const bool inX90 = x < endX90;
    const short* p = tile + (g * rpg) * pitch + x;        // row y-1, col x-1 (tile origin is (-1,-1))
    int r0[3], r1[3], r2[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { r0[k] = p[k]; r1[k] = p[pitch + k]; }
    const int yEnd = min((g + 1) * rpg, height);
    for (int y = g * rpg; y < yEnd; y++)
    {
      const short* q = p + (y - g * rpg + 2) * pitch;
#pragma unroll
      for (int k = 0; k < 3; k++) r2[k] = q[k];
      const int c = r1[1];
      const int d = (int)org[(size_t)(y0 + y) * ostride + x0 + x] - c;
      // sign(c - n) as a 3-way clamp of the difference (one v_med3 after the subtraction)
      auto sg = [](int v) { int r; asm("v_med3_i32 %0, %1, -1, 1" : "=v"(r) : "v"(v)); return r; };
      const int sl = sg(c - r1[0]), sr = sg(c - r1[2]), su = sg(c - r0[1]), sd = sg(c - r2[1]);
      const int sul = sg(c - r0[0]), sdr = sg(c - r2[2]), sur = sg(c - r0[2]), sdl = sg(c - r2[0]);
      const int e[4] = { 2 + sl + sr, 2 + su + sd, 2 + sul + sdr, 2 + sur + sdl };
      bool use[4];
      use[0] = inXe && y < endY0;
      use[1] = inX90 && y >= startY90 && y < endYd;
      use[2] = y == 0 ? (x == 0 ? (aboveLeft && (above ? endXe : 1) > 0) : (above && x < endXe)) : (inXe && y < endYd);
      use[3] = y == 0 ? (above && inXe) : (inXe && y < endYd);
      const unsigned val = (1u << 21) | (unsigned)(d + 1024);
      // category select without compares: m = one-hot of the category (0 when the sample is not used by this class), bit k of it
      // times the packed value goes into accumulator k -- a bit-field extract and a 24-bit multiply-add per category (val < 2^22)
#pragma unroll
      for (int t = 0; t < 4; t++)
      {
        const unsigned m = (use[t] ? 1u : 0u) << e[t];
#pragma unroll
        for (int k = 0; k < 5; k++)
        {
          const unsigned bit = (m >> k) & 1u;                // the compiler would turn a multiply by 0 / 1 back into mask, and, add
          asm("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(acc[t][k]) : "v"(bit), "v"(val));
        }
      }
      if (inX90 && y < endY0)
        atomicAdd(&bo[(tid & 15) * 32 + ((c >> boShift) & 31)], (1ull << 32) + (unsigned long long)(d + 1024));
#pragma unroll
      for (int k = 0; k < 3; k++) { r0[k] = r1[k]; r1[k] = r2[k]; }
    }
  }
  sao_stats_reduce(acc, eo);
  __syncthreads();
  sao_stats_output(eo, bo, out, cx, cy, wCtu, nthreads);
}

// ---- packed form (CTU width a multiple of 64, a thread's rows a multiple of four: the 128 / 64 CTUs of a picture) ------------------------
// The tile is staged TRANSPOSED (a column's rows are consecutive 16-bit words), so a thread reads four rows of its column and of the two
// neighbour columns as dword pairs and works on 2 x 16-bit packed values: one v_pk_sub / v_pk_max / v_pk_min per TWO signs, the +-1 row shifts
// are v_alignbit.  The category of a sample (e - 2 = s1 + s2 in -2..2) becomes a byte SELECTOR of v_perm_b32: per (class, category) one
// v_perm turns the four selectors of a four-row block into four one-hot bytes, and two v_dot4_u32_u8 against the bytes of (org - rec + 1024)
// add the block to the accumulators -- 3 instructions per (class, category) and FOUR samples where the scalar form has 2 per sample.
//   value bytes : LO = (d + 1024) & 255;  HC = ((d + 1024) >> 8) << 5 | 1   (count rides in the low five bits: a thread has <= 28 rows)
//   selectors   : (s1 + s2) & 7 = 6, 7, 0, 1, 2 for the categories 0..4; 0x0C (v_perm: constant zero) for a sample the class does not use
// Results are the integers of the scalar form (tests/test_gpu_stats.py, tests/golden/sao.npz).
// (inline assembly: the elementwise min / max builtins on 2 x i16 are scalarised into compares and selects by the compiler)
__device__ __forceinline__ unsigned sao_pk_sign(unsigned a, unsigned b)          // clamp(a - b, -1, 1) per 16-bit half (the subtraction saturates)
{
  unsigned r;
  asm("v_pk_sub_i16 %0, %1, %2 clamp\n\tv_pk_max_i16 %0, %0, -1\n\tv_pk_min_i16 %0, %0, 1 op_sel_hi:[1,0]" : "=&v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ unsigned sao_pk_add(unsigned a, unsigned b) { unsigned r; asm("v_pk_add_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ unsigned sao_pk_sub(unsigned a, unsigned b) { unsigned r; asm("v_pk_sub_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ unsigned sao_expand4(unsigned m) { return (m & 1u) | ((m & 2u) << 7) | ((m & 4u) << 14) | ((m & 8u) << 21); }   // bit j -> byte j

// The packed body runs in 256-thread workgroups (several per CU: one workgroup's staging and finish run beside the others' statistics) and
// takes a CTU as STRIPS of (256 / ctuW) x 16 rows -- a thread walks 16 rows of a strip, four blocks of four -- staged, walked and reduced one
// after the other; the per-CTU record is written once.  (One 1024-thread workgroup per CTU spent 37 of its 58 us per 4K picture outside
// the statistics: staging, finish and workgroup turnover with nothing else on the CU.)
constexpr int SAO_PK_THREADS = 256, SAO_PK_ROWS = 16;
__host__ __device__ __forceinline__ bool sao_stats_packed_ok(int nthreads, int ctuW, int ctuH)
{
  if (nthreads != SAO_PK_THREADS || (ctuW & 63) || ctuW > nthreads) return false;
  const int stripH = (nthreads / ctuW) * SAO_PK_ROWS;
  return (ctuH % stripH) == 0;
}

__device__ __forceinline__ void sao_stats_body_pk(const int cx, const int cy, const Pel* __restrict__ org, int ostride,
                                                  const Pel* __restrict__ rec, int rstride, int w, int h,
                                                  int ctuW, int ctuH, int wCtu, int boShift,
                                                  const uint8_t* __restrict__ availMap, int skipR, int skipB,
                                                  long long* __restrict__ out)
{
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int nthreads = SAO_PK_THREADS;
  const int groups = nthreads / ctuW, stripH = groups * SAO_PK_ROWS;
  const int pitchT = stripH + 6;                                                  // 16-bit words per column: (stripH / 2 + 3) dwords, odd -> 32 consecutive columns hit 32 banks
  short* tile = reinterpret_cast<short*>(smem);                                   // [ctuW + 2 columns][pitchT]: column x + 1, row (y - strip start) + 2
  const int tileBytes = ((ctuW + 2) * pitchT * 2 + 15) & ~15;
  unsigned long long* bo = reinterpret_cast<unsigned long long*>(smem + tileBytes);   // 16 replicas x 32 packed bands (same-band atomics serialise)
  int* eo = reinterpret_cast<int*>(smem + tileBytes + 16 * 32 * 8);                   // [4][5][2]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x0 = cx * ctuW, y0 = cy * ctuH;
  const int width = min(ctuW, w - x0), height = min(ctuH, h - y0);
  for (int i = tid; i < 16 * 32; i += nthreads) bo[i] = 0ull;
  if (tid < 40) eo[tid] = 0;

  const int a = availMap ? availMap[cy * wCtu + cx] : ((cx > 0 ? 1 : 0) | (cy > 0 ? 4 : 0) | (cx > 0 && cy > 0 ? 16 : 0));
  const bool left = a & 1, above = (a >> 2) & 1, aboveLeft = (a >> 4) & 1;
  const bool right = x0 + ctuW < w, below = y0 + ctuH < h;
  const int startXe = left ? 0 : 1, endXe = right ? width - skipR : width - 1;
  const int endX90 = right ? width - skipR : width;              // also BO
  const int endY0 = below ? height - skipB : height;             // EO_0 and BO
  const int endYd = below ? height - skipB : height - 1;         // EO_90/135/45
  const int startY90 = above ? 0 : 1;

  const int x = tid & (ctuW - 1), g = __builtin_amdgcn_readfirstlane(tid / ctuW);     // a wave lies inside one row group
  const bool inXe = x >= startXe && x < endXe, inX90 = x < endX90;                   // (both imply x < width)
  const unsigned keepE = inXe ? 0x07070707u : 0u, unusedE = inXe ? 0u : 0x0C0C0C0Cu;
  const unsigned keep90 = inX90 ? 0x07070707u : 0u, unused90 = inX90 ? 0u : 0x0C0C0C0Cu;
  // class 135 in row 0 of the CTU: column 0 looks at the above-left CTU, the others at the one above
  const bool c2row0 = x == 0 ? (aboveLeft && (above ? endXe : 1) > 0) : (above && x < endXe);
  const bool c3row0 = above && inXe;
  const unsigned* colL = reinterpret_cast<const unsigned*>(tile) + ((x * pitchT) >> 1);          // dword r of a column = rows 2 r - 2, 2 r - 1 of the strip
  const unsigned* colC = colL + (pitchT >> 1);
  const unsigned* colR = colC + (pitchT >> 1);
  const int xo = min(x, width - 1);
  const Pel* orgCol = org + (size_t)y0 * ostride + x0 + xo;                       // (offsets below: 32 bits, a plane is below 2^31 samples)

  for (int ys = 0; ys < height; ys += stripH)                                     // strips of the CTU (wave-uniform)
  {
    // ---- staging: an item = (pair of tile rows; column) = one dword of the tile.  A wave takes the row pairs wave, wave + 4, ..; its lanes the
    // columns lane and lane + 64; the columns beyond 128 (two of them for a 128-wide CTU) are one more item of the lanes of wave 0.  All loads of a
    // thread are issued before its first LDS store.
    {
      const int nPr = (stripH + 4) >> 1, nC = ctuW + 2;
      constexpr int SB = 5;                                                         // row pairs per wave: ceil(18 / 4), ceil(34 / 4) = 9 -> two rounds
      for (int pr0 = wave; pr0 < nPr; pr0 += 4 * SB)
      {
        unsigned va[SB][2], vb[SB][2];
#pragma unroll
        for (int k = 0; k < SB; k++)
        {
          const int pr = min(pr0 + 4 * k, nPr - 1);                                // (beyond the last pair: the last pair again, not stored)
          const unsigned rowA = (unsigned)(min(max(y0 + ys + 2 * pr - 2, 0), h - 1) * rstride), rowB = (unsigned)(min(max(y0 + ys + 2 * pr - 1, 0), h - 1) * rstride);
#pragma unroll
          for (int q = 0; q < 2; q++)
          {
            const int c = lane + 64 * q;
            const unsigned xc = (unsigned)min(max(x0 + min(c, nC - 1) - 1, 0), w - 1);
            va[k][q] = (unsigned short)rec[rowA + xc]; vb[k][q] = (unsigned short)rec[rowB + xc];
          }
        }
#pragma unroll
        for (int k = 0; k < SB; k++)
#pragma unroll
          for (int q = 0; q < 2; q++)
          {
            const int c = lane + 64 * q, pr = pr0 + 4 * k;
            if (c < nC && c < 128 && pr < nPr) reinterpret_cast<unsigned*>(tile)[((c * pitchT) >> 1) + pr] = va[k][q] | (vb[k][q] << 16);
          }
      }
      if (nC > 128 && wave == 0)                                                   // columns 128 .. nC - 1: (nC - 128) x nPr items
        for (int i = lane; i < (nC - 128) * nPr; i += 64)
        {
          const int c = 128 + (i >= nPr ? 1 : 0), pr = i - (i >= nPr ? nPr : 0);
          const unsigned xc = (unsigned)min(max(x0 + c - 1, 0), w - 1);
          const unsigned rowA = (unsigned)(min(max(y0 + ys + 2 * pr - 2, 0), h - 1) * rstride), rowB = (unsigned)(min(max(y0 + ys + 2 * pr - 1, 0), h - 1) * rstride);
          reinterpret_cast<unsigned*>(tile)[((c * pitchT) >> 1) + pr] = (unsigned)(unsigned short)rec[rowA + xc] | ((unsigned)(unsigned short)rec[rowB + xc] << 16);
        }
    }
    __syncthreads();

    unsigned accLo[4][5], accHc[4][5];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
      for (int e = 0; e < 5; e++) { accLo[t][e] = 0u; accHc[t][e] = 0u; }

    const int yBeg = ys + g * SAO_PK_ROWS, yEnd = min(yBeg + SAO_PK_ROWS, height);
    int orgN[4];                                                                   // the original samples of the NEXT block: requested a block ahead
#pragma unroll
    for (int j = 0; j < 4; j++) orgN[j] = orgCol[(unsigned)(min(yBeg + j, height - 1) * ostride)];
    for (int y = yBeg; y < yEnd; y += 4)
    {
      // dwords (y-2, y-1) (y, y+1) (y+2, y+3) (y+4, y+5) of the three columns
      const int r = (y - ys) >> 1;
      unsigned L[4], C[4], R[4];
#pragma unroll
      for (int i = 0; i < 4; i++) { L[i] = colL[r + i]; C[i] = colC[r + i]; R[i] = colR[r + i]; }
      int t4[4];
#pragma unroll
      for (int j = 0; j < 4; j++)
      {
        const int cj = (int)(short)(C[1 + (j >> 1)] >> ((j & 1) * 16));
        t4[j] = orgN[j] - cj + 1024;
      }
#pragma unroll
      for (int j = 0; j < 4; j++) orgN[j] = orgCol[(unsigned)(min(y + 4 + j, height - 1) * ostride)];
      // the rows one up / one down, per column: U0 = (y-1, y), U1 = D0 = (y+1, y+2), D1 = (y+3, y+4)
      unsigned UL[2], UC[2], UR[2], DL1, DC1, DR1;
      UL[0] = __builtin_amdgcn_alignbit(L[1], L[0], 16); UL[1] = __builtin_amdgcn_alignbit(L[2], L[1], 16); DL1 = __builtin_amdgcn_alignbit(L[3], L[2], 16);
      UC[0] = __builtin_amdgcn_alignbit(C[1], C[0], 16); UC[1] = __builtin_amdgcn_alignbit(C[2], C[1], 16); DC1 = __builtin_amdgcn_alignbit(C[3], C[2], 16);
      UR[0] = __builtin_amdgcn_alignbit(R[1], R[0], 16); UR[1] = __builtin_amdgcn_alignbit(R[2], R[1], 16); DR1 = __builtin_amdgcn_alignbit(R[3], R[2], 16);
      unsigned E[4][2];                                                            // s1 + s2 per class, two samples per dword
      {
        const unsigned sd0 = sao_pk_sign(C[1], UC[1]), sd1 = sao_pk_sign(C[2], DC1);
        const unsigned su0 = sao_pk_sign(C[1], UC[0]);
        // sign(c(y) - c(y-1)) = -sign(c(y-1) - c(y)): the "up" signs of rows y+2, y+3 are the negated "down" signs of rows y+1, y+2
        E[1][0] = sao_pk_add(su0, sd0); E[1][1] = sao_pk_sub(sd1, __builtin_amdgcn_alignbit(sd1, sd0, 16));
        E[0][0] = sao_pk_add(sao_pk_sign(C[1], L[1]), sao_pk_sign(C[1], R[1])); E[0][1] = sao_pk_add(sao_pk_sign(C[2], L[2]), sao_pk_sign(C[2], R[2]));
        E[2][0] = sao_pk_add(sao_pk_sign(C[1], UL[0]), sao_pk_sign(C[1], UR[1])); E[2][1] = sao_pk_add(sao_pk_sign(C[2], UL[1]), sao_pk_sign(C[2], DR1));
        E[3][0] = sao_pk_add(sao_pk_sign(C[1], UR[0]), sao_pk_sign(C[1], UL[1])); E[3][1] = sao_pk_add(sao_pk_sign(C[2], UR[1]), sao_pk_sign(C[2], DL1));
      }
      // selectors: the low byte of each 16-bit sum, then the samples a class does not use
      unsigned sel[4];
#pragma unroll
      for (int t = 0; t < 4; t++) sel[t] = __builtin_amdgcn_perm(E[t][1], E[t][0], 0x06040200u);
      if (y >= 1 && y + 3 < endYd)                                                 // wave-uniform: every row of the block is inside every class's rows
      {
        sel[0] = (sel[0] & keepE) | unusedE; sel[1] = (sel[1] & keep90) | unused90;
        sel[2] = (sel[2] & keepE) | unusedE; sel[3] = (sel[3] & keepE) | unusedE;
      }
      else
      {
        unsigned m0 = 0, m1 = 0, m23 = 0;
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
          m0 |= (y + j < endY0 ? 1u : 0u) << j;
          m1 |= (y + j >= startY90 && y + j < endYd ? 1u : 0u) << j;
          m23 |= (y + j >= 1 && y + j < endYd ? 1u : 0u) << j;
        }
        const unsigned k0 = sao_expand4(m0) * 7u, k1 = sao_expand4(m1) * 7u, k23 = sao_expand4(m23) * 7u;
        const unsigned u0 = sao_expand4(m0 ^ 15u) * 12u, u1 = sao_expand4(m1 ^ 15u) * 12u, u23 = sao_expand4(m23 ^ 15u) * 12u;
        unsigned kp[4] = { inXe ? k0 : 0u, inX90 ? k1 : 0u, inXe ? k23 : 0u, inXe ? k23 : 0u };
        unsigned un[4] = { inXe ? u0 : 0x0C0C0C0Cu, inX90 ? u1 : 0x0C0C0C0Cu, inXe ? u23 : 0x0C0C0C0Cu, inXe ? u23 : 0x0C0C0C0Cu };
        if (y == 0)                                                                // row 0 of the CTU: the diagonal classes look into the CTUs above
        {
          kp[2] = (kp[2] & ~0xFFu) | (c2row0 ? 0x07u : 0u); un[2] = (un[2] & ~0xFFu) | (c2row0 ? 0u : 0x0Cu);
          kp[3] = (kp[3] & ~0xFFu) | (c3row0 ? 0x07u : 0u); un[3] = (un[3] & ~0xFFu) | (c3row0 ? 0u : 0x0Cu);
        }
#pragma unroll
        for (int t = 0; t < 4; t++) sel[t] = (sel[t] & kp[t]) | un[t];
      }
      const unsigned p01 = (unsigned)t4[0] | ((unsigned)t4[1] << 16), p23 = (unsigned)t4[2] | ((unsigned)t4[3] << 16);
      const unsigned LO = __builtin_amdgcn_perm(p23, p01, 0x06040200u);
      const unsigned HC = (__builtin_amdgcn_perm(p23, p01, 0x07050301u) << 5) | 0x01010101u;
#pragma unroll
      for (int t = 0; t < 4; t++)
      {
        // category k <-> selector (k - 2) & 7: 6, 7, 0, 1, 2; bytes 0..3 of {hi, lo} come from lo, 4..7 from hi
        const unsigned oh0 = __builtin_amdgcn_perm(0x00010000u, 0u, sel[t]), oh1 = __builtin_amdgcn_perm(0x01000000u, 0u, sel[t]);
        const unsigned oh2 = __builtin_amdgcn_perm(0u, 0x00000001u, sel[t]), oh3 = __builtin_amdgcn_perm(0u, 0x00000100u, sel[t]);
        const unsigned oh4 = __builtin_amdgcn_perm(0u, 0x00010000u, sel[t]);
        const unsigned oh[5] = { oh0, oh1, oh2, oh3, oh4 };
#pragma unroll
        for (int k = 0; k < 5; k++)
        {
          accLo[t][k] = __builtin_amdgcn_udot4(oh[k], LO, accLo[t][k], false);
          accHc[t][k] = __builtin_amdgcn_udot4(oh[k], HC, accHc[t][k], false);
        }
      }
      if (inX90)
      {
#pragma unroll
        for (int j = 0; j < 4; j++)
          if (y + j < endY0)
          {
            const int cj = (int)(short)(C[1 + (j >> 1)] >> ((j & 1) * 16));
            atomicAdd(&bo[(tid & 15) * 32 + ((cj >> boShift) & 31)], (1ull << 32) + (unsigned long long)t4[j]);
          }
      }
    }
    // the strip's accumulators in the packed form of the scalar body (count << 21 | sum of (d + 1024)), reduced over the wave into eo
    unsigned acc[4][5];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
      for (int e = 0; e < 5; e++) acc[t][e] = ((accHc[t][e] & 31u) << 21) + accLo[t][e] + ((accHc[t][e] >> 5) << 8);
    sao_stats_reduce(acc, eo);
    __syncthreads();                                                               // the tile is free for the next strip / eo and bo are complete
  }
  sao_stats_output(eo, bo, out, cx, cy, wCtu, nthreads);
}

// one plane per launch (vvcgpu_sao_stats): 1024 threads per CTU -- a thread walks 16 rows of a 128 x 128 CTU instead of 64 (the row walk is
// the critical path of the workgroup and a 4K picture has only 510 luma CTUs for 256 CUs)
__global__ __launch_bounds__(1024) void sao_stats_kernel(const Pel* __restrict__ org, int ostride, const Pel* __restrict__ rec, int rstride, int w, int h,
                                                         int ctuW, int ctuH, int wCtu, int boShift, const uint8_t* __restrict__ availMap, int skipR,
                                                         int skipB, long long* __restrict__ out)
{
  if (sao_stats_packed_ok((int)blockDim.x, ctuW, ctuH))
    sao_stats_body_pk((int)blockIdx.x, (int)blockIdx.y, org, ostride, rec, rstride, w, h, ctuW, ctuH, wCtu, boShift, availMap, skipR, skipB, out);
  else
    sao_stats_body((int)blockIdx.x, (int)blockIdx.y, (int)blockDim.x, org, ostride, rec, rstride, w, h, ctuW, ctuH, wCtu, boShift, availMap, skipR, skipB, out);
}

// the three planes of a picture in one launch
struct SaoStatsPlane { const Pel* org; const Pel* rec; long long* out; int ostride, rstride, w, h, ctu, wCtu, skipR, skipB, wgEnd; };
struct SaoStats3 { SaoStatsPlane a[3]; const uint8_t* avail; int boShift, total, xcd; };
__global__ __launch_bounds__(1024) void sao_stats_picture_kernel(SaoStats3 p)
{
  const int b = vvc_xcd_index2((int)blockIdx.x, p.a[0].wgEnd, p.total, p.xcd);
  if (b < 0) return;
  const int c = b < p.a[0].wgEnd ? 0 : b < p.a[1].wgEnd ? 1 : 2;
  const SaoStatsPlane& a = c == 0 ? p.a[0] : c == 1 ? p.a[1] : p.a[2];
  const int r = b - (c == 0 ? 0 : c == 1 ? p.a[0].wgEnd : p.a[1].wgEnd);
  if (sao_stats_packed_ok((int)blockDim.x, a.ctu, a.ctu))
    sao_stats_body_pk(r % a.wCtu, r / a.wCtu, a.org, a.ostride, a.rec, a.rstride, a.w, a.h, a.ctu, a.ctu, a.wCtu, p.boShift, p.avail, a.skipR,
                      a.skipB, a.out);
  else
    sao_stats_body(r % a.wCtu, r / a.wCtu, (int)blockDim.x, a.org, a.ostride, a.rec, a.rstride, a.w, a.h, a.ctu, a.ctu, a.wCtu, p.boShift, p.avail, a.skipR,
                   a.skipB, a.out);
}


// =====================================================================================================
// A3: ALF covariance.  One workgroup per 64x64 tile (256 blocks of 4x4, one per thread): the rec tile with a
// 3-sample halo is staged in LDS (border replicated), each thread builds the 13 (7) symmetric tap sums of its
// 16 pixels in the canonical (transposeIdx 0) order and accumulates the upper triangle of E, y and pixAcc in
// registers with 24-bit integer MADs; the per-class buckets live in LDS (64-bit atomics, indices permuted by
// the block's transposeIdx) and are flushed to the per-CTU output with 64-bit global atomics.
// =====================================================================================================
constexpr int AT = 64, AP = AT + 8, AR = AT + 6;
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
typedef short short2v __attribute__((ext_vector_type(2)));

template <int PITCH>
__device__ __forceinline__ void load_tile_clamped(short* __restrict__ lds, const Pel* __restrict__ src,
                                                  int stride, int w, int h, int x0, int y0, int rows,
                                                  int tid, int nthreads)
{
  constexpr int VPR = PITCH / 4;
  const int nvec = rows * VPR;
  const bool vec_ok = ((stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(src) & 7) == 0);
  for (int v = tid; v < nvec; v += nthreads)
  {
    const int r = v / VPR, c = (v - r * VPR) * 4;
    int y = y0 + r;
    y = y < 0 ? 0 : (y >= h ? h - 1 : y);
    const int x = x0 + c;
    const Pel* row = src + (size_t)y * stride;
    pel4 val;
    if (vec_ok && x >= 0 && x + 3 < w) val = *reinterpret_cast<const pel4*>(row + x);
    else
    {
#pragma unroll
      for (int k = 0; k < 4; k++) { int xx = x + k; xx = xx < 0 ? 0 : (xx >= w ? w - 1 : xx); val[k] = row[xx]; }
    }
    *reinterpret_cast<pel4*>(lds + r * PITCH + c) = val;
  }
}

// the same in two halves, NB vectors per thread: tile_fetch issues every load into registers, tile_store writes them to LDS -- the simple
// loop pays one trip to memory per iteration, and a caller can put other work between the halves
template <int PITCH, int NB>
__device__ __forceinline__ void tile_fetch(pel4 (&val)[NB], const Pel* __restrict__ src, int stride, int w, int h, int x0, int y0, int rows,
                                           int tid, int nthreads)
{
  constexpr int VPR = PITCH / 4;
  const int nvec = rows * VPR;
  const bool vec_ok = ((stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(src) & 7) == 0);
#pragma unroll
  for (int u = 0; u < NB; u++)
  {
    const int v = tid + u * nthreads;
    val[u] = pel4{ 0, 0, 0, 0 };
    if (v >= nvec) continue;
    const int r = v / VPR, c = (v - r * VPR) * 4;
    int y = y0 + r;
    y = y < 0 ? 0 : (y >= h ? h - 1 : y);
    const int x = x0 + c;
    const Pel* row = src + (size_t)y * stride;
    if (vec_ok && x >= 0 && x + 3 < w) val[u] = *reinterpret_cast<const pel4*>(row + x);
    else
    {
#pragma unroll
      for (int k = 0; k < 4; k++) { int xx = x + k; xx = xx < 0 ? 0 : (xx >= w ? w - 1 : xx); val[u][k] = row[xx]; }
    }
  }
}
template <int PITCH, int NB>
__device__ __forceinline__ void tile_store(short* __restrict__ lds, const pel4 (&val)[NB], int rows, int tid, int nthreads)
{
  constexpr int VPR = PITCH / 4;
  const int nvec = rows * VPR;
#pragma unroll
  for (int u = 0; u < NB; u++)
  {
    const int v = tid + u * nthreads;
    if (v >= nvec) continue;
    const int r = v / VPR, c = (v - r * VPR) * 4;
    *reinterpret_cast<pel4*>(lds + r * PITCH + c) = val[u];
  }
}

// canonical tap index of offset (dy,dx), as in alf.hip (AdaptiveLoopFilter.cpp:600-636 == calcCovariance t=0)
template <bool IS7>
__device__ __forceinline__ constexpr int tapIndexS(int dy, int dx)
{
  if (dy < 0 || (dy == 0 && dx < 0)) { dy = -dy; dx = -dx; }
  if (IS7)
  {
    if (dy == 3) return dx == 0 ? 0 : -1;
    if (dy == 2) return dx == 1 ? 1 : dx == 0 ? 2 : dx == -1 ? 3 : -1;
    if (dy == 1) return (dx >= -2 && dx <= 2) ? 6 - dx : -1;
    return dx <= 3 ? 12 - dx : -1;
  }
  else
  {
    if (dy == 2) return dx == 0 ? 0 : -1;
    if (dy == 1) return (dx >= -1 && dx <= 1) ? 2 - dx : -1;
    if (dy == 0) return dx <= 2 ? 6 - dx : -1;
    return -1;
  }
}

// covariance sums of one 4x4 block in the canonical (transposeIdx 0) tap order: blk = LDS tile at (row by - 3, column bx - 4) of a tile
// with row pitch PITCH samples, orgBlk = the original at (by, bx).  A = upper triangle of E (row-major), Y = cross terms, pix = sum d^2.
template <bool IS7, int PITCH>
__device__ __forceinline__ void alf_block_acc(const short* __restrict__ blk, const Pel* __restrict__ orgBlk, int ostride,
                                              int (&A)[(IS7 ? 13 : 7) * ((IS7 ? 13 : 7) + 1) / 2], int (&Y)[IS7 ? 13 : 7], int& pix)
{
  constexpr int N = IS7 ? 13 : 7, R = IS7 ? 3 : 2;
#pragma unroll 1
  for (int i = 0; i < 4; i++)                  // pixel row by+i (not unrolled: one row of tap sums live at a time)
  {
    // two horizontally adjacent pixels share every instruction: tap sums are built as packed 16-bit pairs
    // (v_pk_add_u16, <= 2 * 1023) and multiplied with v_dot2_i32_i16 (two exact MACs per instruction)
    us2 E[2][N];
#pragma unroll
    for (int jp = 0; jp < 2; jp++)
#pragma unroll
      for (int k = 0; k < N; k++) E[jp][k] = us2{ 0, 0 };
    const short* p = blk + (i + 3 - R) * PITCH;     // row by+i-R, col bx-4
    unsigned cen[2];
#pragma unroll
    for (int r = 0; r <= 2 * R; r++)
    {
      unsigned d[6];                                                // samples bx-4 .. bx+7 as pairs
      const uint2 v0 = *reinterpret_cast<const uint2*>(p + r * PITCH);
      const uint2 v1 = *reinterpret_cast<const uint2*>(p + r * PITCH + 4);
      const uint2 v2 = *reinterpret_cast<const uint2*>(p + r * PITCH + 8);
      d[0] = v0.x; d[1] = v0.y; d[2] = v1.x; d[3] = v1.y; d[4] = v2.x; d[5] = v2.y;
      if (r == R) { cen[0] = d[2]; cen[1] = d[3]; }
      const int dy = r - R;
#pragma unroll
      for (int dx = -R; dx <= R; dx++)
      {
        const int k = tapIndexS<IS7>(dy, dx);
        if (k < 0) continue;
#pragma unroll
        for (int jp = 0; jp < 2; jp++)
        {
          const int s0 = 4 + 2 * jp + dx;                           // first sample of the pair (compile-time)
          const unsigned pr = (s0 & 1) ? __builtin_amdgcn_alignbit(d[(s0 + 1) >> 1], d[(s0 - 1) >> 1], 16) : d[s0 >> 1];
          E[jp][k] += __builtin_bit_cast(us2, pr);
        }
      }
    }
    const uint2 ov = *reinterpret_cast<const uint2*>(orgBlk + (size_t)i * ostride);
#pragma unroll
    for (int jp = 0; jp < 2; jp++)
    {
      const short2v yl = __builtin_bit_cast(short2v, jp ? ov.y : ov.x) - __builtin_bit_cast(short2v, cen[jp]);
      int idx = 0;
#pragma unroll
      for (int k = 0; k < N; k++)
      {
        const short2v ek = __builtin_bit_cast(short2v, E[jp][k]);
#pragma unroll
        for (int l = k; l < N; l++) { A[idx] = __builtin_amdgcn_sdot2(ek, __builtin_bit_cast(short2v, E[jp][l]), A[idx], false); idx++; }
        Y[k] = __builtin_amdgcn_sdot2(ek, yl, Y[k], false);
      }
      pix = __builtin_amdgcn_sdot2(yl, yl, pix, false);
    }
  }
}

template <bool IS7>
__device__ __forceinline__ void alf_stats_body(const int bidx, const int bidy, const Pel* __restrict__ org, int ostride,
                                                        const Pel* __restrict__ rec, int rstride, int w, int h,
                                                        int ctu, int wCtu, const uint16_t* __restrict__ cls,
                                                        int nCls, unsigned long long* __restrict__ out)
{
  constexpr int N = IS7 ? 13 : 7, R = IS7 ? 3 : 2;
  constexpr int NT = N * (N + 1) / 2;            // upper triangle
  constexpr int NB = NT + N + 1;                 // bucket entries: tri(E), y, pixAcc
  constexpr int RECSZ = N * N + N + 1;
  __shared__ short tile[AR * AP];
  // REP replicas of the class buckets, picked by the low lane bits: neighbouring blocks mostly share class and transpose, and
  // same-address LDS atomics serialise -- REP copies divide that queue; they are summed when the tile is flushed
  constexpr int REP = IS7 ? 2 : 4;              // 7x7: 2 x 21 KB keeps two workgroups per CU
  __shared__ unsigned long long bucket[REP * 25 * NB];
  const int tid = threadIdx.x;
  const int tx0 = bidx * AT, ty0 = bidy * AT;
  load_tile_clamped<AP>(tile, rec, rstride, w, h, tx0 - 4, ty0 - 3, AR, tid, 256);
  for (int i = tid; i < REP * 25 * NB; i += 256) bucket[i] = 0ull;
  __syncthreads();

  const int bj = tid & 15, bi = tid >> 4;
  const int bx = tx0 + 4 * bj, by = ty0 + 4 * bi;
  if (bx < w && by < h)
  {
    int A[NT], Y[N], pix = 0;
#pragma unroll
    for (int i = 0; i < NT; i++) A[i] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) Y[i] = 0;

    alf_block_acc<IS7, AP>(tile + (4 * bi) * AP + 4 * bj, org + (size_t)by * ostride + bx, ostride, A, Y, pix);
    // flush into the class bucket with the block's transpose permutation
    int classIdx = 0, t = 0;
    if (cls) { const uint16_t c = cls[(size_t)(by >> 2) * (w >> 2) + (bx >> 2)]; classIdx = c & 0xff; t = c >> 8; }
    unsigned long long perm;                    // nibble k = coefficient index that canonical tap k feeds
    if (IS7) perm = t == 0 ? 0xCBA9876543210ull : t == 1 ? 0xC62037B518A49ull : t == 2 ? 0xCBA9456781230ull : 0xC62015B734A89ull;
    else     perm = t == 0 ? 0x6543210ull : t == 1 ? 0x6203514ull : t == 2 ? 0x6541230ull : 0x6201534ull;
    // without a classifier (chroma) every block feeds class 0: spread over all REP * 25 slots instead
    constexpr int SLOTS1 = 1 << (31 - __builtin_clz(REP * 25));           // largest power of two <= REP * 25
    unsigned long long* b = bucket + (cls ? (tid & (REP - 1)) * 25 + classIdx : (tid & (SLOTS1 - 1))) * NB;
    int idx = 0;
#pragma unroll
    for (int k = 0; k < N; k++)
    {
      const int ck = (int)((perm >> (4 * k)) & 15);
#pragma unroll
      for (int l = k; l < N; l++)
      {
        const int cl = (int)((perm >> (4 * l)) & 15);
        const int lo = min(ck, cl), hi = max(ck, cl);
        // position of (lo,hi) in the row-major upper triangle
        const int pos = lo * N - (lo * (lo - 1)) / 2 + (hi - lo);
        atomicAdd(&b[pos], (unsigned long long)(long long)A[idx]);
        idx++;
      }
      atomicAdd(&b[NT + ck], (unsigned long long)(long long)Y[k]);
    }
    atomicAdd(&b[NT + N], (unsigned long long)(long long)pix);
  }
  __syncthreads();
  // flush LDS buckets to the CTU's record (a CTU is covered by several tiles -> global 64-bit atomics)
  const int ctuIdx = (ty0 / ctu) * wCtu + tx0 / ctu;
  unsigned long long* o = out + (size_t)ctuIdx * nCls * RECSZ;
  for (int i = tid; i < nCls * NB; i += 256)
  {
    unsigned long long v = 0ull;
    if (cls)
    {
#pragma unroll
      for (int r = 0; r < REP; r++) v += bucket[r * 25 * NB + i];
    }
    else
    {
      constexpr int SLOTS1 = 1 << (31 - __builtin_clz(REP * 25));
      for (int r = 0; r < SLOTS1; r++) v += bucket[r * NB + i];     // i < NB here (nCls == 1)
    }
    if (v == 0ull) continue;
    const int c = i / NB, e = i - c * NB;
    unsigned long long* oc = o + (size_t)c * RECSZ;
    if (e < NT)
    {
      // invert pos -> (lo,hi)
      int lo = 0, rem = e;
      while (rem >= N - lo) { rem -= N - lo; lo++; }
      const int hi = lo + rem;
      atomicAdd(&oc[lo * N + hi], v);
      if (hi != lo) atomicAdd(&oc[hi * N + lo], v);
    }
    else
      atomicAdd(&oc[N * N + (e - NT)], v);
  }
}

template <bool IS7>
__global__ __launch_bounds__(256) void alf_stats_kernel(const Pel* __restrict__ org, int ostride, const Pel* __restrict__ rec, int rstride, int w, int h,
                                                        int ctu, int wCtu, const uint16_t* __restrict__ cls, int nCls, unsigned long long* __restrict__ out)
{
  alf_stats_body<IS7>((int)blockIdx.x, (int)blockIdx.y, org, ostride, rec, rstride, w, h, ctu, wCtu, cls, nCls, out);
}

// both chroma planes (5x5, no classifier) in one launch: blockIdx.z selects the plane
__global__ __launch_bounds__(256) void alf_stats_chroma2_kernel(const Pel* __restrict__ orgCb, const Pel* __restrict__ orgCr, int ostride,
                                                                const Pel* __restrict__ recCb, const Pel* __restrict__ recCr, int rstride, int w, int h,
                                                                int ctu, int wCtu, unsigned long long* __restrict__ outCb, unsigned long long* __restrict__ outCr)
{
  const bool cr = blockIdx.z != 0;
  alf_stats_body<false>((int)blockIdx.x, (int)blockIdx.y, cr ? orgCr : orgCb, ostride, cr ? recCr : recCb, rstride, w, h, ctu, wCtu, nullptr, 1, cr ? outCr : outCb);
}

// ---------------------------------------------------------------------------------------------------
// Picture form of A3 (vvcgpu_alf_stats_picture): ONE workgroup of 512 threads per CTU, so a CTU's records are finished inside its
// workgroup -- no zero-fill pass, no 64-bit global atomics (the tile form above adds 25 x 183 partial sums per 64x64 tile into the
// CTU's record: 9 M global atomics for a 4K picture) and no 5x5-from-7x7 pass: the workgroup writes the 7x7 and the 5x5 record of
// every class directly.  Luma workgroup: the CTU with its 3-sample halo in LDS, four replicas of the 25 class buckets, two vertically
// adjacent 4x4 blocks per thread (flushed once when they share class and transposition); the (tap pair, transposition) -> bucket slot
// map is a 4 x 105 table in LDS, built once per workgroup, instead of nibble extraction and triangle index arithmetic per atomic.
// Chroma workgroup: both chroma CTUs (no classifier), one block per thread.
// ---------------------------------------------------------------------------------------------------
constexpr int ACT = 512;                          // threads per CTU workgroup
constexpr int AC_REC7 = 13 * 13 + 13 + 1;        // luma record: E, y, pixAcc

template <int C> struct AlfCtuLds
{
  static constexpr int P = C + 8, ROWS = C + 6;                          // luma tile
  static constexpr int tileBytes = (ROWS * P * 2 + 15) & ~15;
  static constexpr int zeroBytes = (3 * P * 2 + 16 + 15) & ~15;          // four rows of zeros at the tile's pitch: what an empty slot of a step reads
  static constexpr int bucketBytes = 25 * AC_REC7 * 8;                   // one 64-bit record per class
  static constexpr int MAXSTEPS = (C / 4) * (C / 4) / 4 + 25;            // steps of four blocks, every class padded to whole steps
  static constexpr int listBytes = MAXSTEPS * 4 * 4;                     // u32 per slot: centre sample index | transposition << 16 | block << 18
  static constexpr int stepClsBytes = (MAXSTEPS + 15) & ~15;
  static constexpr int stageBytes = (ACT / 64) * 128;                    // per wave: the org rows of the current step (4 blocks x 4 rows x 8 bytes)
  static constexpr int lumaBytes = tileBytes + zeroBytes + bucketBytes + listBytes + stepClsBytes + stageBytes + 64 * 4;
  static constexpr int CP = C / 2 + 8, CROWS = C / 2 + 6;                // chroma tiles (two planes)
  static constexpr int ctileBytes = (CROWS * CP * 2 + 15) & ~15;
  static constexpr int chromaBytes = 2 * ctileBytes + 2 * 57 * 8;
  static constexpr int bytes = lumaBytes > chromaBytes ? lumaBytes : chromaBytes;
};

struct AlfStatsPic
{
  const Pel* org[3]; const Pel* rec[3]; int ostride[3], rstride[3];
  int w, h, wCtu, nCtu, xcd;
  const uint16_t* cls;
  uint16_t* clsOut; int clsShift;                  // fused form (vvcgpu_alf_classify_stats_picture): the classes are DERIVED here and written out
  unsigned long long* out7; unsigned long long* out5; unsigned long long* outC[2];
};

typedef int alf_i4 __attribute__((ext_vector_type(4)));

// Luma CTU on the matrix cores.  The statistics of a class are the Gram matrix of x = (s_0 .. s_11, centre, org - rec) over the pixels of the
// class's blocks (s_k = the two samples of tap pair k, in FILTER coefficient order, i.e. behind the block's transposition): E = X X^T restricted
// to 13 x 13, y = its column 13, pixAcc = entry (13, 13).  Values need 11 bits (+ sign for org - rec), so x = 256 H + L with L = the low byte read
// as SIGNED (the matrix cores multiply signed bytes) and H = (x + 128) >> 8 in [-4, 8] = the high byte of x + 128: both limbs are byte picks
// (v_perm_b32), no masks or shifts; four v_mfma_i32_16x16x64_i8 per step of 64 pixels accumulate L L^T, H L^T, L H^T and H H^T exactly in int32
// (a whole CTU of one class stays below 2^29), and E = 65536 HH + 256 (HL + LH) + LL is formed in 64 bits when a class is finished.
//   * the CTU's 4x4 blocks are counting-sorted by class in LDS; a class is padded to whole steps of four blocks;
//   * lane (c, g) of a step builds variable c for the 16 pixels of block g: two unaligned 4-sample reads per row from the LDS tile (offset of
//     the tap that the block's transposition puts at coefficient c), one v_pk_add_u16, limb split, byte pack -- the SAME registers are the A
//     and the B operand, so the k order inside the instruction does not matter;
//   * the eight waves take contiguous ranges of steps; a wave adds its accumulators to the class's 64-bit LDS record when the class changes
//     (distinct addresses inside a wave, so the LDS atomics do not serialise).
// The old form (per-lane upper-triangle accumulation with v_dot2, ~1100 vector instructions and 105 serialising LDS atomics per block) took
// 83 us for a 3840x2160 picture; see DESIGN.md for the measured time of this one.
// CLS: the block classes are not read but derived from the tile (AdaptiveLoopFilter::deriveClassificationBlk, :248-455; the arithmetic of
// alf_classify_kernel in alf.hip: the tile has the same origin and clamping) and written to a.clsOut -- the classifier's own launch and its read
// of the picture are gone.  The Laplacian sums of the (C / 4 + 1)^2 4x4 quads live where the class records are accumulated later.
// `between` runs while the luma tile is on its way into the threads' registers (the kernel puts the CTU's chroma pair there).
template <int C, bool CLS, typename Between>
__device__ __forceinline__ void alf_ctu_luma(const AlfStatsPic& a, int ctuIdx, unsigned char* smem, Between between)
{
  using L = AlfCtuLds<C>;
  constexpr int P = L::P, BPR = C / 4, NBLK = BPR * BPR, S = (NBLK + ACT - 1) / ACT, NW = ACT / 64;
  constexpr int oZero = L::tileBytes, oBucket = oZero + L::zeroBytes, oList = oBucket + L::bucketBytes, oStepCls = oList + L::listBytes,
                oStage = oStepCls + L::stepClsBytes, oCnt = oStage + L::stageBytes;
  constexpr unsigned CEN_ZERO = (unsigned)(oZero / 2), EMPTY = CEN_ZERO | 0x80000000u;     // empty slot: centre in the zero rows, flag in bit 31
  short* tile = reinterpret_cast<short*>(smem);
  unsigned long long* bucket = reinterpret_cast<unsigned long long*>(smem + oBucket);
  unsigned* list = reinterpret_cast<unsigned*>(smem + oList);
  unsigned char* stepCls = smem + oStepCls;
  int* cnt = reinterpret_cast<int*>(smem + oCnt);                          // [32] counts, [32] first step of the class
  int* start = cnt + 32;
  const int tid = threadIdx.x;
  const int x0 = (ctuIdx % a.wCtu) * C, y0 = (ctuIdx / a.wCtu) * C;
  int myKey[S], myPos[S];
  {
    constexpr int NBT = (L::ROWS * (P / 4) + ACT - 1) / ACT;
    pel4 tv[NBT];
    tile_fetch<P, NBT>(tv, a.rec[0], a.rstride[0], a.w, a.h, x0 - 4, y0 - 3, L::ROWS, tid, ACT);
    // the classes of this thread's blocks: requested together with the tile (behind the barrier they were a second memory latency of the set-up)
#pragma unroll
    for (int s = 0; s < S; s++)
    {
      const int blk = tid + s * ACT, bi = blk % BPR, bj = blk / BPR;
      const int bx = x0 + 4 * bj, by = y0 + 4 * bi;
      myKey[s] = -1;
      if (!CLS && blk < NBLK && bx < a.w && by < a.h) myKey[s] = (int)a.cls[(size_t)(by >> 2) * (a.w >> 2) + (bx >> 2)];
    }
    between();                                                              // (the luma tile is on its way into this thread's registers)
    tile_store<P, NBT>(tile, tv, L::ROWS, tid, ACT);
  }
  for (int i = tid; i < L::zeroBytes / 4; i += ACT) reinterpret_cast<unsigned*>(smem + oZero)[i] = 0u;
  if (!CLS) for (int i = tid; i < 25 * AC_REC7; i += ACT) bucket[i] = 0ull;
  for (int i = tid; i < L::MAXSTEPS * 4; i += ACT) list[i] = EMPTY;
  if (tid < 64) cnt[tid] = 0;
  __syncthreads();
  if (CLS)
  {
    constexpr int QN = BPR + 1;
    static_assert(QN * QN * 16 <= L::bucketBytes, "quad sums fit into the record region");
    int* quad = reinterpret_cast<int*>(smem + oBucket);
    typedef unsigned short us2v __attribute__((ext_vector_type(2)));
    auto padd = [](unsigned x, unsigned y) { return __builtin_bit_cast(unsigned, __builtin_bit_cast(us2v, x) + __builtin_bit_cast(us2v, y)); };
    for (int q = tid; q < QN * QN; q += ACT)
    {
      // quad (qi, qj): picture rows y0 + 4 qi - 2 .. + 1, columns x0 + 4 qj - 2 .. + 1; two samples per instruction as in alf_classify_kernel
      const int qi = q / QN, qj = q - qi * QN;                              // row-major: neighbouring lanes read neighbouring 8-byte words of a tile row
      const unsigned* base = reinterpret_cast<const unsigned*>(tile + (4 * qi) * P + 4 * qj);
      unsigned sv = 0, sh = 0, sd0 = 0, sd1 = 0;
      unsigned A[5], B[5], Cc[5];
      auto loadRow = [&](int r, unsigned (&R)[5])
      {
        const uint2 lo = *reinterpret_cast<const uint2*>(base + r * (P / 2)), hi = *reinterpret_cast<const uint2*>(base + r * (P / 2) + 2);
        R[0] = __builtin_amdgcn_alignbit(lo.y, lo.x, 16); R[1] = lo.y; R[2] = __builtin_amdgcn_alignbit(hi.x, lo.y, 16); R[3] = hi.x;
        R[4] = __builtin_amdgcn_alignbit(hi.y, hi.x, 16);
      };
      loadRow(0, A); loadRow(1, B);
#pragma unroll
      for (int y = 0; y < 4; y++)
      {
        loadRow(y + 2, Cc);
        const unsigned c01 = padd(B[1], B[1]), c23 = padd(B[3], B[3]);
        sv  = __builtin_amdgcn_sad_u16(c01, padd(A[1], Cc[1]), sv);   sv  = __builtin_amdgcn_sad_u16(c23, padd(A[3], Cc[3]), sv);
        sh  = __builtin_amdgcn_sad_u16(c01, padd(B[0], B[2]), sh);    sh  = __builtin_amdgcn_sad_u16(c23, padd(B[2], B[4]), sh);
        sd0 = __builtin_amdgcn_sad_u16(c01, padd(A[0], Cc[2]), sd0);  sd0 = __builtin_amdgcn_sad_u16(c23, padd(A[2], Cc[4]), sd0);
        sd1 = __builtin_amdgcn_sad_u16(c01, padd(Cc[0], A[2]), sd1);  sd1 = __builtin_amdgcn_sad_u16(c23, padd(Cc[2], A[4]), sd1);
#pragma unroll
        for (int k = 0; k < 5; k++) { A[k] = B[k]; B[k] = Cc[k]; }
      }
      *reinterpret_cast<int4*>(quad + (qi * QN + qj) * 4) = make_int4((int)sv, (int)sh, (int)sd0, (int)sd1);
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < S; s++)
    {
      const int blk = tid + s * ACT, bi = blk % BPR, bj = blk / BPR;
      const int bx = x0 + 4 * bj, by = y0 + 4 * bi;
      if (blk < NBLK && bx < a.w && by < a.h)
      {
        const int4 q00 = *reinterpret_cast<const int4*>(quad + (bi * QN + bj) * 4), q01 = *reinterpret_cast<const int4*>(quad + (bi * QN + bj + 1) * 4);
        const int4 q10 = *reinterpret_cast<const int4*>(quad + ((bi + 1) * QN + bj) * 4), q11 = *reinterpret_cast<const int4*>(quad + ((bi + 1) * QN + bj + 1) * 4);
        const int sumV = q00.x + q01.x + q10.x + q11.x, sumH = q00.y + q01.y + q10.y + q11.y;
        const int sumD0 = q00.z + q01.z + q10.z + q11.z, sumD1 = q00.w + q01.w + q10.w + q11.w;
        const unsigned long long th = 0x4333333332222210ull;               // th[] of AdaptiveLoopFilter.cpp:294, 4 bits per entry
        const int activity = (short)clip3(0, 15, ((sumV + sumH) * 32) >> a.clsShift);
        int classIdx = (int)((th >> (4 * activity)) & 15);
        int hv1, hv0, d1, d0, dirHV, dirD;
        if (sumV > sumH) { hv1 = sumV; hv0 = sumH; dirHV = 1; } else { hv1 = sumH; hv0 = sumV; dirHV = 3; }
        if (sumD0 > sumD1) { d1 = sumD0; d0 = sumD1; dirD = 0; } else { d1 = sumD1; d0 = sumD0; dirD = 2; }
        int hvd1, hvd0, mainDir, secDir;
        if ((int)((unsigned)d1 * (unsigned)hv0) > (int)((unsigned)hv1 * (unsigned)d0)) { hvd1 = d1; hvd0 = d0; mainDir = dirD; secDir = dirHV; }
        else { hvd1 = hv1; hvd0 = hv0; mainDir = dirHV; secDir = dirD; }
        int strength = 0;
        if (hvd1 > 2 * hvd0) strength = 1;
        if (hvd1 * 2 > 9 * hvd0) strength = 2;
        if (strength) classIdx += (((mainDir & 1) << 1) + strength) * 5;
        const int transposeIdx = (0xDE84u >> (2 * (mainDir * 2 + (secDir >> 1)))) & 3;   // transposeTable (:447), 2 bits per entry
        myKey[s] = classIdx | (transposeIdx << 8);
        a.clsOut[(size_t)(by >> 2) * (a.w >> 2) + (bx >> 2)] = (uint16_t)myKey[s];
      }
    }
    __syncthreads();
    for (int i = tid; i < 25 * AC_REC7; i += ACT) bucket[i] = 0ull;        // (two barriers before the first record is added to)
  }
#pragma unroll
  for (int s = 0; s < S; s++)
  {
    // column-major thread -> block map (blk = tid + s ACT, row blk % BPR): the lanes of a wave are vertical neighbours, so the four blocks of a step
    // mostly are too -- their tile rows sit 16 LDS banks apart (4 rows x 68 dwords), while horizontal neighbours (2 banks apart) collide four-way
    // in every tile read
    myPos[s] = 0;
    // position inside the class: one LDS atomic per wave and distinct class (neighbouring blocks mostly share the class, and same-address
    // atomics serialise), the lanes of a class take consecutive positions behind the returned base
    const int myC = myKey[s] < 0 ? -1 : (myKey[s] & 0xff);
    unsigned long long todo = __ballot(myC >= 0);
    while (todo)
    {
      const int leader = __builtin_ctzll(todo);
      const int cc = __shfl(myC, leader);
      const unsigned long long m = __ballot(myC == cc);
      int base = 0;
      if ((tid & 63) == leader) base = atomicAdd(&cnt[cc], (int)__popcll(m));
      base = __shfl(base, leader);
      if (myC == cc) myPos[s] = base + (int)__popcll(m & ((1ull << (tid & 63)) - 1ull));
      todo &= ~m;
    }
  }
  __syncthreads();
  if (tid < 64)                                                             // first step of every class: an exclusive prefix sum over the lanes of one wave
  {
    const int steps = tid < 25 ? (cnt[tid] + 3) >> 2 : 0;
    int incl = steps;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) { const int v = __shfl_up(incl, o); if (tid >= o) incl += v; }
    if (tid <= 25) start[tid] = incl - steps;                                // (lane 25: the total)
  }
  __syncthreads();
  const int T = start[25];
#pragma unroll
  for (int s = 0; s < S; s++)
    if (myKey[s] >= 0)
    {
      const int blk = tid + s * ACT, bi = blk % BPR, bj = blk / BPR;
      list[start[myKey[s] & 0xff] * 4 + myPos[s]] = (unsigned)((4 * bi + 3) * P + 4 * bj + 4) | (unsigned)((myKey[s] >> 8) & 3) << 16 | (unsigned)(bi * BPR + bj) << 18;
    }
  for (int st = tid; st < T; st += ACT)
  {
    int c = 0;
    while (st >= start[c + 1]) c++;
    stepCls[st] = (unsigned char)c;
  }
  __syncthreads();

  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, c = lane & 15, g = lane >> 4;
  // sample offset of the tap that transposition t puts at coefficient c (geometric tap k sits at coefficient perm_t[k])
  constexpr int geoDy[13] = { 3, 2, 2, 2, 1, 1, 1, 1, 1, 0, 0, 0, 0 }, geoDx[13] = { 0, 1, 0, -1, 2, 1, 0, -1, -2, 3, 2, 1, 0 };
  int offT[4] = { 0, 0, 0, 0 };
#pragma unroll
  for (int t = 0; t < 4; t++)
  {
    const unsigned long long perm = t == 0 ? 0xCBA9876543210ull : t == 1 ? 0xC62037B518A49ull : t == 2 ? 0xCBA9456781230ull : 0xC62015B734A89ull;
#pragma unroll
    for (int k = 0; k < 12; k++)
      if ((int)((perm >> (4 * k)) & 15) == c) offT[t] = geoDy[k] * P + geoDx[k];
  }
  // second operand of the pair sum: + for the taps, 0 for the centre, - for org - rec (first operand = org there)
  const unsigned sgn = c < 12 ? 0x00010001u : (c == 12 ? 0u : 0xFFFFFFFFu);
  const bool isD = c == 13, dead = c >= 14;
  const int s0 = (wv * T) / NW, s1 = ((wv + 1) * T) / NW;
  alf_i4 LLa = { 0, 0, 0, 0 }, HLa = { 0, 0, 0, 0 }, LHa = { 0, 0, 0, 0 }, HHa = { 0, 0, 0, 0 };
  int cur = -1;
  auto flush = [&](int cl)
  {
#pragma unroll
    for (int r = 0; r < 4; r++)
    {
      const int row = 4 * g + r;
      const long long v = (long long)HHa[r] * 65536 + (long long)(HLa[r] + LHa[r]) * 256 + (long long)LLa[r];
      int e = -1;
      if (row < 13 && c < 13) e = row * 13 + c;
      else if (row == 13 && c < 13) e = 169 + c;
      else if (row == 13 && c == 13) e = 182;
      if (e >= 0) atomicAdd(&bucket[cl * AC_REC7 + e], (unsigned long long)v);
    }
    LLa = HLa = LHa = HHa = alf_i4{ 0, 0, 0, 0 };
  };
  // org rows of a step (four blocks x four rows of 8 bytes) are loaded FOUR steps ahead by lanes 0..15 (lane 4 g + i: row i of block g), kept
  // in registers, and written to the wave's 128-byte LDS stage when their step comes: the org - rec lanes read their first operand there
  // (row pitch 8 bytes) exactly as the other lanes read theirs in the tile (row pitch 2 P bytes)
  const unsigned stageOff = (unsigned)(oStage + wv * 128);
  const Pel* orgRowBase = a.org[0] + (size_t)(y0 + (lane & 3)) * a.ostride[0] + x0;
  const unsigned rowStep = isD ? 8u : (unsigned)(2 * P);
  auto issue = [&](int stp, uint2& dst)
  {
    dst = make_uint2(0u, 0u);
    if (lane < 16 && stp < s1)
    {
      const unsigned e2 = list[stp * 4 + (lane >> 2)];
      if ((int)e2 >= 0)
      {
        const unsigned blk2 = e2 >> 18;
        dst = *reinterpret_cast<const uint2*>(orgRowBase + (size_t)(4 * (blk2 / BPR)) * a.ostride[0] + 4 * (blk2 % BPR));
      }
    }
  };
  auto step = [&](int st, const uint2& orgRow)
  {
    const int cl = __builtin_amdgcn_readfirstlane((int)stepCls[st]);
    if (cl != cur) { if (cur >= 0) flush(cur); cur = cl; }
    if (lane < 16) *reinterpret_cast<uint2*>(smem + stageOff + lane * 8) = orgRow;
    const unsigned e = list[st * 4 + g];
    const int t = (int)(e >> 16) & 3;
    const int offLo = (t & 1) ? offT[1] : offT[0], offHi = (t & 1) ? offT[3] : offT[2];
    const int off = (int)e < 0 ? 0 : ((t & 2) ? offHi : offLo);           // empty slot: both operands in the zero rows
    const int cen = dead ? (int)CEN_ZERO : (int)(e & 0xFFFFu);
    const int ap = cen + off, am = cen - off;                               // first samples of the two 4-sample rows (row i: + i P); cen is even
    const unsigned sh = (unsigned)(off & 1) << 4;
    const unsigned adP = isD ? (stageOff + (unsigned)g * 32u) : (unsigned)(ap >> 1) << 2;
    const unsigned adM = (unsigned)(am >> 1) << 2;
    alf_i4 aL, aH;
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
      const unsigned* dp = reinterpret_cast<const unsigned*>(smem + adP + i * rowStep);
      const unsigned* dm = reinterpret_cast<const unsigned*>(smem + adM + i * 2 * P);
      const unsigned p0 = dp[0], p1 = dp[1], p2 = dp[2], m0 = dm[0], m1 = dm[1], m2 = dm[2];
      const unsigned P01 = __builtin_amdgcn_alignbit(p1, p0, sh), P23 = __builtin_amdgcn_alignbit(p2, p1, sh);
      const unsigned M01 = __builtin_amdgcn_alignbit(m1, m0, sh), M23 = __builtin_amdgcn_alignbit(m2, m1, sh);
      const us2 v01 = __builtin_bit_cast(us2, P01) + __builtin_bit_cast(us2, M01) * __builtin_bit_cast(us2, sgn);
      const us2 v23 = __builtin_bit_cast(us2, P23) + __builtin_bit_cast(us2, M23) * __builtin_bit_cast(us2, sgn);
      const us2 bias = { 128, 128 };
      const us2 h01 = v01 + bias, h23 = v23 + bias;
      aL[i] = (int)__builtin_amdgcn_perm(__builtin_bit_cast(unsigned, v23), __builtin_bit_cast(unsigned, v01), 0x06040200u);   // low bytes, signed
      aH[i] = (int)__builtin_amdgcn_perm(__builtin_bit_cast(unsigned, h23), __builtin_bit_cast(unsigned, h01), 0x07050301u);   // (v + 128) >> 8
    }
    LLa = __builtin_amdgcn_mfma_i32_16x16x64_i8(aL, aL, LLa, 0, 0, 0);
    HHa = __builtin_amdgcn_mfma_i32_16x16x64_i8(aH, aH, HHa, 0, 0, 0);
    HLa = __builtin_amdgcn_mfma_i32_16x16x64_i8(aH, aL, HLa, 0, 0, 0);
    LHa = __builtin_amdgcn_mfma_i32_16x16x64_i8(aL, aH, LHa, 0, 0, 0);
  };
  uint2 ring[4];
#pragma unroll
  for (int u = 0; u < 4; u++) issue(s0 + u, ring[u]);
  for (int st = s0; st < s1; st += 4)
  {
#pragma unroll
    for (int u = 0; u < 4; u++)
    {
      if (st + u < s1)                                                      // wave-uniform
      {
        step(st + u, ring[u]);
        issue(st + u + 4, ring[u]);
      }
    }
  }
  if (cur >= 0) flush(cur);
  __syncthreads();
  // records: 7x7 (13 x 13 + 13 + 1 = 183 entries per class) and its 5x5 sub-record (coefficient i of 5x5 = coefficient sig[i] of 7x7)
  unsigned long long* o7 = a.out7 + (size_t)ctuIdx * 25 * AC_REC7;
  for (int i = tid; i < 25 * AC_REC7; i += ACT) o7[i] = bucket[i];
  unsigned long long* o5 = a.out5 + (size_t)ctuIdx * 25 * 57;
  for (int i = tid; i < 25 * 57; i += ACT)
  {
    const int cc = i / 57, e = i - cc * 57;
    const int sig[7] = { 2, 5, 6, 7, 10, 11, 12 };
    o5[i] = bucket[cc * AC_REC7 + (e < 49 ? sig[e / 7] * 13 + sig[e % 7] : e < 56 ? 169 + sig[e - 49] : 182)];
  }
}

// Chroma CTU pair, same form: the 16 variables are (six tap pairs, centre, org - rec) of Cb and of Cr at the same pixel, so one Gram matrix holds
// the Cb record in its upper-left and the Cr record in its lower-right 8 x 8 (the cross terms are dropped).  No classes, no transposition.
// what the chroma part of a CTU needs from memory, requested BEFORE the luma part runs: the org rows of every step of this wave
// (lanes 0..31: plane, block, row) and this thread's share of the two reconstruction tiles
template <int C> struct AlfChromaPre
{
  using L = AlfCtuLds<C>;
  static constexpr int NS = (C / 8) * (C / 8) / 4, NW = ACT / 64, NSW = (NS + NW - 1) / NW, NBT = (L::CROWS * (L::CP / 4) + ACT - 1) / ACT;
  uint2 orgR[NSW];
  pel4 tv[2][NBT];
};

template <int C>
__device__ __forceinline__ void alf_chroma_prefetch(const AlfStatsPic& a, int ctuIdx, AlfChromaPre<C>& pre)
{
  using L = AlfCtuLds<C>;
  using PR = AlfChromaPre<C>;
  constexpr int C2 = C / 2, BPR = C2 / 4;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int w2 = a.w >> 1, h2 = a.h >> 1;
  const int x0 = (ctuIdx % a.wCtu) * C2, y0 = (ctuIdx / a.wCtu) * C2;
  const int s0 = (wv * PR::NS) / PR::NW, s1 = ((wv + 1) * PR::NS) / PR::NW;
  const int pl2 = (lane >> 4) & 1, g2 = (lane >> 2) & 3, i2 = lane & 3;
#pragma unroll
  for (int u = 0; u < PR::NSW; u++)
  {
    pre.orgR[u] = make_uint2(0u, 0u);
    const int blk = (s0 + u) * 4 + g2, bi = blk / BPR, bj = blk % BPR;
    if (lane < 32 && s0 + u < s1 && x0 + 4 * bj < w2 && y0 + 4 * bi < h2)
      pre.orgR[u] = *reinterpret_cast<const uint2*>(a.org[1 + pl2] + (size_t)(y0 + 4 * bi + i2) * a.ostride[1 + pl2] + x0 + 4 * bj);
  }
#pragma unroll
  for (int q = 0; q < 2; q++)
    tile_fetch<L::CP, PR::NBT>(pre.tv[q], a.rec[1 + q], a.rstride[1 + q], w2, h2, x0 - 4, y0 - 3, L::CROWS, tid, ACT);
}

template <int C>
__device__ __forceinline__ void alf_ctu_chroma(const AlfStatsPic& a, int ctuIdx, unsigned char* smem, const AlfChromaPre<C>& pre)
{
  using L = AlfCtuLds<C>;
  constexpr int P = L::CP, C2 = C / 2, BPR = C2 / 4, NBLK = BPR * BPR, NS = NBLK / 4, NW = ACT / 64, NSW = (NS + NW - 1) / NW;
  constexpr int oBucket = 2 * L::ctileBytes, oStage = oBucket + 2 * 57 * 8 + 16;
  static_assert(NS % NW == 0 || NS < NW, "chroma steps per wave");
  const int tid = threadIdx.x;
  const int w2 = a.w >> 1, h2 = a.h >> 1;
  const int x0 = (ctuIdx % a.wCtu) * C2, y0 = (ctuIdx / a.wCtu) * C2;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, c = lane & 15, g = lane >> 4, pl = c >> 3, cc = c & 7;
  const int s0 = (wv * NS) / NW, s1 = ((wv + 1) * NS) / NW;
  unsigned long long* bucket = reinterpret_cast<unsigned long long*>(smem + oBucket);       // [plane][57]
#pragma unroll
  for (int q = 0; q < 2; q++) tile_store<P, AlfChromaPre<C>::NBT>(reinterpret_cast<short*>(smem + q * L::ctileBytes), pre.tv[q], L::CROWS, tid, ACT);
  for (int i = tid; i < 2 * 57; i += ACT) bucket[i] = 0ull;
  __syncthreads();
  constexpr int geoDy[7] = { 2, 1, 1, 1, 0, 0, 0 }, geoDx[7] = { 0, 1, 0, -1, 2, 1, 0 };
  int off = 0;
#pragma unroll
  for (int k = 0; k < 6; k++) if (cc == k) off = geoDy[k] * P + geoDx[k];
  const unsigned sgn = cc < 6 ? 0x00010001u : (cc == 6 ? 0u : 0xFFFFFFFFu);
  const bool isD = cc == 7;
  const unsigned tileOff = (unsigned)(pl * L::ctileBytes), stageOff = (unsigned)(oStage + wv * 256);
  const unsigned rowStep = isD ? 8u : (unsigned)(2 * P);
  const unsigned sh = (unsigned)(off & 1) << 4;
  alf_i4 LLa = { 0, 0, 0, 0 }, HLa = { 0, 0, 0, 0 }, LHa = { 0, 0, 0, 0 }, HHa = { 0, 0, 0, 0 };
#pragma unroll
  for (int u = 0; u < NSW; u++)
  {
    const int st = s0 + u;
    if (st < s1)                                                            // wave-uniform
    {
      if (lane < 32) *reinterpret_cast<uint2*>(smem + stageOff + lane * 8) = pre.orgR[u];
      const int blk = st * 4 + g, bi = blk / BPR, bj = blk % BPR;
      const bool valid = x0 + 4 * bj < w2 && y0 + 4 * bi < h2;
      const int cen = (4 * bi + 3) * P + 4 * bj + 4;
      const int ap = cen + off, am = cen - off;
      const unsigned adP = isD ? stageOff + (unsigned)(pl * 128 + g * 32) : tileOff + ((unsigned)(ap >> 1) << 2);
      const unsigned adM = tileOff + ((unsigned)(am >> 1) << 2);
      alf_i4 aL, aH;
#pragma unroll
      for (int i = 0; i < 4; i++)
      {
        const unsigned* dp = reinterpret_cast<const unsigned*>(smem + adP + i * rowStep);
        const unsigned* dm = reinterpret_cast<const unsigned*>(smem + adM + i * 2 * P);
        const unsigned p0 = dp[0], p1 = dp[1], p2 = dp[2], m0 = dm[0], m1 = dm[1], m2 = dm[2];
        const unsigned P01 = __builtin_amdgcn_alignbit(p1, p0, sh), P23 = __builtin_amdgcn_alignbit(p2, p1, sh);
        const unsigned M01 = __builtin_amdgcn_alignbit(m1, m0, sh), M23 = __builtin_amdgcn_alignbit(m2, m1, sh);
        const us2 v01 = __builtin_bit_cast(us2, P01) + __builtin_bit_cast(us2, M01) * __builtin_bit_cast(us2, sgn);
        const us2 v23 = __builtin_bit_cast(us2, P23) + __builtin_bit_cast(us2, M23) * __builtin_bit_cast(us2, sgn);
        const us2 bias = { 128, 128 };
        const us2 h01 = v01 + bias, h23 = v23 + bias;
        aL[i] = (int)__builtin_amdgcn_perm(__builtin_bit_cast(unsigned, v23), __builtin_bit_cast(unsigned, v01), 0x06040200u);
        aH[i] = (int)__builtin_amdgcn_perm(__builtin_bit_cast(unsigned, h23), __builtin_bit_cast(unsigned, h01), 0x07050301u);
      }
      if (!valid) { aL = alf_i4{ 0, 0, 0, 0 }; aH = alf_i4{ 0, 0, 0, 0 }; }
      LLa = __builtin_amdgcn_mfma_i32_16x16x64_i8(aL, aL, LLa, 0, 0, 0);
      HHa = __builtin_amdgcn_mfma_i32_16x16x64_i8(aH, aH, HHa, 0, 0, 0);
      HLa = __builtin_amdgcn_mfma_i32_16x16x64_i8(aH, aL, HLa, 0, 0, 0);
      LHa = __builtin_amdgcn_mfma_i32_16x16x64_i8(aL, aH, LHa, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; r++)
  {
    const int row = 4 * g + r, r7 = row & 7;
    const long long v = (long long)HHa[r] * 65536 + (long long)(HLa[r] + LHa[r]) * 256 + (long long)LLa[r];
    int e = -1;
    if ((row >> 3) == pl)
    {
      if (r7 < 7 && cc < 7) e = r7 * 7 + cc;
      else if (r7 == 7 && cc < 7) e = 49 + cc;
      else if (r7 == 7 && cc == 7) e = 56;
    }
    if (e >= 0 && s1 > s0) atomicAdd(&bucket[pl * 57 + e], (unsigned long long)v);
  }
  __syncthreads();
  for (int i = tid; i < 2 * 57; i += ACT) a.outC[i / 57][(size_t)ctuIdx * 57 + (i % 57)] = bucket[i];
}

// one workgroup per CTU: the CTU's chroma pair, then luma in the same LDS (the chroma tiles and records lie inside the luma tile's bytes, which are
// written behind the barrier that ends the chroma part).
// As workgroups of their own the chroma pairs were a second round of 80 KB workgroups behind the luma round: 10 us of a 56 us launch.
template <int C, bool CLS>
__global__ __launch_bounds__(ACT) void alf_stats_picture_kernel(AlfStatsPic a)
{
  extern __shared__ __align__(16) unsigned char alfSmem[];
  AlfChromaPre<C> pre;
  const int ctuIdx = vvc_xcd_index((int)blockIdx.x, a.nCtu, a.xcd);
  if (ctuIdx < 0) return;
  // the chroma pair FIRST, with the luma tile requested in front of it: the luma tile's trip through the start-up burst (every workgroup of the chip
  // fetches at once) is covered by the chroma part's arithmetic instead of by nothing (58.9 -> 58.0 us; luma first with the chroma loads behind the tile: 59.4)
  alf_chroma_prefetch<C>(a, ctuIdx, pre);
  alf_ctu_luma<C, CLS>(a, ctuIdx, alfSmem, [&]() { alf_ctu_chroma<C>(a, ctuIdx, alfSmem, pre); __syncthreads(); });
}

// The 5x5 diamond is the centre of the 7x7 diamond under every transposition, so the 5x5 covariance record of a class is a sub-matrix of its 7x7
// record: coefficient i of the 5x5 filter is coefficient SIG[i] of the 7x7 filter (tap (2,0) -> 2, (1,1) -> 5, (1,0) -> 6, (1,-1) -> 7, (0,2) -> 10,
// (0,1) -> 11, centre -> 12).  One thread per output entry; also zeroes nothing: every entry of out5 is written.
__global__ __launch_bounds__(256) void alf_stats_5from7_kernel(const unsigned long long* __restrict__ out7, unsigned long long* __restrict__ out5, int nRecords)
{
  const int gid = blockIdx.x * 256 + threadIdx.x;
  if (gid >= nRecords * 57) return;
  const int rec = gid / 57, e = gid - rec * 57;
  const int sig[7] = { 2, 5, 6, 7, 10, 11, 12 };
  const unsigned long long* r7 = out7 + (size_t)rec * 183;
  unsigned long long v;
  if (e < 49) v = r7[sig[e / 7] * 13 + sig[e % 7]];
  else if (e < 56) v = r7[169 + sig[e - 49]];
  else v = r7[182];
  out5[gid] = v;
}

__global__ __launch_bounds__(256) void zero3_kernel(unsigned long long* a, size_t na, unsigned long long* b, size_t nb, unsigned long long* c, size_t nc)
{
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
  for (size_t i = gid; i < na; i += stride) a[i] = 0ull;
  for (size_t i = gid; i < nb; i += stride) b[i] = 0ull;
  for (size_t i = gid; i < nc; i += stride) c[i] = 0ull;
}


}  // namespace

extern "C" {

int vvcgpu_sao_stats(const vvc_pel* org, int org_stride, const vvc_pel* rec, int rec_stride,
                     int width, int height, int ctu_w, int ctu_h, int bit_depth, const uint8_t* avail,
                     int skip_lines_r, int skip_lines_b, int64_t* out, void* stream)
{
  VVC_CHECK_ARG(org && rec && out, "sao_stats: null pointer");
  VVC_CHECK_ARG(width > 0 && height > 0 && org_stride >= width && rec_stride >= width, "sao_stats: bad size/stride");
  VVC_CHECK_ARG(ctu_w >= 16 && ctu_w <= SAO_MAX_CTU && (ctu_w & (ctu_w - 1)) == 0 && ctu_h >= 16 && ctu_h <= SAO_MAX_CTU &&
                (ctu_h & (ctu_h - 1)) == 0, "sao_stats: CTU %dx%d must be a power of two in 16..128", ctu_w, ctu_h);
  VVC_CHECK_ARG(bit_depth >= 8 && bit_depth <= 10, "sao_stats: bit depth %d outside 8..10", bit_depth);
  VVC_CHECK_ARG(skip_lines_r >= 0 && skip_lines_b >= 0 && skip_lines_r < 16 && skip_lines_b < 16, "sao_stats: bad skip lines");
  const int wCtu = cdiv(width, ctu_w), hCtu = cdiv(height, ctu_h);
  // the packed body (256 threads, strips of (256 / ctu_w) x 16 rows) where the CTU shape allows it, else the scalar one:
  // groups = nthreads / ctu_w row groups of ctu_h / groups rows, at least 4 rows per thread
  const bool packed = sao_stats_packed_ok(SAO_PK_THREADS, ctu_w, ctu_h);
  int nthreads = packed ? SAO_PK_THREADS : 1024;
  while (!packed && nthreads > 256 && (nthreads / ctu_w) * 4 > ctu_h) nthreads >>= 1;
  const size_t tileBytes = packed ? (((size_t)((SAO_PK_THREADS / ctu_w) * SAO_PK_ROWS + 6) * (ctu_w + 2) * 2) + 15) & ~(size_t)15
                                  : (((size_t)(ctu_h + 2) * (ctu_w + 2) * 2) + 15) & ~(size_t)15;
  const size_t smem = tileBytes + 16 * 32 * 8 + 40 * 4;
  if (smem > 48 * 1024)
    VVC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sao_stats_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL(sao_stats_kernel, dim3(wCtu, hCtu), dim3(nthreads), smem, (hipStream_t)stream, org, org_stride, rec,
                     rec_stride, width, height, ctu_w, ctu_h, wCtu, bit_depth - 5, avail, skip_lines_r, skip_lines_b,
                     reinterpret_cast<long long*>(out));
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}

int vvcgpu_alf_stats(const vvc_pel* org, int org_stride, const vvc_pel* rec, int rec_stride,
                     int width, int height, int ctu_size, const uint16_t* cls, int filter_type,
                     int64_t* out, void* stream)
{
  VVC_CHECK_ARG(org && rec && out, "alf_stats: null pointer");
  VVC_CHECK_ARG(width > 0 && height > 0 && (width & 3) == 0 && (height & 3) == 0, "alf_stats: size must be a multiple of 4");
  VVC_CHECK_ARG(org_stride >= width && rec_stride >= width && (org_stride & 3) == 0 && ((uintptr_t)org & 7) == 0,
                "alf_stats: org needs stride %% 4 == 0 and 8-byte alignment");
  VVC_CHECK_ARG(ctu_size >= AT ? (ctu_size % AT) == 0 : false, "alf_stats: ctu size %d must be a multiple of %d", ctu_size, AT);
  VVC_CHECK_ARG(filter_type == 0 || filter_type == 1, "alf_stats: filter_type %d", filter_type);
  const int nCls = cls ? 25 : 1;
  const int N = filter_type ? 13 : 7;
  const int wCtu = cdiv(width, ctu_size), hCtu = cdiv(height, ctu_size);
  hipStream_t st = (hipStream_t)stream;
  VVC_HIP(hipMemsetAsync(out, 0, sizeof(int64_t) * (size_t)(N * N + N + 1) * nCls * wCtu * hCtu, st));
  dim3 grid(cdiv(width, AT), cdiv(height, AT));
  if (filter_type)
    hipLaunchKernelGGL(alf_stats_kernel<true>, grid, dim3(256), 0, st, org, org_stride, rec, rec_stride, width, height,
                       ctu_size, wCtu, cls, nCls, reinterpret_cast<unsigned long long*>(out));
  else
    hipLaunchKernelGGL(alf_stats_kernel<false>, grid, dim3(256), 0, st, org, org_stride, rec, rec_stride, width, height,
                       ctu_size, wCtu, cls, nCls, reinterpret_cast<unsigned long long*>(out));
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}

int vvcgpu_sao_stats_picture(const vvcgpu_planes* org, const vvcgpu_planes* rec, int width, int height, int ctu_size, int bit_depth,
                             const uint8_t* avail, int skip_r_luma, int skip_b_luma, int skip_r_chroma, int skip_b_chroma,
                             int64_t* out_y, int64_t* out_cb, int64_t* out_cr, void* stream)
{
  VVC_CHECK_ARG(org && rec && out_y && out_cb && out_cr, "sao_stats_picture: null pointer");
  VVC_CHECK_ARG(width > 0 && height > 0 && (width & 1) == 0 && (height & 1) == 0, "sao_stats_picture: bad size %dx%d", width, height);
  VVC_CHECK_ARG(ctu_size >= 32 && ctu_size <= SAO_MAX_CTU && (ctu_size & (ctu_size - 1)) == 0, "sao_stats_picture: CTU size %d must be 32, 64 or 128", ctu_size);
  VVC_CHECK_ARG(bit_depth >= 8 && bit_depth <= 10, "sao_stats_picture: bit depth %d outside 8..10", bit_depth);
  VVC_CHECK_ARG(skip_r_luma >= 0 && skip_b_luma >= 0 && skip_r_luma < 16 && skip_b_luma < 16 && skip_r_chroma >= 0 && skip_b_chroma >= 0 &&
                skip_r_chroma < 16 && skip_b_chroma < 16, "sao_stats_picture: bad skip lines");
  SaoStats3 p;
  int64_t* outs[3] = { out_y, out_cb, out_cr };
  int end = 0;
  for (int c = 0; c < 3; c++)
  {
    const int w = c ? width >> 1 : width, h = c ? height >> 1 : height, ctu = c ? ctu_size >> 1 : ctu_size;
    VVC_CHECK_ARG(org->p[c] && rec->p[c] && org->stride[c] >= w && rec->stride[c] >= w, "sao_stats_picture: plane %d", c);
    const int wCtu = cdiv(w, ctu), hCtu = cdiv(h, ctu);
    end += wCtu * hCtu;
    p.a[c] = SaoStatsPlane{ org->p[c], rec->p[c], reinterpret_cast<long long*>(outs[c]), org->stride[c], rec->stride[c], w, h, ctu, wCtu,
                            c ? skip_r_chroma : skip_r_luma, c ? skip_b_chroma : skip_b_luma, end };
  }
  p.avail = avail; p.boShift = bit_depth - 5;
  // workgroup size: 256 threads where the luma CTU takes the packed body (the chroma CTU then takes it as well, or the scalar body with
  // ctu / 2 / 8 rows per thread); else the chroma CTU (ctu / 2 wide) sets the row groups and every thread keeps at least two rows
  const bool packed = sao_stats_packed_ok(SAO_PK_THREADS, ctu_size, ctu_size);
  int nthreads = packed ? SAO_PK_THREADS : 1024;
  while (!packed && nthreads > 256 && (nthreads / (ctu_size >> 1)) * 2 > (ctu_size >> 1)) nthreads >>= 1;
  size_t tileBytes = 0;
  for (int c = 0; c < 2; c++)
  {
    const int ctu = ctu_size >> c;
    const size_t t = sao_stats_packed_ok(nthreads, ctu, ctu) ? (size_t)((SAO_PK_THREADS / ctu) * SAO_PK_ROWS + 6) * (ctu + 2) * 2 : (size_t)(ctu + 2) * (ctu + 2) * 2;
    tileBytes = t > tileBytes ? t : tileBytes;
  }
  tileBytes = (tileBytes + 15) & ~(size_t)15;
  const size_t smem = tileBytes + 16 * 32 * 8 + 40 * 4;
  if (smem > 48 * 1024)
    VVC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sao_stats_picture_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  p.total = end; p.xcd = vvc_xcd_on();
  hipLaunchKernelGGL(sao_stats_picture_kernel, dim3(vvc_xcd_grid2(p.a[0].wgEnd, end, p.xcd)), dim3(nthreads), smem, (hipStream_t)stream, p);
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}

}  // extern "C"

// cls_out != nullptr: the fused form -- the classes are derived inside the CTU workgroups and written to cls_out (cls is not read)
static int alf_stats_picture_impl(const vvcgpu_planes* org, const vvcgpu_planes* rec, int width, int height, int ctu_size, const uint16_t* cls,
                                  uint16_t* cls_out, int bit_depth, int64_t* out7, int64_t* out5, int64_t* out_cb, int64_t* out_cr, void* stream)
{
  VVC_CHECK_ARG(org && rec && (cls || cls_out) && out7 && out5 && out_cb && out_cr, "alf_stats_picture: null pointer");
  VVC_CHECK_ARG(width > 0 && height > 0 && (width & 7) == 0 && (height & 7) == 0, "alf_stats_picture: size must be a multiple of 8");
  VVC_CHECK_ARG(ctu_size == 64 || (ctu_size >= 2 * AT && (ctu_size % (2 * AT)) == 0), "alf_stats_picture: ctu size %d must be 64 or a multiple of %d", ctu_size, 2 * AT);
  for (int c = 0; c < 3; c++)
  {
    const int w = c ? width >> 1 : width;
    VVC_CHECK_ARG(org->p[c] && rec->p[c] && org->stride[c] >= w && rec->stride[c] >= w && (org->stride[c] & 3) == 0 && ((uintptr_t)org->p[c] & 7) == 0,
                  "alf_stats_picture: plane %d (org needs stride %% 4 == 0 and 8-byte alignment)", c);
  }
  VVC_CHECK_ARG(org->stride[1] == org->stride[2] && rec->stride[1] == rec->stride[2], "alf_stats_picture: Cb and Cr strides must match");
  hipStream_t st = (hipStream_t)stream;
  const int wCtu = cdiv(width, ctu_size), hCtu = cdiv(height, ctu_size), nCtu = wCtu * hCtu;
  unsigned long long* o7 = reinterpret_cast<unsigned long long*>(out7);
  unsigned long long* ocb = reinterpret_cast<unsigned long long*>(out_cb);
  unsigned long long* ocr = reinterpret_cast<unsigned long long*>(out_cr);
  // The classifier inside the CTU workgroups lengthens every workgroup's chain of phases by ~4 us and saves the classifier's launch (~9.5 us at 4K, ~3 us
  // at 1080p): it pays when the CTUs fill the machine about twice (510 CTUs at 4K: 67.5 -> 59.4 us), not for a 1080p picture's 135 CTUs (54.6 -> 61.4 us)
  if (cls_out && nCtu < 320)
  {
    const int rt = vvcgpu_alf_classify(rec->p[0], rec->stride[0], width, height, bit_depth, cls_out, stream);
    if (rt) return rt;
    cls = cls_out; cls_out = nullptr;
  }
  if (ctu_size == 128 || ctu_size == 64)
  {
    // CTU form: one launch, every record written by the workgroup that owns the CTU
    AlfStatsPic a;
    for (int c = 0; c < 3; c++) { a.org[c] = org->p[c]; a.rec[c] = rec->p[c]; a.ostride[c] = org->stride[c]; a.rstride[c] = rec->stride[c]; }
    a.w = width; a.h = height; a.wCtu = wCtu; a.nCtu = nCtu; a.cls = cls; a.clsOut = cls_out; a.clsShift = bit_depth + 4; a.xcd = vvc_xcd_on();
    a.out7 = o7; a.out5 = reinterpret_cast<unsigned long long*>(out5); a.outC[0] = ocb; a.outC[1] = ocr;
    auto launch = [&](auto kernel, int ldsBytes) -> int
    {
      VVC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, ldsBytes));
      hipLaunchKernelGGL(kernel, dim3(vvc_xcd_grid(nCtu, a.xcd)), dim3(ACT), ldsBytes, st, a);
      return VVCGPU_OK;
    };
    int rt;
    if (ctu_size == 128) rt = cls_out ? launch(alf_stats_picture_kernel<128, true>, AlfCtuLds<128>::bytes) : launch(alf_stats_picture_kernel<128, false>, AlfCtuLds<128>::bytes);
    else                 rt = cls_out ? launch(alf_stats_picture_kernel<64, true>, AlfCtuLds<64>::bytes) : launch(alf_stats_picture_kernel<64, false>, AlfCtuLds<64>::bytes);
    if (rt) return rt;
    VVC_LAUNCH_CHECK();
    return VVCGPU_OK;
  }
  if (cls_out)                                                              // other CTU sizes: the classifier's own launch in front of the tile form
  {
    const int rt = vvcgpu_alf_classify(rec->p[0], rec->stride[0], width, height, bit_depth, cls_out, stream);
    if (rt) return rt;
    cls = cls_out;
  }
  hipLaunchKernelGGL(zero3_kernel, dim3(512), dim3(256), 0, st, o7, (size_t)nCtu * 25 * 183, ocb, (size_t)nCtu * 57, ocr, (size_t)nCtu * 57);
  hipLaunchKernelGGL(alf_stats_kernel<true>, dim3(cdiv(width, AT), cdiv(height, AT)), dim3(256), 0, st, org->p[0], org->stride[0], rec->p[0], rec->stride[0],
                     width, height, ctu_size, wCtu, cls, 25, o7);
  hipLaunchKernelGGL(alf_stats_chroma2_kernel, dim3(cdiv(width >> 1, AT), cdiv(height >> 1, AT), 2), dim3(256), 0, st, org->p[1], org->p[2], org->stride[1],
                     rec->p[1], rec->p[2], rec->stride[1], width >> 1, height >> 1, ctu_size >> 1, wCtu, ocb, ocr);
  hipLaunchKernelGGL(alf_stats_5from7_kernel, dim3(cdiv(nCtu * 25 * 57, 256)), dim3(256), 0, st, o7, reinterpret_cast<unsigned long long*>(out5), nCtu * 25);
  VVC_LAUNCH_CHECK();
  return VVCGPU_OK;
}

extern "C" {

int vvcgpu_alf_stats_picture(const vvcgpu_planes* org, const vvcgpu_planes* rec, int width, int height, int ctu_size, const uint16_t* cls,
                             int64_t* out7, int64_t* out5, int64_t* out_cb, int64_t* out_cr, void* stream)
{
  VVC_CHECK_ARG(cls, "alf_stats_picture: null pointer");
  return alf_stats_picture_impl(org, rec, width, height, ctu_size, cls, nullptr, 10, out7, out5, out_cb, out_cr, stream);
}

int vvcgpu_alf_classify_stats_picture(const vvcgpu_planes* org, const vvcgpu_planes* rec, int width, int height, int ctu_size, int bit_depth,
                                      uint16_t* cls_out, int64_t* out7, int64_t* out5, int64_t* out_cb, int64_t* out_cr, void* stream)
{
  VVC_CHECK_ARG(cls_out, "alf_classify_stats_picture: null pointer");
  VVC_CHECK_ARG(bit_depth >= 8 && bit_depth <= 10, "alf_classify_stats_picture: bit depth %d outside 8..10", bit_depth);
  return alf_stats_picture_impl(org, rec, width, height, ctu_size, nullptr, cls_out, bit_depth, out7, out5, out_cb, out_cr, stream);
}

}  // extern "C"
