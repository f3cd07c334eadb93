// raster_dev.h -- device pieces shared by the step-5 raster search kernels of dist.hip and the hierarchical search of mehier.hip:
// wave minima, the LDS window fill, and the QUAD SAD loop (a lane owns four consecutive raster columns; the org rows are wave-uniform scalar
// operands from a packed copy of the block).  See the kernel comments in dist.hip for the measurements behind these forms.
#pragma once
#include "common.h"

namespace {

__device__ __forceinline__ unsigned expgolomb_bits(int v)        // RdCost.h:172-184
{
  unsigned len = 1, t = (v <= 0) ? ((unsigned)(-v) << 1) + 1 : (unsigned)(v << 1);
  while (t > 128u) { len += 14; t >>= 7; }
  return len + ((31 - __clz((int)t)) << 1);
}

// minimum of a 32-bit value over the wavefront with DPP row operations (no LDS traffic, 6 VALU); the result is wave-uniform
__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
#define WMIN_STEP(CTRL, ROWMASK) v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, CTRL, ROWMASK, 0xF, false))
  WMIN_STEP(0xB1, 0xF);      // quad_perm [1,0,3,2]
  WMIN_STEP(0x4E, 0xF);      // quad_perm [2,3,0,1]
  WMIN_STEP(0x141, 0xF);     // row_half_mirror
  WMIN_STEP(0x140, 0xF);     // row_mirror: every lane of a 16-lane row holds the row's minimum
  WMIN_STEP(0x142, 0xA);     // row_bcast15 into rows 1 and 3
  WMIN_STEP(0x143, 0xC);     // row_bcast31 into rows 2 and 3: lane 63 holds the minimum of all four rows
#undef WMIN_STEP
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// 64-bit minimum as two 32-bit passes: the high words first, then the low words of the lanes that hold the minimal high word
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long k)
{
  const unsigned hi = (unsigned)(k >> 32), lo = (unsigned)k;
  const unsigned hmin = wave_min_u32(hi);
  const unsigned lmin = wave_min_u32(hi == hmin ? lo : 0xFFFFFFFFu);
  return ((unsigned long long)hmin << 32) | lmin;
}

#define R5C_COST_N 132                                            /* expgolomb_bits <= 65 per component */
#define R5C_WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)       /* lgkmcnt(0), vmcnt/expcnt untouched */


// window fill for r5c: thread = (row r0, quad q), walking down the rows with a constant stride so that the addressing per
// 16-byte load is one add for the global offset and one for the LDS index; FB loads in flight.
template <int FB>
__device__ __forceinline__ void fill_window_cols(unsigned* __restrict__ lds, const uint4* __restrict__ g, int rsQ, int winRows,
                                                 int pitchDw, int nQuads, int tid, int nthreads)
{
  const int r0 = (int)(((float)tid + 0.5f) * __frcp_rn((float)nQuads));   // tid / nQuads (tid < 1024, nQuads < 1024: exact)
  const int q = tid - r0 * nQuads;
  const int R = nthreads / nQuads;                                         // rows per pass; threads beyond R * nQuads idle
  if (r0 >= R) return;
  unsigned goff = (unsigned)(r0 * rsQ + q);
  unsigned loff = (unsigned)(r0 * pitchDw + 4 * q);
  const unsigned gstep = (unsigned)(R * rsQ), lstep = (unsigned)(R * pitchDw);
  for (int r = r0; r < winRows; r += FB * R)
  {
    uint4 v[FB];
#pragma unroll
    for (int u = 0; u < FB; u++)
      if (r + u * R < winRows) v[u] = g[goff + u * gstep];
#pragma unroll
    for (int u = 0; u < FB; u++)
      if (r + u * R < winRows)
      {
        uint2* d = reinterpret_cast<uint2*>(lds + loff + u * lstep);       // pitch is even: 8-byte aligned
        d[0] = make_uint2(v[u].x ^ 0x80008000u, v[u].y ^ 0x80008000u);
        d[1] = make_uint2(v[u].z ^ 0x80008000u, v[u].w ^ 0x80008000u);
      }
    goff += FB * gstep; loff += FB * lstep;
  }
}


// ---- QUAD form of the SAD loop (sad_raster5q_kernel, sad_raster5gq_kernel, me_hier_kernel) ----
struct R5qStage { unsigned ovE[8], ovO[8]; unsigned long long d[8]; unsigned x1; };

template <int OA>
__device__ __forceinline__ void r5q_issue(R5qStage& st, const unsigned* __restrict__ op, unsigned a)
{
#pragma unroll
  for (int k = 0; k < 8; k++) { st.ovE[k] = op[k]; st.ovO[k] = op[8 + k]; }        // wave-uniform: one 64-byte scalar load
  // words 0..7 of the span (dwords 0..15); OA >= 2 also needs dword 16.  Single ds_read_b64 on purpose (see r5c_issue_row).
  if (OA < 2)
  {
    unsigned dummy;
    asm volatile("ds_read_b64 %0, %9\n\tds_read_b64 %1, %9 offset:8\n\tds_read_b64 %2, %9 offset:16\n\tds_read_b64 %3, %9 offset:24\n\t"
                 "ds_read_b64 %4, %9 offset:32\n\tds_read_b64 %5, %9 offset:40\n\tds_read_b64 %6, %9 offset:48\n\tds_read_b64 %7, %9 offset:56"
                 : "=&v"(st.d[0]), "=&v"(st.d[1]), "=&v"(st.d[2]), "=&v"(st.d[3]), "=&v"(st.d[4]), "=&v"(st.d[5]), "=&v"(st.d[6]), "=&v"(st.d[7]), "=&v"(dummy)
                 : "v"(a) : "memory");
  }
  else
    asm volatile("ds_read_b64 %0, %9\n\tds_read_b64 %1, %9 offset:8\n\tds_read_b64 %2, %9 offset:16\n\tds_read_b64 %3, %9 offset:24\n\t"
                 "ds_read_b64 %4, %9 offset:32\n\tds_read_b64 %5, %9 offset:40\n\tds_read_b64 %6, %9 offset:48\n\tds_read_b64 %7, %9 offset:56\n\t"
                 "ds_read_b32 %8, %9 offset:64"
                 : "=&v"(st.d[0]), "=&v"(st.d[1]), "=&v"(st.d[2]), "=&v"(st.d[3]), "=&v"(st.d[4]), "=&v"(st.d[5]), "=&v"(st.d[6]), "=&v"(st.d[7]), "=&v"(st.x1)
                 : "v"(a) : "memory");
}

// position m of the lane starts OA + STEP m samples into the span: dword I = (OA + STEP m) >> 1, parity P = (OA + STEP m) & 1.
// STEP = 5: four columns of the step-5 raster; STEP = 1: four neighbouring columns of a dense (+-4) grid (mehier.hip).
template <int OA, int STEP = 5>
__device__ __forceinline__ void r5q_compute(const R5qStage& st, unsigned (&acc)[4])
{
  unsigned dd[17];
#pragma unroll
  for (int k = 0; k < 8; k++)
  {
    asm volatile("" :: "v"(st.d[k]));                     // whole 64-bit destination stays allocated until here
    dd[2 * k] = (unsigned)st.d[k]; dd[2 * k + 1] = (unsigned)(st.d[k] >> 32);
  }
  if (OA >= 2) { asm volatile("" :: "v"(st.x1)); dd[16] = st.x1; } else dd[16] = 0u;
  // k outer, candidate inner: four independent accumulator chains in flight instead of one 8-deep dependent chain after the other
  // (tools/micro/sadloop_rate.hip: 73.8 vs 83.5 ns per stage at 4 waves per SIMD)
#pragma unroll
  for (int k = 0; k < 8; k++)
  {
#pragma unroll
    for (int m = 0; m < 4; m++)
    {
      const int s = OA + STEP * m, I = s >> 1;
      if (s & 1)
      {
        if (k < 7) acc[m] = __builtin_amdgcn_sad_u16(st.ovO[k], dd[I + 1 + k], acc[m]);
        else       acc[m] = __builtin_amdgcn_sad_u16(st.ovO[7], (dd[I + 8] & 0xFFFFu) | (dd[I] & 0xFFFF0000u), acc[m]);
      }
      else
        acc[m] = __builtin_amdgcn_sad_u16(st.ovE[k], dd[I + k], acc[m]);
    }
  }
}

// walks nStages chunk-rows starting at chunk-row cr0 of the block (CH chunks per row)
template <int OA, int STEP = 5>
__device__ __forceinline__ void r5q_positions(const unsigned* __restrict__ orgQ, unsigned base, int ldsStep, int CH,
                                              int cr0, int nStages, unsigned (&acc)[4])
{
  R5qStage A, B;
  const int chShift = 31 - __clz(CH);
  int ch = cr0 & (CH - 1);
  unsigned oOff = (unsigned)cr0 * 16u;
  unsigned lOff = (unsigned)((cr0 >> chShift) * ldsStep + ch * 8) * 4u;
  const unsigned rowAdv = (unsigned)(ldsStep - 8 * (CH - 1)) * 4u;          // from the last chunk of a row to the first of the next sampled row
  auto issue = [&](R5qStage& st)
  {
    r5q_issue<OA>(st, orgQ + oOff, base + lOff);
    oOff += 16u; ch++;
    if (ch == CH) { ch = 0; lOff += rowAdv; } else lOff += 32u;
  };
  issue(A);
  for (int s = 0; s < nStages; s += 2)
  {
    R5C_WAIT_LGKM0();
    if (s + 1 < nStages) issue(B);
    __builtin_amdgcn_sched_barrier(0);
    r5q_compute<OA, STEP>(A, acc);
    if (s + 1 >= nStages) break;
    R5C_WAIT_LGKM0();
    if (s + 2 < nStages) issue(A);
    __builtin_amdgcn_sched_barrier(0);
    r5q_compute<OA, STEP>(B, acc);
  }
}

// The same walk with everything known at compile time (me_hier_kernel: one chunk per row, NST sampled rows LSTEP dwords apart): the stage loop
// unrolls, the window reads of stage s are ONE base register + immediate offsets (s * LSTEP * 4 + 8 k < 2^16), the original rows of stage s one scalar
// load at an immediate offset from the block's base -- no per-stage address arithmetic, compares or branches on the scalar unit (VERDICT r5 item 7 ii:
// the loop form spent ~14 scalar instructions per stage of 35 vector instructions, on the CU's one scalar unit shared by sixteen waves).
// The original rows as an EXPLICIT scalar load (one s_load_dwordx16 into sixteen scalar registers; "memory": not moved across barriers).  me_hier_kernel
// packs the original rows of a super-block itself and reads them back through the scalar cache: compiler-generated loads of a buffer the kernel also
// writes become vector loads (every lane the same address: 286 us instead of 202, round 5), and through a read-only view they could be moved above
// the packing stores.  The caller waits (lgkmcnt(0)) before the first use, as for the window reads.
typedef unsigned r5q_u16v __attribute__((ext_vector_type(16)));
struct R5qStageS { r5q_u16v ov; unsigned long long d[8]; unsigned x1; };          // ov[0..7]: even pairs, ov[8..15]: odd-shifted pairs of the original row

template <int OA, int OFF, int OOFF = 0>                                         // OFF: byte offset of the window reads, OOFF: of the original row
__device__ __forceinline__ void r5q_issue_at(R5qStageS& st, const unsigned* op, unsigned a)
{
  asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(st.ov) : "s"(op), "n"(OOFF) : "memory");
  if (OA < 2)
  {
    unsigned dummy;
    asm volatile("ds_read_b64 %0, %9 offset:%10\n\tds_read_b64 %1, %9 offset:%11\n\tds_read_b64 %2, %9 offset:%12\n\tds_read_b64 %3, %9 offset:%13\n\t"
                 "ds_read_b64 %4, %9 offset:%14\n\tds_read_b64 %5, %9 offset:%15\n\tds_read_b64 %6, %9 offset:%16\n\tds_read_b64 %7, %9 offset:%17"
                 : "=&v"(st.d[0]), "=&v"(st.d[1]), "=&v"(st.d[2]), "=&v"(st.d[3]), "=&v"(st.d[4]), "=&v"(st.d[5]), "=&v"(st.d[6]), "=&v"(st.d[7]), "=&v"(dummy)
                 : "v"(a), "i"(OFF), "i"(OFF + 8), "i"(OFF + 16), "i"(OFF + 24), "i"(OFF + 32), "i"(OFF + 40), "i"(OFF + 48), "i"(OFF + 56) : "memory");
  }
  else
    asm volatile("ds_read_b64 %0, %9 offset:%10\n\tds_read_b64 %1, %9 offset:%11\n\tds_read_b64 %2, %9 offset:%12\n\tds_read_b64 %3, %9 offset:%13\n\t"
                 "ds_read_b64 %4, %9 offset:%14\n\tds_read_b64 %5, %9 offset:%15\n\tds_read_b64 %6, %9 offset:%16\n\tds_read_b64 %7, %9 offset:%17\n\t"
                 "ds_read_b32 %8, %9 offset:%18"
                 : "=&v"(st.d[0]), "=&v"(st.d[1]), "=&v"(st.d[2]), "=&v"(st.d[3]), "=&v"(st.d[4]), "=&v"(st.d[5]), "=&v"(st.d[6]), "=&v"(st.d[7]), "=&v"(st.x1)
                 : "v"(a), "i"(OFF), "i"(OFF + 8), "i"(OFF + 16), "i"(OFF + 24), "i"(OFF + 32), "i"(OFF + 40), "i"(OFF + 48), "i"(OFF + 56), "i"(OFF + 64) : "memory");
}
// r5q_compute on the explicit stage (the same 32 sums; STEP = 5)
template <int OA>
__device__ __forceinline__ void r5q_compute_s(const R5qStageS& st, unsigned (&acc)[4])
{
  unsigned dd[17];
#pragma unroll
  for (int k = 0; k < 8; k++)
  {
    asm volatile("" :: "v"(st.d[k]));
    dd[2 * k] = (unsigned)st.d[k]; dd[2 * k + 1] = (unsigned)(st.d[k] >> 32);
  }
  if (OA >= 2) { asm volatile("" :: "v"(st.x1)); dd[16] = st.x1; } else dd[16] = 0u;
#pragma unroll
  for (int k = 0; k < 8; k++)
  {
#pragma unroll
    for (int m = 0; m < 4; m++)
    {
      const int s = OA + 5 * m, I = s >> 1;
      if (s & 1)
      {
        if (k < 7) acc[m] = __builtin_amdgcn_sad_u16(st.ov[8 + k], dd[I + 1 + k], acc[m]);
        else       acc[m] = __builtin_amdgcn_sad_u16(st.ov[15], (dd[I + 8] & 0xFFFFu) | (dd[I] & 0xFFFF0000u), acc[m]);
      }
      else
        acc[m] = __builtin_amdgcn_sad_u16(st.ov[k], dd[I + k], acc[m]);
    }
  }
}

template <int OA, int NST, int LSTEP, int S>
struct R5qFixed
{
  static __device__ __forceinline__ void run(const unsigned* orgQ, unsigned base, const unsigned* nextOrgQ, unsigned nextBase,
                                             R5qStageS& A, R5qStageS& B, unsigned (&acc)[4])
  {
    // stage S is in flight in A (even S) / B (odd S); behind the last stage: stage 0 of the walk that follows (NST is even: into A again)
    R5C_WAIT_LGKM0();
    if (S + 1 < NST) r5q_issue_at<OA, (S + 1) * LSTEP * 4, (S + 1) * 64>((S & 1) ? A : B, orgQ, base);
    else r5q_issue_at<OA, 0>(A, nextOrgQ, nextBase);                       // (unconditional: behind the last sub-block the caller names a valid one again)
    __builtin_amdgcn_sched_barrier(0);
    r5q_compute_s<OA>((S & 1) ? B : A, acc);
    R5qFixed<OA, NST, LSTEP, S + 1>::run(orgQ, base, nextOrgQ, nextBase, A, B, acc);
  }
};
template <int OA, int NST, int LSTEP>
struct R5qFixed<OA, NST, LSTEP, NST>
{
  static __device__ __forceinline__ void run(const unsigned*, unsigned, const unsigned*, unsigned, R5qStageS&, R5qStageS&, unsigned (&)[4]) {}
};
// stage 0 of (orgQ, base) is in flight in A on entry; on exit stage 0 of (nextOrgQ, nextBase) is
template <int OA, int NST, int LSTEP>
__device__ __forceinline__ void r5q_positions_fixed(const unsigned* orgQ, unsigned base, const unsigned* nextOrgQ, unsigned nextBase,
                                                    R5qStageS& A, R5qStageS& B, unsigned (&acc)[4])
{
  static_assert((NST - 1) * LSTEP * 4 + 64 < 65536 && (NST & 1) == 0, "ds offset field; stage 0 lives in A");
  R5qFixed<OA, NST, LSTEP, 0>::run(orgQ, base, nextOrgQ, nextBase, A, B, acc);
}

}  // namespace
