// raster7.hip -- step-5 raster search of xTZSearch (InterSearch.cpp:2159-2169, distortion through xTZSearchHelp :249-343, the
// SAD of RdCost.cpp:466-492 with row sub-sampling) for lists of blocks, RING form ("r7").  Same results as the raster kernels of dist.hip:
// arg-min of  sad + uint64(lambda * bits)  in scan order with strict '<' (InterSearch.cpp:1913-1925, RdCost.h:172-199), optional surface.
//
// Why another form.  profiles/r02_raster_parts.txt + r02_pmc_sq.csv: the strip kernels execute 1.8x the v_sad_u16 the arithmetic needs
// (window staging with vector instructions, 42 x 40 x 64/60 position slots for 39 x 39 positions, per-item set-up and arg-min), and a
// workgroup's staging does not overlap its own SAD loop (32-wide blocks: ONE 131 KB workgroup per CU, fill -> barrier -> loop).  Here:
//   * ONE persistent 1024-thread workgroup per CU walks a CHAIN of vertically adjacent steps (a step = up to four blocks that are
//     horizontal neighbours in the reference picture x one slab of <= 32 block rows).  The search windows of consecutive steps of a chain
//     overlap in all but SH rows, so the window lives in a RING of RR = winRows + SH rows in LDS and a step brings in only the SH new
//     rows of the NEXT step -- by LDS-DMA (global_load_lds_dwordx4: no vector registers, no vector instructions), issued before the
//     step's SAD loop and waited for after it.  Window traffic per 32 x 32 block: 19 KB instead of 131 KB, none of it exposed.
//   * No bias: v_sad_u16 needs unsigned operands.  Reference samples are picture samples (>= 0: precondition of this form); the original
//     block may hold any int16 (bi-predictive searches use 2 org - pred), and for r >= 0:  |o - r| = |max(o, 0) - r| + max(-o, 0).  The
//     packing kernel clamps the original at 0 and keeps the sum of the negative parts per block, a constant that is added to the winner's
//     cost afterwards (the arg-min and its tie order do not see it) -- so the DMA lands raw samples and nothing touches the window.
//   * Position slots are flattened: a lane owns four consecutive raster columns of one raster row (the quad form of dist.hip: 16 - 17
//     dwords of LDS serve 32 v_sad_u16), unit u = row * nq + quad; 64 consecutive units are a GROUP.  With the row pitch == 20 (mod 64)
//     dwords the 8-byte slot of unit u is 5 u (mod 32), so ANY 32 consecutive units read distinct slots: no dead lanes, no row padding.
//   * The (block, group, chunk-row) space of a step is cut into 16 equal runs, one per wave; partial sums meet in an LDS surface
//     (ds_add_u32), one pass over the surface yields the arg-min per block and the workgroup writes vvcgpu_search_best itself: no key
//     initialisation, no global atomics, no decode launch.
// A list that is not a regular grid is still served exactly (chains are verified step by step; a step whose blocks are not neighbours is
// served block by block, each with a full window fill).
#include "common.h"

namespace {

constexpr int R7_P = 592;                    // ring row pitch in bytes: 148 dwords == 20 (mod 64)
constexpr int R7_PIECES = R7_P / 16;         // 16-byte DMA pieces per ring row
constexpr int R7_THREADS = 1024;
constexpr int R7_WAVES = R7_THREADS / 64;
constexpr int R7_MAXNB = 4;
constexpr int R7_COST_N = 132;               // expgolomb_bits <= 65 per component
#define R7_WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)       /* lgkmcnt(0), vmcnt/expcnt untouched */

__device__ __forceinline__ unsigned r7_expgolomb_bits(int v)        // RdCost.h:172-184
{
  unsigned len = 1, t = (v <= 0) ? ((unsigned)(-v) << 1) + 1 : (unsigned)(v << 1);
  while (t > 128u) { len += 14; t >>= 7; }
  return len + ((31 - __clz((int)t)) << 1);
}

__device__ __forceinline__ unsigned r7_wave_min_u32(unsigned v)
{
#define WMIN_STEP(CTRL, ROWMASK) v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, CTRL, ROWMASK, 0xF, false))
  WMIN_STEP(0xB1, 0xF);      // quad_perm [1,0,3,2]
  WMIN_STEP(0x4E, 0xF);      // quad_perm [2,3,0,1]
  WMIN_STEP(0x141, 0xF);     // row_half_mirror
  WMIN_STEP(0x140, 0xF);     // row_mirror
  WMIN_STEP(0x142, 0xA);     // row_bcast15 into rows 1 and 3
  WMIN_STEP(0x143, 0xC);     // row_bcast31 into rows 2 and 3
#undef WMIN_STEP
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned long long r7_wave_min_u64(unsigned long long k)
{
  const unsigned hi = (unsigned)(k >> 32), lo = (unsigned)k;
  const unsigned hmin = r7_wave_min_u32(hi);
  const unsigned lmin = r7_wave_min_u32(hi == hmin ? lo : 0xFFFFFFFFu);
  return ((unsigned long long)hmin << 32) | lmin;
}
__device__ __forceinline__ unsigned r7_wave_sum_u32(unsigned v)        // DPP row operations as in the minimum: the total ends in lane 63
{
#define WSUM_STEP(CTRL, ROWMASK) v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xF, false)
  WSUM_STEP(0xB1, 0xF); WSUM_STEP(0x4E, 0xF); WSUM_STEP(0x141, 0xF); WSUM_STEP(0x140, 0xF); WSUM_STEP(0x142, 0xA); WSUM_STEP(0x143, 0xC);
#undef WSUM_STEP
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// x / d by multiplication: M = ceil(2^32 / d) (exact for x < 2^32 / d), M = 0 stands for d = 1 (2^32 does not fit)
__device__ __forceinline__ int r7_div(int x, unsigned M) { return M ? (int)__umulhi((unsigned)x, M) : x; }

// The four sums of a lane into the surface.  Inline asm on purpose: behind an LDS-DMA the compiler puts s_waitcnt vmcnt(0) in front of every
// LDS access it knows of (the DMA might write there), which would make a wave wait for its rows of the NEXT step at its first flush.  The
// adds are complete at the step's barrier (explicit lgkmcnt(0) in front of it).
__device__ __forceinline__ void r7_flush4(unsigned* sp, const unsigned (&acc)[4])
{
  const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)sp;
  asm volatile("ds_add_u32 %0, %1\n\tds_add_u32 %0, %2 offset:4\n\tds_add_u32 %0, %3 offset:8\n\tds_add_u32 %0, %4 offset:12"
               :: "v"(a), "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]) : "memory");
}

typedef __attribute__((address_space(3))) void* r7_lds_ptr;
typedef const __attribute__((address_space(1))) void* r7_gbl_ptr;
__device__ __forceinline__ void r7_glds16(const void* g, void* l)      // 16 bytes per lane: global (per-lane address) -> LDS (uniform base + 16 * lane id)
{
  __builtin_amdgcn_global_load_lds((r7_gbl_ptr)g, (r7_lds_ptr)l, 16, 0, 0);
}

// ---- packing of the original blocks: [block][chunk-row][even 8 | odd 8] dwords, clamped at 0, row sub-sampling and odd origins
// resolved (layouts: r7_compute); negBlk[block] = sum of max(-o, 0) over the block's sub-sampled samples.  One thread per chunk-row; a block's
// perBlk chunk-rows (a power of two, <= the workgroup size) are consecutive threads of one workgroup, so the sum needs no atomics.
__global__ __launch_bounds__(512) void r7_pack_org_kernel(const Pel* __restrict__ org, int os, const vvcgpu_search_blk* __restrict__ blocks,
                                                          int nblocks, int w, int hs, int subShift, unsigned* __restrict__ packed,
                                                          unsigned* __restrict__ negBlk)
{
  __shared__ unsigned wsum[8];
  const int CH = w >> 4, perBlockUnits = hs * CH;
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = gid < (size_t)nblocks * perBlockUnits;
  const int b = live ? (int)(gid / (unsigned)perBlockUnits) : 0, rem = live ? (int)(gid - (size_t)b * perBlockUnits) : 0;
  unsigned neg = 0;
  if (live)
  {
    const int row = rem / CH, chunk = rem - row * CH;
    const vvcgpu_search_blk blk = blocks[b];
    const Pel* o = org + (size_t)(blk.org_y + (row << subShift)) * os + blk.org_x + 16 * chunk;
    unsigned d[8];
    if ((reinterpret_cast<uintptr_t>(o) & 3) == 0)
    {
      const unsigned* q = reinterpret_cast<const unsigned*>(o);
#pragma unroll
      for (int k = 0; k < 8; k++) d[k] = q[k];
    }
    else
    {
      unsigned short sm[16];
#pragma unroll
      for (int k = 0; k < 16; k++) sm[k] = (unsigned short)o[k];
#pragma unroll
      for (int k = 0; k < 8; k++) d[k] = (unsigned)sm[2 * k] | ((unsigned)sm[2 * k + 1] << 16);
    }
#pragma unroll
    for (int k = 0; k < 8; k++)
    {
      const int lo = (int)(short)(d[k] & 0xFFFFu), hi = (int)(short)(d[k] >> 16);
      neg += (unsigned)max(-lo, 0) + (unsigned)max(-hi, 0);
      d[k] = (unsigned)max(lo, 0) | ((unsigned)max(hi, 0) << 16);
    }
    unsigned E[8], O[8];
#pragma unroll
    for (int k = 0; k < 8; k++)
    {
      E[k] = d[k];
      O[k] = __builtin_amdgcn_alignbit(d[(k + 1) & 7], d[k], 16);            // k < 7: samples (2k+1, 2k+2); k = 7: (15, 0)
    }
    unsigned* pe = packed + ((size_t)b * perBlockUnits + rem) * 16;
    reinterpret_cast<uint4*>(pe)[0] = make_uint4(E[0], E[1], E[2], E[3]); reinterpret_cast<uint4*>(pe)[1] = make_uint4(E[4], E[5], E[6], E[7]);
    reinterpret_cast<uint4*>(pe)[2] = make_uint4(O[0], O[1], O[2], O[3]); reinterpret_cast<uint4*>(pe)[3] = make_uint4(O[4], O[5], O[6], O[7]);
  }
  // block sums: inside a wavefront over aligned groups of min(perBlk, 64) lanes, then over the block's wavefronts through LDS
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int seg = perBlockUnits < 64 ? perBlockUnits : 64;
  for (int o2 = 1; o2 < seg; o2 <<= 1) neg += __shfl_xor(neg, o2);
  if (perBlockUnits <= 64)
  {
    if (live && (lane & (seg - 1)) == 0) negBlk[b] = neg;
  }
  else
  {
    if (lane == 0) wsum[wv] = neg;
    __syncthreads();
    const int wavesPerBlk = perBlockUnits >> 6;
    if (live && rem == 0)
    {
      unsigned t = 0;
      for (int k = 0; k < wavesPerBlk; k++) t += wsum[wv + k];
      negBlk[b] = t;
    }
  }
}

// ---- one stage = one chunk-row (16 samples of one original row) against the four positions of the lane.  Both layouts of the chunk-row
// (even: dwords (0,1) (2,3) ...; odd-shifted: (1,2) ... (13,14) (15,0)) arrive as one 64-byte scalar load and are v_sad_u16 operands from
// SGPRs; the lane's span of the window row (16 dwords, 17 when the span starts 2 or 3 samples into its 8-byte word) as 8 ds_read_b64.
// CO = byte offset of the chunk inside the row, folded into the ds_read immediates (no address arithmetic per stage).
struct R7Stage { unsigned ovE[8], ovO[8]; unsigned long long d[8]; unsigned x1; };

template <int OA, int CO>
__device__ __forceinline__ void r7_issue_at(R7Stage& st, const unsigned* __restrict__ op, unsigned a)
{
#pragma unroll
  for (int k = 0; k < 8; k++) { st.ovE[k] = op[k]; st.ovO[k] = op[8 + k]; }        // wave-uniform: one 64-byte scalar load
#define R7_OFFS "i"(CO), "i"(CO + 8), "i"(CO + 16), "i"(CO + 24), "i"(CO + 32), "i"(CO + 40), "i"(CO + 48), "i"(CO + 56), "i"(CO + 64)
  //             %10       %11          %12           %13           %14           %15           %16           %17           %18
  if (OA < 2)
  {
    unsigned dummy;
    asm volatile("ds_read_b64 %0, %9 offset:%10\n\tds_read_b64 %1, %9 offset:%11\n\tds_read_b64 %2, %9 offset:%12\n\tds_read_b64 %3, %9 offset:%13\n\t"
                 "ds_read_b64 %4, %9 offset:%14\n\tds_read_b64 %5, %9 offset:%15\n\tds_read_b64 %6, %9 offset:%16\n\tds_read_b64 %7, %9 offset:%17"
                 : "=&v"(st.d[0]), "=&v"(st.d[1]), "=&v"(st.d[2]), "=&v"(st.d[3]), "=&v"(st.d[4]), "=&v"(st.d[5]), "=&v"(st.d[6]), "=&v"(st.d[7]), "=&v"(dummy)
                 : "v"(a), R7_OFFS : "memory");
  }
  else
    asm volatile("ds_read_b64 %0, %9 offset:%10\n\tds_read_b64 %1, %9 offset:%11\n\tds_read_b64 %2, %9 offset:%12\n\tds_read_b64 %3, %9 offset:%13\n\t"
                 "ds_read_b64 %4, %9 offset:%14\n\tds_read_b64 %5, %9 offset:%15\n\tds_read_b64 %6, %9 offset:%16\n\tds_read_b64 %7, %9 offset:%17\n\t"
                 "ds_read_b32 %8, %9 offset:%18"
                 : "=&v"(st.d[0]), "=&v"(st.d[1]), "=&v"(st.d[2]), "=&v"(st.d[3]), "=&v"(st.d[4]), "=&v"(st.d[5]), "=&v"(st.d[6]), "=&v"(st.d[7]), "=&v"(st.x1)
                 : "v"(a), R7_OFFS : "memory");
#undef R7_OFFS
}
template <int OA, int LGCH>
__device__ __forceinline__ void r7_issue(R7Stage& st, const unsigned* __restrict__ op, unsigned a, int ch)      // ch: wave-uniform chunk index
{
  if (LGCH == 0) r7_issue_at<OA, 0>(st, op, a);
  else if (LGCH == 1) { if (ch == 0) r7_issue_at<OA, 0>(st, op, a); else r7_issue_at<OA, 32>(st, op, a); }
  else
  {
    if (ch == 0) r7_issue_at<OA, 0>(st, op, a); else if (ch == 1) r7_issue_at<OA, 32>(st, op, a);
    else if (ch == 2) r7_issue_at<OA, 64>(st, op, a); else r7_issue_at<OA, 96>(st, op, a);
  }
}

// position m of the lane starts OA + 5 m samples into the span: dword I = (OA + 5 m) >> 1, parity P = (OA + 5 m) & 1.  The four accumulators
// take turns (k outer, m inner): chains of eight dependent v_sad_u16 cost 12 % (tools/micro/sadloop_rate.hip).
template <int OA>
__device__ __forceinline__ void r7_compute(const R7Stage& st, unsigned (&acc)[4])
{
  unsigned dd[17];
#pragma unroll
  for (int k = 0; k < 8; k++)
  {
    asm volatile("" :: "v"(st.d[k]));                     // whole 64-bit destination stays allocated until here
    dd[2 * k] = (unsigned)st.d[k]; dd[2 * k + 1] = (unsigned)(st.d[k] >> 32);
  }
  if (OA >= 2) { asm volatile("" :: "v"(st.x1)); dd[16] = st.x1; } else dd[16] = 0u;
  unsigned mg[4] = { 0u, 0u, 0u, 0u };
#pragma unroll
  for (int m = 0; m < 4; m++)
  {
    const int s = OA + 5 * m, I = s >> 1;
    if (s & 1) mg[m] = (dd[I + 8] & 0xFFFFu) | (dd[I] & 0xFFFF0000u);
  }
#pragma unroll
  for (int k = 0; k < 8; k++)
#pragma unroll
    for (int m = 0; m < 4; m++)
    {
      const int s = OA + 5 * m, I = s >> 1;
      if (s & 1) acc[m] = __builtin_amdgcn_sad_u16(st.ovO[k], k < 7 ? dd[I + 1 + k] : mg[m], acc[m]);
      else       acc[m] = __builtin_amdgcn_sad_u16(st.ovE[k], dd[I + k], acc[m]);
    }
}

// nStages chunk-rows from chunk ch0 of an original row.  a = the lane's LDS byte address of its span in the current ring row (the ring
// starts at LDS address 0): a row step is  a = min(a + step, a + step - ringBytes)  as unsigned numbers -- the second term wraps to a
// huge value until the lane passes the end of the ring.
template <int OA, int LGCH>
__device__ __forceinline__ void r7_positions(const unsigned* __restrict__ op, unsigned a, unsigned ringBytes, unsigned rowStepB,
                                             int ch0, int nStages, unsigned (&acc)[4])
{
  constexpr int CH = 1 << LGCH;
  R7Stage A, B;
  int ch = ch0;
  unsigned oOff = 0;
  auto issue = [&](R7Stage& st)
  {
    r7_issue<OA, LGCH>(st, op + oOff, a, ch);
    oOff += 16u; ch++;
    if (ch == CH)
    {
      asm volatile("" ::: "memory");                                           // keeps the row step in its (wave-uniform) branch
      ch = 0;
      const unsigned t = a + rowStepB;
      a = min(t, t - ringBytes);
    }
  };
  issue(A);
  for (int s = 0; s < nStages; s += 2)
  {
    R7_WAIT_LGKM0();
    if (s + 1 < nStages) issue(B);
    __builtin_amdgcn_sched_barrier(0);
    r7_compute<OA>(A, acc);
    if (s + 1 >= nStages) break;
    R7_WAIT_LGKM0();
    if (s + 2 < nStages) issue(A);
    __builtin_amdgcn_sched_barrier(0);
    r7_compute<OA>(B, acc);
  }
}

// The units beyond the last full group (39 x 39 positions: 390 units = 6 groups + 6 units) as ONE wave item: lane = (unit, phase), a lane
// takes the chunk-rows phase, phase + P, ... of its unit, so the wave needs ceil(S / P) rounds instead of S stages.  Lanes of one round
// are at different chunk-rows: the packed original row is a per-lane vector load here (VGPR operands), everything else is the stage body.
template <int OA, int LGCH>
__device__ __forceinline__ void r7_tail(const unsigned* __restrict__ orgSlab, int S, int P, int ph, bool valid, unsigned rowBase, unsigned colB,
                                        unsigned RRu, int ss, unsigned (&acc)[4])
{
  constexpr int CH = 1 << LGCH;
  for (int s0 = 0; s0 < S; s0 += P)
  {
    const int s = s0 + ph;
    const bool act = valid && s < S;
    const int sc = act ? s : 0;
    const int y = sc >> LGCH, ch = sc & (CH - 1);
    const unsigned rr0 = rowBase + (unsigned)(y << ss);
    const unsigned rr = min(rr0, rr0 - RRu);
    R7Stage st;
    r7_issue_at<OA, 0>(st, orgSlab + (size_t)sc * 16, rr * R7_P + colB + 32u * (unsigned)ch);
    R7_WAIT_LGKM0();
    __builtin_amdgcn_sched_barrier(0);
    unsigned a2[4] = { acc[0], acc[1], acc[2], acc[3] };
    r7_compute<OA>(st, a2);
#pragma unroll
    for (int m = 0; m < 4; m++) acc[m] = act ? a2[m] : acc[m];
  }
}

struct R7Params
{
  int rs, nblocks;
  int w, h, subShift, SH, nSlab, lgCH, S, lgS;   // S = stages (chunk-rows) per slab, a power of two
  int dx0, dy0, nx, ny, nq, nUnits, nGroups;         // nGroups = groups of 64 units that run in the scalar-original form
  int tailRem, tailP, tailCost; unsigned tpM;     // the remaining tailRem units: tailP chunk-row phases per unit in ONE wave (r7_tail); cost per round in stages
  int svcCost;                                    // stages charged to the service wave per block (arg-min pass)
  unsigned nqM, ngM, nbM, pwM, nxM;               // ceil(2^32 / d) for d = nq, nGroups, NB, 4 nq, nx (0 for d = 1): r7_div
  int NB, winRows, RR, CS;
  int SP, SPpad;                                  // surface entries per block: ny rows of 4 nq (whole quads: the flush needs no column test); padded to whole waves
  int surfOff, keyOff, recOff, dscOff, miscOff;   // LDS byte offsets behind the ring
  vvcgpu_mvcost mv; int useBest;
  int dbg;                                    // timing-only ablations (tools/r7_parts.py): 1 no SAD loop, 2 no arg-min pass, 4 no DMA, 32 one block's packed rows
};

// rows [0, nRows) of src (row stride rs samples, 16-byte aligned, `pieces` <= 37 16-byte pieces per row) -> ring rows ringRow0 .. (mod RR):
// one DMA instruction per row (lane = piece), rows dealt to the first nW waves
__device__ __forceinline__ void r7_dma_rows(unsigned char* ring, int ringRow0, int nRows, int RR, const Pel* src, int rs, int pieces, int wave, int lane, int nW)
{
  const Pel* lp = src + lane * 8;
  if (wave >= nW) return;                                                     // the waves with vector loads of their own keep their vmcnt queue free of DMA
  for (int r = wave; r < nRows; r += nW)
  {
    int rr = ringRow0 + r; if (rr >= RR) rr -= RR;
    if (lane < pieces) r7_glds16(lp + (size_t)r * rs, ring + (size_t)rr * R7_P);
  }
}

struct R7Rec { int v[8]; };          // step record (32 bytes, LDS): written once per chunk of steps, see the kernel

constexpr int R7_SVC = R7_WAVES - 2;           // service wave: arg-min pass of the finished step, beside the others' SAD loops of the next one
constexpr int R7_TAIL = R7_WAVES - 1;          // the wave that runs the tail items

// packed: [block][hs * CH][16] dwords, negBlk: [block] (r7_pack_org_kernel).  The pointers are separate __restrict__ arguments so
// that the wave-uniform reads of the packed rows become scalar loads (no vmcnt traffic beside the DMA).
// Schedule of a workgroup.  Steps alternate between TWO surfaces: the waves add step k into surface k & 1, one barrier, then fourteen waves
// go straight on to step k + 1 while the service wave scans surface k & 1 (arg-min in scan order, motion-vector cost per position, record
// written, surface cleared) and the tail wave runs the tail items; both are charged their extra work in the split of the SAD runs, so all
// sixteen waves reach the next barrier together.  One barrier per (step, slab); nothing but that barrier serialises a step.
template <int LGCH>
__global__ __launch_bounds__(R7_THREADS, 4) void sad_raster7_kernel(const unsigned* __restrict__ packed, const unsigned* __restrict__ negBlk,
                                                                    const Pel* __restrict__ ref, const vvcgpu_search_blk* __restrict__ blocks,
                                                                    unsigned* __restrict__ out, vvcgpu_search_best* __restrict__ best, unsigned long long* __restrict__ diag, const R7Params p)
{
  extern __shared__ __align__(16) unsigned char r7lds[];                               // everything in the dynamic region (no static in front of it)
  unsigned char* ring = r7lds;
  unsigned* surf = reinterpret_cast<unsigned*>(r7lds + p.surfOff);                      // [2][NB][SP]
  unsigned long long* waveKey = reinterpret_cast<unsigned long long*>(r7lds + p.keyOff); // [2][NB][16] (cost << 32 | entry) of each wave
  unsigned* waveV = reinterpret_cast<unsigned*>(r7lds + p.keyOff + 2 * p.NB * R7_WAVES * 8);   // [2][NB][16] the SAD sum of that entry
  R7Rec* recs = reinterpret_cast<R7Rec*>(r7lds + p.recOff);                             // [CS][NB]
  vvcgpu_search_blk* dsc = reinterpret_cast<vvcgpu_search_blk*>(r7lds + p.dscOff);      // [CS][NB]
  int& sG = *reinterpret_cast<int*>(r7lds + p.miscOff);
  constexpr int CH = 1 << LGCH;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = p.nblocks, W = p.w, npos = p.nx * p.ny, NB = p.NB, SP = p.SP, SPA = p.SPpad, PW = 4 * p.nq;
  if ((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ring != 0u) __builtin_trap();   // r7_positions: ring at LDS address 0

  // ---- once per workgroup: cleared surfaces, bit counts of the raster columns / rows, and the list's row length G (blocks [0, G) are a run
  // of horizontal neighbours; G only shapes the partition -- every chain link and every step is verified below, so any list is served exactly)
  if (tid == 0) sG = n;
  __syncthreads();
  {
    const int g = tid + 1;
    if (g < n)
    {
      const vvcgpu_search_blk a = blocks[g - 1], b = blocks[g];
      if (!(b.ref_x == a.ref_x + W && b.ref_y == a.ref_y)) atomicMin(&sG, g);
    }
  }
  __syncthreads();
  const int G = __builtin_amdgcn_readfirstlane(min(sG, R7_THREADS));
  const int nRowsL = (n + G - 1) / G, gpr = (G + NB - 1) / NB;                    // list rows, steps per list row
  const long long Q = (long long)gpr * nRowsL;                                    // steps; a block's slabs stay in one workgroup
  // XCD-aware order (speed only): workgroups are dealt round-robin over the 8 XCDs; give every XCD one contiguous run of the chains
  const int nwg = (int)gridDim.x;
  const int rho = (nwg & 7) == 0 ? (int)(blockIdx.x & 7) * (nwg >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int q0 = (int)(Q * rho / nwg), q1 = (int)(Q * (rho + 1) / nwg);

  const int ss = p.subShift, SH = p.SH, RR = p.RR, S = p.S, lgS = p.lgS, CS = p.CS;
  const unsigned ringBytes = (unsigned)RR * R7_P, rowStepB = (unsigned)R7_P << ss;
  const int perBlkAll = p.nSlab * S;                                              // chunk-rows per block
  int ringTop = 0;                                                                // ring row of the current window's first row
  bool nextReady = false;                                                         // the rows of the coming (step, slab) are in the ring / in flight
  int buf = 0;                                                                    // surface of the current step
  int svcBp = 0, svcNb = 0, svcBuf = 0;                                           // the finished step whose surface waits for the service wave
  // diagnostic launches (VVCGPU_R7_DIAG): cycles between stamps 0 .. 7 of a phase, summed in registers, written once at the end
  unsigned long long dT[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, dPrev = 0; unsigned dN = 0;
#define R7_STAMP(i) do { if (diag) { const unsigned long long t_ = __builtin_readcyclecounter(); if (dPrev) dT[i] += t_ - dPrev; dPrev = t_; } } while (0)
  unsigned svcNeg = 0;                                                            // service wave: negBlk of the finished step's block t in lane t

  // the thread's two surface entries (tid, tid + 1024: the same in every step) with their motion-vector cost uint32(lambda * bits)
  // (RdCost.h:172-199); 0x7FFFFFFF marks a padding column or an entry beyond the surface (above every cost: host check)
  unsigned pc0 = 0x7FFFFFFFu, pc1 = 0x7FFFFFFFu;
  {
    auto entry = [&](int e, unsigned& pc)
    {
      if (e >= SP) return;
      const int jj = r7_div(e, p.pwM), ii = e - jj * PW;
      if (ii >= p.nx) return;
      pc = 0u;
      if (p.useBest)
      {
        const int vx = ((p.dx0 + ii * 5) << p.mv.cost_scale) - p.mv.pred_hor, vy = ((p.dy0 + jj * 5) << p.mv.cost_scale) - p.mv.pred_ver;
        const unsigned bits = r7_expgolomb_bits(vx >> p.mv.imv_shift) + r7_expgolomb_bits(vy >> p.mv.imv_shift);
        pc = __double2uint_rz(p.mv.lambda * (double)bits);                       // == uint64(lambda * bits): the product is < 2^31
      }
    };
    entry(tid, pc0);
    entry(tid + R7_THREADS, pc1);
  }
  for (int i = tid; i < 2 * NB * SPA; i += R7_THREADS) surf[i] = 0u;

  // winner of a step whose wave keys are complete (one barrier after its arg-min pass): the service wave reduces the sixteen keys of each block
  auto final_reduce = [&]()
  {
    for (int t = 0; t < svcNb; t++)
    {
      const unsigned nk = (unsigned)__builtin_amdgcn_readlane((int)svcNeg, t);        // sum of the negative parts of block t
      const int ki = (svcBuf * NB + t) * R7_WAVES;
      const unsigned long long k = lane < R7_WAVES ? waveKey[ki + lane] : ~0ull;
      const unsigned long long m = r7_wave_min_u64(k);
      const int src = (int)__builtin_ctzll(__ballot(k == m));
      const unsigned vw = lane < R7_WAVES ? waveV[ki + lane] : 0u;
      const unsigned vwin = (unsigned)__builtin_amdgcn_readlane((int)vw, src);
      if (lane == 0)
      {
        const int e = (int)(unsigned)m, jj = r7_div(e, p.pwM), ii = e - jj * PW;
        vvcgpu_search_best r;
        r.x = p.dx0 + ii * 5; r.y = p.dy0 + jj * 5; r.cost = (m >> 32) + ((unsigned long long)nk << ss); r.sad = (unsigned long long)(vwin + nk) << ss;
        best[svcBp + t] = r;
      }
    }
    svcNb = 0;
  };

  // the pending arg-min pass: the finished step whose surface has not been scanned yet (scanned by every wave behind its SAD runs of the
  // NEXT phase, so that the pass's LDS round trip and its two wave reductions run beside the other waves' SAD loops)
  int argBp = 0, argNb = 0, argBuf = 0; unsigned argNeg = 0;
  auto argmin_pass = [&]()
  {
    for (int t = 0; t < argNb; t++)
    {
      unsigned* st = surf + (argBuf * NB + t) * SPA;
      unsigned v0 = 0u, v1 = 0u;
      if (tid < SPA) { v0 = st[tid]; st[tid] = 0u; }
      if (tid + R7_THREADS < SPA) { v1 = st[tid + R7_THREADS]; st[tid + R7_THREADS] = 0u; }
      if (out)                                                                 // the surface leaves with the constant of the clamped original
      {
        const unsigned nk = negBlk[argBp + t];
        unsigned* ob = out + (size_t)(argBp + t) * npos;
        if (pc0 != 0x7FFFFFFFu) { const int jj = r7_div(tid, p.pwM), ii = tid - jj * PW; ob[jj * p.nx + ii] = (v0 + nk) << ss; }
        if (pc1 != 0x7FFFFFFFu) { const int e = tid + R7_THREADS, jj = r7_div(e, p.pwM), ii = e - jj * PW; ob[jj * p.nx + ii] = (v1 + nk) << ss; }
      }
      if (p.useBest)
      {
        const unsigned c0 = (v0 << ss) + pc0, c1 = (v1 << ss) + pc1;
        const unsigned c = min(c0, c1);                                        // entry tid comes first in scan order: it wins a tie
        const unsigned e = c1 < c0 ? (unsigned)(tid + R7_THREADS) : (unsigned)tid;
        const unsigned vv = c1 < c0 ? v1 : v0;
        const unsigned cmin = r7_wave_min_u32(c);
        const unsigned emin = r7_wave_min_u32(c == cmin ? e : 0xFFFFFFFFu);
        const unsigned vw = (unsigned)__builtin_amdgcn_readlane((int)vv, (int)__builtin_ctzll(__ballot(c == cmin && e == emin)));
        if (lane == 0)
        {
          waveKey[(argBuf * NB + t) * R7_WAVES + wave] = ((unsigned long long)cmin << 32) | emin;
          waveV[(argBuf * NB + t) * R7_WAVES + wave] = vw;
        }
      }
    }
    argNb = 0;
  };

  __builtin_amdgcn_s_setprio(2);
  bool carryChained = false;                                                      // chain bit of a chunk's first step (known from the chunk before)
  const int winStepRows = p.nSlab * SH;
  for (int qc = q0; qc < q1; qc += CS - 1)
  {
    // ---- records of steps [qc, qc + CS): everything a step needs, decided once by one thread per step (the waves then read 32 bytes):
    //   A = { first block, nb | parts << 8 | chained << 16 | next chained << 17 | pieces << 20 | off << 28, source address of slab 0 }
    //   B = { first block of the next step or -1 }
    const int nLocal = min(CS, q1 - qc);
    __syncthreads();                                                              // the previous chunk's records are no longer read
    if (tid < CS * NB)
    {
      const int s = r7_div(tid, p.nbM), t = tid - s * NB;
      int4v* dq = reinterpret_cast<int4v*>(dsc);
      dq[tid] = int4v{ 0, 0, (int)0x80000000, 0 };
      if (s < nLocal)
      {
        const int q = qc + s, x = q / nRowsL, y = q - x * nRowsL, b0 = y * G + x * NB;
        if (x * NB + t < G && b0 + t < n) dq[tid] = reinterpret_cast<const int4v*>(blocks)[b0 + t];
      }
    }
    __syncthreads();
    if (tid < CS * NB)
    {
      const int s = r7_div(tid, p.nbM), t = tid - s * NB;
      // (nb, adjacent, first block) of local step ls
      auto shape = [&](int ls, int& b0, int& nbStep, bool& adj)
      {
        const int q = qc + ls, x = q / nRowsL, y = q - x * nRowsL;
        b0 = y * G + x * NB;
        nbStep = min(NB, min(G - x * NB, n - b0));
        if (nbStep < 0) nbStep = 0;
        adj = true;
        const vvcgpu_search_blk d0 = dsc[ls * NB];
        for (int k = 1; k < nbStep; k++)
        {
          const vvcgpu_search_blk dk = dsc[ls * NB + k];
          adj = adj && dk.ref_x == d0.ref_x + k * W && dk.ref_y == d0.ref_y;
        }
      };
      // step lb continues the chain of step la: both one part of the same width, same columns, window one block height further down
      auto chains = [&](int la, int lb)
      {
        if (la < 0 || lb >= nLocal) return false;
        int b0a, nba, b0b, nbb; bool aa, ab;
        shape(la, b0a, nba, aa); shape(lb, b0b, nbb, ab);
        const vvcgpu_search_blk da = dsc[la * NB], db = dsc[lb * NB];
        return aa && ab && nba > 0 && nba == nbb && da.ref_x == db.ref_x && db.ref_y == da.ref_y + winStepRows;
      };
      int4v ra = { 0, 0, 0, 0 }, rb = { -1, 0, 0, 0 };
      if (s < nLocal)
      {
        int b0, nbStep; bool adj;
        shape(s, b0, nbStep, adj);
        const vvcgpu_search_blk dt = dsc[tid];
        const int nbRec = adj ? nbStep : 1, parts = adj ? (nbStep > 0 ? 1 : 0) : nbStep;
        const long long wx = (long long)dt.ref_x + p.dx0, wy = (long long)dt.ref_y + p.dy0;
        const int off = (int)(wx & 7);
        const int pieces = (((p.nx - 1) * 5 + nbRec * W + off) * 2 + 15) >> 4;
        const unsigned long long src = (unsigned long long)(size_t)(ref + wy * p.rs + (wx - off));
        const int ch = (t == 0 && chains(s - 1, s)) ? 1 : 0, chn = (t == 0 && chains(s, s + 1)) ? 1 : 0;
        ra = int4v{ b0 + t, nbRec | parts << 8 | ch << 16 | chn << 17 | pieces << 20 | off << 28, (int)(unsigned)src, (int)(unsigned)(src >> 32) };
        if (s + 1 < nLocal)
        {
          int b1, nb1; bool a1;
          shape(s + 1, b1, nb1, a1);
          if (nb1 > 0) rb.x = b1;
        }
      }
      reinterpret_cast<int4v*>(recs)[2 * tid] = ra;
      reinterpret_cast<int4v*>(recs)[2 * tid + 1] = rb;
    }
    __syncthreads();
    const bool lastChunk = qc + CS >= q1;
    const int nProc = lastChunk ? nLocal : CS - 1;                                // the chunk's last step is the next chunk's first
    for (int s = 0; s < nProc; s++)
    {
      const int4v A0 = reinterpret_cast<const int4v*>(recs)[2 * s * NB];
      const int info0 = __builtin_amdgcn_readfirstlane(A0.y);
      const int nParts = (info0 >> 8) & 255;
      for (int part = 0; part < nParts; part++)
      {
        const int4v A = reinterpret_cast<const int4v*>(recs)[2 * (s * NB + part)];
        const int bp = __builtin_amdgcn_readfirstlane(A.x), info = __builtin_amdgcn_readfirstlane(A.y);
        const unsigned long long srcA = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(A.z) | ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(A.w) << 32);
        const int nb = info & 255, pieces = (info >> 20) & 63, off = (info >> 28) & 7, OA = off & 3;
        const bool chainedIn = part == 0 && (s == 0 ? carryChained : ((info >> 16) & 1) != 0);
        const bool chainsOn = part + 1 == nParts && ((info >> 17) & 1) != 0;
        const int nbp = part + 1 == nParts ? __builtin_amdgcn_readfirstlane(reinterpret_cast<const int4v*>(recs)[2 * (s * NB) + 1].x) : -1;
        R7_STAMP(0);
        // service wave: the sums of the negative parts of this step's blocks (lane t: block t), requested now, used behind the step's last barrier
        unsigned negReq = 0;
        if (wave == R7_SVC && lane < nb) negReq = negBlk[bp + lane];
        unsigned* sf = surf + buf * NB * SPA;
        for (int slab = 0; slab < p.nSlab; slab++)
        {
          const Pel* src = reinterpret_cast<const Pel*>((size_t)srcA) + (size_t)slab * SH * p.rs;
          if (slab > 0 || chainedIn) { ringTop += SH; if (ringTop >= RR) ringTop -= RR; }
          else
          {
            __syncthreads();                                                      // nobody reads the ring any more
            ringTop = 0;
            if (!(p.dbg & 4)) r7_dma_rows(ring, 0, p.winRows, RR, src, p.rs, pieces, wave, lane, R7_WAVES);
            __syncthreads();                                                      // (waits for the DMA: vmcnt(0) in front of the barrier)
          }
          // the SH new rows of the coming (step, slab), if it continues this chain: in flight during the SAD loop
          if (slab + 1 < p.nSlab || chainsOn)
          {
            int r0 = ringTop + p.winRows; if (r0 >= RR) r0 -= RR;
            if (!(p.dbg & 4)) r7_dma_rows(ring, r0, SH, RR, src + (size_t)p.winRows * p.rs, p.rs, pieces, wave, lane, R7_SVC);
          }
          R7_STAMP(1);
          // the packed rows of the coming (step, slab) into this XCD's L2 (the SAD loop's scalar loads then miss the scalar cache only)
          unsigned pf = 0u;
          if (wave < R7_SVC)
          {
            const unsigned* nxt = slab + 1 < p.nSlab ? packed + ((size_t)bp * perBlkAll + (size_t)(slab + 1) * S) * 16
                                                      : (nbp >= 0 ? packed + (size_t)nbp * perBlkAll * 16 : nullptr);
            const int li = wave * 64 + lane;
            if (nxt && li < nb * S && !(p.dbg & 4)) pf = nxt[(size_t)(li >> lgS) * perBlkAll * 16 + (size_t)(li & (S - 1)) * 16];   // used (as a dummy) behind the barrier
          }

          // ---- the service wave first writes the records of the step whose wave keys were complete at the last barrier
          R7_STAMP(2);
          const int svcBlocks = svcNb;
          if (wave == R7_SVC && svcNb > 0 && p.useBest)
          {
            final_reduce();
          }
          svcNb = 0;
          R7_STAMP(3);

          // ---- SAD loop: the (block, group, chunk-row) space of the step in 16 runs; the service wave and the tail wave get shorter ones
          {
            const int T = (nb * p.nGroups) << lgS;
            const int rounds = p.tailP ? (S + p.tailP - 1) / p.tailP : 0;
            const int costA = svcBlocks * p.svcCost, costB = nb * rounds * p.tailCost;
            const int tot = T + costA + costB;
            auto bnd = [&](int w) { return (int)((unsigned)(tot * w) >> 4); };
            int f, fEnd;
            {
              const int e14 = min(T, max(bnd(14), bnd(15) - costA));
              if (wave < R7_SVC) { f = min(T, bnd(wave)); fEnd = min(T, bnd(wave + 1)); }
              else if (wave == R7_SVC) { f = min(T, bnd(14)); fEnd = e14; }
              else { f = e14; fEnd = T; }
              if (p.dbg & 1) f = fEnd;
            }
            const int colU = 8 * (off >> 2);
            R7_STAMP(4);
            __builtin_amdgcn_s_setprio(0);                                        // (the serial parts of a phase run at priority 2: see below)
            if (p.tailP && wave == R7_TAIL && !(p.dbg & 1))
            {
              const int ui = r7_div(lane, p.tpM), ph = lane - ui * p.tailP;
              const bool valid = ui < p.tailRem;
              const int uu = min(64 * p.nGroups + ui, p.nUnits - 1);
              const int j = r7_div(uu, p.nqM), qd = uu - j * p.nq;
              for (int t = 0; t < nb; t++)
              {
                const unsigned* orgSlab = packed + ((size_t)(bp + t) * perBlkAll + (size_t)slab * S) * 16;
                const unsigned colB = (unsigned)(40 * qd + (t * W) * 2 + colU);
                unsigned acc[4] = { 0u, 0u, 0u, 0u };
                if (OA == 0)      r7_tail<0, LGCH>(orgSlab, S, p.tailP, ph, valid, (unsigned)(ringTop + 5 * j), colB, (unsigned)RR, ss, acc);
                else if (OA == 1) r7_tail<1, LGCH>(orgSlab, S, p.tailP, ph, valid, (unsigned)(ringTop + 5 * j), colB, (unsigned)RR, ss, acc);
                else if (OA == 2) r7_tail<2, LGCH>(orgSlab, S, p.tailP, ph, valid, (unsigned)(ringTop + 5 * j), colB, (unsigned)RR, ss, acc);
                else              r7_tail<3, LGCH>(orgSlab, S, p.tailP, ph, valid, (unsigned)(ringTop + 5 * j), colB, (unsigned)RR, ss, acc);
                if (valid) r7_flush4(sf + t * SPA + j * PW + 4 * qd, acc);
              }
            }
            while (f < fEnd)
            {
              const int gq = f >> lgS, s0 = f & (S - 1), len = min(S - s0, fEnd - f);
              const int t = r7_div(gq, p.ngM), g = gq - t * p.nGroups;
              const int u = 64 * g + lane;                                         // groups of the scalar form are full: every lane has a unit
              const int j = r7_div(u, p.nqM), qd = u - j * p.nq;
              const int y0 = s0 >> LGCH, ch0 = s0 & (CH - 1);
              const unsigned rr0 = (unsigned)(ringTop + (y0 << ss) + 5 * j);
              const unsigned rr = min(rr0, rr0 - (unsigned)RR);                    // rr0 < 2 RR
              const unsigned a = rr * R7_P + (unsigned)(40 * qd + (t * W) * 2 + colU);
              const unsigned* orgQ = packed + ((size_t)((p.dbg & 32) ? 0 : bp + t) * perBlkAll + (size_t)(slab * S + s0)) * 16;
              unsigned acc[4] = { 0u, 0u, 0u, 0u };
              if (OA == 0)      r7_positions<0, LGCH>(orgQ, a, ringBytes, rowStepB, ch0, len, acc);
              else if (OA == 1) r7_positions<1, LGCH>(orgQ, a, ringBytes, rowStepB, ch0, len, acc);
              else if (OA == 2) r7_positions<2, LGCH>(orgQ, a, ringBytes, rowStepB, ch0, len, acc);
              else              r7_positions<3, LGCH>(orgQ, a, ringBytes, rowStepB, ch0, len, acc);
              if (u < p.nUnits) r7_flush4(sf + t * SPA + j * PW + 4 * qd, acc);
              f += len;
            }
          }
          R7_STAMP(5);
          __builtin_amdgcn_s_setprio(2);                                          // scalar / latency chains from here to the next SAD runs: a wave in them is not starved by the SIMD's other waves
          // ---- the arg-min pass of the step BEFORE, behind this phase's SAD runs (its surface was complete at the last barrier); its keys are
          // complete at the barrier below
          const int argBlocks = argNb;
          if (argNb > 0 && !(p.dbg & 2)) { svcBp = argBp; svcBuf = argBuf; svcNeg = argNeg; argmin_pass(); }
          argNb = 0;
          R7_STAMP(6);
          R7_WAIT_LGKM0();                                                        // the flushes (inline asm: not counted by the compiler)
          __syncthreads();                                                        // the phase's barrier: surface complete, rows free, DMA landed, keys complete
          asm volatile("" :: "v"(pf));
          if (argBlocks > 0 && !(p.dbg & 2)) svcNb = argBlocks;                   // (svcNeg was latched when that step finished)
          R7_STAMP(7);
          dN++;
        }
        // the step is complete: its surface waits for the arg-min pass (behind the next phase's SAD runs)
        argBp = bp; argNb = nb; argBuf = buf; argNeg = negReq;
        buf ^= 1;
      }
    }
    if (!lastChunk) carryChained = ((__builtin_amdgcn_readfirstlane(reinterpret_cast<const int4v*>(recs)[2 * (CS - 1) * NB].y) >> 16) & 1) != 0;
    if (lastChunk) break;
  }
  // drain: the records of the step before the last, then the last step's arg-min pass, a barrier, its records
  if (wave == R7_SVC && svcNb > 0 && p.useBest) final_reduce();
  svcNb = 0;
  const int lastBlocks = (p.dbg & 2) ? 0 : argNb;
  if (lastBlocks > 0) { svcBp = argBp; svcBuf = argBuf; svcNeg = argNeg; argmin_pass(); }
  __syncthreads();
  svcNb = lastBlocks;
  if (wave == R7_SVC && svcNb > 0 && p.useBest) final_reduce();
  if (diag && blockIdx.x == 0 && lane == 0)
  {
    for (int i = 0; i < 8; i++) diag[wave * 8 + i] = dT[i];
    diag[128 + wave] = dN;
  }
#undef R7_STAMP
}

}  // namespace

// host side: 0 = launched, 1 = this form does not apply (the caller takes another kernel), < 0 = error
int vvcgpu_raster7_launch(const vvc_pel* org, int org_stride, const vvc_pel* ref, int ref_stride, const vvcgpu_search_blk* blocks, int nblocks,
                          int w, int h, int sub_shift, int dx0, int dy0, int nx, int ny, uint32_t* sad_out, const vvcgpu_mvcost* mvcost_host,
                          vvcgpu_search_best* best, hipStream_t st)
{
  if (!(w == 16 || w == 32 || w == 64)) return 1;
  if (h < 8 || h > 128 || (h & (h - 1)) != 0) return 1;
  if (nx < 1 || ny < 1 || nx > 40 || ny > 40) return 1;
  if ((ref_stride & 7) != 0 || ((uintptr_t)ref & 15) != 0) return 1;
  if (best && !(mvcost_host->lambda >= 0.0 && mvcost_host->lambda < 8.0e6)) return 1;      // 32-bit cost: SAD < 2^27, lambda * bits < 2^30
  const int SH = h < 32 ? h : 32;
  if ((SH >> sub_shift) < 1 || (SH & ((1 << sub_shift) - 1)) != 0) return 1;
  R7Params p = {};
  auto lg2 = [](int v) { int l = 0; while ((1 << l) < v) l++; return l; };
  auto magic = [](int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); };
  const int CHh = w >> 4;
  p.w = w; p.h = h; p.subShift = sub_shift; p.SH = SH; p.nSlab = h / SH; p.lgCH = lg2(CHh); p.S = (SH >> sub_shift) * CHh; p.lgS = lg2(p.S);
  if ((1 << p.lgS) != p.S) return 1;
  p.dx0 = dx0; p.dy0 = dy0; p.nx = nx; p.ny = ny; p.nq = (nx + 3) >> 2; p.nUnits = ny * p.nq;
  {
    // full groups of 64 units in the scalar-original form; a remainder of at most 32 units as one wave item with 64 / rem phases (r7_tail)
    static const int tailOff = getenv("VVCGPU_R7_NOTAIL") ? 1 : 0;                          // A/B switch
    static const int tcEnv = getenv("VVCGPU_R7_TC") ? atoi(getenv("VVCGPU_R7_TC")) : 0;     // experiment: stages charged per tail round
    const int nFull = p.nUnits >> 6, rem = p.nUnits & 63;
    static const int scEnv = getenv("VVCGPU_R7_SC") ? atoi(getenv("VVCGPU_R7_SC")) : 0;     // experiment: stages charged per block of the arg-min pass
    p.nGroups = nFull; p.tailRem = 0; p.tailP = 0; p.tailCost = tcEnv > 0 ? tcEnv : 3; p.svcCost = scEnv > 0 ? scEnv : 1;
    if (rem > 0 && rem <= 32 && nFull >= 1 && !tailOff) { p.tailRem = rem; p.tailP = 64 / rem; }
    else if (rem > 0) return 1;                                                           // (a partial group in the scalar form: not built -- strip kernels)
  }
  p.SP = ny * 4 * p.nq; p.SPpad = p.SP;
  if (p.SPpad > 2 * R7_THREADS) return 1;                                           // two surface entries per thread
  const int pitchSamples = R7_P / 2;
  if ((nx - 1) * 5 + w + 7 > pitchSamples - 8) return 1;
  int nbGeom = (pitchSamples - 8 - 7 - (nx - 1) * 5) / w;
  const int stagesPerBlk = p.nGroups * p.S + (p.tailP ? ((p.S + p.tailP - 1) / p.tailP) * p.tailCost : 0) + p.svcCost / p.nSlab;
  int NB = (180 + stagesPerBlk - 1) / stagesPerBlk;                                 // >= ~11 chunk-row stages per wave and step
  if (NB > nbGeom) NB = nbGeom;
  if (NB > R7_MAXNB) NB = R7_MAXNB;
  if (NB < 1) NB = 1;
  static const int nbEnv = getenv("VVCGPU_R7_NB") ? atoi(getenv("VVCGPU_R7_NB")) : 0;       // experiment: blocks per step
  if (nbEnv >= 1 && nbEnv <= R7_MAXNB && nbEnv <= nbGeom) NB = nbEnv;
  p.winRows = (ny - 1) * 5 + SH; p.RR = p.winRows + SH;
  p.CS = 8;
  size_t smem = 0;
  for (;; NB--)                                                                     // fewer blocks per step until both surfaces fit
  {
    size_t o = (size_t)p.RR * R7_P;
    p.surfOff = (int)o; o += (size_t)2 * NB * p.SPpad * 4;
    p.keyOff = (int)o; o += (size_t)2 * NB * R7_WAVES * 12;
    p.recOff = (int)o; o += (size_t)p.CS * NB * 32;
    p.dscOff = (int)o; o += (size_t)p.CS * NB * 16;
    p.miscOff = (int)o; o += 16;                                                    // the list's row length
    smem = o;                                                                       // (a span that ends past the last ring row reads surface bytes: harmless)
    if (smem <= 160 * 1024 - 16 || NB == 1) break;
  }
  if (smem > 160 * 1024 - 16) return 1;
  p.NB = NB;
  p.nqM = magic(p.nq); p.ngM = magic(p.nGroups); p.nbM = magic(NB); p.tpM = magic(p.tailP); p.pwM = magic(4 * p.nq); p.nxM = magic(nx);
  const int hs = h >> sub_shift, perBlk = hs * CHh;
  const size_t packedDw = (size_t)nblocks * perBlk * 16, negDw = ((size_t)nblocks + 1) & ~(size_t)1;
  static const int diagOn = getenv("VVCGPU_R7_DIAG") ? 1 : 0;                           // diagnostic build path: phase cycles of workgroup 0 to stderr (synchronises)
  const size_t diagDw = diagOn ? 2 * 160 : 0;
  unsigned* scratch = static_cast<unsigned*>(vvcgpu_scratch(st, (packedDw + negDw + diagDw) * sizeof(unsigned)));
  if (!scratch) return VVCGPU_E_DEVICE;
  p.rs = ref_stride; p.nblocks = nblocks;
  if (best) p.mv = *mvcost_host;
  p.useBest = best ? 1 : 0;
  p.dbg = getenv("VVCGPU_R7_DBG") ? atoi(getenv("VVCGPU_R7_DBG")) : 0;
  const int packT = perBlk > 256 ? 512 : 256;                                       // a block's chunk-rows inside one workgroup
  if (perBlk > 512) return 1;
  hipLaunchKernelGGL(r7_pack_org_kernel, dim3((unsigned)(((size_t)nblocks * perBlk + packT - 1) / packT)), dim3(packT), 0, st, org, org_stride, blocks, nblocks,
                     w, hs, sub_shift, scratch, scratch + packedDw);
  VVC_LAUNCH_CHECK();
  static int nCu = 0;
  if (!nCu)
  {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    nCu = v;
  }
  int grid = nCu;                                                                   // one persistent workgroup per CU, fewer for short lists
  if (nblocks < grid) grid = nblocks;
  static const int trace = getenv("VVCGPU_TRACE_PATH") ? 1 : 0;
  if (trace)
    fprintf(stderr, "[vvcgpu] sad_search %dx%d %dx%d raster: ring form, %d blocks per step, %d + %d slab rows, %d groups%s, %s, %zu B LDS, %d workgroups\n", w, h, nx, ny, NB,
            p.winRows, SH, p.nGroups, p.tailP ? " + tail item" : "", "two surfaces", smem, grid);
  unsigned long long* dg = diagOn ? reinterpret_cast<unsigned long long*>(scratch + packedDw + negDw) : nullptr;
  if (dg) VVC_HIP(hipMemsetAsync(dg, 0, 160 * 8, st));
#define R7_LAUNCH(LG)                                                                                                             \
  do {                                                                                                                            \
    VVC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sad_raster7_kernel<LG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
    hipLaunchKernelGGL(sad_raster7_kernel<LG>, dim3(grid), dim3(R7_THREADS), smem, st, scratch, scratch + packedDw, ref, blocks, sad_out, best, dg, p); \
  } while (0)
  if (p.lgCH == 0) R7_LAUNCH(0); else if (p.lgCH == 1) R7_LAUNCH(1); else R7_LAUNCH(2);
#undef R7_LAUNCH
  VVC_LAUNCH_CHECK();
  if (dg)
  {
    unsigned long long h[160];
    VVC_HIP(hipMemcpyAsync(h, dg, sizeof h, hipMemcpyDeviceToHost, st));
    VVC_HIP(hipStreamSynchronize(st));
    fprintf(stderr, "[vvcgpu] r7 diag %dx%d (workgroup 0, mean cycles per phase over %llu phases), per wave:\n"
                    "   barrier->records read | DMA rows issued | prefetch issued | key reduce | run split | SAD runs | arg-min pass | wait at the barrier\n", w, h, h[128]);
    for (int wv = 0; wv < R7_WAVES; wv++)
    {
      const double nn = (double)(h[128 + wv] ? h[128 + wv] : 1);
      fprintf(stderr, "   wave %2d:", wv);
      for (int i = 0; i < 8; i++) fprintf(stderr, " %7.0f", (double)h[wv * 8 + i] / nn);
      fprintf(stderr, "\n");
    }
  }
  return 0;
}
