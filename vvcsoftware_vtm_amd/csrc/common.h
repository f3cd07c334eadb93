// common.h -- shared plumbing of the HIP hot-path library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include "vvcgpu.h"

typedef int16_t Pel;
typedef int32_t TCoeff;

void vvcgpu_set_error(const char* fmt, ...);
// Library-internal device scratch, cached per (device, stream): work on one stream is ordered, so the buffer of the previous
// call on that stream is free again when the next call's kernels start.  Grow-only; returns nullptr (error text set) on failure.
// ONE call of an entry point may take each REGION once: an entry point that sizes its own workspace uses VVC_SCRATCH_ENTRY (vvcgpu_scratch), a
// launch helper that other entry points call with their workspace still live (vvcgpu_frac_refine_launch under vvcgpu_me_batch,
// vvcgpu_mc_batch_impl under the affine entry points) uses VVC_SCRATCH_HELPER -- two requests for the same region on one call path alias.
enum { VVC_SCRATCH_ENTRY = 0, VVC_SCRATCH_HELPER = 1, VVC_SCRATCH_IOTA = 2, VVC_SCRATCH_REGIONS = 3 };
int* vvcgpu_iota(hipStream_t stream, int n);                                 // device array 0, 1, .. of at least n ints, persistent per stream (lib.hip)
void* vvcgpu_scratch(hipStream_t stream, size_t bytes);                      // region VVC_SCRATCH_ENTRY
void* vvcgpu_scratch_region(hipStream_t stream, int region, size_t bytes);
// The ONE behaviour switch of the library (read per call): VVCGPU_NO_MFMA=1 keeps the interpolation filters, the Hadamard refinement and the
// transforms off the matrix cores (the vector-pipe bodies that otherwise serve flagged PUs / TUs take everything) -- tests/test_gpu_no_mfma.py runs the
// parity cases of those bodies under it.  Every other environment variable the library reads is a measurement aid (VVCGPU_*_DIAG: cycle stamps to
// stderr) or a test hook (VVCGPU_MH_WGS: number of persistent workgroups of the hierarchical search).
int vvcgpu_no_mfma(void);
// eager construction of the per-device table images (vvcgpu_warmup, lib.hip): each returns VVCGPU_OK or an error code with the text set
int vvcgpu_mc_image_build(int bit_depth);          // interp.hip
int vvcgpu_frac_image_build(int bit_depth);        // fracsearch.hip
int vvcgpu_tr_image_build(void);                   // resichain.hip (transform tables + f16 image)
int vvcgpu_cu_count(void);                         // compute units of the current device (queried once per device; 256 if the query fails)
constexpr int VVC_CTR_INTS = 32;                       // ints per counter set of vvcgpu_counters
int* vvcgpu_counters(hipStream_t stream, int* cur);     // two persistent zeroed work counters per (device, stream), see lib.hip
void vvcgpu_counters_failed(hipStream_t stream);       // a launch that took a counter set failed: both sets are cleared before their next use
int vvcgpu_frac_refine_launch(const vvc_pel* org, int org_stride, const vvc_pel* ref, int ref_stride, const vvcgpu_frac_blk* blocks, int nblocks,
                              int w, int h, int bit_depth, int clp_min, int clp_max, int use_hadamard, const vvcgpu_mvcost* mvcost_host,
                              const int* preds, vvcgpu_frac_result* results, void* stream);

// Raster stage of whole-PU TZ searches as its own launch (tzsearch.hip -> dist.hip): one record per PU, written on the device.  An active
// PU's raster is the nx x ny grid of step 5 whose position (0, 0) is the motion vector (x0, y0); blocks[] holds the block's origin in the
// original and the reference position of the ZERO vector (as in vvcgpu_sad_search); the best candidate comes back as the packed
// key (cost << 24 | j * nx + i) in best[].cost (all-ones: no candidate).  Every PU of the launch is w x h with row sub-sampling sub_shift.
struct VvcRasterPer { int active, nx, ny, x0, y0, pred_hor, pred_ver, reserved; };
int vvcgpu_raster_per_block_launch(const vvc_pel* org, int org_stride, const vvc_pel* ref, int ref_stride, const vvcgpu_search_blk* blocks,
                                   const VvcRasterPer* per, int nblocks, int w, int h, int sub_shift, int nx_max, int ny_max,
                                   const vvcgpu_mvcost* mvcost_host, vvcgpu_search_best* best, unsigned* packed_workspace, hipStream_t stream);

// device addresses (current device) of the transform matrices as int32 (tr32[type][size] row-major T[k][n] at type * 5460 + (n n - 4) / 3, tr32t its
// transpose), of the raster position -> scan index tables (dqInv + scanOff[(log2 w - 1) * 6 + log2 h - 1]); uploaded on first use (transform.hip)
struct VvcTrTables { const int* tr32; const int* tr32t; const unsigned short* dqInv; const int* scanOff; };
int vvcgpu_tr_tables(VvcTrTables* out);

#define VVC_CHECK_ARG(cond, ...)                                   \
  do { if (!(cond)) { vvcgpu_set_error(__VA_ARGS__); return VVCGPU_E_ARG; } } while (0)

#define VVC_HIP(call)                                                                        \
  do { hipError_t e_ = (call); if (e_ != hipSuccess) {                                       \
         vvcgpu_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
         return VVCGPU_E_DEVICE; } } while (0)

#define VVC_LAUNCH_CHECK()                                                                    \
  do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) {                            \
         vvcgpu_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
         return VVCGPU_E_DEVICE; } } while (0)

// the same for launches that own a counter set of vvcgpu_counters: a failure leaves the sets in an unknown state
#define VVC_LAUNCH_CHECK_COUNTERS(st)                                                         \
  do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) {                            \
         vvcgpu_counters_failed(st);                                                         \
         vvcgpu_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
         return VVCGPU_E_DEVICE; } } while (0)

// XCD-aware order of a one-dimensional grid (speed only, never correctness): workgroups are dealt round-robin over the 8 XCDs, each with its own L2.
// A picture pass whose neighbouring tiles share halo lines gives every XCD one CONTIGUOUS run of the logical tile indices, so that a 128-byte
// line is fetched from the fabric by one L2 instead of by up to three (profiles/r03_fabric_requests.csv: the picture passes read 1.6 - 3.1 x
// their planes, every request a whole 128-byte line).  Launch vvc_xcd_grid(total) workgroups; padding workgroups get -1 and leave.
static inline int vvc_xcd_on() { return 1; }
static inline int vvc_xcd_grid(int total, int on) { return on ? ((total + 7) >> 3) << 3 : total; }
// two index ranges [0, nA) and [nA, total) (luma tiles, then chroma tiles), each spread over the XCDs on its own: the heavy and the light part of a
// launch both reach every XCD
static inline int vvc_xcd_grid2(int nA, int total, int on) { return on ? vvc_xcd_grid(nA, 1) + vvc_xcd_grid(total - nA, 1) : total; }
#ifdef __HIPCC__
__device__ __forceinline__ int vvc_xcd_index(int bid, int total, int on)
{
  if (!on) return bid;
  const int chunk = (total + 7) >> 3, i = (bid & 7) * chunk + (bid >> 3);
  return i < total ? i : -1;
}
__device__ __forceinline__ int vvc_xcd_index2(int bid, int nA, int total, int on)
{
  if (!on) return bid;
  const int gridA = ((nA + 7) >> 3) << 3;
  if (bid < gridA) return vvc_xcd_index(bid, nA, 1);
  const int i = vvc_xcd_index(bid - gridA, total - nA, 1);
  return i < 0 ? -1 : nA + i;
}
#endif

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

__device__ __forceinline__ int clip3(int lo, int hi, int v) { return min(max(v, lo), hi); }
__device__ __forceinline__ int sgn(int v) { return (v > 0) - (v < 0); }

// 8 Pels = 16 bytes, the coalescing sweet spot for int16 planes (guide G13)
typedef short pel8 __attribute__((ext_vector_type(8)));
typedef short pel4 __attribute__((ext_vector_type(4)));
typedef short pel2 __attribute__((ext_vector_type(2)));
typedef int   int4v __attribute__((ext_vector_type(4)));
