/* vvcgpu.h -- C ABI of the MI355X-native pixel hot path for VTM (reference studied: VTM 2.1).
 *
 * This is the drop-in boundary: every entry point replaces one of the reference's function-pointer
 * table slots or picture-level methods; the reference-side binding that
 * installs them is shown in INTEGRATION.md.  Plain C: pointers, sizes, no C++ or torch types.
 *
 * Conventions
 *  - All sample pointers are DEVICE pointers (HBM) unless the parameter name ends in `_host`.
 *    `Pel` = int16_t, `TCoeff` = int32_t (reference: CommonLib/TypeDef.h:370-371).
 *  - Strides are in ELEMENTS, as in the reference's AreaBuf (CommonLib/Buffer.h:78-91).
 *  - A plane pointer addresses sample (0,0) of the valid picture; kernels that need samples outside
 *    the picture (ALF) replicate the border themselves, so no margin is required for those.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls are asynchronous on
 *    that stream; the caller synchronises.  Small parameter tables (`*_host`) are copied into the
 *    kernel argument block at call time and may be reused immediately.
 *  - Return value: 0 on success, negative VVCGPU_E_* otherwise; vvcgpu_last_error() gives text.
 *    The reference convention (CHECK/THROW -> Exception, TypeDef.h:1187-1208) is restored by the shim.
 *  - All entry points are re-entrant (reference may call from OpenMP jobs, EncLib.cpp:94-113).
 */
#ifndef VVCGPU_H
#define VVCGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int16_t vvc_pel;
typedef int32_t vvc_coef;

#define VVCGPU_OK            0
#define VVCGPU_E_ARG        -1   /* bad argument (shape, alignment, enum) */
#define VVCGPU_E_DEVICE     -2   /* HIP runtime error */
#define VVCGPU_E_UNSUPPORTED -3  /* outside the precondition (e.g. bit depth > 10) */

/* ---- library ------------------------------------------------------------------------------- */
int         vvcgpu_version(void);                 /* ABI version, currently 1 */
const char* vvcgpu_last_error(void);              /* thread-local text of the last failure */
int         vvcgpu_device_count(void);
int         vvcgpu_set_device(int device);
/* sizeof() of the parameter structs, for binding self-checks: 0 sao_ctu, 1 deblock_cfg, 2 dist_desc, 3 search_blk,
 * 4 mvcost, 5 search_best, 6 if_desc, 7 mc_desc, 8 pelop_desc, 9 pelop_cfg, 10 tr_desc, 11 frac_blk, 12 frac_result,
 * 13 dqtr_desc, 14 afg_desc, 15 afe_desc, 16 tz_pu, 17 tz_cfg, 18 intra_desc, 19 cclm_desc, 20 intra_fill_desc, 21 imv_pu, 22 imv_result, 23 quant_desc,
 * 24 dq_rates, 25 depquant_desc, 26 rdoq_rates, 27 rdoq_desc, 28 intra_satd_desc, 29 affine_iter, 30 me_hier_cfg; -1 for unknown ids.          */
int         vvcgpu_sizeof(int struct_id);

/* ---- device memory helpers for host-side callers (the reference keeps pictures in host memory; the shim stages them).
 *      2-D copies take pitches in BYTES.  All are asynchronous on `stream` except vvcgpu_stream_sync. */
int vvcgpu_malloc(void** dev_ptr, size_t bytes);
int vvcgpu_free(void* dev_ptr);
int vvcgpu_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes, void* stream);
int vvcgpu_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes, void* stream);
int vvcgpu_memcpy2d_h2d(void* dst_dev, size_t dst_pitch, const void* src_host, size_t src_pitch, size_t width_bytes,
                        size_t height, void* stream);
int vvcgpu_memcpy2d_d2h(void* dst_host, size_t dst_pitch, const void* src_dev, size_t src_pitch, size_t width_bytes,
                        size_t height, void* stream);
int vvcgpu_memcpy2d_d2d(void* dst_dev, size_t dst_pitch, const void* src_dev, size_t src_pitch, size_t width_bytes,
                        size_t height, void* stream);
int vvcgpu_stream_sync(void* stream);
/* Library-internal per-stream resources (work lists, packed search blocks, counters) are created on the first call that needs them and kept
 * for later calls on the same (device, stream).  A host that creates streams per thread / job calls vvcgpu_stream_release(stream) before it
 * destroys a stream: the call waits for the stream's queued work and frees what the library holds for it (any number of streams may come and
 * go; the stream's slot is found whichever device is current).  vvcgpu_shutdown() does the same for every stream of every device (e.g. before
 * unloading the library).  Both return VVCGPU_OK when there was nothing to free and VVCGPU_E_DEVICE when a device could not be reached (what
 * could not be freed is kept, not leaked).  The scratch of a stream grows geometrically; a buffer it has outgrown is freed as soon as the work
 * queued before the growth has completed.  One host thread drives a stream at a time (per-thread streams for concurrent callers). */
int vvcgpu_stream_release(void* stream);
int vvcgpu_shutdown(void);
/* Per device (and bit depth) the library keeps a few constant table images in device memory: the transform matrices (TrQuant.cpp:72-84, Rom.cpp:245-299)
 * as int32 and as the f16 image of the matrix-core kernels, the Toeplitz tap tables of the matrix-core interpolation (InterpolationFilter.cpp:59-138).
 * They are built by the FIRST call that needs them -- on the NULL stream, with a device synchronisation, under a library mutex: that call must not run
 * inside a stream capture and briefly stalls other threads' streams.  vvcgpu_warmup(bit_depth) builds them all for the current device at a time the host
 * chooses (start-up), after which no entry point synchronises the device.  Optional; idempotent. */
int vvcgpu_warmup(int bit_depth);

/* ---- A1: ALF classification  (AdaptiveLoopFilter::deriveClassification, AdaptiveLoopFilter.cpp:274-463;
 *          table slot m_deriveClassificationBlk, AdaptiveLoopFilter.h:90) -----------------------
 * src: luma plane after deblock+SAO (W x H valid samples, border replicated by the kernel exactly as
 *      ALFProcess does with extendBorderPel(3), AdaptiveLoopFilter.cpp:90-92).
 * cls: one uint16 per 4x4 luma block, row-major (H/4) x (W/4): low byte classIdx 0..24, high byte
 *      transposeIdx 0..3 (reference stores the same pair per pixel, AdaptiveLoopFilter.h:46-56).
 * W and H must be multiples of 4.                                                              */
int vvcgpu_alf_classify(const vvc_pel* src, int src_stride, int width, int height, int bit_depth,
                        uint16_t* cls, void* stream);

/* ---- A2: ALF filtering  (AdaptiveLoopFilter::ALFProcess CTU loop + filterBlk<5|7>,
 *          AdaptiveLoopFilter.cpp:68-139,465-650; table slots m_filter5x5Blk/m_filter7x7Blk) ------
 * src must not alias dst (reference filters from a copy, AdaptiveLoopFilter.cpp:87-92).
 * filter_type: 0 = 5x5 (7 coeff), 1 = 7x7 (13 coeff)  (AlfFilterType, TypeDef.h).
 * coeff_host: luma: 25 classes x 13 int16 (m_coeffFinal layout, MAX_NUM_ALF_LUMA_COEFF = 13);
 *             chroma: 7 int16.
 * ctu_enable: device array, one byte per CTU in raster order (Picture::getAlfCtuEnableFlag); CTUs with 0
 *             are left untouched in dst.  NULL = all enabled.
 * For chroma pass the chroma plane, its width/height and ctu_size = luma CTU size >> 1.          */
int vvcgpu_alf_filter_luma(const vvc_pel* src, int src_stride, vvc_pel* dst, int dst_stride,
                           int width, int height, int ctu_size, const uint16_t* cls,
                           int filter_type, const int16_t* coeff_host, const uint8_t* ctu_enable,
                           int clp_min, int clp_max, void* stream);
int vvcgpu_alf_filter_chroma(const vvc_pel* src, int src_stride, vvc_pel* dst, int dst_stride,
                             int width, int height, int ctu_size, const int16_t* coeff_host,
                             const uint8_t* ctu_enable, int clp_min, int clp_max, void* stream);

/* ---- S1: SAO apply  (SampleAdaptiveOffset::SAOProcess/offsetCTU/offsetBlock,
 *          SampleAdaptiveOffset.cpp:292-612) -----------------------------------------------------
 * One call per component.  src = deblocked plane (the reference's m_tempBuf copy), dst = recon plane;
 * samples of CTUs whose mode is OFF and samples skipped at unavailable borders are NOT written (dst is
 * expected to already hold the deblocked picture, as in the reference where dst is the source of the copy).
 * params: device array of vvcgpu_sao_ctu, one per CTU in raster order, already merge-resolved and
 *         de-quantised (xReconstructBlkSAOParams, SampleAdaptiveOffset.cpp:262-290).
 * avail bits (deriveLoopFilterBoundaryAvailibility, :685-760).                                   */
typedef struct vvcgpu_sao_ctu {
  int8_t  type;        /* -1 = off, 0 EO_0, 1 EO_90, 2 EO_135, 3 EO_45, 4 BO  (SAOModeNewTypes) */
  uint8_t avail;       /* bit0 L, 1 R, 2 A, 3 B, 4 AL, 5 AR, 6 BL, 7 BR */
  int16_t offset[32];  /* EO: offset[0..4] = classes (edgeType+2); BO: offset[band]            */
} vvcgpu_sao_ctu;
int vvcgpu_sao_apply(const vvc_pel* src, int src_stride, vvc_pel* dst, int dst_stride,
                     int width, int height, int ctu_w, int ctu_h, int bit_depth,
                     const vvcgpu_sao_ctu* params, int clp_min, int clp_max, void* stream);

/* ---- L1+L2: deblocking  (LoopFilter::loopFilterPic, LoopFilter.cpp:149-230: pass 1 all vertical edges,
 *          pass 2 all horizontal edges; xEdgeFilterLuma :543-681, xEdgeFilterChroma :684-838,
 *          xPelFilterLuma :856-916, xPelFilterChroma :928-949) -------------------------------------------
 * In place on the three reconstruction planes (4:2:0).  The CU/TU/motion walk that decides WHICH edge
 * segments are filtered and with which boundary strength (xDeblockCU :243-369, xGetBoundaryStrengthSingle
 * :419-541) stays on the host; it is handed over as maps with one entry per 4x4 luma unit, row-major
 * (height/4) x (width/4):
 *   edge_ver[u] / edge_hor[u]: the 4-sample segment on the LEFT / TOP border of unit u
 *        bits 0-1  luma   BS (0 = segment not filtered; only 8x8-grid positions may be non-zero)
 *        bits 2-3  chroma BS (chroma is filtered when it is 2 and the edge lies on the 8-sample chroma grid)
 *        bit  4    P side must not be modified (IPCM+pcm_loop_filter_disable / transquant bypass, :640-651)
 *        bit  5    Q side must not be modified
 *   qp_luma[u], qp_chroma[u]: QP of the CU covering unit u in the luma tree / chroma tree (CodingUnit::qp;
 *        identical unless the slice uses the dual tree).
 * width and height must be multiples of 8 (minimum CU size 4 with the 8x8 deblocking grid, TypeDef.h:60). */
typedef struct vvcgpu_deblock_cfg {
  int32_t bit_depth_luma, bit_depth_chroma;
  int32_t beta_offset_div2, tc_offset_div2;     /* slice deblocking offsets            */
  int32_t cb_qp_offset, cr_qp_offset;           /* PPS chroma QP offsets (:809)        */
  int32_t clp_min[3], clp_max[3];               /* ClpRng per component                */
} vvcgpu_deblock_cfg;
int vvcgpu_deblock(vvc_pel* y, int stride_y, vvc_pel* cb, vvc_pel* cr, int stride_c, int width, int height,
                   const uint8_t* edge_ver, const uint8_t* edge_hor, const int8_t* qp_luma,
                   const int8_t* qp_chroma, const vvcgpu_deblock_cfg* cfg_host, void* stream);

/* ---- S2: SAO statistics  (EncSampleAdaptiveOffset::getStatistics / getBlkStats,
 *          EncoderLib/EncSampleAdaptiveOffset.cpp:278-330, 1122-1490; isCalculatePreDeblockSamples == false,
 *          i.e. SAOLcuBoundary 0 as in all shipped cfgs) -------------------------------------------------
 * One call per component.  org = original plane, rec = deblocked plane.
 * out: per CTU (raster) 5 types x { int64 diff[32]; int64 count[32]; }  (SAOStatData, EncSampleAdaptiveOffset.h:53-80);
 *      EO classes use indices 0..4 (edgeType + 2), BO the 32 bands.  320 int64 per CTU.
 * avail: device array, one byte per CTU with the vvcgpu_sao_ctu.avail bit layout; only L (bit0), A (bit2) and
 *      AL (bit4) are read -- right/below/above-right come from the picture geometry exactly as in :300-306.
 * skip_lines_r / skip_lines_b: m_skipLinesR/B of the component (5/4 luma, 3/2 chroma; :122-128).
 * Precondition: org and rec samples within the bit depth (|org - rec| <= 1023: the per-thread accumulators pack count and sum; a band index is
 * taken modulo 32); a plane holds fewer than 2^31 samples.                                               */
int vvcgpu_sao_stats(const vvc_pel* org, int org_stride, const vvc_pel* rec, int rec_stride,
                     int width, int height, int ctu_w, int ctu_h, int bit_depth, const uint8_t* avail,
                     int skip_lines_r, int skip_lines_b, int64_t* out, void* stream);

/* ---- A3: ALF covariance statistics  (EncAdaptiveLoopFilter::deriveStatsForFiltering / getBlkStats /
 *          calcCovariance, EncoderLib/EncAdaptiveLoopFilter.cpp:1317-1515) ---------------------------------
 * One call per (component, filter shape).  rec = deblocked+SAO plane (border replicated by the kernel, :250-252),
 * org = original plane.  cls: A1 output for luma, NULL for chroma (single class).
 * filter_type 0: 5x5 (N = 7 coefficients), 1: 7x7 (N = 13).
 * out: per CTU (raster) x class (25 luma / 1 chroma): int64 E[N][N] (full symmetric), y[N], pixAcc
 *      = N*N + N + 1 values (183 / 57).  The reference accumulates the same integers into doubles
 *      (AlfCovariance, EncAdaptiveLoopFilter.h:46-52); all sums are < 2^53, so int64 is exact and
 *      order-independent.                                                                             */
int vvcgpu_alf_stats(const vvc_pel* org, int org_stride, const vvc_pel* rec, int rec_stride,
                     int width, int height, int ctu_size, const uint16_t* cls, int filter_type,
                     int64_t* out, void* stream);

/* ---- D1/D2/D3: block distortion, batched  (RdCost::m_afpDistortFunc table, RdCost.h:104; scalar bodies
 *          xGetSAD* RdCost.cpp:450-1000, xGetHADs :2855-2974 + xCalcHADs* :2205-2853, xGetSSE* :1820-2200;
 *          SIMD twins x86/RdCostX86.h:215-432, 2292-2436) ------------------------------------------------------
 * One descriptor per reference call `distFunc(DistParam)`: offsets are in elements from org_base / cur_base
 * (two device allocations, e.g. the original picture and a reference picture or a fractional-plane buffer).
 * kind 0 = SAD  (vertical subsampling: rows step 1<<sub_shift, sum <<= sub_shift, RdCost.cpp:466-492)
 *      1 = HAD  (Hadamard SATD, tile selection of xGetHADs with isQtbt = true; rectangular tiles use the
 *                reference's double arithmetic (int)(sad / sqrt(128.0) * 2), RdCost.cpp:2561,2698,2771,2850)
 *      2 = SSE  (sum of squared differences; the chroma distortion weight of getDistPart stays on the host)
 *      3 = MR-SAD  (row D4: xGetMRSAD and its size twins, RdCost.cpp:1008-1815: offset = sum(org - cur) / N over the sub-sampled
 *                block, truncating; sum |org - cur - offset| << sub_shift -- the m_afpDistortFunc[DF_MRSAD..] entries for useMR)
 *      4 = MR-HAD  (xGetMRHADs :3433-3446: kind 1 on org - Pel(meanDiff))
 * out[i] is the Distortion (uint64) of descriptor i.  Bit depth <= 10 (the reference's SIMD precondition). */
typedef struct vvcgpu_dist_desc {
  int64_t org_off, cur_off;
  int32_t org_stride, cur_stride;
  int16_t w, h;
  int16_t sub_shift;          /* SAD / MR-SAD only */
  int16_t reserved;
} vvcgpu_dist_desc;
int vvcgpu_dist_batch(int kind, const vvc_pel* org_base, const vvc_pel* cur_base, const vvcgpu_dist_desc* descs,
                      int n, int bit_depth, uint64_t* out, void* stream);

/* ---- D1 (search form): SAD surface of a block over a regular grid of integer positions
 *          (the distFunc loops of InterSearch::xPatternSearch, InterSearch.cpp:1887-1935, and of the raster stage of
 *          xTZSearch, :2159-2169, both through xTZSearchHelp/distFunc :249-343) --------------------------------
 * All blocks of one call share w, h, sub_shift and the position grid  x = dx0 + i*sx (i < nx), y = dy0 + j*sy (j < ny),
 * relative to (ref_x, ref_y) of the block.  Every probed sample must lie inside the reference plane's allocation
 * (the reference clips MVs to the padded picture, Mv.cpp:64-80).  org samples may be any int16 (bi-pred refinement
 * searches on 2*org - otherPred, InterSearch.cpp:1682-1692).
 * sad_out: nblocks x ny x nx uint32, row-major.  May be NULL when only `best` is wanted (what xPatternSearch and the raster
 *          loop of xTZSearch keep, InterSearch.cpp:1887-1935, 1979-2000): the arg-min is fused into the SAD kernels and no
 *          surface is written.  The fused arg-min packs (cost << 24 | scan index): cost < 2^40 and nx * ny < 2^24 are
 *          preconditions whenever `best` is given (motion lambda * MV bits stays below 2^20 in the reference).
 * best (optional): per block the argmin of  sad + uint64(lambda * bits(x,y))  in the reference's scan order
 *          (y outer, x inner, strict '<': InterSearch.cpp:1913-1925) with the MV-bit cost of RdCost.h:172-199.   */
typedef struct vvcgpu_search_blk { int32_t org_x, org_y, ref_x, ref_y; } vvcgpu_search_blk;
typedef struct vvcgpu_mvcost {
  double  lambda;             /* m_motionLambda                         */
  int32_t pred_hor, pred_ver; /* m_mvPredictor (after setPredictor)     */
  int32_t cost_scale;         /* m_iCostScale                           */
  int32_t imv_shift;          /* imvShift                               */
} vvcgpu_mvcost;
typedef struct vvcgpu_search_best { int32_t x, y; uint64_t cost; uint64_t sad; } vvcgpu_search_best;
int vvcgpu_sad_search(const vvc_pel* org, int org_stride, const vvc_pel* ref, int ref_stride,
                      const vvcgpu_search_blk* blocks, int nblocks, int w, int h, int sub_shift,
                      int dx0, int dy0, int nx, int ny, int sx, int sy, uint32_t* sad_out,
                      const vvcgpu_mvcost* mvcost_host, vvcgpu_search_best* best, void* stream);

/* ---- D1 + D5, hierarchical form: the step-5 raster stage of xTZSearch (InterSearch.cpp:2159-2169) AND the +-dense_range full search of
 *          xPatternSearch (:1886-1935) for EVERY 16x16, 32x32 and 64x64 block of a regular block grid, in one launch ---------------------------
 * The blocks of a CU tree that share the displacement grid and the MV predictor: the SAD of a 32x32 / 64x64 block at a displacement is the exact
 * sum of the SADs of its 16x16 sub-blocks at that displacement (same rows under row sub-sampling), so each 16x16 SAD is computed ONCE and the
 * larger blocks are sums.  Per block the result equals vvcgpu_sad_search on that block with
 *   raster: dx0 = dy0 = -5 (raster_range / 5), nx = ny = 2 (raster_range / 5) + 1, sx = sy = 5;   dense: dx0 = dy0 = -dense_range, nx = ny = 2 dense_range + 1,
 * the same sub_shift and mvcost (cost, arg-min in visiting order, strict '<').
 * Grid: 16x16 block (i, j), i < n16x, j < n16y, has its origin at (org_x + 16 i, org_y + 16 j) in the original and its zero-vector position at
 * (ref_x + 16 i, ref_y + 16 j) in the reference plane; 32x32 block (i, j) covers 16x16 blocks (2i .. 2i+1, 2j .. 2j+1), i < n16x / 2, j < n16y / 2
 * (whole blocks only), 64x64 likewise with 4.  Results: raster_best[0 / 1 / 2] = arrays of n16x n16y / (n16x/2)(n16y/2) / (n16x/4)(n16y/4) records
 * for the 16 / 32 / 64 blocks, row-major; dense_best[] likewise (NULL with dense_range 0).  Arrays of sizes that have no block may be NULL.
 * Reads of the reference plane: exactly the samples the per-size searches of the existing blocks read (window of +-5 (raster_range / 5) around
 * every block).  Preconditions of the kernel: raster_step 5, raster_range <= 99 (39 x 39 positions), dense_range <= 4, sub_shift 0 or 1, org_stride
 * even, ref_stride a multiple of 8, org 4-byte and ref 16-byte aligned, 0 <= lambda < 4e6; anything else returns VVCGPU_E_UNSUPPORTED (nothing
 * launched) and the caller takes vvcgpu_sad_search per size.                                                                                  */
typedef struct vvcgpu_me_hier_cfg {
  int32_t org_x, org_y;       /* grid origin in the original plane */
  int32_t ref_x, ref_y;       /* zero-vector position of the grid origin in the reference plane */
  int32_t n16x, n16y;         /* 16x16 blocks of the grid */
  int32_t sub_shift;          /* rows step 1 << sub_shift (DistParam::subShift) */
  int32_t raster_range;       /* iSearchRange of the raster stage */
  int32_t raster_step;        /* iRaster: 5 */
  int32_t dense_range;        /* BipredSearchRange of xPatternSearch; 0 = no dense grid */
} vvcgpu_me_hier_cfg;
int vvcgpu_me_hier_search(const vvc_pel* org, int org_stride, const vvc_pel* ref, int ref_stride, const vvcgpu_me_hier_cfg* cfg_host,
                          const vvcgpu_mvcost* mvcost_host, vvcgpu_search_best* const* raster_best, vvcgpu_search_best* const* dense_best, void* stream);

/* ---- I1: interpolation filter table slots, batched  (InterpolationFilter::m_filterHor/m_filterVer[N][isFirst][isLast]
 *          and m_filterCopy[isFirst][isLast], InterpolationFilter.h:84-86; bodies InterpolationFilter.cpp:205-379;
 *          SIMD twins x86/InterpolationFilterX86.h:195-1125) ------------------------------------------------------
 * One descriptor per reference call.  taps = 8, 4 or 2 (filter) or 0 (filterCopy, coeff ignored).  src_off addresses
 * the first OUTPUT-aligned sample exactly as the reference's `src` argument does (the kernel steps back taps/2-1). */
typedef struct vvcgpu_if_desc {
  int64_t src_off, dst_off;           /* elements from src_base / dst_base */
  int32_t src_stride, dst_stride;
  int16_t w, h;
  int8_t  taps, is_vertical, is_first, is_last;
  int16_t coeff[8];
  int16_t reserved[4];                /* sizeof == 56 */
} vvcgpu_if_desc;
int vvcgpu_if_batch(const vvc_pel* src_base, vvc_pel* dst_base, const vvcgpu_if_desc* descs, int n,
                    int bit_depth, int clp_min, int clp_max, void* stream);

/* ---- I3 (+B1): motion compensation of prediction blocks, batched  (InterPrediction::xPredInterBlk,
 *          InterPrediction.cpp:480-547, uni: rndRes = true; bi: both lists with rndRes = false followed by
 *          PelBuf::addAvg, :743-760 / Buffer.cpp:114-151).  Affine (xPredAffineBlk :550-722) is the same call with
 *          one descriptor per 4x4 (2x2 chroma) sub-block and the sub-block MVs derived on the host. -----------------
 * ref*_off: element offset (from ref0_base / ref1_base) of the block position displaced by the INTEGER part of the MV
 *           (pu pos + (mv >> shift)); frac_x/frac_y in 1/16 (luma) or 1/32 (chroma) sample units as computed at
 *           :497-504; is_luma selects the 8-tap table m_lumaFilter[16][8] or the 4-tap m_chromaFilter[32][4].
 * bi = 0: dst = clipped uni-prediction from ref0.   bi = 1: dst = addAvg(pred(ref0), pred(ref1)).
 * bi = 2: dst = the unrounded 14-bit intermediate of ref0 (what motionCompensation leaves in m_acYuvPred).
 * Reads: the rows / columns the reference's branch reads ((N - 1) extra rows only when frac_y != 0, columns likewise), as whole ALIGNED
 * 16-byte words (16x16 luma PUs on the matrix-core path, reference strides that are multiples of 8 samples) or aligned dwords (everything else)
 * -- i.e. at most SEVEN samples left of and seven samples right of them IN THE SAME ROW and inside the aligned 16-byte words that hold them (the
 * x86 filters over-read as far: picture margins cover it; a plane whose rows start 16-byte aligned is never read outside its rows).  16x16 luma
 * and 8x8 chroma PUs with bi 0 / 1 and quarter- (chroma: eighth-) sample phases take the matrix-core path (both filter passes as exact f16
 * products); other phases, bi = 2 and reference samples outside the bit depth take the packed vector path; PUs that are grids of such tiles
 * walk it tile by tile; every result is bit-equal to the reference. */
typedef struct vvcgpu_mc_desc {
  int64_t ref0_off, ref1_off, dst_off;
  int32_t ref0_stride, ref1_stride, dst_stride;
  int16_t w, h;
  int8_t  frac_x0, frac_y0, frac_x1, frac_y1;
  int8_t  is_luma, bi;
  int16_t reserved;
} vvcgpu_mc_desc;
int vvcgpu_mc_batch(const vvc_pel* ref0_base, const vvc_pel* ref1_base, vvc_pel* dst_base,
                    const vvcgpu_mc_desc* descs, int n, int bit_depth, int clp_min, int clp_max, void* stream);
/* The same call for a list that is (mostly) 16x16 luma / 8x8 chroma PUs -- a picture's PU list under the common partition, as the motion
 * compensation of a whole picture hands it over (InterPrediction::motionCompensation per PU, InterPrediction.cpp:480-547): ONE launch.  vvcgpu_mc_batch
 * runs the matrix-core kernel and, behind it, a launch of the generic kernel for every descriptor the first one left (4.9 us per 4K picture when that
 * list is empty); here a descriptor the matrix-core kernel cannot take -- another shape, phase or stride, reference samples outside the bit depth -- is
 * served by the wavefront that found it, behind its walk, through the same generic body.  Any list is valid and every result is the one
 * vvcgpu_mc_batch gives; a list of mostly other shapes is slower here (they are served by the matrix-core kernel's waves one after the other). */
int vvcgpu_mc_picture_batch(const vvc_pel* ref0_base, const vvc_pel* ref1_base, vvc_pel* dst_base,
                            const vvcgpu_mc_desc* descs, int n, int bit_depth, int clp_min, int clp_max, void* stream);
/* "predict a candidate -> distortion against the original" in one pass: the cost of an AMVP candidate (InterSearch::xGetTemplateCost,
 * EncoderLib/InterSearch.cpp:1606-1640: motionCompensation of the candidate vector, then getDistPart(DF_SAD)) and of a merge candidate
 * (EncCu::xCheckRDCostMerge2Nx2N, EncoderLib/EncCu.cpp:1565-1592: motionCompensation of the candidate, then the Hadamard distParam.distFunc).
 * Descriptors as vvcgpu_mc_batch with bi = 0 or 1, EXCEPT that dst_off / dst_stride address the ORIGINAL block in org_base and `reserved` is the
 * row sub-sampling shift of the SAD (DistParam::subShift; 0 for the other kinds).  kind: 0 SAD, 1 Hadamard (xGetHADs), 2 SSE.  out[i] = what
 * vvcgpu_mc_batch followed by vvcgpu_dist_batch(kind) returns for the pair; the prediction stays in LDS.  w, h <= 128.  The vector bits of
 * the candidates (getCostOfVectorWithPredictor) and the candidate lists stay with the caller.  Descriptors live in device memory, so the library
 * cannot validate them on the host: a descriptor outside the contract (w or h outside 1..128, bi outside 0..1) is skipped and its out[i] is the
 * sentinel ~0 (UINT64_MAX); the same sentinel convention holds for vvcgpu_intra_satd_batch (w, h outside 1..64) and for the distortion output of
 * vvcgpu_affine_me_iter_batch (w, h outside 1..128).  The descriptor array of vvcgpu_mc_batch / vvcgpu_mc_dist_batch is 16-byte aligned.       */
int vvcgpu_mc_dist_batch(int kind, const vvc_pel* ref0_base, const vvc_pel* ref1_base, const vvc_pel* org_base, const vvcgpu_mc_desc* descs, int n,
                         int bit_depth, int clp_min, int clp_max, uint64_t* out, void* stream);

/* ---- B1-B4: PelBuffer element-wise operations, batched  (g_pelBufOP table, Buffer.h:57-73: addAvg4/8, reco4/8,
 *          linTf4/8; cores Buffer.cpp:50-94; plus AreaBuf::subtract Buffer.h:321-339, removeHighFreq :389-416,
 *          copyClip Buffer.cpp:197-222) --------------------------------------------------------------------------
 * op: 0 addAvg   dst = clip((s0 + s1 + offset) >> shift)
 *     1 reco     dst = clip(s0 + s1)                      (dst may alias s0, DecCu.cpp:188)
 *     2 linTf    dst = (scale*s0 >> shift) + offset, clipped when clip != 0   (shift < 0 shifts left)
 *     3 subtract dst = s0 - s1
 *     4 removeHighFreq  dst = 2*s0 - s1, clipped when clip != 0
 *     5 copyClip dst = clip(s0)                                                                           */
typedef struct vvcgpu_pelop_desc {
  int64_t src0_off, src1_off, dst_off;
  int32_t src0_stride, src1_stride, dst_stride;
  int16_t w, h;
} vvcgpu_pelop_desc;
typedef struct vvcgpu_pelop_cfg { int32_t scale, shift, offset, clip, clp_min, clp_max; } vvcgpu_pelop_cfg;
int vvcgpu_pelop_batch(int op, const vvc_pel* src0_base, const vvc_pel* src1_base, vvc_pel* dst_base,
                       const vvcgpu_pelop_desc* descs, int n, const vvcgpu_pelop_cfg* cfg_host, void* stream);

/* ---- T1/T2/T3: 2-D separable transforms, batched  (free functions xTrMxN_EMT / xITrMxN_EMT, TrQuant.cpp:138-310, as
 *          called by TrQuant::xT / xIT :694-791 through the fastFwdTrans/fastInvTrans tables :72-84; transform skip
 *          xTransformSkip / xITransformSkip :795-847, 1112-1163) ------------------------------------------------------
 * tr_hor / tr_ver: 0 DCT-II (sizes 2..64), 1 DCT-VIII, 2 DST-VII (sizes 4..32) (TransType, TypeDef.h:402-410);
 * tr_hor = 3 selects transform skip (tr_ver ignored).  Coefficients are W x H int32, contiguous (stride W), as the
 * reference's CoeffBuf.  Zero-out as with useQTBT/m_rectTUs: columns/rows >= 32 are not computed and written as 0
 * (forward, TrQuant.cpp:157-162) / not read (inverse, :755-759).  maxLog2TrDynamicRange is 15 (no extended precision).
 * The integer matrices are the reference's run-time generated tables shipped as golden data (csrc/tr_tables.inc).   */
typedef struct vvcgpu_tr_desc {
  int64_t resi_off, coeff_off;          /* elements from resi_base (Pel) / coeff_base (TCoeff) */
  int32_t resi_stride;
  int16_t w, h;
  int8_t  tr_hor, tr_ver;
  int16_t reserved;
  int32_t reserved2;                    /* sizeof == 32 */
} vvcgpu_tr_desc;
int vvcgpu_tr_fwd_batch(const vvc_pel* resi_base, vvc_coef* coeff_base, const vvcgpu_tr_desc* descs, int n,
                        int bit_depth, void* stream);
int vvcgpu_tr_inv_batch(const vvc_coef* coeff_base, vvc_pel* resi_base, const vvcgpu_tr_desc* descs, int n,
                        int bit_depth, void* stream);
/* ---- N1 (first "next" row): de-quantisation fused with the inverse transform at the ABI
 *          (TrQuant::invTransformNxN = m_quant->dequant + xIT, TrQuant.cpp:559-571; Quant::dequant, Quant.cpp:277-428 with
 *          flat scaling; dependent quantisation DQIntern::Quantizer::dequantBlock, DepQuant.cpp:708-785) ---------------
 * One descriptor per TU; binary compatible with vvcgpu_tr_desc (level_off sits where coeff_off does).  level_base holds
 * the entropy-decoded levels, W x H int32 contiguous.  qp = QpParam::Qp of the component (bit-depth offset included).
 * dep_quant = 1 replays the 4-state machine over the diagonal 4x4-grouped scan (state transitions 32040, :782).
 * ONE launch: the de-quantiser and the inverse transform of a TU run in the same wave, the de-quantised coefficients stay in LDS.
 * coeff_out (optional, may be NULL; same offsets as level_base; the reference's m_plTempCoeff) additionally receives the de-quantised
 * coefficients.  Intermediate products are formed in 64 bits (the reference's `int` cannot overflow for levels in
 * the entropy-coding range +-32768, which is the precondition).                                                          */
typedef struct vvcgpu_dqtr_desc {
  int64_t resi_off, level_off;          /* elements from resi_base (Pel) / level_base (TCoeff) */
  int32_t resi_stride;
  int16_t w, h;
  int8_t  tr_hor, tr_ver;               /* as vvcgpu_tr_desc; tr_hor = 3: transform skip */
  int8_t  dep_quant;                    /* 0: Quant::dequant, 1: dependent quantisation */
  int8_t  reserved;
  int32_t qp;                           /* sizeof == 32 */
} vvcgpu_dqtr_desc;
int vvcgpu_dequant_tr_inv_batch(const vvc_coef* level_base, vvc_pel* resi_base, const vvcgpu_dqtr_desc* descs, int n,
                                int bit_depth, vvc_coef* coeff_out, void* stream);
/* ---- fused residual chain of one TU  (InterSearch::xEstimateInterResidualQT, EncoderLib/InterSearch.cpp:4254-4504:
 *          residual = org - pred (CodingStructure::getResiBuf / subtract, CommonLib/Buffer.h:321-339), TrQuant::transformNxN :4409
 *          = xT (TrQuant.cpp:694-739) + Quant::quant (Quant.cpp:721-834, sign bit hiding :142-273), TrQuant::invTransformNxN :4497
 *          = Quant::dequant (Quant.cpp:277-428) + xIT (TrQuant.cpp:743-791), reconstruction clip(pred + resi') (Buffer.cpp:66-79)) ----
 * What vvcgpu_pelop_batch(subtract) -> vvcgpu_tr_fwd_batch -> vvcgpu_quant_batch -> vvcgpu_dequant_tr_inv_batch -> vvcgpu_pelop_batch(reco)
 * compute for the same TUs, in ONE pass: the residual, the coefficients and the de-quantised coefficients never leave the chip; per TU the
 * levels (W x H int32 contiguous at level_off, zero outside the kept 32 x 32 low-frequency region), their absolute sum (abs_sum[i], as
 * Quant::quant's uiAbsSum) and the reconstructed samples are written once.  Bit-exact with that sequence (tests/test_gpu_resichain.py).
 * Preconditions: org / pred samples within the bit depth (|residual| <= 1023), tr_hor / tr_ver in 0..2 (transform skip and RDPCM TUs go
 * through the separate entry points), qp as vvcgpu_quant_desc.  TUs with both sides in 16 / 32 / 64 run their four 1-D stages on the matrix cores
 * (v_mfma_f32_16x16x32_f16 on exact integer limbs), one wave per TU; a 16- / 32- / 64-point side with an 8- or 4-point one as well, two or four
 * TUs packed into one multi-tile with block-diagonal short stages; 8x8 / 8x4 / 4x8 / 4x4 in lane groups; 2-wide (chroma) TUs on a generic
 * wave-per-TU path.  Results do not depend on the path.                                                                                      */
typedef struct vvcgpu_resi_chain_desc {
  int64_t org_off, pred_off, rec_off;   /* elements from org_base / pred_base / rec_base (Pel) */
  int64_t level_off;                    /* elements from level_base (TCoeff) */
  int32_t org_stride, pred_stride, rec_stride;
  int16_t w, h;                         /* 2..64, powers of two */
  int8_t  tr_hor, tr_ver;               /* 0 DCT-II, 1 DCT-VIII, 2 DST-VII */
  int8_t  intra_slice;                  /* quantiser rounding offset 171 (I slice) or 85, as vvcgpu_quant_desc */
  int8_t  sign_hiding;
  int32_t qp;                           /* QpParam::Qp of the component (bit-depth offset included) */
  int32_t reserved[2];                  /* sizeof == 64 */
} vvcgpu_resi_chain_desc;
int vvcgpu_resi_chain_batch(const vvc_pel* org_base, const vvc_pel* pred_base, vvc_pel* rec_base, vvc_coef* level_base,
                            const vvcgpu_resi_chain_desc* descs, int n, int bit_depth, int clp_min, int clp_max, uint32_t* abs_sum,
                            void* stream);
/* The same for descriptors the caller has GROUPED BY SHAPE (an encoder knows its TU shapes when it builds the list): runs_host holds n_runs triples
 * (w, h, count) -- descs[] is run 0's TUs, then run 1's, ...; every descriptor of a run has the run's w x h; a shape appears in at most one run.  The
 * library then needs no classification pass over the list (one launch fewer: 8.7 us of a 3840x2160 picture).  Same outputs, same preconditions; in
 * addition the descriptors must be valid (tr_hor / tr_ver in 0..2, DCT-II on a 64-point side): the un-grouped entry marks an invalid TU with
 * abs_sum = 0xFFFFFFFF and skips it, this one does not look.  A call that holds a shape with a side of 2 is served as vvcgpu_resi_chain_batch.     */
int vvcgpu_resi_chain_runs_batch(const vvc_pel* org_base, const vvc_pel* pred_base, vvc_pel* rec_base, vvc_coef* level_base,
                                 const vvcgpu_resi_chain_desc* descs, int n, const int32_t* runs_host, int n_runs,
                                 int bit_depth, int clp_min, int clp_max, uint32_t* abs_sum, void* stream);
/* ---- T3: residual DPCM of transform-skipped / lossless TUs  (TrQuant::applyForwardRDPCM, CommonLib/TrQuant.cpp:991-1045, with
 *          Quant::transformSkipQuantOneSample / invTrSkipDeQuantOneSample, Quant.cpp:911-1090; TrQuant::invRdpcmNxN, TrQuant.cpp:632-688).
 *          A range-extension tool (CU::isRDPCMEnabled, UnitTools.cpp:105-108): off in the shipped cfgs, here for completeness of row T3. --------
 * mode: 0 off, 1 horizontal, 2 vertical (RDPCMMode).  fwd: per sample delta = residual - reconstructed running sum of its line, level =
 * transform-skip quantiser of the delta (rounding offset 256 when mode != 0, else 171 / 85 by slice type), coefficient index reversed when
 * `rotate` (TU::isNonTransformedResidualRotated); lossless: level = delta.  abs_sum[i] = sum |level| (32-bit, as the reference's uiAbsSum).
 * inv: in place, every sample after the first of a line becomes the clipped running sum of the line.                                        */
typedef struct vvcgpu_rdpcm_desc {
  int64_t resi_off, coeff_off;          /* elements from resi_base (Pel) / coeff_base (TCoeff, W x H contiguous) */
  int32_t resi_stride;
  int16_t w, h;
  int8_t  mode, lossless, rotate, intra_slice;
  int32_t qp;                           /* QpParam::Qp */
  int32_t reserved;                     /* sizeof == 40 */
} vvcgpu_rdpcm_desc;
int vvcgpu_rdpcm_fwd_batch(const vvc_pel* resi_base, vvc_coef* coeff_base, const vvcgpu_rdpcm_desc* descs, int n, int bit_depth, uint32_t* abs_sum,
                           void* stream);
int vvcgpu_rdpcm_inv_batch(vvc_pel* resi_base, const vvcgpu_rdpcm_desc* descs, int n, void* stream);
/* ---- I3: affine sub-block motion vectors on the device  (InterPrediction::xPredAffineBlk, CommonLib/InterPrediction.cpp:550-722: the per
 *          4x4 (chroma 2x2) sub-block vector from the control-point vectors, roundAffineMv (Mv.cpp:56-61), the clip of :675-676, the split into
 *          integer position and 1/16 (chroma 1/32) phase :681-701) -------------------------------------------------------------------------
 * One vvcgpu_affine_pu per PU; the kernel writes one vvcgpu_mc_desc per sub-block (w/4 x h/4 of them, row-major, starting at first_desc), ready for
 * vvcgpu_mc_batch: the caller no longer derives sub-block vectors on the host.  Control-point vectors are in 1/16 luma sample units (Mv::setHighPrec).
 * comp 0: luma descriptors (4x4), 1: chroma descriptors (2x2, phase in 1/32).  ref_origin: position of picture sample (0,0) inside the reference
 * plane of that component (its margin), so that ref offsets address the padded plane.  bi = 1: both lists (ref0 / ref1), else list 0 only
 * (bi field of the descriptors: 1 / 0).                                                                                                     */
typedef struct vvcgpu_affine_pu {
  int32_t pos_x, pos_y;                 /* luma position of the PU */
  int16_t w, h;                         /* luma size */
  int16_t six_param, bi;
  int32_t mv[2][3][2];                  /* [list][LT, RT, LB][hor, ver] */
  int64_t dst_off;                      /* elements from dst_base of the component */
  int32_t dst_stride, first_desc;       /* sizeof == 80 */
} vvcgpu_affine_pu;
int vvcgpu_affine_subblock_descs(const vvcgpu_affine_pu* pus, int n, int comp, int pic_w, int pic_h, int max_cu_w, int max_cu_h,
                                 int ref_origin_x, int ref_origin_y, int ref0_stride, int ref1_stride, vvcgpu_mc_desc* out, void* stream);
/* The whole of xPredAffineBlk for a list of PUs in one call: vvcgpu_affine_subblock_descs into subblock_ws (n_subblocks = sum of the PUs' sub-block
 * counts, device memory the call overwrites), then their interpolation as vvcgpu_mc_batch does it -- for luma with four 4x4 sub-blocks per wavefront
 * through the packed filter code (a list of 4x4 descriptors handed to vvcgpu_mc_batch itself goes one sub-block per wavefront: 0.6 ms instead of
 * 0.15 for the 518 k sub-blocks of a 4K picture).  Arguments as the two calls it bundles; ref1_base may be NULL when no PU has bi = 1.          */
int vvcgpu_affine_pred_batch(const vvc_pel* ref0_base, const vvc_pel* ref1_base, vvc_pel* dst_base, const vvcgpu_affine_pu* pus, int n, int n_subblocks,
                             vvcgpu_mc_desc* subblock_ws, int comp, int pic_w, int pic_h, int max_cu_w, int max_cu_h, int ref_origin_x, int ref_origin_y,
                             int ref0_stride, int ref1_stride, int bit_depth, int clp_min, int clp_max, void* stream);
/* ---- picture-level forms of the in-loop entry points: the three planes of a 4:2:0 picture in ONE launch each.  A chroma plane of a 4K picture is
 *          about one workgroup per CU, so a launch of its own costs its latency floor; these are also the forms a binding uses that keeps the
 *          reconstruction resident on the device across loopFilterPic -> SAOProcess -> ALFProcess (EncGOP.cpp:2122-2153, DecLib.cpp:506-533).
 * Same arithmetic and preconditions as the per-plane entry points they bundle (vvcgpu_sao_apply, vvcgpu_sao_stats, vvcgpu_alf_filter_luma /
 * _chroma, vvcgpu_alf_stats); width / height are the luma size, chroma planes are half size, ctu_size is the luma CTU size.            */
typedef struct vvcgpu_planes { vvc_pel* p[3]; int32_t stride[3]; } vvcgpu_planes;     /* Y, Cb, Cr: device pointers, strides in samples */
int vvcgpu_sao_apply_picture(const vvcgpu_planes* src, const vvcgpu_planes* dst, int width, int height, int ctu_size, int bit_depth,
                             const vvcgpu_sao_ctu* params_y, const vvcgpu_sao_ctu* params_cb, const vvcgpu_sao_ctu* params_cr,
                             int clp_min, int clp_max, void* stream);
/* out_y / out_cb / out_cr as vvcgpu_sao_stats' out (nCtu x 5 x 2 x 32 int64 each); skip lines: luma (r, b), chroma (r, b) */
int vvcgpu_sao_stats_picture(const vvcgpu_planes* org, const vvcgpu_planes* rec, int width, int height, int ctu_size, int bit_depth,
                             const uint8_t* avail, int skip_r_luma, int skip_b_luma, int skip_r_chroma, int skip_b_chroma,
                             int64_t* out_y, int64_t* out_cb, int64_t* out_cr, void* stream);
/* luma with the per-4x4 classifier and filter_type (0: 5x5, 1: 7x7), chroma always 5x5 with one filter; enable_* may be NULL (all CTUs on) */
int vvcgpu_alf_filter_picture(const vvcgpu_planes* src, const vvcgpu_planes* dst, int width, int height, int ctu_size, const uint16_t* cls,
                              int filter_type, const int16_t* luma_coeff_host, const int16_t* chroma_coeff_host, const uint8_t* enable_y,
                              const uint8_t* enable_cb, const uint8_t* enable_cr, int clp_min, int clp_max, void* stream);
/* the four covariance sets EncAdaptiveLoopFilter::deriveStatsForFiltering builds per picture (EncAdaptiveLoopFilter.cpp:1317-1392): luma 7x7
 * (out7: nCtu x 25 x 183) and luma 5x5 (out5: nCtu x 25 x 57) per class, Cb and Cr 5x5 (nCtu x 1 x 57 each).  The 5x5 diamond is a sub-diamond of
 * the 7x7 one under every transposition, so the luma 5x5 set is gathered from the 7x7 sums instead of being accumulated a second time.   */
int vvcgpu_alf_stats_picture(const vvcgpu_planes* org, const vvcgpu_planes* rec, int width, int height, int ctu_size, const uint16_t* cls,
                             int64_t* out7, int64_t* out5, int64_t* out_cb, int64_t* out_cr, void* stream);
/* The encoder's ALF front end for a picture in ONE launch: ALFProcess derives the block classes of the SAO output
 * (AdaptiveLoopFilter::deriveClassification, EncAdaptiveLoopFilter.cpp:1218-1226) and then accumulates the covariances over them
 * (deriveStatsForFiltering, :1317-1392).  cls_out (one uint16 per 4x4 luma block, as vvcgpu_alf_classify writes it) is an OUTPUT here:
 * the CTU workgroups of vvcgpu_alf_stats_picture classify their blocks from the tile they hold anyway.  Results are those of
 * vvcgpu_alf_classify followed by vvcgpu_alf_stats_picture, bit for bit (CTU sizes other than 64 / 128 and pictures of fewer than 320 CTUs --
 * where a classifier launch of its own is the faster form -- run exactly these two).  */
int vvcgpu_alf_classify_stats_picture(const vvcgpu_planes* org, const vvcgpu_planes* rec, int width, int height, int ctu_size, int bit_depth,
                                      uint16_t* cls_out, int64_t* out7, int64_t* out5, int64_t* out_cb, int64_t* out_cr, void* stream);
/* The coefficient scan the library replays (host copy, out[scanIdx] = raster position; w, h in 2..64 powers of two). */
int vvcgpu_scan_order_host(int w, int h, uint16_t* out);
/* ---- N3 ("next" row): affine gradient search kernels  (AffineGradientSearch table slots m_HorizontalSobelFilter /
 *          m_VerticalSobelFilter / m_EqualCoeffComputer, AffineGradientSearch.h:50-54; bodies AffineGradientSearch.cpp:66-174;
 *          called per iteration of xAffineMotionEstimation, InterSearch.cpp:3456-3534) --------------------------------
 * sobel: deriv = 3x3 Sobel response of the W x H prediction block (horizontal: [-1 0 1; -2 0 2; -1 0 1], vertical its
 *        transpose); the outermost ring repeats the nearest interior value (:83-96, :116-129).  w, h >= 3.
 * equal_coeff: out[col+1][row] = sum iC[col]*iC[row], out[col+1][P] = sum (iC[col]*resi) << 3 over the block, P = 4 or 6
 *        parameters, iC built from the two derivative planes and the sample position (:139-157); out is n x 7 x 7 int64
 *        (row 0 and unused columns are written as 0; the reference ACCUMULATES into a zeroed matrix, the caller adds).
 *        The residue is indexed with the DERIVATIVE stride, as the reference does (:144 uses `idx` for both).           */
typedef struct vvcgpu_afg_desc {
  int64_t pred_off, deriv_off;          /* elements from pred_base (Pel) / deriv_base (int32) */
  int32_t pred_stride, deriv_stride;
  int16_t w, h;
  int32_t reserved;                     /* sizeof == 32 */
} vvcgpu_afg_desc;
int vvcgpu_affine_sobel_batch(int vertical, const vvc_pel* pred_base, int32_t* deriv_base, const vvcgpu_afg_desc* descs, int n,
                              void* stream);
typedef struct vvcgpu_afe_desc {
  int64_t resi_off, deriv_off;          /* elements from resi_base (Pel) / derivx_base, derivy_base (int32, same offset) */
  int32_t deriv_stride;                 /* also the stride of the residue block */
  int16_t w, h;
  int32_t six_param;                    /* 0: 4-parameter model, 1: 6-parameter model */
  int32_t reserved;                     /* sizeof == 32 */
} vvcgpu_afe_desc;
int vvcgpu_affine_equal_coeff_batch(const vvc_pel* resi_base, const int32_t* derivx_base, const int32_t* derivy_base,
                                    const vvcgpu_afe_desc* descs, int n, int64_t* out, void* stream);

/* One iteration of the affine gradient search behind its prediction (the loop body of InterSearch::xAffineMotionEstimation, InterSearch.cpp:3456-3534:
 * xPredAffineBlk with the current control-point vectors, error = org - pred, the two Sobel planes, the normal-equation sums, and the distortion of
 * that prediction for the cost check) in one call: sub-block vectors (as vvcgpu_affine_subblock_descs, luma, list 0 only: pu.bi must be 0), the
 * sub-block prediction (as vvcgpu_mc_batch, left in pred_base at pu.dst_off / pu.dst_stride), then ONE pass over the PU that writes neither a
 * residue nor a derivative plane.  PUs: both sides 16..128 (AFFINE_MIN_BLOCK_SIZE sub-blocks of 4x4).  For a bi-predictive search org_base holds the
 * caller's "2 org - other prediction" block, as in the reference.  pu.first_desc: index of the PU's first sub-block in subblock_ws (n_subblocks
 * entries = sum of (w / 4) (h / 4), device memory the call may overwrite).  coeff_out: n x 7 x 7 int64 as vvcgpu_affine_equal_coeff_batch;
 * dist_out (may be NULL): n x uint64, dist_kind 0 SAD / 1 Hadamard (what xAffineMotionEstimation's cost uses) of org against the prediction.
 * The caller solves the 4 x 4 / 6 x 6 system and updates the vectors on the host (double arithmetic, InterSearch.cpp:3536-3600).                    */
typedef struct vvcgpu_affine_iter {
  vvcgpu_affine_pu pu;
  int64_t org_off;                      /* elements from org_base */
  int32_t org_stride, reserved;         /* sizeof == 96 */
} vvcgpu_affine_iter;
int vvcgpu_affine_me_iter_batch(const vvc_pel* org_base, const vvc_pel* ref_base, vvc_pel* pred_base, const vvcgpu_affine_iter* items, int n,
                                int n_subblocks, vvcgpu_mc_desc* subblock_ws, int dist_kind, int pic_w, int pic_h, int max_cu_w, int max_cu_h,
                                int ref_origin_x, int ref_origin_y, int ref_stride, int bit_depth, int clp_min, int clp_max, int64_t* coeff_out,
                                uint64_t* dist_out, void* stream);

/* ---- N2 ("next" row): integer-sample TZ search of whole PUs, on the device  (InterSearch::xTZSearch,
 *          EncoderLib/InterSearch.cpp:1971-2252, with xTZSearchHelp :249-343, xTZ2PointSearch :349-374,
 *          xTZ8PointDiamondSearch :431-632, xSetSearchRange :1820-1883, clipMv CommonLib/Mv.cpp:64-80) -----------------
 * One wavefront walks one PU through the complete reference control flow (start point / zero vector / 2Nx2N predictor,
 * search range, first diamond rounds, zero neighbourhood, 2-point search, raster, star refinement); the up-to-16 candidates
 * of one diamond round (64 raster points) are evaluated together, and the arg-min keeps the reference's visiting order
 * and strict '<' rule, so position, cost and SAD are those of the sequential search.
 * Per PU: start_x/start_y = rcMv on entry (quarter units, before clipMv); pred2_x/pred2_y = *pIntegerMv2Nx2NPred (integer
 * units) when flag VVCGPU_TZ_PRED2; pos_x/pos_y = pu.cu->lumaPos(); pred_hor/pred_ver = m_mvPredictor (quarter units);
 * sub_shift = DistParam::subShift as set by RdCost::setDistParam for subShiftMode 0/2 (RdCost.cpp:256-283; mode 1 is
 * MESEARCH_SELECTIVE only and not served); VVCGPU_TZ_EXTENDED / VVCGPU_TZ_FAST = bExtendedSettings / bFastSettings.
 * cfg: lambda, cost_scale (2 in xMotionEstimation), imv_shift, search_range (m_iSearchRange), first_search_stop
 * (FastMEAssumingSmootherMV), picture and CTU size for clipMv, and the rectangle of reference samples that may be read
 * [ref_x0, ref_x1) x [ref_y0, ref_y1) in plane coordinates: the reference probes positions up to search_range / 2 beyond
 * the clipped range (zero-neighbourhood test) and relies on the picture margin; probes are clamped to the rectangle here,
 * so a too small margin gives a wrong SAD, never a fault; the plane allocation must extend 4 bytes beyond the rectangle's last sample
 * (reference rows are read as aligned dwords) and reference samples must be non-negative (picture samples).  Composite reference (JVET_K0157 inCtuSearch) and MR-SAD
 * (weighted prediction) are not served.  results: x, y = rcMv (integer units), cost = uiBestSad, sad = ruiSAD.          */
enum { VVCGPU_TZ_PRED2 = 1, VVCGPU_TZ_EXTENDED = 2, VVCGPU_TZ_FAST = 4 };
typedef struct vvcgpu_tz_pu {
  int32_t org_x, org_y, ref_x, ref_y;
  int32_t start_x, start_y, pred2_x, pred2_y;
  int32_t pos_x, pos_y, pred_hor, pred_ver;
  int16_t w, h, sub_shift, flags;
  int32_t reserved[2];                  /* reserved[0] > 0: this PU's own search range (m_aaiAdaptSR[list][refIdx]: the adaptive search range is per
                                           reference picture), 0: cfg.search_range; a batch may then mix the (list, reference) searches of one PU.  Must not exceed
                                           cfg.search_range (the caller passes the maximum over the batch there); a larger value is clamped to it.
                                           reserved[1]: 0.  sizeof == 64 */
} vvcgpu_tz_pu;
typedef struct vvcgpu_tz_cfg {
  double  lambda;
  int32_t cost_scale, imv_shift;
  int32_t search_range, first_search_stop;
  int32_t pic_w, pic_h, max_cu_w, max_cu_h;
  int32_t ref_x0, ref_y0, ref_x1, ref_y1;
  int32_t wg_per_pu;                    /* 0: one wavefront per PU (PUs up to about 32x32); 1: one workgroup of four per PU   */
  int32_t uniform_pu;                   /* 0: PUs of any size.  h << 16 | w (16 / 32 / 64 each): the caller states that the PUs of the batch are w x h with
                                           2:1 row sub-sampling; the raster stage (iRaster 5, InterSearch.cpp:2159-2169) then runs as its own launch
                                           between two launches of the search (a PU that does not match, or whose raster touches the border of the
                                           readable rectangle, keeps the one-launch form: results are identical either way).  sizeof == 64 */
} vvcgpu_tz_cfg;
int vvcgpu_tz_search_batch(const vvc_pel* org, int org_stride, const vvc_pel* ref, int ref_stride,
                           const vvcgpu_tz_pu* pus, int n, const vvcgpu_tz_cfg* cfg_host,
                           vvcgpu_search_best* results, void* stream);

/* ---- N4 ("next" row, picture-level passes): border extension and picture hash --------------------------------------------
 * vvcgpu_extend_border: Picture::extendPicBorder (CommonLib/Picture.cpp:996-1041) for one plane: every sample of the margin
 *   becomes the nearest picture sample (left/right columns replicated, then whole rows replicated upwards/downwards).
 *   plane points to sample (0,0) INSIDE the padded allocation; margins are in samples of this plane (the reference shifts the
 *   luma margin by the chroma scale, :1010-1011).
 * vvcgpu_picture_hash: the per-plane digests of CommonLib/PicYuvMD5.cpp -- method 1 (HASHTYPE_CRC): compCRC :83-125, CRC-16
 *   CCITT over the low byte then (bit depth > 8) the high byte of every sample in raster order, 16 zero bits appended;
 *   method 2 (HASHTYPE_CHECKSUM): compChecksum :143-169.  The CRC is linear over GF(2): every lane reduces 8 samples, lane and
 *   block remainders are shifted to their position by multiplication with x^n mod P and XORed.  out: one uint32 on the device
 *   (CRC in the low 16 bits).  method 0 (MD5, calcMD5 :181-207) is a serial chain of 64-byte blocks per plane and is NOT
 *   offered on the device (VVCGPU_E_UNSUPPORTED).                                                                          */
int vvcgpu_extend_border(vvc_pel* plane, int stride, int w, int h, int margin_x, int margin_y, void* stream);
int vvcgpu_picture_hash(int method, const vvc_pel* plane, int stride, int w, int h, int bit_depth, uint32_t* out, void* stream);

/* ---- N4 ("next" row): intra sample prediction  (IntraPrediction::predIntraAng, CommonLib/IntraPrediction.cpp:251-347 =
 *          xPredIntraPlanar :424-477 | xPredIntraDc :482-493 | xPredIntraAng :540-773 with wide-angle mapping :213-231, followed by
 *          the simplified PDPC; optional xFilterReferenceSamples :1071-1104 in front) ---------------------------------------
 * One descriptor per prediction block (a TU of the intra PU), any mix of sizes and modes in one call.  The caller gathers the
 * reference samples (xFillReferenceSamples, :807-1004: availability and substitution are control logic over the coding
 * structure) and hands them over PACKED:  refs[0] = top-left, refs[1 .. T] = the row above, refs[T + 1 .. T + L] = the column to
 * the left, with T / L = m_topRefLength / m_leftRefLength of setReferenceArrayLengths (:233-249; vvcgpu_intra_ref_lengths).
 * mode = PU::getFinalIntraMode (0 planar, 1 DC, 2..66 angular; the wide-angle remapping happens inside, as in the reference).
 * filter_refs = the decision of useFilteredIntraRefSamples (:1107-1150).  w, h: powers of two 4..64 (the reference predicts per
 * TU, <= 64; its DC divisor table g_aucLog2 ends at 128).  clp_min / clp_max: slice clip range of the component.  CCLM
 * (predIntraChromaLM, JVET_K0190) is not built.                                                                             */
typedef struct vvcgpu_intra_desc {
  int64_t ref_off, dst_off;             /* samples, relative to refs_base / dst_base */
  int32_t dst_stride;
  int16_t w, h;
  int8_t  mode, filter_refs;
  int16_t reserved;
  int32_t reserved2;                    /* sizeof == 32 */
} vvcgpu_intra_desc;
int vvcgpu_intra_ref_lengths(int w, int h, int* top_len, int* left_len);
int vvcgpu_intra_pred_batch(const vvc_pel* refs_base, vvc_pel* dst_base, const vvcgpu_intra_desc* descs, int n, int clp_min, int clp_max,
                            void* stream);

/* N4, intra mode pre-selection  (IntraSearch::estIntraPredLumaQT, EncoderLib/IntraSearch.cpp:397-480: for every candidate mode
 *          predIntraAng into the prediction buffer, then distParam.distFunc = the Hadamard distortion RdCost::xGetHADs against the
 *          original, :423-433 and the second round :470-480) -- one descriptor per (block, candidate mode): the prediction of
 *          vvcgpu_intra_pred_batch goes into LDS and only out[i] = xGetHADs(org, pred) (what vvcgpu_dist_batch kind 1 returns for the
 *          same two blocks) leaves the chip.  The mode bits and the candidate lists (xFracModeBitsIntra, updateCandList) stay with
 *          the caller.  refs as vvcgpu_intra_desc; org_off / org_stride: the block in the original plane.                            */
typedef struct vvcgpu_intra_satd_desc {
  int64_t ref_off, org_off;             /* samples, relative to refs_base / org_base */
  int32_t org_stride;
  int16_t w, h;                         /* powers of two 4..64 */
  int8_t  mode, filter_refs;
  int16_t reserved;
  int32_t reserved2;                    /* sizeof == 32 */
} vvcgpu_intra_satd_desc;
int vvcgpu_intra_satd_batch(const vvc_pel* refs_base, const vvc_pel* org_base, const vvcgpu_intra_satd_desc* descs, int n, int clp_min,
                            int clp_max, uint64_t* out, void* stream);

/* N4, reference sample gathering: IntraPrediction::xFillReferenceSamples (:807-1004) for the packed layout above.  rec_off: the
 * block's top-left sample in the reconstruction plane; flags_off: the reference's neighborFlags of the block in flags_base, one
 * byte per unit in chain order  below-left (bottom first) ... left ... top-left ... above ... above-right,
 * ceil(L / unit_h) + 1 + ceil(T / unit_w) entries (the availability walk over the coding structure, :853-858, stays with the
 * caller); unit_w / unit_h: pcv.minCUWidth / Height shifted by the component scale (:824-826).  Unavailable units are padded
 * exactly like the reference's line buffer; with nothing available every sample is 1 << (bit_depth - 1).  ref_off: where the
 * T + L + 1 packed samples go in refs_base -- feed them to vvcgpu_intra_pred_batch.                                         */
typedef struct vvcgpu_intra_fill_desc {
  int64_t rec_off, flags_off, ref_off;
  int32_t rec_stride;
  int16_t w, h;
  int8_t  unit_w, unit_h;
  int16_t reserved;
  int32_t reserved2;                    /* sizeof == 40 */
} vvcgpu_intra_fill_desc;
int vvcgpu_intra_fill_refs_batch(const vvc_pel* rec_base, const uint8_t* flags_base, vvc_pel* refs_base, const vvcgpu_intra_fill_desc* descs,
                                 int n, int bit_depth, void* stream);

/* N4, CCLM: cross-component linear model prediction of a chroma block  (IntraPrediction::xGetLumaRecPixels :1283-1581, the
 * JVET_K0190 branch, + xGetLMParameters :1597-1857 + predIntraChromaLM :390-403).  4:2:0 only.  luma_off: the co-located luma
 * block's top-left sample in the luma RECONSTRUCTION plane (rows -2, -1 are read when above_avail, columns -3 .. -1 when left_avail);
 * nb_off: reconstructed chroma neighbours of this component in nb_base, the w samples above followed by the h samples to the left
 * (what getPredictorPtr(compID) holds in row 0 / column 0); above_avail / left_avail: the reference's bAboveAvaillable /
 * bLeftAvaillable (ALL units of that side available, :1633-1637 -- control logic over the coding structure, decided by the
 * caller); w, h: chroma block, powers of two 2..64.                                                                        */
typedef struct vvcgpu_cclm_desc {
  int64_t luma_off, nb_off, dst_off;
  int32_t luma_stride, dst_stride;
  int16_t w, h;
  int8_t  above_avail, left_avail;
  int16_t reserved;
  int32_t reserved2[2];                 /* sizeof == 48 */
} vvcgpu_cclm_desc;
int vvcgpu_cclm_pred_batch(const vvc_pel* luma_base, const vvc_pel* nb_base, vvc_pel* dst_base, const vvcgpu_cclm_desc* descs, int n,
                           int bit_depth_luma, int bit_depth_chroma, int clp_min, int clp_max, void* stream);

/* N1, forward direction without RDOQ: scalar quantisation of transform coefficients  (Quant::quant, Quant.cpp:721-834: level =
 * (|c| * g_quantScales[qp % 6] * whScale + add) >> qBits with add = (intra slice ? 171 : 85) << (qBits - 9), flat scaling) and,
 * when the slice enables it, sign bit hiding per 4x4 coefficient group (xSignBitHidingHDQ :142-273, JVET_K0072 keeps it for
 * slices without dependent quantisation).  abs_sum[i] = uiAbsSum of TU i (sum of the magnitudes before hiding).  The rate-
 * distortion optimised quantisers (QuantRDOQ, the DepQuant trellis) walk CABAC context state sequentially and are NOT built.  */
typedef struct vvcgpu_quant_desc {
  int64_t coeff_off, level_off;         /* in elements of coeff_base / level_base; both blocks are w x h, row pitch w */
  int16_t w, h;                         /* powers of two 2..64 */
  int8_t  intra_slice, sign_hiding;
  int16_t reserved;
  int32_t qp;                           /* QpParam::Qp (bit-depth offset included) */
  int32_t reserved2;                    /* sizeof == 32 */
} vvcgpu_quant_desc;
int vvcgpu_quant_batch(const vvc_coef* coeff_base, vvc_coef* level_base, const vvcgpu_quant_desc* descs, int n, int bit_depth, uint32_t* abs_sum,
                       void* stream);

/* N1, the rate-distortion optimised quantiser of the default configuration: dependent quantisation as a 4-state trellis
 * (DQIntern::DepQuant::quant, CommonLib/DepQuant.cpp:1323-1391 with xDecideAndUpdate :1252-1320, State :861-1102, CommonCtx::update
 * :1104-1164, Quantizer::initQuantBlock / preQuantCoeff :647-706, :786-808).  The trellis is sequential along the scan but
 * independent between TUs.  The CABAC side enters as RATE TABLES: what DQIntern::RateEstimator (:335-485) derives from the
 * current context states -- last-position bits per column / row (incl. the cbf bit difference), coded-sub-block flag bits,
 * significance bits of the three state-dependent context sets, and the greater-than / parity bit sums -- in 2^-15 bit units
 * (SCALE_BITS); the caller fills one vvcgpu_dq_rates per (component, TU width x height class) and points TUs at it.
 * luma != 0 selects the luma template context offsets (:566-577).  lambda = Quant::m_dLambda of the component.
 * level_out receives the signed levels (row pitch w), abs_sum[i] the sum of absolute levels.  total_coeffs: extent of the
 * coefficient buffer the descriptors address (coeff_off + w * h <= total_coeffs, coeff_off a multiple of 16, blocks disjoint);
 * ws: device workspace of vvcgpu_depquant_workspace_bytes(total_coeffs, n) bytes, 16-byte aligned (trellis decisions and the
 * per-state level histories, both indexed by coeff_off).                                                                                                */
typedef struct vvcgpu_dq_rates {
  int32_t last_x[64], last_y[64];       /* m_lastBitsX / m_lastBitsY                                     */
  int32_t sig_sbb[2][2];                /* m_sigSbbFracBits[ctx].intBits[bin]                            */
  int32_t sig[3][18][2];                /* m_sigFracBits[ctxSet][ctx].intBits[bin]                       */
  int32_t gtx[21][7];                   /* m_gtxFracBits[ctx].bits[0..6]                                 */
} vvcgpu_dq_rates;
typedef struct vvcgpu_depquant_desc {
  int64_t coeff_off, level_off;         /* elements of coeff_base / level_base, blocks are w x h with row pitch w */
  double  lambda;
  int32_t qp;                           /* QpParam::Qp                                                   */
  int32_t rates_idx;                    /* index into the rates array                                    */
  int16_t w, h;                         /* powers of two 4..64 (2-wide chroma blocks are not served)     */
  int8_t  luma;
  int8_t  reserved[3];                  /* sizeof == 40 */
} vvcgpu_depquant_desc;
size_t vvcgpu_depquant_workspace_bytes(size_t total_coeffs, int n);
int vvcgpu_depquant_batch(const vvc_coef* coeff_base, vvc_coef* level_base, const vvcgpu_depquant_desc* descs, int n,
                          const vvcgpu_dq_rates* rates, int bit_depth, uint32_t* abs_sum, size_t total_coeffs, void* ws, size_t ws_bytes,
                          void* stream);

/* N1, the rate-distortion optimised quantiser used when dependent quantisation is off (DepQuant::quant hands the TU over,
 * CommonLib/DepQuant.cpp:1411-1421): QuantRDOQ::xRateDistOptQuant, CommonLib/QuantRDOQ.cpp:694-1409 with xGetCodedLevel :107-162, xGetICRate :235-313,
 * xGetRateLast :407-421, xGetErrScaleCoeff :482-506 and the JVET_K0072 template contexts (CommonLib/ContextModelling.h:135-219);
 * what QuantRDOQ::quant :652-690 dispatches to for blocks wider and higher than 2.  The decision chain of one TU is sequential (each
 * level changes the contexts of the following ones); TUs are independent.  All costs are IEEE doubles evaluated in the reference's
 * order.  The CABAC side enters as the fractional-bit tables (FracBitsAccess, 2^-15 bit units) the function reads:
 *   sig[ofs]      Ctx::SigFlag[chType]( ofs )                 (sigCtxIdAbs with state 0; 18 luma / 12 chroma offsets)
 *   par/gt1/gt2   Ctx::ParFlag[chType], Ctx::GtxFlag[2 + chType], Ctx::GtxFlag[chType] ( ctxOffsetAbs: 21 luma / 11 chroma offsets )
 *   sig_group[c]  Ctx::SigCoeffGroup[chType]( c )
 *   last_x/last_y the prefix tables lastBitsX / lastBitsY as built at :1172-1200
 *   cbf           Ctx::QtRootCbf() (luma of an inter CU at depth 0) or Ctx::QtCbf[compID]( CtxQtCbf ) :1134-1160
 * sign_hiding = slice->getSignDataHidingEnabledFlag().  lambda = Quant::m_dLambda.  Layout rules, abs_sum and the workspace as
 * for vvcgpu_depquant_batch (workspace: the per-coefficient cost / rate-delta arrays m_pdCostCoeff ... m_deltaU of the reference). */
typedef struct vvcgpu_rdoq_rates {
  int32_t sig[18][2];
  int32_t par[21][2], gt1[21][2], gt2[21][2];
  int32_t sig_group[2][2];
  int32_t last_x[14], last_y[14];
  int32_t cbf[2];                       /* sizeof == 784 */
} vvcgpu_rdoq_rates;
typedef struct vvcgpu_rdoq_desc {
  int64_t coeff_off, level_off;         /* elements of coeff_base / level_base, blocks are w x h with row pitch w */
  double  lambda;
  int32_t qp;                           /* QpParam::Qp                                                   */
  int32_t rates_idx;                    /* index into the rates array                                    */
  int16_t w, h;                         /* powers of two 4..64                                           */
  int8_t  luma, sign_hiding;
  int8_t  reserved[2];                  /* sizeof == 40 */
} vvcgpu_rdoq_desc;
size_t vvcgpu_rdoq_workspace_bytes(size_t total_coeffs, int n);
int vvcgpu_rdoq_batch(const vvc_coef* coeff_base, vvc_coef* level_base, const vvcgpu_rdoq_desc* descs, int n,
                      const vvcgpu_rdoq_rates* rates, int bit_depth, uint32_t* abs_sum, size_t total_coeffs, void* ws, size_t ws_bytes,
                      void* stream);

/* The shipped matrix [type][log2(N)-1] as N x N int16 (host copy; for the shim's table check against initROM()). */
const int16_t* vvcgpu_tr_matrix_host(int type, int n);

/* ---- I2 (+D2, D5): fractional-sample refinement of a PU, fused  (InterSearch::xPatternSearchFracDIF,
 *          EncoderLib/InterSearch.cpp:2503-2552 = xExtDIFUpSamplingH :3813-3869 -> xPatternRefinement(half) :634-689 ->
 *          xExtDIFUpSamplingQ :3882-4093 -> xPatternRefinement(quarter)) ------------------------------------------------
 * The reference materialises up to 12 fractional planes (m_filteredBlock[4][4]) per PU and reference picture and then
 * evaluates 9 half-sample and 9 quarter-sample candidates with distFunc (Hadamard when HadamardME is on) + MV cost.
 * Here the planes never exist in HBM: every candidate block is interpolated in LDS (horizontal first-stage filter, then
 * vertical last-stage filter, the exact stage flags of the reference) and consumed by the distortion directly.
 * All blocks of one call share w, h.  ref_x/ref_y: block position displaced by the integer MV; mv_x/mv_y: that integer
 * MV (rcMvInt, integer-sample units) for the MV cost.  mvcost_host: lambda and predictor (m_mvPredictor, quarter units);
 * cost_scale/imv_shift are ignored (the reference uses scale 1 for the half stage and 0 for the quarter stage, shift 0).
 * result: half_x/half_y in {-1,0,1} (rcMvHalf), qter_x/qter_y in {-1,0,1} (rcMvQter), cost = ruiCost of the quarter stage,
 * cost_half = best cost of the half stage.  Candidate order and strict '<' tie rule as s_acMvRefineH/Q (:59-83).        */
typedef struct vvcgpu_frac_blk { int32_t org_x, org_y, ref_x, ref_y, mv_x, mv_y; } vvcgpu_frac_blk;
typedef struct vvcgpu_frac_result { int32_t half_x, half_y, qter_x, qter_y; uint64_t cost_half, cost; } vvcgpu_frac_result;
int vvcgpu_frac_refine(const vvc_pel* org, int org_stride, const vvc_pel* ref, int ref_stride,
                       const vvcgpu_frac_blk* blocks, int nblocks, int w, int h, int bit_depth, int clp_min, int clp_max,
                       int use_hadamard, const vvcgpu_mvcost* mvcost_host, vvcgpu_frac_result* results, void* stream);

/* N2, AMVR: integer / 4-sample refinement of the integer search result with both AMVP candidates
 * (InterSearch::xPatternSearchIntRefine, InterSearch.cpp:2408-2501; taken instead of the fractional refinement when cu.imv != 0).
 * 9 positions (the input vector and its 8 neighbours at distance 1 << imv_shift) x num_cand predictors; distortion = SATD
 * (use_hadamard: getUseHADME() && !transQuantBypass) or SAD at the clipped position, times `weight` in double precision and
 * truncated (:2456), + the MV cost against that predictor; strict '<' in visiting order.  Per PU: mv = rcMv on entry (integer
 * units), cand = amvpInfo.mvCand[0..1] (quarter units), idx_cost = m_auiMVPIdxCost[0..1][AMVP_MAX_NUM_CANDS], mvp_idx / bits =
 * riMVPIdx / ruiBits on entry.  Result: mv (quarter units), mvp_idx, bits = ruiBits, cost = ruiCost on exit.  cfg: the fields
 * lambda, imv_shift (1 or 2 + ...: 2 for integer, 4 for 4-sample in quarter units), picture geometry and readable rectangle of
 * vvcgpu_tz_cfg are used.                                                                                                  */
typedef struct vvcgpu_imv_pu {
  int32_t org_x, org_y, ref_x, ref_y;
  int32_t mv_x, mv_y;
  int32_t cand_x[2], cand_y[2];
  int32_t pos_x, pos_y;
  uint32_t idx_cost[2];
  uint32_t bits;
  int16_t w, h;
  int8_t  num_cand, mvp_idx;
  int16_t reserved;
  int32_t reserved2;                    /* sizeof == 72 */
} vvcgpu_imv_pu;
typedef struct vvcgpu_imv_result { int32_t mv_x, mv_y, mvp_idx; uint32_t bits; uint64_t cost; } vvcgpu_imv_result;
int vvcgpu_imv_refine_batch(const vvc_pel* org, int org_stride, const vvc_pel* ref, int ref_stride, const vvcgpu_imv_pu* pus, int n,
                            const vvcgpu_tz_cfg* cfg_host, int use_hadamard, double weight, vvcgpu_imv_result* results, void* stream);

/* N2, chained: integer TZ search followed by the fused fractional refinement (I2) of the same PUs, on one stream with no host
 * round trip -- the device form of InterSearch::xMotionEstimation's  xPatternSearchFast -> xPatternSearchFracDIF  sequence
 * (InterSearch.cpp:1775-1816).  All PUs of one call share w x h (the refinement kernel's rule); every PU keeps its own
 * predictor (pred_hor / pred_ver) in both stages.  int_results as vvcgpu_tz_search_batch, frac_results as vvcgpu_frac_refine;
 * the final vector is  (int << 2) + (half << 1) + qter  (:1813-1815).                                                     */
int vvcgpu_me_batch(const vvc_pel* org, int org_stride, const vvc_pel* ref, int ref_stride, const vvcgpu_tz_pu* pus, int n, int w, int h,
                    const vvcgpu_tz_cfg* cfg_host, int bit_depth, int clp_min, int clp_max, int use_hadamard,
                    vvcgpu_search_best* int_results, vvcgpu_frac_result* frac_results, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VVCGPU_H */
