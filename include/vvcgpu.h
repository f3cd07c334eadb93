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

#ifdef __cplusplus
}
#endif
#endif /* VVCGPU_H */
