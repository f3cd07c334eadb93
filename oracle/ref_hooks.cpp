// oracle/ref_hooks.cpp -- TEST INFRASTRUCTURE (part of the reference drop-in harness, see vtm_hip_shim.cpp).
// EncSampleAdaptiveOffset::getStatistics and EncAdaptiveLoopFilter::deriveStatsForFiltering are called from inside their own
// translation units (EncSampleAdaptiveOffset.cpp:227, EncAdaptiveLoopFilter.cpp:260), where GNU ld --wrap does not reach.  The
// reference objects are position independent, so those calls are PLT calls to interposable symbols: this tiny library, loaded
// with RTLD_GLOBAL ahead of libvtmref_hip.so, defines the two symbols, offers the call to the GPU shim and otherwise forwards
// to the reference's own definition (looked up in the handle the driver passes to vtmhooks_set_target).  No reference header is
// needed: `this` and references are pointers.
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>

#define SAO_SYM "_ZN23EncSampleAdaptiveOffset13getStatisticsERSt6vectorIPP11SAOStatDataSaIS3_EER7UnitBufIsES9_R15CodingStructureb"
#define ALF_SYM "_ZN21EncAdaptiveLoopFilter23deriveStatsForFilteringER7UnitBufIsES2_"
// the 2-D transforms are free functions called from TrQuant::xT / xIT in their own translation unit (TrQuant.cpp:694-791)
#define TRF_SYM "_Z10xTrMxN_EMTiPKsmPiiiihhb"
#define TRI_SYM "_Z11xITrMxN_EMTiPKiPsmiijjihh"
// the full integer search (FastSearch 0), called from xMotionEstimation in its own translation unit
#define FS_SYM "_ZN11InterSearch14xPatternSearchERNS_17IntTZSearchStructER2MvRm"
// the integer TZ search, called from xPatternSearchFast / xMotionEstimation in its own translation unit (InterSearch.cpp:1764, 1962)
#define TZ_SYM "_ZN11InterSearch9xTZSearchERK14PredictionUnitRNS_17IntTZSearchStructER2MvRmPKS5_bb"
// the picture hashes, called from calcAndPrintHashStatus in their own translation unit (PicYuvMD5.cpp:228-240) and from the encoder's SEI writer
#define CRC_SYM "_Z7calcCRCRK7UnitBufIKsER11PictureHashRK9BitDepths"
#define SUM_SYM "_Z12calcChecksumRK7UnitBufIKsER11PictureHashRK9BitDepths"
// the dependent-quantisation trellis: a virtual function, reached through the vtable (dynamic relocation against this symbol)
#define DQ_SYM "_ZN8DepQuant5quantER13TransformUnitRK11ComponentIDRK7AreaBufIKiERiRK7QpParamRK3Ctx"
#define RQ_SYM "_ZN9QuantRDOQ5quantER13TransformUnitRK11ComponentIDRK7AreaBufIKiERiRK7QpParamRK3Ctx"
// the fractional motion refinement, called from xMotionEstimation in its own translation unit (InterSearch.cpp:1816)
#define FRAC_SYM "_ZN11InterSearch21xPatternSearchFracDIFERK14PredictionUnit10RefPicListiRNS_17IntTZSearchStructERK2MvRS6_S9_Rm"

// the two deblocking sample filters, called per CU edge from LoopFilter::xDeblockCU in their own translation unit (LoopFilter.cpp:340-347)
#define EFL_SYM "_ZN10LoopFilter15xEdgeFilterLumaERK10CodingUnit14DeblockEdgeDiri"
#define EFC_SYM "_ZN10LoopFilter17xEdgeFilterChromaERK10CodingUnit14DeblockEdgeDiri"

extern "C" {
typedef void (*ef_real_t)(void*, const void*, int, int);
typedef int (*ef_shim_t)(void*, const void*, int, int, int);
void hook_edge_luma(void* self, const void* cu, int dir, int edge) asm(EFL_SYM);
void hook_edge_chroma(void* self, const void* cu, int dir, int edge) asm(EFC_SYM);
typedef void (*sao_real_t)(void*, void*, void*, void*, void*, bool);
typedef int (*sao_shim_t)(void*, void*, void*, void*, void*, bool);
typedef void (*alf_real_t)(void*, void*, void*);
typedef int (*alf_shim_t)(void*, void*, void*);

typedef void (*trf_real_t)(int, const short*, size_t, int*, int, int, int, unsigned char, unsigned char, bool);
typedef int (*trf_shim_t)(int, const short*, size_t, int*, int, int, int, unsigned char, unsigned char, bool);
typedef void (*tri_real_t)(int, const int*, short*, size_t, int, int, unsigned, unsigned, int, unsigned char, unsigned char);
typedef int (*tri_shim_t)(int, const int*, short*, size_t, int, int, unsigned, unsigned, int, unsigned char, unsigned char);
void hook_tr_fwd(int bd, const short* resi, size_t stride, int* coeff, int w, int h, int maxLog2, unsigned char mode, unsigned char idx, bool qtbt) asm(TRF_SYM);
void hook_tr_inv(int bd, const int* coeff, short* resi, size_t stride, int w, int h, unsigned skipW, unsigned skipH, int maxLog2, unsigned char mode, unsigned char idx) asm(TRI_SYM);
typedef void (*frac_real_t)(void*, void*, int, int, void*, void*, void*, void*, void*);
typedef int (*frac_shim_t)(void*, void*, int, int, void*, void*, void*, void*, void*);
void hook_frac(void* self, void* pu, int list, int refIdx, void* cStruct, void* mvInt, void* mvHalf, void* mvQter, void* cost) asm(FRAC_SYM);
typedef void (*fs_real_t)(void*, void*, void*, void*);
typedef int (*fs_shim_t)(void*, void*, void*, void*);
void hook_fullsearch(void* self, void* cStruct, void* mv, void* sad) asm(FS_SYM);
typedef void (*tz_real_t)(void*, void*, void*, void*, void*, const void*, bool, bool);
typedef int (*tz_shim_t)(void*, void*, void*, void*, void*, const void*, bool, bool);
void hook_tzsearch(void* self, void* pu, void* cStruct, void* mv, void* sad, const void* pred2, bool ext, bool fast) asm(TZ_SYM);
typedef unsigned (*hash_real_t)(const void*, void*, const void*);
typedef int (*hash_shim_t)(int, const void*, void*, const void*);
unsigned hook_crc(const void* pic, void* digest, const void* bitDepths) asm(CRC_SYM);
unsigned hook_checksum(const void* pic, void* digest, const void* bitDepths) asm(SUM_SYM);
typedef void (*dq_real_t)(void*, void*, const void*, const void*, void*, const void*, const void*);
typedef int (*dq_shim_t)(void*, void*, const void*, const void*, void*, const void*, const void*);
void hook_depquant(void* self, void* tu, const void* compID, const void* src, void* absSum, const void* qp, const void* ctx) asm(DQ_SYM);
void hook_rdoq(void* self, void* tu, const void* compID, const void* src, void* absSum, const void* qp, const void* ctx) asm(RQ_SYM);
void hook_sao_stats(void* self, void* blkStats, void* org, void* src, void* cs, bool pre) asm(SAO_SYM);
void hook_alf_stats(void* self, void* org, void* rec) asm(ALF_SYM);

static void* g_target = nullptr;                 // dlopen handle of libvtmref_hip.so (holds the reference's own definitions)
void vtmhooks_set_target(void* handle) { g_target = handle; }
static void* must(void* p, const char* what) { if (!p) { fprintf(stderr, "ref_hooks: %s not found\n", what); abort(); } return p; }

// (1 = the shim recorded the edge for the device-side filter, 0 = run the reference's own sample filter)
void hook_edge_luma(void* self, const void* cu, int dir, int edge)
{
  static ef_shim_t shim = (ef_shim_t)dlsym(RTLD_DEFAULT, "vvcshim_edge_filter");
  static ef_real_t real = (ef_real_t)must(g_target ? dlsym(g_target, EFL_SYM) : nullptr, EFL_SYM);
  if (shim && shim(self, cu, dir, edge, 0)) return;
  real(self, cu, dir, edge);
}
void hook_edge_chroma(void* self, const void* cu, int dir, int edge)
{
  static ef_shim_t shim = (ef_shim_t)dlsym(RTLD_DEFAULT, "vvcshim_edge_filter");
  static ef_real_t real = (ef_real_t)must(g_target ? dlsym(g_target, EFC_SYM) : nullptr, EFC_SYM);
  if (shim && shim(self, cu, dir, edge, 1)) return;
  real(self, cu, dir, edge);
}
void hook_sao_stats(void* self, void* blkStats, void* org, void* src, void* cs, bool pre)
{
  static sao_shim_t shim = (sao_shim_t)dlsym(RTLD_DEFAULT, "vvcshim_sao_stats");
  static sao_real_t real = (sao_real_t)must(g_target ? dlsym(g_target, SAO_SYM) : nullptr, SAO_SYM);
  if (shim && shim(self, blkStats, org, src, cs, pre)) return;
  real(self, blkStats, org, src, cs, pre);
}
void hook_alf_stats(void* self, void* org, void* rec)
{
  static alf_shim_t shim = (alf_shim_t)dlsym(RTLD_DEFAULT, "vvcshim_alf_stats");
  static alf_real_t real = (alf_real_t)must(g_target ? dlsym(g_target, ALF_SYM) : nullptr, ALF_SYM);
  if (shim && shim(self, org, rec)) return;
  real(self, org, rec);
}
void hook_tr_fwd(int bd, const short* resi, size_t stride, int* coeff, int w, int h, int maxLog2, unsigned char mode, unsigned char idx, bool qtbt)
{
  static trf_shim_t shim = (trf_shim_t)dlsym(RTLD_DEFAULT, "vvcshim_tr_fwd");
  static trf_real_t real = (trf_real_t)must(g_target ? dlsym(g_target, TRF_SYM) : nullptr, TRF_SYM);
  if (shim && shim(bd, resi, stride, coeff, w, h, maxLog2, mode, idx, qtbt)) return;
  real(bd, resi, stride, coeff, w, h, maxLog2, mode, idx, qtbt);
}
void hook_tr_inv(int bd, const int* coeff, short* resi, size_t stride, int w, int h, unsigned skipW, unsigned skipH, int maxLog2, unsigned char mode, unsigned char idx)
{
  static tri_shim_t shim = (tri_shim_t)dlsym(RTLD_DEFAULT, "vvcshim_tr_inv");
  static tri_real_t real = (tri_real_t)must(g_target ? dlsym(g_target, TRI_SYM) : nullptr, TRI_SYM);
  if (shim && shim(bd, coeff, resi, stride, w, h, skipW, skipH, maxLog2, mode, idx)) return;
  real(bd, coeff, resi, stride, w, h, skipW, skipH, maxLog2, mode, idx);
}
void hook_frac(void* self, void* pu, int list, int refIdx, void* cStruct, void* mvInt, void* mvHalf, void* mvQter, void* cost)
{
  static frac_shim_t shim = (frac_shim_t)dlsym(RTLD_DEFAULT, "vvcshim_frac");
  static frac_real_t real = (frac_real_t)must(g_target ? dlsym(g_target, FRAC_SYM) : nullptr, FRAC_SYM);
  if (shim && shim(self, pu, list, refIdx, cStruct, mvInt, mvHalf, mvQter, cost)) return;
  real(self, pu, list, refIdx, cStruct, mvInt, mvHalf, mvQter, cost);
}
void hook_fullsearch(void* self, void* cStruct, void* mv, void* sad)
{
  static fs_shim_t shim = (fs_shim_t)dlsym(RTLD_DEFAULT, "vvcshim_fullsearch");
  static fs_real_t real = (fs_real_t)must(g_target ? dlsym(g_target, FS_SYM) : nullptr, FS_SYM);
  if (shim && shim(self, cStruct, mv, sad)) return;
  real(self, cStruct, mv, sad);
}
static tz_real_t tz_real() { static tz_real_t real = (tz_real_t)must(g_target ? dlsym(g_target, TZ_SYM) : nullptr, TZ_SYM); return real; }
void hook_tzsearch(void* self, void* pu, void* cStruct, void* mv, void* sad, const void* pred2, bool ext, bool fast)
{
  static tz_shim_t shim = (tz_shim_t)dlsym(RTLD_DEFAULT, "vvcshim_tzsearch");
  if (shim && shim(self, pu, cStruct, mv, sad, pred2, ext, fast)) return;
  tz_real()(self, pu, cStruct, mv, sad, pred2, ext, fast);
}
// the reference's own body, for the shim's optional A/B check (VVCGPU_SHIM_TZ_VERIFY)
void vtmhooks_real_tzsearch(void* self, void* pu, void* cStruct, void* mv, void* sad, const void* pred2, bool ext, bool fast)
{
  tz_real()(self, pu, cStruct, mv, sad, pred2, ext, fast);
}
unsigned hook_crc(const void* pic, void* digest, const void* bitDepths)
{
  static hash_shim_t shim = (hash_shim_t)dlsym(RTLD_DEFAULT, "vvcshim_pichash");
  static hash_real_t real = (hash_real_t)must(g_target ? dlsym(g_target, CRC_SYM) : nullptr, CRC_SYM);
  if (shim) { const int n = shim(1, pic, digest, bitDepths); if (n) return (unsigned)n; }
  return real(pic, digest, bitDepths);
}
unsigned hook_checksum(const void* pic, void* digest, const void* bitDepths)
{
  static hash_shim_t shim = (hash_shim_t)dlsym(RTLD_DEFAULT, "vvcshim_pichash");
  static hash_real_t real = (hash_real_t)must(g_target ? dlsym(g_target, SUM_SYM) : nullptr, SUM_SYM);
  if (shim) { const int n = shim(2, pic, digest, bitDepths); if (n) return (unsigned)n; }
  return real(pic, digest, bitDepths);
}
static dq_real_t dq_real() { static dq_real_t real = (dq_real_t)must(g_target ? dlsym(g_target, DQ_SYM) : nullptr, DQ_SYM); return real; }
void hook_depquant(void* self, void* tu, const void* compID, const void* src, void* absSum, const void* qp, const void* ctx)
{
  static dq_shim_t shim = (dq_shim_t)dlsym(RTLD_DEFAULT, "vvcshim_depquant");
  if (shim && shim(self, tu, compID, src, absSum, qp, ctx)) return;
  dq_real()(self, tu, compID, src, absSum, qp, ctx);
}
void vtmhooks_real_depquant(void* self, void* tu, const void* compID, const void* src, void* absSum, const void* qp, const void* ctx)
{
  dq_real()(self, tu, compID, src, absSum, qp, ctx);
}
// QuantRDOQ::quant: virtual as well, and called by DepQuant::quant (qualified, through the PLT) when dependent quantisation is off
static dq_real_t rq_real() { static dq_real_t real = (dq_real_t)must(g_target ? dlsym(g_target, RQ_SYM) : nullptr, RQ_SYM); return real; }
void hook_rdoq(void* self, void* tu, const void* compID, const void* src, void* absSum, const void* qp, const void* ctx)
{
  static dq_shim_t shim = (dq_shim_t)dlsym(RTLD_DEFAULT, "vvcshim_rdoq");
  if (shim && shim(self, tu, compID, src, absSum, qp, ctx)) return;
  rq_real()(self, tu, compID, src, absSum, qp, ctx);
}
void vtmhooks_real_rdoq(void* self, void* tu, const void* compID, const void* src, void* absSum, const void* qp, const void* ctx)
{
  rq_real()(self, tu, compID, src, absSum, qp, ctx);
}
}
