// InitHIP.h -- declarations the patched reference files see (integration/vtm-2.1-hip.patch adds `#include "hip/InitHIP.h"` to them).
// To be copied to source/Lib/CommonLib/hip/InitHIP.h of a VTM 2.1 tree together with InitHIP.cpp; see integration/README.md.
#pragma once
#include <vector>
#include <cstddef>

class LoopFilter; class SampleAdaptiveOffset; class AdaptiveLoopFilter; class EncSampleAdaptiveOffset; class EncAdaptiveLoopFilter;
class RdCost; class InterpolationFilter; struct PelBufferOps; class AffineGradientSearch; class CodingStructure;
struct SAOBlkParam; struct AlfSliceParam; struct UnitArea; struct SAOStatData;
template <typename T> struct UnitBuf;

// one id per hooked function: vvcHipEnter(id) is true when SIMD=HIP was selected and the call is not the library's own re-entry into the function
enum VvcHipHook { VVC_HIP_LOOPFILTER, VVC_HIP_SAO, VVC_HIP_OFFSETCTU, VVC_HIP_ALF, VVC_HIP_ENCSAO, VVC_HIP_ENCALF,
                  VVC_HIP_INIT_RDCOST, VVC_HIP_INIT_IF, VVC_HIP_INIT_PELBUF, VVC_HIP_INIT_ALF, VVC_HIP_INIT_AGS,
                  VVC_HIP_INVTR, VVC_HIP_PREDANG, VVC_HIP_PREDLM, VVC_HIP_INITPATTERN, VVC_HIP_EXTEND, VVC_HIP_HOOKS };
void vvcHipSelect();                      // called by read_x86_extension_flags("HIP")
bool vvcHipSelected();
bool vvcHipEnter( VvcHipHook id );

// the bodies (vtm_hip_shim.cpp): each does the work on the device, or calls back into the reference's own function (hook disarmed) when it cannot
void wrap_loopFilterPic( LoopFilter*, CodingStructure& );
void wrap_SAOProcess( SampleAdaptiveOffset*, CodingStructure&, SAOBlkParam* );
void wrap_offsetCTU( SampleAdaptiveOffset*, const UnitArea&, const UnitBuf<const short>&, UnitBuf<short>&, SAOBlkParam&, CodingStructure& );
void wrap_ALFProcess( AdaptiveLoopFilter*, CodingStructure&, AlfSliceParam& );
void wrap_EncSAOProcess( EncSampleAdaptiveOffset*, CodingStructure&, bool*, const double*, const bool, const double, const double, bool, bool );
void wrap_EncALFProcess( EncAdaptiveLoopFilter*, CodingStructure&, const double*, AlfSliceParam& );
void wrap_initRdCostX86( RdCost* );
void wrap_initIfX86( InterpolationFilter* );
void wrap_initPelBufX86( PelBufferOps* );
void wrap_initAlfX86( AdaptiveLoopFilter* );
void wrap_initAgsX86( AffineGradientSearch* );
// the "next" rows N1 / N4 (same shape: one line at the top of the member)
class TrQuant; class IntraPrediction; struct Picture; struct TransformUnit; struct PredictionUnit; struct CodingUnit; struct CompArea; class QpParam;
template <typename T> struct AreaBuf;
void wrap_invTransformNxN( TrQuant*, TransformUnit&, const ComponentID&, AreaBuf<short>&, const QpParam& );
void wrap_predIntraAng( IntraPrediction*, const ComponentID, AreaBuf<short>&, const PredictionUnit&, const bool );
void wrap_predIntraChromaLM( IntraPrediction*, const ComponentID, AreaBuf<short>&, const PredictionUnit&, const CompArea&, int );
void wrap_initIntraPatternChType( IntraPrediction*, const CodingUnit&, const CompArea&, const bool );
void wrap_extendPicBorder( Picture* );
// functions the library may serve whole: 1 (or the digest length) = done on the device, 0 = run the reference's own body.  Untyped here (the patched
// files pass the objects they have); the typed definitions are in vtm_hip_shim.cpp.
#ifndef VVCSHIM_SOURCE_HOOKS
extern "C" int vvcshim_tr_fwd( int bd, const short* resi, size_t stride, int* coeff, int w, int h, int maxLog2, unsigned char ucMode, unsigned char ucTrIdx, bool useQTBT );
extern "C" int vvcshim_tr_inv( int bd, const int* coeff, short* resi, size_t stride, int w, int h, unsigned skipW, unsigned skipH, int maxLog2, unsigned char ucMode, unsigned char ucTrIdx );
extern "C" int vvcshim_depquant( void* self, void* tu, const void* compID, const void* src, void* absSum, const void* qp, const void* ctx );
extern "C" int vvcshim_rdoq( void* self, void* tu, const void* compID, const void* src, void* absSum, const void* qp, const void* ctx );
extern "C" int vvcshim_edge_filter( void* self, const void* cu, int edgeDir, int iEdge, int chroma );
extern "C" int vvcshim_frac( void* self, const void* pu, int eRefPicList, int iRefIdx, void* cStruct, const void* mvInt, void* mvHalf, void* mvQter, void* cost );
extern "C" int vvcshim_fullsearch( void* self, void* cStruct, void* mv, void* sad );
extern "C" int vvcshim_tzsearch( void* self, const void* pu, void* cStruct, void* mv, void* sad, const void* pred2Nx2N, bool extended, bool fast );
extern "C" int vvcshim_pichash( int method, const void* pic, void* digest, const void* bitDepths );
#endif
// the two encoder statistics passes: 1 = done on the device, 0 = run the reference's own body
extern "C" int vvcshim_sao_stats( EncSampleAdaptiveOffset* self, std::vector<SAOStatData**>* blkStats, UnitBuf<short>* orgYuv, UnitBuf<short>* srcYuv,
                                  CodingStructure* cs, bool isCalculatePreDeblockSamples );
extern "C" int vvcshim_alf_stats( EncAdaptiveLoopFilter* self, UnitBuf<short>* orgYuv, UnitBuf<short>* recYuv );
