// InitHIP.h -- declarations the patched reference files see (integration/vtm-2.1-hip.patch adds `#include "hip/InitHIP.h"` to them).
// To be copied to source/Lib/CommonLib/hip/InitHIP.h of a VTM 2.1 tree together with InitHIP.cpp; see integration/README.md.
#pragma once
#include <vector>

class LoopFilter; class SampleAdaptiveOffset; class AdaptiveLoopFilter; class EncSampleAdaptiveOffset; class EncAdaptiveLoopFilter;
class RdCost; class InterpolationFilter; struct PelBufferOps; class AffineGradientSearch; class CodingStructure;
struct SAOBlkParam; struct AlfSliceParam; struct UnitArea; struct SAOStatData;
template <typename T> struct UnitBuf;

// one id per hooked function: vvcHipEnter(id) is true when SIMD=HIP was selected and the call is not the library's own re-entry into the function
enum VvcHipHook { VVC_HIP_LOOPFILTER, VVC_HIP_SAO, VVC_HIP_OFFSETCTU, VVC_HIP_ALF, VVC_HIP_ENCSAO, VVC_HIP_ENCALF,
                  VVC_HIP_INIT_RDCOST, VVC_HIP_INIT_IF, VVC_HIP_INIT_PELBUF, VVC_HIP_INIT_ALF, VVC_HIP_INIT_AGS, VVC_HIP_HOOKS };
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
// the two encoder statistics passes: 1 = done on the device, 0 = run the reference's own body
extern "C" int vvcshim_sao_stats( EncSampleAdaptiveOffset* self, std::vector<SAOStatData**>* blkStats, UnitBuf<short>* orgYuv, UnitBuf<short>* srcYuv,
                                  CodingStructure* cs, bool isCalculatePreDeblockSamples );
extern "C" int vvcshim_alf_stats( EncAdaptiveLoopFilter* self, UnitBuf<short>* orgYuv, UnitBuf<short>* recYuv );
