#!/usr/bin/env python3
"""Generates integration/vtm-2.1-hip.patch from a VTM 2.1 source tree (default /root/reference): a zero-context unified diff (`diff -U0`: file names,
line numbers and ADDED lines only -- no reference text is copied into this repository) that
  * makes `SIMD=HIP` a value of the reference's SIMD selector (CommonLib/x86/CommonDefX86.cpp, read_x86_extension_flags),
  * puts ONE line at the top of the five table-initialisation functions of CommonLib/x86/InitX86.cpp and of the picture-level in-loop entry points,
    which hands the call to the bodies of integration/InitHIP.cpp when HIP is selected.
Every insertion point is found by the function's signature, so the script also tells when a tree is not the one the patch was made for.
usage: python integration/make_patch.py [reference root] > integration/vtm-2.1-hip.patch"""
import os
import re
import subprocess
import sys
import tempfile

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
LIB = "source/Lib"
INC = '#include "hip/InitHIP.h"   // SIMD=HIP'

# file -> list of (signature regex of the function, line(s) to insert right after its opening brace)
HOOKS = {
    "CommonLib/x86/InitX86.cpp": [
        (r"^void InterpolationFilter::initInterpolationFilterX86\(", ["  if( vvcHipEnter( VVC_HIP_INIT_IF ) ) { wrap_initIfX86( this ); return; }"]),
        (r"^void PelBufferOps::initPelBufOpsX86\(", ["  if( vvcHipEnter( VVC_HIP_INIT_PELBUF ) ) { wrap_initPelBufX86( this ); return; }"]),
        (r"^void RdCost::initRdCostX86\(", ["  if( vvcHipEnter( VVC_HIP_INIT_RDCOST ) ) { wrap_initRdCostX86( this ); return; }"]),
        (r"^void AffineGradientSearch::initAffineGradientSearchX86\(", ["  if( vvcHipEnter( VVC_HIP_INIT_AGS ) ) { wrap_initAgsX86( this ); return; }"]),
        (r"^void AdaptiveLoopFilter::initAdaptiveLoopFilterX86\(", ["  if( vvcHipEnter( VVC_HIP_INIT_ALF ) ) { wrap_initAlfX86( this ); return; }"]),
    ],
    "CommonLib/LoopFilter.cpp": [
        (r"^void LoopFilter::loopFilterPic\(", ["  if( vvcHipEnter( VVC_HIP_LOOPFILTER ) ) { wrap_loopFilterPic( this, cs ); return; }"]),
        (r"^void LoopFilter::xEdgeFilterLuma\(", ["  if( vvcHipSelected() && vvcshim_edge_filter( this, &cu, (int) edgeDir, iEdge, 0 ) ) return;"]),
        (r"^void LoopFilter::xEdgeFilterChroma\(", ["  if( vvcHipSelected() && vvcshim_edge_filter( this, &cu, (int) edgeDir, iEdge, 1 ) ) return;"]),
    ],
    "CommonLib/SampleAdaptiveOffset.cpp": [
        (r"^void SampleAdaptiveOffset::offsetCTU\(", ["  if( vvcHipEnter( VVC_HIP_OFFSETCTU ) ) { wrap_offsetCTU( this, area, src, res, saoblkParam, cs ); return; }"]),
        (r"^void SampleAdaptiveOffset::SAOProcess\(", ["  if( vvcHipEnter( VVC_HIP_SAO ) ) { wrap_SAOProcess( this, cs, saoBlkParams ); return; }"]),
    ],
    "CommonLib/AdaptiveLoopFilter.cpp": [
        (r"^void AdaptiveLoopFilter::ALFProcess\(", ["  if( vvcHipEnter( VVC_HIP_ALF ) ) { wrap_ALFProcess( this, cs, alfSliceParam ); return; }"]),
    ],
    "EncoderLib/EncSampleAdaptiveOffset.cpp": [
        (r"^void EncSampleAdaptiveOffset::SAOProcess\(", [
            "#if K0238_SAO_GREEDY_MERGE_ENCODING",
            "  if( vvcHipEnter( VVC_HIP_ENCSAO ) ) { wrap_EncSAOProcess( this, cs, sliceEnabled, lambdas, bTestSAODisableAtPictureLevel, saoEncodingRate, saoEncodingRateChroma, isPreDBFSamplesUsed, isGreedymergeEncoding ); return; }",
            "#endif"]),
        (r"^void EncSampleAdaptiveOffset::getStatistics\(", ["  if( vvcHipSelected() && vvcshim_sao_stats( this, &blkStats, &orgYuv, &srcYuv, &cs, isCalculatePreDeblockSamples ) ) return;"]),
    ],
    "CommonLib/TrQuant.cpp": [
        (r"^void xTrMxN_EMT\(", ["  if( vvcHipSelected() && vvcshim_tr_fwd( bitDepth, residual, stride, coeff, iWidth, iHeight, maxLog2TrDynamicRange, ucMode, ucTrIdx, useQTBT ) ) return;"]),
        (r"^void xITrMxN_EMT\(", ["  if( vvcHipSelected() && vvcshim_tr_inv( bitDepth, coeff, residual, stride, iWidth, iHeight, uiSkipWidth, uiSkipHeight, maxLog2TrDynamicRange, ucMode, ucTrIdx ) ) return;"]),
        (r"^void TrQuant::invTransformNxN\(", ["  if( vvcHipEnter( VVC_HIP_INVTR ) ) { wrap_invTransformNxN( this, tu, compID, pResi, cQP ); return; }"]),
    ],
    "CommonLib/IntraPrediction.cpp": [
        (r"^void IntraPrediction::predIntraAng\(", ["  if( vvcHipEnter( VVC_HIP_PREDANG ) ) { wrap_predIntraAng( this, compId, piPred, pu, useFilteredPredSamples ); return; }"]),
        (r"^void IntraPrediction::predIntraChromaLM\(", ["  if( vvcHipEnter( VVC_HIP_PREDLM ) ) { wrap_predIntraChromaLM( this, compID, piPred, pu, chromaArea, intraDir ); return; }"]),
        (r"^void IntraPrediction::initIntraPatternChType\(", ["  if( vvcHipEnter( VVC_HIP_INITPATTERN ) ) { wrap_initIntraPatternChType( this, cu, area, bFilterRefSamples ); return; }"]),
    ],
    "CommonLib/Picture.cpp": [
        (r"^void Picture::extendPicBorder\(", ["  if( vvcHipEnter( VVC_HIP_EXTEND ) ) { wrap_extendPicBorder( this ); return; }"]),
    ],
    "CommonLib/PicYuvMD5.cpp": [
        (r"^uint32_t calcCRC\(", ["  if( vvcHipSelected() ) { const int n = vvcshim_pichash( 1, &pic, &digest, &bitDepths ); if( n ) return (uint32_t) n; }"]),
        (r"^uint32_t calcChecksum\(", ["  if( vvcHipSelected() ) { const int n = vvcshim_pichash( 2, &pic, &digest, &bitDepths ); if( n ) return (uint32_t) n; }"]),
    ],
    "CommonLib/DepQuant.cpp": [
        (r"^void DepQuant::quant\(", ["  if( vvcHipSelected() && vvcshim_depquant( this, &tu, &compID, &pSrc, &uiAbsSum, &cQP, &ctx ) ) return;"]),
    ],
    "CommonLib/QuantRDOQ.cpp": [
        (r"^void QuantRDOQ::quant\(", ["  if( vvcHipSelected() && vvcshim_rdoq( this, &tu, &compID, &pSrc, &uiAbsSum, &cQP, &ctx ) ) return;"]),
    ],
    "EncoderLib/InterSearch.cpp": [
        (r"^void InterSearch::xPatternSearch\(", ["  if( vvcHipSelected() && vvcshim_fullsearch( this, &cStruct, &rcMv, &ruiSAD ) ) return;"]),
        (r"^void InterSearch::xTZSearch\(", ["  if( vvcHipSelected() && vvcshim_tzsearch( this, &pu, &cStruct, &rcMv, &ruiSAD, pIntegerMv2Nx2NPred, bExtendedSettings, bFastSettings ) ) return;"]),
        (r"^void InterSearch::xPatternSearchFracDIF\(", ["  if( vvcHipSelected() && vvcshim_frac( this, &pu, (int) eRefPicList, iRefIdx, &cStruct, &rcMvInt, &rcMvHalf, &rcMvQter, &ruiCost ) ) return;"]),
    ],
    "EncoderLib/EncAdaptiveLoopFilter.cpp": [
        (r"^void EncAdaptiveLoopFilter::ALFProcess\(", ["  if( vvcHipEnter( VVC_HIP_ENCALF ) ) { wrap_EncALFProcess( this, cs, lambdas, alfSliceParam ); return; }"]),
        (r"^void EncAdaptiveLoopFilter::deriveStatsForFiltering\(", ["  if( vvcHipSelected() && vvcshim_alf_stats( this, &orgYuv, &recYuv ) ) return;"]),
    ],
}
# the selector: "HIP" = the best x86 tables the CPU has, with the library's slots and picture-level bodies on top
SELECTOR = ("CommonLib/x86/CommonDefX86.cpp", r"^X86_VEXT read_x86_extension_flags\(", r"^\s*if\( !b_detection_finished \)",
            ['      if( extStrId == "HIP" ) { vvcHipSelect(); ext_flags = _get_x86_extensions(); b_detection_finished = true; return ext_flags; }'])


def after_brace(lines, start):
    for i in range(start, min(start + 16, len(lines))):
        if lines[i].strip() == "{":
            return i + 1
    raise SystemExit("no opening brace behind line %d" % (start + 1))


def add_include(lines):
    for i, l in enumerate(lines):
        if l.startswith('#include "'):
            return lines[:i + 1] + [INC] + lines[i + 1:]
    raise SystemExit("no include line")


def patched(rel, lines):
    out = list(lines)
    ins = []                                                       # (index, new lines), applied from the bottom up
    for sig, new in HOOKS.get(rel, []):
        hits = [i for i, l in enumerate(out) if re.search(sig, l)]
        if not hits:
            raise SystemExit("%s: signature %s not found -- not a VTM 2.1 tree?" % (rel, sig))
        for h in hits:                                              # the encoder's SAOProcess has two signatures under #if / #else and one body
            pos = after_brace(out, h)
            if (pos, new) not in ins:
                ins.append((pos, new))
    if rel == SELECTOR[0]:
        f = next(i for i, l in enumerate(out) if re.search(SELECTOR[1], l))
        g = next(i for i in range(f, len(out)) if re.search(SELECTOR[2], out[i]))
        ins.append((after_brace(out, g), SELECTOR[3]))
    for pos, new in sorted(ins, key=lambda t: -t[0]):
        out[pos:pos] = new
    return add_include(out)


def main():
    files = sorted(set(HOOKS) | {SELECTOR[0]})
    with tempfile.TemporaryDirectory() as tmp:
        for rel in files:
            src = open(os.path.join(REF, LIB, rel)).read().split("\n")
            for side, content in (("a", src), ("b", patched(rel, src))):
                path = os.path.join(tmp, side, LIB, rel)
                os.makedirs(os.path.dirname(path), exist_ok=True)
                open(path, "w").write("\n".join(content))
            r = subprocess.run(["diff", "-U0", "--label", "a/%s/%s" % (LIB, rel), "--label", "b/%s/%s" % (LIB, rel),
                                os.path.join(tmp, "a", LIB, rel), os.path.join(tmp, "b", LIB, rel)], capture_output=True, text=True)
            assert r.returncode == 1, (rel, r.stderr)
            sys.stdout.write(r.stdout)


if __name__ == "__main__":
    main()
