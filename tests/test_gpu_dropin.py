"""Drop-in proof on the GPU box: the reference DecoderApp / EncoderApp classes (oracle/_ref/libvtmref_hip.so = the
reference objects, unmodified, with LoopFilter::loopFilterPic / SampleAdaptiveOffset::SAOProcess /
AdaptiveLoopFilter::ALFProcess redirected by ld --wrap to vvcsoftware_vtm_amd/shim/vtm_hip_shim.cpp) run their in-loop
chain on the MI355X through the C ABI.

  * decoder: every picture's MD5 must match the hash SEI the reference ENCODER embedded -> `(OK)` for all pictures, and
    the written YUV must have the md5 the reference CPU decoder produced (tests/golden/bitstreams/manifest.json);
  * encoder: with deblocking on the GPU inside the encoding loop the bitstream must be byte-identical to the fixture.
This pins the deblocking path (host map derivation + kernels) and the SAO/ALF parameter plumbing end to end."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
APP = os.path.join(ROOT, "oracle", "_ref", "vtmref_app")
HIPLIB = os.path.join(ROOT, "oracle", "_ref", "libvtmref_hip.so")
BS = os.path.join(ROOT, "tests", "golden", "bitstreams")


def md5(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


def manifest():
    return json.load(open(os.path.join(BS, "manifest.json")))


needs_ref = pytest.mark.skipif(not (os.path.exists(APP) and os.path.exists(HIPLIB)), reason="oracle/_ref not built")


@needs_ref
@pytest.mark.parametrize("name", sorted(manifest()))
def test_decoder_inloop_chain_on_gpu(name, tmp_path):
    m = manifest()[name]
    out = str(tmp_path / "dec.yuv")
    r = subprocess.run([APP, "--hip", "dec", "-b", os.path.join(BS, name + ".bin"), "-o", out, "-d", str(m["bd"])],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "ERROR" not in r.stdout and r.stdout.count("(OK)") >= m["frames"], r.stdout[-2000:]
    assert md5(out) == m["dec_yuv_md5"]
    # the GPU really ran: the shim reports its call counts at exit
    line = [l for l in r.stderr.splitlines() if "[vvcgpu shim]" in l]
    assert line, r.stderr[-1000:]
    calls = [int(x) for x in line[-1].replace(",", " ").split() if x.isdigit()]
    assert calls[0] >= m["frames"]            # one deblocking call per picture
    assert calls[21] > 0, line[-1]            # IntraPrediction::predIntraAng of the intra TUs ran on the GPU (next row N4)
    assert calls[24] > 0, line[-1]            # ... and predIntraChromaLM (CCLM) of the chroma blocks that chose it
    assert calls[25] > 0, line[-1]            # ... from reference samples gathered by vvcgpu_intra_fill_refs_batch
    if m["frames"] > 1:
        assert calls[22] > 0, line[-1]        # Picture::extendPicBorder of the reference pictures
    if m.get("hash", 1) != 1:
        assert calls[23] >= m["frames"], line[-1]   # the CRC / checksum that produced every (OK) above came from vvcgpu_picture_hash


@needs_ref
def test_decoder_fallthrough_matches(tmp_path):
    """VVCGPU_SHIM=0 routes the same binary through the reference CPU filters: same output."""
    name = "ra_208x120_10b_q32"
    m = manifest()[name]
    out = str(tmp_path / "dec.yuv")
    env = dict(os.environ, VVCGPU_SHIM="0")
    r = subprocess.run([APP, "--hip", "dec", "-b", os.path.join(BS, name + ".bin"), "-o", out, "-d", str(m["bd"])],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and md5(out) == m["dec_yuv_md5"]


@needs_ref
@pytest.mark.parametrize("name", ["ldp_208x120_10b_q27", "ai_416x240_8b_q37own", "ldpfs_208x120_10b_q32", "ldpcrc_208x120_10b_q32", "aisum_208x120_8b_q37", "ldprdoq_208x120_10b_q32", "rab_208x120_10b_q32"])
def test_encoder_with_gpu_inloop_is_bitstream_exact(tmp_path, name):
    """the reference ENCODER with deblocking, the SAO statistics (getStatistics) and the ALF covariances
    (deriveStatsForFiltering), the per-CTU SAO offsetting (offsetCTU) and the three ALF table slots computed on the GPU inside its loop: every SAO / ALF decision and therefore the bitstream must be
    byte-identical to the fixture produced by the unmodified CPU encoder."""
    sys.path.insert(0, ROOT)
    from vvcsoftware_vtm_amd import synth
    m = manifest()[name]
    yuv = str(tmp_path / "in.yuv")
    synth.write_yuv(yuv, synth.gen_yuv(m["w"], m["h"], m["frames"], m["bd"], m["seed"]), m["bd"])
    assert m["cfg"].startswith("@")            # this repository's own test cfg (the reference's cfg files do not travel)
    cfg = os.path.join(ROOT, m["cfg"][1:])
    binf = str(tmp_path / "out.bin")
    r = subprocess.run([APP, "--hip", "enc", "-c", cfg, "-i", yuv, "-wdt", str(m["w"]), "-hgt", str(m["h"]), "-fr", "30",
                        "-f", str(m["frames"]), "-q", str(m["qp"]), "--InputBitDepth=%d" % m["bd"], "--InternalBitDepth=%d" % m["bd"],
                        "--OutputBitDepth=%d" % m["bd"], "-b", binf, "-o", str(tmp_path / "rec.yuv"), "--SEIDecodedPictureHash=%d" % m.get("hash", 1)] + m.get("extra", []),
                       capture_output=True, text=True, timeout=1200, env=dict(os.environ, VVCGPU_SHIM_TZ_VERIFY="1", VVCGPU_SHIM_CCLM_VERIFY="1", VVCGPU_SHIM_FILL_VERIFY="1", VVCGPU_SHIM_DEPQUANT_VERIFY="1", VVCGPU_SHIM_RDOQ_VERIFY="1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert md5(binf) == m["bin_md5"]
    line = [l for l in r.stderr.splitlines() if "[vvcgpu shim]" in l]
    assert line, r.stderr[-1000:]
    calls = [int(x) for x in line[-1].replace(",", " ").split() if x.isdigit()]
    assert calls[0] >= m["frames"] and calls[3] >= m["frames"] and calls[4] >= m["frames"], line[-1]
    # per-CTU SAO offsetCTU, and the ALF table slots (m_filter5x5Blk / m_filter7x7Blk / m_deriveClassificationBlk) installed
    # where the reference installs its SIMD functions
    assert calls[5] > 0 and calls[7] > 0, line[-1]
    if name.startswith("ldpfs_") :
        assert calls[15] > 0, line[-1]         # xPatternSearch: full search = vvcgpu_sad_search with the fused arg-min
    if name.startswith("ldp_") or name.startswith("ldpcrc_"):
        assert calls[8] > 0 and calls[17] > 0, line[-1]   # RdCost table slots DF_SAD64 and DF_SSE64 (64-wide blocks) ran on the GPU
        assert calls[10] > 0, line[-1]         # InterpolationFilter table slots (64-wide calls) ran on the GPU
        assert calls[11] > 0, line[-1]         # PelBufferOps table slots (addAvg8 / reco8 / linTf8, 64-wide calls)
        assert calls[14] > 0, line[-1]         # xPatternSearchFracDIF: the fused half/quarter refinement kernel, every inter PU
        assert calls[18] > 0 and calls[19] > 0, line[-1]   # affine gradient table slots (Sobel planes, equal coefficients)
        assert calls[20] > 0, line[-1]         # xTZSearch: the whole integer TZ search of every PU on the device (next row N2)
        assert "TZ mismatch" not in r.stderr, r.stderr[-2000:]
    if name.startswith("rab_"):
        # hierarchical-B random access: uni- and bi-predictive searches of every B picture (fractional refinement, TZ search)
        assert calls[14] > 0 and calls[20] > 0 and "TZ mismatch" not in r.stderr, line[-1] + r.stderr[-1500:]
    assert calls[12] > 0 and calls[16] > 0, line[-1]   # forward transforms (32/64-side TUs); de-quantisation + inverse (every TU)
    assert calls[21] > 0, line[-1]             # predIntraAng: every intra mode candidate of the first calls (capped, VVCGPU_SHIM_INTRA_LIMIT)
    assert calls[24] > 0 and "CCLM mismatch" not in r.stderr, line[-1] + r.stderr[-1500:]   # predIntraChromaLM on the GPU, A/B-checked per call
    assert calls[25] > 0 and "reference sample mismatch" not in r.stderr, line[-1] + r.stderr[-1500:]   # xFillReferenceSamples on the GPU, A/B-checked
    if "--DepQuant=0" in m.get("extra", []):
        # dependent quantisation off: DepQuant::quant hands every TU to QuantRDOQ::quant (DepQuant.cpp:1411-1421); capped by VVCGPU_SHIM_RDOQ_LIMIT, A/B-checked per call
        assert calls[26] == 0 and calls[27] >= 20000 and "RDOQ mismatch" not in r.stderr, line[-1] + r.stderr[-1500:]
    else:
        assert calls[26] > 0 and "DepQuant mismatch" not in r.stderr, line[-1] + r.stderr[-1500:]   # the dependent-quantisation trellis on the GPU, A/B-checked
    if m["frames"] > 1:
        assert calls[22] > 0, line[-1]         # extendPicBorder of every reference picture
    if m.get("hash", 1) != 1:
        assert calls[23] >= m["frames"], line[-1]   # the hash SEI payload itself was computed by vvcgpu_picture_hash
    print(line[-1])
    # production form of the picture-level binding: the reconstruction goes up once per picture (after CTU coding), the original once (for the
    # SAO / ALF statistics), and the filtered picture comes down once, however many stages ran in between
    res = [l for l in r.stderr.splitlines() if "[vvcgpu resident]" in l]
    assert res and "resident form on" in res[-1], r.stderr[-1000:]
    pics, ups, downs = [int(x) for x in res[-1].replace(",", " ").replace("(", " ").split() if x.isdigit()][:3]
    assert pics == m["frames"] and downs == m["frames"] and ups <= 2 * m["frames"], res[-1]
    print(res[-1])
    if name.startswith("ai_"):
        assert calls[6] > 0, line[-1]          # on this clip the encoder enables ALF: the filter table slots ran


@needs_ref
def test_encoder_per_call_form_still_exact(tmp_path):
    """VVCGPU_SHIM_RESIDENT=0: every picture-level hook uploads and downloads the picture itself (the round-1 proof form) -- same bitstream"""
    sys.path.insert(0, ROOT)
    from vvcsoftware_vtm_amd import synth
    name = "ai_416x240_8b_q37own"
    m = manifest()[name]
    yuv = str(tmp_path / "in.yuv")
    synth.write_yuv(yuv, synth.gen_yuv(m["w"], m["h"], m["frames"], m["bd"], m["seed"]), m["bd"])
    cfg = os.path.join(ROOT, m["cfg"][1:])
    binf = str(tmp_path / "out.bin")
    r = subprocess.run([APP, "--hip", "enc", "-c", cfg, "-i", yuv, "-wdt", str(m["w"]), "-hgt", str(m["h"]), "-fr", "30",
                        "-f", str(m["frames"]), "-q", str(m["qp"]), "--InputBitDepth=%d" % m["bd"], "--InternalBitDepth=%d" % m["bd"],
                        "--OutputBitDepth=%d" % m["bd"], "-b", binf, "-o", str(tmp_path / "rec.yuv"), "--SEIDecodedPictureHash=%d" % m.get("hash", 1)] + m.get("extra", []),
                       capture_output=True, text=True, timeout=1200, env=dict(os.environ, VVCGPU_SHIM_RESIDENT="0", VVCGPU_SHIM_NO_TABLES="1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert md5(binf) == m["bin_md5"]
    res = [l for l in r.stderr.splitlines() if "[vvcgpu resident]" in l]
    assert res and "resident form off" in res[-1]



def _encode_fixture(tmp_path, name, env, flag="--hip", extra=()):
    sys.path.insert(0, ROOT)
    from vvcsoftware_vtm_amd import synth
    m = manifest()[name]
    yuv = str(tmp_path / "in.yuv")
    synth.write_yuv(yuv, synth.gen_yuv(m["w"], m["h"], m["frames"], m["bd"], m["seed"]), m["bd"])
    cfg = os.path.join(ROOT, m["cfg"][1:])
    binf = str(tmp_path / "out.bin")
    r = subprocess.run([APP, flag, "enc", "-c", cfg, "-i", yuv, "-wdt", str(m["w"]), "-hgt", str(m["h"]), "-fr", "30",
                        "-f", str(m["frames"]), "-q", str(m["qp"]), "--InputBitDepth=%d" % m["bd"], "--InternalBitDepth=%d" % m["bd"],
                        "--OutputBitDepth=%d" % m["bd"], "-b", binf, "-o", str(tmp_path / "rec.yuv"), "--SEIDecodedPictureHash=%d" % m.get("hash", 1)] + m.get("extra", []) + list(extra),
                       capture_output=True, text=True, timeout=3300, env=dict(os.environ, **env))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert md5(binf) == m["bin_md5"]

    def numbers(tag):
        line = [l for l in r.stderr.splitlines() if tag in l]
        assert line, r.stderr[-1000:]
        return [int(x) for x in line[-1].split(":", 2)[-1].replace(",", " ").replace("(", " ").split() if x.isdigit()], line[-1]
    return m, r, numbers


@needs_ref
def test_encoder_random_access_gop16_fixture(tmp_path):
    """the repository's random-access cfg with the reference's RA values (GOP 16, hierarchical B, search range 384 -> minimum window 96,
    bi-prediction refinement, DepQuant, SAO, ALF; tests/golden/bitstreams/test_ra_gop16.cfg), 17 pictures of 416x240: the production form
    of the binding (picture-level hooks on the device-resident reconstruction) must reproduce the CPU encoder's bitstream"""
    m, r, numbers = _encode_fixture(tmp_path, "ragop16_416x240_10b_q32", {"VVCGPU_SHIM_HOOKS": "pic"})
    calls, line = numbers("[vvcgpu shim]")
    assert calls[0] == m["frames"] and calls[3] == m["frames"] and calls[4] == m["frames"] and calls[7] >= m["frames"], line
    assert sum(calls[8:22]) == 0, line                   # no block- or PU-level hook at this level
    (pics, ups, downs), res = numbers("[vvcgpu resident]")[0][:3], numbers("[vvcgpu resident]")[1]
    assert pics == m["frames"] and downs == m["frames"] and ups <= 2 * m["frames"] and "resident form on" in res, res


@needs_ref
def test_encoder_batched_pu_searches(tmp_path):
    """VVCGPU_SHIM_HOOKS=pub: the uni-prediction searches of a CU (InterSearch::predInterSearch, InterSearch.cpp:876-960: one xMotionEstimation per
    list / reference pair) run as ONE vvcgpu_me_batch on device-resident pictures in front of the reference's own function, which is then served from
    that session; every served integer search is also A/B-checked against the reference's own xTZSearch body (VVCGPU_SHIM_TZ_VERIFY) -- same bitstream"""
    m, r, numbers = _encode_fixture(tmp_path, "rab_208x120_10b_q32", {"VVCGPU_SHIM_HOOKS": "pub", "VVCGPU_SHIM_TZ_LIMIT": "0", "VVCGPU_SHIM_TZ_VERIFY": "1"})
    assert "TZ mismatch" not in r.stderr, [l for l in r.stderr.splitlines() if "mismatch" in l][:5]
    (sessions, searches, served, fell, uploads), line = numbers("[vvcgpu batched]")[0][:5], numbers("[vvcgpu batched]")[1]
    assert sessions > 0 and searches >= 2 * sessions and served > searches, line      # most searches of a session are served twice (integer + fractional)
    print(line)


@needs_ref
@pytest.mark.skipif(not os.environ.get("VVCGPU_NIGHTLY"), reason="nightly-style run (minutes of synchronous round trips): VVCGPU_NIGHTLY=1")
@pytest.mark.parametrize("name,level", [("rab_208x120_10b_q32", "all"), ("ragop16_416x240_10b_q32", "pu")])
def test_nightly_uncapped_hooks(tmp_path, name, level):
    """call caps lifted (VVCGPU_SHIM_*_LIMIT=0): every eligible call of every hook is served by the library, none is left to the CPU by a cap"""
    env = {"VVCGPU_SHIM_HOOKS": level}
    env.update({"VVCGPU_SHIM_%s_LIMIT" % k: "0" for k in ("INTRA", "FILL", "DEPQUANT", "RDOQ", "DQIT", "TZ")})
    m, r, numbers = _encode_fixture(tmp_path, name, env)
    capped, line = numbers("[vvcgpu caps]")
    assert capped[-6:] == [0] * 6, line
    calls, cl = numbers("[vvcgpu shim]")
    assert calls[14] > 0 and calls[20] > 0, cl
    if level == "all":
        assert calls[16] > 60000 and calls[21] > 60000 and calls[26] > 20000, cl      # beyond the default caps
    print(cl)


@needs_ref
@pytest.mark.skipif(not os.environ.get("VVCGPU_NIGHTLY"), reason="nightly-style run (millions of synchronous round trips): VVCGPU_NIGHTLY=1")
@pytest.mark.parametrize("name", ["ldpcrc_208x120_10b_q32"])
def test_nightly_table_slots_every_width(tmp_path, name):
    """VVCGPU_SHIM_HOOKS=slots: the function-pointer tables of SURVEY 8(b) taken literally -- EVERY distortion slot (SAD / HAD / SSE of every width,
    DF_SAD12/24/48 included), every interpolation slot and the PelBuffer slots serve calls of every width through the library, while the reference's own
    searches, transforms and quantisers run on the host and issue those calls.  The bitstream must not change."""
    m, r, numbers = _encode_fixture(tmp_path, name, {"VVCGPU_SHIM_HOOKS": "slots"})
    widths, wl = numbers("[vvcgpu slots]")
    served = [int(t.split(":")[1]) for t in wl.split("width:", 1)[1].split(",")]      # "4: n, 8: n, 12-16: n, 24-32: n, 48-64: n, 128: n"
    assert len(served) == 6 and all(v > 0 for v in served[:5]), wl          # widths 4 .. 64 all carried traffic
    calls, cl = numbers("[vvcgpu shim]")
    assert calls[8] > 100000 and calls[9] > 10000 and calls[10] > 10000 and calls[11] > 1000, cl     # SAD, HAD, interpolation, PelBuffer slots
    assert calls[14] == 0 and calls[20] == 0 and calls[26] == 0, cl       # no PU-level or N1 hooks in this mode: the reference's own code issued the calls
    print(wl)
    print(cl)


SELLIB = os.path.join(ROOT, "oracle", "_ref", "libvtmref_hipsel.so")
needs_sel = pytest.mark.skipif(not (os.path.exists(APP) and os.path.exists(SELLIB)), reason="oracle/_ref/libvtmref_hipsel.so not built (make -C oracle ref)")


@needs_sel
def test_patched_tree_selector_encoder(tmp_path):
    """the binding WITHOUT linker tricks: the reference built from a tree that carries integration/vtm-2.1-hip.patch (SIMD= selector value HIP, one line
    at the top of every hooked function; bodies in integration/InitHIP.cpp).  With --SIMD=HIP the
    whole in-loop chain runs on the device-resident picture and the bitstream is the fixture's, byte for byte; without it the same library is the
    plain CPU encoder (no GPU call at all)."""
    m, r, numbers = _encode_fixture(tmp_path, "ragop16_416x240_10b_q32", {"VVCGPU_SHIM_HOOKS": "pic"}, flag="--hipsel", extra=["--SIMD=HIP"])
    calls, line = numbers("[vvcgpu shim]")
    assert calls[0] == m["frames"] and calls[3] == m["frames"] and calls[4] == m["frames"] and calls[7] >= m["frames"], line
    (pics, ups, downs), res = numbers("[vvcgpu resident]")[0][:3], numbers("[vvcgpu resident]")[1]
    assert pics == m["frames"] and downs == m["frames"] and "resident form on" in res, res
    m, r, numbers = _encode_fixture(tmp_path, "ldp_208x120_10b_q27", {}, flag="--hipsel")
    calls, line = numbers("[vvcgpu shim]")
    assert sum(calls) == 0, line


@needs_sel
def test_patched_tree_selector_table_slots_and_decoder(tmp_path):
    """--SIMD=HIP at the default hook level: the five function-pointer tables carry the library's slots (64-wide distortion, interpolation and
    PelBuffer calls, ALF table slots); then the decoder of the same library on the fixture stream"""
    m, r, numbers = _encode_fixture(tmp_path, "ldp_208x120_10b_q27", {}, flag="--hipsel", extra=["--SIMD=HIP"])
    calls, line = numbers("[vvcgpu shim]")
    assert calls[0] == m["frames"] and calls[8] + calls[9] + calls[10] + calls[11] > 0, line
    # the "next" rows through their source hooks: transforms, fractional refinement, de-quantiser + T2, whole-PU TZ search, intra prediction, borders
    assert calls[12] + calls[13] > 0 and calls[14] > 0 and calls[16] > 0 and calls[20] > 0 and calls[21] > 0 and calls[22] > 0, line
    name = "ldp_208x120_10b_q27"
    out = str(tmp_path / "dec.yuv")
    r = subprocess.run([APP, "--hipsel", "dec", "-b", os.path.join(BS, name + ".bin"), "-o", out, "-d", str(m["bd"]), "--SIMD=HIP"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "ERROR" not in r.stdout and r.stdout.count("(OK)") >= m["frames"], r.stdout[-2000:]
    assert md5(out) == m["dec_yuv_md5"]
    line = [l for l in r.stderr.splitlines() if "[vvcgpu shim]" in l]
    assert line and int(line[-1].replace(",", " ").split("deblock")[1].split()[0]) >= m["frames"], r.stderr[-600:]
    calls = [int(x) for x in line[-1].replace(",", " ").split() if x.isdigit()]
    assert calls[21] > 0 and calls[25] > 0 and calls[22] > 0, line[-1]        # intra prediction, reference gathering, border extension on the device
