#!/usr/bin/env python3
"""Multi-GPU exactness experiment (CPU only, build container): is chunk-parallel encoding by intra period byte-exact with the sequential encode?

  A = the reference encoder on N pictures, sequentially                                   (oracle/_ref/vtmref_app enc)
  B = the same encoder RE-ENTERED at an intra-period boundary with the reference's own mechanism: pictures with POC < P are taken from A
      (--DebugBitstream=A.bin --DebugPOC=P: decoded, not encoded), pictures with POC >= P are encoded fresh -- what a worker that starts at the
      chunk boundary with the hand-over pictures in its DPB does   (EncGOP.cpp:1146-1300)

The tool splits both streams into NAL units, pairs them in order, parses every slice header up to the reference-picture-set signalling
(HLSWriter::codeSliceHeader, EncoderLib/VLCWriter.cpp:891-988: first_slice_segment_in_pic_flag, [no_output_of_prior_pics_flag], pps id, slice type,
pic_order_cnt_lsb (8 bits), short_term_ref_pic_set_sps_flag, then the SPS index or an explicitly coded set) and reports, for every NAL unit whose
bytes differ, which field differs first.  Then the stitch test: the access units A coded before the boundary followed by the access units B coded
after it must decode (reference decoder, hash SEI checked) to exactly the pictures of A.

usage: python tools/chunk_exactness.py [--frames 65] [--poc 33] > profiles/rNN_chunk_exactness.txt"""
import argparse
import hashlib
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vvcsoftware_vtm_amd import synth  # noqa: E402

APP = os.path.join(ROOT, "oracle", "_ref", "vtmref_app")
CFG = os.path.join(ROOT, "tests", "golden", "bitstreams", "test_ra_gop16.cfg")


def md5(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


def nal_units(data):
    """Annex B byte stream -> list of NAL unit payloads (start codes stripped)"""
    out, i, n = [], 0, len(data)
    starts = []
    while i + 3 <= n:
        if data[i] == 0 and data[i + 1] == 0 and data[i + 2] == 1:
            starts.append(i + 3)
            i += 3
        else:
            i += 1
    for k, s in enumerate(starts):
        e = starts[k + 1] - 3 if k + 1 < len(starts) else n
        while e > s and data[e - 1] == 0:          # trailing zero bytes belong to the next start code
            e -= 1
        out.append(data[s:e])
    return out


class Bits:
    def __init__(self, nal):
        rb, z = bytearray(), 0
        for b in nal[2:]:                            # skip the two NAL header bytes; drop emulation prevention bytes
            if z >= 2 and b == 3:
                z = 0
                continue
            rb.append(b)
            z = z + 1 if b == 0 else 0
        self.d, self.p = bytes(rb), 0

    def u(self, n):
        v = 0
        for _ in range(n):
            v = (v << 1) | ((self.d[self.p >> 3] >> (7 - (self.p & 7))) & 1)
            self.p += 1
        return v

    def ue(self):
        z = 0
        while self.u(1) == 0:
            z += 1
        return (1 << z) - 1 + (self.u(z) if z else 0)


def se(b):
    k = b.ue()
    return (k + 1) >> 1 if k & 1 else -(k >> 1)


def pps_fields(nal):
    """HLSWriter::codePPS (VLCWriter.cpp:215-240): the fields the slice header parse depends on"""
    b = Bits(nal)
    p = {"pps_id": b.ue(), "sps_id": b.ue(), "output_flag_present": b.u(1), "num_extra_slice_header_bits": b.u(3), "cabac_init_present": b.u(1)}
    p["num_ref_idx_default"] = (b.ue() + 1, b.ue() + 1)
    return p


def slice_header(nal, num_rps_sps, pps=None, cfg=None):
    """HLSWriter::codeSliceHeader (VLCWriter.cpp:891-1303) as this repository's cfg produces it: no long-term pictures, no weighted
    prediction, no list modification, SAO / ALF / TMVP / DepQuant / QTBT / SubPuMvp on, slice chroma QP offsets present (dual tree),
    no deblocking control, loop filter across slices signalled.  With ALF switched on for the slice the parse stops at alf()."""
    t = (nal[0] >> 1) & 0x3F
    if t > 21 or (10 <= t <= 15):
        return None
    cfg = cfg or {}
    b = Bits(nal)
    h = {"nal_type": t, "first_slice": b.u(1)}
    if 16 <= t <= 23:
        h["no_output_of_prior_pics"] = b.u(1)
    h["pps_id"] = b.ue()
    h["slice_type"] = b.ue()                      # 0 B, 1 P, 2 I
    if t not in (19, 20):
        h["poc_lsb"] = b.u(8)
        h["rps_sps_flag"] = b.u(1)
        if h["rps_sps_flag"]:
            nb = 0
            while (1 << nb) < num_rps_sps:
                nb += 1
            h["rps_idx"] = b.u(nb) if nb else 0
        else:
            h["rps"] = "coded explicitly in the slice header"
            return h
        h["slice_temporal_mvp_enabled_flag"] = b.u(1)
    h["slice_sao_luma_flag"] = b.u(1)
    h["slice_sao_chroma_flag"] = b.u(1)
    h["alf_slice_enable_flag"] = b.u(1)
    if h["alf_slice_enable_flag"] or pps is None:
        h["parsed_to_bit"] = b.p
        return h
    intra = h["slice_type"] == 2
    nref = list(pps["num_ref_idx_default"])
    if not intra:
        h["num_ref_idx_active_override_flag"] = b.u(1)
        if h["num_ref_idx_active_override_flag"]:
            nref[0] = b.ue() + 1
            if h["slice_type"] == 0:
                nref[1] = b.ue() + 1
            h["num_ref_idx_active"] = tuple(nref)
        if h["slice_type"] == 0:
            h["mvd_l1_zero_flag"] = b.u(1)
        if pps["cabac_init_present"]:
            h["cabac_init_flag"] = b.u(1)
        if h.get("slice_temporal_mvp_enabled_flag"):
            col_l0 = 1
            if h["slice_type"] == 0:
                col_l0 = h["collocated_from_l0_flag"] = b.u(1)
            if nref[0 if col_l0 else 1] > 1:
                h["collocated_ref_idx"] = b.ue()
    h["dep_quant_enable_flag"] = b.u(1)
    if not h["dep_quant_enable_flag"]:
        h["sign_data_hiding_enable_flag"] = b.u(1)
    if not intra:
        h["max_binary_tree_unit_size"] = b.ue()
        h["seven_minus_max_num_merge_cand" if cfg.get("subpumvp", 1) else "five_minus_max_num_merge_cand"] = b.ue()
    h["slice_qp_delta"] = se(b)
    h["slice_cb_qp_offset"] = se(b)
    h["slice_cr_qp_offset"] = se(b)
    h["slice_loop_filter_across_slices_enabled_flag"] = b.u(1)
    if not intra and cfg.get("subpumvp", 1):
        h["slice_atmvp_subblk_size_enable_flag"] = b.u(1)
        if h["slice_atmvp_subblk_size_enable_flag"]:
            h["log2_slice_sub_pu_tmvp_size_minus2"] = b.u(3)
    h["alignment_ok"] = b.u(1) == 1 and all(b.u(1) == 0 for _ in range((-b.p) % 8))     # byte_alignment(): a one, then zeros
    h["header_bits"] = b.p
    return h


def encode(yuv, binf, rec, frames, extra, size, env=None):
    cmd = [APP, "enc", "-c", CFG, "-i", yuv, "-wdt", str(size[0]), "-hgt", str(size[1]), "-fr", "30", "-f", str(frames), "-q", "32", "--InputBitDepth=8",
           "--InternalBitDepth=8", "--OutputBitDepth=8", "-b", binf, "-o", rec, "--SEIDecodedPictureHash=1"] + extra
    t0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=3500, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return time.perf_counter() - t0, r.stdout


def run_with_record(frames=65, poc=33, keep=None, size=(416, 240), out=print):
    """The same experiment with the chunk hand-over SIDE RECORD (vvcsoftware_vtm_amd/shard.py SIDE_RECORD: the per-layer ATMVP statistics of the
    sequential encoder at the boundary, oracle/ref_wrap.cpp): A writes the record when it reaches the first slice with POC >= poc, C is re-entered
    at the boundary like B but installs the record before it codes that slice.  Returns (C byte-identical to A, record bytes)."""
    with tempfile.TemporaryDirectory() as tmp0:
        tmp = keep or tmp0
        os.makedirs(tmp, exist_ok=True)
        yuv = os.path.join(tmp, "in.yuv")
        synth.write_yuv(yuv, synth.gen_yuv(size[0], size[1], frames, 8, 20261013), 8)
        A, C, R = os.path.join(tmp, "A.bin"), os.path.join(tmp, "C.bin"), os.path.join(tmp, "record.bin")
        ta, _ = encode(yuv, A, os.path.join(tmp, "A_rec.yuv"), frames, [], size, env={"VVCGPU_ATMVP_POC": str(poc), "VVCGPU_ATMVP_DUMP": R})
        tc, _ = encode(yuv, C, os.path.join(tmp, "C_rec.yuv"), frames, ["--DebugBitstream=" + A, "--DebugPOC=%d" % poc], size,
                       env={"VVCGPU_ATMVP_POC": str(poc), "VVCGPU_ATMVP_LOAD": R})
        da, dc, rec = open(A, "rb").read(), open(C, "rb").read(), open(R, "rb").read()
        out("chunk exactness with the side record: %d pictures %dx%d, SubPuMvp 1 (fixture cfg), boundary POC %d" % (frames, size[0], size[1], poc))
        out("  A sequential encode (writes the record)  : %6.1f s, %d bytes, md5 %s" % (ta, len(da), hashlib.md5(da).hexdigest()))
        out("  C re-entered, record installed           : %6.1f s, %d bytes, md5 %s" % (tc, len(dc), hashlib.md5(dc).hexdigest()))
        import struct
        v = struct.unpack("<22I", rec)
        out("  record (%d bytes): subMergeBlkSize %s  subMergeBlkNum %s  prevPOC %d  clear %d" % (len(rec), list(v[:10]), list(v[10:20]), v[20], v[21]))
        out("  C byte-identical to A: %s" % (da == dc))
        return da == dc, rec


def decode(binf, out):
    r = subprocess.run([APP, "dec", "-b", binf, "-o", out, "-d", "8"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ERROR" not in r.stdout, r.stdout[-1500:]
    return r.stdout.count("(OK)")


def run(frames=65, poc=33, enc=(), keep=None, size=(416, 240), out=print):
    """returns (byte_exact, stitched_pictures_identical, first differing fields)"""
    a = argparse.Namespace(frames=frames, poc=poc, enc=list(enc), keep=keep, size=size)
    return experiment(a, out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=65)
    ap.add_argument("--poc", type=int, default=33)
    ap.add_argument("--enc", action="append", default=[], help="extra encoder option for both encodes, e.g. --enc=--SubPuMvp=0")
    ap.add_argument("--keep", default=None, help="directory to keep (and re-use) the streams in")
    ap.add_argument("--size", default="416x240")
    ap.add_argument("--record", action="store_true", help="the run with the hand-over side record instead (A writes it, C installs it)")
    a = ap.parse_args()
    a.size = tuple(int(v) for v in a.size.split("x"))
    if a.record:
        run_with_record(a.frames, a.poc, a.keep, a.size)
        return
    experiment(a, print)


def experiment(a, print):
    fields = []
    with tempfile.TemporaryDirectory() as tmp0:
        tmp = a.keep or tmp0
        os.makedirs(tmp, exist_ok=True)
        yuv = os.path.join(tmp, "in.yuv")
        synth.write_yuv(yuv, synth.gen_yuv(a.size[0], a.size[1], a.frames, 8, 20261013), 8)
        A, B = os.path.join(tmp, "A.bin"), os.path.join(tmp, "B.bin")
        ta = tb = float("nan")
        if not (os.path.exists(A) and os.path.exists(B)):
            ta, _ = encode(yuv, A, os.path.join(tmp, "A_rec.yuv"), a.frames, a.enc, a.size)
            tb, _ = encode(yuv, B, os.path.join(tmp, "B_rec.yuv"), a.frames, a.enc + ["--DebugBitstream=" + A, "--DebugPOC=%d" % a.poc], a.size)
        da, db = open(A, "rb").read(), open(B, "rb").read()
        print("chunk exactness: %d pictures %dx%d 8-bit," % (a.frames, a.size[0], a.size[1]) + " cfg test_ra_gop16.cfg (GOP 16, intra period 32), QP 32%s" % ((", " + " ".join(a.enc)) if a.enc else ""))
        print("  A sequential encode           : %6.1f s, %d bytes, md5 %s" % (ta, len(da), hashlib.md5(da).hexdigest()))
        print("  B re-entered at POC %-3d        : %6.1f s, %d bytes, md5 %s   (POC < %d decoded from A, the rest encoded)" % (a.poc, tb, len(db), hashlib.md5(db).hexdigest(), a.poc))
        na, nb = nal_units(da), nal_units(db)
        print("  NAL units: A %d, B %d; byte-identical streams: %s" % (len(na), len(nb), da == db))
        # the SPS of this cfg carries GOPSize + 1 reference picture sets (EncLib::xInitRPS: one per GOP entry + the intra set)
        num_rps = 17
        pps = next(pps_fields(u) for u in na if ((u[0] >> 1) & 0x3F) == 34)
        cfg = {"subpumvp": 0 if any("SubPuMvp=0" in e for e in a.enc) else 1}
        print("  PPS: %s" % pps)
        ndiff = 0
        first_b_idx = None
        for i, (x, y) in enumerate(zip(na, nb)):
            hx, hy = slice_header(x, num_rps, pps, cfg), slice_header(y, num_rps, pps, cfg)
            if hx and first_b_idx is None and "poc_lsb" in hx and hx["poc_lsb"] >= a.poc and hx["nal_type"] < 16:
                first_b_idx = i
            if x == y:
                continue
            ndiff += 1
            nbytes = sum(1 for p, q in zip(x, y) if p != q) + abs(len(x) - len(y))
            first = next((k for k, (p, q) in enumerate(zip(x, y)) if p != q), min(len(x), len(y)))
            print("  NAL %3d differs: %d byte(s), first at byte %d of the NAL unit, sizes %d / %d" % (i, nbytes, first, len(x), len(y)))
            if hx and hy:
                keys = [k for k in list(hx) + [k for k in hy if k not in hx] if hx.get(k) != hy.get(k) and k not in ("header_bits", "alignment_ok")]
                fields.append(keys[0] if keys else "slice data")
                print("      POC lsb %d, nal type %d: %s" % (hx.get("poc_lsb", 0), hx["nal_type"], "; ".join("%s A=%s B=%s" % (k, hx.get(k), hy.get(k)) for k in keys)
                                                             if keys else "slice header identical (the difference is in alf() or the slice data)"))
        if ndiff == 0:
            print("  no NAL unit differs: the re-entered encode is BYTE-EXACT")
        # stitch: A's NAL units coded before the first re-encoded picture + B's from there on
        cut = first_b_idx if first_b_idx is not None else len(na)
        # step back to the start of that access unit (its leading SEI / parameter-set NAL units: types >= 32)
        while cut > 0 and ((nb[cut - 1][0] >> 1) & 0x3F) >= 32:
            cut -= 1
        S = os.path.join(tmp, "S.bin")
        with open(S, "wb") as f:
            for u in na[:cut] + nb[cut:]:
                f.write(b"\x00\x00\x00\x01" + u)
        oka = decode(A, os.path.join(tmp, "A_dec.yuv"))
        oks = decode(S, os.path.join(tmp, "S_dec.yuv"))
        print("  stitch: A's first %d NAL units + B's remaining %d -> decoder: %d / %d picture hashes (OK); decoded YUV identical to A's: %s"
              % (cut, len(nb) - cut, oks, oka, md5(os.path.join(tmp, "S_dec.yuv")) == md5(os.path.join(tmp, "A_dec.yuv"))))
        same = md5(os.path.join(tmp, "S_dec.yuv")) == md5(os.path.join(tmp, "A_dec.yuv"))
        if os.path.exists(os.path.join(tmp, "A_rec.yuv")):
            print("  recon of A == recon of B (encoder side): %s" % (md5(os.path.join(tmp, "A_rec.yuv")) == md5(os.path.join(tmp, "B_rec.yuv"))))
        if fields:
            print("  FIRST differing slice-header field, in coding order: %s" % fields[0])
        return da == db, same and oks == oka == a.frames, fields


if __name__ == "__main__":
    main()
