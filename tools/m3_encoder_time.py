#!/usr/bin/env python3
"""Measurement M3 of SURVEY 8(d): the reference ENCODER itself (oracle/_ref/vtmref_app = the reference's unmodified
objects), timed on this box

  (a) as shipped: its own SIMD (AVX2) kernels on one host core, and
  (b) with the drop-in library bound in (`--hip`: ld --wrap shim + table slots + pre-empted functions, every hot-path
      call a synchronous round trip to the MI355X),

on the random-access fixture of this repository (hierarchical B, own cfg; the reference's cfg files do not travel),
bitstream md5 checked against the fixture in both legs.  Then the Amdahl projection 1 / ((1 - f) + f / s) the SURVEY asks
for: f = share of encoder time inside the hot path (gprof split of the reference encoder), s = M1 / M2 of
the batched kernels (bench.py: GPU pictures/s over the reference's SIMD kernels on one host core).

This is a test/measurement tool: it runs the compiled reference, which is not part of the product path.
usage: python tools/m3_encoder_time.py [--projection-only] [bench.json] > profiles/rNN_m3_encoder.txt"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vvcsoftware_vtm_amd import synth  # noqa: E402

APP = os.path.join(ROOT, "oracle", "_ref", "vtmref_app")
BS = os.path.join(ROOT, "tests", "golden", "bitstreams")
# gprof self-time split of the reference encoder (RA QP32 416x240, 17 frames)
SHARE_8A = {"transforms": 7.9, "SAD/SATD/SSE": 7.3, "PelBuf ops": 6.1, "DCTIF": 3.6, "deblock": 1.2, "SAO+ALF": 1.0}
SHARE_NEXT = {"DepQuant/Quant": 47.0, "affine gradient": 2.5, "intra prediction": 2.1}


def md5(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


def run(name, m, hip, tmp):
    yuv = os.path.join(tmp, "in.yuv")
    if not os.path.exists(yuv):
        synth.write_yuv(yuv, synth.gen_yuv(m["w"], m["h"], m["frames"], m["bd"], m["seed"]), m["bd"])
    cfg = os.path.join(ROOT, m["cfg"][1:])
    binf = os.path.join(tmp, "out_%d.bin" % hip)
    cmd = [APP] + (["--hip"] if hip else []) + ["enc", "-c", cfg, "-i", yuv, "-wdt", str(m["w"]), "-hgt", str(m["h"]), "-fr", "30",
           "-f", str(m["frames"]), "-q", str(m["qp"]), "--InputBitDepth=%d" % m["bd"], "--InternalBitDepth=%d" % m["bd"],
           "--OutputBitDepth=%d" % m["bd"], "-b", binf, "-o", os.path.join(tmp, "rec.yuv"),
           "--SEIDecodedPictureHash=%d" % m.get("hash", 1)] + m.get("extra", [])
    t0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=3000)
    dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    shim = [l for l in r.stderr.splitlines() if "[vvcgpu shim]" in l]
    return dt, md5(binf) == m["bin_md5"], (shim[-1] if shim else "")


def main():
    name = "rab_208x120_10b_q32"
    m = json.load(open(os.path.join(BS, "manifest.json")))[name]
    print("M3: reference encoder (VTM 2.1 objects, unmodified), fixture %s: %dx%d %d-bit, %d pictures, QP %d, hierarchical-B random access (own cfg)"
          % (name, m["w"], m["h"], m["bd"], m["frames"], m["qp"]))
    proj_only = "--projection-only" in sys.argv
    if proj_only:
        sys.argv.remove("--projection-only")
    with tempfile.TemporaryDirectory() as tmp:
        if proj_only:
            return projection()
        t_cpu, ok_cpu, _ = run(name, m, 0, tmp)
        print("  (a) own SIMD kernels, 1 host core           : %7.2f s  = %.3f pictures/s   bitstream == fixture: %s" % (t_cpu, m["frames"] / t_cpu, ok_cpu))
        t_hip, ok_hip, shim = run(name, m, 1, tmp)
        print("  (b) drop-in library bound in (per-call round trips): %7.2f s  = %.3f pictures/s   bitstream == fixture: %s" % (t_hip, m["frames"] / t_hip, ok_hip))
        print("      " + shim.strip())
        print("      (b)/(a) wall time = %.2f: the block-level hooks make one synchronous launch + copy per call and exist to prove the boundary;" % (t_hip / t_cpu))
        print("      the batched picture-level entry points (what bench.py times) are the production form.")
    projection()


def projection():
    f8a, fnext = sum(SHARE_8A.values()) / 100.0, sum(SHARE_NEXT.values()) / 100.0
    print("Amdahl projection (SURVEY 8(d) M3), f from the gprof self-time split of the reference encoder (SURVEY section 6):")
    print("  f(8(a) rows) = %.3f  %s" % (f8a, SHARE_8A))
    print("  f(8(a) + next rows N1/N3/N4) = %.3f  + %s" % (f8a + fnext, SHARE_NEXT))
    if len(sys.argv) > 1:
        b = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
        s = b["value"] / b["cpu_baseline"]["value"]
        print("  s = M1 / M2 = %.1f / %.4f = %.0f (batched kernels on one MI355X over the reference's SIMD kernels on one host core, %s)"
              % (b["value"], b["cpu_baseline"]["value"], s, os.path.basename(sys.argv[1])))
        for label, f in (("8(a) rows only", f8a), ("8(a) + next rows", f8a + fnext)):
            print("  projected whole-encoder speed-up, %-18s: 1 / ((1 - %.3f) + %.3f / %.0f) = %.2fx  (upper bound %.2fx as s -> inf)"
                  % (label, f, f, s, 1.0 / ((1 - f) + f / s), 1.0 / (1 - f)))
        print("  -> the hot path alone cannot reach the north star's 20x: the remaining %.0f %% (CABAC, mode control, partitioner) is the"
              % (100 * (1 - f8a - fnext)))
        print("     serial RDO control loop, out of scope for this tier (SURVEY section 0.3).  bench.py's value is M1, not encoder fps.")


if __name__ == "__main__":
    main()
