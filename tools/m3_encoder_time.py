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


LEGS = {  # label -> (use the drop-in library, VVCGPU_SHIM_HOOKS level, description)
    "cpu": (0, None, "own SIMD kernels, 1 host core"),
    "pic": (1, "pic", "library bound in, picture-level hooks only (resident reconstruction: deblock, SAO stats + apply, ALF classify + stats + filter)"),
    "pu":  (1, "pu", "picture-level hooks + whole-PU searches (xTZSearch, xPatternSearchFracDIF, xPatternSearch), one round trip per PU"),
    "pub": (1, "pub", "as pu, the uni-prediction searches of a CU (every list / reference pair) batched into ONE round trip on device-resident pictures"),
    "all": (1, "all", "every hook, block-level table slots included (one synchronous round trip per call: the proof form)"),
}


def run(m, leg, tmp, limits0=False):
    hip, level, _ = LEGS[leg]
    yuv = os.path.join(tmp, "in.yuv")
    if not os.path.exists(yuv):
        synth.write_yuv(yuv, synth.gen_yuv(m["w"], m["h"], m["frames"], m["bd"], m["seed"]), m["bd"])
    cfg = os.path.join(ROOT, m["cfg"][1:])
    binf = os.path.join(tmp, "out_%s.bin" % leg)
    cmd = [APP] + (["--hip"] if hip else []) + ["enc", "-c", cfg, "-i", yuv, "-wdt", str(m["w"]), "-hgt", str(m["h"]), "-fr", "30",
           "-f", str(m["frames"]), "-q", str(m["qp"]), "--InputBitDepth=%d" % m["bd"], "--InternalBitDepth=%d" % m["bd"],
           "--OutputBitDepth=%d" % m["bd"], "-b", binf, "-o", os.path.join(tmp, "rec.yuv"),
           "--SEIDecodedPictureHash=%d" % m.get("hash", 1)] + m.get("extra", [])
    env = dict(os.environ)
    if level:
        env["VVCGPU_SHIM_HOOKS"] = level
    if limits0:                                           # every call of the capped hooks is served (nightly-style run)
        for k in ("INTRA", "FILL", "DEPQUANT", "RDOQ", "DQIT", "TZ"):
            env["VVCGPU_SHIM_%s_LIMIT" % k] = "0"
    if leg in ("pu", "pub"):
        env["VVCGPU_SHIM_TZ_LIMIT"] = "0"                 # every integer search is served: the two forms are compared on the same calls
    t0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=3300, env=env)
    dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    shim = [l for l in r.stderr.splitlines() if "[vvcgpu shim]" in l or "[vvcgpu resident]" in l or "[vvcgpu batched]" in l or "mismatch" in l]
    return dt, md5(binf), shim


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--fixture", default="ragop16_416x240_10b_q32")
    ap.add_argument("--legs", default="cpu,pic,pu")
    ap.add_argument("--big", action="store_true", help="1920x1080, 3 pictures with the fixture's cfg (no committed bitstream: the legs are compared with the cpu leg)")
    ap.add_argument("--uncapped", action="store_true", help="lift the call caps of the block-level hooks (leg all)")
    ap.add_argument("--projection-only", action="store_true")
    ap.add_argument("bench", nargs="?")
    a = ap.parse_args()
    if a.bench:
        sys.argv = [sys.argv[0], a.bench]
    else:
        sys.argv = [sys.argv[0]]
    if a.projection_only:
        return projection()
    m = dict(json.load(open(os.path.join(BS, "manifest.json")))[a.fixture])
    want = m["bin_md5"]
    if a.big:
        m.update(w=1920, h=1080, frames=3)
        want = None
    print("M3: reference encoder (VTM 2.1 objects, unmodified), %s: %dx%d %d-bit, %d pictures, QP %d, cfg %s"
          % (a.fixture + (" (picture size raised)" if a.big else ""), m["w"], m["h"], m["bd"], m["frames"], m["qp"], m["cfg"]))
    with tempfile.TemporaryDirectory() as tmp:
        base = None
        for leg in a.legs.split(","):
            dt, h, shim = run(m, leg, tmp, a.uncapped and leg == "all")
            if want is None and leg == "cpu":
                want = h
            ok = "n/a (no cpu leg)" if want is None else str(h == want)
            if leg == "cpu":
                base = dt
            print("  (%-3s) %-70s: %8.2f s = %.3f pictures/s   bitstream identical: %s%s"
                  % (leg, LEGS[leg][2][:70], dt, m["frames"] / dt, ok, "" if base is None or leg == "cpu" else "   wall time / cpu leg = %.2f" % (dt / base)))
            for l in shim:
                print("        " + l.strip())
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
