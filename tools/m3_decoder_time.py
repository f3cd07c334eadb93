#!/usr/bin/env python3
"""Measurement M3 of SURVEY 8(d), decoder side: the reference DECODER itself (oracle/_ref/vtmref_app dec = the reference's unmodified DecApp /
DecLib objects), timed on this box

  cpu : as shipped, its own SIMD kernels on one host core;
  pic : with the drop-in library bound in, picture-level hooks only -- the production form of the in-loop chain: the reconstruction is uploaded
        once per picture after CTU decoding, deblocking (maps from the reference's own CU walk), SAO and ALF (classification + filtering) run on
        the device-resident picture, the filtered picture comes down once (DecLib.cpp:506-533);
  all : plus every block-level hook (64-wide interpolation / PelBuffer slots, inverse transforms + de-quantiser per TU, intra prediction, picture
        hash): one synchronous round trip per call -- the proof form, not a performance form.

on the 1920x1080 10-bit random-access fixture stream (tests/golden/bitstreams/rab_1920x1080_10b_q32.bin: 9 pictures, hierarchical B, affine, ALF,
SAO; encoded by the compiled reference in the build container) and, for the fixed cost of a leg (process start, library load, HIP start-up, first
launches), on the 208x120 fixture of the same structure.  Decoded YUV md5 compared between the legs and with the manifest.

This is a measurement tool: it runs the compiled reference, which is not part of the product path.
usage: python tools/m3_decoder_time.py [--legs cpu,pic,all] [--reps 3] > profiles/rNN_m3_decoder.txt"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
APP = os.path.join(ROOT, "oracle", "_ref", "vtmref_app")
BS = os.path.join(ROOT, "tests", "golden", "bitstreams")
LEGS = {"cpu": (0, None), "pic": (1, "pic"), "all": (1, "all")}


def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 22), b""):
            h.update(blk)
    return h.hexdigest()


def decode(name, bd, leg, tmp):
    hip, level = LEGS[leg]
    out = os.path.join(tmp, "%s_%s.yuv" % (name, leg))
    cmd = [APP] + (["--hip"] if hip else []) + ["dec", "-b", os.path.join(BS, name + ".bin"), "-o", out, "-d", str(bd)]
    env = dict(os.environ)
    if level:
        env["VVCGPU_SHIM_HOOKS"] = level
    t0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=3000, env=env)
    dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    ok = r.stdout.count("(OK)")
    bad = r.stdout.count("***ERROR***")
    shim = [l for l in r.stderr.splitlines() if "[vvcgpu shim]" in l or "[vvcgpu resident]" in l]
    return dt, md5(out), ok, bad, shim


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--legs", default="cpu,pic,all")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--stream", default="rab_1920x1080_10b_q32")
    ap.add_argument("--small", default="rab_208x120_10b_q32")
    a = ap.parse_args()
    man = json.load(open(os.path.join(BS, "manifest.json")))
    big, small = man[a.stream], man[a.small]
    legs = a.legs.split(",")
    print("M3, decoder: the reference DecoderApp classes, %dx%d x %d pictures (%s, %d bytes), %d-bit; wall seconds, best of %d; one host core"
          % (big["w"], big["h"], big["frames"], a.stream, big["bytes"], big["bd"], a.reps))
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        for leg in legs:
            best, fixed = None, None
            for _ in range(a.reps):
                dt, m, ok, bad, shim = decode(a.stream, big["bd"], leg, tmp)
                assert m == big["dec_yuv_md5"], "%s: decoded YUV differs from the manifest" % leg
                assert bad == 0 and ok == big["frames"], "%s: picture hash SEI check failed (%d OK, %d errors)" % (leg, ok, bad)
                best = dt if best is None else min(best, dt)
                dts, ms, _, _, _ = decode(a.small, small["bd"], leg, tmp)
                assert ms == small["dec_yuv_md5"]
                fixed = dts if fixed is None else min(fixed, dts)
            res[leg] = (best, fixed, shim)
    cpu = res.get("cpu")
    print("%-4s %9s %9s %10s %9s   %s" % ("leg", "wall s", "fixed s", "net s", "net/pic ms", "decoded YUV md5 == manifest, every picture hash SEI (OK)"))
    for leg in legs:
        best, fixed, shim = res[leg]
        net = best - fixed
        line = "%-4s %9.3f %9.3f %10.3f %9.1f" % (leg, best, fixed, net, net / big["frames"] * 1e3)
        if cpu and leg != "cpu":
            cnet = cpu[0] - cpu[1]
            line += "   wall %.2fx, net %.2fx of cpu" % (best / cpu[0], net / cnet)
        print(line)
        for s in shim:
            print("       " + s.strip())
    print("fixed = the same leg on the %dx%d fixture of the same structure (%d pictures): process start, library load, HIP start-up and first launches dominate it"
          % (small["w"], small["h"], small["frames"]))


if __name__ == "__main__":
    main()
