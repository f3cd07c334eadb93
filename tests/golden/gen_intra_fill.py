#!/usr/bin/env python3
"""Captures fixtures for the intra reference sample gathering from the COMPILED REFERENCE: runs the reference encoder (CPU only,
shim disabled) with VVCGPU_FILL_DUMP set, so that the drop-in shim's initIntraPatternChType hook records, for real calls of the
reference's own xFillReferenceSamples, the block shape, unit size, bit depth, the neighbour availability flags, the row above
and the column to the left of the block in the reconstruction, and the reference's unfiltered reference samples (packed).  At
most two calls per (shape, unit, availability pattern) and clip are kept.  -> tests/golden/intra_fill.npz.  Build container only."""
import os
import struct
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from vvcsoftware_vtm_amd import synth  # noqa: E402

APP = os.path.join(ROOT, "oracle", "_ref", "vtmref_app")


def capture(name, cfg, w, h, bd, frames, qp, seed):
    yuv, dump = "/tmp/fill_%s.yuv" % name, "/tmp/fill_%s.bin" % name
    if os.path.exists(dump):
        os.remove(dump)
    synth.write_yuv(yuv, synth.gen_yuv(w, h, frames, bd, seed), bd)
    env = dict(os.environ, VVCGPU_SHIM="0", VVCGPU_FILL_DUMP=dump)
    subprocess.check_call([APP, "--hip", "enc", "-c", os.path.join(ROOT, cfg), "-i", yuv, "-wdt", str(w), "-hgt", str(h), "-fr", "30", "-f", str(frames),
                           "-q", str(qp), "--InputBitDepth=%d" % bd, "--InternalBitDepth=%d" % bd, "--OutputBitDepth=%d" % bd, "-b", "/tmp/fill_%s.vvc" % name,
                           "-o", "/dev/null"], env=env, stdout=subprocess.DEVNULL)
    data = open(dump, "rb").read()
    recs, pos = [], 0
    while pos < len(data):
        hdr = struct.unpack_from("<8i", data, pos); pos += 32
        w_, h_, uw, uh, bd_, T, L, total = hdr
        flags = np.frombuffer(data, "u1", total, pos); pos += total
        aboveUnits, leftUnits = (T + uw - 1) // uw, (L + uh - 1) // uh
        topN, leftN = 1 + aboveUnits * uw, leftUnits * uh
        top = np.frombuffer(data, "<i2", topN, pos); pos += 2 * topN
        left = np.frombuffer(data, "<i2", leftN, pos); pos += 2 * leftN
        out = np.frombuffer(data, "<i2", T + L + 1, pos); pos += 2 * (T + L + 1)
        recs.append((hdr, flags, top, left, out))
    return recs


def main():
    recs = capture("ai8", "tests/golden/bitstreams/test_intra.cfg", 208, 120, 8, 1, 37, 20261013)
    recs += capture("ldp10", "tests/golden/bitstreams/test_lowdelay.cfg", 208, 120, 10, 2, 32, 20261014)
    out = {"hdr": np.array([r[0] for r in recs], np.int32)}
    for i, k in enumerate(("flags", "top", "left", "out")):
        out[k] = np.concatenate([r[i + 1] for r in recs])
    path = os.path.join(HERE, "intra_fill.npz")
    np.savez_compressed(path, **out)
    partial = sum(1 for r in recs if 0 < r[1].sum() < len(r[1]))
    print("intra_fill", len(recs), "records,", partial, "with partially available neighbours,", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
