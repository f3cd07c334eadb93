#!/usr/bin/env python3
"""Reduce a shim call trace (VVCGPU_SHIM_TRACE, vvcsoftware_vtm_amd/shim/vtm_hip_shim.cpp: six int32 per block-level call of the reference encoder --
entry, width, height, three parameters; shapes and parameters only) to the committed fixture tests/golden/trace_*.npz:
  hist   [n, 7] int64: entry, w, h, a, b, c, calls   -- every distinct call signature of the run with its call count
  first  [m, 6] int32: the first FIRST records of every entry in call order (the start of the first inter picture's CTU rows for the block entries)
  meta   json: the command the trace came from
usage: python tools/trace_reduce.py trace.bin out.npz "description of the encode" """
import json
import sys

import numpy as np

FIRST = 4096
ENTRIES = ["dist", "interp", "pelop", "tr_fwd", "tr_inv", "dequant_tr_inv", "depquant", "rdoq", "intra_pred"]


def main():
    src, dst, desc = sys.argv[1], sys.argv[2], sys.argv[3]
    rec = np.fromfile(src, dtype=np.int32).reshape(-1, 6)
    uniq, counts = np.unique(rec, axis=0, return_counts=True)
    hist = np.concatenate([uniq.astype(np.int64), counts[:, None].astype(np.int64)], axis=1)
    first = []
    for e in range(len(ENTRIES)):
        idx = np.nonzero(rec[:, 0] == e)[0][:FIRST]
        first.append(rec[idx])
    first = np.concatenate(first, axis=0)
    meta = {"description": desc, "records": int(rec.shape[0]), "entries": ENTRIES, "record": ["entry", "w", "h", "a", "b", "c"]}
    np.savez_compressed(dst, hist=hist, first=first, meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))
    for e, name in enumerate(ENTRIES):
        h = hist[hist[:, 0] == e]
        tot = int(h[:, 6].sum())
        if not tot:
            continue
        samples = int((h[:, 1] * h[:, 2] * h[:, 6]).sum())
        top = h[np.argsort(-h[:, 6])][:6]
        print("%-15s %10d calls %14d samples; most frequent: %s" % (name, tot, samples, ", ".join("%dx%d:%d" % (r[1], r[2], r[6]) for r in top)))


if __name__ == "__main__":
    main()
