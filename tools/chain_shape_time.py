#!/usr/bin/env python3
"""Per-shape throughput of vvcgpu_resi_chain_batch and vvcgpu_dequant_tr_inv_batch: one batch of `samples` samples of ONE TU shape (DCT-II both ways,
the parameter mix of vvcsoftware_vtm_amd/shape_mix.py) per line, next to the share of that shape in the committed encoder trace -- which shapes the real
mix pays for (profiles/rNN_chain_shapes.txt).  usage: python tools/chain_shape_time.py [samples]"""
import os
import sys
from collections import defaultdict

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vvcsoftware_vtm_amd import shape_mix as sm  # noqa: E402


def main():
    samples = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 21
    hist, meta = sm.load_trace()
    h = hist[hist[:, 0] == sm.ENTRY["tr_fwd"]][:, 1:]
    share = defaultdict(int)
    for w, hh, a, b, c, n in h:
        if w >= 4 and hh >= 4:
            share[(int(w), int(hh))] += int(n) * int(w) * int(hh)
    tot = sum(share.values())
    rng = np.random.default_rng(5)
    print("one shape per batch, %d samples; share = samples of the shape among the forward-transformed samples of the trace" % samples)
    print("%-8s %7s | %10s %12s | %10s %12s" % ("shape", "share", "chain ms", "Gsamples/s", "dq+T2 ms", "Gsamples/s"))
    acc = [0.0, 0.0]
    for (w, hh), v in sorted(share.items(), key=lambda kv: -kv[1]):
        n = max(1, samples // (w * hh))
        calls = np.zeros((n, 6), np.int64)
        calls[:, 0], calls[:, 1] = w, hh
        f1, n1, _ = sm.build_chain(calls, rng)
        f2, n2, _ = sm.build_dqtr(calls, rng)
        t1, t2 = sm.gpu_ms(f1), sm.gpu_ms(f2)
        acc[0] += t1 * v / tot
        acc[1] += t2 * v / tot
        print("%-8s %6.1f%% | %10.4f %12.2f | %10.4f %12.2f" % ("%dx%d" % (w, hh), 100.0 * v / tot, t1, n1 / t1 * 1e-6, t2, n2 / t2 * 1e-6))
    print("share-weighted ms per %d samples: chain %.4f, dequant + inverse %.4f" % (samples, acc[0], acc[1]))


if __name__ == "__main__":
    main()
