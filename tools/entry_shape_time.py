#!/usr/bin/env python3
"""Per-shape throughput of one batch entry point on the committed encoder trace: every (w, h) of the entry's call signatures (the most frequent
parameter set of that shape) as ONE batch of `samples` samples, with the shape's share of the entry's samples -- which shapes the real mix pays for.
usage: python tools/entry_shape_time.py "if_batch" [samples] [max shapes]"""
import os
import sys
from collections import defaultdict

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vvcsoftware_vtm_amd import shape_mix as sm  # noqa: E402


def main():
    name = sys.argv[1]
    samples = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 22
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 24
    hist, _ = sm.load_trace()
    _, entry, keep, build = [r for r in sm.rows(hist) if r[0] == name][0]
    sig = sm.signatures(hist, entry, keep)
    share, best = defaultdict(int), {}
    for r in sig:
        k = (int(r[0]), int(r[1]))
        share[k] += int(r[5]) * k[0] * k[1]
        if k not in best or r[5] > best[k][5]:
            best[k] = r
    tot = sum(share.values())
    rng = np.random.default_rng(5)
    print("%s: one shape per batch, %d samples; share = samples of the shape among the entry's samples in the trace" % (name, samples))
    print("%-8s %7s %9s | %10s %12s" % ("shape", "share", "a,b,c", "ms", "Gsamples/s"))
    acc = 0.0
    for (w, h), v in sorted(share.items(), key=lambda kv: -kv[1])[:top]:
        n = max(1, samples // (w * h))
        calls = np.repeat(best[(w, h)][None, :5], n, axis=0)
        f, ns, _ = build(calls, rng)
        t = sm.gpu_ms(f)
        acc += t * v / tot
        print("%-8s %6.1f%% %9s | %10.4f %12.2f" % ("%dx%d" % (w, h), 100.0 * v / tot, ",".join(str(int(x)) for x in best[(w, h)][2:5]), t, ns / t * 1e-6))
    print("share-weighted ms per %d samples (listed shapes): %.4f" % (samples, acc))


if __name__ == "__main__":
    main()
