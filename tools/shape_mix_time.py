#!/usr/bin/env python3
"""Real-shape benchmark (VERDICT r2 #4b): every batch entry point on the call-signature mix of the committed encoder trace
(tests/golden/trace_ragop16_416x240_10b_q32.npz) against the same number of samples in 16 x 16 blocks.  See vvcsoftware_vtm_amd/shape_mix.py.
usage: python tools/shape_mix_time.py [samples] ["entry point,entry point"] > profiles/rNN_shape_mix.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vvcsoftware_vtm_amd import shape_mix  # noqa: E402


def main():
    samples = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 21
    hist, meta = shape_mix.load_trace()
    print("real-shape mix against squares; %d samples per batch; trace: %s (%d calls)" % (samples, meta["description"], meta["records"]))
    print("%-22s %9s %10s | %9s %12s | %9s %12s | %6s" % ("entry point", "calls", "<= 8 wide", "real ms", "Gsamples/s", "16x16 ms", "Gsamples/s", "ratio"))
    only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
    res = shape_mix.run(samples, only=only)
    for name, r in res.items():
        print("%-22s %9d %9.0f%% | %9.4f %12.2f | %9.4f %12.2f | %6.2f%s" % (name, r["calls"], 100 * r["calls_8_wide_or_less"], r["real_ms"], r["real_Gsamples_s"],
                                                                          r["square_ms"], r["square_Gsamples_s"], r["ratio"], "" if r["ratio"] >= 0.5 else "   < 0.5"))


if __name__ == "__main__":
    main()
