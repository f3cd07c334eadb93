#!/usr/bin/env python3
"""Runs selected launch groups of the canonical workload a few times (for rocprofv3 / quick A-B timing).
usage: python3 tools/run_stage.py [--width W --height H] [--reps N] [--only substring]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from vvcsoftware_vtm_amd.workload import Workload  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=3840)
ap.add_argument("--height", type=int, default=2160)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--only", default="")
a = ap.parse_args()
wl = Workload(a.width, a.height, 10)


class Timer:
    def __init__(self):
        self.ev = {}

    def __call__(self, name):
        return Span(self, name)


class Span:
    def __init__(self, t, n):
        self.t, self.n = t, n

    def __enter__(self):
        self.a = torch.cuda.Event(enable_timing=True)
        self.b = torch.cuda.Event(enable_timing=True)
        self.a.record()

    def __exit__(self, *x):
        self.b.record()
        self.t.ev.setdefault(self.n, []).append((self.a, self.b))


st = None
t = Timer()
for i in range(a.reps + 1):
    st, out = wl.run_gpu(st, t)
torch.cuda.synchronize()
for k, v in t.ev.items():
    if a.only in k:
        ms = [x.elapsed_time(y) for x, y in v[1:]]
        print("%-34s %8.4f ms" % (k, sum(ms) / len(ms)))
