#!/usr/bin/env python3
"""what a synchronous per-PU round trip of the drop-in shim is made of (M3 `pu` leg: 1.03 M of them at ~68 us): host->device copies of the
original block, of the search window and of the PU record, one small launch, the 32-byte result back, the stream synchronisation."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vvcsoftware_vtm_amd import capi, ops  # noqa: E402

lib = capi.lib()
torch.zeros(1, device="cuda")
pic = np.random.default_rng(0).integers(0, 1024, (1080 + 288, 1920 + 288)).astype(np.int16)
dwin = torch.empty(256 * 256, dtype=torch.int16, device="cuda")
dorg = torch.empty(64 * 64, dtype=torch.int16, device="cuda")
dsmall = torch.empty(64, dtype=torch.uint8, device="cuda")
host_small = np.zeros(64, np.uint8)
res = np.zeros(32, np.uint8)
P = lambda a: a.ctypes.data_as(C.c_void_p)
D = lambda t: C.c_void_p(t.data_ptr())


def timed(fn, n=2000):
    for _ in range(50):
        fn()
    lib.vvcgpu_stream_sync(None)
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    lib.vvcgpu_stream_sync(None)
    return (time.perf_counter() - t0) / n * 1e6


def win(w):
    return lambda: lib.vvcgpu_memcpy2d_h2d(D(dwin), C.c_size_t(w * 2), C.c_void_p(pic.ctypes.data + 2 * (300 * pic.shape[1] + 400)), C.c_size_t(pic.shape[1] * 2),
                                           C.c_size_t(w * 2), C.c_size_t(w), None)


print("h2d window 208x208 (pageable, strided)  %.1f us" % timed(win(208)))
print("h2d window 144x144                      %.1f us" % timed(win(144)))
print("h2d block 16x16 (strided)               %.1f us" % timed(lambda: lib.vvcgpu_memcpy2d_h2d(D(dorg), C.c_size_t(32), C.c_void_p(pic.ctypes.data), C.c_size_t(pic.shape[1] * 2), C.c_size_t(32), C.c_size_t(16), None)))
print("h2d 64 bytes                            %.1f us" % timed(lambda: lib.vvcgpu_memcpy_h2d(D(dsmall), P(host_small), C.c_size_t(64), None)))
print("d2h 32 bytes + sync                     %.1f us" % timed(lambda: (lib.vvcgpu_memcpy_d2h(P(res), D(dsmall), C.c_size_t(32), None), lib.vvcgpu_stream_sync(None))))
print("sync of an idle stream                  %.1f us" % timed(lambda: lib.vvcgpu_stream_sync(None)))
x = torch.zeros(64, device="cuda")
print("one small launch + sync                 %.1f us" % timed(lambda: (x.add_(1), lib.vvcgpu_stream_sync(None))))
for (wb, hh) in ((64, 1), (32, 2), (16, 4), (8, 8)):
    print("h2d 64 bytes as %2d x %d through the 2-D call   %.1f us" % (wb, hh, timed(lambda: lib.vvcgpu_memcpy2d_h2d(D(dsmall), C.c_size_t(wb), P(host_small), C.c_size_t(wb), C.c_size_t(wb), C.c_size_t(hh), None))))
big = np.zeros(4096, np.uint8)
dbig = torch.empty(4096, dtype=torch.uint8, device="cuda")
for (wb, hh) in ((4096, 1), (2048, 2), (64, 64)):
    print("h2d 4096 bytes as %4d x %2d                    %.1f us" % (wb, hh, timed(lambda: lib.vvcgpu_memcpy2d_h2d(D(dbig), C.c_size_t(wb), P(big), C.c_size_t(wb), C.c_size_t(wb), C.c_size_t(hh), None))))


def trip(use2d):
    def f():
        win(208)()
        lib.vvcgpu_memcpy2d_h2d(D(dorg), C.c_size_t(32), C.c_void_p(pic.ctypes.data), C.c_size_t(pic.shape[1] * 2), C.c_size_t(32), C.c_size_t(16), None)
        if use2d:
            lib.vvcgpu_memcpy2d_h2d(D(dsmall), C.c_size_t(64), P(host_small), C.c_size_t(64), C.c_size_t(64), C.c_size_t(1), None)
        else:
            lib.vvcgpu_memcpy_h2d(D(dsmall), P(host_small), C.c_size_t(64), None)
        x.add_(1)
        if use2d:
            lib.vvcgpu_memcpy2d_d2h(P(res), C.c_size_t(32), D(dsmall), C.c_size_t(32), C.c_size_t(32), C.c_size_t(1), None)
        else:
            lib.vvcgpu_memcpy_d2h(P(res), D(dsmall), C.c_size_t(32), None)
        lib.vvcgpu_stream_sync(None)
    return f


print("whole round trip (window + block + record up, launch, result down, sync): 1-D entry points %.1f us, 2-D calls for the small copies %.1f us"
      % (timed(trip(False), 1000), timed(trip(True), 1000)))
