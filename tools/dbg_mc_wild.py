"""debug: the 'wild' case of tests/test_gpu_interp.py::test_mc_every_phase -- which PUs / samples differ (run on the GPU box)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from oraclelib import oracle, p
from vvcsoftware_vtm_amd import ops
import test_gpu_interp as T
dev = lambda a: torch.from_numpy(a).cuda()
bd, kind, W, clip = int(sys.argv[1]) if len(sys.argv) > 1 else 8, "wild", 392, None
rng = np.random.default_rng(1000 * bd + W)
mx = (1 << bd) - 1
H, M = 200, 8
mk = lambda: T._wild_plane(rng, H, W, bd, "smooth")
r0, r1 = mk(), mk()
rows = []
for (w, luma, nf, q) in [(16, 1, 16, 4), (8, 0, 32, 4)]:
    for fx in range(nf):
        for fy in range(nf):
            for bi in (0, 1):
                x0, y0 = int(rng.integers(M, W - w - M)), int(rng.integers(M, H - w - M))
                x1, y1 = int(rng.integers(M, W - w - M)), int(rng.integers(M, H - w - M))
                fx1, fy1 = (fx + q * int(rng.integers(0, nf // q))) % nf, (fy + q * int(rng.integers(0, nf // q))) % nf
                rows.append([y0 * W + x0, y1 * W + x1, 0, W, W, w, w, w, fx, fy, fx1, fy1, luma, bi, 0])
rows.append(rows[700][:])
order = rng.permutation(len(rows))
doff = 3
out = []
for k in order:
    r = rows[k]; r[2] = doff; doff += r[6] * r[7]; out.append(tuple(r))
d = np.array(out, dtype=ops.MC_DESC)
want = np.full(doff, -5, np.int16)
oracle().orc_mc_batch(p(r0), p(r1), p(want), p(d), len(d), bd, 0, mx)
got = torch.full((doff,), -5, dtype=torch.int16, device="cuda")
ops.mc_batch(dev(r0), dev(r1), got, ops.struct_to_device(d), len(d), bd, (0, mx))
got = got.cpu().numpy()
for i, r in enumerate(d):
    w = r["w"]
    a, b = got[r["dst_off"]:r["dst_off"] + w * w].reshape(w, w), want[r["dst_off"]:r["dst_off"] + w * w].reshape(w, w)
    if not np.array_equal(a, b):
        ys, xs = np.nonzero(a != b)
        print("PU", i, tuple(r), "differ at", list(zip(ys.tolist(), xs.tolist()))[:8])
        print(" got ", a[ys[0], max(0, xs[0] - 2):xs[0] + 3], " want", b[ys[0], max(0, xs[0] - 2):xs[0] + 3])
        for rf, (pl, off) in enumerate([(r0, r["ref0_off"]), (r1, r["ref1_off"])]):
            y0, x0 = divmod(int(off), W)
            win = pl[y0 - 3:y0 + w + 4, x0 - 3:x0 + w + 4]
            bad = np.nonzero((win < 0) | (win > mx))
            print("  ref%d window: out-of-range samples at" % rf, list(zip(bad[0].tolist(), bad[1].tolist())), win[bad])
