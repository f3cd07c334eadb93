import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vvcsoftware_vtm_amd import ops
W, H = 3840, 2160
resi = torch.randint(-300, 300, (H, W), dtype=torch.int16, device="cuda")
def run(sizes, label):
    rows = []; coff = 0
    ci = 0
    for y0 in range(0, H - H % 64, 64):
        for x0 in range(0, W - W % 64, 64):
            s = sizes[ci % len(sizes)]; ci += 1
            for ty in range(0, 64, s):
                for tx in range(0, 64, s):
                    rows.append(((y0 + ty) * W + x0 + tx, coff, W, s, s, (ci + tx) % 3 if s <= 32 else 0, (ci + ty) % 3 if s <= 32 else 0, 0, 0)); coff += s * s
    d = np.array(rows, dtype=ops.TR_DESC)
    dd = ops.struct_to_device(d)
    coef = torch.empty(coff, dtype=torch.int32, device="cuda")
    out = torch.empty_like(resi)
    for fn, name in ((lambda: ops.tr_fwd_batch(resi, coef, dd, d.size, 10), "fwd"), (lambda: ops.tr_inv_batch(coef, out, dd, d.size, 10), "inv")):
        for _ in range(3): fn()
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): fn()
        b.record(); torch.cuda.synchronize()
        print("%-10s %s n=%7d  %.4f ms" % (label, name, d.size, a.elapsed_time(b) / 10))
for s in (4, 8, 16, 32, 64):
    run([s], "all%d" % s)
run([4, 8, 16, 32, 64], "mix")
# launch-overhead floor: one tiny batch
rows = [(0, 0, W, 4, 4, 0, 0, 0, 0)]
d = np.array(rows, dtype=ops.TR_DESC); dd = ops.struct_to_device(d)
coef = torch.empty(16, dtype=torch.int32, device="cuda")
for _ in range(3): ops.tr_fwd_batch(resi, coef, dd, 1, 10)
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): ops.tr_fwd_batch(resi, coef, dd, 1, 10)
b.record(); torch.cuda.synchronize()
print("tiny fwd %.4f ms" % (a.elapsed_time(b) / 20))
import time
t0 = time.perf_counter()
for _ in range(200): ops.tr_fwd_batch(resi, coef, dd, 1, 10)
t1 = time.perf_counter(); torch.cuda.synchronize()
print("host time per call %.1f us" % ((t1 - t0) / 200 * 1e6))
# N1: de-quantisation + inverse transform on the mixed TU list of the canonical workload
rows = []; coff = 0; ci = 0
sizes = [4, 8, 16, 32, 64]
for y0 in range(0, H - H % 64, 64):
    for x0 in range(0, W - W % 64, 64):
        s = sizes[ci % 5]; ci += 1
        for ty in range(0, 64, s):
            for tx in range(0, 64, s):
                rows.append(((y0 + ty) * W + x0 + tx, coff, W, s, s, 0, 0, (ci + tx // s) & 1, 0, 32 + 12)); coff += s * s
d = np.array(rows, dtype=ops.DQTR_DESC); dd = ops.struct_to_device(d)
lv = (torch.randint(-20, 21, (coff,), dtype=torch.int32, device="cuda") * (torch.rand(coff, device="cuda") < 0.3)).to(torch.int32)
tmp = torch.empty(coff, dtype=torch.int32, device="cuda"); out = torch.empty_like(resi)
fn = lambda: ops.dequant_tr_inv_batch(lv, out, dd, d.size, 10, tmp)
for _ in range(3): fn()
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): fn()
b.record(); torch.cuda.synchronize()
print("dequant+inv mix n=%d  %.4f ms (levels 33 MB in, residual 16.6 MB out)" % (d.size, a.elapsed_time(b) / 10))
