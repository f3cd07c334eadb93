import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oraclelib import oracle, p
from vvcsoftware_vtm_amd import ops
dev = lambda a: torch.from_numpy(a).cuda()
bd, mx, W, H = 10, 1023, 64, 64
yy, xx = np.mgrid[0:H, 0:W]
mode = sys.argv[1] if len(sys.argv) > 1 else "const"
r0 = (np.full((H, W), 512) if mode == "const" else (xx * 8) if mode == "xramp" else (yy * 8)).astype(np.int16)
rows = []; doff = 0
for (fx, fy, bi) in [(4, 0, 0), (4, 4, 0), (8, 12, 0), (4, 0, 1)]:
    for rep in range(2):
        rows.append((20 * W + 20, 20 * W + 20, doff, W, W, 8, 8, 8, fx, fy, fx, fy, 0, bi, 0)); doff += 64
d = np.array(rows, dtype=ops.MC_DESC)
want = np.full(doff, -5, np.int16)
oracle().orc_mc_batch(p(r0), p(r0), p(want), p(d), len(d), bd, 0, mx)
got = torch.full((doff,), -5, dtype=torch.int16, device="cuda")
ops.mc_batch(dev(r0), dev(r0), got, ops.struct_to_device(d), len(d), bd, (0, mx))
got = got.cpu().numpy()
for i in range(0, len(d), 2):
    print("desc", i, rows[i][8:10], "bi", rows[i][13]); print(got[i * 64:i * 64 + 64].reshape(8, 8)[:3]); print(want[i * 64:i * 64 + 64].reshape(8, 8)[:3])
