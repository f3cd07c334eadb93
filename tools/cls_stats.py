import os
import sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vvcsoftware_vtm_amd.workload import Workload
wl = Workload(3840, 2160, 10)
st, out = wl.run_gpu(None, None)
torch.cuda.synchronize()
cls = out["cls"].cpu().numpy().view(np.uint16).reshape(2160 // 4, 3840 // 4)
print("distinct keys", len(np.unique(cls)))
print("horizontal neighbours equal", float((cls[:, 1:] == cls[:, :-1]).mean()))
print("vertical neighbours equal", float((cls[1:, :] == cls[:-1, :]).mean()))
c = cls & 0xff
print("class only: h", float((c[:, 1:] == c[:, :-1]).mean()), "v", float((c[1:] == c[:-1]).mean()))
vals, cnt = np.unique(cls, return_counts=True)
print("top keys share", np.sort(cnt)[::-1][:8] / cls.size)
