import os
import sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vvcsoftware_vtm_amd import ops
rng = np.random.default_rng(1)
for (w, h) in ((3840, 2160), (1920, 1080), (960, 544)):
    Y = torch.from_numpy(rng.integers(0, 1024, (h, w), dtype=np.int16)).cuda()
    for _ in range(6):
        cls = ops.alf_classify(Y, 10)
    torch.cuda.synchronize()
print("done")
