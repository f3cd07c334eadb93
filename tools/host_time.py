import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vvcsoftware_vtm_amd.workload import Workload
wl = Workload(3840, 2160, 10)
st = None
for ov in (False, True):
    for _ in range(3):
        st, out = wl.run_gpu(st, overlap=ov)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        st, out = wl.run_gpu(st, overlap=ov)
    t1 = time.perf_counter()          # host has issued everything
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("overlap=%s host issue %.3f ms/step, wall %.3f ms/step" % (ov, (t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
