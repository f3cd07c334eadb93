#!/usr/bin/env python3
"""Reduce rocprofv3 CSV output (gpurun_out/prof_*) to the per-launch-group summary committed under profiles/.

  python profiles/summarize.py gpurun_out r01

Groups dispatches by (kernel, grid size, LDS bytes) so that the six sad_search launches of one step stay separate; joins
the FETCH_SIZE / WRITE_SIZE passes (collected in their own runs) by the same key.  FETCH_SIZE/WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section) -- both the raw and the
corrected figure are listed."""
import csv
import os
import shutil
import sys
from collections import defaultdict


def short(name):
    n = name.replace("(anonymous namespace)::", "")
    return n.split("(")[0].replace("void ", "")


def load_trace(path):
    g = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            key = (short(r["Kernel_Name"]), int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1)) * int(r.get("Grid_Size_Z", 1)),
                   int(r.get("LDS_Block_Size", 0)))
            g[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return g


def load_pmc(path, counter):
    g = defaultdict(list)
    if not os.path.exists(path):
        return g
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            key = (short(r["Kernel_Name"]), int(r["Grid_Size"]), int(r.get("LDS_Block_Size", 0)))
            g[key].append(float(r["Counter_Value"]))
    return g


def main():
    src, tag = sys.argv[1], sys.argv[2]
    here = os.path.dirname(os.path.abspath(__file__))
    kt = os.path.join(src, "prof_kt")
    shutil.copy(os.path.join(kt, tag + "_kernel_stats.csv"), os.path.join(here, tag + "_kernel_stats.csv"))
    tr = load_trace(os.path.join(kt, tag + "_kernel_trace.csv"))
    fe = load_pmc(os.path.join(src, "prof_fetch", tag + "_counter_collection.csv"), "FETCH_SIZE")
    wr = load_pmc(os.path.join(src, "prof_write", tag + "_counter_collection.csv"), "WRITE_SIZE")
    # fabric read requests by size (round 3): the exact byte count behind FETCH_SIZE -- 32 n32 + 64 n64 + 128 n128 (the remainder of RDREQ, if the
    # size counters do not add up to it, is priced at 64 bytes) -- and fabric write requests (64-byte ones, the rest 32 bytes)
    rq = {c: load_pmc(os.path.join(src, "prof_rdreq", tag + "_counter_collection.csv"), c) for c in
          ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum")}
    wq = {c: load_pmc(os.path.join(src, "prof_wrreq", tag + "_counter_collection.csv"), c) for c in ("TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum")}
    mean = lambda v: (sum(v) / len(v)) if v else None
    rows = []
    for key, d in tr.items():
        f = fe.get(key, [])
        w = wr.get(key, [])
        rows.append((sum(d), key, len(d), sum(d) / len(d) / 1e3, (sum(f) / len(f)) if f else None, (sum(w) / len(w)) if w else None))
    rows.sort(reverse=True)
    by_size = {}
    for key in tr:
        n, n32, n64, n128 = (mean(rq[c].get(key, [])) for c in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"))
        wn, wn64 = (mean(wq[c].get(key, [])) for c in ("TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"))
        if n is not None and n32 is not None and n64 is not None and n128 is not None:
            rest = max(0.0, n - n32 - n64 - n128)
            rd = 32 * n32 + 64 * (n64 + rest) + 128 * n128
            wb = None if wn is None or wn64 is None else 64 * wn64 + 32 * max(0.0, wn - wn64)
            by_size[key] = (n, n32, n64, n128, rd, wb)
    if by_size:
        out2 = os.path.join(here, tag + "_fabric_requests.csv")
        with open(out2, "w") as fh:
            fh.write("kernel,grid_threads,lds_bytes,avg_us,rdreq,rdreq_32B,rdreq_64B,rdreq_128B,read_MB_by_size,FETCH_SIZE_MB_raw,read_by_size_over_FETCH_SIZE,write_MB_by_size,WRITE_SIZE_MB\n")
            for tot, key, n, avg, f, w in rows:
                if key not in by_size:
                    continue
                q = by_size[key]
                kname = '"%s"' % key[0] if "," in key[0] else key[0]
                fmb = None if f is None else f * 1024 / 1e6
                fh.write("%s,%d,%d,%.2f,%.0f,%.0f,%.0f,%.0f,%.2f,%s,%s,%s,%s\n" % (kname, key[1], key[2], avg, q[0], q[1], q[2], q[3], q[4] / 1e6,
                         "" if fmb is None else "%.2f" % fmb, "" if not fmb else "%.3f" % (q[4] / 1e6 / fmb),
                         "" if q[5] is None else "%.2f" % (q[5] / 1e6), "" if w is None else "%.2f" % (w * 1024 / 1e6)))
        print(open(out2).read())
    out = os.path.join(here, tag + "_launch_groups.csv")
    with open(out, "w") as fh:
        # fetch_MB_x2_corrected keeps its name (bench.py reads it) but holds the bytes BY REQUEST SIZE when that pass exists (round 3 on):
        # the blanket x2 of rounds 1 / 2 is right for 128-byte requests only
        fh.write("kernel,grid_threads,lds_bytes,calls,avg_us,fetch_KiB_raw,fetch_MB_x2_corrected,write_MB,hbm_GBps,hbm_frac_of_8TBps\n")
        for tot, key, n, avg, f, w in rows:
            kname = '"%s"' % key[0] if "," in key[0] else key[0]       # template argument lists contain commas: quote the field
            if key in by_size and f is not None:
                f = by_size[key][4] / 2048.0                             # so that 2 f KiB = the exact read bytes
            hbm = None if (f is None or w is None) else (2 * f + w) * 1024 / (avg * 1e-6) / 1e9       # counter traffic / time
            fh.write("%s,%d,%d,%d,%.2f,%s,%s,%s,%s,%s\n" % (kname, key[1], key[2], n, avg,
                                                             "" if f is None else "%.1f" % f,
                                                             "" if f is None else "%.2f" % (2 * f * 1024 / 1e6),
                                                             "" if w is None else "%.2f" % (w * 1024 / 1e6),
                                                             "" if hbm is None else "%.1f" % hbm,
                                                             "" if hbm is None else "%.4f" % (hbm / 8000.0)))
    print(open(out).read())
    # which kernel sources the profile was taken from: bench.py quotes `traffic` from it only while they are unchanged
    import json
    sys.path.insert(0, os.path.dirname(here))
    import bench
    with open(os.path.join(here, tag + "_meta.json"), "w") as fh:
        json.dump({"tag": tag, "lib_digest": bench.lib_digest(), "launch_groups": tag + "_launch_groups.csv",
                   "command": "rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --serial (tools/profile_round.sh)"}, fh, indent=1)


if __name__ == "__main__":
    main()
