#!/usr/bin/env python3
"""Reduce the SQ counter passes of tools/profile_round.sh (gpurun_out/pmc_sq1, pmc_sq2: rocprofv3 --pmc ... -- python3
tools/run_stage.py) to one row per (kernel, grid, LDS) launch group: profiles/<tag>_pmc_sq.csv.

  python profiles/summarize_pmc.py gpurun_out r02

Columns: per-launch averages of every counter collected, plus
  valu_per_wave, salu_per_wave   SQ_INSTS_VALU / SQ_WAVES, SQ_INSTS_SALU / SQ_WAVES
  sq_busy_frac                   SQ_BUSY_CYCLES / (GRBM-free estimate: launch duration x 2.4 GHz x 32 SEs)   [indicative]
  issue_ns_per_valu              launch duration x 1024 SIMDs / SQ_INSTS_VALU   (4.4 cycles = 1.8 ns is the measured issue floor)
  lds_conflict_ratio             SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS   (extra LDS-array cycles per cycle an LDS instruction is active)
SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md, cycle constants)."""
import csv
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize import short  # noqa: E402


def load(path):
    cnt = defaultdict(lambda: defaultdict(list))
    if not os.path.exists(path):
        return cnt
    with open(path) as f:
        for r in csv.DictReader(f):
            key = (short(r["Kernel_Name"]), int(r["Grid_Size"]), int(r.get("LDS_Block_Size", 0)))
            cnt[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return cnt


def durations(path):
    d = defaultdict(list)
    if not os.path.exists(path):
        return d
    with open(path) as f:
        for r in csv.DictReader(f):
            g = int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1)) * int(r.get("Grid_Size_Z", 1))
            d[(short(r["Kernel_Name"]), g, int(r.get("LDS_Block_Size", 0)))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return d


def main():
    src, tag = sys.argv[1], sys.argv[2]
    here = os.path.dirname(os.path.abspath(__file__))
    a = load(os.path.join(src, "pmc_sq1", "p_counter_collection.csv"))
    b = load(os.path.join(src, "pmc_sq2", "p_counter_collection.csv"))
    # durations from the plain kernel trace of the bench run (un-profiled by counters) when present, else from the PMC pass
    dur = durations(os.path.join(src, "prof_kt", tag + "_kernel_trace.csv"))
    dur_pmc = durations(os.path.join(src, "pmc_sq1", "p_kernel_trace.csv"))
    names = sorted({n for c in (a, b) for k in c for n in c[k]})
    keys = sorted(set(a) | set(b), key=lambda k: -sum(dur.get(k, dur_pmc.get(k, [0]))))
    out = os.path.join(here, tag + "_pmc_sq.csv")
    with open(out, "w") as fh:
        fh.write("kernel,grid_threads,lds_bytes,avg_us," + ",".join(names) + ",valu_per_wave,salu_per_wave,issue_ns_per_valu,lds_conflict_ratio\n")
        for k in keys:
            d = dur.get(k) or dur_pmc.get(k) or [0]
            us = sum(d) / len(d) / 1e3
            vals = {}
            for c in (a, b):
                for n, v in c.get(k, {}).items():
                    vals[n] = sum(v) / len(v)
            waves = vals.get("SQ_WAVES", 0.0)
            valu, salu = vals.get("SQ_INSTS_VALU", 0.0), vals.get("SQ_INSTS_SALU", 0.0)
            kname = '"%s"' % k[0] if "," in k[0] else k[0]
            fh.write("%s,%d,%d,%.2f," % (kname, k[1], k[2], us) + ",".join("%.0f" % vals[n] if n in vals else "" for n in names))
            ldsA, ldsC = vals.get("SQ_ACTIVE_INST_LDS", 0.0), vals.get("SQ_LDS_BANK_CONFLICT", 0.0)
            fh.write(",%s,%s,%s,%s\n" % ("%.1f" % (valu / waves) if waves else "", "%.1f" % (salu / waves) if waves else "",
                                         "%.2f" % (us * 1e3 * 1024 / valu) if valu else "", "%.3f" % (ldsC / ldsA) if ldsA else ""))
    print(open(out).read())


if __name__ == "__main__":
    main()
