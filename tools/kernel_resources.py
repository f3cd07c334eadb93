#!/usr/bin/env python3
"""Register / LDS / scratch use of every kernel in a built object of vvcsoftware_vtm_amd/lib/obj (the gfx950 code object inside the clang offload
bundle, read with llvm-readelf --notes).  usage: python tools/kernel_resources.py resichain [name filter]"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    obj = os.path.join(ROOT, "vvcsoftware_vtm_amd", "lib", "obj", sys.argv[1] + ".o")
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    data = open(obj, "rb").read()
    i = data.find(b"__CLANG_OFFLOAD_BUNDLE__")
    n = struct.unpack_from("<Q", data, i + 24)[0]
    off = i + 32
    co = None
    for _ in range(n):
        o, sz, tl = struct.unpack_from("<QQQ", data, off)
        off += 24
        name = data[off:off + tl].decode()
        off += tl
        if "gfx950" in name:
            co = data[i + o:i + o + sz]
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(co)
        f.flush()
        txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
    cur = {}
    rows = []
    for ln in txt.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", ln)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count" and cur.get("name"):
            rows.append(cur)
            cur = {}
        cur[k] = v
    if cur.get("name"):
        rows.append(cur)
    print("%-60s %5s %5s %6s %7s %8s" % ("kernel", "vgpr", "agpr", "spill", "LDS", "scratch"))
    for r in rows:
        nm = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        nm = re.sub(r"\(.*", "", nm).replace("(anonymous namespace)::", "")
        if flt in nm:
            print("%-60s %5s %5s %6s %7s %8s" % (nm[:60], r.get("vgpr_count"), r.get("agpr_count"), r.get("vgpr_spill_count"), r.get("group_segment_fixed_size"),
                                                r.get("private_segment_fixed_size")))


if __name__ == "__main__":
    main()
