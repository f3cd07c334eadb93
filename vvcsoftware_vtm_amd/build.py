"""Builds the HIP hot-path library in-tree: vvcsoftware_vtm_amd/lib/libvvcgpu.so (gfx950 only).

hipcc cross-compiles without a GPU, so this runs on the CPU-only build container; the .so travels to
the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored)."""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libvvcgpu.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fno-fast-math", "-ffp-contract=off",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def _digest(paths):
    h = hashlib.sha256()
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(p.encode())
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    hs += [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    return sorted(hs)


def build_lib(verbose=False, force=False):
    os.makedirs(OBJDIR, exist_ok=True)
    hdr = headers()
    jobs = []
    objs = []
    for src in sources():
        obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")
        stamp = obj + ".sha"
        dig = _digest([src] + hdr)
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
            continue
        jobs.append((src, obj, stamp, dig))

    def compile_one(job):
        src, obj, stamp, dig = job
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr))
        with open(stamp, "w") as f:
            f.write(dig)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    if jobs or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr)
    return LIB


if __name__ == "__main__":
    print(build_lib(verbose=True, force="--force" in sys.argv))
