#!/usr/bin/env python3
"""bench.py -- hot-path throughput of the MI355X-native VTM pixel path (measurement M1 of SURVEY.md §8(d)).

A "step" is one INTRA PERIOD of the random-access structure: `--pictures-per-step` (default 32) passes of the canonical
per-picture hot-path workload (integer-ME SAD searches, fused fractional refinement, bi-pred MC, residual + forward
transform + quantiser + de-quantiser + inverse transform + reconstruction, deblocking, SAO stats+apply, ALF
classify+stats+filter) over 3840x2160 10-bit 4:2:0 pictures whose planes are resident in HBM, followed by the chunk
hand-over: the last reconstructed picture of the intra period becomes a reference picture of the NEXT chunk (copied into a
padded DPB slot, borders extended on the device, Picture::extendPicBorder).  N > 1: one process per GPU, rank r works on
intra periods r, r+N, ... (random-access intra periods shard with no data-path collective); the hand-over
picture then travels point-to-point to the next rank (RCCL send/recv over xGMI) once per step and is installed there.
value = pictures all ranks processed / max-over-ranks time.  This is NOT EncoderApp fps: the serial RDO control loop is
outside the path (M1 must never be presented as encoder fps).

`python bench.py --gpus N` without a torch.distributed environment starts the N ranks itself (torch.distributed.run, before
anything touches a GPU).  Prints ONE JSON line on rank 0."""
import argparse
import ctypes as C
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
# Issue model of the search kernels, per OPERATION (profiles/r03_valu_rate.txt, tools/micro/valu_rate.hip on MI355X): v_sad_u16 -- the only
# two-sample SAD instruction for 10-bit data -- and every other half-rate operation (v_bfi, v_alignbit, v_pk_*, v_dot2, v_max) issue once per
# 1.75 ns and SIMD from four waves per SIMD on, the whole chip busy (1.82 ns with two waves); full-rate operations (v_add_u32, v_and_b32)
# once per 0.94 ns.  The encoding (VOP2 / VOP3) is not what decides: v_add_u32_e64 1.0 ns, v_max_u32 (VOP2) 1.71 ns.
N_SIMD = 1024
SAD_ISSUE_NS = 1.75
SAD_ISSUE_PEAK = N_SIMD / (SAD_ISSUE_NS * 1e-9)          # v_sad_u16 wave-instructions per second, whole chip
# the same instruction inside the kernels' real stage body (LDS reads and scalar loads beside it: the chip drops to 2.0 - 2.26 GHz) costs
# 2.2 - 2.5 ns (profiles/r03_sadloop_rate.txt): no LDS-fed v_sad_u16 kernel gets beyond ~0.7 - 0.8 of SAD_ISSUE_PEAK
SAD_STAGE_BODY_NS = 2.46
SQ_CLOCK_HZ = 2.4e9                                        # SQ_ACTIVE_INST_* count quad-cycles: busy fraction below is against the maximum clock
BASELINE_METRIC = "encoded frames/sec (bit-exact bitstream) at 4K10 RA QP32, 1/2/4/8 GPU"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pictures-per-step", type=int, default=32, help="pictures of one intra period (one hand-over per step)")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--qp", type=int, default=32, help="base QP of the workload (BASELINE configs[2] sweeps 22 / 27 / 32 / 37): quantiser, de-quantiser, motion lambda and deblocking QP field follow it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-real-mix", action="store_true", help="skip the real-shape block (kernels_real_mix)")
    ap.add_argument("--serial", action="store_true", help="one stream, stage order: the default since round 4 (kept as a flag for older command lines)")
    ap.add_argument("--overlap", action="store_true",
                    help="independent stages on side streams beside the reconstruction chain (the default of rounds 1-3; measured slower than the serial "
                         "schedule once the integer search became one launch: 1363 vs 1426 pictures/s, profiles/r04_schedule.txt)")
    ap.add_argument("--rotate", type=int, default=12,
                    help="resident copies of the per-picture inputs (original + first reference picture) the pictures cycle through: 12 x 56 MB at 4K exceeds "
                         "the 256 MB memory-side cache, so the input reads and the counters behind hbm_frac are HBM-side (1 = every picture re-reads the same buffers)")
    ap.add_argument("--no-input-stream", action="store_true", help="skip the second timed region that uploads one original picture per picture from pinned host memory")
    ap.add_argument("--no-depquant-leg", action="store_true", help="skip the leg that runs the same pictures with the dependent-quantisation trellis (DepQuant 1 of the shipped cfgs) in place of the stand-in quantiser")
    ap.add_argument("--rehearse", action="store_true",
                    help="N > 1 on ONE GPU: every rank takes device (local rank mod device count) and the process group is gloo (RCCL refuses two ranks on "
                         "one device); the hand-over keeps the grouped form the RCCL run takes.  A rehearsal of the N > 1 control flow, not a measurement")
    ap.add_argument("--dry-launch", action="store_true",
                    help="launcher check without a GPU: start the ranks (gloo), verify the world size, hand one dummy boundary picture round the ring")
    return ap.parse_args()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args):
    """Parent of a multi-GPU run started as plain `python bench.py --gpus N`: starts N worker processes under
    torch.distributed.run as CHILDREN (this process has not touched a GPU and never execs) and returns their exit code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def lib_digest():
    """sha256 over the HIP sources of libvvcgpu.so: a committed rocprofv3 profile is only quoted when it was taken from the same kernels."""
    from vvcsoftware_vtm_amd import build
    h = hashlib.sha256()
    for p in build.sources() + build.headers():
        with open(p, "rb") as f:
            h.update(os.path.basename(p).encode())
            h.update(f.read())
    return h.hexdigest()[:16]


def profile_traffic():
    """-> ({launch-group kernel name: [(avg_us, hbm_bytes)]}, source file) from the newest committed rocprofv3 PMC summary
    (profiles/rNN_launch_groups.csv: FETCH_SIZE x2-corrected for gfx950 + WRITE_SIZE, collected in separate passes), or
    ({}, None) when that profile was taken from different kernel sources (profiles/rNN_meta.json `lib_digest`)."""
    import csv
    import glob
    metas = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_meta.json")))
    if not metas:
        return {}, None
    meta = json.load(open(metas[-1]))
    if meta.get("lib_digest") != lib_digest():
        return {}, None
    path = os.path.join(ROOT, "profiles", meta["launch_groups"])
    if not os.path.exists(path):
        return {}, None
    out = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["fetch_MB_x2_corrected"] and r["write_MB"]:
                out.setdefault(r["kernel"], []).append((float(r["avg_us"]), (float(r["fetch_MB_x2_corrected"]) + float(r["write_MB"])) * 1e6))
    return out, os.path.basename(path)


def profile_valu():
    """-> {kernel name: [(avg_us, SQ_INSTS_VALU per launch, SQ_ACTIVE_INST_VALU per launch)]} from the SQ counter summary of the same committed profile
    (profiles/rNN_pmc_sq.csv), {} when there is none for these kernel sources"""
    import csv
    import glob
    metas = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_meta.json")))
    if not metas:
        return {}
    meta = json.load(open(metas[-1]))
    path = os.path.join(ROOT, "profiles", meta.get("pmc_sq", meta.get("tag", "") + "_pmc_sq.csv"))
    if meta.get("lib_digest") != lib_digest() or not os.path.exists(path):
        return {}
    out = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get("SQ_INSTS_VALU") and float(r["avg_us"]) > 0:
                out.setdefault(r["kernel"], []).append((float(r["avg_us"]), float(r["SQ_INSTS_VALU"]), float(r.get("SQ_ACTIVE_INST_VALU") or 0.0)))
    return out


def profile_launches():
    """-> every kernel launch of ONE picture of the serial schedule from the committed rocprofv3 kernel trace of these kernel sources
    (profiles/rNN_launch_groups.csv: name, launches per picture, average us), the small fixed-cost ones included -- the launch groups of `kernels{}`
    bracket several launches each (org packing + memset + raster + key decode, classifier + chain + generic, fast + generic MC).  None without a
    matching profile."""
    import csv
    import glob
    import re
    metas = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_meta.json")))
    if not metas:
        return None
    meta = json.load(open(metas[-1]))
    path = os.path.join(ROOT, "profiles", meta["launch_groups"])
    if meta.get("lib_digest") != lib_digest() or not os.path.exists(path):
        return None
    rows, pictures = [], None
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["kernel"]
            if k.startswith("at::") or "elementwise" in k or k.startswith("__amd_rocclr"):
                continue                                             # torch kernels and runtime copies / fills (hand-over, state set-up): not launches of the library
            m = re.search(r"(\d*)([a-z][a-z0-9_]*_kernel)", k)        # readable name out of a (possibly mangled) symbol
            name = (m.group(2) if m else k[:48]) + (k[k.index("<"):k.index(">") + 1] if "<" in k and ">" in k else "")
            rows.append((name, int(r["calls"]), float(r["avg_us"])))
    # pictures of the profiled run = calls of a once-per-picture kernel
    for k, c, _ in rows:
        if "rc_chain_kernel" in k:
            pictures = c
    if not pictures:
        return None
    out = [{"kernel": k, "per_picture": round(c / pictures, 2), "avg_us": round(u, 2)} for k, c, u in rows if c >= pictures // 2]
    small = sum(e["per_picture"] * e["avg_us"] for e in out if e["avg_us"] < 15.0)
    return {"source": os.path.basename(path), "launches_per_picture": round(sum(e["per_picture"] for e in out), 1),
            "serial_us_per_picture": round(sum(e["per_picture"] * e["avg_us"] for e in out), 1), "launches_under_15us_total_us": round(small, 1), "launches": out}


def cpu_baseline(wl, budget_s=25.0):
    """VTM's own SIMD kernels (oracle/_ref/libvtmref.so, kind 'reference') or the scalar restatement (kind 'port') on ONE host
    core over the bench's own workload object (the whole 3840x2160 picture, no extrapolation) when one pass fits the budget;
    the scalar port falls back to a 960x544 picture of the same workload scaled by pixel count."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from workload_cpu import run_cpu                      # checker side (tests/), timed here as the reported CPU baseline
    odir = os.path.join(ROOT, "oracle")
    port_so = os.path.join(odir, "liboracle.so")
    ref_so = os.path.join(odir, "_ref", "libvtmref.so")
    if not os.path.exists(port_so):
        return None
    port = C.CDLL(port_so)
    kind, lib = "port", port
    if os.path.exists(ref_so):
        try:
            lib, kind = (port, C.CDLL(ref_so)), "reference"
        except OSError:
            pass
    scale, sample_wl = 1.0, wl
    if kind == "port":
        from vvcsoftware_vtm_amd.workload import Workload
        sample_wl = Workload(960, 544, wl.bd, seed=wl.seed, qp=wl.qp)
        scale = (wl.w * wl.h) / float(960 * 544)
    t0 = time.perf_counter()
    reps, secs_tot = 0, {}
    while True:
        _, secs = run_cpu(sample_wl, lib, kind)
        for k, v in secs.items():
            secs_tot[k] = secs_tot.get(k, 0.0) + v
        reps += 1
        el = time.perf_counter() - t0
        if el + el / reps > budget_s:
            break
    per_picture = sum(secs_tot.values()) / reps * scale
    what = ("the bench's own %dx%d workload object, every stage, %d repetition(s)" % (wl.w, wl.h, reps) if scale == 1.0 else
            "same workload on a 960x544 picture (%d repetitions), scaled by pixel count to %dx%d" % (reps, wl.w, wl.h))
    return {"value": 1.0 / per_picture, "unit": "frames/s", "cores": 1, "kind": kind,
            "sample": "%s, %.1f s of CPU work; deblocking and plane add/subtract use the scalar port (no reference entry point); "
                      "stage seconds per picture: %s" % (what, sum(secs_tot.values()), {k: round(v / reps * scale, 4) for k, v in secs_tot.items()})}


def dry_launch(args):
    """Launcher check on CPU (gloo): every rank joins, the world size equals --gpus, one dummy boundary picture goes round the ring."""
    import torch
    import torch.distributed as dist
    from vvcsoftware_vtm_amd import shard
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    if world > 1:
        dist.init_process_group("gloo")
        assert dist.get_world_size() == args.gpus
    planes = [torch.full((8, 8), rank, dtype=torch.int16), torch.full((4, 4), rank + 100, dtype=torch.int16)]
    rec = shard.empty_side_record()
    rec["sub_merge_blk_num"][0, 3] = 1000 + rank
    # the form the RCCL run takes: ONE grouped batch_isend_irecv per hand-over (forced on gloo here), nothing posted early
    h = shard.Handover(planes, rank, world, batched=True).post_recv()
    early = list(h.issued)
    got, grec = h.send(planes, rec).wait()
    ok = int(got[0][0, 0]) == (rank - 1) % world and int(got[1][0, 0]) == (rank - 1) % world + 100
    ok = ok and (world == 1 or int(grec["sub_merge_blk_num"][0, 3]) == 1000 + (rank - 1) % world)
    ok = ok and early == [] and (world == 1 or h.issued == [("batch", 6)])
    # and the plain gloo form (receives posted at chunk start) gives the same picture
    got2 = shard.exchange_boundary(planes, rank, world, record=rec)
    ok = ok and bool((got2[0] == got[0]).all())
    seen = [None] * world
    if world > 1:
        dist.all_gather_object(seen, (rank, ok))
        dist.destroy_process_group()
    else:
        seen = [(rank, ok)]
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": world, "ranks_seen": sorted(r for r, _ in seen), "handover_ok": all(o for _, o in seen),
                          "p2p": "one batch_isend_irecv per hand-over"}))


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))                      # nothing GPU-related has been imported or called at this point
    if args.dry_launch:
        return dry_launch(args)

    import numpy as np
    import torch
    import torch.distributed as dist
    from vvcsoftware_vtm_amd import capi, shard
    from vvcsoftware_vtm_amd.workload import Workload

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d (start it as `python bench.py --gpus %d`, or under "
                         "torch.distributed.run --nproc-per-node %d)" % (args.gpus, world, args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if args.rehearse:
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    capi.call("vvcgpu_set_device", local_rank)
    if world > 1:
        if args.rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: RCCL sees %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))

    bd = 10
    wl = Workload(args.width, args.height, bd, seed=20261003 + rank, qp=args.qp)
    alg = wl.algorithmic_bytes()
    pps = args.pictures_per_step

    class Timer:
        def __init__(self):
            self.ev = {}
            self.on = False
            self.only = None          # when set: only this launch group is bracketed by events

        def __call__(self, name):
            return _Span(self, name)

    class _Span:
        def __init__(self, t, name):
            self.t, self.name = t, name

        def __enter__(self):
            self.rec = self.t.on and (self.t.only is None or self.t.only == self.name)
            if self.rec:
                self.a = torch.cuda.Event(enable_timing=True)
                self.b = torch.cuda.Event(enable_timing=True)
                self.a.record()
            return self

        def __exit__(self, *a):
            if self.rec:
                self.b.record()
                self.t.ev.setdefault(self.name, []).append((self.a, self.b))
            return False

    timer = Timer()
    overlap = bool(args.overlap) and not args.serial

    # ---- picture 0 on a fresh state: the md5 the 4K parity test pins (tests/golden/bench_md5.json), then one serial, bracketed
    # picture to find the dominant launch group (in the timed region only THAT group carries HIP events: an event pair per
    # launch group would put ~40 markers into every picture)
    state, out = wl.run_gpu(None, None, overlap=overlap)
    torch.cuda.synchronize()
    first_md5 = shard.picture_hash(out["final"])
    timer.on = True
    state, out = wl.run_gpu(state, timer, overlap=False)
    torch.cuda.synchronize()
    warm = {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in timer.ev.items()}
    warm = {k: v for k, v in warm.items() if k.split('/')[1] in alg.get(k.split('/')[0], {})}
    timer.only = max(warm, key=warm.get)
    timer.ev, timer.on = {}, False
    # the overlapped schedule launches ONE search group first and alone: make it the dominant one when that is a search, so that its
    # event-timed duration in the timed region is a kernel time (any other dominant group is quoted from the serial pass below)
    alone = None
    if timer.only.startswith("me/sad_search_"):
        sz, grid = timer.only[len("me/sad_search_"):].split("_")
        alone = (int(sz.split("x")[0]), 0 if grid == "9x9" else 1)
    # the hierarchical search IS the first launch of a picture, on the main stream, behind the join of the previous picture's side streams: alone
    dom_alone = alone is not None or timer.only == "me/hier_search"
    rotate = max(1, args.rotate)

    # ---- input-stream leg: one original picture (24.9 MB at 4K) per picture from pinned host memory on a copy stream, double-buffered through the
    # rotating input sets -- the upload for picture i + 1 runs beside picture i; the picture that uses a set waits for that set's upload only
    up = {"on": False, "ev": [None] * rotate, "host": None, "side": {}, "stream": None, "bytes": 0, "side_bytes": 0}

    def on_input_set(k, st_):
        if not up["on"]:
            return
        main = torch.cuda.current_stream()
        if up["ev"][k] is not None:
            main.wait_event(up["ev"][k])                   # this picture's original has arrived
            up["ev"][k] = None
        if rotate < 2:
            return
        nxt = (k + 1) % rotate                             # last read by picture i + 1 - rotate, complete on `main` (every picture joins its side streams)
        e = torch.cuda.Event()
        e.record(main)
        up["stream"].wait_event(e)
        with torch.cuda.stream(up["stream"]):
            for dst, src in zip(st_["in_sets"][nxt][0], up["host"]):
                dst.copy_(src, non_blocking=True)
            # ... and the side information an encoder derives per picture: PU / TU descriptor lists, deblocking maps, SAO parameters, ALF switches
            for key, src in up["side"].items():
                dst = st_["side_sets"][nxt][key]
                for d_, s_ in zip(dst if isinstance(dst, list) else [dst], src if isinstance(src, list) else [src]):
                    d_.view(torch.uint8).reshape(-1).copy_(s_, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(up["stream"])
        up["ev"][nxt] = ev

    pending = [None]

    def install_pending():
        """in front of the first motion compensation that reads the boundary picture: the current (main) stream waits for the transfer, then the
        picture is copied into the padded reference slot and border-extended on the device"""
        if pending[0] is not None:
            boundary, _rec = pending[0].wait()
            shard.install_reference(boundary, state["ref1"], wl.margins())
            pending[0] = None

    def one_step(tm):
        """one intra period: pps pictures, then the chunk hand-over (next rank's reference picture + the 88-byte side record; own picture at N = 1).
        The receive from rank r - 1 and the send to rank r + 1 are issued as ONE grouped RCCL operation when the boundary picture exists
        (shard.Handover: no unbatched point-to-point on the eagerly initialised group); the wait comes only where the NEXT chunk first needs the
        picture -- in front of the motion compensation of its first picture, behind that picture's searches -- so a rank never waits at a step boundary."""
        nonlocal state, out
        h = shard.Handover(out["final"], rank, world, batched=True if args.rehearse else None).post_recv()
        for i in range(pps):
            state, out = wl.run_gpu(state, tm, overlap=overlap, alone=alone, pre_mc=install_pending if i == 0 else None, rotate=rotate, on_input_set=on_input_set)
        install_pending()                                   # (pps == 0 guard; a no-op otherwise)
        h.send(out["final"], shard.empty_side_record())     # (the kernels carry no encoder statistics: the record of a fresh encoder travels)
        pending[0] = h

    def finish_steps():
        install_pending()

    for _ in range(args.warmup):
        one_step(None)
    finish_steps()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timer.on = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step(timer)
    finish_steps()                                          # the last hand-over belongs to the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    timed_groups = dict(timer.ev)

    # ---- the same K steps once more WITH the input stream (reported beside `value`, never as `value`: the contract's figure has the inputs resident)
    dt_up = None
    if not args.no_input_stream and rotate >= 2:
        up["host"] = [torch.from_numpy(np.ascontiguousarray(p_)).pin_memory() for p_ in wl.org]
        pin = lambda a: torch.from_numpy(np.ascontiguousarray(a)).pin_memory()
        up["side"] = {k: ([pin(a) for a in v] if isinstance(v, list) else pin(v)) for k, v in wl.side_host().items()}
        up["side_bytes"] = sum(sum(int(t_.numel()) for t_ in (v if isinstance(v, list) else [v])) for v in up["side"].values())
        up["bytes"] = sum(int(t_.numel()) * 2 for t_ in up["host"]) + up["side_bytes"]
        up["stream"] = torch.cuda.Stream()
        up["on"] = True
        timer.on = False
        one_step(None)                                      # untimed: fills the upload pipeline
        finish_steps()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            one_step(None)
        finish_steps()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt_up = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([dt_up], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_up = float(t.item())
        up["on"] = False
        torch.cuda.current_stream().wait_stream(up["stream"])
        timer.on = True

    # ---- with_depquant leg: the shipped configurations run `DepQuant 1` (cfg/encoder_randomaccess_vtm.cfg); the headline workload uses the
    # Quant::quant stand-in SURVEY 8(d) allows.  The same pictures once more with vvcgpu_depquant_batch + the dependent-quantisation de-quantiser between the
    # separate transform entry points -- reported beside `value`, never as `value`.  A bounded number of pictures: the trellis is the serial walk of its longest TU.
    dq_leg = None
    if rank == 0 and world == 1 and not args.no_depquant_leg:
        wl_dq = Workload(args.width, args.height, bd, seed=20261003 + rank, qp=args.qp, depquant=True)
        tdq = Timer()
        st_dq, _ = wl_dq.run_gpu(None, None, overlap=False)
        torch.cuda.synchronize()
        n_dq = max(4, min(pps, 16))
        t2 = time.perf_counter()
        for _ in range(n_dq):
            st_dq, o_dq = wl_dq.run_gpu(st_dq, None, overlap=False)
        torch.cuda.synchronize()
        dt_dq = time.perf_counter() - t2
        tdq.on = True
        for _ in range(3):
            st_dq, o_dq = wl_dq.run_gpu(st_dq, tdq, overlap=False)
        torch.cuda.synchronize()
        g_dq = {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in tdq.ev.items()}
        dq_leg = {"value": n_dq / dt_dq, "unit": "frames/s", "ms_per_picture": dt_dq / n_dq * 1e3, "pictures_timed": n_dq,
                  "resi_stage_ms": {k.split("/")[1]: round(v, 4) for k, v in g_dq.items() if k.startswith("resi/")},
                  "last_picture_md5": shard.picture_hash(o_dq["final"]),
                  "what": "the same per-picture workload with the quantiser of the shipped cfgs: residual -> vvcgpu_tr_fwd_batch -> vvcgpu_depquant_batch (DepQuant::quant, "
                          "eight seeded rate tables, lambda of the QP) -> vvcgpu_dequant_tr_inv_batch (dep_quant = 1) -> reconstruction, in place of the one-pass chain with "
                          "Quant::quant; parity of this leg: tests/test_gpu_workload.py::test_workload_depquant_leg_matches_oracle_416x240.  Reported beside `value`, never as it"}
        del wl_dq, st_dq, o_dq
        torch.cuda.empty_cache()

    # final gather of per-picture output hashes (control path, outside the timed region)
    hashes = shard.gather_hashes({"rank%d" % rank: shard.picture_hash(out["final"])}, world)

    # dominant kernel: device time over the timed region (HIP events on the stream the kernels were launched on)
    timed_ms = {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in timed_groups.items()}
    n_timed = {k: len(v) for k, v in timed_groups.items()}
    # table of all launch groups: a separate, untimed pass (serial schedule) with an event pair around every group
    dom_name = timer.only
    timer.only, timer.ev = None, {}
    for _ in range(5):
        state, out = wl.run_gpu(state, timer, overlap=False, rotate=rotate)
    torch.cuda.synchronize()
    kern_ms = {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in timer.ev.items()}
    if dom_alone or not overlap:
        kern_ms.update(timed_ms)                          # measured over the timed region, launched alone
    else:
        n_timed = {}                                      # dominant group runs beside other kernels in the overlapped schedule: serial-pass time
    stage_ms = {}
    for k, v in kern_ms.items():
        stage_ms[k.split("/")[0]] = stage_ms.get(k.split("/")[0], 0.0) + v
    dom = dom_name
    dstage, dname = dom.split("/")
    abytes = alg[dstage][dname]
    achieved = abytes / (kern_ms[dom] * 1e-3) / 1e9
    uniq = wl.unique_bytes()
    useful = wl.useful_sad_insts()
    prof, prof_src = profile_traffic() if rank == 0 else ({}, None)
    pvalu = profile_valu() if rank == 0 else {}
    hint = wl.profile_kernel_hint(dom)

    def valu_busy_of(group, ms):
        """fraction of the SIMD cycles in which a vector instruction of the launch group's main kernel executes: SQ_ACTIVE_INST_VALU (quad-cycles,
        summed over the SIMDs) of the committed counter profile (the row whose duration is closest and within 35 %) x 4 / (1024 SIMDs x time x 2.4 GHz).
        A measured busy count, so it cannot exceed 1; it reads LOW when the chip runs below 2.4 GHz (the searches hold ~2.0 GHz)."""
        h = wl.profile_kernel_hint(group)
        cands = [c for k, v in pvalu.items() if h and h in k for c in v]
        if not cands:
            return None, None
        best = min(cands, key=lambda c: abs(c[0] - ms * 1e3))
        if abs(best[0] - ms * 1e3) > 0.35 * ms * 1e3 or best[2] <= 0:
            return None, None
        return min(1.0, best[2] * 4.0 / (N_SIMD * best[0] * 1e-6 * SQ_CLOCK_HZ)), best[1]

    def hbm_of(group, ms):
        """HBM bytes per launch of the dominant launch group's main kernel from the committed profile: rows of that kernel, the one
        whose average duration is closest (and within 35 %)"""
        if group != dom or hint is None:
            return None
        cands = [c for k, v in prof.items() if hint in k for c in v]
        if not cands:
            return None
        best = min(cands, key=lambda c: abs(c[0] - ms * 1e3))
        return best[1] if abs(best[0] - ms * 1e3) <= 0.35 * ms * 1e3 else None

    per_kernel = {}
    for k, v in kern_ms.items():
        s, n = k.split("/")
        e = {"ms": round(v, 4)}
        if n in alg.get(s, {}):
            if k in useful:                                    # a search: the SURVEY 8(d) per-PU sum is not traffic (one staged window serves thousands of PUs)
                e["per_pu_byte_sum_MB_not_traffic"] = round(alg[s][n] / 1e6, 2)
            else:
                e.update({"alg_MB": round(alg[s][n] / 1e6, 2), "alg_GBps": round(alg[s][n] / (v * 1e-3) / 1e9, 1)})
        if n in uniq.get(s, {}):
            e["unique_MB"] = round(uniq[s][n] / 1e6, 2)
            e["unique_frac"] = round(uniq[s][n] / (v * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        hb = hbm_of(k, v)
        if hb is not None:
            e["hbm_MB"] = round(hb / 1e6, 2)
            e["hbm_frac"] = round(hb / (v * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        if k in useful:
            e["issue_frac"] = round(useful[k] / (v * 1e-3) / SAD_ISSUE_PEAK, 4)
        vb, ninst = valu_busy_of(k, v)
        if vb is not None:
            e["valu_busy"] = round(vb, 4)
            e["valu_insts_M"] = round(ninst / 1e6, 2)
            if k in useful:
                e["useful_share_of_valu"] = round(useful[k] / ninst, 4)       # needed v_sad_u16 / executed vector instructions
        per_kernel[k] = e

    if rank == 0:
        pictures = args.steps * pps * world
        dk = per_kernel[dom]
        ms_pic = dt / (args.steps * pps) * 1e3
        uniq_total = sum(v for st_ in uniq.values() for v in st_.values() if isinstance(v, (int, float)))
        # roofline{}: ONE bound, and frac against THAT bound's peak, never above 1.
        #   a search launch: bound "valu" -- useful (non-redundant) v_sad_u16 wave-instructions of the launch / its HIP-event duration against the
        #     issue peak of the instruction, one per 4 cycles and SIMD at 2.4 GHz (1024 SIMDs: 614.4 G/s); its bytes are the UNION of the windows + the
        #     original + the records (what the launch must touch once).  The SURVEY 8(d) per-PU byte sum counts every window sample once per PU that
        #     reads it -- thousands of PUs share a window that is staged once -- so it is kept under a name that says it is not traffic.
        #   any other launch: bound "hbm" -- unique bytes of the launch / duration / 8 TB/s.
        VALU_PEAK = N_SIMD * SQ_CLOCK_HZ / 4.0
        common = {"kernel": dom, "traffic": (dk["hbm_MB"] * 1e6 if "hbm_MB" in dk else None), "traffic_source": prof_src if "hbm_MB" in dk else None,
                  "avg_launch_ms": kern_ms[dom], "launches_timed": n_timed.get(dom, 0),
                  "unique_bytes_per_launch": uniq[dstage][dname] if dname in uniq.get(dstage, {}) else abytes,
                  "hbm_frac": dk.get("hbm_frac"), "unique_frac": dk.get("unique_frac"), "valu_busy": dk.get("valu_busy"),
                  "picture_unique_MB": round(uniq_total / 1e6, 1), "picture_unique_frac": round(uniq_total / (ms_pic * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        if dom in useful:
            ach = useful[dom] / (kern_ms[dom] * 1e-3)
            roofline = dict(common, bound="valu", achieved=ach / 1e9, peak=VALU_PEAK / 1e9, unit="G wave-instr/s", frac=ach / VALU_PEAK,
                            per_pu_byte_sum_not_traffic=abytes,
                            stage_body_frac=ach / (N_SIMD / (SAD_STAGE_BODY_NS * 1e-9)), measured_issue_frac=ach / SAD_ISSUE_PEAK,
                            note="bound = vector-instruction issue: achieved = useful v_sad_u16 wave-instructions of the launch (16x16 positions x samples / 2 / 64, every SAD "
                                 "counted ONCE: the 32x32 / 64x64 results are sums) / HIP-event duration of the launch group on its stream; peak = 1024 SIMDs x 2.4 GHz / 4 "
                                 "cycles per instruction; frac = achieved / peak.  measured_issue_frac prices the same count at the 1.75 ns the instruction issues at in "
                                 "a micro-benchmark (profiles/r03_valu_rate.txt), stage_body_frac at the 2.46 ns inside the real stage body (r03_sadloop_rate.txt).  "
                                 "traffic = counter bytes of the committed profile by request size (TCC_EA0_RDREQ 32 / 64 / 128 B + WRITE_SIZE), hbm_frac = traffic / time "
                                 "/ 8 TB/s, unique_frac = union of windows + original + records / time / 8 TB/s.  per_pu_byte_sum_not_traffic = the SURVEY 8(d) "
                                 "formula summed over the PUs the launch answers: every window sample counted once per PU that reads it, NOT bytes moved")
        else:
            ub = common["unique_bytes_per_launch"]
            ach = ub / (kern_ms[dom] * 1e-3) / 1e9
            roofline = dict(common, bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS,
                            note="achieved = bytes the launch must touch once / HIP-event duration of the launch group on its stream; traffic = counter bytes of the "
                                 "committed profile (FETCH by request size + WRITE_SIZE)")
        res = {
            "metric": BASELINE_METRIC + " [M1: hot-path pictures/s of the kernels behind the call sites, NOT EncoderApp fps]",
            "value": pictures / dt,
            "unit": "frames/s",
            "n_gpus": world,
            "rehearsal": bool(args.rehearse),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "timed_s": dt,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int16",
            "data": "synthetic",
            "config": {"workload": "hot-path canonical per-picture workload (SURVEY 8(d) M1: ME SAD searches 16/32/64 +-4 & raster +-96 (one hierarchical launch, every SAD once), fused half/quarter refinement 16x16 (9+9 SATD), "
                                   "bi-pred MC 16x16 (PUs without residual stored straight into the reconstruction), residual+fwd transform+quantiser (Quant::quant, sign hiding)+dequant+inv transform+reco "
                                   "(TUs handed over grouped by shape: vvcgpu_resi_chain_runs_batch), deblock, SAO stats+apply, ALF classify+stats+filter) "
                                   "on %dx%d 10-bit 4:2:0 (BASELINE configs[3] picture format; configs[1] is the same workload at 1920x1080), planes resident in HBM; "
                                   "step = one intra period of %d pictures + hand-over of the last reconstructed picture as the next chunk's reference; "
                                   "NOT EncoderApp fps (RDO control loop out of scope)" % (args.width, args.height, pps),
                       "width": args.width, "height": args.height, "bit_depth": bd, "qp": args.qp, "pictures_per_step": pps, "ms_per_picture": ms_pic,
                       "schedule": ("serial: one HIP stream, stage order (every kernel alone on the device: event-timed durations are kernel times)" if not overlap else
                                    "overlap: reconstruction chain on the main stream, searches / refinement / statistics on three side streams "
                                    "(their real dependencies only); the integer search is the first launch of a picture, alone on the main stream"),
                       "parallelism": "one chunk stream per GPU (intra-period sharding), one point-to-point boundary picture per step; no data-path collective"},
            "roofline": roofline,
            "input_rotation": {"sets": rotate, "MB_per_set": round((sum(int(p_.size) for p_ in wl.org) + sum(int(p_.size) for p_ in wl.ref0_pad)) * 2 / 1e6, 1),
                               "what": "original + first reference picture cycle through this many resident copies (beyond the 256 MB memory-side cache from 5 sets on)"},
            "input_stream": (None if dt_up is None else {
                "value": pictures / dt_up, "unit": "frames/s", "ms_per_picture": dt_up / (args.steps * pps) * 1e3, "timed_s": dt_up,
                "upload_MB_per_picture": round(up["bytes"] / 1e6, 2), "side_info_MB_per_picture": round(up["side_bytes"] / 1e6, 2),
                "upload_GBps": round(up["bytes"] * args.steps * pps / dt_up / 1e9, 2),
                "what": "the same steps with, per picture, one original picture AND the picture's side information (PU / TU descriptor lists, deblocking edge / QP "
                        "maps, SAO parameters, ALF switches) uploaded from pinned host memory on a copy stream inside the timed region (double-buffered through the "
                        "input sets); `value` above has the inputs resident, as the contract asks"}),
            "with_depquant": dq_leg,
            "picture_hashes": {"gathered": len(hashes), "rank0_first_picture_md5": first_md5, "rank0_last_picture_md5": hashes.get("rank0")},
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "serial_kernel_ms_per_picture": round(sum(stage_ms.values()), 4),
            "kernels": per_kernel,
            "kernel_launches": profile_launches(),
            "lib_digest": lib_digest(),
            "encoder_fps_m3": "see profiles/*_m3_encoder.txt (reference encoder with and without the library, measured separately; never this line's value)",
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(wl)
        if world == 1 and not args.no_real_mix:
            # the batch entry points on the call-signature mix of a real encode (tests/golden/trace_*.npz: 80 % of the calls are 4 or 8 wide) next to
            # the same number of samples in 16 x 16 blocks: the canonical workload above is squares only (vvcsoftware_vtm_amd/shape_mix.py)
            from vvcsoftware_vtm_amd import shape_mix
            del wl
            torch.cuda.empty_cache()
            res["kernels_real_mix"] = {"trace": os.path.basename(shape_mix.TRACE), "samples_per_batch": 1 << 21, "square": 16,
                                       "entries": shape_mix.run(1 << 21, reps=3)}
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
