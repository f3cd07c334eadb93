"""Real-shape benchmark of the batch entry points (VERDICT r2 #4b).

The canonical workload of bench.py is squares only; the reference encoder's calls are not: QTBT / MTT partitioning makes W != H the common case and,
on the committed call trace (tests/golden/trace_*.npz: every block-level call of a 3-picture random-access encode, shapes and parameters only, taken
by the shim's trace mode), four- and eight-wide blocks dominate.  Here every entry point is driven with a batch whose call signatures follow the
trace's histogram (scaled to `samples` samples, shuffled), next to a batch of the same number of samples in 16 x 16 blocks with the same parameter
mix.  `ratio` = real-mix throughput per sample / square throughput per sample: the bar is >= 0.5.

Timing only; parity on the trace's shapes is tests/test_gpu_shape_mix.py (every distinct signature against the oracle)."""
import json
import os

import numpy as np
import torch

from . import ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRACE = os.path.join(ROOT, "tests", "golden", "trace_ragop16_416x240_10b_q32.npz")
ENTRY = {"dist": 0, "interp": 1, "pelop": 2, "tr_fwd": 3, "tr_inv": 4, "dequant_tr_inv": 5, "depquant": 6, "rdoq": 7, "intra_pred": 8}
FILTERS = {8: [-1, 4, -11, 40, 40, -11, 4, -1], 4: [-4, 36, 36, -4], 2: [32, 32]}


def load_trace(path=TRACE):
    z = np.load(path)
    return z["hist"], json.loads(bytes(z["meta"]).decode())


def pow2(v):
    return (v & (v - 1)) == 0


def signatures(hist, entry, keep):
    """rows (w, h, a, b, c, calls) of one entry that satisfy keep(w, h, a, b, c)"""
    h = hist[hist[:, 0] == ENTRY[entry]][:, 1:]
    m = np.array([bool(keep(*r[:5])) for r in h], dtype=bool) if len(h) else np.zeros(0, bool)
    return h[m]


def draw(sig, samples, rng, square=None):
    """a shuffled call list (w, h, a, b, c) whose signature mix follows `sig`, about `samples` samples in total; square = N replaces every shape
    by N x N (same parameters, same number of samples)"""
    tot = float((sig[:, 0] * sig[:, 1] * sig[:, 5]).sum())
    scale = samples / tot
    reps = np.floor(sig[:, 5] * scale + rng.random(len(sig))).astype(np.int64)
    calls = np.repeat(sig[:, :5], reps, axis=0)
    if square:
        cum = np.cumsum(calls[:, 0] * calls[:, 1])
        nsq = int(cum[-1] // (square * square))
        calls = calls[np.searchsorted(cum, (np.arange(nsq) + 0.5) * square * square)].copy()     # one square per square's worth of samples: its
        calls[:, 0] = calls[:, 1] = square                                                      # parameters from the call those samples belong to
    rng.shuffle(calls, axis=0)
    return calls


def gpu_ms(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def _offsets(w, h, pad_w=0, pad_h=0):
    """every block in its own region of a 1-D buffer: row pitch = width (+ padding) rounded up to 8 samples -> (offsets, pitches, total)"""
    pitch = ((w + pad_w + 7) // 8) * 8
    size = pitch * (h + pad_h)
    off = np.concatenate([[0], np.cumsum(size)[:-1]])
    return off.astype(np.int64), pitch.astype(np.int32), int(size.sum())


# ---- one builder per entry point: calls [n, 5] -> (callable, samples) ------------------------------------------------------------------
def build_dist(calls, rng, kind, bd=10):
    w, h = calls[:, 0], calls[:, 1]
    off, pitch, total = _offsets(w, h)
    d = np.zeros(len(calls), ops.DIST_DESC)
    d["org_off"] = d["cur_off"] = off
    d["org_stride"] = d["cur_stride"] = pitch
    d["w"], d["h"] = w, h
    d["sub_shift"] = calls[:, 3] if kind == 0 else 0
    horg, hcur = rng.integers(0, 1 << bd, total, dtype=np.int16), rng.integers(0, 1 << bd, total, dtype=np.int16)
    org, cur = torch.from_numpy(horg).cuda(), torch.from_numpy(hcur).cuda()
    dd = ops.struct_to_device(d)

    def check(orc, P):
        want = np.zeros(len(d), np.uint64)
        orc.orc_dist_batch(kind, P(horg), P(hcur), P(d), len(d), P(want))
        return np.array_equal(ops.dist_batch(kind, org, cur, dd, len(calls), bd).cpu().numpy().view(np.uint64), want)
    return (lambda: ops.dist_batch(kind, org, cur, dd, len(calls), bd)), int((w * h).sum()), check


def build_interp(calls, rng, bd=10):
    w, h, taps, flags = calls[:, 0], calls[:, 1], calls[:, 2], calls[:, 3]
    soff, sp, stotal = _offsets(w, h, 8, 8)
    doff, dp, dtotal = _offsets(w, h)
    d = np.zeros(len(calls), ops.IF_DESC)
    ver, first, last = flags & 1, (flags >> 1) & 1, (flags >> 2) & 1
    before = taps // 2 - 1
    d["src_off"] = soff + np.where(ver == 1, before * sp, before)
    d["dst_off"], d["src_stride"], d["dst_stride"], d["w"], d["h"] = doff, sp, dp, w, h
    d["taps"], d["is_vertical"], d["is_first"], d["is_last"] = taps, ver, first, last
    for t, c in FILTERS.items():
        d["coeff"][taps == t, :t] = c
    # a second-stage call (is_first = 0) reads 14-bit intermediates, a first-stage call reads samples
    hsrc = rng.integers(0, 1 << bd, stotal, dtype=np.int16)
    src = torch.from_numpy(hsrc).cuda()
    dst = torch.zeros(dtotal, dtype=torch.int16, device="cuda")
    dd = ops.struct_to_device(d)

    def check(orc, P):
        want = np.zeros(dtotal, np.int16)
        orc.orc_if_batch(P(hsrc), P(want), P(d), len(d), bd, 0, (1 << bd) - 1)
        dst.zero_()
        ops.if_batch(src, dst, dd, len(calls), bd, (0, (1 << bd) - 1))
        return np.array_equal(dst.cpu().numpy(), want)
    return (lambda: ops.if_batch(src, dst, dd, len(calls), bd, (0, (1 << bd) - 1))), int((w * h).sum()), check


def build_pelop(calls, rng, op, bd=10):
    w, h = calls[:, 0], calls[:, 1]
    off, pitch, total = _offsets(w, h)
    d = np.zeros(len(calls), ops.PELOP_DESC)
    d["src0_off"] = d["src1_off"] = d["dst_off"] = off
    d["src0_stride"] = d["src1_stride"] = d["dst_stride"] = pitch
    d["w"], d["h"] = w, h
    h0, h1 = rng.integers(-2000, 2000, total, dtype=np.int16), rng.integers(-2000, 2000, total, dtype=np.int16)
    s0, s1 = torch.from_numpy(h0).cuda(), torch.from_numpy(h1).cuda()
    dst = torch.zeros(total, dtype=torch.int16, device="cuda")
    cfg = ops.PelopCfg(0, 15 - bd, (1 << (14 - bd)) + 2 * 8192, 1, 0, (1 << bd) - 1) if op == 0 else \
        ops.PelopCfg(0, 0, 0, 1, 0, (1 << bd) - 1) if op == 1 else ops.PelopCfg(3, 2, 1, 1, 0, (1 << bd) - 1)
    dd = ops.struct_to_device(d)

    def check(orc, P):
        import ctypes as C
        want = np.zeros(total, np.int16)
        orc.orc_pelop_batch(op, P(h0), P(h1) if op != 2 else None, P(want), P(d), len(d), C.byref(cfg))
        dst.zero_()
        ops.pelop_batch(op, s0, s1 if op != 2 else None, dst, dd, len(calls), cfg)
        return np.array_equal(dst.cpu().numpy(), want)
    return (lambda: ops.pelop_batch(op, s0, s1 if op != 2 else None, dst, dd, len(calls), cfg)), int((w * h).sum()), check


def _tr_desc(calls):
    w, h = calls[:, 0], calls[:, 1]
    off, pitch, total = _offsets(w, h)
    d = np.zeros(len(calls), ops.TR_DESC)
    d["resi_off"], d["resi_stride"], d["w"], d["h"] = off, pitch, w, h
    d["coeff_off"] = np.concatenate([[0], np.cumsum(w * h)[:-1]])
    d["tr_hor"], d["tr_ver"] = calls[:, 2], calls[:, 3]
    return d, total, int((w * h).sum())


def build_tr(calls, rng, inverse, bd=10):
    d, total, n = _tr_desc(calls)
    hresi, hcoef = rng.integers(-300, 301, total, dtype=np.int16), rng.integers(-2000, 2001, n).astype(np.int32)
    dd = ops.struct_to_device(d)

    def check(orc, P):
        if inverse:
            want, got = np.zeros(total, np.int16), torch.zeros(total, dtype=torch.int16, device="cuda")
            orc.orc_tr_inv_batch(P(hcoef), P(want), P(d), len(d), bd)
            ops.tr_inv_batch(torch.from_numpy(hcoef).cuda(), got, dd, len(calls), bd)
        else:
            want, got = np.zeros(n, np.int32), torch.zeros(n, dtype=torch.int32, device="cuda")
            orc.orc_tr_fwd_batch(P(hresi), P(want), P(d), len(d), bd)
            ops.tr_fwd_batch(torch.from_numpy(hresi).cuda(), got, dd, len(calls), bd)
        return np.array_equal(got.cpu().numpy(), want)
    resi, coef = torch.from_numpy(hresi).cuda(), torch.from_numpy(hcoef).cuda()
    if inverse:
        return (lambda: ops.tr_inv_batch(coef, resi, dd, len(calls), bd)), n, check
    return (lambda: ops.tr_fwd_batch(resi, coef, dd, len(calls), bd)), n, check


def build_chain(calls, rng, bd=10):
    """vvcgpu_resi_chain_batch on the forward-transform call mix (its TUs are the TUs the encoder transforms)"""
    w, h = calls[:, 0], calls[:, 1]
    off, pitch, total = _offsets(w, h)
    d = np.zeros(len(calls), ops.RC_DESC)
    d["org_off"] = d["pred_off"] = d["rec_off"] = off
    d["org_stride"] = d["pred_stride"] = d["rec_stride"] = pitch
    d["level_off"] = np.concatenate([[0], np.cumsum(w * h)[:-1]])
    d["w"], d["h"], d["tr_hor"], d["tr_ver"] = w, h, calls[:, 2], calls[:, 3]
    d["qp"], d["sign_hiding"] = 32 + 12, 1
    n = int((w * h).sum())
    org = torch.from_numpy(rng.integers(0, 1 << bd, total, dtype=np.int16)).cuda()
    pred = torch.clamp(org + torch.from_numpy(rng.integers(-40, 41, total, dtype=np.int16)).cuda(), 0, (1 << bd) - 1).to(torch.int16)
    rec = torch.zeros(total, dtype=torch.int16, device="cuda")
    level = torch.zeros(n, dtype=torch.int32, device="cuda")
    dd = ops.struct_to_device(d)
    return (lambda: ops.resi_chain_batch(org, pred, rec, level, dd, len(calls), bd, (0, (1 << bd) - 1))), n, None


def build_dqtr(calls, rng, bd=10):
    w, h, ts, dep = calls[:, 0], calls[:, 1], calls[:, 3], calls[:, 4]
    off, pitch, total = _offsets(w, h)
    d = np.zeros(len(calls), ops.DQTR_DESC)
    d["resi_off"], d["resi_stride"], d["w"], d["h"] = off, pitch, w, h
    d["level_off"] = np.concatenate([[0], np.cumsum(w * h)[:-1]])
    d["tr_hor"] = d["tr_ver"] = np.where(ts == 1, 3, 0)
    d["dep_quant"], d["qp"] = dep, 32 + 12
    n = int((w * h).sum())
    hlv = (rng.integers(-12, 13, n) * (rng.random(n) < 0.35)).astype(np.int32)
    lv = torch.from_numpy(hlv).cuda()
    resi = torch.zeros(total, dtype=torch.int16, device="cuda")
    dd = ops.struct_to_device(d)

    def check(orc, P):
        want, wcoef = np.zeros(total, np.int16), np.zeros(n, np.int32)
        orc.orc_dequant_tr_inv_batch(P(hlv), P(want), P(d), len(d), bd, P(wcoef))
        resi.zero_()
        ops.dequant_tr_inv_batch(lv, resi, dd, len(calls), bd, None)
        return np.array_equal(resi.cpu().numpy(), want)
    return (lambda: ops.dequant_tr_inv_batch(lv, resi, dd, len(calls), bd, None)), n, check


def build_depquant(calls, rng, bd=10):
    w, h = calls[:, 0], calls[:, 1]
    g = np.load(os.path.join(ROOT, "tests", "golden", "depquant.npz"))
    rates = np.ascontiguousarray(g["rates"][:4]).view(ops.DQ_RATES)
    n = int((w * h).sum())
    d = np.zeros(len(calls), ops.DEPQUANT_DESC)
    d["coeff_off"] = d["level_off"] = np.concatenate([[0], np.cumsum(w * h)[:-1]])
    d["lambda"], d["qp"], d["rates_idx"], d["w"], d["h"] = 60.0, 44, rng.integers(0, 4, len(calls)), w, h
    d["luma"] = (calls[:, 2] == 0)
    # coefficient magnitudes falling off with the frequency, as after a transform
    coef = np.zeros(n, np.int32)
    pos = 0
    for ww, hh in zip(w, h):
        yy, xx = np.mgrid[0:hh, 0:ww]
        coef[pos:pos + ww * hh] = (rng.normal(0, 1500, (hh, ww)) * np.exp(-(xx / ww * 3 + yy / hh * 3))).astype(np.int32).reshape(-1)
        pos += ww * hh
    dc, dd, dr = torch.from_numpy(coef).cuda(), ops.struct_to_device(d), ops.struct_to_device(rates)
    level = torch.zeros(n, dtype=torch.int32, device="cuda")
    return (lambda: ops.depquant_batch(dc, level, dd, len(calls), dr, n, bd)), n, None


def rows(hist):
    """(name, entry in the trace, filter on a signature, builder) of every measured entry point.  The filters are the shim's own eligibility tests
    (vtm_hip_shim.cpp): what it would hand to the library."""
    dist_ok = lambda w, h, a, b, c: w >= 4 and h >= 4 and w % 2 == 0 and w <= 128 and h <= 128 and not (a == 0 and w == 4 and b) and \
        not (a == 0 and b and h % (1 << b))
    tu_ok = lambda w, h, a, b, c: 2 <= w <= 64 and 2 <= h <= 64 and pow2(w) and pow2(h)
    return [
        ("dist_batch SAD", "dist", lambda w, h, a, b, c: a == 0 and dist_ok(w, h, a, b, c), lambda c, r: build_dist(c, r, 0)),
        ("dist_batch Hadamard", "dist", lambda w, h, a, b, c: a == 1 and dist_ok(w, h, a, b, c) and w % 4 == 0 and h % 4 == 0, lambda c, r: build_dist(c, r, 1)),
        ("dist_batch SSE", "dist", lambda w, h, a, b, c: a == 2 and dist_ok(w, h, a, b, c), lambda c, r: build_dist(c, r, 2)),
        ("if_batch", "interp", lambda w, h, a, b, c: 2 <= w <= 256 and h <= 256, build_interp),
        ("pelop_batch addAvg", "pelop", lambda w, h, a, b, c: a == 0 and 8 <= w <= 128 and h <= 128, lambda c, r: build_pelop(c, r, 0)),
        ("pelop_batch reco", "pelop", lambda w, h, a, b, c: a == 1 and 8 <= w <= 128 and h <= 128, lambda c, r: build_pelop(c, r, 1)),
        ("tr_fwd_batch", "tr_fwd", tu_ok, lambda c, r: build_tr(c, r, False)),
        ("tr_inv_batch", "tr_inv", tu_ok, lambda c, r: build_tr(c, r, True)),
        ("resi_chain_batch", "tr_fwd", tu_ok, build_chain),
        ("dequant_tr_inv_batch", "dequant_tr_inv", tu_ok, build_dqtr),
        ("depquant_batch", "depquant", lambda w, h, a, b, c: tu_ok(w, h, a, b, c) and w >= 4 and h >= 4, build_depquant),
    ]


def run(samples=1 << 21, seed=3, square=16, reps=5, only=None):
    """-> {entry point: {real_ms, square_ms, samples, real_Gs, square_Gs, ratio, calls, narrow_share}}"""
    hist, _ = load_trace()
    out = {}
    for name, entry, keep, build in rows(hist):
        if only and name not in only:
            continue
        sig = signatures(hist, entry, keep)
        if not len(sig):
            continue
        rng = np.random.default_rng(seed)
        res = {}
        for form, sq in (("real", None), ("square", square)):
            calls = draw(sig, samples, rng, sq)
            fn, n, _ = build(calls, rng)
            ms = gpu_ms(fn, reps)
            res[form] = (ms, n, len(calls))
            del fn
            torch.cuda.empty_cache()
        share = float((sig[(sig[:, 0] <= 8), 5]).sum()) / float(sig[:, 5].sum())
        r = {"calls": res["real"][2], "samples": res["real"][1], "real_ms": round(res["real"][0], 4), "square_ms": round(res["square"][0], 4),
             "real_Gsamples_s": round(res["real"][1] / res["real"][0] / 1e6, 2), "square_Gsamples_s": round(res["square"][1] / res["square"][0] / 1e6, 2),
             "calls_8_wide_or_less": round(share, 3)}
        r["ratio"] = round(r["real_Gsamples_s"] / r["square_Gsamples_s"], 3)
        out[name] = r
    return out


def parity(oracle, P, per_entry=600, seed=11):
    """every distinct call signature of the trace (up to per_entry per entry point, the most frequent first), one call each in one batch, against the
    oracle (tests only).  -> {entry point: (signatures checked, identical)}"""
    hist, _ = load_trace()
    out = {}
    for name, entry, keep, build in rows(hist):
        sig = signatures(hist, entry, keep)
        if not len(sig):
            continue
        sig = sig[np.argsort(-sig[:, 5], kind="stable")][:per_entry]
        rng = np.random.default_rng(seed)
        calls = sig[:, :5].copy()
        rng.shuffle(calls, axis=0)
        _, _, check = build(calls, rng)
        if check is None:
            continue
        out[name] = (len(calls), bool(check(oracle, P)))
    return out
