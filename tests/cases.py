"""Seeded input builders shared by the CPU (oracle-vs-reference / golden) and GPU (HIP-vs-oracle) tests."""
import numpy as np

from oraclelib import SAO_DTYPE


def rand_plane(rng, h, w, bd, kind="uniform"):
    mx = (1 << bd) - 1
    if kind == "uniform":
        return rng.integers(0, mx + 1, (h, w)).astype(np.int16)
    if kind == "flat":      # many equal neighbours (sign == 0 paths)
        return ((rng.integers(0, mx + 1, (h, w)) >> (bd - 3)) + (mx // 2)).astype(np.int16)
    if kind == "smooth":
        yy, xx = np.mgrid[0:h, 0:w]
        v = (np.sin(xx / 17.0) + np.cos(yy / 11.0) + 2) / 4 * mx + rng.normal(0, 3 * 2 ** (bd - 8), (h, w))
        return np.clip(np.rint(v), 0, mx).astype(np.int16)
    if kind == "extreme":   # saturating values at both ends (clip paths)
        return rng.choice(np.array([0, 1, mx - 1, mx], dtype=np.int16), (h, w))
    raise ValueError(kind)


def alf_coeffs(rng, amp=60):
    lc = rng.integers(-amp, amp + 1, (25, 13)).astype(np.int16)
    lc[:, 12] = 512 - 2 * lc[:, :12].sum(1)
    cc = rng.integers(-amp, amp + 1, 7).astype(np.int16)
    cc[6] = 512 - 2 * cc[:6].sum()
    return lc, cc


def n_ctus(w, h, cw, ch=None):
    ch = ch or cw
    return ((w + cw - 1) // cw), ((h + ch - 1) // ch)


def sao_params(rng, w, h, cw, ch, full_avail=True, types=None):
    nx, ny = n_ctus(w, h, cw, ch)
    prm = np.zeros(nx * ny, SAO_DTYPE)
    prm["type"] = rng.integers(-1, 5, nx * ny) if types is None else rng.choice(types, nx * ny)
    prm["offset"] = rng.integers(-31, 32, (nx * ny, 32))
    for j in range(ny):
        for i in range(nx):
            L, R, A, B = i > 0, i < nx - 1, j > 0, j < ny - 1
            bits = [L, R, A, B, A and L, A and R, B and L, B and R]
            a = 0
            for k, b in enumerate(bits):
                if b and (full_avail or rng.random() < 0.7):
                    a |= 1 << k
            prm["avail"][j * nx + i] = a
    return prm


def deblock_maps(rng, w, h, mode="cu"):
    """Edge/BS/QP maps per 4x4 luma unit (see include/vvcgpu.h).  mode 'cu': a seeded quadtree CU grid with
    intra/inter blocks and cbf-like BS; mode 'random': arbitrary map bytes (stress: off-grid edges, flags)."""
    w4, h4 = w // 4, h // 4
    ev = np.zeros((h4, w4), np.uint8)
    eh = np.zeros((h4, w4), np.uint8)
    qpl = np.zeros((h4, w4), np.int8)
    qpc = np.zeros((h4, w4), np.int8)
    if mode == "random":
        ev[:] = rng.integers(0, 64, (h4, w4))
        eh[:] = rng.integers(0, 64, (h4, w4))
        # BS value 3 never occurs in the reference
        for m in (ev, eh):
            m[(m & 3) == 3] &= 0xFE
            m[((m >> 2) & 3) == 3] &= 0xFB
        qpl[:] = rng.integers(0, 64, (h4, w4))
        qpc[:] = rng.integers(0, 64, (h4, w4))
        return ev, eh, qpl, qpc
    intra = np.zeros((h4, w4), bool)

    def split(x, y, s):
        if s > 8 and (s > 64 or rng.random() < 0.55):
            for dy in (0, s // 2):
                for dx in (0, s // 2):
                    split(x + dx, y + dy, s // 2)
            return
        x1, y1 = min(x + s, w), min(y + s, h)
        if x >= w or y >= h:
            return
        ux0, uy0, ux1, uy1 = x // 4, y // 4, x1 // 4, y1 // 4
        isint = rng.random() < 0.3
        intra[uy0:uy1, ux0:ux1] = isint
        q = int(rng.integers(22, 46))
        qpl[uy0:uy1, ux0:ux1] = q
        qpc[uy0:uy1, ux0:ux1] = q if rng.random() < 0.7 else int(rng.integers(22, 46))
        nf = rng.random() < 0.03
        for uy in range(uy0, uy1):          # left border
            if x > 0:
                pi = intra[uy, ux0 - 1] or isint
                bs = 2 if pi else int(rng.integers(0, 2))
                ev[uy, ux0] = bs | ((2 if pi else 0) << 2) | (0x20 if nf else 0)
        for ux in range(ux0, ux1):          # top border
            if y > 0:
                pi = intra[uy0 - 1, ux] or isint
                bs = 2 if pi else int(rng.integers(0, 2))
                eh[uy0, ux] = bs | ((2 if pi else 0) << 2) | (0x10 if nf and rng.random() < 0.5 else 0)

    for y in range(0, h, 128):
        for x in range(0, w, 128):
            split(x, y, 128)
    return ev, eh, qpl, qpc


# ---- N2: integer TZ search ------------------------------------------------------------------------------------------
TZ_PU = np.dtype([("org_x", "<i4"), ("org_y", "<i4"), ("ref_x", "<i4"), ("ref_y", "<i4"), ("start_x", "<i4"), ("start_y", "<i4"),
                  ("pred2_x", "<i4"), ("pred2_y", "<i4"), ("pos_x", "<i4"), ("pos_y", "<i4"), ("pred_hor", "<i4"), ("pred_ver", "<i4"),
                  ("w", "<i2"), ("h", "<i2"), ("sub_shift", "<i2"), ("flags", "<i2"), ("reserved", "<i4", (2,))])
TZ_CFG = np.dtype([("lambda", "<f8"), ("cost_scale", "<i4"), ("imv_shift", "<i4"), ("search_range", "<i4"), ("first_search_stop", "<i4"),
                   ("pic_w", "<i4"), ("pic_h", "<i4"), ("max_cu_w", "<i4"), ("max_cu_h", "<i4"),
                   ("ref_x0", "<i4"), ("ref_y0", "<i4"), ("ref_x1", "<i4"), ("ref_y1", "<i4"), ("wg_per_pu", "<i4"), ("reserved", "<i4")])   # "reserved" = vvcgpu_tz_cfg.uniform_pu (the field keeps its name: the golden fixtures store this dtype)
BEST = np.dtype([("x", "<i4"), ("y", "<i4"), ("cost", "<u8"), ("sad", "<u8")])
assert TZ_PU.itemsize == 64 and TZ_CFG.itemsize == 64 and BEST.itemsize == 24


def tz_planes(rng, W, H, M, bd, motion=(7, -5), noise=3):
    """ref: (H+2M) x (W+2M) blob texture (the padded reference picture); org: the picture displaced by `motion` + noise,
    so that searches have a real optimum away from most start vectors."""
    from scipy.ndimage import gaussian_filter
    mx = (1 << bd) - 1
    f = gaussian_filter(rng.normal(0, 1, (H + 2 * M, W + 2 * M)), 2.5) * 6 + gaussian_filter(rng.normal(0, 1, (H + 2 * M, W + 2 * M)), 9) * 30
    ref = np.clip(np.rint((f * 0.12 + 0.5) * mx), 0, mx).astype(np.int16)
    dx, dy = motion
    org = ref[M + dy:M + dy + H, M + dx:M + dx + W].astype(np.int32) + rng.integers(-noise, noise + 1, (H, W)) * (1 << (bd - 8))
    return np.ascontiguousarray(np.clip(org, 0, mx).astype(np.int16)), ref


def tz_pus(rng, n, W, H, M, sizes, flags_choices=(0, 1, 2, 3, 4, 5), spread=40, sub_mode2=None):
    pus = np.zeros(n, TZ_PU)
    for i in range(n):
        w, h = sizes[int(rng.integers(0, len(sizes)))]
        x = int(rng.integers(0, (W - w) // 4 + 1)) * 4
        y = int(rng.integers(0, (H - h) // 4 + 1)) * 4
        r = pus[i]
        r["org_x"], r["org_y"], r["ref_x"], r["ref_y"], r["pos_x"], r["pos_y"] = x, y, M + x, M + y, x, y
        far = rng.random() < 0.3
        s = spread * 4 if far else 24
        r["start_x"], r["start_y"] = int(rng.integers(-s, s + 1)), int(rng.integers(-s, s + 1))
        r["pred2_x"], r["pred2_y"] = int(rng.integers(-12, 13)), int(rng.integers(-12, 13))
        r["pred_hor"], r["pred_ver"] = int(r["start_x"]) + int(rng.integers(-8, 9)), int(r["start_y"]) + int(rng.integers(-8, 9))
        r["w"], r["h"] = w, h
        m2 = bool(rng.integers(0, 2)) if sub_mode2 is None else sub_mode2
        r["sub_shift"] = 1 if (m2 and h > 8 and w <= 64) else 0
        r["flags"] = flags_choices[int(rng.integers(0, len(flags_choices)))]
    return pus


def tz_cfg(W, H, M, lam, search_range=64, first_stop=0, max_cu=128, cost_scale=2, imv_shift=0, wg_per_pu=0):
    c = np.zeros(1, TZ_CFG)
    c[0] = (lam, cost_scale, imv_shift, search_range, first_stop, W, H, max_cu, max_cu, 0, 0, W + 2 * M, H + 2 * M, wg_per_pu, 0)
    return c


# ---- N2: AMVR integer refinement -----------------------------------------------------------------------------------
IMV_PU = np.dtype([("org_x", "<i4"), ("org_y", "<i4"), ("ref_x", "<i4"), ("ref_y", "<i4"), ("mv_x", "<i4"), ("mv_y", "<i4"),
                   ("cand_x", "<i4", (2,)), ("cand_y", "<i4", (2,)), ("pos_x", "<i4"), ("pos_y", "<i4"), ("idx_cost", "<u4", (2,)), ("bits", "<u4"),
                   ("w", "<i2"), ("h", "<i2"), ("num_cand", "i1"), ("mvp_idx", "i1"), ("reserved", "<i2"), ("reserved2", "<i4")])
IMV_RESULT = np.dtype([("mv_x", "<i4"), ("mv_y", "<i4"), ("mvp_idx", "<i4"), ("bits", "<u4"), ("cost", "<u8")])
assert IMV_PU.itemsize == 72 and IMV_RESULT.itemsize == 24


def imv_pus(rng, n, W, H, M, sizes, imv_shift):
    """PUs whose AMVP candidates satisfy what the encoder guarantees on entry (mv - cand is a multiple of 4 quarter units)."""
    pus = np.zeros(n, IMV_PU)
    for i in range(n):
        w, h = sizes[int(rng.integers(0, len(sizes)))]
        x = int(rng.integers(0, (W - w) // 4 + 1)) * 4
        y = int(rng.integers(0, (H - h) // 4 + 1)) * 4
        r = pus[i]
        r["org_x"], r["org_y"], r["ref_x"], r["ref_y"], r["pos_x"], r["pos_y"] = x, y, M + x, M + y, x, y
        r["mv_x"], r["mv_y"] = int(rng.integers(-40, 41)), int(rng.integers(-40, 41))
        step = 1 << imv_shift
        for c in range(2):
            r["cand_x"][c] = (int(rng.integers(-60, 61)) * step) // 1 if rng.random() < 0.5 else int(rng.integers(-50, 51)) * 4
            r["cand_y"][c] = int(rng.integers(-50, 51)) * 4
            r["cand_x"][c] = (int(r["cand_x"][c]) // 4) * 4
        if rng.random() < 0.25:
            r["cand_x"][1], r["cand_y"][1] = r["cand_x"][0], r["cand_y"][0]          # equal candidates: the SATD is reused (:2449-2461)
        r["num_cand"] = 2 if rng.random() < 0.85 else 1
        r["mvp_idx"] = int(rng.integers(0, int(r["num_cand"])))
        r["idx_cost"] = (1, 1) if r["num_cand"] == 2 else (0, 0)
        r["bits"] = int(rng.integers(8, 40))
        r["w"], r["h"] = w, h
    return pus


def unpack_plane(lo, hi):
    """inverse of tests/golden/gen_deblock.py:pack_plane: low / high bytes of the horizontal-then-vertical differences modulo 2^16 -> int16 plane"""
    d = lo.astype(np.int64) | (hi.astype(np.int64) << 8)
    v = np.cumsum(np.cumsum(d, axis=0), axis=1) & 0xFFFF
    return v.astype(np.uint16).view(np.int16)


def deblock_golden():
    """-> list of pictures of tests/golden/deblock.npz (the compiled reference's own loopFilterPic: planes in front, the maps its xDeblockCU walk
    produced, slice / PPS parameters, planes behind): dicts with hdr (field -> int), ev, eh, qp_luma, qp_chroma, pre[3], post[3]"""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "deblock.npz"))
    fields = [str(f) for f in z["hdr_fields"]]
    pics = []
    for i in range(int(z["n"])):
        r = {"hdr": dict(zip(fields, (int(v) for v in z["hdr%d" % i])))}
        for k in ("ev", "eh", "qp_luma", "qp_chroma"):
            r[k] = np.ascontiguousarray(z["%s_%d" % (k, i)])
        r["pre"] = [np.ascontiguousarray(unpack_plane(z["pre%d_lo_%d" % (c, i)], z["pre%d_hi_%d" % (c, i)])) for c in range(3)]
        r["post"] = [(r["pre"][c].astype(np.int32) + z["delta%d_%d" % (c, i)]).astype(np.int16) for c in range(3)]
        pics.append(r)
    return pics
