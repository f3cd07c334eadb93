"""Canonical per-picture hot-path workload (measurement M1).

Everything is derived from a seed: three synthetic frames (current = org, two references), block/PU/TU
descriptors, deblocking maps, SAO and ALF parameters.  The same `Workload` object drives

  * the HIP path (`run_gpu`, through vvcsoftware_vtm_amd.ops -> C ABI),
  * the checkers (tests/workload_cpu.py `run_cpu`: oracle/liboracle.so or oracle/_ref/libvtmref.so over the same object) --
    test infrastructure, not part of this package.

Stages (one picture = one step):
  me     integer ME: SAD surface of every 16x16 / 32x32 / 64x64 block at the 81 positions of a +-4 full search and
         at the 5-stride raster of a +-96 window (39x39), sub_shift 1 (FEN mode 2), with the MV-cost argmin
  frac   fused fractional refinement (half + quarter sample, 9 + 9 Hadamard candidates, MV cost) of every 16x16 block
         around a seeded integer MV (the device form of xPatternSearchFracDIF; fractional planes stay in LDS)
  mc     bi-predictive MC of the whole picture as 16x16 PUs (luma 8-tap + chroma 4-tap) + addAvg
  resi   residual = org - pred; forward + inverse transforms over a seeded tiling {64,32,16,8,4} in equal pixel
         shares (DST-VII/DCT-VIII pairs on tiles <= 32), scalar quantisation with sign bit hiding (Quant::quant, no RDOQ) at
         QP 32, de-quantisation (Quant::dequant); reconstruction -- one pass per TU (vvcgpu_resi_chain_batch; `fused_resi=False`
         runs the five separate entry points instead: same outputs)
  dbk    deblocking with a seeded CU grid / BS / QP field
  sao    SAO statistics + apply with seeded per-CTU parameters (all five types)
  alf    ALF classification + covariance statistics (7x7 and 5x5 luma, 5x5 chroma) + 7x7 luma / 5x5 chroma filtering
"""
import ctypes as C

import numpy as np

from . import synth

MARGIN = 144          # reference picture margin (maxCUWidth + 16, Picture.cpp:737-742)
CTU = 128
STAGES = ["me", "frac", "mc", "resi", "dbk", "sao", "alf"]

DIST_DESC = np.dtype([("org_off", "<i8"), ("cur_off", "<i8"), ("org_stride", "<i4"), ("cur_stride", "<i4"),
                      ("w", "<i2"), ("h", "<i2"), ("sub_shift", "<i2"), ("reserved", "<i2")])
SEARCH_BLK = np.dtype([("org_x", "<i4"), ("org_y", "<i4"), ("ref_x", "<i4"), ("ref_y", "<i4")])
SEARCH_BEST = np.dtype([("x", "<i4"), ("y", "<i4"), ("cost", "<u8"), ("sad", "<u8")])
MC_DESC = np.dtype([("ref0_off", "<i8"), ("ref1_off", "<i8"), ("dst_off", "<i8"), ("ref0_stride", "<i4"),
                    ("ref1_stride", "<i4"), ("dst_stride", "<i4"), ("w", "<i2"), ("h", "<i2"), ("frac_x0", "i1"),
                    ("frac_y0", "i1"), ("frac_x1", "i1"), ("frac_y1", "i1"), ("is_luma", "i1"), ("bi", "i1"),
                    ("reserved", "<i2")])
PELOP_DESC = np.dtype([("src0_off", "<i8"), ("src1_off", "<i8"), ("dst_off", "<i8"), ("src0_stride", "<i4"),
                       ("src1_stride", "<i4"), ("dst_stride", "<i4"), ("w", "<i2"), ("h", "<i2")])
TR_DESC = np.dtype([("resi_off", "<i8"), ("coeff_off", "<i8"), ("resi_stride", "<i4"), ("w", "<i2"), ("h", "<i2"),
                    ("tr_hor", "i1"), ("tr_ver", "i1"), ("reserved", "<i2"), ("reserved2", "<i4")])
SAO_DTYPE = np.dtype([("type", "i1"), ("avail", "u1"), ("offset", "<i2", (32,))])
FRAC_BLK = np.dtype([("org_x", "<i4"), ("org_y", "<i4"), ("ref_x", "<i4"), ("ref_y", "<i4"), ("mv_x", "<i4"), ("mv_y", "<i4")])
FRAC_RESULT = np.dtype([("half_x", "<i4"), ("half_y", "<i4"), ("qter_x", "<i4"), ("qter_y", "<i4"), ("cost_half", "<u8"), ("cost", "<u8")])


QUANT_DESC = np.dtype([("coeff_off", "<i8"), ("level_off", "<i8"), ("w", "<i2"), ("h", "<i2"), ("intra_slice", "i1"), ("sign_hiding", "i1"),
                       ("reserved", "<i2"), ("qp", "<i4"), ("reserved2", "<i4")])
DQTR_DESC = np.dtype([("resi_off", "<i8"), ("level_off", "<i8"), ("resi_stride", "<i4"), ("w", "<i2"), ("h", "<i2"),
                      ("tr_hor", "i1"), ("tr_ver", "i1"), ("dep_quant", "i1"), ("reserved", "i1"), ("qp", "<i4")])
RC_DESC = np.dtype([("org_off", "<i8"), ("pred_off", "<i8"), ("rec_off", "<i8"), ("level_off", "<i8"), ("org_stride", "<i4"), ("pred_stride", "<i4"),
                    ("rec_stride", "<i4"), ("w", "<i2"), ("h", "<i2"), ("tr_hor", "i1"), ("tr_ver", "i1"), ("intra_slice", "i1"), ("sign_hiding", "i1"),
                    ("qp", "<i4"), ("reserved", "<i4", (2,))])
DQ_RATES = np.dtype([("last_x", "<i4", (64,)), ("last_y", "<i4", (64,)), ("sig_sbb", "<i4", (2, 2)), ("sig", "<i4", (3, 18, 2)), ("gtx", "<i4", (21, 7))])
DEPQUANT_DESC = np.dtype([("coeff_off", "<i8"), ("level_off", "<i8"), ("lambda", "<f8"), ("qp", "<i4"), ("rates_idx", "<i4"), ("w", "<i2"), ("h", "<i2"),
                          ("luma", "i1"), ("reserved", "i1", (3,))])
assert QUANT_DESC.itemsize == 32 and DQTR_DESC.itemsize == 32 and RC_DESC.itemsize == 64 and DEPQUANT_DESC.itemsize == 40


class MvCost(C.Structure):
    _fields_ = [("lambda_", C.c_double), ("pred_hor", C.c_int32), ("pred_ver", C.c_int32),
                ("cost_scale", C.c_int32), ("imv_shift", C.c_int32)]


class PelopCfg(C.Structure):
    _fields_ = [("scale", C.c_int32), ("shift", C.c_int32), ("offset", C.c_int32), ("clip", C.c_int32),
                ("clp_min", C.c_int32), ("clp_max", C.c_int32)]


class DeblockCfg(C.Structure):
    _fields_ = [("bit_depth_luma", C.c_int32), ("bit_depth_chroma", C.c_int32),
                ("beta_offset_div2", C.c_int32), ("tc_offset_div2", C.c_int32),
                ("cb_qp_offset", C.c_int32), ("cr_qp_offset", C.c_int32),
                ("clp_min", C.c_int32 * 3), ("clp_max", C.c_int32 * 3)]


def _plane_offsets(sizes):
    """start (in samples) of each plane when the planes of one picture share an allocation: starts rounded up to 64 samples;
    the last entry is the total"""
    off = [0]
    for n in sizes:
        off.append((off[-1] + n + 63) & ~63)
    return off


def _pad(plane, m):
    return np.ascontiguousarray(np.pad(plane, m, mode="edge"))


class Workload:
    def __init__(self, width, height, bit_depth=10, seed=20261003, raster_range=96, me_sizes=(16, 32, 64), qp=32, fused_resi=True, hier_me=True, depquant=False, fuse_alf=True):
        assert width % 8 == 0 and height % 8 == 0
        self.w, self.h, self.bd = width, height, bit_depth
        self.mx = (1 << bit_depth) - 1
        self.seed = seed
        self.raster_range = raster_range
        # depquant: the quantiser of the shipped configurations (cfg/encoder_randomaccess_vtm.cfg: DepQuant 1) instead of the Quant::quant stand-in --
        # the dependent-quantisation trellis (vvcgpu_depquant_batch) between the separate transform entry points, rate tables from a fixed seed
        # (bench.py's `with_depquant` leg; the headline workload keeps the stand-in SURVEY 8(d) allows)
        self.depquant = bool(depquant)
        self.fused_resi = fused_resi and not self.depquant   # residual chain as ONE pass (vvcgpu_resi_chain_batch) or as its five separate entry points
        # integer ME as ONE hierarchical launch (vvcgpu_me_hier_search: every 16x16 SAD once, 32x32 / 64x64 by addition, raster and +-4 grid from
        # the same LDS window) or as six per-size searches (vvcgpu_sad_search): same results
        # (the entry's own preconditions, csrc/mehier.hip: at most 39 raster columns, and the raster's window must contain the +-4 grid's: 5 (R // 5) >= 4 + 15)
        self.fuse_alf = bool(fuse_alf)        # classifier inside the covariance launch (vvcgpu_alf_classify_stats_picture) in the serial schedule
        self.hier_me = bool(hier_me) and tuple(sorted(me_sizes)) == (16, 32, 64) and raster_range <= 99 and 5 * (raster_range // 5) >= 19
        self.qp = qp                                   # base QP (BASELINE configs: 22 / 27 / 32 / 37); quantiser, de-quantiser and the deblocking QP field follow it
        rng = np.random.default_rng(seed)
        frames = synth.gen_yuv(width, height, 3, bit_depth, seed)
        to16 = lambda fr: [p.astype(np.int16) for p in fr]
        self.ref1, self.ref0, self.org = to16(frames[0]), to16(frames[1]), to16(frames[2])
        m, mc = MARGIN, MARGIN // 2
        self.ref0_pad = [_pad(self.ref0[0], m), _pad(self.ref0[1], mc), _pad(self.ref0[2], mc)]
        self.ref1_pad = [_pad(self.ref1[0], m), _pad(self.ref1[1], mc), _pad(self.ref1[2], mc)]
        self.pw, self.pwc = width + 2 * m, width // 2 + 2 * mc

        # ---- ME blocks ---------------------------------------------------------------------------------------
        self.me = {}
        for s in me_sizes:
            xs, ys = np.arange(0, width - s + 1, s), np.arange(0, height - s + 1, s)
            gx, gy = np.meshgrid(xs, ys)
            blk = np.zeros(gx.size, SEARCH_BLK)
            blk["org_x"], blk["org_y"] = gx.reshape(-1), gy.reshape(-1)
            blk["ref_x"], blk["ref_y"] = blk["org_x"] + m, blk["org_y"] + m
            self.me[s] = blk
        nr = 2 * (raster_range // 5) + 1
        self.me_grids = [(-4, -4, 9, 9, 1, 1), (-5 * (nr // 2), -5 * (nr // 2), nr, nr, 5, 5)]
        lam = float(np.sqrt(57.0 * 2.0 ** ((self.qp - 32) / 3.0)))     # sqrt(lambda) of the motion cost, doubling every 3 QP steps
        self.mvcost = MvCost(lam, 0, 0, 2, 0)

        # ---- fractional refinement: every full 16x16 block around a seeded integer MV --------------------------
        b16 = self.me[16] if 16 in self.me else None
        if b16 is None:
            xs, ys = np.arange(0, width - 15, 16), np.arange(0, height - 15, 16)
            gx, gy = np.meshgrid(xs, ys)
            b16 = np.zeros(gx.size, SEARCH_BLK)
            b16["org_x"], b16["org_y"] = gx.reshape(-1), gy.reshape(-1)
        fb = np.zeros(b16.size, FRAC_BLK)
        imv = rng.integers(-8, 9, (b16.size, 2))
        fb["org_x"], fb["org_y"] = b16["org_x"], b16["org_y"]
        fb["mv_x"], fb["mv_y"] = imv[:, 0], imv[:, 1]
        fb["ref_x"], fb["ref_y"] = b16["org_x"] + m + imv[:, 0], b16["org_y"] + m + imv[:, 1]
        self.frac = fb
        self.frac_mvcost = MvCost(lam, 3, -2, 0, 0)

        # ---- MC: bi-pred 16x16 PUs, quarter-pel MVs ----------------------------------------------------------
        xs, ys = np.arange(0, width, 16), np.arange(0, height, 16)
        gx, gy = np.meshgrid(xs, ys)
        gx, gy = gx.reshape(-1), gy.reshape(-1)
        n = gx.size
        bw = np.minimum(16, width - gx).astype(np.int16)
        bh = np.minimum(16, height - gy).astype(np.int16)
        mv = rng.integers(-32, 33, (n, 4))                       # quarter-pel luma units: (x0, y0, x1, y1)
        dl = np.zeros(n, MC_DESC)
        dl["ref0_off"] = (gy + m + (mv[:, 1] >> 2)) * self.pw + gx + m + (mv[:, 0] >> 2)
        dl["ref1_off"] = (gy + m + (mv[:, 3] >> 2)) * self.pw + gx + m + (mv[:, 2] >> 2)
        dl["dst_off"] = gy * width + gx
        dl["ref0_stride"] = dl["ref1_stride"] = self.pw
        dl["dst_stride"] = width
        dl["w"], dl["h"] = bw, bh
        dl["frac_x0"], dl["frac_y0"] = (mv[:, 0] & 3) << 2, (mv[:, 1] & 3) << 2
        dl["frac_x1"], dl["frac_y1"] = (mv[:, 2] & 3) << 2, (mv[:, 3] & 3) << 2
        dl["is_luma"], dl["bi"] = 1, 1
        dc = np.zeros(n, MC_DESC)
        cx, cy = gx // 2, gy // 2
        dc["ref0_off"] = (cy + mc + (mv[:, 1] >> 3)) * self.pwc + cx + mc + (mv[:, 0] >> 3)
        dc["ref1_off"] = (cy + mc + (mv[:, 3] >> 3)) * self.pwc + cx + mc + (mv[:, 2] >> 3)
        dc["dst_off"] = cy * (width // 2) + cx
        dc["ref0_stride"] = dc["ref1_stride"] = self.pwc
        dc["dst_stride"] = width // 2
        dc["w"], dc["h"] = bw // 2, bh // 2
        dc["frac_x0"], dc["frac_y0"] = (mv[:, 0] & 7) << 2, (mv[:, 1] & 7) << 2
        dc["frac_x1"], dc["frac_y1"] = (mv[:, 2] & 7) << 2, (mv[:, 3] & 7) << 2
        dc["is_luma"], dc["bi"] = 0, 1
        self.mc_luma, self.mc_chroma = dl, dc
        # picture list: the three components in one descriptor list over planes that share one allocation each (run_gpu lays them
        # out back to back, plane starts rounded to 64 samples), so one call - one fast launch - predicts the whole picture
        self.ref_plane_off = _plane_offsets([(height + 2 * m) * self.pw] + 2 * [(height // 2 + 2 * mc) * self.pwc])
        self.pic_plane_off = _plane_offsets([height * width] + 2 * [(height // 2) * (width // 2)])
        parts = []
        for c, src in enumerate((dl, dc, dc)):
            d = src.copy()
            d["ref0_off"] += self.ref_plane_off[c]
            d["ref1_off"] += self.ref_plane_off[c]
            d["dst_off"] += self.pic_plane_off[c]
            parts.append(d)
        # row of PUs by row of PUs, the three components of a row together: every part of the list costs the same, which the launch's
        # XCD-contiguous split of the list relies on (luma first and chroma after it left the chroma XCDs idle: 0.094 ms against 0.076)
        per_row = xs.size
        self.mc_pic = np.concatenate([p[r * per_row:(r + 1) * per_row] for r in range(ys.size) for p in parts])
        # PUs without a residual in this workload (chroma; luma below / right of the TU tiling): their reconstruction IS the clipped prediction
        # (xReconInter with cbf == 0: copyClip).  In the one-pass form the motion compensation stores those PUs straight into the reconstruction
        # picture -- the copy-clip launch of rounds 1-5 (8.4 us per 4K picture) is gone; run_gpu patches dst_off of these entries
        tiled_w, tiled_h = width - width % 64, height - height % 64
        rest_luma = (gy >= tiled_h) | (gx >= tiled_w)
        masks = [rest_luma, np.ones(n, bool), np.ones(n, bool)]
        self.mc_rec_mask = np.concatenate([m[r * per_row:(r + 1) * per_row] for r in range(ys.size) for m in masks])
        self._mc_pic_patched = None

        # ---- residual / transform tiling (luma) --------------------------------------------------------------
        rows = []
        coff = 0
        sizes = [64, 32, 16, 8, 4]
        ci = 0
        for y0 in range(0, height - height % 64, 64):
            for x0 in range(0, width - width % 64, 64):
                s = sizes[ci % 5]
                ci += 1
                for ty in range(0, 64, s):
                    for tx in range(0, 64, s):
                        if s <= 32 and rng.random() < 0.5:
                            th, tv = [(1, 1), (1, 2), (2, 1), (2, 2)][int(rng.integers(0, 4))]
                        else:
                            th, tv = 0, 0
                        rows.append(((y0 + ty) * width + x0 + tx, coff, width, s, s, th, tv, 0, 0))
                        coff += s * s
        self.tr = np.array(rows, dtype=TR_DESC)
        # the TU list an encoder hands over is grouped by shape (largest first; it knows the shapes when it builds the list): the one-pass chain then runs
        # without a classification launch (vvcgpu_resi_chain_runs_batch).  Stable, so inside a shape the TUs keep their picture order.
        order = np.argsort(-self.tr["w"].astype(np.int64), kind="stable")
        self.tr = np.ascontiguousarray(self.tr[order])
        self.tu_runs = [(int(s_), int(s_), int((self.tr["w"] == s_).sum())) for s_ in sizes if (self.tr["w"] == s_).any()]
        self.n_coef = coff
        # quantiser between the transforms: Quant::quant without RDOQ (P slice, sign bit hiding) and Quant::dequant at the base QP
        qp = self.qp + 6 * (bit_depth - 8)
        self.quant = np.zeros(self.tr.size, QUANT_DESC)
        self.quant["coeff_off"] = self.quant["level_off"] = self.tr["coeff_off"]
        self.quant["w"], self.quant["h"], self.quant["sign_hiding"], self.quant["qp"] = self.tr["w"], self.tr["h"], 1, qp
        self.dqtr = np.zeros(self.tr.size, DQTR_DESC)
        self.dqtr["resi_off"], self.dqtr["level_off"], self.dqtr["resi_stride"] = self.tr["resi_off"], self.tr["coeff_off"], self.tr["resi_stride"]
        self.dqtr["w"], self.dqtr["h"], self.dqtr["tr_hor"], self.dqtr["tr_ver"], self.dqtr["qp"] = self.tr["w"], self.tr["h"], self.tr["tr_hor"], self.tr["tr_ver"], qp
        if self.depquant:
            # DepQuant::quant per TU: eight rate tables of plausible magnitudes (fractional bits, 1 bit = 2^15) from their own seed -- an encoder would
            # fill them from its CABAC states per TU (DepQuant.cpp:1323-1409); lambda follows the QP as the reference's RD lambda does
            rq = np.random.default_rng(seed ^ 0x5EED)
            K = 8
            self.dq_rates = np.zeros(K, DQ_RATES)
            for f in ("last_x", "last_y", "sig_sbb", "sig", "gtx"):
                self.dq_rates[f] = rq.integers(6000, 140000, self.dq_rates[f].shape)
            self.dq = np.zeros(self.tr.size, DEPQUANT_DESC)
            self.dq["coeff_off"] = self.dq["level_off"] = self.tr["coeff_off"]
            self.dq["lambda"] = 0.57 * 2.0 ** ((self.qp - 12) / 3.0)
            self.dq["qp"], self.dq["rates_idx"] = qp, rq.integers(0, K, self.tr.size)
            self.dq["w"], self.dq["h"], self.dq["luma"] = self.tr["w"], self.tr["h"], 1
            self.dqtr["dep_quant"] = 1
        # the same TUs for the fused chain (residual -> T1 -> quant -> dequant -> T2 -> reconstruction in one pass)
        self.rc = np.zeros(self.tr.size, RC_DESC)
        self.rc["org_off"] = self.rc["pred_off"] = self.rc["rec_off"] = self.tr["resi_off"]
        self.rc["level_off"] = self.tr["coeff_off"]
        self.rc["org_stride"] = self.rc["pred_stride"] = self.rc["rec_stride"] = width
        self.rc["w"], self.rc["h"], self.rc["tr_hor"], self.rc["tr_ver"] = self.tr["w"], self.tr["h"], self.tr["tr_hor"], self.tr["tr_ver"]
        self.rc["sign_hiding"], self.rc["qp"] = 1, qp
        self.tiled_w, self.tiled_h = width - width % 64, height - height % 64      # the TU tiling covers whole 64x64 cells only
        # plane-wide element-wise ops as one descriptor per CTU (the reference calls them per CU, <= 128x128)
        def bands(wp, hp):
            r = []
            for y0 in range(0, hp, 128):
                for x0 in range(0, wp, 128):
                    r.append((y0 * wp + x0, y0 * wp + x0, y0 * wp + x0, wp, wp, wp, min(128, wp - x0), min(128, hp - y0)))
            return np.array(r, dtype=PELOP_DESC)
        self.bands_luma = bands(width, height)
        self.bands_chroma = bands(width // 2, height // 2)
        # what the fused chain leaves to copy (reconstruction = clipped prediction): both chroma planes and the luma samples outside
        # the TU tiling, as ONE list relative to the luma plane of a shared allocation
        rest = []
        for c in (1, 2):
            b = self.bands_chroma.copy()
            for f in ("src0_off", "src1_off", "dst_off"):
                b[f] += self.pic_plane_off[c]
            rest.append(b)
        strips = [(y0, x0, min(128, width - x0), min(128, height - y0)) for y0 in range(self.tiled_h, height, 128) for x0 in range(0, width, 128)]
        strips += [(y0, x0, min(128, width - x0), min(128, self.tiled_h - y0)) for y0 in range(0, self.tiled_h, 128)
                   for x0 in range(self.tiled_w, width, 128)]
        rest.append(np.array([(y0 * width + x0,) * 3 + (width,) * 3 + (bw_, bh_) for (y0, x0, bw_, bh_) in strips], dtype=PELOP_DESC).reshape(-1))
        self.bands_rest = np.concatenate(rest)
        self.cfg_sub = PelopCfg(0, 0, 0, 0, 0, self.mx)
        self.cfg_reco = PelopCfg(0, 0, 0, 1, 0, self.mx)

        # ---- deblocking maps: seeded CU grid -----------------------------------------------------------------
        w4, h4 = width // 4, height // 4
        cell = rng.choice(np.array([8, 16, 32, 64]), ((height + 63) // 64, (width + 63) // 64))
        cu = np.repeat(np.repeat(cell, 16, axis=0), 16, axis=1)[:h4, :w4]          # CU size per 4x4 unit
        ux, uy = np.meshgrid(np.arange(w4) * 4, np.arange(h4) * 4)
        intra = np.repeat(np.repeat(rng.random(((height + 7) // 8, (width + 7) // 8)) < 0.3, 2, axis=0), 2, axis=1)[:h4, :w4]
        bs_rand = rng.integers(0, 2, (h4, w4))
        def edge_map(pos, shift_axis):
            on = (pos % cu == 0) & (pos > 0)
            nb = np.roll(intra, 1, axis=shift_axis)
            either = intra | nb
            bs = np.where(either, 2, bs_rand)
            return np.where(on, bs | (np.where(either, 2, 0) << 2), 0).astype(np.uint8)
        self.edge_ver, self.edge_hor = edge_map(ux, 1), edge_map(uy, 0)
        self.qp_luma = np.repeat(np.repeat(rng.integers(self.qp - 6, self.qp + 8, ((height + 7) // 8, (width + 7) // 8)), 2, axis=0), 2, axis=1)[:h4, :w4].astype(np.int8)
        self.qp_chroma = self.qp_luma.copy()
        self.dbk_cfg = DeblockCfg(bit_depth, bit_depth, 0, 0, 0, 0, (C.c_int32 * 3)(0, 0, 0), (C.c_int32 * 3)(self.mx, self.mx, self.mx))

        # ---- SAO / ALF parameters ----------------------------------------------------------------------------
        self.nctu_x, self.nctu_y = (width + CTU - 1) // CTU, (height + CTU - 1) // CTU
        nctu = self.nctu_x * self.nctu_y
        self.sao = []
        for comp in range(3):
            prm = np.zeros(nctu, SAO_DTYPE)
            prm["type"] = rng.integers(-1, 5, nctu)
            prm["offset"] = rng.integers(-7, 8, (nctu, 32))
            ix, iy = np.meshgrid(np.arange(self.nctu_x), np.arange(self.nctu_y))
            L, R, A, B = ix > 0, ix < self.nctu_x - 1, iy > 0, iy < self.nctu_y - 1
            av = (L * 1) | (R * 2) | (A * 4) | (B * 8) | ((A & L) * 16) | ((A & R) * 32) | ((B & L) * 64) | ((B & R) * 128)
            prm["avail"] = av.reshape(-1).astype(np.uint8)
            self.sao.append(prm)
        lc = rng.integers(-40, 41, (25, 13)).astype(np.int16)
        lc[:, 12] = 512 - 2 * lc[:, :12].sum(1)
        cc = rng.integers(-40, 41, 7).astype(np.int16)
        cc[6] = 512 - 2 * cc[:6].sum()
        self.alf_luma_coeff, self.alf_chroma_coeff = lc, cc
        self.alf_enable = [(rng.random(nctu) < 0.8).astype(np.uint8) for _ in range(3)]

    # ------------------------------------------------------------------------------------------------------
    def algorithmic_bytes(self):
        """Algorithmic (compulsory) bytes per stage and per launch group, SURVEY.md §8(d) formulas."""
        w, h = self.w, self.h
        P = w * h * 3              # picture bytes (4:2:0, 16-bit samples)
        Y = w * h * 2
        out = {}
        me = {}
        for s, blk in self.me.items():
            hs = s // 2
            for (dx0, dy0, nx, ny, sx, sy) in self.me_grids:
                Ww, Wh = (nx - 1) * sx + s, (ny - 1) * sy + s
                # every search returns the best candidate only (24 B per block): xPatternSearch / xTZSearch keep nothing else
                outb = 24
                me["sad_search_%dx%d_%dx%d" % (s, s, nx, ny)] = blk.size * (Ww * Wh * 2 + s * s * 2 + outb)
        if self.hier_me:                               # one launch serves all six searches: the SURVEY 8(d) figure of every PU it answers
            me = {"hier_search": sum(me.values())}
        out["me"] = me
        out["frac"] = {"frac_refine_16x16": self.frac.size * (24 * 24 * 2 + 16 * 16 * 2 + 32)}
        nl = self.mc_luma.size
        # per PU: two reference windows (W+7)^2 (luma) / (W/2+3)^2 (chroma, x2 components) + the written block
        out["mc"] = {"mc_picture": nl * (2 * 23 * 23 * 2 + 16 * 16 * 2) + 2 * nl * (2 * 11 * 11 * 2 + 8 * 8 * 2)}
        ncoef = self.n_coef
        out["resi"] = {"subtract": 3 * Y, "tr_fwd": ncoef * 6, "quant": ncoef * 8, "depquant": ncoef * 8, "dequant_tr_inv": ncoef * 6, "reco": 3 * Y,
                       # fused: org + pred in, levels + reconstruction out per covered sample; the rest of the plane is copied
                       "resi_chain": ncoef * (2 + 2 + 4 + 2)}
        maps = (w // 4) * (h // 4) * 4
        out["dbk"] = {"deblock": 2 * P + maps}
        out["sao"] = {"sao_stats": 2 * P + self.nctu_x * self.nctu_y * 3 * 2560, "sao_apply": 2 * P}
        nctu = self.nctu_x * self.nctu_y
        # (with the classifier inside the covariance launch its class map is written instead of read: the same bytes)
        out["alf"] = {"alf_classify": Y + Y // 16, "alf_stats": (2 * Y + Y // 16) * 2 + 2 * (P - Y) + nctu * 25 * (183 + 57) * 8 + nctu * 2 * 57 * 8,
                      "alf_filter": 2 * P + Y // 16}
        return out

    def unique_bytes(self):
        """Bytes a launch must touch at least once (every sample of its inputs and outputs counted ONCE, however many blocks'
        windows overlap it): the figure the counter traffic (FETCH_SIZE x2 + WRITE_SIZE) is to be compared with."""
        w, h = self.w, self.h
        P, Y = w * h * 3, w * h * 2
        out = {s: dict(v) for s, v in self.algorithmic_bytes().items()}
        for s, blk in self.me.items():
            for (dx0, dy0, nx, ny, sx, sy) in self.me_grids:
                out["me"]["sad_search_%dx%d_%dx%d" % (s, s, nx, ny)] = (w + (nx - 1) * sx) * (h + (ny - 1) * sy) * 2 + Y + blk.size * 24
        if self.hier_me:     # the hierarchical launch reads the reference picture + its search margin and the original once and writes six result arrays
            R = 5 * (self.raster_range // 5)
            out["me"] = {"hier_search": (w + 2 * R) * (h + 2 * R) * 2 + Y + 2 * sum(b.size for b in self.me.values()) * 24}
        out["frac"] = {"frac_refine_16x16": 2 * Y + self.frac.size * 32}
        out["mc"] = {"mc_picture": 3 * P}
        return out

    def useful_sad_insts(self):
        """launch group -> v_sad_u16 wave-instructions its arithmetic needs (2 abs-differences per lane-operation, 64 lanes): the
        numerator of the issue-rate roofline of the search kernels."""
        out = {}
        for s, blk in self.me.items():
            for (dx0, dy0, nx, ny, sx, sy) in self.me_grids:
                out["me/sad_search_%dx%d_%dx%d" % (s, s, nx, ny)] = blk.size * nx * ny * s * (s >> 1) / 2.0 / 64.0
        # hierarchical launch: the NON-REDUNDANT count -- every 16x16 SAD of both grids once; the 32x32 / 64x64 SADs are sums
        if 16 in self.me:
            out["me/hier_search"] = sum(self.me[16].size * g[2] * g[3] * 16 * 8 / 2.0 / 64.0 for g in self.me_grids)
        return out

    def profile_kernel_hint(self, group):
        """name (substring) of the rocprofv3 kernel that does the work of a launch group (None: not mapped)"""
        stage, name = group.split("/")
        if stage == "me" and name == "hier_search":
            return "me_hier_kernel"
        if stage == "me":
            return "sad_raster5" if name.endswith("%dx%d" % (self.me_grids[1][2], self.me_grids[1][3])) else "sad_dense"
        return {"frac_refine_16x16": "frac16m_kernel", "mc_picture": "mc_mfma_kernel", "deblock": "deblock_block_kernel",
                "sao_stats": "sao_stats_picture_kernel", "sao_apply": "sao_apply_strip_kernel", "alf_classify": "alf_classify_kernel",
                "alf_stats": "alf_stats_picture_kernel", "alf_filter": "alf_filter_picture_kernel", "resi_chain": "rc_chain_kernel"}.get(name)

    def margins(self):
        return [(MARGIN, MARGIN), (MARGIN // 2, MARGIN // 2), (MARGIN // 2, MARGIN // 2)]

    # device-state keys of the per-picture side information (run_gpu), and the host arrays they are uploaded from
    SIDE_KEYS = ("frac_blk", "mc_pic", "rc", "bands_rest", "edge_ver", "edge_hor", "qp_luma", "qp_chroma", "sao", "alf_en")

    def side_host(self):
        """{state key: numpy byte array (or list of them)} -- what bench.py's upload leg sends per picture besides the original"""
        b = lambda a: np.ascontiguousarray(a).view(np.uint8).reshape(-1)
        return {"frac_blk": b(self.frac), "mc_pic": b(self.mc_pic if self._mc_pic_patched is None else self._mc_pic_patched), "rc": b(self.rc), "bands_rest": b(self.bands_rest),
                "edge_ver": b(self.edge_ver), "edge_hor": b(self.edge_hor), "qp_luma": b(self.qp_luma), "qp_chroma": b(self.qp_chroma),
                "sao": [b(p) for p in self.sao], "alf_en": [b(e) for e in self.alf_enable]}

    # ------------------------------------------------------------------------------------------------------
    def run_gpu(self, dev_state=None, timer=None, overlap=False, alone=None, pre_mc=None, rotate=1, on_input_set=None):
        """One step on the GPU through ops/C-ABI.  Returns (state, outputs dict of CUDA tensors).

        overlap=False: every launch on the current stream, in stage order.
        overlap=True: the stages' real dependencies only.  The searches, the fractional refinement and the statistics do not
        feed the reconstruction chain (mc -> residual -> transforms -> reco -> deblock -> SAO -> ALF), so they run on three
        side HIP streams beside it; each of these kernels alone leaves SIMDs idle while its workgroups stage their windows /
        tiles, and concurrent kernels fill those gaps.  ONE search launch group runs first and alone on the main stream, so that
        its event-timed duration is a kernel time and not a share of an overlapped interval: `alone` = (block size, grid index)
        names it (bench.py passes the dominant one); default: the first size's raster search.
        `pre_mc`: called on the main stream right before the motion compensation, the first reader of the second reference picture
        (bench.py installs the boundary picture of a chunk hand-over there: the searches in front of it do not wait for the transfer).
        `rotate` = K > 1: the picture's inputs that are NEW for every picture of an encode -- the original and the first reference picture -- cycle
        through K resident copies at different addresses (same content, same results), so that consecutive pictures do not find their inputs in the
        256 MB memory-side cache: K x 56 MB at 4K.  The second reference picture (the hand-over slot, re-used by every picture of a chunk as a
        reference picture is), the descriptor lists and the intermediates between the stages of one picture stay where they are.
        `on_input_set(k, state)`: called at the start of the picture with the index of the input set it uses (bench.py's upload leg)."""
        import torch
        from . import ops
        T = timer or (lambda name: _NullCtx())
        st = dev_state
        if st is None:
            d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()

            def planes(offsets, shapes, arrs=None):       # the planes of one picture as views of one allocation
                flat = torch.zeros(offsets[-1], dtype=torch.int16, device="cuda")
                views = [flat[o:o + hh * ww].view(hh, ww) for o, (hh, ww) in zip(offsets, shapes)]
                for v, a in zip(views, arrs or []):
                    v.copy_(torch.from_numpy(np.ascontiguousarray(a)))
                return views
            st = {"org": [d(p) for p in self.org],
                  "ref0": planes(self.ref_plane_off, [p.shape for p in self.ref0_pad], self.ref0_pad),
                  "ref1": planes(self.ref_plane_off, [p.shape for p in self.ref1_pad], self.ref1_pad),
                  "me_blk": {s: ops.struct_to_device(b) for s, b in self.me.items()},
                  "frac_blk": ops.struct_to_device(self.frac),
                  "mc_pic": ops.struct_to_device(self.mc_pic),
                  "tr": ops.struct_to_device(self.tr), "bands_luma": ops.struct_to_device(self.bands_luma),
                  "bands_chroma": ops.struct_to_device(self.bands_chroma), "bands_rest": ops.struct_to_device(self.bands_rest),
                  "edge_ver": d(self.edge_ver), "edge_hor": d(self.edge_hor), "qp_luma": d(self.qp_luma), "qp_chroma": d(self.qp_chroma),
                  "sao": [ops.sao_params_to_device(p) for p in self.sao], "alf_en": [d(e) for e in self.alf_enable]}
            e16 = lambda hh, ww: torch.empty((hh, ww), dtype=torch.int16, device="cuda")
            w, h = self.w, self.h
            st["pred"] = planes(self.pic_plane_off, [(h, w), (h // 2, w // 2), (h // 2, w // 2)])
            st["resi"] = e16(h, w)
            st["resi2"] = torch.zeros((h, w), dtype=torch.int16, device="cuda")   # rows below the last 64-multiple stay 0
            st["coef"] = torch.empty(self.n_coef, dtype=torch.int32, device="cuda")
            st["level"] = torch.empty(self.n_coef, dtype=torch.int32, device="cuda")
            st["dqcoef"] = torch.empty(self.n_coef, dtype=torch.int32, device="cuda")
            st["quant"], st["dqtr"] = ops.struct_to_device(self.quant), ops.struct_to_device(self.dqtr)
            if self.depquant:
                st["dq"], st["dq_rates"] = ops.struct_to_device(self.dq), ops.struct_to_device(self.dq_rates)
            st["rc"] = ops.struct_to_device(self.rc)
            st["rec"] = planes(self.pic_plane_off, [(h, w), (h // 2, w // 2), (h // 2, w // 2)])
            if self.fused_resi:
                delta = (st["rec"][0].data_ptr() - st["pred"][0].data_ptr()) // 2       # samples from the prediction allocation to the reconstruction allocation
                patched = self.mc_pic.copy()
                patched["dst_off"][self.mc_rec_mask] += delta
                self._mc_pic_patched = patched
                st["mc_pic"] = ops.struct_to_device(patched)
            st["sao_out"] = [e16(h, w), e16(h // 2, w // 2), e16(h // 2, w // 2)]
            st["alf_out"] = [e16(h, w), e16(h // 2, w // 2), e16(h // 2, w // 2)]
            st["in_sets"], st["rot"] = [(st["org"], st["ref0"])], 0
            st["_planes"] = planes
        K = max(1, int(rotate))
        while len(st["in_sets"]) < K:
            d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
            st["in_sets"].append(([d(p) for p in self.org], st["_planes"](self.ref_plane_off, [p.shape for p in self.ref0_pad], self.ref0_pad)))
        # the side information that an encoder derives anew for every picture (PU / TU descriptor lists, deblocking maps, SAO parameters, ALF switches):
        # one resident copy per input set, so that the upload leg of bench.py can fill the next set while the current one is in use
        if "side_sets" not in st:
            st["side_sets"] = [{k: st[k] for k in self.SIDE_KEYS}]
        while len(st["side_sets"]) < K:
            st["side_sets"].append({k: ([t.clone() for t in st["side_sets"][0][k]] if isinstance(st["side_sets"][0][k], list) else st["side_sets"][0][k].clone())
                                    for k in self.SIDE_KEYS})
        if K > 1:
            st["rot"] = (st["rot"] + 1) % K
            st["org"], st["ref0"] = st["in_sets"][st["rot"]]
            st.update(st["side_sets"][st["rot"]])
        if on_input_set is not None:
            on_input_set(st["rot"], st)
        out = {}
        bd, mx = self.bd, self.mx
        cfg_mv = ops.MvCost(self.mvcost.lambda_, self.mvcost.pred_hor, self.mvcost.pred_ver, self.mvcost.cost_scale, self.mvcost.imv_shift)
        fmv = ops.MvCost(self.frac_mvcost.lambda_, self.frac_mvcost.pred_hor, self.frac_mvcost.pred_ver, 0, 0)
        main = torch.cuda.current_stream()
        if overlap and "streams" not in st:
            st["streams"] = [torch.cuda.Stream() for _ in range(3)]
        side = st["streams"] if overlap else [main, main, main]

        class _On:                                  # run a block on a side stream after the given events of other streams
            def __init__(self, stream, *events):
                self.stream, self.events = stream, events

            def __enter__(self):
                self.ctx = torch.cuda.stream(self.stream)
                self.ctx.__enter__()
                for e in self.events:
                    if e is not None:
                        self.stream.wait_event(e)

            def __exit__(self, *a):
                return self.ctx.__exit__(*a)

        def mark():                                 # event on the current stream (None in the serial schedule)
            if not overlap:
                return None
            e = torch.cuda.Event()
            e.record()
            return e

        def search(s, grid):
            dx0, dy0, nx, ny, sx, sy = grid
            with T("me/sad_search_%dx%d_%dx%d" % (s, s, nx, ny)):
                # xPatternSearch and the raster stage of xTZSearch keep only the best candidate
                # (InterSearch.cpp:1887-1935, 1979-2000): no SAD surface
                sad, best = ops.sad_search(st["org"][0], st["ref0"][0], st["me_blk"][s], self.me[s].size, s, s, 1,
                                           dx0, dy0, nx, ny, sx, sy, cfg_mv, want_sad=False)
            out["me_sad_%d_%d" % (s, nx)] = sad
            out["me_best_%d_%d" % (s, nx)] = best

        sizes = sorted(self.me)
        dense, raster = self.me_grids[0], self.me_grids[1]

        def frac():
            with T("frac/frac_refine_16x16"):
                out["frac"] = ops.frac_refine(st["org"][0], st["ref0"][0], st["frac_blk"], self.frac.size, 16, 16, bd, fmv, True, (0, mx))

        if self.hier_me:
            # ---- me, hierarchical: ONE launch answers the six searches (InterSearch.cpp:2159-2169 raster, :1886-1935 +-4) on the main stream, alone;
            # the fractional refinement follows on a side stream
            with T("me/hier_search"):
                rb, db = ops.me_hier_search(st["org"][0], st["ref0"][0], (0, 0), (MARGIN, MARGIN), self.w // 16, self.h // 16, 1, self.raster_range, 4, cfg_mv)
            for k, s in enumerate((16, 32, 64)):
                out["me_sad_%d_%d" % (s, raster[2])] = out["me_sad_%d_%d" % (s, dense[2])] = None
                out["me_best_%d_%d" % (s, dense[2])] = db[k]
                out["me_best_%d_%d" % (s, raster[2])] = rb[k]          # (a raster of 9 x 9 positions shares the key: the raster result stays, as in the per-size order)
            if not overlap:
                frac()
            else:
                with _On(side[1], mark()):
                    frac()
        # ---- me / frac, per-size form.  Serial order: per size the +-4 grid, then the raster.  Overlapped: the first size's raster alone on
        # the main stream, everything else on side streams 0 / 1 after it.
        elif not overlap:
            for s in sizes:
                search(s, dense)
                search(s, raster)
            frac()
        else:
            grids = {0: dense, 1: raster}
            first = alone if alone is not None else (sizes[0], 1)
            rest = [(sz, gi) for gi in (1, 0) for sz in sizes if (sz, gi) != first]
            search(first[0], grids[first[1]])
            e_first = mark()
            with _On(side[0], e_first):
                for sz, gi in rest[0::2]:
                    search(sz, grids[gi])
            with _On(side[1], e_first):
                for sz, gi in rest[1::2]:
                    search(sz, grids[gi])
                frac()
        # ---- mc
        if pre_mc is not None:
            pre_mc()
        with T("mc/mc_picture"):                  # offsets in the list are relative to the luma planes; the chroma planes follow
            ops.mc_picture_batch(st["ref0"][0], st["ref1"][0], st["pred"][0], st["mc_pic"], self.mc_pic.size, bd, (0, mx))
        # ---- residual / transforms / reconstruction
        if self.fused_resi:
            with T("resi/resi_chain"):
                out["abs_sum"] = ops.resi_chain_runs_batch(st["org"][0], st["pred"][0], st["rec"][0], st["level"], st["rc"], self.tr.size, self.tu_runs, bd, (0, mx))
                # (outside the TU tiling the residual is zero, and chroma carries no residual in this workload: reconstruction = clipped prediction,
                # B4 copyClip / xReconInter with cbf == 0 -- written there by the motion compensation itself: mc_rec_mask)
        else:
            sub = ops.PelopCfg(0, 0, 0, 0, 0, mx)
            rec_cfg = ops.PelopCfg(0, 0, 0, 1, 0, mx)
            with T("resi/subtract"):
                ops.pelop_batch(3, st["org"][0], st["pred"][0], st["resi"], st["bands_luma"], self.bands_luma.size, sub)
            with T("resi/tr_fwd"):
                ops.tr_fwd_batch(st["resi"], st["coef"], st["tr"], self.tr.size, bd)
            if self.depquant:
                with T("resi/depquant"):
                    out["abs_sum"] = ops.depquant_batch(st["coef"], st["level"], st["dq"], self.tr.size, st["dq_rates"], self.n_coef, bd)
            else:
                with T("resi/quant"):
                    out["abs_sum"] = ops.quant_batch(st["coef"], st["level"], st["quant"], self.tr.size, bd)
            with T("resi/dequant_tr_inv"):
                ops.dequant_tr_inv_batch(st["level"], st["resi2"], st["dqtr"], self.tr.size, bd, st["dqcoef"])
            with T("resi/reco"):
                ops.pelop_batch(1, st["pred"][0], st["resi2"], st["rec"][0], st["bands_luma"], self.bands_luma.size, rec_cfg)
                st["rec"][1].copy_(st["pred"][1])
                st["rec"][2].copy_(st["pred"][2])
        out["coef"] = st["level"]
        # ---- deblock (in place on rec)
        dcfg = ops.deblock_cfg(bd)
        with T("dbk/deblock"):
            ops.deblock(st["rec"][0], st["rec"][1], st["rec"][2], st["edge_ver"], st["edge_hor"], st["qp_luma"], st["qp_chroma"], dcfg)
        e_dbk = mark()
        # ---- SAO (the statistics only read the deblocked picture: side stream 2); picture-level entry points: three planes per launch
        with _On(side[2], e_dbk):
            with T("sao/sao_stats"):
                sao_stats = ops.sao_stats_picture(st["org"], st["rec"], CTU, bd)
        with T("sao/sao_apply"):
            ops.sao_apply_picture(st["rec"], st["sao_out"], CTU, bd, st["sao"], (0, mx))
        out["sao_stats"] = sao_stats
        # ---- ALF (the covariances read the SAO output and the classifier: side stream 2)
        if self.fuse_alf and not overlap:
            # ALFProcess' front end as ONE launch: the covariance workgroups derive the classes of their CTU from the tile they hold
            with T("alf/alf_stats"):
                cls, a7, a5, ac = ops.alf_classify_stats_picture(st["org"], st["sao_out"], CTU, bd)
        else:
            with T("alf/alf_classify"):
                cls = ops.alf_classify(st["sao_out"][0], bd)
            e_cls = mark()
            with _On(side[2], e_cls):
                with T("alf/alf_stats"):
                    a7, a5, ac = ops.alf_stats_picture(st["org"], st["sao_out"], CTU, cls)
        with T("alf/alf_filter"):
            ops.alf_filter_picture(st["sao_out"], st["alf_out"], CTU, cls, 1, self.alf_luma_coeff, self.alf_chroma_coeff, st["alf_en"], (0, mx))
        if overlap:                                 # join: the step is complete (and its buffers reusable) when `main` is
            for sd in side:
                e = torch.cuda.Event()
                e.record(sd)
                main.wait_event(e)
        out.update({"cls": cls, "alf_stats7": a7, "alf_stats5": a5, "alf_stats_c": ac, "final": st["alf_out"],
                    # the prediction picture; one-pass form: luma inside the TU tiling only (every other PU was stored into the reconstruction picture)
                    "pred": st["pred"], "pred_valid": ((self.tiled_h, self.tiled_w) if self.fused_resi else None)})
        return st, out


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
