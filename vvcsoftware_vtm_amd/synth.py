"""Synthetic YUV 4:2:0 clips: no test sequences ship with the reference, so every
workload in tests/ and bench.py is generated from a seed.

gen_yuv(W, H, N, bd, seed): luma = window of a 1/f^1.2 power-law noise field, global pan (+1,+2)
px/frame, one 64x96 foreign-texture object moving (+3,-4) px/frame, plus N(0, 6*2^(bd-10)) noise;
Cb/Cr = mid +- (Y-mid)/3, /5 subsampled.  Returns a list of (Y, Cb, Cr) uint16 arrays.
"""
import numpy as np


def _powerlaw_field(h, w, rng, alpha=1.2):
    fy = np.fft.fftfreq(h)[:, None]
    fx = np.fft.rfftfreq(w)[None, :]
    f = np.sqrt(fx * fx + fy * fy)
    f[0, 0] = 1.0
    amp = 1.0 / f ** alpha
    amp[0, 0] = 0.0
    ph = rng.uniform(0, 2 * np.pi, size=amp.shape)
    spec = amp * np.exp(1j * ph)
    fld = np.fft.irfft2(spec, s=(h, w))
    fld -= fld.mean()
    fld /= fld.std() + 1e-12
    return fld


def gen_yuv(W, H, N, bd=10, seed=20261003, warp=None, noise=6.0):
    """`warp` = (degrees, zoom) per frame: the background of frame n is the field rotated by n * degrees about the picture centre and scaled by
    zoom ** n instead of panned (content for the affine tools of the encoder; used by tests/golden/gen_deblock.py only -- the default clip is
    unchanged).  `noise`: standard deviation of the per-frame noise at 10 bit."""
    rng = np.random.default_rng(seed)
    mid = 1 << (bd - 1)
    mx = (1 << bd) - 1
    amp = 0.22 * (1 << bd)
    big = _powerlaw_field(H + 200, W + 200, rng) * amp + mid
    obj = _powerlaw_field(96, 64, rng, 0.8) * amp * 1.3 + mid
    frames = []
    for n in range(N):
        oy, ox = 100 - 2 * n, 100 - 1 * n           # pan (+1,+2) px/frame of the content
        oy %= 200
        ox %= 200
        if warp is None:
            y = big[oy:oy + H, ox:ox + W].copy()
        else:
            from scipy import ndimage
            a, z = np.deg2rad(warp[0] * n), warp[1] ** n
            yy, xx = np.meshgrid(np.arange(H) - H / 2.0, np.arange(W) - W / 2.0, indexing="ij")
            sy = (np.cos(a) * yy - np.sin(a) * xx) / z + H / 2.0 + 100.0
            sx = (np.sin(a) * yy + np.cos(a) * xx) / z + W / 2.0 + 100.0
            y = ndimage.map_coordinates(big, [sy, sx], order=3, mode="reflect")
        py = (H // 2 - 48 - 4 * n) % max(1, H - 96)
        px = (W // 4 + 3 * n) % max(1, W - 64)
        y[py:py + 96, px:px + 64] = obj
        y += rng.normal(0.0, noise * 2.0 ** (bd - 10), size=y.shape)
        Y = np.clip(np.rint(y), 0, mx).astype(np.uint16)
        ys = Y.astype(np.float64)
        sub = (ys[0::2, 0::2] + ys[1::2, 0::2] + ys[0::2, 1::2] + ys[1::2, 1::2]) / 4.0
        Cb = np.clip(np.rint(mid + (sub - mid) / 3.0), 0, mx).astype(np.uint16)
        Cr = np.clip(np.rint(mid - (sub - mid) / 5.0), 0, mx).astype(np.uint16)
        frames.append((Y, Cb, Cr))
    return frames


def write_yuv(path, frames, bd=10):
    with open(path, "wb") as f:
        for planes in frames:
            for p in planes:
                if bd > 8:
                    f.write(p.astype("<u2").tobytes())
                else:
                    f.write(p.astype(np.uint8).tobytes())


def random_plane(h, w, bd=10, seed=1):
    """Uniform-random plane for kernel micro-benchmarks (no early-out can help)."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 1 << bd, size=(h, w), dtype=np.int16)
