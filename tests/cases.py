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
