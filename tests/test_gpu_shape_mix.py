"""Parity on the shapes the reference encoder really calls with: every distinct call signature of the committed trace
(tests/golden/trace_ragop16_416x240_10b_q32.npz, taken by the shim's trace mode: shapes and parameters only) through the batch entry points,
bit-exact against the oracle."""
import numpy as np
import pytest

from oraclelib import oracle, p

pytestmark = pytest.mark.gpu


def test_every_traced_call_signature_matches_the_oracle():
    from vvcsoftware_vtm_amd import shape_mix
    res = shape_mix.parity(oracle(), p)
    assert len(res) >= 8, res
    for name, (n, same) in res.items():
        assert n > 0 and same, (name, n)


def test_trace_fixture_layout():
    from vvcsoftware_vtm_amd import shape_mix
    hist, meta = shape_mix.load_trace()
    assert hist.shape[1] == 7 and meta["record"] == ["entry", "w", "h", "a", "b", "c"]
    assert int(hist[:, 6].sum()) == meta["records"]
    # the point of the fixture: most distortion calls of the real encoder are 4 or 8 wide
    d = hist[hist[:, 0] == 0]
    assert d[d[:, 1] <= 8][:, 6].sum() > 0.5 * d[:, 6].sum()


def test_dequant_tr_inv_long_batch_in_class_order():
    """a batch long enough (n >= 16384) that vvcgpu_dequant_tr_inv_batch bins the descriptor indices by phase class on the device and walks the class
    lists (dqtr_classify_kernel): the trace's call mix, ~0.5 M samples, plus a run of 16-wide TUs that fills whole 64-descriptor batches of the
    lane-group-of-16 class (position 63 of a bin's list once collided with the "not a lane-group TU" mark) -- bit-exact against the oracle"""
    from vvcsoftware_vtm_amd import shape_mix as sm
    hist, _ = sm.load_trace()
    name, entry, keep, build = [r for r in sm.rows(hist) if r[0] == "dequant_tr_inv_batch"][0]
    sig = sm.signatures(hist, entry, keep)
    rng = np.random.default_rng(23)
    calls = sm.draw(sig, 1 << 20, rng)
    extra = np.zeros((400, 5), np.int64)
    extra[:, 0] = rng.choice([16, 16, 8, 4], 400)
    extra[:, 1] = np.where(extra[:, 0] == 16, rng.choice([16, 8, 4], 400), 16)
    extra[:, 4] = rng.integers(0, 2, 400)
    calls = np.concatenate([calls, extra])
    assert len(calls) >= 16384
    _, _, check = build(calls, rng)
    assert check(oracle(), p)


@pytest.mark.parametrize("name", ["tr_fwd_batch", "tr_inv_batch"])
def test_plain_transforms_long_batch_in_bin_order(name):
    """n >= 16384: vvcgpu_tr_fwd_batch / vvcgpu_tr_inv_batch bin the small TUs on the device too and the small kernel walks the bin lists
    (tr_collect_large_kernel, small_setup) -- the trace's call mix, ~1 M samples, bit-exact against the oracle"""
    from vvcsoftware_vtm_amd import shape_mix as sm
    hist, _ = sm.load_trace()
    _, entry, keep, build = [r for r in sm.rows(hist) if r[0] == name][0]
    sig = sm.signatures(hist, entry, keep)
    rng = np.random.default_rng(29)
    calls = sm.draw(sig, 1 << 20, rng)
    assert len(calls) >= 16384
    _, _, check = build(calls, rng)
    assert check(oracle(), p)
