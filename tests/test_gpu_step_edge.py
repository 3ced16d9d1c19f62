"""The resident batch-of-8 kernel (one launch per pretrain_from_features call) at the edges of its
one-XCD exchange, whose partial sums travel as fixed point below 2^19: partial sums beyond that, and
non-finite values, must behave as the f32 paths and the reference's arithmetic do (lib.rs:1016-1059
carries them on), not be clamped; and the dropout draws of a seeded call must not depend on which
path ran."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIMS = (60, 512, 256, 16)


@pytest.fixture(scope="module")
def sz():
    import streamz_amd
    return streamz_amd


def _train(sz, params, wins, perm, one_xcd, lr=0.02, cls=3):
    """(parameters after the call, loss, launches of the resident kernel)"""
    old = os.environ.get("STREAMZ_STEP_ONE_XCD")
    os.environ["STREAMZ_STEP_ONE_XCD"] = "1" if one_xcd else "0"   # read at every call
    try:
        net = sz.SimpleNeuralNet(*DIMS, init=params)
        sz.profile_enable(True)
        sz.profile_report()
        loss = sz.pretrain_from_features(net, wins, cls, DIMS[3], perm.shape[0], lr, 0.0, 8, perm=perm)
        tags = sz.profile_report()
        sz.profile_enable(False)
    finally:
        if old is None:
            os.environ.pop("STREAMZ_STEP_ONE_XCD", None)
        else:
            os.environ["STREAMZ_STEP_ONE_XCD"] = old
    assert "fwd2" not in tags, tags            # the per-layer path did not run
    return net.get_flat(), loss, tags["steps"][0]


def _data(seed, n=41, epochs=2):
    rng = np.random.default_rng(seed)
    shapes = [(60, 512), (512,), (512, 256), (256,), (256, 16), (16,)]
    params = [rng.uniform(-0.5, 0.5, s).astype(np.float32) for s in shapes]
    wins = rng.standard_normal((n, 60)).astype(np.float32)
    perm = np.stack([rng.permutation(n) for _ in range(epochs)]).astype(np.uint32)
    return params, wins, perm


def test_partial_sums_beyond_the_fixed_point_range_take_the_f32_exchange(sz):
    params, wins, perm = _data(3)
    # benign weights: one launch, and the one-XCD exchange agrees with the chip-wide one to rounding
    # (its sums are formed in fixed point: more exact than 15 float additions, not the same bits)
    a, loss_a, launches_a = _train(sz, params, wins, perm, one_xcd=True)
    b, loss_b, launches_b = _train(sz, params, wins, perm, one_xcd=False)
    assert launches_a == launches_b == 1
    np.testing.assert_allclose(a, b, atol=2e-5)
    # second-layer weights of tens of thousands: a workgroup's partial sums pass 2^19 while the sums over
    # the workgroups partly cancel.  Clamped partials would give other sums; the call has to notice,
    # repeat on the f32 exchange (a second launch) and return exactly what that exchange gives.
    big = [p.copy() for p in params]
    big[2] *= 1.0e5
    assert np.abs(np.maximum(wins[perm[0, :8]] @ big[0][:, :32] + big[1][:32], 0) @ big[2][:32]).max() > 2 ** 19
    a, loss_a, launches_a = _train(sz, big, wins, perm, one_xcd=True)
    b, loss_b, launches_b = _train(sz, big, wins, perm, one_xcd=False)
    assert (launches_a, launches_b) == (2, 1)
    np.testing.assert_array_equal(a, b)
    assert loss_a == loss_b and np.isfinite(a).all()
    # the fallback says nothing about later calls: a benign one is a single launch again
    _, _, launches = _train(sz, params, wins, perm, one_xcd=True)
    assert launches == 1


def _same_with_nans(a, b, atol=0.0):
    np.testing.assert_array_equal(np.isnan(a), np.isnan(b))
    ok = ~np.isnan(a)
    if atol:
        np.testing.assert_allclose(a[ok], b[ok], atol=atol)
    else:
        np.testing.assert_array_equal(a[ok], b[ok])


def test_non_finite_values_are_carried_on_like_the_f32_paths(sz):
    from oracle import pyoracle as po
    params, wins, perm = _data(5)

    def oracle(w):
        onet = po.Net(*DIMS)
        onet.set_params(params)
        onet.pretrain_from_features(w, 3, DIMS[3], perm.shape[0], 0.02, 0.0, 8, perm=perm)
        return np.concatenate([p.ravel() for p in onet.get_params()])

    # A NaN input: the reference's ReLU (v > 0 ? v : 0, lib.rs:882) turns the NaN pre-activations into
    # zeros, so the forward pass stays finite and the partial sums fit; the NaN reaches row 17 of w1
    # through dW1 = x (x) delta1 (lib.rs:1041-1044) and nothing else.  No fallback is needed.
    bad = wins.copy()
    bad[int(perm[0, 9]), 17] = np.nan        # enters the second batch of the first epoch
    a, _, launches_a = _train(sz, params, bad, perm, one_xcd=True)
    b, _, launches_b = _train(sz, params, bad, perm, one_xcd=False)
    assert (launches_a, launches_b) == (1, 1)
    w1 = a[:60 * 512].reshape(60, 512)
    assert np.isnan(w1[17]).all() and not np.isnan(np.delete(w1, 17, axis=0)).any() and not np.isnan(a[60 * 512:]).any()
    _same_with_nans(a, b, atol=2e-5)
    _same_with_nans(a, oracle(bad), atol=1e-4)
    # An infinite input: +-inf pre-activations, an infinite h1, partial sums that are infinite or NaN:
    # the fixed-point exchange cannot carry them, the f32 one does.
    inf = wins.copy()
    inf[int(perm[0, 0]), 3] = np.inf
    a, _, launches_a = _train(sz, params, inf, perm, one_xcd=True)
    b, _, launches_b = _train(sz, params, inf, perm, one_xcd=False)
    assert (launches_a, launches_b) == (2, 1)
    assert np.isnan(a).any()
    _same_with_nans(a, b)
    np.testing.assert_array_equal(np.isnan(a), np.isnan(oracle(inf)))


CHILD = r"""
import sys
import numpy as np
sys.path.insert(0, %(root)r)
import streamz_amd as sz
rng = np.random.default_rng(77)
dims = (60, 512, 256, 16)
shapes = [(60, 512), (512,), (512, 256), (256,), (256, 16), (16,)]
params = [rng.uniform(-0.5, 0.5, s).astype(np.float32) for s in shapes]
wins = rng.standard_normal((53, 60)).astype(np.float32)
net = sz.SimpleNeuralNet(*dims, init=params)
sz.profile_enable(True)
sz.profile_report()
loss = sz.pretrain_from_features(net, wins, 5, 16, 3, 0.02, 0.2, 8, seed=4242)
tags = sz.profile_report()
np.savez(sys.argv[1], params=net.get_flat(), loss=loss, per_layer="fwd2" in tags)
"""


def test_seeded_dropout_draws_do_not_depend_on_the_path(tmp_path):
    """pretrain_from_features(dropout 0.2, seed) through the resident kernel and, with the kernel made
    to give up, through one launch per layer and step: the same shuffles and the same dropout draws, so
    the same training up to the kernels' rounding."""
    outs = []
    for force in (None, "1"):
        env = dict(os.environ)
        env.pop("STREAMZ_STEP_FORCE_ABORT", None)
        if force:
            env["STREAMZ_STEP_FORCE_ABORT"] = force
        out = str(tmp_path / f"run{force}.npz")
        r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}, out], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        outs.append(np.load(out))
    assert not outs[0]["per_layer"] and outs[1]["per_layer"]
    np.testing.assert_allclose(outs[0]["params"], outs[1]["params"], atol=2e-5)   # other draws would move them by 1e-2
    assert abs(float(outs[0]["loss"]) - float(outs[1]["loss"])) < 1e-5
