"""Parity of the calls the benchmark times, under their own randomness and at the wide
configuration's own shape, plus train_from_feature_map and the seeded initialisation."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from helpers import dropout_keep, grad_view, mix64

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sz():
    import streamz_amd
    return streamz_amd


@pytest.fixture(scope="module")
def po():
    from oracle import pyoracle
    return pyoracle


def _oracle_grad_threads(onet, x, labels, threads=16, pieces=None):
    """oracle gradient of many rows, the rows spread over host threads (ctypes releases the GIL);
    each of the `pieces` partial sums is the oracle's f32 sum, the pieces are added in f64"""
    parts = np.array_split(np.arange(len(x)), pieces or threads)
    with ThreadPoolExecutor(threads) as pool:
        gs = list(pool.map(lambda ix: onet.grad(x[ix[0]:ix[-1] + 1], labels=labels[ix[0]:ix[-1] + 1]).astype(np.float64),
                           [p for p in parts if len(p)]))
    return np.sum(gs, axis=0)


def _entries_of_near_zero_units(x, params, dims, size, eps=1.5e-6):
    """Mask over the flat gradient of the entries that one first-layer ReLU decision can move: column j of
    dW1, db1[j] and row j of dW2 for every unit j that has a pre-activation within eps of zero on some row."""
    w1, b1 = params[0].astype(np.float64), params[1].astype(np.float64)
    near = np.zeros(dims[1], bool)
    for r0 in range(0, len(x), 20000):
        z = x[r0:r0 + 20000].astype(np.float64) @ w1 + b1
        near |= (np.abs(z) < eps).any(axis=0)
    loose = np.zeros(size, bool)
    loose[:60 * dims[1]].reshape(60, dims[1])[:, near] = True
    loose[60 * dims[1]:61 * dims[1]][near] = True
    off_w2 = 61 * dims[1]
    loose[off_w2:off_w2 + dims[1] * dims[2]].reshape(dims[1], dims[2])[near, :] = True
    return loose


@pytest.mark.parametrize("gemm", ["f32", "bf16x3"])
@pytest.mark.parametrize("dims,n", [((60, 512, 256, 16), 5000), ((60, 512, 256, 64), 3000)])
def test_accumulate_dev_with_seeded_dropout_matches_oracle(sz, po, dims, n, gemm):
    """szh_net_accumulate_dev(dropout=0.2, seed) as bench.py calls it: the mask is rebuilt on the
    host from the same counter-based generator and the masked rows go to the oracle
    (lib.rs:119-129 no rescale, lib.rs:607-609 rows that end all-zero do not count)."""
    import torch
    rng = np.random.default_rng(31)
    onet = po.Net(*dims)
    params = [rng.uniform(-0.5, 0.5, s).astype(np.float32) for s in onet.shapes()]
    onet.set_params(params)
    net = sz.SimpleNeuralNet(*dims, init=params)
    net.set_gemm(gemm)   # bf16x3: the three large products on the split-bf16 kernels, same tolerances
    assert net.get_gemm() == gemm
    x = rng.standard_normal((n, 60)).astype(np.float32)
    x[11, 3:] = 0.0   # three non-zero inputs: some seeds drop all of them
    x[12] = 0.0
    labels = rng.integers(0, dims[3], n).astype(np.int32)
    p, seed = 0.2, 1003
    keep = dropout_keep(seed, n, p)
    assert 0.78 < keep.mean() < 0.82
    xm = np.where(keep, x, np.float32(0.0))
    live = np.any(xm != 0.0, axis=1)
    assert not live[12]
    dx, dl = torch.from_numpy(x).cuda(), torch.from_numpy(labels).cuda()
    s = torch.cuda.current_stream().cuda_stream
    net.zero_grad_dev(s)
    net.accumulate_dev(dx.data_ptr(), dl.data_ptr(), n, p, seed, s)
    torch.cuda.synchronize()
    g = grad_view(net).cpu().numpy()
    want = onet.grad(xm[live], labels=labels[live])
    scale = np.abs(want).max()
    np.testing.assert_allclose(g[:-2], want, atol=2e-5 * scale)
    assert g[-1] == live.sum()
    probs = onet.forward(xm[live])
    loss = -np.log(np.maximum(probs[np.arange(live.sum()), labels[live]], 1e-12)).sum()
    assert abs(g[-2] - loss) < 1e-4 * loss
    # another seed is another mask
    net.zero_grad_dev(s)
    net.accumulate_dev(dx.data_ptr(), dl.data_ptr(), n, p, seed + 1, s)
    torch.cuda.synchronize()
    assert np.abs(grad_view(net).cpu().numpy()[:-2] - g[:-2]).max() > 1e-3 * scale


@pytest.mark.parametrize("gemm", ["f32", "bf16x3"])
def test_accumulate_dev_at_the_wide_configuration(sz, po, gemm):
    """60-2048-2048-256 (the widest configuration): the chunked accumulate path with the GEMM output
    layer and the wave-per-row softmax, on 4 096 rows, against the oracle's gradient."""
    import torch
    rng = np.random.default_rng(41)
    dims = (60, 2048, 2048, 256)
    onet = po.Net(*dims)
    # weights scaled so that the activations stay in tanh's responsive range at this width
    params = [(rng.uniform(-0.5, 0.5, s) * (0.08 if len(s) == 2 and s[0] > 60 else 1.0)).astype(np.float32)
              for s in onet.shapes()]
    onet.set_params(params)
    net = sz.SimpleNeuralNet(*dims, init=params)
    net.set_gemm(gemm)
    n = 4096
    x = rng.standard_normal((n, 60)).astype(np.float32)
    labels = rng.integers(0, 256, n).astype(np.int32)
    dx, dl = torch.from_numpy(x).cuda(), torch.from_numpy(labels).cuda()
    s = torch.cuda.current_stream().cuda_stream
    net.zero_grad_dev(s)
    net.accumulate_dev(dx.data_ptr(), dl.data_ptr(), n, stream=s)
    torch.cuda.synchronize()
    g = grad_view(net).cpu().numpy()
    want = _oracle_grad_threads(onet, x, labels)
    scale = np.abs(want).max()
    np.testing.assert_allclose(g[:-2], want, atol=3e-5 * scale)
    assert g[-1] == n
    probs = onet.forward(x[:1280])   # >= 1024 rows, so that the forward pass also takes the chosen kernels
    np.testing.assert_allclose(net.forward(x[:1280]), probs, atol=1e-4)


def test_accumulate_dev_at_the_wide_configuration_on_the_walking_f32_kernel(sz, po):
    """60-2048-2048-256 at 16 421 rows: enough 256 x 256 tiles that the f32 mode runs every large product,
    the output layer's three included, on the walking kernel (ragged last tile, rows zeroed up to the next
    multiple of 32 for the k-major reads).  Against the oracle's gradient and probabilities."""
    import torch
    rng = np.random.default_rng(47)
    dims = (60, 2048, 2048, 256)
    onet = po.Net(*dims)
    params = [(rng.uniform(-0.5, 0.5, s) * (0.08 if len(s) == 2 and s[0] > 60 else 1.0)).astype(np.float32)
              for s in onet.shapes()]
    onet.set_params(params)
    net = sz.SimpleNeuralNet(*dims, init=params)
    n = 16421
    x = rng.standard_normal((n, 60)).astype(np.float32)
    labels = rng.integers(0, 256, n).astype(np.int32)
    dx, dl = torch.from_numpy(x).cuda(), torch.from_numpy(labels).cuda()
    s = torch.cuda.current_stream().cuda_stream
    net.zero_grad_dev(s)
    net.accumulate_dev(dx.data_ptr(), dl.data_ptr(), n, stream=s)
    torch.cuda.synchronize()
    g = grad_view(net).cpu().numpy()
    want = _oracle_grad_threads(onet, x, labels, pieces=64)
    scale = np.abs(want).max()
    loose = _entries_of_near_zero_units(x, params, dims, want.size)
    np.testing.assert_allclose(g[:-2][~loose], want[~loose], atol=3e-5 * scale)
    np.testing.assert_allclose(g[:-2][loose], want[loose], atol=3e-4 * scale)
    assert g[-1] == n
    tail = slice(n - 300, n)
    np.testing.assert_allclose(net.forward(x)[tail], onet.forward(x[tail]), atol=1e-4)


def test_accumulate_dev_at_a_row_count_that_takes_the_walking_f32_kernel(sz, po):
    """From 512 tiles of 256 x 256 per product on (131 072 rows at 60-512-256-16) fwd1, fwd2 and delta1 run
    on the 256-tile walking kernel: 140 003 rows (a ragged last tile) against the oracle's gradient and
    probabilities, and against the same rows fed in pieces small enough for the 128-tile kernel."""
    import torch
    rng = np.random.default_rng(43)
    dims = (60, 512, 256, 16)
    onet = po.Net(*dims)
    params = [rng.uniform(-0.5, 0.5, s).astype(np.float32) for s in onet.shapes()]
    onet.set_params(params)
    net = sz.SimpleNeuralNet(*dims, init=params)
    assert net.get_gemm() == "f32"
    n = 140003
    x = rng.standard_normal((n, 60)).astype(np.float32)
    x[n - 2] = 0.0   # an all-zero row does not count (lib.rs:607-609)
    live = np.any(x != 0.0, axis=1)
    labels = rng.integers(0, 16, n).astype(np.int32)
    dx, dl = torch.from_numpy(x).cuda(), torch.from_numpy(labels).cuda()
    s = torch.cuda.current_stream().cuda_stream
    net.zero_grad_dev(s)
    net.accumulate_dev(dx.data_ptr(), dl.data_ptr(), n, stream=s)
    torch.cuda.synchronize()
    g = grad_view(net).cpu().numpy()
    want = _oracle_grad_threads(onet, x[live], labels[live], pieces=512)   # short f32 sums on the oracle's side
    scale = np.abs(want).max()
    # Among 72 M first-layer pre-activations a few lie within an f32 rounding of zero, and there the
    # ReLU decision depends on how the 60-term sum was rounded (the oracle multiplies and adds, the
    # matrix instruction fuses).  Such a decision moves the unit's column of dW1, its db1 entry and its
    # row of dW2 by one row's contribution: those entries are compared at a looser bound.
    loose = _entries_of_near_zero_units(x, params, dims, want.size)
    assert loose[60 * dims[1]:61 * dims[1]].sum() < 64   # of 512 units
    np.testing.assert_allclose(g[:-2][~loose], want[~loose], atol=2e-5 * scale)
    np.testing.assert_allclose(g[:-2][loose], want[loose], atol=2e-4 * scale)
    assert g[-1] == n - 1
    # the same rows in four calls, each below the walking kernel's threshold
    net.zero_grad_dev(s)
    step = 40000
    for r0 in range(0, n, step):
        rows = min(step, n - r0)
        net.accumulate_dev(dx[r0:].data_ptr(), dl[r0:].data_ptr(), rows, stream=s)
    torch.cuda.synchronize()
    g4 = grad_view(net).cpu().numpy()
    # (the pieces' first layer runs on the other kernel too, with its own rounding of the near-zero sums)
    np.testing.assert_allclose(g[:-2][~loose], g4[:-2][~loose], atol=2e-6 * scale)
    np.testing.assert_allclose(g[:-2][loose], g4[:-2][loose], atol=2e-4 * scale)
    assert g4[-1] == n - 1 and abs(g4[-2] - g[-2]) < 1e-6 * g[-2]
    # forward: the last rows sit in the ragged tile
    probs = net.forward(x)
    tail = slice(n - 600, n)
    np.testing.assert_allclose(probs[tail], onet.forward(x[tail]), atol=2e-5)
    np.testing.assert_allclose(probs[:600], onet.forward(x[:600]), atol=2e-5)


def _shuffle_orders(seed, n_win, epochs):
    """the Fisher-Yates orders pretrain_from_features draws from `seed` (one draw counter per call)"""
    seed = np.uint64(seed)
    draw, out = 0, []
    for _ in range(epochs):
        order = list(range(n_win))
        if n_win > 1:
            js = mix64(seed ^ mix64(np.arange(draw, draw + n_win - 1, dtype=np.uint64)))
            draw += n_win - 1
            for t, i in enumerate(range(n_win, 1, -1)):
                j = int(js[t] % np.uint64(i))
                order[i - 1], order[j] = order[j], order[i - 1]
        out.append(order)
    return np.array(out, np.uint32).reshape(epochs, n_win)


@pytest.mark.parametrize("dims,batch", [((60, 64, 32, 3), 8), ((60, 48, 20, 3), 5)])
def test_train_from_feature_map_matches_per_file_loop_and_oracle(sz, po, dims, batch):
    """train_from_feature_map (lib.rs:632-665): files in list order, a path missing from the map is
    skipped, the mean of the per-file losses comes back, every trained file is recorded.  Compared
    with the library's own per-file loop (bit for bit) and with the oracle fed the same shuffles."""
    rng = np.random.default_rng(51)
    params = [rng.uniform(-0.5, 0.5, s).astype(np.float32) for s in po.Net(*dims).shapes()]
    fmap = {"a.wav": rng.standard_normal((37, 60)).astype(np.float32),
            "b.wav": rng.standard_normal((8, 60)).astype(np.float32),
            "d.wav": rng.standard_normal((21, 60)).astype(np.float32)}
    fmap["b.wav"][2] = 0.0
    files = [("a.wav", 0), ("b.wav", 2), ("missing.wav", 1), ("d.wav", 1)]
    epochs, lr, seed = 3, 0.05, 99
    net = sz.SimpleNeuralNet(*dims, init=params)
    loss = sz.train_from_feature_map(net, fmap, files, epochs, lr, 0.0, batch, seed=seed)
    assert net.file_lists() == [["a.wav"], ["d.wav"], ["b.wav"]]   # record_training_file, lib.rs:655

    loop = sz.SimpleNeuralNet(*dims, init=params)
    onet = po.Net(*dims)
    onet.set_params(params)
    losses, olosses = [], []
    for f, (path, cls) in enumerate(files):
        if path not in fmap:
            continue
        fseed = int(mix64(np.uint64(seed + f)))
        losses.append(sz.pretrain_from_features(loop, fmap[path], cls, dims[3], epochs, lr, 0.0, batch, seed=fseed))
        perm = _shuffle_orders(fseed, len(fmap[path]), epochs)
        olosses.append(onet.pretrain_from_features(fmap[path], cls, dims[3], epochs, lr, 0.0, batch, perm=perm))
    np.testing.assert_array_equal(net.get_flat(), loop.get_flat())
    assert loss == np.float32(np.sum(np.array(losses, np.float32), dtype=np.float32) / np.float32(len(losses)))
    want = np.concatenate([p.ravel() for p in onet.get_params()])
    np.testing.assert_allclose(net.get_flat(), want, atol=1e-4)
    assert abs(loss - np.mean(olosses)) < 1e-4 * max(1.0, abs(np.mean(olosses)))
    # nothing to train on: 0.0 (lib.rs:660-664)
    assert sz.train_from_feature_map(net, {}, files, 1, lr, 0.0, batch) == 0.0


def test_seeded_initialisation(sz):
    """SimpleNeuralNet::new (lib.rs:767-790): weights uniform in [-0.5, 0.5), biases zero; here the
    draw is a function of the seed."""
    dims = (60, 512, 256, 16)
    a = sz.SimpleNeuralNet(*dims, seed=42).get_params()
    b = sz.SimpleNeuralNet(*dims, seed=42).get_params()
    c = sz.SimpleNeuralNet(*dims, seed=43).get_params()
    for p, q in zip(a, b):
        np.testing.assert_array_equal(p, q)
    for i in (1, 3, 5):
        assert not a[i].any()
    for i in (0, 2, 4):
        w = a[i]
        assert w.min() >= -0.5 and w.max() < 0.5
        assert abs(w.mean()) < 4 * np.sqrt(1 / 12 / w.size)
        assert abs(w.var() - 1 / 12) < 0.01
        assert np.unique(w).size > 0.99 * min(w.size, 1 << 22) * 0.5
        assert not np.array_equal(w, c[i])
    # layers do not repeat each other's stream
    assert not np.array_equal(a[0].ravel()[:1000], a[2].ravel()[:1000])
