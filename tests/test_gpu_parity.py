"""Parity of the HIP path (through the C ABI) with the CPU oracle and the golden fixtures.

Bars: window counts, vote counts and speaker lists exact; float32 features, probabilities
and updated parameters within 1e-4 absolute of the oracle (the tolerance the task states),
with the tighter values actually observed noted next to each check.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CLIPS = ["noise", "tones", "voiced", "quiet", "fullscale", "short799", "exact800", "len1199", "len1200"]


@pytest.fixture(scope="module")
def sz():
    import streamz_amd
    return streamz_amd


@pytest.fixture(scope="module")
def po():
    from oracle import pyoracle
    return pyoracle


@pytest.fixture(scope="module")
def ext(sz):
    return sz.FeatureExtractor()


@pytest.fixture(scope="module")
def oext(po):
    return po.Extractor()


def test_mel_filterbank_equals_oracle(ext, po):
    np.testing.assert_allclose(ext.mel_filters(), po.mel_filterbank(), rtol=0, atol=1e-9)


@pytest.mark.parametrize("clip", CLIPS)
def test_extract_matches_oracle_and_golden(ext, oext, golden_features, clip):
    pcm = golden_features[f"pcm_{clip}"]
    got = ext.extract(pcm)
    want = oext.extract(pcm, fft_mode=0)
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, atol=1e-4)
    np.testing.assert_allclose(got, golden_features[f"feat_{clip}"], atol=1e-4)


def test_extract_silence_and_clamps(ext, oext, golden_features):
    # digital silence sits on the ln(1e-12) clamp (lib.rs:309): every frame equals the oracle's
    pcm = golden_features["pcm_zeros"]
    np.testing.assert_allclose(ext.extract(pcm), oext.extract(pcm, fft_mode=0), atol=1e-4)
    # silence followed by a loud half: frames that are entirely silent or entirely loud and not
    # within two frames of the boundary are held to 1e-4; the boundary frames mix clamped and
    # unclamped energies through the deltas and are held to 5e-3
    pcm = golden_features["pcm_silence_then_loud"]
    got, want = ext.extract(pcm), oext.extract(pcm, fft_mode=0)
    silent = np.array([not pcm[i * 400:i * 400 + 800].any() for i in range(len(got))])
    near = np.convolve(silent.astype(int), np.ones(5, int), "same")
    tight = (near == 0) | (near == 5)
    np.testing.assert_allclose(got[tight], want[tight], atol=1e-4)
    np.testing.assert_allclose(got, want, atol=5e-3)


def test_extract_batch_ragged_files(ext, oext):
    rng = np.random.default_rng(3)
    lens = [0, 5, 799, 800, 2000, 48 * 400 + 400, 48 * 400 + 401, 49 * 400 + 400 + 6 * 400, 30000, 1201]
    clips = [np.clip(np.round(0.2 * rng.standard_normal(n) * 32767), -32768, 32767).astype(np.int16) for n in lens]
    outs = ext.extract_batch(clips)
    for c, o in zip(clips, outs):
        want = oext.extract(c, fft_mode=0)
        assert o.shape == want.shape
        if len(want):
            np.testing.assert_allclose(o, want, atol=1e-4)


def _random_clip(rng, n):
    kind = rng.integers(7)
    t = np.arange(n, dtype=np.float64) / 44100.0
    if kind == 0:
        x = 0.3 * rng.standard_normal(n)
    elif kind == 1:
        x = 0.5 * np.sin(2 * np.pi * rng.uniform(50, 15000) * t) + 0.2 * np.sin(2 * np.pi * rng.uniform(50, 15000) * t)
    elif kind == 2:
        x = np.zeros(n)                                   # digital silence: both clamps
    elif kind == 3:
        x = np.sign(np.sin(2 * np.pi * rng.uniform(80, 400) * t)) * rng.uniform(0.1, 1.2)   # square wave, may clip
    elif kind == 4:
        x = np.zeros(n)
        x[rng.integers(0, max(n, 1), size=max(n // 500, 1))] = rng.uniform(-1, 1)          # sparse impulses
    elif kind == 5:
        x = rng.uniform(-2, 2) / 32767.0 * rng.standard_normal(n)                          # a few LSBs of noise
    else:
        x = 0.25 * rng.standard_normal(n) + rng.uniform(-0.5, 0.5)                         # DC offset
    return np.clip(np.round(x * 32767), -32768, 32767).astype(np.int16)


@pytest.mark.parametrize("seed", [1, 2])
def test_random_clips_match_oracle(ext, oext, seed):
    """A seeded sweep over clip lengths (ragged, shorter than a window, exact multiples of the
    hop) and signal kinds (noise, tones, silence, clipping squares, impulses, LSB noise, DC)."""
    rng = np.random.default_rng(seed)
    lens = [int(v) for v in rng.integers(0, 30000, size=14)] + [799, 800, 801, 1200, 1199, 400 * 20 + 800]
    clips = [_random_clip(rng, n) for n in lens]
    got = ext.extract_batch(clips)
    for n, c, g in zip(lens, clips, got):
        want = oext.extract(c, fft_mode=0)
        assert g.shape == want.shape == (max(0, (n - 800) // 400 + 1) if n >= 800 else 0, 60)
        if g.size:
            np.testing.assert_allclose(g, want, atol=1e-4, err_msg=f"seed {seed}, {n} samples")
            np.testing.assert_allclose(ext.extract(c), g, atol=0, rtol=0)  # one clip alone: identical


def test_extract_long_clip_properties(ext, oext):
    # 60 s clip (6 614 windows): spot-check windows against the oracle and check the
    # size-independent properties of the z-score: zero mean, unit variance per frame
    rng = np.random.default_rng(11)
    t = np.arange(2_646_000) / 44100.0
    x = 0.2 * np.sin(2 * np.pi * 150 * t) + 0.05 * np.sin(2 * np.pi * 1234 * t) + 0.01 * rng.standard_normal(t.size)
    pcm = np.round(x * 32767).astype(np.int16)
    got = ext.extract(pcm)
    assert got.shape == (6614, 60)
    np.testing.assert_allclose(got.mean(axis=1), 0, atol=2e-6)
    np.testing.assert_allclose(got.var(axis=1), 1, atol=2e-5)
    head = oext.extract(pcm[:400 * 60 + 800], fft_mode=1)
    np.testing.assert_allclose(got[:58], head[:58], atol=1e-4)  # last 2 frames of `head` see its own end
    tail = oext.extract(pcm[-(400 * 60 + 800):], fft_mode=1)
    np.testing.assert_allclose(got[-58:], tail[-58:], atol=1e-4)


@pytest.mark.parametrize("name", ["tiny", "mid", "wide"])
def test_forward_and_train_batch_match_golden(sz, golden_net, name):
    g = golden_net
    dims = [int(v) for v in g[f"{name}_dims"]]
    net = sz.SimpleNeuralNet(*dims, init=[g[f"{name}_p{i}"] for i in range(6)])
    x = g[f"{name}_x"]
    np.testing.assert_allclose(net.forward(x), g[f"{name}_probs"], atol=1e-5)
    net.train_batch(x, g[f"{name}_target"], 0.05)
    for i, p in enumerate(net.get_params()):
        np.testing.assert_allclose(p, g[f"{name}_q{i}"].reshape(p.shape), atol=1e-5)


@pytest.mark.parametrize("B", [1, 8, 1000])
@pytest.mark.parametrize("C", [16, 3, 24, 64])
def test_cli_sized_net_matches_oracle(sz, po, B, C):
    rng = np.random.default_rng(100 + B + C)
    dims = (60, 512, 256, C)
    onet = po.Net(*dims)
    params = [rng.uniform(-0.5, 0.5, s).astype(np.float32) for s in onet.shapes()]
    params[1] = rng.uniform(-0.1, 0.1, 512).astype(np.float32)
    onet.set_params(params)
    net = sz.SimpleNeuralNet(*dims, init=params)
    x = rng.standard_normal((B, 60)).astype(np.float32)
    np.testing.assert_allclose(net.forward(x), onet.forward(x), atol=1e-4)
    np.testing.assert_allclose(net.embed(x), onet.embed(x), atol=1e-4)
    np.testing.assert_allclose(net.forward_embedding(x), onet.forward_embedding(x), atol=1e-3, rtol=1e-5)
    target = np.zeros(C, np.float32)
    target[1] = 1.0
    net.train_batch(x, target, 0.01)
    onet.train_batch(x, target, 0.01)
    for p, q in zip(net.get_params(), onet.get_params()):
        np.testing.assert_allclose(p, q, atol=1e-5)
    np.testing.assert_allclose(net.forward(x), onet.forward(x), atol=1e-4)


@pytest.mark.parametrize("B", [8, 300])
def test_wide_net_config(sz, po, B):
    """hidden 2048 / 2048, 256 speakers (the widest configuration): same f32 kernels, same bar."""
    rng = np.random.default_rng(2048 + B)
    dims = (60, 2048, 2048, 256)
    onet = po.Net(*dims)
    params = [rng.uniform(-0.5, 0.5, s).astype(np.float32) for s in onet.shapes()]
    onet.set_params(params)
    net = sz.SimpleNeuralNet(*dims, init=params)
    x = rng.standard_normal((B, 60)).astype(np.float32)
    want = onet.forward(x)
    np.testing.assert_allclose(net.forward(x), want, atol=1e-4)
    ids, counts = sz.identify_from_features(net, x, 0.0)
    # rows whose two best probabilities are closer than the tolerance may vote either way
    srt = np.sort(want, axis=1)
    clear = srt[:, -1] - srt[:, -2] > 1e-4
    want_counts = np.bincount(np.argmax(want[clear], axis=1), minlength=256)
    assert np.abs(counts.astype(np.int64) - want_counts).sum() <= (~clear).sum() and counts.sum() == B
    target = np.zeros(256, np.float32)
    target[200] = 1.0
    net.train_batch(x, target, 0.01)
    onet.train_batch(x, target, 0.01)
    for p, q in zip(net.get_params(), onet.get_params()):
        np.testing.assert_allclose(p, q, atol=1e-5)
    np.testing.assert_allclose(net.forward(x), onet.forward(x), atol=1e-4)


@pytest.mark.parametrize("seed", [20240611, 7, 99])
def test_random_shapes_match_oracle(sz, po, seed):
    """A seeded sweep over layer widths, class counts and batch sizes: whichever kernels the
    dispatch picks (matrix-core heads, vector head, GEMM fallback, 64- or 128-row tiles, padded
    inputs, the resident step kernel), forward and one train_batch step stay within the bar."""
    rng = np.random.default_rng(seed)
    for case in range(24):
        inp = int(rng.choice([60, 60, 60, 13, 64, 65, 100]))
        h1 = int(rng.choice([32, 64, 96, 128, 160, 256, 512]))
        h2 = int(rng.choice([16, 24, 32, 64, 128, 200, 256]))
        C = int(rng.choice([1, 2, 5, 16, 17, 31, 32, 33, 64, 65, 100]))
        B = int(rng.choice([1, 3, 8, 9, 63, 64, 65, 129, 700]))
        dims = (inp, h1, h2, C)
        onet = po.Net(*dims)
        params = [rng.uniform(-0.5, 0.5, s).astype(np.float32) for s in onet.shapes()]
        onet.set_params(params)
        net = sz.SimpleNeuralNet(*dims, init=params)
        x = rng.standard_normal((B, inp)).astype(np.float32)
        x[rng.random(B) < 0.1] = 0.0  # some all-zero rows (train_batch still counts them, lib.rs:1002-1060)
        msg = f"seed {seed} case {case}: dims {dims}, B {B}"
        np.testing.assert_allclose(net.forward(x), onet.forward(x), atol=1e-4, err_msg=msg)
        np.testing.assert_allclose(net.embed(x), onet.embed(x), atol=1e-4, err_msg=msg)
        target = np.zeros(C, np.float32)
        target[int(rng.integers(C))] = 1.0
        net.train_batch(x, target, 0.02)
        onet.train_batch(x, target, 0.02)
        for p, q in zip(net.get_params(), onet.get_params()):
            np.testing.assert_allclose(p, q, atol=2e-5, err_msg=msg)
        if inp == 60:
            perm = np.stack([rng.permutation(B) for _ in range(2)]).astype(np.uint32)
            keep = (rng.random((2, B, 60)) >= 0.2).astype(np.uint8)
            bs = int(rng.choice([1, 4, 8, 16]))
            tc = int(rng.integers(C))
            got = sz.pretrain_from_features(net, x, tc, C, 2, 0.01, 0.2, bs, perm=perm, keep=keep)
            want = onet.pretrain_from_features(x, tc, C, 2, 0.01, 0.2, bs, perm=perm, keep=keep)
            assert abs(got - want) <= 1e-4 * max(1.0, abs(want)), msg
            for p, q in zip(net.get_params(), onet.get_params()):
                np.testing.assert_allclose(p, q, atol=1e-4, err_msg=msg)


@pytest.mark.parametrize("dims", [(100, 32, 16, 3), (64, 40, 24, 2), (7, 8, 4, 3), (63, 33, 17, 33)])
def test_other_input_widths_match_oracle(sz, po, dims):
    # SimpleNeuralNet::new takes any sizes (lib.rs:767); the prepared rows are padded to a tile
    # ([x | 1 | 0...]), so widths around the 64-column tile boundary are the interesting ones
    rng = np.random.default_rng(sum(dims))
    onet = po.Net(*dims)
    params = [rng.uniform(-0.5, 0.5, s).astype(np.float32) for s in onet.shapes()]
    onet.set_params(params)
    net = sz.SimpleNeuralNet(*dims, init=params)
    for B in (1, 5, 130):
        x = rng.standard_normal((B, dims[0])).astype(np.float32)
        target = np.zeros(dims[3], np.float32)
        target[B % dims[3]] = 1.0
        np.testing.assert_allclose(net.forward(x), onet.forward(x), atol=1e-4)
        net.train_batch(x, target, 0.05)
        onet.train_batch(x, target, 0.05)
        for p, q in zip(net.get_params(), onet.get_params()):
            np.testing.assert_allclose(p, q, atol=1e-4)


def test_train_equals_train_batch_of_one_and_changes_weights(sz, golden_net):
    g = golden_net
    dims = [int(v) for v in g["tiny_dims"]]
    init = [g[f"tiny_p{i}"] for i in range(6)]
    a, b = sz.SimpleNeuralNet(*dims, init=init), sz.SimpleNeuralNet(*dims, init=init)
    a.train(g["tiny_x"][0], g["tiny_target"], 0.05)
    b.train_batch(g["tiny_x"][:1], g["tiny_target"], 0.05)
    for p, q in zip(a.get_params(), b.get_params()):
        np.testing.assert_array_equal(p, q)
    # the reference's own unit test (lib.rs:1831-1851)
    assert any(not np.array_equal(p, init[i].reshape(p.shape)) for i, p in enumerate(a.get_params()))


def test_empty_batch_is_a_no_op(sz, golden_net):
    g = golden_net
    dims = [int(v) for v in g["mid_dims"]]
    init = [g[f"mid_p{i}"] for i in range(6)]
    net = sz.SimpleNeuralNet(*dims, init=init)
    net.train_batch(np.zeros((0, 60), np.float32), g["mid_target"], 0.05)
    for i, p in enumerate(net.get_params()):
        np.testing.assert_array_equal(p, init[i].reshape(p.shape))


def test_pretrain_from_features_with_injected_rng(sz, po, golden_net):
    g = golden_net
    dims = [int(v) for v in g["loop_dims"]]
    cls, ncls, epochs, batch = [int(v) for v in g["loop_cfg"]]
    init = [g[f"loop_p{i}"] for i in range(6)]
    net = sz.SimpleNeuralNet(*dims, init=init)
    loss = sz.pretrain_from_features(net, g["loop_wins"], cls, ncls, epochs, 0.05, 0.2, batch,
                                     perm=g["loop_perm"], keep=g["loop_keep"])
    onet = po.Net(*dims)
    onet.set_params(init)
    oloss = onet.pretrain_from_features(g["loop_wins"], cls, ncls, epochs, 0.05, 0.2, batch,
                                        perm=g["loop_perm"], keep=g["loop_keep"])
    assert abs(loss - oloss) < 1e-5
    assert abs(loss - float(g["loop_loss"])) < 2e-5
    for p, q, i in zip(net.get_params(), onet.get_params(), range(6)):
        np.testing.assert_allclose(p, q, atol=1e-5)
        np.testing.assert_allclose(p, g[f"loop_q{i}"].reshape(p.shape), atol=2e-5)


def test_pretrain_cli_schedule_small_file(sz, po):
    # the CLI's schedule (batch 8, lr 0.01, dropout 0.2, main.rs:33-37, 658-666) on 100 windows
    # for 3 epochs with the same injected streams on both sides
    rng = np.random.default_rng(5)
    dims = (60, 512, 256, 2)
    onet = po.Net(*dims)
    params = [rng.uniform(-0.5, 0.5, s).astype(np.float32) for s in onet.shapes()]
    onet.set_params(params)
    net = sz.SimpleNeuralNet(*dims, init=params)
    wins = rng.standard_normal((100, 60)).astype(np.float32)
    perm = np.stack([rng.permutation(100) for _ in range(3)]).astype(np.uint32)
    keep = (rng.random((3, 100, 60)) >= 0.2).astype(np.uint8)
    loss = sz.pretrain_from_features(net, wins, 1, 2, 3, 0.01, 0.2, 8, perm=perm, keep=keep)
    oloss = onet.pretrain_from_features(wins, 1, 2, 3, 0.01, 0.2, 8, perm=perm, keep=keep)
    assert abs(loss - oloss) < 1e-4 * max(1.0, abs(oloss))
    for p, q in zip(net.get_params(), onet.get_params()):
        np.testing.assert_allclose(p, q, atol=1e-4)


@pytest.mark.parametrize("one_xcd", ["1", "0"])
@pytest.mark.parametrize("dims,batch", [((60, 64, 128, 5), 3), ((60, 512, 256, 16), 8), ((60, 96, 48, 1), 1), ((60, 64, 32, 20), 8),
                                        ((60, 1024, 256, 16), 8), ((60, 1056, 64, 3), 5)])
def test_pretrain_resident_step_kernel_edge_cases(sz, po, dims, batch, one_xcd, monkeypatch):
    # batches of <= 8 rows run as one resident kernel (several workgroups, one exchange per
    # step); here with a ragged last batch, one batch whose rows are all dropped to zero (no
    # update, lib.rs:1003-1005), single dead rows, and three epochs.  Both exchanges: the
    # workgroups on one XCD adding fixed-point partials in its L2 (up to 32 workgroups = 1024
    # hidden-1 units), and spread over the chip with flags and slots (STREAMZ_STEP_ONE_XCD=0,
    # or more than 32 workgroups)
    monkeypatch.setenv("STREAMZ_STEP_ONE_XCD", one_xcd)
    rng = np.random.default_rng(11)
    n, epochs = 4 * batch + max(1, batch // 2), 3
    onet = po.Net(*dims)
    params = [rng.uniform(-0.5, 0.5, s).astype(np.float32) for s in onet.shapes()]
    onet.set_params(params)
    net = sz.SimpleNeuralNet(*dims, init=params)
    wins = rng.standard_normal((n, 60)).astype(np.float32)
    perm = np.stack([rng.permutation(n) for _ in range(epochs)]).astype(np.uint32)
    keep = (rng.random((epochs, n, 60)) >= 0.2).astype(np.uint8)
    keep[0, batch:2 * batch] = 0      # second batch of the first epoch: nothing survives
    keep[1, 0] = 0                    # one dead row inside a live batch
    cls = min(1, dims[3] - 1)
    sz.profile_enable(True)
    sz.profile_report()
    loss = sz.pretrain_from_features(net, wins, cls, dims[3], epochs, 0.02, 0.2, batch, perm=perm, keep=keep)
    tags = sz.profile_report()
    sz.profile_enable(False)
    assert "steps" in tags and "fwd2" not in tags, tags  # the resident kernel ran, not the per-layer launches
    oloss = onet.pretrain_from_features(wins, cls, dims[3], epochs, 0.02, 0.2, batch, perm=perm, keep=keep)
    assert abs(loss - oloss) < 1e-4 * max(1.0, abs(oloss))
    for p, q in zip(net.get_params(), onet.get_params()):
        np.testing.assert_allclose(p, q, atol=1e-4)
    assert np.isfinite(net.get_flat()).all()


def test_augment_is_bit_exact(sz, po):
    # augment (lib.rs:103-116): integer output, so equality is exact; includes clamping at both
    # rails, a clip shorter than a window (shift range = len) and the identity setting
    rng = np.random.default_rng(3)
    x = rng.integers(-32768, 32768, 5000).astype(np.int16)
    x[:4] = [32767, -32768, 32767, -32768]
    for gain, level, shift in [(1.05, 0.005, 799), (0.95, 0.0049, 0), (1.0, 0.0, 123), (1.049, 0.003, 4999)]:
        noise = rng.uniform(-level, level, x.size).astype(np.float32) if level > 0 else np.zeros(x.size, np.float32)
        got = sz.augment(x, gain, level, shift, noise=noise)
        np.testing.assert_array_equal(got, po.augment(x, gain, shift, noise))
    np.testing.assert_array_equal(sz.augment(x, 1.0, 0.0, 0, noise=np.zeros(x.size, np.float32)), x)
    short = x[:300]
    nz = rng.uniform(-0.001, 0.001, 300).astype(np.float32)
    np.testing.assert_array_equal(sz.augment(short, 1.02, 0.001, 299, noise=nz), po.augment(short, 1.02, 299, nz))
    # without injected noise the draws stay inside (-level, level): |out - round-trip| is bounded
    free = sz.augment(x, 1.0, 0.004, 0, seed=9).astype(np.int64)
    assert np.abs(free - x.astype(np.int64)).max() <= int(0.004 * 32767) + 1 and (free != x).any()


def test_pretrain_network_matches_composed_oracle(sz, po, ext, oext):
    # pretrain_network (lib.rs:348-397): per epoch augment -> extract -> shuffle -> batches of 8.
    # Every draw is injected on both sides; the oracle side is composed from its augment, extract
    # and pretrain_from_features.
    rng = np.random.default_rng(17)
    n, epochs, C = 44100, 3, 4
    t = np.arange(n) / 44100.0
    pcm = np.clip(np.round(8000 * np.sin(2 * np.pi * 180 * t) + 600 * rng.standard_normal(n)), -32768, 32767).astype(np.int16)
    nw = sz.window_count(n)
    gain = rng.uniform(0.95, 1.05, epochs).astype(np.float32)
    level = rng.uniform(0.0005, 0.005, epochs).astype(np.float32)
    shift = rng.integers(0, 800, epochs).astype(np.uint64)
    noise = np.stack([rng.uniform(-l, l, n) for l in level]).astype(np.float32)
    perm = np.stack([rng.permutation(nw) for _ in range(epochs)]).astype(np.uint32)
    keep = (rng.random((epochs, nw, 60)) >= 0.2).astype(np.uint8)
    dims = (60, 64, 32, C)
    onet = po.Net(*dims)
    params = [rng.uniform(-0.5, 0.5, s).astype(np.float32) for s in onet.shapes()]
    onet.set_params(params)
    net = sz.SimpleNeuralNet(*dims, init=params)
    loss = sz.pretrain_network(net, pcm, 2, C, epochs, 0.01, 0.2, 8, ext, gain=gain, noise_level=level, shift=shift,
                               noise=noise, perm=perm, keep=keep)
    tot, cnt = 0.0, 0
    for ep in range(epochs):
        aug = po.augment(pcm, float(gain[ep]), int(shift[ep]), noise[ep])
        feats = oext.extract(aug)
        assert feats.shape == (nw, 60)
        live = int((np.where(keep[ep] != 0, feats[perm[ep]], 0.0) != 0).any(axis=1).sum())
        l = onet.pretrain_from_features(feats, 2, C, 1, 0.01, 0.2, 8, perm=perm[ep:ep + 1], keep=keep[ep:ep + 1])
        tot += l * live
        cnt += live
    assert abs(loss - tot / cnt) < 1e-4 * max(1.0, tot / cnt)
    for p, q in zip(net.get_params(), onet.get_params()):
        np.testing.assert_allclose(p, q, atol=1e-4)
    # no samples / too short for a window: nothing happens, loss 0 (lib.rs:392-396)
    before = net.get_flat()
    assert sz.pretrain_network(net, pcm[:500], 2, C, 2, 0.01, 0.2, 8, ext) == 0.0
    np.testing.assert_array_equal(net.get_flat(), before)
    # free-running draws: finite result, weights move
    assert np.isfinite(sz.pretrain_network(net, pcm, 1, C, 2, 0.01, 0.2, 8, ext, seed=5))
    assert (net.get_flat() != before).any()


def test_pretrain_no_windows_returns_zero(sz):
    net = sz.SimpleNeuralNet(60, 8, 4, 2, seed=1)
    before = net.get_flat()
    assert sz.pretrain_from_features(net, np.zeros((0, 60), np.float32), 0, 2, 3, 0.01, 0.2, 8) == 0.0
    np.testing.assert_array_equal(net.get_flat(), before)


def test_identify_last_max_wins_and_stable_order(sz, golden_net):
    g = golden_net
    dims = [int(v) for v in g["ident_dims"]]
    net = sz.SimpleNeuralNet(*dims, init=[g[f"ident_p{i}"] for i in range(6)])
    ids, counts = sz.identify_from_features(net, g["ident_feats"], 0.3)
    np.testing.assert_array_equal(counts, g["ident_counts"])
    np.testing.assert_array_equal(ids, g["ident_ids"])


def test_identify_speaker_list_end_to_end(sz, po, ext, oext):
    rng = np.random.default_rng(9)
    dims = (60, 512, 256, 5)
    onet = po.Net(*dims)
    params = [rng.uniform(-0.5, 0.5, s).astype(np.float32) for s in onet.shapes()]
    onet.set_params(params)
    net = sz.SimpleNeuralNet(*dims, init=params)
    pcm = np.clip(np.round(0.3 * rng.standard_normal(44100 * 3) * 32767), -32768, 32767).astype(np.int16)
    feats = oext.extract(pcm, fft_mode=0)
    probs = onet.forward(feats)
    top = np.sort(probs, axis=1)[:, -1]
    thr = 0.6
    # the comparison is exact unless a window's top probability sits within 1e-4 of the threshold
    assert np.abs(top - thr).min() > 1e-4
    assert sz.identify_speaker_list(net, pcm, thr, ext) == [int(v) for v in onet.identify_speaker_list(oext, pcm, thr)]
    assert sz.identify_speaker_list(net, pcm[:700], thr, ext) == []


def _ref_normalize(v):
    v = v.astype(np.float32).copy()
    ss = np.float32(0.0)
    for x in v:  # sum::<f32>() in order, lib.rs:133
        ss = np.float32(ss + np.float32(x * x))
    norm = np.float32(np.sqrt(ss))
    return v / norm if norm > 1e-6 else v


def _ref_median(e):
    n = e.shape[0]
    if n == 0:
        return np.zeros(e.shape[1], np.float32)
    srt = np.sort(e, axis=0)
    mid = n // 2
    med = (srt[mid - 1] + srt[mid]) / np.float32(2.0) if n % 2 == 0 else srt[mid]
    return _ref_normalize(med)


@pytest.mark.parametrize("dims", [(60, 64, 24, 3), (60, 128, 40, 5), (60, 512, 256, 16)])
def test_pooled_embeddings(sz, po, ext, oext, dims):
    """extract_embedding, extract_embedding_from_features, median_embedding_from_features
    (lib.rs:1418-1495): pooled on the device, against the oracle's per-window embeddings pooled
    with the reference's formulas, and the selection itself exactly against a sort of the
    device's own embeddings."""
    rng = np.random.default_rng(sum(dims))
    onet = po.Net(*dims)
    params = [rng.uniform(-0.5, 0.5, s).astype(np.float32) for s in onet.shapes()]
    onet.set_params(params)
    net = sz.SimpleNeuralNet(*dims, init=params)
    pcm = np.clip(np.round(0.3 * rng.standard_normal(400 * 120 + 800) * 32767), -32768, 32767).astype(np.int16)
    feats = oext.extract(pcm, fft_mode=0)
    for n in (feats.shape[0], feats.shape[0] - 1, 2, 1, 0):
        f = feats[:n]
        e_relu = onet.forward_embedding(f).reshape(n, dims[2])
        e_tanh = onet.embed(f).reshape(n, dims[2])
        # mean: f32 accumulation in window order, / count, unit norm (lib.rs:1451-1470)
        acc = np.zeros(dims[2], np.float32)
        for row in e_relu:
            acc = (acc + row).astype(np.float32)
        want_mean = _ref_normalize(acc / np.float32(n)) if n else acc
        np.testing.assert_allclose(sz.extract_embedding_from_features(net, f), want_mean, atol=1e-4)
        np.testing.assert_allclose(sz.median_embedding_from_features(net, f), _ref_median(e_relu), atol=1e-4)
        # the order statistic is exact: same answer as sorting the device's own embeddings
        # (ReLU leaves many exact zeros, so ties are covered)
        own = net.forward_embedding(f).reshape(n, dims[2]) if n else np.zeros((0, dims[2]), np.float32)
        np.testing.assert_allclose(sz.median_embedding_from_features(net, f), _ref_median(own), atol=1e-7)
    np.testing.assert_allclose(sz.extract_embedding(net, pcm, ext), _ref_median(onet.embed(feats).reshape(-1, dims[2])), atol=1e-4)
    own = net.embed(ext.extract(pcm))
    np.testing.assert_allclose(sz.extract_embedding(net, pcm, ext), _ref_median(own.reshape(-1, dims[2])), atol=1e-7)
    assert not sz.extract_embedding(net, pcm[:100], ext).any()  # lib.rs:1429-1431


def test_add_output_class_and_clone(sz):
    rng = np.random.default_rng(2)
    net = sz.SimpleNeuralNet(60, 16, 8, 2, seed=3)
    before = net.get_params()
    col = rng.uniform(-0.5, 0.5, 8).astype(np.float32)
    twin = net.clone()
    net.add_output_class(col)
    after = net.get_params()
    assert net.output_size() == 3 and twin.output_size() == 2
    np.testing.assert_array_equal(after[4][:, :2], before[4])  # lib.rs:803-806
    np.testing.assert_array_equal(after[4][:, 2], col)
    np.testing.assert_array_equal(after[5], np.array([before[5][0], before[5][1], 0.0], np.float32))
    for p, q in zip(twin.get_params(), before):
        np.testing.assert_array_equal(p, q)
    x = rng.standard_normal((4, 60)).astype(np.float32)
    assert net.forward(x).shape == (4, 3)
    np.testing.assert_allclose(net.forward(x).sum(axis=1), 1.0, atol=1e-6)


def test_device_resident_epoch_matches_oracle_gradient(sz, po):
    """zero_grad / accumulate / apply on device buffers equals the oracle's gradient sum and
    update with per-row labels (the multi-file accumulate step)."""
    import torch
    rng = np.random.default_rng(21)
    dims = (60, 512, 256, 16)
    onet = po.Net(*dims)
    params = [rng.uniform(-0.5, 0.5, s).astype(np.float32) for s in onet.shapes()]
    onet.set_params(params)
    net = sz.SimpleNeuralNet(*dims, init=params)
    n = 3000
    x = rng.standard_normal((n, 60)).astype(np.float32)
    labels = rng.integers(0, 16, n).astype(np.int32)
    dx = torch.from_numpy(x).cuda()
    dl = torch.from_numpy(labels).cuda()
    s = torch.cuda.current_stream().cuda_stream
    net.zero_grad_dev(s)
    net.accumulate_dev(dx.data_ptr(), dl.data_ptr(), 1700, stream=s)
    net.accumulate_dev(dx[1700:].data_ptr(), dl[1700:].data_ptr(), n - 1700, stream=s)
    torch.cuda.synchronize()
    import ctypes
    g = np.zeros(net.grad_len(), np.float32)
    hip = ctypes.CDLL("libamdhip64.so")  # the gradient vector is library-owned device memory
    assert hip.hipMemcpy(ctypes.c_void_p(g.ctypes.data), ctypes.c_void_p(net.grad_ptr()), g.nbytes, 2) == 0
    want = onet.grad(x, labels=labels)
    scale = np.abs(want).max()
    np.testing.assert_allclose(g[:-2], want, atol=2e-5 * scale)
    assert g[-1] == n
    probs = onet.forward(x)
    loss = -np.log(np.maximum(probs[np.arange(n), labels], 1e-12)).sum()
    assert abs(g[-2] - loss) < 1e-4 * loss
    net.apply_grad_dev(0.5, s)
    torch.cuda.synchronize()
    onet.apply(want, 0.5 / n)
    for p, q in zip(net.get_params(), onet.get_params()):
        np.testing.assert_allclose(p, q, atol=1e-5)
