"""The C oracle against the independent numpy/scipy golden fixtures (CPU only).

Tolerances: the fixtures are float64; the oracle works in float32 like the reference.
Features are compared on the function's real output (the z-scored 60-vector): 1e-4 absolute,
the bar the HIP path is later held to.  Near-silent and digitally silent frames sit on the
1e-12 / 1e-6 clamps of lib.rs:309 and lib.rs:337, where float32 rounding noise is amplified,
so for those only the frames whose mel energies are all well above the clamp are compared
tightly and the rest for shape and finiteness.
"""
import numpy as np
import pytest

from oracle import pyoracle as po

CLIPS = ["noise", "tones", "voiced", "quiet", "silence_then_loud", "fullscale",
         "short799", "exact800", "len1199", "len1200", "zeros"]


def test_window_count_formula():
    # lib.rs:288-291, 317
    for n, w in [(0, 0), (799, 0), (800, 1), (1199, 1), (1200, 2), (1201, 2), (220500, 550), (2646000, 6614)]:
        assert po.window_count(n) == w
    assert po.window_count(15_876_000_000) == 39_689_999  # 100 h stream


def test_mel_filterbank_matches_independent_formula(golden_features):
    w = po.mel_filterbank()
    g = golden_features["mel_filterbank"]
    assert w.shape == (26, 401)
    np.testing.assert_allclose(w, g, rtol=0, atol=2e-7)
    nz = (g > 0).sum(axis=1)
    # the Nyquist bin of the last filter is 4e-19 in float64 and exactly 0 in float32
    assert abs(int((g > 1e-12).sum()) - int((w > 1e-12).sum())) == 0
    assert nz.min() >= 4 and nz.max() <= 110


@pytest.mark.parametrize("clip", CLIPS)
@pytest.mark.parametrize("fft_mode", [0, 1])
def test_extract_matches_golden(golden_features, clip, fft_mode):
    ext = po.Extractor()
    pcm = golden_features[f"pcm_{clip}"]
    want = golden_features[f"feat_{clip}"]
    wbase = golden_features[f"base_{clip}"]
    got, base = ext.extract(pcm, fft_mode=fft_mode, return_base=True)
    assert got.shape == want.shape == (po.window_count(pcm.size), 60)
    assert np.isfinite(got).all()
    if clip == "zeros":
        # all mel energies clamp to ln(1e-12); c0 = 26 ln(1e-12), deltas 0
        np.testing.assert_allclose(base[:, 0], 26 * np.log(1e-12), rtol=1e-6)
        np.testing.assert_allclose(got, want, atol=2e-3)
        return
    if clip == "silence_then_loud":
        silent = np.array([not pcm[i * 400:i * 400 + 800].any() for i in range(len(got))])
        ok = ~silent
        # frames within 2 of a silent frame mix clamped and unclamped values through the deltas
        near = np.convolve(silent.astype(int), np.ones(5, int), "same") > 0
        tight = ok & ~near
        assert tight.sum() >= 3
        np.testing.assert_allclose(got[tight], want[tight], atol=1e-4)
        np.testing.assert_allclose(got, want, atol=5e-3)
        return
    np.testing.assert_allclose(base, wbase, atol=2e-3, rtol=1e-5)
    np.testing.assert_allclose(got, want, atol=1e-4)


@pytest.mark.parametrize("name", ["tiny", "mid", "wide"])
def test_forward_and_train_batch_match_golden(golden_net, name):
    g = golden_net
    dims = [int(v) for v in g[f"{name}_dims"]]
    net = po.Net(*dims)
    net.set_params([g[f"{name}_p{i}"] for i in range(6)])
    x = g[f"{name}_x"]
    np.testing.assert_allclose(net.forward(x), g[f"{name}_probs"], atol=2e-6)
    net.train_batch(x, g[f"{name}_target"], 0.05)
    for i, p in enumerate(net.get_params()):
        np.testing.assert_allclose(p, g[f"{name}_q{i}"].reshape(p.shape), atol=3e-6)


def test_train_is_train_batch_of_one(golden_net):
    g = golden_net
    dims = [int(v) for v in g["tiny_dims"]]
    a, b = po.Net(*dims), po.Net(*dims)
    for n in (a, b):
        n.set_params([g[f"tiny_p{i}"] for i in range(6)])
    a.train(g["tiny_x"][0], g["tiny_target"], 0.05)
    b.train_batch(g["tiny_x"][:1], g["tiny_target"], 0.05)
    for p, q in zip(a.get_params(), b.get_params()):
        np.testing.assert_array_equal(p, q)
    # the reference's own unit test (lib.rs:1831-1851): some weight changed
    assert any(not np.array_equal(p, g[f"tiny_p{i}"].reshape(p.shape)) for i, p in enumerate(a.get_params()))


def test_empty_batch_is_a_no_op(golden_net):
    g = golden_net
    dims = [int(v) for v in g["mid_dims"]]
    net = po.Net(*dims)
    net.set_params([g[f"mid_p{i}"] for i in range(6)])
    net.train_batch(np.zeros((0, 60), np.float32), g["mid_target"], 0.05)
    for i, p in enumerate(net.get_params()):
        np.testing.assert_array_equal(p, g[f"mid_p{i}"].reshape(p.shape))


def test_pretrain_from_features_with_injected_rng(golden_net):
    g = golden_net
    dims = [int(v) for v in g["loop_dims"]]
    cls, ncls, epochs, batch = [int(v) for v in g["loop_cfg"]]
    net = po.Net(*dims)
    net.set_params([g[f"loop_p{i}"] for i in range(6)])
    loss = net.pretrain_from_features(g["loop_wins"], cls, ncls, epochs, 0.05, 0.2, batch,
                                      perm=g["loop_perm"], keep=g["loop_keep"])
    assert abs(loss - float(g["loop_loss"])) < 2e-5
    for i, p in enumerate(net.get_params()):
        np.testing.assert_allclose(p, g[f"loop_q{i}"].reshape(p.shape), atol=1e-5)


def test_pretrain_from_features_no_windows_returns_zero():
    net = po.Net(60, 8, 4, 2)
    assert net.pretrain_from_features(np.zeros((0, 60), np.float32), 0, 2, 3, 0.01, 0.2, 8) == 0.0


def test_identify_last_max_wins_and_stable_order(golden_net):
    g = golden_net
    dims = [int(v) for v in g["ident_dims"]]
    net = po.Net(*dims)
    net.set_params([g[f"ident_p{i}"] for i in range(6)])
    ids, counts = net.identify_from_features(g["ident_feats"], 0.3)
    np.testing.assert_array_equal(counts, g["ident_counts"])
    np.testing.assert_array_equal(ids, g["ident_ids"])
    assert counts[2] == 0  # classes 2 and 3 tie exactly: the later index takes every such vote


def test_embedding_variants():
    rng = np.random.default_rng(1)
    net = po.Net(60, 16, 8, 3)
    p = [rng.uniform(-0.5, 0.5, s).astype(np.float32) for s in net.shapes()]
    net.set_params(p)
    x = rng.standard_normal((5, 60)).astype(np.float32)
    h1 = np.maximum(x.astype(np.float64) @ p[0] + p[1], 0)
    np.testing.assert_allclose(net.embed(x), np.tanh(h1 @ p[2] + p[3]), atol=2e-6)
    np.testing.assert_allclose(net.forward_embedding(x), np.maximum(h1 @ p[2] + p[3], 0), atol=5e-6)
