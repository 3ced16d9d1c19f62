"""Pins the CPU oracle (and, on a GPU, the HIP path) against outputs of the REAL reference when they
are present under tests/golden/reference/ (made by tools/reference_fixtures/ on a machine with
cargo: the mel filterbank of mel_filter 0.1.1, FeatureExtractor::extract of every golden clip,
SimpleNeuralNet::forward and one train_batch).  The directory cannot be produced where this
repository is developed (no Rust toolchain), so until somebody adds it these tests skip, and
parity stays pinned only by the independent numpy formulation of tests/golden/make_golden.py."""
import glob
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "tests", "golden", "reference")

needs_reference = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "reference_mel.npy")),
                                     reason="tests/golden/reference/ not present (needs the reference built with cargo)")


def _clips():
    return sorted(os.path.basename(p)[len("reference_feat_"):-4] for p in glob.glob(os.path.join(REF, "reference_feat_*.npy")))


def _model(path):
    z = np.load(path)
    c = int(z["num_speakers"][0])
    w3 = np.stack([z[f"w3_{k + 1}"] for k in range(c)], axis=1)
    b3 = np.array([z[f"b3_{k + 1}"][0] for k in range(c)], np.float32)
    return [z["w1"], z["b1"], z["w2"], z["b2"], w3, b3]


def _net_case(golden_net):
    dims = tuple(int(v) for v in golden_net["mid_dims"])
    params = [golden_net[f"mid_p{k}"] for k in range(6)]
    return dims, params, golden_net["mid_x"].reshape(-1, dims[0]), golden_net["mid_target"]


@needs_reference
def test_oracle_filterbank_is_the_reference_filterbank():
    from oracle import pyoracle as po
    np.testing.assert_allclose(po.mel_filterbank(), np.load(os.path.join(REF, "reference_mel.npy")), rtol=0, atol=1e-7)


@needs_reference
def test_oracle_extract_matches_reference(golden_features):
    from oracle import pyoracle as po
    ext = po.Extractor()
    assert _clips(), "no reference_feat_*.npy"
    for clip in _clips():
        want = np.load(os.path.join(REF, f"reference_feat_{clip}.npy"))
        got = ext.extract(golden_features[f"pcm_{clip}"], fft_mode=0)
        assert got.shape == want.shape, clip
        np.testing.assert_allclose(got, want, atol=1e-4, err_msg=clip)


@needs_reference
def test_oracle_forward_and_train_batch_match_reference(golden_net):
    from oracle import pyoracle as po
    dims, params, x, target = _net_case(golden_net)
    net = po.Net(*dims)
    net.set_params(params)
    np.testing.assert_allclose(net.forward(x), np.load(os.path.join(REF, "reference_probs.npy")), atol=1e-5)
    net.train_batch(x, target, 0.05)
    for got, want in zip(net.get_params(), _model(os.path.join(REF, "reference_model_after.npz"))):
        np.testing.assert_allclose(got, want.reshape(got.shape), atol=1e-5)


@needs_reference
@pytest.mark.gpu
def test_hip_path_matches_reference(golden_features, golden_net):
    import streamz_amd as sz
    ext = sz.FeatureExtractor()
    np.testing.assert_allclose(ext.mel_filters(), np.load(os.path.join(REF, "reference_mel.npy")), rtol=0, atol=1e-7)
    for clip in _clips():
        want = np.load(os.path.join(REF, f"reference_feat_{clip}.npy"))
        got = ext.extract(golden_features[f"pcm_{clip}"])
        assert got.shape == want.shape, clip
        np.testing.assert_allclose(got, want, atol=1e-4, err_msg=clip)
    dims, params, x, target = _net_case(golden_net)
    net = sz.SimpleNeuralNet(*dims, init=params)
    np.testing.assert_allclose(net.forward(x), np.load(os.path.join(REF, "reference_probs.npy")), atol=1e-4)
    net.train_batch(x, target, 0.05)
    for got, want in zip(net.get_params(), _model(os.path.join(REF, "reference_model_after.npz"))):
        np.testing.assert_allclose(got, want.reshape(got.shape), atol=1e-4)
