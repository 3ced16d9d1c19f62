"""Known-answer checks of the oracle's speaker-embedding bookkeeping (normalize, average_vectors,
cosine_similarity, identify_speaker_from_embedding, the pooled embeddings and
compute_speaker_embeddings; lib.rs:132-159, 1450-1599).  CPU only."""
import numpy as np
import pytest

from oracle import pyoracle as po


def test_normalize_and_average_vectors():
    np.testing.assert_allclose(po.normalize([3.0, 4.0]), [0.6, 0.8], rtol=1e-7)
    tiny = np.array([5e-7, 0.0], np.float32)
    np.testing.assert_array_equal(po.normalize(tiny), tiny)          # norm <= 1e-6: left alone (lib.rs:135)
    assert po.average_vectors([]).size == 0                          # lib.rs:145-147
    avg = po.average_vectors([[1.0, 0.0], [0.0, 1.0]])
    np.testing.assert_allclose(avg, [2 ** -0.5, 2 ** -0.5], rtol=1e-6)
    np.testing.assert_allclose(po.average_vectors([[2.0, 0.0]]), [1.0, 0.0])


def test_cosine_similarity():
    assert po.cosine_similarity([1, 0], [0, 1]) == 0.0
    assert abs(po.cosine_similarity([1, 2, 3], [2, 4, 6]) - 1.0) < 1e-6
    assert abs(po.cosine_similarity([1, 0], [-1, 0]) + 1.0) < 1e-6
    assert po.cosine_similarity([0, 0], [1, 1]) == 0.0               # a zero norm gives 0 (lib.rs:1536-1537)


def test_identify_speaker_from_embedding_thresholds():
    cents = {0: np.array([1.0, 0.0], np.float32), 3: np.array([0.0, 1.0], np.float32)}
    emb = po.normalize([0.8, 0.6])                                   # similarity 0.8 to speaker 0, 0.6 to speaker 3
    # fewer than 20 speakers: the threshold is relaxed to 0.7 x (lib.rs:1519-1523)
    assert po.identify_speaker_from_embedding(emb, cents, 0.8) == 0          # 0.8 > 0.56
    assert po.identify_speaker_from_embedding(emb, cents, 1.2) is None       # 0.8 < 0.84
    assert po.identify_speaker_from_embedding(po.normalize([0.1, 0.9]), cents, 0.8) == 3
    assert po.identify_speaker_from_embedding(emb, {}, 0.5) is None          # nothing to match: usize::MAX
    many = {i: po.normalize(np.eye(32, dtype=np.float32)[i] + 0.01) for i in range(20)}
    probe = po.normalize(np.eye(32, dtype=np.float32)[5] * 0.75 + np.eye(32, dtype=np.float32)[31] * 0.66)
    sim = float(po.cosine_similarity(probe, many[5]))
    assert 0.56 < sim < 0.8
    assert po.identify_speaker_from_embedding(probe, many, 0.8) is None      # 20 speakers: the full threshold
    fewer = {i: many[i] for i in range(19)}
    assert po.identify_speaker_from_embedding(probe, fewer, 0.8) == 5


class _FakeNet:
    """forward_embedding = the first four features: makes the pooled embeddings checkable by hand"""
    h2 = 4

    def forward_embedding(self, x):
        return np.asarray(x, np.float32).reshape(-1, 60)[:, :4].copy()


def _rows(vals):
    x = np.zeros((len(vals), 60), np.float32)
    x[:, :4] = vals
    return x


def test_pooled_embeddings_and_speaker_statistics():
    net = _FakeNet()
    f = _rows([[1, 0, 0, 0], [3, 0, 4, 0], [2, 0, 2, 0]])
    np.testing.assert_allclose(po.extract_embedding_from_features(net, f), po.normalize([2.0, 0.0, 2.0, 0.0]), rtol=1e-6)
    np.testing.assert_allclose(po.median_embedding_from_features(net, f), po.normalize([2.0, 0.0, 2.0, 0.0]), rtol=1e-6)
    even = _rows([[1, 0, 0, 0], [3, 0, 4, 0], [2, 0, 2, 0], [10, 0, 0, 8]])   # medians: (2+3)/2, 0, (0+2)/2, 0
    np.testing.assert_allclose(po.median_embedding_from_features(net, even), po.normalize([2.5, 0.0, 1.0, 0.0]), rtol=1e-6)
    assert not po.median_embedding_from_features(net, np.zeros((0, 60), np.float32)).any()
    a, b = _rows([[1, 0, 0, 0]]), _rows([[0, 1, 0, 0]])
    stats = po.compute_speaker_embeddings(net, [[a, b], [], [a]], 4)
    mean, mean_sim, std_sim = stats[0]
    np.testing.assert_allclose(mean, [2 ** -0.5, 2 ** -0.5, 0, 0], rtol=1e-6)
    assert abs(mean_sim - 2 ** -0.5) < 1e-6 and std_sim < 1e-6
    assert not stats[1][0].any() and stats[1][1] == 0.0 and stats[1][2] == 0.0   # no files: zeros (lib.rs:1569-1570)
    np.testing.assert_allclose(stats[2][0], [1, 0, 0, 0])
    assert abs(stats[2][1] - 1.0) < 1e-6 and stats[2][2] == 0.0
    c = _rows([[1, 1, 0, 0]])
    m2, ms2, sd2 = po.compute_speaker_embeddings(net, [[a, c]], 4)[0]
    sims = [float(po.cosine_similarity(po.normalize(v), m2)) for v in ([1, 0, 0, 0], [1, 1, 0, 0])]
    assert abs(ms2 - np.mean(sims)) < 1e-6 and abs(sd2 - np.std(sims)) < 1e-6
