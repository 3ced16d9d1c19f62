"""Long host recordings and large batches of host clips travel through pinned pieces filled by helper threads
(from 32 MB on; STREAMZ_COPY_THREADS=0 keeps the plain pageable copies): both ways must give the same results."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sz():
    import streamz_amd
    return streamz_amd


def noise_pcm(rng, n):
    return np.clip(np.round(0.3 * rng.standard_normal(n) * 32767), -32768, 32767).astype(np.int16)


def test_extract_batch_staged_copies_equal_plain_ones(sz, monkeypatch):
    """44 MB of PCM in clips of ragged lengths (one too short for a window, one a single window, pieces of 4 MB
    ending inside clips), features returned through the same ring."""
    ext = sz.FeatureExtractor()
    rng = np.random.default_rng(12)
    lengths = [60 * 44100, 799, 800, 47 * 44100 + 13, 60 * 44100, 3 * 44100 + 399, 90 * 44100, 60 * 44100 + 1,
               120 * 44100, 2_097_152, 60 * 44100]
    assert sum(lengths) * 2 >= 32 << 20
    clips = [noise_pcm(rng, n) for n in lengths]
    staged = ext.extract_batch(clips)
    monkeypatch.setenv("STREAMZ_COPY_THREADS", "0")
    plain = ext.extract_batch(clips)
    for c, a, b in zip(clips, staged, plain):
        assert a.shape == (sz.window_count(c.size), 60)
        assert np.array_equal(a, b)
    # and each clip on its own (a small call: plain copies)
    for i in (0, 2, 5):
        assert np.array_equal(ext.extract(clips[i]), staged[i])
    # twice in a row: the second call finds nothing left over from the first
    monkeypatch.delenv("STREAMZ_COPY_THREADS")
    again = ext.extract_batch(clips[::-1])[::-1]
    for a, b in zip(again, plain):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("tile", ["8192", "3000", None])
def test_identify_from_a_long_host_recording_staged_and_plain(sz, monkeypatch, tile):
    """37 MB of host PCM, tile by tile (pieces span tiles of 6.5 MB and of 2.4 MB) and in one piece (the default
    tile): the order of speakers equals the device-resident call's, the probability sums the plain copies'."""
    import torch
    rng = np.random.default_rng(5)
    net = sz.SimpleNeuralNet(60, 512, 256, 16, seed=11)
    ext = sz.FeatureExtractor()
    pcm = noise_pcm(rng, 420 * 44100 + 77)
    assert pcm.nbytes >= 32 << 20
    d = torch.from_numpy(pcm).cuda()
    torch.cuda.synchronize()
    want_ids, want_counts = sz.identify_speaker_list_dev(net, d.data_ptr(), pcm.size, 0.1, ext)
    assert 0 < want_counts.sum() <= sz.window_count(pcm.size) and len(want_ids) > 1
    if tile:
        monkeypatch.setenv("STREAMZ_TILE_WINDOWS", tile)
    ids = sz.identify_speaker_list(net, pcm, 0.1, ext)
    best, sums = sz.identify_speaker(net, pcm, ext, return_sums=True)
    monkeypatch.setenv("STREAMZ_COPY_THREADS", "0")
    ids0 = sz.identify_speaker_list(net, pcm, 0.1, ext)
    best0, sums0 = sz.identify_speaker(net, pcm, ext, return_sums=True)
    assert ids == ids0 == want_ids
    assert best == best0
    np.testing.assert_allclose(sums, sums0, rtol=1e-6)
