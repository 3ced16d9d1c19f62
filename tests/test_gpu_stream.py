"""identify_speaker_list over long recordings: the tile-by-tile path (HIP graph replay, two
staging buffers) against the oracle at small sizes, against the one-piece path at ragged
lengths, and at the full 100 h size through the periodicity of a repeated block."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sz():
    import streamz_amd
    return streamz_amd


@pytest.fixture(scope="module")
def po():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="module")
def nets(sz, po):
    rng = np.random.default_rng(41)
    dims = (60, 512, 256, 16)
    onet = po.Net(*dims)
    params = [rng.uniform(-0.5, 0.5, s).astype(np.float32) for s in onet.shapes()]
    # small output weights keep the top probability away from 1.0, so that a threshold bites
    params[4] *= 0.3
    onet.set_params(params)
    return sz.SimpleNeuralNet(*dims, init=params), onet


def noise_pcm(rng, n):
    return np.clip(np.round(0.3 * rng.standard_normal(n) * 32767), -32768, 32767).astype(np.int16)


def dev_counts(sz, net, ext, pcm, thr, tile):
    import torch
    d = torch.from_numpy(pcm).cuda()
    torch.cuda.synchronize()
    return sz.identify_speaker_list_dev(net, d.data_ptr(), pcm.size, thr, ext, tile_windows=tile)


def test_tiles_match_oracle(sz, po, nets):
    net, onet = nets
    ext, oext = sz.FeatureExtractor(), po.Extractor()
    rng = np.random.default_rng(7)
    pcm = noise_pcm(rng, 400 * 549 + 800 + 123)  # 550 windows and a dropped partial one
    feats = oext.extract(pcm, fft_mode=0)
    assert feats.shape[0] == 550
    probs = onet.forward(feats)
    srt = np.sort(probs, axis=1)
    thr = float(np.median(srt[:, -1]))
    # exact unless a decision sits within the fp tolerance (1e-4) of a tie or of the threshold
    keep = (np.abs(srt[:, -1] - thr) > 1e-4) & (srt[:, -1] - srt[:, -2] > 1e-4)
    assert keep.all(), "pick another seed: a window sits on a decision boundary"
    want_ids, want_counts = onet.identify_from_features(feats, thr)
    assert 0 < want_counts.sum() < 550
    for tile in (4, 5, 47, 48, 49, 100, 273, 546, 547, 548, 549, 550, 8192):
        ids, counts = dev_counts(sz, net, ext, pcm, thr, tile)
        np.testing.assert_array_equal(counts, want_counts, err_msg=f"tile {tile}")
        assert ids == [int(v) for v in want_ids], tile
    # host PCM: staged tile by tile through the two device buffers
    old = os.environ.get("STREAMZ_TILE_WINDOWS")
    try:
        for tile in ("4", "64", "137"):
            os.environ["STREAMZ_TILE_WINDOWS"] = tile
            assert sz.identify_speaker_list(net, pcm, thr, ext) == [int(v) for v in want_ids], tile
    finally:
        if old is None:
            os.environ.pop("STREAMZ_TILE_WINDOWS", None)
        else:
            os.environ["STREAMZ_TILE_WINDOWS"] = old


def test_ragged_lengths_equal_one_piece(sz, nets):
    """Window counts around every tile boundary case give the votes of the one-piece path."""
    net, _ = nets
    ext = sz.FeatureExtractor()
    rng = np.random.default_rng(6)
    B = 16
    full = noise_pcm(rng, 400 * 80 + 800)
    for nw in (1, 2, 3, 4, B, B + 1, B + 2, B + 3, B + 4, 2 * B + 2, 2 * B + 3, 2 * B + 4, 3 * B + 2, 3 * B + 3, 5 * B - 1):
        for extra in (0, 399):
            pcm = full[: 400 * (nw - 1) + 800 + extra]
            assert sz.window_count(pcm.size) == nw
            _, one = dev_counts(sz, net, ext, pcm, 0.0, 1 << 20)
            _, tiled = dev_counts(sz, net, ext, pcm, 0.0, B)
            assert one.sum() == nw  # threshold 0: every window votes exactly once
            np.testing.assert_array_equal(tiled, one, err_msg=f"{nw} windows (+{extra} samples)")
    # too short for a window: nobody is identified (lib.rs:288-291)
    ids, counts = dev_counts(sz, net, ext, full[:799], 0.0, B)
    assert ids == [] and counts.sum() == 0


@pytest.mark.parametrize("lanes", ["1", "2", "3", "4"])
def test_votes_do_not_depend_on_the_tiles_in_flight(sz, nets, lanes, monkeypatch):
    """Interior tiles of device-resident PCM run on several streams at once (each with its own plan, feature
    rows, workspace rows and graph); the votes equal the one-piece path's for every number of lanes,
    including window counts that leave a lane without a tile in the last round."""
    net, _ = nets
    ext = sz.FeatureExtractor()
    rng = np.random.default_rng(16)
    monkeypatch.setenv("STREAMZ_TILE_LANES", lanes)
    B = 32
    full = noise_pcm(rng, 400 * 14 * B + 800)
    for nw in (4 * B + 3, 7 * B + 5, 13 * B + 2):
        pcm = full[: 400 * (nw - 1) + 800]
        _, one = dev_counts(sz, net, ext, pcm, 0.1, 1 << 20)
        _, tiled = dev_counts(sz, net, ext, pcm, 0.1, B)
        np.testing.assert_array_equal(tiled, one, err_msg=f"{nw} windows, {lanes} lanes")


def test_profiled_run_gives_the_same_votes(sz, nets):
    net, _ = nets
    ext = sz.FeatureExtractor()
    pcm = noise_pcm(np.random.default_rng(8), 400 * 300 + 800)
    _, plain = dev_counts(sz, net, ext, pcm, 0.2, 32)
    sz.profile_enable(True)
    try:
        _, prof = dev_counts(sz, net, ext, pcm, 0.2, 32)
        sz.profile_report()
    finally:
        sz.profile_enable(False)
    np.testing.assert_array_equal(prof, plain)


def test_100h_stream_by_periodicity(sz, nets):
    """The inference-only configuration at full size: 100 h of 44.1 kHz PCM (39 689 999 windows)
    in tiles of 8192 windows and in the library's default tiles.  The recording repeats a block of P windows' worth of samples, so
    every interior window w has the features of window w + P, and the votes of n windows obey
    c(R*P + k) = c(P + k) + (R - 1) * (c(2P + k) - c(P + k)) exactly."""
    import torch
    net, _ = nets
    ext = sz.FeatureExtractor()
    n_samples = 100 * 3600 * 44100
    nw = sz.window_count(n_samples)
    assert nw == 39_689_999
    P = 3000
    R, k = divmod(nw, P)
    assert k >= 4
    block = torch.from_numpy(noise_pcm(np.random.default_rng(12), 400 * P)).cuda()
    reps = (n_samples + block.numel() - 1) // block.numel()
    stream = block.repeat(reps)[:n_samples].contiguous()
    assert stream.numel() == n_samples
    torch.cuda.synchronize()
    thr = 0.3

    def counts(n_windows, tile):
        ns = 400 * (n_windows - 1) + 800
        return sz.identify_speaker_list_dev(net, stream.data_ptr(), ns, thr, ext, tile_windows=tile)[1].astype(np.int64)

    c1 = counts(P + k, 1 << 20)
    c2 = counts(2 * P + k, 1 << 20)
    assert 0 < c1.sum() < P + k
    ids, full = sz.identify_speaker_list_dev(net, stream.data_ptr(), n_samples, thr, ext, tile_windows=8192)
    np.testing.assert_array_equal(full.astype(np.int64), c1 + (R - 1) * (c2 - c1))
    order = sorted((i for i in range(16) if full[i] > 0), key=lambda i: (-int(full[i]), i))
    assert ids == order
    # the library's own tile size (tile_windows 0: larger tiles, two rows blocks per workgroup of the vote kernel)
    ids0, full0 = sz.identify_speaker_list_dev(net, stream.data_ptr(), n_samples, thr, ext)
    np.testing.assert_array_equal(full0, full)
    assert ids0 == ids
    # every window is counted exactly once when the threshold lets all of them vote
    _, everyone = sz.identify_speaker_list_dev(net, stream.data_ptr(), n_samples, 0.0, ext, tile_windows=8192)
    assert int(everyone.sum()) == nw
    del stream, block
    torch.cuda.empty_cache()  # 64 GB of recording go back to the driver, not to the caching allocator


@pytest.mark.parametrize("classes", [16, 40, 70, 1])
def test_identify_speaker_and_threshold_siblings(sz, po, classes):
    """identify_speaker / _with_threshold / _with_threshold_feats (lib.rs:1285-1377): per-class sums
    of the window probabilities, arg-max, confidence = sum / windows against the threshold."""
    rng = np.random.default_rng(300 + classes)
    dims = (60, 128, 64, classes)
    onet = po.Net(*dims)
    params = [rng.uniform(-0.5, 0.5, s).astype(np.float32) for s in onet.shapes()]
    params[4] *= 0.3
    onet.set_params(params)
    net = sz.SimpleNeuralNet(*dims, init=params)
    ext, oext = sz.FeatureExtractor(), po.Extractor()
    pcm = noise_pcm(rng, 400 * 299 + 800 + 17)
    feats = oext.extract(pcm, fft_mode=0)
    nw = feats.shape[0]
    want_best, want_sums = onet.identify_speaker_feats(feats)
    best, sums = sz.identify_speaker(net, pcm, ext, return_sums=True)
    # every window's probabilities are within 1e-4 of the oracle's, so the sums within 1e-4 * windows
    np.testing.assert_allclose(sums, want_sums, atol=1e-4 * nw)
    np.testing.assert_allclose(sums, want_sums, atol=5e-3)  # in practice far tighter
    np.testing.assert_allclose(sums.sum(), nw, rtol=1e-5)
    if classes > 1:
        srt = np.sort(want_sums)
        assert srt[-1] - srt[-2] > 0.05, "pick another seed: the two best sums are too close"
    assert best == want_best
    conf = float(want_sums[want_best]) / nw
    for thr in (conf - 0.01, conf + 0.01, 0.0, 1.1):
        want = onet.identify_speaker_with_threshold_feats(feats, thr)
        assert sz.identify_speaker_with_threshold(net, pcm, thr, ext) == want, thr
        got, fsums = sz.identify_speaker_with_threshold_feats(net, feats, thr, return_sums=True)
        assert got == want, thr
        np.testing.assert_allclose(fsums, want_sums, atol=5e-3)
    if classes == 1:
        assert onet.identify_speaker_with_threshold_feats(feats, 0.0) is None  # lib.rs:1316-1318
    # no window: all sums are 0.0 and the last maximum wins (lib.rs:1298-1302); None (lib.rs:1329-1331)
    assert sz.identify_speaker(net, pcm[:500], ext) == classes - 1 == onet.identify_speaker_feats(feats[:0])[0]
    assert sz.identify_speaker_with_threshold(net, pcm[:500], 0.0, ext) is None
    assert sz.identify_speaker_with_threshold_feats(net, feats[:0], 0.0) is None
    # tile by tile: the same sums up to the order of the f64 additions
    old = os.environ.get("STREAMZ_TILE_WINDOWS")
    try:
        os.environ["STREAMZ_TILE_WINDOWS"] = "64"
        tbest, tsums = sz.identify_speaker(net, pcm, ext, return_sums=True)
    finally:
        if old is None:
            os.environ.pop("STREAMZ_TILE_WINDOWS", None)
        else:
            os.environ["STREAMZ_TILE_WINDOWS"] = old
    assert tbest == best
    np.testing.assert_allclose(tsums, sums, rtol=1e-6)
