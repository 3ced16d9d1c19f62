"""The one-kernel forward + vote (identify_speaker_list's loop body, lib.rs:880-891, 1391-1401) against the oracle
over the shapes and row counts that select its variants: one or two blocks of 32 rows per workgroup, one or two
blocks of 32 classes, fewer first-layer tiles than waves, ragged last tiles, fewer than 60 inputs."""
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


def make_nets(sz, po, dims, seed):
    rng = np.random.default_rng(seed)
    onet = po.Net(*dims)
    params = [rng.uniform(-0.5, 0.5, s).astype(np.float32) for s in onet.shapes()]
    params[4] *= 0.3   # keeps the top probability away from 1.0, so that a threshold bites
    onet.set_params(params)
    return sz.SimpleNeuralNet(*dims, init=params), onet


def expected_votes(probs, thr):
    """Votes per class, and how many rows sit within the fp tolerance of a tie or of the threshold."""
    c = probs.shape[1]
    top = probs.max(axis=1)
    # the last maximum wins (lib.rs:1393-1397)
    win = c - 1 - np.argmax(probs[:, ::-1], axis=1)
    counts = np.bincount(win[top >= thr], minlength=c).astype(np.int64)
    srt = np.sort(probs, axis=1)
    gap = srt[:, -1] - srt[:, -2] if c > 1 else np.ones(len(probs))
    doubtful = int(((np.abs(top - thr) <= 1e-4) | (gap <= 1e-4)).sum())
    return counts, doubtful


@pytest.mark.parametrize("dims", [(60, 512, 256, 16), (60, 128, 64, 5), (60, 256, 128, 33), (60, 1024, 256, 64),
                                  (40, 128, 32, 3), (60, 384, 96, 16)])
def test_vote_dev_matches_oracle_at_small_and_ragged_row_counts(sz, po, dims):
    import torch
    net, onet = make_nets(sz, po, dims, 900 + dims[1] + dims[3])
    rng = np.random.default_rng(dims[2])
    n_max = 4097
    x = rng.standard_normal((n_max, dims[0])).astype(np.float32)
    probs = onet.forward(x)
    thr = float(np.median(probs.max(axis=1)))
    dx = torch.from_numpy(x).cuda()
    s = torch.cuda.current_stream().cuda_stream
    for n in (1, 31, 32, 33, 64, 100, 2049, n_max):
        want, doubtful = expected_votes(probs[:n], thr)
        counts = torch.zeros(dims[3], dtype=torch.int64, device="cuda")
        net.vote_dev(dx.data_ptr(), n, thr, counts.data_ptr(), s)
        torch.cuda.synchronize()
        got = counts.cpu().numpy()
        assert np.abs(got - want).sum() <= 2 * doubtful, (dims, n, got, want, doubtful)
        # votes add to what is already there
        net.vote_dev(dx.data_ptr(), n, 0.0, counts.data_ptr(), s)
        torch.cuda.synchronize()
        assert int(counts.sum().item()) == int(got.sum()) + n


@pytest.mark.parametrize("dims", [(60, 512, 256, 16), (60, 256, 128, 40)])
def test_vote_and_probability_sums_with_two_row_blocks_per_workgroup(sz, po, dims):
    """Enough rows that a workgroup takes 64 at a time (from 128 rows per compute unit on), a ragged last tile."""
    import torch
    net, onet = make_nets(sz, po, dims, 77 + dims[3])
    rng = np.random.default_rng(5)
    n = 128 * 256 + 4097 + 37
    x = rng.standard_normal((n, 60)).astype(np.float32)
    probs = onet.forward(x)
    thr = float(np.median(probs.max(axis=1)))
    want, doubtful = expected_votes(probs, thr)
    dx = torch.from_numpy(x).cuda()
    counts = torch.zeros(dims[3], dtype=torch.int64, device="cuda")
    net.vote_dev(dx.data_ptr(), n, thr, counts.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = counts.cpu().numpy()
    assert np.abs(got - want).sum() <= 2 * doubtful, (got, want, doubtful)
    # the same rows through the host entry points: votes, and the per-class probability sums (f64 on the device)
    _, hcounts = sz.identify_from_features(net, x, thr)
    assert np.abs(hcounts.astype(np.int64) - want).sum() <= 2 * doubtful
    best, sums = sz.identify_speaker_with_threshold_feats(net, x, 0.0, return_sums=True)
    want_sums = probs.astype(np.float64).sum(axis=0)
    np.testing.assert_allclose(sums, want_sums, rtol=2e-5, atol=1e-4 * 16)
    assert best == int(len(want_sums) - 1 - np.argmax(want_sums[::-1]))
