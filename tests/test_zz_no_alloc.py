"""szh_extract_dev_noalloc (one clip, nothing allocated) against szh_plan_run_dev on the same
PCM: bit-identical features, on two calls in a row, the second with a smaller clip; the size
limit is enforced; a call of zero windows is a no-op."""
import numpy as np
import pytest

import streamz_amd as sz

pytestmark = pytest.mark.gpu

MAX_WINDOWS = 16384


def _clip(seed, n_samples):
    rng = np.random.default_rng(seed)
    t = np.arange(n_samples) / 44100.0
    x = 0.2 * np.sin(2 * np.pi * (110 + 7 * seed) * t) + 0.05 * rng.standard_normal(n_samples)
    return np.clip(np.round(x * 32767), -32768, 32767).astype(np.int16)


def _plan_features(ext, d_pcm, n_samples):
    import torch
    plan = sz.Plan(ext, [0], [n_samples])
    out = torch.full((plan.total_windows, 60), float("nan"), device="cuda")
    plan.run(d_pcm.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return out.cpu().numpy()


def test_noalloc_extract_matches_plan_run_twice_second_smaller():
    import torch
    sz.lib().szh_init(0)
    ext = sz.FeatureExtractor()
    stream = torch.cuda.current_stream().cuda_stream
    # the largest clip the entry point takes, then a much smaller one (leftovers of the first call
    # stay in the shared array behind the second call's part), then an odd size
    for seed, n_win in ((1, MAX_WINDOWS), (2, 1000), (3, 49)):
        n_samples = 800 + 400 * (n_win - 1) + 123
        assert sz.window_count(n_samples) == n_win
        pcm = _clip(seed, n_samples)
        d_pcm = torch.from_numpy(pcm).cuda()
        want = _plan_features(ext, d_pcm, n_samples)
        got = torch.full((n_win, 60), float("nan"), device="cuda")
        ext.extract_dev_noalloc(d_pcm.data_ptr(), n_samples, got.data_ptr(), stream)
        torch.cuda.synchronize()
        got = got.cpu().numpy()
        assert np.isfinite(got).all()
        assert np.array_equal(got, want), (n_win, float(np.abs(got - want).max()))


def test_noalloc_extract_rejects_oversize_and_accepts_empty():
    import torch
    ext = sz.FeatureExtractor()
    n_samples = 800 + 400 * MAX_WINDOWS  # one window too many
    d_pcm = torch.zeros(n_samples, dtype=torch.int16, device="cuda")
    out = torch.zeros((MAX_WINDOWS + 1, 60), device="cuda")
    with pytest.raises(sz.StreamzError) as ei:
        ext.extract_dev_noalloc(d_pcm.data_ptr(), n_samples, out.data_ptr())
    assert ei.value.code == -1 and "limit" in str(ei.value)
    ext.extract_dev_noalloc(d_pcm.data_ptr(), 799, out.data_ptr())  # no window: nothing to do
    torch.cuda.synchronize()
    assert float(out.abs().max()) == 0.0
