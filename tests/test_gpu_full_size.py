"""Parity of the benchmarked launches at their own size: 6 614 000 rows at 60-512-256-16 in ONE
szh_net_accumulate_dev call (h1 alone is 13.5 GB: element indices beyond 2^31, 25 836 row tiles,
weight-gradient K ranges of 51 672 rows), and szh_plan_run_dev over 1 000 clips of 60 s.

The oracle cannot walk 6.6 M rows in a test, so the inputs are periodic: rows (clips) repeat a block
that the oracle does walk, every row's result must not depend on where the row sits, and sums over
the rows are R times the block's sums plus the tail's."""
import numpy as np
import pytest

from helpers import dropout_keep, grad_view

pytestmark = pytest.mark.gpu

DIMS = (60, 512, 256, 16)
N_ROWS = 6_614_000          # 1 000 clips x 6 614 windows: what bench.py hands to one call
BLOCK = 140_003             # rows of the repeated block (ragged against every tile size)


@pytest.fixture(scope="module")
def sz():
    import streamz_amd
    return streamz_amd


@pytest.fixture(scope="module")
def po():
    from oracle import pyoracle
    return pyoracle


def _t_mix64(z):
    """rng.h mix64 on int64 tensors (two's-complement wrap-around, logical shifts by masking)"""
    def c(v):
        return v - (1 << 64) if v >= (1 << 63) else v
    z = z + c(0x9E3779B97F4A7C15)
    z = (z ^ ((z >> 30) & ((1 << 34) - 1))) * c(0xBF58476D1CE4E5B9)
    z = (z ^ ((z >> 27) & ((1 << 37) - 1))) * c(0x94D049BB133111EB)
    return z ^ ((z >> 31) & ((1 << 33) - 1))


def _t_dropout_keep(torch, seed, row0, n_rows, p, dev):
    """keep[r][j] of szh_net_accumulate_dev(dropout=p, seed) for rows row0 .. row0 + n_rows, on the device"""
    r = torch.arange(row0, row0 + n_rows, dtype=torch.int64, device=dev)[:, None]
    j = torch.arange(60, dtype=torch.int64, device=dev)[None, :]
    k = 0xD1342543DE82EF95 - (1 << 64)
    h = _t_mix64(_t_mix64((r * k) ^ seed) + j)
    u = ((h >> 40) & 0xFFFFFF).to(torch.float32) * (1.0 / 16777216.0)
    return ~(u < p)


def _oracle_sums(po, onet, x, labels, pieces=64, threads=16):
    """oracle gradient (f32 sums of short pieces, added in f64), loss sum and live rows of x"""
    from concurrent.futures import ThreadPoolExecutor
    live = np.any(x != 0.0, axis=1)
    xl, ll = x[live], labels[live]
    parts = [p for p in np.array_split(np.arange(len(xl)), pieces) if len(p)]
    with ThreadPoolExecutor(threads) as pool:
        gs = list(pool.map(lambda ix: onet.grad(xl[ix[0]:ix[-1] + 1], labels=ll[ix[0]:ix[-1] + 1]).astype(np.float64), parts))
        ps = list(pool.map(lambda ix: onet.forward(xl[ix[0]:ix[-1] + 1]), parts))
    probs = np.concatenate(ps)
    loss = -np.log(np.maximum(probs[np.arange(len(xl)), ll], 1e-12).astype(np.float64)).sum()
    return np.sum(gs, axis=0), loss, int(live.sum())


def _near_zero_units(x, params, eps=1.5e-6):
    w1, b1 = params[0].astype(np.float64), params[1].astype(np.float64)
    near = np.zeros(DIMS[1], bool)
    for r0 in range(0, len(x), 20000):
        near |= (np.abs(x[r0:r0 + 20000].astype(np.float64) @ w1 + b1) < eps).any(axis=0)
    loose = np.zeros(61 * DIMS[1] + DIMS[1] * DIMS[2] + DIMS[2] + (DIMS[2] + 1) * DIMS[3], bool)
    loose[:60 * DIMS[1]].reshape(60, DIMS[1])[:, near] = True
    loose[60 * DIMS[1]:61 * DIMS[1]][near] = True
    off_w2 = 61 * DIMS[1]
    loose[off_w2:off_w2 + DIMS[1] * DIMS[2]].reshape(DIMS[1], DIMS[2])[near, :] = True
    return loose


@pytest.mark.parametrize("gemm", ["f32", "bf16x3"])
def test_accumulate_dev_at_the_benchmarked_row_count(sz, po, gemm):
    """zero_grad / accumulate_dev over 6 614 000 rows in one call (lib.rs:1013-1045 per row), without
    dropout against the oracle by periodicity, with seeded dropout (lib.rs:119-129, 607-609) against
    the same kernels fed block by block with rows masked outside the library, then forward."""
    import torch
    dev = torch.device("cuda")
    rng = np.random.default_rng(61)
    onet = po.Net(*DIMS)
    params = [rng.uniform(-0.5, 0.5, s).astype(np.float32) for s in onet.shapes()]
    onet.set_params(params)
    net = sz.SimpleNeuralNet(*DIMS, init=params)
    net.set_gemm(gemm)
    R, tail = divmod(N_ROWS, BLOCK)
    assert R == 47 and tail == 33_859
    xb = rng.standard_normal((BLOCK, 60)).astype(np.float32)
    xb[5] = 0.0            # an all-zero row does not count (lib.rs:607-609), in the tail as well
    xb[77, 2:] = 0.0       # two non-zero inputs: dropout removes both on some rows
    xb[BLOCK - 1] = 0.0
    lb = rng.integers(0, DIMS[3], BLOCK).astype(np.int32)
    dxb, dlb = torch.from_numpy(xb).to(dev), torch.from_numpy(lb).to(dev)
    x = torch.cat([dxb.repeat(R, 1), dxb[:tail]]).contiguous()
    labels = torch.cat([dlb.repeat(R), dlb[:tail]]).contiguous()
    assert x.shape == (N_ROWS, 60)
    s = torch.cuda.current_stream().cuda_stream

    def run(xp, lp, n, dropout=0.0, seed=0):
        net.zero_grad_dev(s)
        net.accumulate_dev(xp.data_ptr(), lp.data_ptr(), n, dropout, seed, s)
        torch.cuda.synchronize()
        return grad_view(net).cpu().numpy().astype(np.float64)

    # ---- no dropout: one call over everything
    g = run(x, labels, N_ROWS)
    if gemm == "f32":
        assert sz.lib().szh_net_param_count(net._h) + 2 == g.size
    g_block, loss_block, live_block = _oracle_sums(po, onet, xb, lb)
    g_tail, loss_tail, live_tail = _oracle_sums(po, onet, xb[:tail], lb[:tail])
    want = R * g_block + g_tail
    scale = np.abs(want).max()
    assert g[-1] == R * live_block + live_tail == N_ROWS - 2 * R - 1   # rows 5 and BLOCK - 1 of every block, row 5 of the tail
    assert abs(g[-2] - (R * loss_block + loss_tail)) < 2e-5 * (R * loss_block + loss_tail)
    loose = _near_zero_units(xb, params)
    assert loose[60 * DIMS[1]:61 * DIMS[1]].sum() < 64
    err = np.abs(g[:-2] - want)
    assert err[~loose].max() < 2e-5 * scale, err[~loose].max() / scale
    # a first-layer ReLU decision that differs from the oracle's (a pre-activation within an f32 rounding
    # of zero) is repeated in every copy of the block
    assert err[loose].max() < 2e-4 * scale, err[loose].max() / scale
    # ... and against the same kernels on one block and on the tail (the same decisions: tight)
    gb, gt = run(dxb, dlb, BLOCK), run(dxb, dlb, tail)
    assert gb[-1] == live_block and gt[-1] == live_tail
    np.testing.assert_allclose(g[:-2], R * gb[:-2] + gt[:-2], atol=3e-6 * scale)

    # ---- seeded dropout at full size
    p, seed = 0.2, 1003
    gd = run(x, labels, N_ROWS, p, seed)
    acc = np.zeros_like(gd)
    kept = 0
    for r0 in range(0, N_ROWS, BLOCK):
        n = min(BLOCK, N_ROWS - r0)
        keep = _t_dropout_keep(torch, seed, r0, n, p, dev)
        if r0 + n == N_ROWS:   # the device-side restatement of the generator agrees with the numpy one
            np.testing.assert_array_equal(keep[-2000:].cpu().numpy(), dropout_keep(seed, 2000, p, row0=N_ROWS - 2000))
        kept += int(keep.sum())
        xm = torch.where(keep, x[r0:r0 + n], torch.zeros((), device=dev)).contiguous()
        acc += run(xm, labels[r0:r0 + n], n)
        del xm, keep
    assert 0.7995 < kept / (60.0 * N_ROWS) < 0.8005
    assert gd[-1] == acc[-1] and gd[-1] < g[-1]       # row 77's two inputs both dropped in about 4 % of the blocks
    assert abs(gd[-2] - acc[-2]) < 1e-6 * acc[-2]
    np.testing.assert_allclose(gd[:-2], acc[:-2], atol=3e-6 * np.abs(acc[:-2]).max())
    assert np.abs(gd[:-2] - g[:-2]).max() > 1e-3 * scale
    # the masked rows of the last (ragged) piece against the oracle
    r0 = N_ROWS - 5000
    keep = _t_dropout_keep(torch, seed, r0, 5000, p, dev)
    xm = torch.where(keep, x[r0:], torch.zeros((), device=dev)).contiguous()
    gm = run(xm, labels[r0:], 5000)
    wm, _, live_m = _oracle_sums(po, onet, xm.cpu().numpy(), labels[r0:].cpu().numpy(), pieces=16)
    assert gm[-1] == live_m
    np.testing.assert_allclose(gm[:-2], wm, atol=2e-5 * np.abs(wm).max())

    # ---- forward over all rows in one call: every copy of the block gives the block's bits
    probs = torch.empty((N_ROWS, DIMS[3]), dtype=torch.float32, device=dev)
    net.forward_dev(x.data_ptr(), N_ROWS, probs.data_ptr(), s)
    torch.cuda.synchronize()
    first = probs[:BLOCK]
    assert torch.equal(probs[:R * BLOCK].view(R, BLOCK, DIMS[3]), first.unsqueeze(0).expand(R, BLOCK, DIMS[3]))
    assert torch.equal(probs[R * BLOCK:], first[:tail])
    np.testing.assert_allclose(probs[-600:].cpu().numpy(), onet.forward(xb[tail - 600:tail]), atol=2e-5)
    np.testing.assert_allclose(first[:600].cpu().numpy(), onet.forward(xb[:600]), atol=2e-5)


def test_plan_run_over_one_thousand_minute_clips(sz, po):
    """szh_plan_run_dev over 1 000 x 60 s (6 614 000 windows, 5.3 GB of PCM, one launch per kernel):
    clip i repeats one of seven base clips, so its features must equal that clip's bit for bit; three base
    clips against the oracle (lib.rs:279-345); every row z-scored (lib.rs:323-342)."""
    import torch
    dev = torch.device("cuda")
    rng = np.random.default_rng(67)
    n = 60 * 44100
    n_al = (n + 7) // 8 * 8
    t = np.arange(n) / 44100.0
    base = []
    for b in range(7):
        f0 = 95.0 + 17.0 * b
        sig = sum(a * np.sin(2 * np.pi * f0 * (h + 1) * t + 0.3 * h * b) for h, a in enumerate((1.0, 0.6, 0.4, 0.25)))
        sig = 0.25 * sig / 2.25 + 0.008 * rng.standard_normal(n)
        base.append(np.clip(np.round(sig * 32767.0), -32768, 32767).astype(np.int16))
    n_clips = 1000
    pcm = torch.zeros(n_clips * n_al, dtype=torch.int16, device=dev)
    dbase = [torch.from_numpy(b).to(dev) for b in base]
    for i in range(n_clips):
        pcm[i * n_al:i * n_al + n] = dbase[i % 7]
    ext = sz.FeatureExtractor()
    plan = sz.Plan(ext, np.arange(n_clips, dtype=np.uint64) * n_al, np.full(n_clips, n, np.uint64))
    per = sz.window_count(n)
    assert per == 6614 and plan.total_windows == N_ROWS
    np.testing.assert_array_equal(plan.window_offsets, np.arange(n_clips + 1, dtype=np.uint64) * per)
    feats = torch.full((N_ROWS, 60), float("nan"), dtype=torch.float32, device=dev)
    plan.run(pcm.data_ptr(), feats.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    f = feats.view(n_clips, per, 60)
    for b in range(7):
        same = f[b::7]
        assert torch.equal(same, same[:1].expand_as(same)), b
    oext = po.Extractor()
    for b in (0, 3, 6):
        want = oext.extract(base[b], fft_mode=0)
        np.testing.assert_allclose(f[b].cpu().numpy(), want, atol=1e-4)
        np.testing.assert_allclose(f[987 + b].cpu().numpy(), want, atol=1e-4)   # 987 = 7 * 141: late copies of the same clips
    assert bool(torch.isfinite(feats).all())
    assert float(feats.mean(dim=1).abs().max()) < 1e-5
    assert float((feats.var(dim=1, unbiased=False) - 1.0).abs().max()) < 1e-4
