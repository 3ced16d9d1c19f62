"""The N>1 training step on CPU (gloo, world size 2): every rank sums gradients over its own
shard of files, the [gradient | loss | live rows] vector is all-reduced, every rank applies
lr / global_live_rows.  The compute on each rank is the CPU oracle (no GPU here); the result
must equal one process doing the same step over all files.  It checks the sharding arithmetic and
the sum-then-update algebra of that protocol only: no product code runs here.  The product's own
path (szh_net_*_dev, the library's RCCL communicator, bench.py's multi-rank branch) is exercised by
tests/test_gpu_two_rank.py on a GPU."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import pyoracle as po

DIMS = (60, 24, 12, 4)


def _data():
    rng = np.random.default_rng(77)
    params = [rng.uniform(-0.5, 0.5, s).astype(np.float32)
              for s in [(60, 24), (24,), (24, 12), (12,), (12, 4), (4,)]]
    files = [rng.standard_normal((n, 60)).astype(np.float32) for n in (31, 8, 50, 17, 23)]
    files[1][3] = 0.0  # an all-zero window is skipped and does not count (lib.rs:607-609)
    labels = [0, 3, 1, 2, 3]
    return params, files, labels


def _shard_grad(net, files, labels, idx):
    g = np.zeros(net.param_count() + 2, np.float64)
    for i in idx:
        x = files[i]
        live = np.any(x != 0.0, axis=1)
        xs = x[live]
        g[:-2] += net.grad(xs, labels=np.full(len(xs), labels[i], np.int32))
        p = net.forward(xs)
        g[-2] += -np.log(np.maximum(p[:, labels[i]], 1e-12)).sum()
        g[-1] += len(xs)
    return g


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    params, files, labels = _data()
    net = po.Net(*DIMS)
    net.set_params(params)
    # contiguous file ranges balanced by window count
    counts = np.array([len(f) for f in files])
    cut = int(np.searchsorted(np.cumsum(counts), counts.sum() / world)) + 1
    mine = list(range(0, cut)) if rank == 0 else list(range(cut, len(files)))
    g = torch.from_numpy(_shard_grad(net, files, labels, mine))
    dist.all_reduce(g)  # sum
    g = g.numpy()
    if g[-1] > 0:
        net.apply(g[:-2].astype(np.float32), 0.05 / g[-1])
    np.save(os.path.join(out, f"rank{rank}.npy"), np.concatenate([p.ravel() for p in net.get_params()] + [g[-2:]]))
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_equals_single_process(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npy"), np.load(tmp_path / "rank1.npy")
    np.testing.assert_array_equal(r0, r1)  # identical update on every rank
    params, files, labels = _data()
    net = po.Net(*DIMS)
    net.set_params(params)
    g = _shard_grad(net, files, labels, range(len(files)))
    assert g[-1] == sum(len(f) for f in files) - 1
    net.apply(g[:-2].astype(np.float32), 0.05 / g[-1])
    want = np.concatenate([p.ravel() for p in net.get_params()])
    np.testing.assert_allclose(r0[:-2], want, atol=2e-6)
    np.testing.assert_allclose(r0[-2:], g[-2:], rtol=1e-6)
