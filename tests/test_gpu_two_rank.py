"""The sharded epoch (files split over ranks, one sum of the gradient vector, identical update)
through the product's own device entry points, on the one GPU a test box has:
  * two nets in one process, each accumulating half of the rows, their gradient vectors
    (loss / live-row tail included) summed as the all-reduce does;
  * two processes sharing device 0, gloo standing in for RCCL (which refuses two ranks on one GPU), and the
    same two processes through the library's own communicator on its host transport;
  * the library's RCCL communicator with a single rank (dlopen, ncclCommInitRank, in-place
    ncclAllReduce / ncclBroadcast on library-owned memory);
  * bench.py's multi-rank branch, rehearsed with two ranks on one GPU and with RCCL at one rank.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _oracle_step(po, params, files, labels, dims, lr):
    onet = po.Net(*dims)
    onet.set_params(params)
    x = np.concatenate(files)
    lab = np.concatenate([np.full(len(f), l, np.int32) for f, l in zip(files, labels)])
    live = np.any(x != 0.0, axis=1)
    g = onet.grad(x[live], labels=lab[live])
    probs = onet.forward(x[live])
    loss = -np.log(np.maximum(probs[np.arange(live.sum()), lab[live]], 1e-12)).sum()
    onet.apply(g, lr / live.sum())
    return g, loss, int(live.sum()), np.concatenate([p.ravel() for p in onet.get_params()])


def test_two_nets_summed_like_the_allreduce():
    import torch
    import streamz_amd as sz
    from oracle import pyoracle as po
    from helpers import grad_view
    import two_rank_child as tr

    params, files, labels = tr.make_data()
    nets = [sz.SimpleNeuralNet(*tr.DIMS, init=params) for _ in range(2)]
    s = torch.cuda.current_stream().cuda_stream
    views = []
    for rank, net in enumerate(nets):
        idx = tr.shard(files, 2, rank)
        x = np.concatenate([files[i] for i in idx])
        lab = np.concatenate([np.full(len(files[i]), labels[i], np.int32) for i in idx])
        dx, dl = torch.from_numpy(x).cuda(), torch.from_numpy(lab).cuda()
        net.zero_grad_dev(s)
        net.accumulate_dev(dx.data_ptr(), dl.data_ptr(), len(x), stream=s)
        torch.cuda.synchronize()
        views.append(grad_view(net))
    lives = [float(v[-1]) for v in views]
    assert lives[0] != lives[1] and sum(lives) == sum(len(f) for f in files) - 1  # the all-zero row sits in one shard
    total = views[0] + views[1]
    for v in views:
        v.copy_(total)
    for net in nets:
        net.apply_grad_dev(tr.LR, s)
    torch.cuda.synchronize()
    p0, p1 = nets[0].get_flat(), nets[1].get_flat()
    np.testing.assert_array_equal(p0, p1)
    # one net over all rows, and the oracle
    whole = sz.SimpleNeuralNet(*tr.DIMS, init=params)
    _, summed = tr.epoch(whole, files, labels, range(len(files)))
    g, loss, live, want = _oracle_step(po, params, files, labels, tr.DIMS, tr.LR)
    scale = np.abs(g).max()
    np.testing.assert_allclose(total.cpu().numpy()[:-2], g, atol=2e-5 * scale)
    np.testing.assert_allclose(summed[:-2], g, atol=2e-5 * scale)
    assert summed[-1] == live == sum(lives)
    assert abs(float(total[-2]) - loss) < 1e-4 * loss
    np.testing.assert_allclose(p0, whole.get_flat(), atol=1e-5)
    np.testing.assert_allclose(p0, want, atol=1e-5)


def test_two_rank_processes_on_one_gpu(tmp_path):
    import streamz_amd as sz
    from oracle import pyoracle as po
    import two_rank_child as tr

    for attempt in range(3):  # the port is free when it is picked, not necessarily when the ranks bind it
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2")
        procs = [subprocess.Popen([sys.executable, "-X", "faulthandler", os.path.join(ROOT, "tests", "two_rank_child.py"), str(tmp_path)],
                                  env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT, text=True) for r in range(2)]
        outs = [p.communicate(timeout=300)[0] for p in procs]
        if attempt < 2 and any(p.returncode != 0 and "EADDRINUSE" in o for p, o in zip(procs, outs)):
            continue
        break
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-4000:]
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    np.testing.assert_array_equal(r0["summed"], r1["summed"])
    np.testing.assert_array_equal(r0["params"], r1["params"])  # identical update on every rank, bit for bit
    assert r0["local"][-1] != r1["local"][-1]  # the shards differ in live rows
    np.testing.assert_allclose(r0["local"] + r1["local"], r0["summed"], rtol=1e-6, atol=1e-6)
    params, files, labels = tr.make_data()
    whole = sz.SimpleNeuralNet(*tr.DIMS, init=params)
    _, summed = tr.epoch(whole, files, labels, range(len(files)))
    g, loss, live, want = _oracle_step(po, params, files, labels, tr.DIMS, tr.LR)
    scale = np.abs(g).max()
    np.testing.assert_allclose(r0["summed"][:-2], g, atol=2e-5 * scale)
    assert r0["summed"][-1] == live == summed[-1]
    assert abs(r0["summed"][-2] - loss) < 1e-4 * loss
    np.testing.assert_allclose(r0["params"], whole.get_flat(), atol=1e-5)
    np.testing.assert_allclose(r0["params"], want, atol=1e-5)


def test_two_rank_processes_through_the_library_host_transport(tmp_path):
    """The same two processes with the sum done by szh_net_allreduce_grad_dev on a communicator whose id
    names a shared-memory segment (STREAMZ_COMM=host): rank 0 == rank 1 bit for bit, and both equal the
    gloo run's result."""
    import two_rank_child as tr
    env = dict(os.environ, WORLD_SIZE="2", STREAMZ_COMM="host", TWO_RANK_LIBRARY_COMM=str(tmp_path / "comm.id"))
    procs = [subprocess.Popen([sys.executable, "-X", "faulthandler", os.path.join(ROOT, "tests", "two_rank_child.py"), str(tmp_path)],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-4000:]
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    np.testing.assert_array_equal(r0["summed"], r1["summed"])
    np.testing.assert_array_equal(r0["params"], r1["params"])
    np.testing.assert_array_equal(r0["local"] + r1["local"], r0["summed"])   # rank order: slot 0 + slot 1, in f32
    import streamz_amd as sz
    params, files, labels = tr.make_data()
    whole = sz.SimpleNeuralNet(*tr.DIMS, init=params)
    _, summed = tr.epoch(whole, files, labels, range(len(files)))
    assert r0["summed"][-1] == summed[-1]
    np.testing.assert_allclose(r0["summed"][:-2], summed[:-2], atol=2e-6 * np.abs(summed[:-2]).max())
    np.testing.assert_allclose(r0["params"], whole.get_flat(), atol=1e-6)
    assert not os.path.exists("/dev/shm/" + open(tmp_path / "comm.id", "rb").read()[8:].split(b"\0")[0].decode().lstrip("/"))


def test_rccl_communicator_single_rank_in_place():
    import torch
    import streamz_amd as sz
    from helpers import grad_view

    sz._capi._check(sz.lib().szh_init(0))
    rng = np.random.default_rng(5)
    net = sz.SimpleNeuralNet(60, 64, 32, 5, seed=3)
    x = torch.from_numpy(rng.standard_normal((300, 60)).astype(np.float32)).cuda()
    lab = torch.from_numpy(rng.integers(0, 5, 300).astype(np.int32)).cuda()
    s = torch.cuda.current_stream().cuda_stream
    net.zero_grad_dev(s)
    net.accumulate_dev(x.data_ptr(), lab.data_ptr(), 300, stream=s)
    torch.cuda.synchronize()
    before = grad_view(net).cpu().numpy().copy()
    comm = sz.Comm(sz.comm_unique_id(), 0, 1)  # a real RCCL communicator of one rank
    assert (comm.rank, comm.world) == (0, 1)
    comm.allreduce_grad(net, s)
    params = net.get_flat()
    comm.broadcast_params(net, 0, s)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(grad_view(net).cpu().numpy(), before)  # the sum over one rank
    np.testing.assert_array_equal(net.get_flat(), params)
    local = sz.Comm(None, 0, 1)  # no id: a communicator whose calls do nothing
    local.allreduce_grad(net, s)
    with pytest.raises(sz.StreamzError):
        sz.Comm(None, 0, 2)
    with pytest.raises(sz.StreamzError):
        comm.broadcast_params(net, 3, s)


def _bench_line(cmd, env):
    """cmd / env may be callables of a rendezvous port: the launch is repeated with another port when
    the one picked was taken between picking and binding"""
    for attempt in range(3):
        port = _free_port()
        c = cmd(port) if callable(cmd) else cmd
        e = env(port) if callable(env) else env
        r = subprocess.run(c, env=e, capture_output=True, text=True, timeout=600, cwd=ROOT)
        if r.returncode != 0 and "EADDRINUSE" in r.stdout + r.stderr and attempt < 2:
            continue
        break
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_multi_rank_branch():
    # (no dropout: its mask is keyed by a rank's own row numbers, so only without it do two ranks repeat one rank's sums)
    small = ["--clips", "6", "--seconds", "5", "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline", "--dropout", "0"]
    one = _bench_line([sys.executable, "bench.py", "--gpus", "1"] + small, dict(os.environ))
    assert one["scaling"] == "strong" and one["config"]["windows_by_rank"] == [one["config"]["windows_total"]]
    # two ranks on the one GPU (the library's communicator on its host transport); strong scaling: the same
    # six clips cut in two by window count, so the summed loss is that of the single-rank run
    torchrun = lambda port: [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",  # noqa: E731
                             "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2", "--share-gpu"]
    two = _bench_line(lambda port: torchrun(port) + small, dict(os.environ))
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and "host transport" in two["config"]["collective"]
    assert two["config"]["windows_total"] == one["config"]["windows_total"]
    assert two["config"]["windows_by_rank"] == [one["config"]["windows_total"] // 2] * 2
    assert abs(two["mean_loss_last_step"] - one["mean_loss_last_step"]) < 1e-4 * one["mean_loss_last_step"]
    # weak scaling: every rank brings its own six clips
    weak = _bench_line(lambda port: torchrun(port) + small + ["--scaling", "weak"], dict(os.environ))
    assert weak["scaling"] == "weak" and weak["config"]["windows_by_rank"] == [one["config"]["windows_total"]] * 2
    assert weak["mean_loss_last_step"] is not None and np.isfinite(weak["mean_loss_last_step"])
    # the RCCL branch with a single rank: same numbers as the plain single-GPU run
    rccl = _bench_line([sys.executable, "bench.py", "--gpus", "1"] + small,
                       lambda port: dict(os.environ, BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                                         RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"))
    # (a line that carries collective_note did NOT run the library's collective: that counts as a failure here)
    assert "library RCCL" in rccl["config"]["collective"] and "collective_note" not in rccl["config"]
    assert rccl["mean_loss_last_step"] == one["mean_loss_last_step"]
    # the stand-in for a library communicator that does not come up: torch's RCCL group on the same vector
    alt = _bench_line([sys.executable, "bench.py", "--gpus", "1"] + small,
                      lambda port: dict(os.environ, BENCH_FORCE_DIST="1", BENCH_FORCE_TORCH_RCCL="1", MASTER_ADDR="127.0.0.1",
                                        MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"))
    assert "torch.distributed RCCL group" in alt["config"]["collective_note"] and "fallback" in alt["config"]["collective"]
    assert alt["mean_loss_last_step"] == one["mean_loss_last_step"]


def test_train_epochs_sharded_is_the_device_epoch_loop_and_matches_the_oracle():
    """szh_net_train_epochs_sharded on one rank (no communicator, and a one-rank RCCL communicator):
    bit for bit the zero_grad / accumulate_dev / apply_grad_dev loop, and the oracle's full-batch
    epochs within tolerance (body of lib.rs:1013-1059 over all rows, one update per epoch)."""
    import torch
    import streamz_amd as sz
    from oracle import pyoracle as po
    from helpers import grad_view
    dims, epochs, lr = (60, 512, 256, 4), 3, 0.05
    rng = np.random.default_rng(5)
    onet = po.Net(*dims)
    params = [rng.uniform(-0.5, 0.5, s).astype(np.float32) for s in onet.shapes()]
    x = rng.standard_normal((1500, 60)).astype(np.float32)
    x[7] = 0.0
    labels = rng.integers(0, 4, len(x)).astype(np.int32)
    a = sz.SimpleNeuralNet(*dims, init=params)
    loss_a = a.train_epochs_sharded(x, labels, epochs, lr, dropout=0.2, seed=77)
    # the same steps through the device entry points
    b = sz.SimpleNeuralNet(*dims, init=params)
    dx, dl = torch.from_numpy(x).cuda(), torch.from_numpy(labels).cuda()
    for ep in range(epochs):
        b.zero_grad_dev(0)
        b.accumulate_dev(dx.data_ptr(), dl.data_ptr(), len(x), 0.2, 77 + ep, 0)
        torch.cuda.synchronize()
        tail = grad_view(b)[-2:].cpu().numpy()
        b.apply_grad_dev(lr, 0)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(a.get_flat(), b.get_flat())
    assert loss_a == np.float32(tail[0]) / np.float32(tail[1])
    # a communicator of one rank: the all-reduce leaves the sums as they are
    c = sz.SimpleNeuralNet(*dims, init=params)
    comm = sz.Comm(sz.comm_unique_id(), 0, 1)
    loss_c = c.train_epochs_sharded(x, labels, epochs, lr, dropout=0.2, seed=77, comm=comm)
    np.testing.assert_array_equal(c.get_flat(), a.get_flat())
    assert loss_c == loss_a
    # without dropout against the oracle
    d = sz.SimpleNeuralNet(*dims, init=params)
    loss_d = d.train_epochs_sharded(x, labels, epochs, lr)
    onet.set_params(params)
    live = np.any(x != 0.0, axis=1)
    for ep in range(epochs):
        probs = onet.forward(x[live])
        oloss = -np.log(np.maximum(probs[np.arange(live.sum()), labels[live]], 1e-12)).mean()
        onet.apply(onet.grad(x[live], labels=labels[live]), lr / live.sum())
    want = np.concatenate([p.ravel() for p in onet.get_params()])
    np.testing.assert_allclose(d.get_flat(), want, atol=2e-5)
    assert abs(loss_d - oloss) < 1e-4 * max(1.0, oloss)
