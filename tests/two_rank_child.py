"""One rank of tests/test_gpu_two_rank.py: a sharded epoch through the library's device entry
points (zero_grad / accumulate_dev / [sum over ranks] / apply_grad_dev) on this rank's
contiguous range of files.  The ranks of the test share one GPU, where RCCL refuses to form a
communicator, so the sum of the gradient vectors goes through gloo on the host; on a node with
one GPU per rank the same step uses szh_net_allreduce_grad_dev (bench.py)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import streamz_amd as sz  # noqa: E402
from helpers import grad_view  # noqa: E402

DIMS = (60, 512, 256, 16)
LR = 0.05


def make_data():
    rng = np.random.default_rng(77)
    shapes = [(60, 512), (512,), (512, 256), (256,), (256, 16), (16,)]
    params = [rng.uniform(-0.5, 0.5, s).astype(np.float32) for s in shapes]
    files = [rng.standard_normal((n, 60)).astype(np.float32) for n in (310, 80, 500, 170, 230, 41)]
    files[1][3] = 0.0  # an all-zero window is skipped and does not count (lib.rs:607-609)
    labels = [0, 3, 1, 2, 15, 7]
    return params, files, labels


def shard(files, world, rank):
    """contiguous file ranges balanced by window count: the library's own cut (szh_shard_ranges)"""
    edges = sz.shard_ranges([len(f) for f in files], world)
    return list(range(edges[rank], edges[rank + 1]))


def epoch(net, files, labels, idx, reduce=None):
    x = np.concatenate([files[i] for i in idx])
    lab = np.concatenate([np.full(len(files[i]), labels[i], np.int32) for i in idx])
    dx, dl = torch.from_numpy(x).cuda(), torch.from_numpy(lab).cuda()
    s = torch.cuda.current_stream().cuda_stream
    net.zero_grad_dev(s)
    net.accumulate_dev(dx.data_ptr(), dl.data_ptr(), len(x), stream=s)
    g = grad_view(net)
    local = g.cpu().numpy().copy()
    if reduce is not None:
        host = g.cpu()
        reduce(host)
        g.copy_(host)
    summed = g.cpu().numpy().copy()
    net.apply_grad_dev(LR, s)
    torch.cuda.synchronize()
    return local, summed


def library_comm_main(out, rank, world, id_path):
    """the sum through the library's communicator; its id travels through a file"""
    import time
    torch.cuda.set_device(0)
    sz._capi._check(sz.lib().szh_init(0))
    if rank == 0:
        ident = sz.comm_unique_id()
        with open(id_path + ".tmp", "wb") as f:
            f.write(ident)
        os.replace(id_path + ".tmp", id_path)
    else:
        for _ in range(600):
            if os.path.exists(id_path):
                break
            time.sleep(0.1)
        ident = open(id_path, "rb").read()
    comm = sz.Comm(ident, rank, world)
    params, files, labels = make_data()
    net = sz.SimpleNeuralNet(*DIMS, init=params)
    comm.broadcast_params(net, 0, 0)

    def reduce(host):   # epoch() hands over a host copy; the library sums its own device vector in place
        comm.allreduce_grad(net, torch.cuda.current_stream().cuda_stream)
        host.copy_(grad_view(net).cpu())
    local, summed = epoch(net, files, labels, shard(files, world, rank), reduce=reduce)
    np.savez(os.path.join(out, f"rank{rank}.npz"), params=net.get_flat(), local=local, summed=summed)


def main():
    out = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if os.environ.get("TWO_RANK_LIBRARY_COMM"):
        return library_comm_main(out, rank, world, os.environ["TWO_RANK_LIBRARY_COMM"])
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    sz._capi._check(sz.lib().szh_init(0))
    params, files, labels = make_data()
    net = sz.SimpleNeuralNet(*DIMS, init=params)
    local, summed = epoch(net, files, labels, shard(files, world, rank), reduce=dist.all_reduce)
    np.savez(os.path.join(out, f"rank{rank}.npz"), params=net.get_flat(), local=local, summed=summed)
    if not os.environ.get("TWO_RANK_KEEP_NET"):   # (set: the handle outlives the process group, up to interpreter exit)
        del net
    torch.cuda.synchronize()
    dist.barrier()   # no rank leaves while its peer may still be inside the all-reduce
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
