"""Shared pieces of the GPU tests: a host-side restatement of the library's counter-based
generator (streamz_amd/csrc/rng.h), so that the dropout mask a device call draws from a seed can
be rebuilt and fed to the oracle, and zero-copy views of library-owned device memory."""
import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def mix64(z):
    z = np.asarray(z, np.uint64)
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(seed, a, b):
    """rng.h uniform01(seed, a, b): 24-bit uniform in [0, 1) as float32; a, b broadcast."""
    a = np.asarray(a, np.uint64)
    b = np.asarray(b, np.uint64)
    with np.errstate(over="ignore"):
        h = mix64(mix64(np.uint64(seed) ^ (a * np.uint64(0xD1342543DE82EF95))) + b)
    return (h >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def dropout_keep(seed, n_rows, p, n_in=60, row0=0):
    """keep[r][j] of szh_net_accumulate_dev(dropout=p, seed): element (r, j) is zeroed when
    uniform01(seed, row0 + r, j) < p (prep_rows_kernel)."""
    r = np.arange(row0, row0 + n_rows, dtype=np.uint64)[:, None]
    j = np.arange(n_in, dtype=np.uint64)[None, :]
    return ~(uniform01(seed, r, j) < np.float32(p))


class _DevView:
    def __init__(self, ptr, n, typestr="<f4"):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def grad_view(net):
    """torch tensor aliasing the net's device gradient vector [params | loss | live rows]."""
    import torch
    return torch.as_tensor(_DevView(net.grad_ptr(), net.grad_len()), device="cuda")
