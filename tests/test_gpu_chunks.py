"""The chunked gradient accumulation with small chunks: a call of 1 000 rows split into chunks of
256 rows (three full ones and a ragged tail) must give the same gradient as the oracle.  The chunk
size is read when the library is loaded, so the check runs in a child process with
STREAMZ_CHUNK_ROWS set; the split-K slabs then accumulate over several chunks (first chunk
stores, later ones add), the head kernels keep per-workgroup slabs over chunks, and the tail
chunk uses fewer K ranges and workgroups than the first."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import ctypes, sys
import numpy as np
import torch
sys.path.insert(0, %(root)r)
import streamz_amd as sz
from oracle import pyoracle as po

def grad_of(net):
    g = np.zeros(net.grad_len(), np.float32)
    hip = ctypes.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(ctypes.c_void_p(g.ctypes.data), ctypes.c_void_p(net.grad_ptr()), g.nbytes, 2) == 0
    return g

rng = np.random.default_rng(4)
# MFMA head, VALU head, GEMM output path, MFMA head + dW3 GEMM with 2, 3 and 4 blocks of 16 classes
for dims in %(dims)s:
    onet = po.Net(*dims)
    params = [rng.uniform(-0.5, 0.5, s).astype(np.float32) for s in onet.shapes()]
    onet.set_params(params)
    net = sz.SimpleNeuralNet(*dims, init=params)
    n = %(n)d
    x = rng.standard_normal((n, 60)).astype(np.float32)
    x[17] = 0.0                                   # an all-zero row does not count (lib.rs:607-609)
    labels = rng.integers(0, dims[3], n).astype(np.int32)
    dx, dl = torch.from_numpy(x).cuda(), torch.from_numpy(labels).cuda()
    s = torch.cuda.current_stream().cuda_stream
    net.zero_grad_dev(s)
    net.accumulate_dev(dx.data_ptr(), dl.data_ptr(), n, stream=s)
    torch.cuda.synchronize()
    g = grad_of(net)
    live = np.ones(n, bool); live[17] = False
    want = onet.grad(x[live], labels=labels[live])
    scale = np.abs(want).max()
    np.testing.assert_allclose(g[:-2], want, atol=3e-5 * scale, err_msg=str(dims))
    assert g[-1] == n - 1, (dims, g[-1])
    probs = onet.forward(x[live])
    loss = -np.log(np.maximum(probs[np.arange(n - 1), labels[live]], 1e-12)).sum()
    assert abs(g[-2] - loss) < 1e-4 * loss, (dims, g[-2], loss)
    # a second call adds on top (two epochs' worth of gradient), and inference is chunked too
    net.accumulate_dev(dx.data_ptr(), dl.data_ptr(), n, stream=s)
    torch.cuda.synchronize()
    np.testing.assert_allclose(grad_of(net)[:-2], 2 * want, atol=6e-5 * scale)
    np.testing.assert_allclose(net.forward(x), onet.forward(x), atol=1e-4)
print("chunks ok")
"""


# MFMA head, VALU head, GEMM output path, MFMA head + dW3 GEMM with 2, 3 and 4 blocks of 16 classes,
# and one net wider than any head kernel (GEMM output path)
DIMS = "((60, 512, 256, 16), (60, 64, 32, 5), (60, 96, 40, 40), (60, 64, 128, 17), (60, 128, 64, 40), (60, 512, 256, 64), (60, 160, 320, 70))"


def test_accumulate_over_small_chunks_matches_oracle():
    env = dict(os.environ, STREAMZ_CHUNK_ROWS="256")
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "dims": DIMS, "n": 1000}], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "chunks ok" in r.stdout


def test_split_bf16_mode_over_small_chunks_matches_oracle():
    """The same with the split-bf16 kernels (STREAMZ_GEMM=bf16x3): 3 000 rows in eight chunks of 384 rows
    (the last one 312 rows: not a multiple of the 32 rows the k-major reads take at a time, so the rows
    behind it are zeroed), weight-gradient slabs adding over chunks with workgroups that walk over several
    tiles; a 16-class net (MFMA head) and a 256-class one (output layer on the same kernels)."""
    env = dict(os.environ, STREAMZ_CHUNK_ROWS="384", STREAMZ_GEMM="bf16x3")
    dims = "((60, 512, 256, 16), (60, 256, 256, 256))"
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "dims": dims, "n": 3000}], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "chunks ok" in r.stdout
