"""Accuracy of the two arithmetic modes of the large products against a float64 evaluation of the
same network (forward lib.rs:880-890, gradient lib.rs:1013-1045): the split-bf16 mode must be as
close to the exact result as the f32 matrix-instruction mode is."""
import numpy as np
import pytest

from helpers import grad_view

pytestmark = pytest.mark.gpu


def _f64_forward_backward(params, x, labels):
    w1, b1, w2, b2, w3, b3 = [p.astype(np.float64) for p in params]
    x = x.astype(np.float64)
    z1 = x @ w1 + b1
    h1 = np.maximum(z1, 0.0)
    h2 = np.tanh(h1 @ w2 + b2)
    logits = h2 @ w3 + b3
    logits -= logits.max(axis=1, keepdims=True)
    p = np.exp(logits)
    p /= p.sum(axis=1, keepdims=True)
    d3 = p.copy()
    d3[np.arange(len(x)), labels] -= 1.0
    d2 = (d3 @ w3.T) * (1.0 - h2 * h2)
    d1 = (d2 @ w2.T) * (z1 > 0.0)
    grads = [x.T @ d1, d1.sum(0), h1.T @ d2, d2.sum(0), h2.T @ d3, d3.sum(0)]
    return p, np.concatenate([g.ravel() for g in grads])


@pytest.mark.parametrize("dims,scale2", [((60, 512, 256, 16), 1.0), ((60, 1024, 768, 40), 0.15)])
def test_split_bf16_mode_is_as_accurate_as_the_f32_mode(dims, scale2):
    import torch
    import streamz_amd as sz
    rng = np.random.default_rng(17)
    shapes = [(dims[0], dims[1]), (dims[1],), (dims[1], dims[2]), (dims[2],), (dims[2], dims[3]), (dims[3],)]
    params = [(rng.uniform(-0.5, 0.5, s) * (scale2 if len(s) == 2 and s[0] > 60 else 1.0)).astype(np.float32) for s in shapes]
    n = 3000   # not a multiple of the 256-row tiles
    x = rng.standard_normal((n, 60)).astype(np.float32)
    labels = rng.integers(0, dims[3], n).astype(np.int32)
    probs64, grad64 = _f64_forward_backward(params, x, labels)
    err = {}
    for mode in ("f32", "bf16x3"):
        net = sz.SimpleNeuralNet(*dims, init=params)
        net.set_gemm(mode)
        probs = net.forward(x)
        dx, dl = torch.from_numpy(x).cuda(), torch.from_numpy(labels).cuda()
        net.zero_grad_dev(0)
        net.accumulate_dev(dx.data_ptr(), dl.data_ptr(), n, stream=0)
        torch.cuda.synchronize()
        g = grad_view(net).cpu().numpy()[:-2].astype(np.float64)
        err[mode] = (np.abs(probs - probs64).max(), np.abs(g - grad64).max() / np.abs(grad64).max())
    print("max |probability error|, max relative gradient error against float64:", err)
    assert err["f32"][0] < 1e-4 and err["bf16x3"][0] < 1e-4                      # the task's bar on the outputs
    assert err["bf16x3"][0] <= max(2.0 * err["f32"][0], 2e-6)
    assert err["bf16x3"][1] <= max(2.0 * err["f32"][1], 2e-6)
