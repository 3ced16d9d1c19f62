"""The resident batch-of-8 kernel when a workgroup gives up waiting for the others (forced here by
STREAMZ_STEP_FORCE_ABORT, which is read when the library is loaded, hence child processes):
  mode 2  the call reports an error and the parameters are exactly as before the call;
  mode 1  the call falls back to one launch per layer and step and still matches the oracle."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
import numpy as np
sys.path.insert(0, %(root)r)
import streamz_amd as sz
from oracle import pyoracle as po

mode = %(mode)d
rng = np.random.default_rng(23)
dims = (60, 512, 256, 16)
onet = po.Net(*dims)
params = [rng.uniform(-0.5, 0.5, s).astype(np.float32) for s in onet.shapes()]
onet.set_params(params)
net = sz.SimpleNeuralNet(*dims, init=params)
n, epochs = 37, 2
wins = rng.standard_normal((n, 60)).astype(np.float32)
perm = np.stack([rng.permutation(n) for _ in range(epochs)]).astype(np.uint32)
keep = (rng.random((epochs, n, 60)) >= 0.2).astype(np.uint8)
before = net.get_flat()
sz.profile_enable(True)
sz.profile_report()
if mode == 2:
    try:
        sz.pretrain_from_features(net, wins, 3, 16, epochs, 0.02, 0.2, 8, perm=perm, keep=keep)
    except sz.StreamzError as e:
        assert "timed out" in str(e), str(e)
    else:
        raise SystemExit("the forced time-out was not reported")
    np.testing.assert_array_equal(net.get_flat(), before)
else:
    loss = sz.pretrain_from_features(net, wins, 3, 16, epochs, 0.02, 0.2, 8, perm=perm, keep=keep)
    tags = sz.profile_report()
    assert "steps" in tags and "fwd2" in tags, tags   # the resident kernel was tried, the per-layer path did the work
    oloss = onet.pretrain_from_features(wins, 3, 16, epochs, 0.02, 0.2, 8, perm=perm, keep=keep)
    assert abs(loss - oloss) < 1e-4 * max(1.0, abs(oloss)), (loss, oloss)
    for p, q in zip(net.get_params(), onet.get_params()):
        np.testing.assert_allclose(p, q, atol=1e-4)
print("abort ok")
"""


@pytest.mark.parametrize("mode", [2, 1])
def test_step_kernel_time_out_leaves_parameters_or_falls_back(mode):
    env = dict(os.environ, STREAMZ_STEP_FORCE_ABORT=str(mode))
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "mode": mode}], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "abort ok" in r.stdout
