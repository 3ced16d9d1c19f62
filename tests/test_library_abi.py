"""CPU-only checks of the C-ABI library: it loads, exports every symbol the header declares,
and the pure-integer entry points agree with the oracle.  No compute call is made."""
import os
import re

import pytest

import streamz_amd as sz
from oracle import pyoracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "streamz_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(szh_[a-z0-9_]+)\s*\(", text)))


def test_library_is_built_in_tree():
    assert os.path.exists(sz.library_path()), "run __graft_entry__.build() first"


def test_every_declared_symbol_is_exported():
    L = sz.lib()
    names = header_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(L, n), f"{n} declared in streamz_hip.h but not exported"
    # the Python binding covers the whole header too
    assert set(names) == set(sz.exported_symbols())


@pytest.mark.parametrize("n", [0, 1, 799, 800, 801, 1199, 1200, 1201, 220500, 2646000, 15_876_000_000])
def test_window_count_matches_oracle(n):
    assert sz.window_count(n) == po.window_count(n)


def test_compute_fails_loudly_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(sz.StreamzError):
        sz.FeatureExtractor()
    with pytest.raises(sz.StreamzError):
        sz.SimpleNeuralNet(60, 8, 4, 2)
