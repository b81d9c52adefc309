"""A short randomised parity sweep on the GPU (tools/fuzz_parity.py): random shapes, widths 2-16, minmax / log, per-channel /
per-tensor, symmetric / asymmetric, ranks 0-128, several input distributions -- every case within the forward tolerance."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_random_layers_against_oracle():
    import fuzz_parity
    failures = fuzz_parity.sweep(cases=40, seed=20260104, verbose=False)
    assert not failures, failures


def test_random_backward_against_oracle():
    import fuzz_parity
    failures = fuzz_parity.sweep(cases=16, seed=7, verbose=False, mode="bwd")
    assert not failures, failures


def test_random_cpt_layers_against_oracle():
    import fuzz_parity
    failures = fuzz_parity.sweep(cases=24, seed=11, verbose=False, mode="cpt")
    assert not failures, failures

