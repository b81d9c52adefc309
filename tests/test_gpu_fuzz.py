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



@pytest.mark.parametrize("switch", ["SPQ_XPASS_STREAM=0", "SPQ_XPASS_STREAM=16", "SPQ_XPASS_STREAM=32", "SPQ_PREP_ROLE=0", "SPQ_SPLIT_K=0",
                                    "SPQ_AUTO_I8=0"])
def test_alternative_kernels_stay_parity_green(switch):
    """The remaining switches of DESIGN.md 'Run-time switches' (read once per process, hence a child process each): a short
    forward sweep under each."""
    import subprocess
    name, value = switch.split("=")
    env = dict(os.environ, **{name: value})
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py")
    modes = ["fwd"]
    for mode in modes:
        r = subprocess.run([sys.executable, tool, "--cases", "14", "--seed", "5", "--mode", mode], env=env, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, f"{switch} ({mode}):\n" + "\n".join(l for l in r.stdout.splitlines() if l.startswith("FAIL")) + r.stderr[-800:]
