"""Shared helpers for the parity tests (test infrastructure)."""
import glob
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

LAYER_CASES = sorted(os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(GOLDEN, "case_*.npz"))
                     if not os.path.basename(p).startswith("case_q_"))
QUANT_CASES = sorted(os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(GOLDEN, "case_q_*.npz")))


def load_case(name):
    z = np.load(os.path.join(GOLDEN, f"case_{name}.npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    t = {k: torch.from_numpy(z[k]) for k in z.files if k != "meta"}
    return meta, t


def assert_close_y(y, y_ref, what="y", rel=1e-5):
    """The tolerance of BASELINE.json north_star as made precise in SURVEY.md §7 hard part 4:
    |d| <= 1e-5*|y_ref| + 1e-5*rms(y_ref)  (a pure elementwise rtol fails on the reference itself)."""
    y = y.detach().double().cpu()
    y_ref = y_ref.detach().double().cpu()
    assert y.shape == y_ref.shape, (what, y.shape, y_ref.shape)
    rms = float(y_ref.pow(2).mean().sqrt())
    err = (y - y_ref).abs()
    bound = rel * y_ref.abs() + rel * rms
    worst = float((err / bound).max())
    assert worst <= 1.0, f"{what}: max err/bound = {worst:.3f} (max abs err {float(err.max()):.3e}, rms {rms:.3e})"
    return worst
