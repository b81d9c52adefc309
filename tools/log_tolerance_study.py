#!/usr/bin/env python3
"""How much of a log-path output difference is the reference's own last-bit noise?   

north_star asks for 1e-5 relative; the log-quantised GPU cases are asserted at 2e-5.  The reference's log quantizer calls ATen's CPU
`log2` and `pow(2, .)`, both <= 1-ulp SLEEF kernels, NOT correctly rounded.  A log quantizer has only 2^b output magnitudes per
channel, so one last-bit difference in `pow(2, x_hat)` for a (channel, level) pair is shared by every element at that level -- a
systematic, not a random, perturbation of a contraction operand.

For every log fixture (tests/golden/case_log*.npz) this script evaluates, with the fixture's own (reference-produced) scales:
  A = the oracle with ATen's log2 / pow (bit-identical to the reference, tests/golden/make_golden.py)
  B = the oracle with correctly rounded (fp64 -> fp32) log2 / exp2
and records the distribution of |y_A - y_B| / (1e-5 |y_A| + 1e-5 rms(y_A)) -- "err/bound" at north_star's tolerance.  That is the
noise floor any implementation with differently-rounded transcendentals sits on.  With --gpu the HIP path is measured against both
A and B on the same inputs.      CPU:  python tools/log_tolerance_study.py            GPU:  python tools/log_tolerance_study.py --gpu
Writes one JSON object (profiles/r02_log_tolerance_{cpu,gpu}.json when --out is given)."""
import argparse
import glob
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_cpu as O  # noqa: E402  (this tool is test infrastructure: it measures the checker against itself)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_case(path):
    z = np.load(path, allow_pickle=False)
    return json.loads(str(z["meta"])), {k: torch.from_numpy(z[k]) for k in z.files if k != "meta"}


def oracle_layer(meta, t):
    """The oracle's layer with the FIXTURE's scales (what the reference derived), so A and B differ in the quantize/dequantize
    transcendentals only, not in calibration."""
    bits, qt, pc = meta["bits"], meta["qtype"], meta["per_channel"]
    def q(tag, cd):
        s = O.QuantState(bits, qt, cd, pc)
        s.scale, s.zero_point, s.calibrated = t[f"{tag}.scale"], t[f"{tag}.zero_point"], True
        return s
    return O.OracleLayer(t["W"], t["bias"], t["A"], t["B"], q("qx", -1), q("qw", 0), q("qA", 1), q("qB", 1),
                         meta["alpha"] / meta["r"] if meta["r"] else 0.0, bits)


def dist(y, ref, tol=1e-5):
    y, ref = y.double().reshape(-1), ref.double().reshape(-1)
    rms = float(ref.pow(2).mean().sqrt())
    ratio = (y - ref).abs() / (tol * ref.abs() + tol * rms)
    srt = ratio.sort().values
    pick = lambda p: float(srt[min(len(srt) - 1, int(p * len(srt)))])
    return {"max": round(float(srt[-1]), 4), "p99.9": round(pick(0.999), 4), "p99": round(pick(0.99), 4),
            "median": round(pick(0.5), 4), "frac_over_1": float(f"{float((ratio > 1).float().mean()):.3e}"), "n": len(srt)}


def level_diff(lv_a, lv_b):
    d = (lv_a != lv_b)
    return {"mismatches": int(d.sum()), "of": d.numel()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpu", action="store_true")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    cases = sorted(glob.glob(os.path.join(GOLDEN, "case_log*.npz")))
    out = {"tolerance": "err/bound = |dy| / (1e-5 |y_ref| + 1e-5 rms(y_ref))  (north_star's 1e-5; the log tests assert 2e-5 = 2.0 here)",
           "cases": {}}
    if args.gpu:
        import llm_qat_on_gpt2_amd as pkg
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from test_gpu_parity import build_layer, DEV
    for path in cases:
        meta, t = load_case(path)
        name = os.path.basename(path)[5:-4]
        lay = oracle_layer(meta, t)
        x = t["x2"]
        with torch.no_grad():
            yA = lay.forward(x)
            fqA_x, fqA_w = lay.qx(x), lay.qw(t["W"])
            lvA_x, lvA_w = lay.qx.levels(x), lay.qw.levels(t["W"])
            with O.correctly_rounded_transcendentals():
                yB = lay.forward(x)
                fqB_x, fqB_w = lay.qx(x), lay.qw(t["W"])
                lvB_x, lvB_w = lay.qx.levels(x), lay.qw.levels(t["W"])
        assert torch.allclose(yA, t["y_x2"].reshape(yA.shape), rtol=1e-5, atol=1e-6), name      # A is the reference's result
        rec = {"shape": {k: meta[k] for k in ("M", "K", "N", "r", "bits") if k in meta},
               "reference_vs_correctly_rounded": {
                   "y": dist(yB, yA),
                   "fq_x_values_differing": float(f"{float((fqA_x != fqB_x).float().mean()):.3e}"),
                   "fq_W_values_differing": float(f"{float((fqA_w != fqB_w).float().mean()):.3e}"),
                   "levels_x": level_diff(lvA_x, lvB_x), "levels_W": level_diff(lvA_w, lvB_w)}}
        if args.gpu:
            layer, key = build_layer(pkg, meta, t)
            lora = layer.lora_adapters[key]
            quants = {"qx": layer.quantizers_input[key], "qw": layer.quantizers_weight[key], "qA": lora.quantize_A, "qB": lora.quantize_B}
            with torch.no_grad():
                for tag, q in quants.items():
                    if tuple(q.scale.shape) == tuple(t[f"{tag}.scale"].shape):
                        q.scale = t[f"{tag}.scale"].to(DEV); q.zero_point = t[f"{tag}.zero_point"].to(DEV); q._epoch += 1
                res = {}
                for pname, pth in (("f16x3", pkg._lib.PATH_F16X3), ("f32", pkg._lib.PATH_F32)):
                    layer.operand_path = pth
                    y = layer(x.to(DEV)).cpu()
                    res[pname] = {"vs_reference": dist(y, yA), "vs_correctly_rounded": dist(y, yB)}
                fq_x = quants["qx"](x.to(DEV)).cpu()
                fq_w = quants["qw"](t["W"].to(DEV)).cpu()
                lv_x = quants["qx"].quantize_levels(x.to(DEV)).cpu()
            res["fq_x_differs_from_reference"] = float(f"{float((fq_x != fqA_x.reshape(fq_x.shape)).float().mean()):.3e}")
            res["fq_x_differs_from_correctly_rounded"] = float(f"{float((fq_x != fqB_x.reshape(fq_x.shape)).float().mean()):.3e}")
            res["fq_W_differs_from_reference"] = float(f"{float((fq_w != fqA_w).float().mean()):.3e}")
            res["fq_W_differs_from_correctly_rounded"] = float(f"{float((fq_w != fqB_w).float().mean()):.3e}")
            res["levels_x_vs_reference"] = level_diff(lv_x.float(), lvA_x.reshape(lv_x.shape))
            res["levels_x_vs_correctly_rounded"] = level_diff(lv_x.float(), lvB_x.reshape(lv_x.shape))
            rec["gpu"] = res
        out["cases"][name] = rec
        print(name, json.dumps(rec), flush=True)
    worst_floor = max(c["reference_vs_correctly_rounded"]["y"]["max"] for c in out["cases"].values())
    out["summary"] = {"reference_noise_floor_max_err_over_bound": worst_floor}
    if args.gpu:
        for pname in ("f16x3", "f32"):
            out["summary"][f"gpu_{pname}_vs_reference_max"] = max(c["gpu"][pname]["vs_reference"]["max"] for c in out["cases"].values())
            out["summary"][f"gpu_{pname}_vs_correctly_rounded_max"] = max(c["gpu"][pname]["vs_correctly_rounded"]["max"] for c in out["cases"].values())
    print("SUMMARY", json.dumps(out["summary"]))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
