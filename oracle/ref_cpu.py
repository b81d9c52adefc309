"""CPU oracle for the fake-quantized linear hot path.  TEST INFRASTRUCTURE ONLY.

This module restates, with plain torch CPU ops, the arithmetic of the reference's
``SPLinearWithLoRA.forward`` and everything it calls.  It is the *checker*: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it.  The product
package (``llm-qat-on-gpt2_amd``) never does; it fails loudly when the HIP library is missing.

Parity pin: the functions below are verified bit-for-bit against the imported reference by
``tests/golden/make_golden.py`` (run in the build container, where ``/root/reference`` exists) and against
the committed fixtures in ``tests/golden/*.npz`` by ``tests/test_oracle_golden.py`` (runs anywhere).
The reference's own tests pin no numeric value on this path, so those fixtures are the pin.

Every function cites the reference lines it follows (paths relative to
``part1_switchable_precision/``).  Elementwise torch CPU ops are deterministic, so "same ops, same
order" gives the same bits; the GEMMs (``F.linear`` / ``matmul``) have unspecified summation order and are
compared with a tolerance everywhere.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Optional, Tuple

import torch
import torch.nn.functional as F

LOG_FN_EPS = 1e-5  # quantization_methods.py:35 -- hard-coded, independent of the module's eps

# The two transcendental ops of the log path.  Default: exactly what the reference calls (ATen's CPU kernels, <= 1 ulp SLEEF
# routines).  `correctly_rounded_transcendentals()` swaps in fp64-computed, fp32-rounded twins so that tools/log_tolerance_study.py
# can measure how much of a log-path difference is the reference's own last-bit noise (test infrastructure; nothing else uses it).
_log2 = torch.log2


def _pow2(t):
    return torch.pow(2, t)


class correctly_rounded_transcendentals:
    def __enter__(self):
        global _log2, _pow2
        self._saved = (_log2, _pow2)
        _log2 = lambda t: torch.log2(t.double()).float()            # noqa: E731
        _pow2 = lambda t: torch.pow(2.0, t.double()).float()        # noqa: E731
        return self

    def __exit__(self, *exc):
        global _log2, _pow2
        _log2, _pow2 = self._saved
        return False


# --------------------------------------------------------------------------------------------------
# a1: MinMaxQuantizationFunction.forward            quantization_methods.py:8-22
# --------------------------------------------------------------------------------------------------
def minmax_levels(x: torch.Tensor, scale: torch.Tensor, zero_point: torch.Tensor, bits: int,
                  symmetric: bool = True) -> torch.Tensor:
    """Integer quantization levels (held in fp32), before dequantisation.

    symmetric  : clamp(round(x/scale), -n, n), n = 2^(b-1)-1       (:14-15)
    asymmetric : clamp(round(x/scale + zp), 0, 2^b-1)              (:18-19)
    ``torch.round`` is round-half-to-even; the divide is a true IEEE divide.
    """
    if symmetric:
        n = 2 ** (bits - 1) - 1
        return torch.clamp(torch.round(x / scale), -n, n)
    return torch.clamp(torch.round(x / scale + zero_point), 0, 2 ** bits - 1)


def minmax_fakequant(x, scale, zero_point, bits: int, symmetric: bool = True) -> torch.Tensor:
    """Quantize-dequantize.  symmetric: q*scale (:16); asymmetric: (q-zp)*scale (:20)."""
    q = minmax_levels(x, scale, zero_point, bits, symmetric)
    if symmetric:
        return q * scale
    return (q - zero_point) * scale


# --------------------------------------------------------------------------------------------------
# a3: LogQuantizationFunction.forward               quantization_methods.py:33-79
# --------------------------------------------------------------------------------------------------
def log_levels(x, log_min, log_range, bits: int, symmetric: bool = True
               ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Integer level q (sym: [-n,n]; asym: [0,2^b-1]) and the pre-round value it was rounded from.

    Follows :41-61.  The pre-round value is returned so tests can classify a mismatch as tie-adjacent.
    """
    mag = torch.abs(x).clamp(min=LOG_FN_EPS)                                   # :45
    lg = _log2(mag)                                                            # :47  torch.log2
    ln = (lg - log_min) / (log_range.clamp(min=LOG_FN_EPS))                    # :49
    ln = torch.clamp(ln, 0, 1)                                                 # :50
    if symmetric:
        n = 2 ** (bits - 1) - 1
        pre = (ln - 0.5) * 2 * n                                               # :54-55
        q = torch.clamp(torch.round(pre), -n, n)                               # :55-56
    else:
        n = 2 ** bits - 1
        pre = ln * n                                                           # :60
        q = torch.clamp(torch.round(pre), 0, n)                                # :60-61
    return q, pre


def log_fakequant(x, log_min, log_range, bits: int, symmetric: bool = True) -> torch.Tensor:
    """Log2-domain quantize-dequantize, op for op as :41-74 (note the (2^b-1) round trip, :57/:64,
    and that x_hat uses the *unclamped* log_range, :68)."""
    zero_mask = torch.abs(x) < LOG_FN_EPS                                      # :41
    sgn = torch.sign(x)                                                        # :43
    q, _ = log_levels(x, log_min, log_range, bits, symmetric)
    if symmetric:
        n = 2 ** (bits - 1) - 1
        qq = (q / (2 * n) + 0.5) * (2 ** bits - 1)                             # :57
        qn = qq / (2 ** bits - 1)                                              # :64
    else:
        n = 2 ** bits - 1
        qn = q / n                                                             # :66
    x_hat = qn * log_range + log_min                                           # :68
    out = _pow2(x_hat) * sgn                                                   # :70-72  torch.pow(2, x_hat)
    return torch.where(zero_mask, torch.zeros_like(x), out)                    # :74


# --------------------------------------------------------------------------------------------------
# a6: statistics    quantization.py:141-209       a7: finish_calibration   quantization.py:104-139
# --------------------------------------------------------------------------------------------------
def _reduce_dims(ndim: int, per_channel: bool, channel_dim: Optional[int]):
    dims = list(range(ndim))                                                   # :144-150
    if per_channel and channel_dim is not None:
        actual = channel_dim if channel_dim >= 0 else ndim + channel_dim
        if actual in dims:
            dims.remove(actual)
    return dims


def keepdim_min_max(t: torch.Tensor, dims) -> Tuple[torch.Tensor, torch.Tensor]:
    lo, hi = t, t                                                              # :152-162
    for d in sorted(dims, reverse=True):
        lo = lo.min(dim=d, keepdim=True)[0]
        hi = hi.max(dim=d, keepdim=True)[0]
    return lo, hi


@dataclass
class QuantState:
    """The calibration state machine of ``LearnableFakeQuantize`` (quantization.py:15-139), as data."""
    bits: int
    qtype: str = "minmax"            # 'minmax' | 'log'
    channel_dim: Optional[int] = 0
    per_channel: bool = True
    symmetric: bool = True
    eps: float = 1e-5
    scale: torch.Tensor = field(default_factory=lambda: torch.ones(1))
    zero_point: torch.Tensor = field(default_factory=lambda: torch.zeros(1))
    running_min: torch.Tensor = field(default_factory=lambda: torch.zeros(1))
    running_max: torch.Tensor = field(default_factory=lambda: torch.zeros(1))
    calibrated: bool = False
    collecting: bool = False
    nbatches: int = 0
    tmin: Optional[torch.Tensor] = None
    tmax: Optional[torch.Tensor] = None

    def start(self):                                                           # :96-102
        self.collecting, self.calibrated, self.nbatches = True, False, 0
        self.tmin = self.tmax = None

    def observe(self, x: torch.Tensor):                                        # :174-209
        cd = self.channel_dim if self.per_channel else None
        if self.qtype == "log":
            ax = torch.abs(x)
            if (ax > self.eps).any():                                          # :179-181
                lg = _log2(torch.clamp(ax, min=self.eps))                      # :182-183  torch.log2
                lo, hi = keepdim_min_max(lg, _reduce_dims(lg.dim(), self.per_channel, cd))
                self._merge(lo, hi)
            elif self.nbatches == 0:                                           # :194-197
                le = torch.log2(torch.tensor(self.eps))
                if self.per_channel and cd is not None:
                    shape = list(x.shape)
                    shape[cd if cd >= 0 else len(shape) + cd] = 1
                    self.tmin = torch.full(shape, le)
                    self.tmax = torch.full(shape, le)
                else:
                    self.tmin = torch.tensor(le)
                    self.tmax = torch.tensor(le)
        else:
            lo, hi = keepdim_min_max(x, _reduce_dims(x.dim(), self.per_channel, cd))
            self._merge(lo, hi)
        self.nbatches += 1                                                     # :209

    def _merge(self, lo, hi):
        if self.nbatches == 0:                                                 # :188-193 / :202-207
            self.tmin, self.tmax = lo.clone(), hi.clone()
        else:
            self.tmin = torch.minimum(self.tmin, lo)
            self.tmax = torch.maximum(self.tmax, hi)

    def finish(self):                                                          # :104-139
        if self.nbatches > 0 and self.tmin is not None:
            self.running_min, self.running_max = self.tmin.clone(), self.tmax.clone()
            self.scale, self.zero_point = finish_scale(self.running_min, self.running_max, self.bits,
                                                       self.qtype, self.symmetric, self.eps)
            self.calibrated = True
        self.collecting = False
        self.tmin = self.tmax = None

    def calibrate_on(self, t: torch.Tensor):
        """start -> one batch -> finish: how weights and LoRA factors are calibrated
        (train_sp.py:58-83, 133-159)."""
        self.start()
        self.observe(t)
        self.finish()
        return self

    def levels(self, x):
        if self.qtype == "minmax":
            return minmax_levels(x, self.scale, self.zero_point, self.bits, self.symmetric)
        return log_levels(x, self.zero_point, self.scale, self.bits, self.symmetric)[0]

    def __call__(self, x):                                                     # :211-226
        if self.bits >= 32:
            return x
        if self.collecting:
            self.observe(x)
            return x
        if not self.calibrated:
            raise RuntimeError("Quantizer not calibrated")
        if self.qtype == "minmax":
            return minmax_fakequant(x, self.scale, self.zero_point, self.bits, self.symmetric)
        if self.qtype == "log":                                                # arg order :237-239
            return log_fakequant(x, self.zero_point, self.scale, self.bits, self.symmetric)
        raise ValueError(f"Unknown quantizer type: {self.qtype}")


def finish_scale(rmin, rmax, bits: int, qtype: str, symmetric: bool, eps: float):
    """running min/max -> (scale, zero_point).                          quantization.py:110-127"""
    if qtype == "log":
        return (rmax - rmin).clone(), rmin.clone()          # scale=log_range, zero_point=log_min
    if symmetric:
        amax = torch.clamp(torch.max(torch.abs(rmin), torch.abs(rmax)), min=eps)
        return amax / (2 ** (bits - 1) - 1), torch.zeros_like(amax)
    rng = torch.clamp(rmax - rmin, min=eps)
    scale = rng / (2 ** bits - 1)
    return scale, torch.round(-rmin / scale)


# --------------------------------------------------------------------------------------------------
# a9: LoRALayer.forward  lora.py:45-54         a10/a11: SPLinearWithLoRA.forward  lora.py:127-150
# --------------------------------------------------------------------------------------------------
def lora_forward(x, A, B, qA: QuantState, qB: QuantState, scaling: float):
    """((x @ FQ(A)) @ FQ(B)) * scaling on the RAW (un-quantized) x."""
    out = torch.matmul(x, qA(A))
    out = torch.matmul(out, qB(B))
    return out * scaling


def sp_linear_forward(x, W, bias, qx: QuantState, qw: QuantState, A=None, B=None,
                      qA: Optional[QuantState] = None, qB: Optional[QuantState] = None,
                      scaling: float = 0.0, calibration_mode: bool = False, bits: int = 8):
    if bits >= 32:                                                             # lora.py:129-131
        return F.linear(x, W, bias)
    xq = qx(x)                                                                 # :141
    wq = qw(W)                                                                 # :142
    base = F.linear(xq, wq, bias)                                              # :144
    if calibration_mode:                                                       # :146-147
        return base
    if A is None or scaling == 0:                                              # lora.py:46-48
        return base + torch.zeros_like(base)
    return base + lora_forward(x, A, B, qA, qB, scaling)                       # :149-150


def sp_linear_backward(layer: "OracleLayer", x, g, calibration_mode: bool = False):
    """Closed form of what autograd gives the reference for one layer with the base weight frozen (main_sp.py:83):
    straight-through estimators -- identity for minmax (quantization_methods.py:25-28), clamp(grad, -10, 10) for log
    (:82-90) -- around F.linear and the two LoRA matmuls.  Returns (grad_x, grad_lora_A, grad_lora_B)."""
    def ste(grad, q):
        return torch.clamp(grad, -10, 10) if q.qtype == "log" else grad
    K, N = layer.W.shape[1], layer.W.shape[0]
    g2, x2 = g.reshape(-1, N), x.reshape(-1, K)
    gx = ste(g2 @ layer.qw(layer.W), layer.qx)                                 # through F.linear, then the input quantizer
    if calibration_mode or layer.scaling == 0:
        return gx.reshape(x.shape), None, None
    aq, bq = layer.qA(layer.A), layer.qB(layer.B)
    gt = (g2 * layer.scaling) @ bq.t()                                         # d/d(x@Aq) after "* scaling" (lora.py:53)
    gx = gx + gt @ aq.t()                                                      # the LoRA branch sees the raw x
    gA = ste(x2.t() @ gt, layer.qA)
    gB = ste((x2 @ aq).t() @ (g2 * layer.scaling), layer.qB)
    return gx.reshape(x.shape), gA, gB


# --------------------------------------------------------------------------------------------------
# Synthetic workload of BASELINE.md §3 / SURVEY.md §8(d) -- shared by tests and bench.py's CPU leg.
# --------------------------------------------------------------------------------------------------
from llm_qat_on_gpt2_amd.synthetic import kaiming_uniform_a5, make_workload  # noqa: E402,F401  (shared input generator)


@dataclass
class OracleLayer:
    """One calibrated layer at one bit-width: what SPLinearWithLoRA holds under one '{b}bit' key."""
    W: torch.Tensor
    bias: torch.Tensor
    A: torch.Tensor
    B: torch.Tensor
    qx: QuantState
    qw: QuantState
    qA: QuantState
    qB: QuantState
    scaling: float
    bits: int

    def forward(self, x, calibration_mode=False):
        return sp_linear_forward(x, self.W, self.bias, self.qx, self.qw, self.A, self.B, self.qA,
                                 self.qB, self.scaling, calibration_mode, self.bits)


def build_calibrated_layer(W, bias, A, B, calib_batches, bits: int, qtype: str, per_channel: bool,
                           alpha: float, rank: int, eps: float = 1e-5, symmetric: bool = True
                           ) -> OracleLayer:
    """The CalibrationManager protocol (train_sp.py:47-123,125-163) on one layer."""
    qw = QuantState(bits, qtype, 0, per_channel, symmetric, eps).calibrate_on(W)
    qA = QuantState(bits, qtype, 1, per_channel, symmetric, eps).calibrate_on(A)
    qB = QuantState(bits, qtype, 1, per_channel, symmetric, eps).calibrate_on(B)
    qx = QuantState(bits, qtype, -1, per_channel, symmetric, eps)
    layer = OracleLayer(W, bias, A, B, qx, qw, qA, qB, alpha / rank, bits)
    qx.start()
    for xb in calib_batches:                       # forwards with LoRA off; qx only records stats
        layer.forward(xb, calibration_mode=True)
    qx.finish()
    return layer


# --------------------------------------------------------------------------------------------------
# f1: the producers either side of the linears inside a block
# --------------------------------------------------------------------------------------------------
def switchable_layernorm(x, weight, bias, eps: float = 1e-5):
    """SwitchableLayerNorm.forward over the last axis (switchable_batchnorm.py:102-109)."""
    mean = x.mean(dim=[-1], keepdim=True)                                      # :104
    var = x.var(dim=[-1], keepdim=True, unbiased=False)                        # :105
    x_normalized = (x - mean) / torch.sqrt(var + eps)                          # :106
    return weight * x_normalized + bias                                        # :109


def attention_core(qkv, n_head: int):
    """SPAttention.forward between its two projections (models_sp.py:61-75): split heads, q k^T / sqrt(d), causal mask,
    softmax, times v, merge heads."""
    import math
    B, T, C3 = qkv.shape
    C = C3 // 3
    d = C // n_head
    q, k, v = qkv.split(C, dim=2)                                              # :62
    q = q.view(B, T, n_head, d).transpose(1, 2)                                # :64-66
    k = k.view(B, T, n_head, d).transpose(1, 2)
    v = v.view(B, T, n_head, d).transpose(1, 2)
    att = (q @ k.transpose(-2, -1)) / math.sqrt(d)                             # :68
    mask = torch.tril(torch.ones(T, T))                                        # :50, :69
    att = att.masked_fill(mask == 0, float('-inf'))                            # :70
    att = F.softmax(att, dim=-1)                                               # :71
    out = att @ v                                                              # :72
    return out.transpose(1, 2).contiguous().view(B, T, C)                      # :73


def sp_mlp_forward(x, fc: "OracleLayer", proj: "OracleLayer", calibration_mode: bool = False):
    """SPMLP.forward (models_sp.py:124-128): c_fc -> nn.GELU() (exact erf) -> c_proj."""
    h = F.gelu(fc.forward(x, calibration_mode))
    return proj.forward(h, calibration_mode), h


def build_calibrated_mlp(fc_t, proj_t, calib_batches, bits: int, qtype: str, per_channel: bool, alpha: float, rank: int,
                         eps: float = 1e-5):
    """CalibrationManager protocol (train_sp.py:47-123) on an SPMLP: weights and LoRA factors on themselves, both input
    quantizers through LoRA-free forwards of the whole MLP (c_proj sees gelu of c_fc's calibration-mode output)."""
    layers = []
    for (W, bias, A, B) in (fc_t, proj_t):
        qw = QuantState(bits, qtype, 0, per_channel, True, eps).calibrate_on(W)
        qA = QuantState(bits, qtype, 1, per_channel, True, eps).calibrate_on(A)
        qB = QuantState(bits, qtype, 1, per_channel, True, eps).calibrate_on(B)
        qx = QuantState(bits, qtype, -1, per_channel, True, eps)
        layers.append(OracleLayer(W, bias, A, B, qx, qw, qA, qB, alpha / rank, bits))
    fc, proj = layers
    fc.qx.start(); proj.qx.start()
    for xb in calib_batches:
        sp_mlp_forward(xb, fc, proj, calibration_mode=True)
    fc.qx.finish(); proj.qx.finish()
    return fc, proj
