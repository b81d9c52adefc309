"""``LoRALayer`` and ``SPLinearWithLoRA`` on the fused HIP forward.

Host-side mirror of the reference's ``part1_switchable_precision/lora.py:13-150`` (``part5_squad/lora.py`` is an
identical copy): same constructors, attribute names, ``ModuleDict`` keys (``'{b}bit'``), buffers and state-dict
layout, so the classes drop into ``models_sp.py`` unchanged.  ``forward`` enqueues ``spq_linear_lora_fwd``
(include/spq.h) instead of ~20 ATen kernels; the weight-side operands FQ(W), FQ(A), FQ(B) are prepared once per
(weights, scales) instead of on every call (lora.py:142, :49-50 recompute them each forward; same values).
"""
import ctypes
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .fake_quantize import LearnableFakeQuantize


class LoRALayer(nn.Module):
    """Low-rank adapter with fake-quantised factors (lora.py:13-54)."""

    def __init__(self, in_features, out_features, rank, alpha, bits, quantizer_type, eps=1e-5, per_channel=True):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.rank = rank
        self.alpha = alpha
        self.bits = bits
        self.enabled = not (bits >= 32 or rank <= 0)
        if not self.enabled:
            self.scaling = 0
            self.register_buffer('lora_A', torch.zeros(1, 1))
            self.register_buffer('lora_B', torch.zeros(1, 1))
            self.quantize_A = None
            self.quantize_B = None
            return
        self.scaling = alpha / rank
        self.lora_A = nn.Parameter(torch.zeros(in_features, rank))
        self.lora_B = nn.Parameter(torch.zeros(rank, out_features))
        nn.init.kaiming_uniform_(self.lora_A, a=math.sqrt(5))
        nn.init.zeros_(self.lora_B)
        common = dict(num_bits=bits, quantizer_type=quantizer_type, channel_dim=1, eps=eps, per_channel=per_channel)
        self.quantize_A = LearnableFakeQuantize(**common)
        self.quantize_B = LearnableFakeQuantize(**common)
        # never written by the reference either, but present in every state_dict (lora.py:42-43)
        self.register_buffer('lora_A_quantized', torch.empty(in_features, rank))
        self.register_buffer('lora_B_quantized', torch.empty(rank, out_features))

    def forward(self, x):
        """Stand-alone adapter output ((x @ FQ(A)) @ FQ(B)) * scaling, lora.py:45-54.  SPLinearWithLoRA does not
        call this in its fused forward; it is here for callers that use the adapter on its own and for autograd."""
        if not self.enabled or self.scaling == 0:
            return torch.zeros(*x.shape[:-1], self.out_features, device=x.device, dtype=x.dtype)
        a_q = self.quantize_A(self.lora_A)
        b_q = self.quantize_B(self.lora_B)
        if torch.is_grad_enabled() and (x.requires_grad or self.lora_A.requires_grad or self.lora_B.requires_grad):
            return torch.matmul(torch.matmul(x, a_q), b_q) * self.scaling
        _lib.require_gpu(x, "LoRA input")
        x2 = x.contiguous().view(-1, self.in_features)
        y = _gemm_nt(_gemm_nt(x2, a_q.t().contiguous()), b_q.t().contiguous())
        return (y * self.scaling).view(*x.shape[:-1], self.out_features)


def _gemm_nt(a, b_nk, bias=None):
    """a[M,K] . b[N,K]^T on the fp32 MFMA kernel (spq_gemm_f32_nt)."""
    M, K = a.shape
    N = b_nk.shape[0]
    out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        rc = _lib.load().spq_gemm_f32_nt(a.data_ptr(), K, b_nk.data_ptr(), K, K, None, 0, None, 0, 0, 1.0,
                                         _lib.ptr(bias), out.data_ptr(), N, M, N, _lib.stream_ptr(a.device))
    _lib.check(rc, "spq_gemm_f32_nt")
    return out


def _gemm_tn(p, q, alpha=1.0):
    """alpha * p[M,I]^T . q[M,J] -> [I, J]: the token contraction of d/dA and d/dB (spq_gemm_f32_tn, deterministic)."""
    M, I = p.shape
    J = q.shape[1]
    assert q.shape[0] == M and p.is_contiguous() and q.is_contiguous()
    if I > J:                                   # the kernel's 64 x 128 tile wants the rank-thin operand on the left
        return _gemm_tn(q, p, alpha).t().contiguous()
    lib = _lib.load()
    out = torch.empty(I, J, dtype=torch.float32, device=p.device)
    nbytes = lib.spq_gemm_f32_tn_workspace_bytes(M, I, J)
    ws = _lib.workspace(p.device, nbytes)
    with torch.cuda.device(p.device):
        rc = lib.spq_gemm_f32_tn(p.data_ptr(), I, q.data_ptr(), J, M, I, J, float(alpha), out.data_ptr(), ws.data_ptr(),
                                 ws.numel(), _lib.stream_ptr(p.device))
    _lib.check(rc, "spq_gemm_f32_tn")
    return out


class _LimbGemm:
    """out[M, R] = a[M, C] . b[R, C]^T with fp32-level accuracy on the f16 MFMA kernel of SPQ_PATH_F16X3: ``b`` is split
    into two fp16 limbs per row (spq_prepare_f16x2 with an identity quantizer), ``a`` into two limbs of a * 2^G with G
    from max|a| found on the device (spq_dynamic_limb_scale).  Used by the backward pass, whose left operand -- the
    incoming gradient -- has no calibrated range.  Buffers are kept per (R, C) and reused."""

    def __init__(self):
        self.key = None

    @staticmethod
    def supported(R: int, C: int) -> bool:
        return R > 0 and C > 0

    def _buffers(self, R, C, device):
        if self.key != (R, C, device):
            lib = _lib.load()
            self.w = torch.empty(lib.spq_prep_f16x2_bytes(R, C, 0), dtype=torch.uint8, device=device)
            self.rowscale = torch.empty((R + 127) // 128 * 128, dtype=torch.float32, device=device)
            self.xscale = torch.empty(2, dtype=torch.float32, device=device)
            self.key = (R, C, device)

    def __call__(self, a: torch.Tensor, b: torch.Tensor, down: torch.Tensor = None, down_scale: torch.Tensor = None):
        """``down`` [r, C] (optional, r <= 128): also returns a . down^T [M, r], computed inside the activation pass."""
        M, C = a.shape
        R = b.shape[0]
        assert b.shape[1] == C and a.is_contiguous() and b.is_contiguous() and self.supported(R, C)
        dev = a.device
        self._buffers(R, C, dev)
        lib = _lib.load()
        one = _ones(dev)
        out = torch.empty(M, R, dtype=torch.float32, device=dev)
        st = _lib.stream_ptr(dev)
        r, t, down_p = 0, None, None
        if down is not None:
            r = down.shape[0]
            r_pad = (r + 63) // 64 * 64
            down_p = down.contiguous()
            if r_pad != r:                                   # the activation pass copies whole 64-row pieces
                down_p = torch.zeros(r_pad, C, dtype=torch.float32, device=dev)
                down_p[:r] = down
            t = torch.empty(M, r, dtype=torch.float32, device=dev)
        sws = _lib.LIMB_SCALE_WORKSPACE_BYTES
        ws = _lib.workspace(dev, lib.spq_fwd_workspace_bytes(M, C, R, r, _lib.PATH_F16X3) + sws + 256)
        stats_ptr = ws.data_ptr() + (ws.numel() - sws) // 256 * 256
        with torch.cuda.device(dev):
            rc = lib.spq_prepare_f16x2(b.data_ptr(), R, C, one.data_ptr(), one.data_ptr(), 0, 32, 0, 1,
                                       None, 0, None, None, 0, 0, 0, 1, 0.0, None, None, None, 0, 0, 0, 1,
                                       one.data_ptr(), 0, self.w.data_ptr(), self.w.numel(), self.rowscale.data_ptr(),
                                       None, st)
            _lib.check(rc, "spq_prepare_f16x2(backward)")
            rc = lib.spq_dynamic_limb_scale(a.data_ptr(), M * C, self.xscale.data_ptr(), stats_ptr, sws, st)
            _lib.check(rc, "spq_dynamic_limb_scale")
            args = _lib.FwdArgs(
                M=M, K=C, N=R, r=r, bits=32, qtype=0, symmetric=1, quantize_input=0, x_per_channel=0,
                path=_lib.PATH_F16X3, x=a.data_ptr(), sx=None, zx=None, x_limb_scale=self.xscale.data_ptr(),
                w_prep=self.w.data_ptr(), w_rowscale=self.rowscale.data_ptr(), bias=None, a_prep=_lib.ptr(down_p),
                b_prep=None, lora_scaling=0.0, y=out.data_ptr(), workspace=ws.data_ptr(),
                workspace_bytes=ws.numel() - sws - 256, ev_gemm_begin=None, ev_gemm_end=None, t_out=_lib.ptr(t),
                a_limb_scale=None)
            rc = lib.spq_linear_lora_fwd(ctypes.byref(args), st)
            _lib.check(rc, "spq_linear_lora_fwd(backward)")
        return out if down is None else (out, t)


class _Prepared:
    """Weight-side GEMM operands of one bit-width plus the signature of what they were built from."""
    __slots__ = ("sig", "path", "w", "w_rowscale", "a", "b", "r", "x_limb_scale", "a_limb_scale", "pending", "keep")

    def __init__(self):
        self.sig = None
        self.w = self.w_rowscale = self.a = self.b = self.x_limb_scale = self.a_limb_scale = None
        self.pending = None      # _lib.PrepareArgs not launched yet: the forward that follows makes the operands itself
        self.keep = None         # tensors the pending struct points into
        self.r = 0


def _sig(t):
    return (t.data_ptr(), t._version, tuple(t.shape))


class SPLinearWithLoRA(nn.Module):
    """Switchable-precision linear layer: one frozen fp32 weight, per-bit-width quantizers and adapters
    (lora.py:56-150)."""

    def __init__(self, in_features, out_features, bit_widths, lora_rank_per_bit, lora_alpha_per_bit,
                 quantizer_per_bit, eps=1e-5, per_channel=True):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.bit_widths = bit_widths
        self.lora_rank_per_bit = lora_rank_per_bit
        student_bits = [b for b in bit_widths if b < 32]
        self.current_bits = sorted(bit_widths, reverse=True)[1]     # second-largest width (lora.py:71)

        self.linear = nn.Linear(in_features, out_features, bias=True)

        def quantizer(bits, channel_dim, **kw):
            return LearnableFakeQuantize(num_bits=bits, quantizer_type=quantizer_per_bit[bits],
                                         channel_dim=channel_dim, eps=eps, per_channel=per_channel, **kw)

        self.quantizers_weight = nn.ModuleDict({f'{b}bit': quantizer(b, 0) for b in student_bits})
        self.quantizers_input = nn.ModuleDict({f'{b}bit': quantizer(b, -1, is_input=True) for b in student_bits})
        self.lora_adapters = nn.ModuleDict({
            f'{b}bit': LoRALayer(in_features, out_features, rank=lora_rank_per_bit[b], alpha=lora_alpha_per_bit[b],
                                 bits=b, quantizer_type=quantizer_per_bit[b], eps=eps, per_channel=per_channel)
            for b in student_bits})

        self.register_buffer('weight_quantized', torch.empty(out_features, in_features))
        self.register_buffer('input_quantized', None)
        self.calibration_mode = False

        # --- not part of the reference surface -----------------------------------------------------------
        self.operand_path = _lib.PATH_AUTO        # enum spq_path; AUTO picks the fastest valid one
        # reuse prepared operands in eval mode (see _operands); SPQ_CACHE_OPERANDS=0 turns the reuse off process-wide
        # (every forward re-quantises, as the reference does) for code that writes weights through `.data` in place
        self.cache_operands = os.environ.get("SPQ_CACHE_OPERANDS", "1") != "0"
        self._prepared = {}
        self._gemm_events = None                  # (hipEvent_t, hipEvent_t) around the dominant kernel, for bench.py
        self.backward_limbs = True                # d/dx on the f16 MFMA limb kernel (False: fp32 MFMA kernel)
        # True (default): a re-quantising forward (training mode, or cache_operands off) hands the weight-side preparation to the
        # forward call (spq_fwd_args.prepare).  Where the streaming activation kernel runs, the row work then rides in the SAME
        # launch as extra workgroups beside the activation workgroups (xpass_stream_prep_kernel; FQ(A)^T, which the pass
        # consumes, goes first as a launch of a few dozen workgroups): 0.0955 -> 0.092 ms per forward at the headline shape.
        # Elsewhere the library issues the ordinary preparation launch itself.  SPQ_FUSE_PREPARE=0: prepare from Python, as before
        self.fuse_prepare = os.environ.get('SPQ_FUSE_PREPARE', '1') != '0'
        self._bwd_gemm = None
        self._wq_t = None                         # (signature, FQ(W)^T) for the backward, see _fq_weight_t
        self._last_t = None                       # LoRA-down product of the last training forward (consumed by autograd)
        self._activation_fused = False
        self._last_path = None                    # operand path of the most recent fused forward
        self.fuse_norm = True                     # apply a preceding SwitchableLayerNorm inside the activation pass (forward(pre_norm=))
        self._pre_norm = None
        self._norm_fused = False

    # ---- precision switching (lora.py:105-125): attribute flips only ---------------------------------------
    def set_precision(self, bits) -> int:
        if bits >= 32:
            self.current_bits = 32
            return bits
        self.current_bits = bits
        key = f'{bits}bit'
        self.quantizers_weight[key].set_num_bits(bits)
        self.quantizers_input[key].set_num_bits(bits)
        lora = self.lora_adapters[key]
        if lora.quantize_A is not None:
            lora.quantize_A.set_num_bits(bits)
        if lora.quantize_B is not None:
            lora.quantize_B.set_num_bits(bits)
        return self.current_bits

    def get_active_lora(self):
        return self.lora_adapters[f'{self.current_bits}bit']

    def invalidate_operand_cache(self):
        """Drop prepared operands (needed only after writing weights through ``.data`` in eval mode)."""
        self._prepared.clear()
        self._wq_t = None

    # ---- forward (lora.py:127-150) ---------------------------------------------------------------------------
    def forward(self, x, activation=None, pre_norm=None):
        """``activation='gelu'`` (not in the reference's signature; used by this package's SPMLP): the exact-erf GELU that
        follows mlp.c_fc (models_sp.py:124-126) is applied to the output -- inside the contraction's store when the fused
        no-grad path runs, as a separate ``F.gelu`` otherwise.

        ``pre_norm`` (this package's SPBlock): the SwitchableLayerNorm whose output this layer consumes (models_sp.py:160-171:
        ln_1 -> c_attn, ln_2 -> c_fc); ``x`` is then that LayerNorm's INPUT.  Where the fused no-grad path allows, the
        normalisation happens inside the activation pass (spq_fwd_args.ln_weight) and the normalised tensor is never stored;
        otherwise ``pre_norm(x)`` is simply computed first.  Same values either way (the kernels share the arithmetic)."""
        if activation not in (None, 'gelu'):
            raise ValueError(f"unknown activation {activation!r}")
        self._norm_fused = False
        if pre_norm is not None:
            if self._can_fuse_norm(x, pre_norm):
                self._pre_norm = pre_norm
            else:
                x = pre_norm(x)
        try:
            y = self._forward(x, activation)
        finally:
            self._pre_norm = None
        if activation == 'gelu' and not self._activation_fused:
            y = F.gelu(y)
        return y

    def _can_fuse_norm(self, x, norm) -> bool:
        """The LayerNorm prologue of the activation pass exists in the panel kernels of the limb / int8 paths: fp32 CUDA input,
        no gradient, a calibrated symmetric-minmax-or-any input quantizer that is not recording statistics (those need the
        normalised tensor itself), K % 64 == 0, K <= 1024, rank <= 64."""
        if not self.fuse_norm or self.current_bits >= 32 or not x.is_cuda or x.dtype != torch.float32:
            return False
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())
                                        or any(p.requires_grad for p in norm.parameters())):
            return False
        key = f'{self.current_bits}bit'
        if key not in self.quantizers_input or key not in self.quantizers_weight:
            return False
        qx, qw, lora = self.quantizers_input[key], self.quantizers_weight[key], self.lora_adapters[key]
        if qx.collecting_stats or qw.collecting_stats or not qx.calibrated or qx.num_bits >= 32:
            return False
        if lora.enabled and (lora.quantize_A.collecting_stats or lora.quantize_B.collecting_stats or lora.rank > 64):
            return False
        K = self.in_features
        if K % 64 != 0 or K > 1024 or tuple(norm.normalized_shape) != (K,) or x.shape[-1] != K:
            return False
        use_lora = (not self.calibration_mode) and lora.enabled and lora.scaling != 0
        return self._choose_path(qx, qw, lora, use_lora, 1) in (_lib.PATH_F16X2, _lib.PATH_F16X3, _lib.PATH_I8)

    def _forward(self, x, activation=None):
        self._activation_fused = False
        if self.current_bits >= 32:
            return F.linear(x, self.linear.weight, self.linear.bias)
        key = f'{self.current_bits}bit'
        if key not in self.quantizers_weight or key not in self.quantizers_input:
            raise KeyError(f"No weight quantizer for {key}")
        qw, qx, lora = self.quantizers_weight[key], self.quantizers_input[key], self.lora_adapters[key]

        needs_grad = torch.is_grad_enabled() and (
            x.requires_grad or self.linear.weight.requires_grad
            or (self.linear.bias is not None and self.linear.bias.requires_grad)
            or (lora.enabled and not self.calibration_mode and (lora.lora_A.requires_grad or lora.lora_B.requires_grad)))
        if x.is_cuda and x.dtype in (torch.float16, torch.bfloat16):
            x = x.float()       # under torch.autocast the producer may hand over half precision; the kernels are fp32
        collecting = qw.collecting_stats or qx.collecting_stats or (
            lora.enabled and (lora.quantize_A.collecting_stats or lora.quantize_B.collecting_stats))
        if needs_grad and not collecting and x.is_cuda:
            # fused forward + straight-through backward on the MFMA kernels
            use_lora = (not self.calibration_mode) and lora.enabled and lora.scaling != 0
            return _SPLinearFunction.apply(x, self.linear.weight, self.linear.bias,
                                           lora.lora_A if use_lora else None, lora.lora_B if use_lora else None,
                                           self, key)
        if needs_grad or qw.collecting_stats or (lora.enabled and (lora.quantize_A.collecting_stats or lora.quantize_B.collecting_stats)):
            return self._forward_composed(x, qx, qw, lora)
        return self._forward_fused(x, key, qx, qw, lora, activation=activation)

    def _forward_composed(self, x, qx, qw, lora):
        """Autograd-capable composition: HIP fake-quant kernels with straight-through backward, GEMMs by
        torch-ROCm.  Used only when a gradient is required (training; SURVEY.md §8 f2 is the fused backward)."""
        base = F.linear(qx(x), qw(self.linear.weight), self.linear.bias)
        if self.calibration_mode:
            return base
        return base + lora(x)

    def _forward_fused(self, x, key, qx, qw, lora, keep_t=False, activation=None):
        _lib.require_gpu(x, "SPLinearWithLoRA input")
        _lib.check_device(x.device)
        W = self.linear.weight
        if W.device != x.device:
            raise RuntimeError(f"SPLinearWithLoRA: weight on {W.device}, input on {x.device}")
        if x.shape[-1] != self.in_features:
            raise RuntimeError(f"SPLinearWithLoRA: input has {x.shape[-1]} features, expected {self.in_features}")
        quantize_input = 1
        if qx.num_bits >= 32:
            quantize_input = 0
        elif qx.collecting_stats:                      # quantization.py:214-216: record, pass x through
            qx._collect_statistics_batch(x)
            quantize_input = 0
        elif not qx.calibrated:
            raise RuntimeError(
                f"Quantizer not calibrated. Please run calibration first for {qx.quantizer_type} quantizer.")
        elif qx.quantizer_type not in _lib.QTYPE_CODE:
            raise ValueError(f"Unknown quantizer type: {qx.quantizer_type}. Supported types: 'minmax', 'log'")
        self._last_t = None
        use_lora = (not self.calibration_mode) and lora.enabled and lora.scaling != 0
        prep = self._operands(key, qx, qw, lora, use_lora, quantize_input)
        self._last_path = prep.path

        x2 = x.detach().contiguous().view(-1, self.in_features)
        M, K, N = x2.shape[0], self.in_features, self.out_features
        # a 3-D keep-dim input scale lifts a 2-D input to 3-D, as x / scale does in the reference
        lead = tuple(x.shape[:-1])
        if quantize_input and qx.scale.dim() > x.dim():
            lead = (1,) * (qx.scale.dim() - x.dim()) + lead
        y = torch.empty(M, N, dtype=torch.float32, device=x.device)
        if M == 0:
            return y.view(*lead, N)
        r = prep.r if use_lora else 0
        lib = _lib.load()
        st = _lib.stream_ptr(x.device)
        ws = _lib.workspace(x.device, lib.spq_fwd_workspace_bytes(M, K, N, r, prep.path), st)
        sx, zx = qx.qparams_for(K) if quantize_input else (None, None)
        if quantize_input and sx.device != x.device:
            raise RuntimeError(f"input scale on {sx.device}, input on {x.device}")
        bias = self.linear.bias
        # training: keep the LoRA-down product x . FQ(A) for d/dB instead of recomputing it in the backward
        self._last_t = torch.empty(M, r, dtype=torch.float32, device=x.device) if (keep_t and r) else None
        a = _lib.FwdArgs(
            M=M, K=K, N=N, r=r, bits=int(qx.num_bits), qtype=_lib.QTYPE_CODE.get(qx.quantizer_type, 0),
            symmetric=1 if qx.symmetric else 0, quantize_input=quantize_input,
            x_per_channel=1 if (quantize_input and sx.numel() > 1) else 0, path=prep.path,
            x=x2.data_ptr(), sx=_lib.ptr(sx), zx=_lib.ptr(zx), x_limb_scale=_lib.ptr(prep.x_limb_scale),
            w_prep=prep.w.data_ptr(),
            w_rowscale=_lib.ptr(prep.w_rowscale), bias=_lib.ptr(bias), a_prep=_lib.ptr(prep.a) if r else None,
            b_prep=_lib.ptr(prep.b) if r else None, lora_scaling=float(lora.scaling) if r else 0.0,
            y=y.data_ptr(), workspace=ws.data_ptr(), workspace_bytes=ws.numel(),
            ev_gemm_begin=self._gemm_events[0] if self._gemm_events else None,
            ev_gemm_end=self._gemm_events[1] if self._gemm_events else None, t_out=_lib.ptr(self._last_t),
            a_limb_scale=_lib.ptr(prep.a_limb_scale) if r else None)
        if prep.pending is not None:
            a.prepare = ctypes.pointer(prep.pending)
        norm = self._pre_norm
        if norm is not None:                              # the activation pass normalises the rows itself
            nw, nb = norm.weights[str(norm.current_precision)], norm.biases[str(norm.current_precision)]
            a.ln_weight, a.ln_bias, a.ln_eps = nw.data_ptr(), nb.data_ptr(), float(norm.eps)
            self._norm_fused = True
        if activation == 'gelu' and (prep.path == _lib.PATH_I8 or (prep.path in (_lib.PATH_F16X2, _lib.PATH_F16X3) and N % 4 == 0)):
            a.epilogue = _lib.EPILOGUE_GELU
            self._activation_fused = True
        with _lib.on_device(x.device):
            rc = lib.spq_linear_lora_fwd(ctypes.byref(a), st)
        prep.pending = prep.keep = None
        _lib.check(rc, "spq_linear_lora_fwd")
        return y.view(*lead, N)

    def _fq_weight_t(self, qw):
        """FQ(W)^T, contiguous [K, N], for the backward contraction.  The base weight is frozen during QAT (main_sp.py:83), so
        it is rebuilt only when the weight (identity, autograd version) or the weight quantizer's calibration changes."""
        W = self.linear.weight
        sig = (_sig(W), qw._epoch, int(qw.num_bits), _sig(qw.scale))
        if self._wq_t is None or self._wq_t[0] != sig:
            with torch.no_grad():
                self._wq_t = (sig, qw(W.detach()).t().contiguous())
        return self._wq_t[1]

    # ---- weight-side operands --------------------------------------------------------------------------------
    def _choose_path(self, qx, qw, lora, use_lora, quantize_input):
        """SPQ_PATH_F16X2 (exact integer levels x 2-limb fp16 weights) whenever the input quantizer allows it:
        symmetric minmax, <= 12 bits, actually quantising; otherwise the always-valid fp32-MFMA path."""
        shape_ok = not use_lora or lora.rank <= 128
        f16_ok = (quantize_input and qx.quantizer_type == 'minmax' and qx.symmetric and 2 <= qx.num_bits <= 12
                  and shape_ok)
        # any other calibrated input quantizer (log, asymmetric, > 12 bit): FQ(x) as two fp16 limbs
        x3_ok = (quantize_input and not f16_ok and qx.quantizer_type in _lib.QTYPE_CODE and 1 <= qx.num_bits <= 16
                 and shape_ok)
        # int8 matrix cores (SPQ_PATH_I8): levels of <= 8 bits x the weight's own integer levels -- one product per algorithmic
        # product at twice the f16 rate -- valid when those levels exist (symmetric minmax weights, <= 8 bit) and the input scale
        # is per tensor (it then leaves the sum).  Measured 50-53 us against ~80 us for the contraction at the c_fc shape.
        # The byte-level activation operand has no NaN: a NaN activation (whose level the reference keeps NaN, quantization_methods.py
        # :14-15 on torch.clamp) becomes the level 0 there.  With the LoRA branch active that row still comes out NaN -- the fp32
        # LoRA-down product t = x . FQ(A) carries it -- but with no LoRA term (calibration_mode, a disabled or rank-0 adapter) the
        # row would be finite where F.linear gives NaN, so those forwards take the fp16-level path, whose levels do hold a NaN.
        i8_ok = bool(f16_ok and qx.num_bits <= 8 and self.in_features % 4 == 0 and self.in_features <= 4096
                     and qw is not None and qw.quantizer_type == 'minmax' and qw.symmetric and 2 <= qw.num_bits <= 8
                     and qx.scale.numel() == 1 and use_lora and lora is not None and lora.rank > 0)
        if self.operand_path == _lib.PATH_I8:
            return _lib.PATH_I8 if i8_ok else (_lib.PATH_F16X2 if f16_ok else (_lib.PATH_F16X3 if x3_ok else _lib.PATH_F32))
        if self.operand_path == _lib.PATH_AUTO:
            if _AUTO_I8 and i8_ok:
                return _lib.PATH_I8
            return _lib.PATH_F16X2 if f16_ok else (_lib.PATH_F16X3 if x3_ok else _lib.PATH_F32)
        if self.operand_path == _lib.PATH_F16X3:
            return _lib.PATH_F16X3 if (x3_ok or f16_ok) else _lib.PATH_F32
        if self.operand_path == _lib.PATH_F16X2 and not f16_ok:
            return _lib.PATH_F32            # e.g. calibration forwards (raw x) of a layer pinned to F16X2
        return self.operand_path

    def _operands(self, key, qx, qw, lora, use_lora, quantize_input):
        """Weight-side GEMM operands for the active bit-width: FQ(W), FQ(A)^T, FQ(B)^T in the layout of the chosen
        operand path (fp32, or 2-limb fp16 with the input scale folded in).

        The reference re-quantises these on every forward.  Here they are rebuilt on every call in training mode
        and, in eval mode, only when their inputs changed: tensor identity + autograd version of W/A/B and the
        calibration epoch of the quantizers.  (A write through ``weight.data`` does not bump the version; call
        ``invalidate_operand_cache()`` after one in eval mode.)
        """
        if not qw.calibrated:
            raise RuntimeError(
                f"Quantizer not calibrated. Please run calibration first for {qw.quantizer_type} quantizer.")
        W = self.linear.weight
        path = self._choose_path(qx, qw, lora, use_lora, quantize_input)
        use_cache = self.cache_operands and not self.training
        if use_lora:
            for q in (lora.quantize_A, lora.quantize_B):
                if not q.calibrated:
                    raise RuntimeError(
                        f"Quantizer not calibrated. Please run calibration first for {q.quantizer_type} quantizer.")
        sig = None
        if use_cache:                                   # (training re-quantises on every call: no signature needed)
            sig = [path, use_lora, _sig(W), qw._epoch, _sig(qw.scale), _sig(qw.zero_point)]
            if path in _LIMB_PATHS:
                sig += [qx._epoch, _sig(qx.scale), _sig(qx.zero_point)]
            if use_lora:
                for q, t in ((lora.quantize_A, lora.lora_A), (lora.quantize_B, lora.lora_B)):
                    sig += [_sig(t), q._epoch, _sig(q.scale), _sig(q.zero_point)]
            sig = tuple(sig)
        ckey = (key, path)
        prep = self._prepared.get(ckey)
        if prep is not None and use_cache and prep.sig == sig:
            return prep
        prep = prep or _Prepared()
        prep.path = path
        if path == _lib.PATH_F32:
            prep.w_rowscale = None
        prep.r = lora.rank if use_lora else 0
        with torch.no_grad():
            if path == _lib.PATH_F32 or not use_lora:
                prep.a = _fq_transposed(lora.quantize_A, lora.lora_A.detach(), 64) if use_lora else None   # [r->64k, K]
            if path == _lib.PATH_F32:
                prep.w = qw(W.detach())                                                            # [N,K]
                prep.b = _fq_transposed(lora.quantize_B, lora.lora_B.detach()) if use_lora else None  # [N,r]
            else:
                # re-quantising every call (training mode / cache off): hand the job to the forward, which spreads the row work
                # over its activation pass instead of launching it (spq_fwd_args.prepare)
                self._prepare_f16x2(prep, qx, qw, lora, use_lora, defer=not use_cache and self.fuse_prepare)
        prep.sig = sig
        self._prepared[ckey] = prep
        return prep

    def _prepare_f16x2(self, prep, qx, qw, lora, use_lora, defer=False):
        W = self.linear.weight.detach().contiguous()
        N, K, r = self.out_features, self.in_features, prep.r
        lib = _lib.load()
        nbytes = lib.spq_prep_bytes(N, K, r, prep.path)
        if getattr(prep, "w", None) is None or prep.w.dtype != torch.uint8 or prep.w.numel() < nbytes:
            prep.w = torch.empty(nbytes, dtype=torch.uint8, device=W.device)
        n_pad = (N + 127) // 128 * 128
        if prep.w_rowscale is None or prep.w_rowscale.numel() != n_pad or prep.w_rowscale.device != W.device:
            prep.w_rowscale = torch.empty(n_pad, dtype=torch.float32, device=W.device)
        qb = lora.quantize_B if use_lora else None
        qa = lora.quantize_A if use_lora else None
        B = lora.lora_B.detach().contiguous() if use_lora else None
        A = lora.lora_A.detach().contiguous() if use_lora else None
        if use_lora:
            r_pad = (r + 63) // 64 * 64
            if prep.a is None or tuple(prep.a.shape) != (r_pad, K) or prep.a.device != W.device:
                prep.a = torch.zeros(r_pad, K, dtype=torch.float32, device=W.device)     # rows >= r stay zero
        # (scale, zero_point) per quantizer: one value or one per channel (a reference checkpoint in the default-fill state
        # carries other -- constant -- shapes: LearnableFakeQuantize.qparams_for)
        sw, zw = qw.qparams_for(N)
        sb, zb = qb.qparams_for(N) if qb is not None else (None, None)
        sa, za = qa.qparams_for(r) if qa is not None else (None, None)
        if prep.path == _lib.PATH_F16X3:
            # no scale folding: the activation operand is FQ(x) itself, as two limbs of FQ(x) * 2^G (G from the quantizer's
            # own range bound -- FQ clamps, so the bound holds for any input); everything stays on the device
            sx_t = _ones(W.device)
            prep.x_limb_scale = _limb_scale(qx)
        else:
            sx_t = qx.qparams_for(K)[0]
            prep.x_limb_scale = None
        prep.a_limb_scale = None
        cur = torch.cuda.current_stream(W.device)
        prep.pending = prep.keep = None
        pa = _lib.PrepareArgs(
            W=W.data_ptr(), N=N, K=K, sw=sw.data_ptr(), zw=zw.data_ptr(), w_per_channel=1 if sw.numel() > 1 else 0,
            w_bits=int(qw.num_bits), w_qtype=_lib.QTYPE_CODE[qw.quantizer_type], w_symmetric=1 if qw.symmetric else 0,
            B=_lib.ptr(B), r=r, sb=_lib.ptr(sb), zb=_lib.ptr(zb), b_per_channel=(1 if sb.numel() > 1 else 0) if qb else 0,
            b_bits=int(qb.num_bits) if qb else 0, b_qtype=_lib.QTYPE_CODE[qb.quantizer_type] if qb else 0,
            b_symmetric=(1 if qb.symmetric else 0) if qb else 1, scaling=float(lora.scaling) if use_lora else 0.0,
            A=_lib.ptr(A), sa=_lib.ptr(sa), za=_lib.ptr(za), a_per_channel=(1 if sa.numel() > 1 else 0) if qa else 0,
            a_bits=int(qa.num_bits) if qa else 0, a_qtype=_lib.QTYPE_CODE[qa.quantizer_type] if qa else 0,
            a_symmetric=(1 if qa.symmetric else 0) if qa else 1, sx=sx_t.data_ptr(), x_per_channel=1 if sx_t.numel() > 1 else 0,
            w_prep=prep.w.data_ptr(), w_prep_bytes=prep.w.numel(), w_rowscale=prep.w_rowscale.data_ptr(),
            a_prep=_lib.ptr(prep.a) if use_lora else None, path=prep.path)
        if defer:
            prep.pending, prep.keep = pa, (W, B, A, sw, zw, sb, zb, sa, za, sx_t)
        else:
            with torch.cuda.device(W.device):
                rc = lib.spq_prepare_f16x2_args(ctypes.byref(pa), cur.cuda_stream)
            _lib.check(rc, "spq_prepare_f16x2")
        prep.b = prep.w      # LoRA-B limbs live inside the same buffer


_LIMB_PATHS = (_lib.PATH_F16X2, _lib.PATH_F16X3, _lib.PATH_I8)
# SPQ_AUTO_I8=0: PATH_AUTO never picks the int8 operand path (then: F16X2 wherever it would be valid)
_AUTO_I8 = os.environ.get('SPQ_AUTO_I8', '1')[:1] != '0'
_ones_cache = {}


def _ones(device):
    t = _ones_cache.get(device)
    if t is None:
        t = _ones_cache[device] = torch.ones(1, dtype=torch.float32, device=device)
    return t


def _limb_scale(q: LearnableFakeQuantize) -> torch.Tensor:
    """Device tensor {2^G, 2^-G} with bound(|FQ(x)|) * 2^G in [2^13, 2^14), from the quantizer's own range:
    minmax symmetric n*scale; asymmetric max(|0-zp|, |2^b-1-zp|)*scale; log 2^(log_min + log_range).  No host sync."""
    key = (q._epoch, int(q.num_bits), q.scale.data_ptr(), q.scale._version)
    cached = getattr(q, "_limb_scale_cache", None)
    if cached is not None and cached[0] == key:
        return cached[1]
    out = _limb_scale_uncached(q)
    q._limb_scale_cache = (key, out)           # a dozen tiny launches: keep it per calibration epoch
    return out


def _limb_scale_uncached(q) -> torch.Tensor:
    s, z = q.scale.detach().float(), q.zero_point.detach().float()
    if q.quantizer_type == 'log':
        bound = torch.exp2((z + s.clamp(min=0)).amax())
    elif q.symmetric:
        bound = s.amax() * float(2 ** (q.num_bits - 1) - 1)
    else:
        bound = (torch.maximum(z.abs(), (float(2 ** q.num_bits - 1) - z).abs()) * s).amax()
    bound = bound * (1.0 + 2.0 ** -10)                                  # rounding margin
    _, e = torch.frexp(bound)                                           # bound = m * 2^e, m in [0.5, 1)
    # exact powers of two, built from the exponent field (torch.ldexp goes through a float pow: 2^16 came out as 65535.996)
    k = torch.clamp(14 - e.to(torch.int32), -100, 100)
    k = torch.where(torch.isfinite(bound) & (bound > 0), k, torch.zeros_like(k))
    p = ((k + 127) << 23).view(torch.float32)
    pinv = ((127 - k) << 23).view(torch.float32)
    return torch.stack([p.reshape(()), pinv.reshape(())]).contiguous()


class _SPLinearFunction(torch.autograd.Function):
    """Fused forward (``spq_linear_lora_fwd``) with the reference's straight-through backward
    (quantization_methods.py:25-28 un-masked identity for minmax, :82-90 clamp to [-10, 10] for log):

        d/dx     = STE_x( g . FQ(W) ) + s * (g . FQ(B)^T) . FQ(A)^T         [one spq_gemm_f32_nt launch, two segments]
        d/dA     = STE_A( s * x^T . (g . FQ(B)^T) )        d/dB = STE_B( s * (x . FQ(A))^T . g )
        d/dW     = STE_W( g^T . FQ(x) )   (only if the base weight is trainable; main_sp.py:83 freezes it)
        d/dbias  = sum_m g

    Every contraction runs on this library's kernels: the M-contractions (d/dA, d/dB, d/dW) on spq_gemm_f32_tn, the N- and
    r-contractions on the f16-limb / fp32-MFMA kernels (no vendor BLAS on the path).
    """

    @staticmethod
    def forward(ctx, x, W, bias, A, B, module, key):
        qx, qw, lora = module.quantizers_input[key], module.quantizers_weight[key], module.lora_adapters[key]
        with torch.no_grad():
            y = module._forward_fused(x, key, qx, qw, lora, keep_t=A is not None)
        ctx.module, ctx.key = module, key
        ctx.use_lora = A is not None
        t, module._last_t = module._last_t, None
        ctx.save_for_backward(x, t)
        return y

    @staticmethod
    def backward(ctx, g):
        module, key = ctx.module, ctx.key
        x, t_saved = ctx.saved_tensors
        qx, qw, lora = module.quantizers_input[key], module.quantizers_weight[key], module.lora_adapters[key]
        K, N = module.in_features, module.out_features
        need_x, need_W, need_b, need_A, need_B = ctx.needs_input_grad[:5]
        g2 = g.contiguous().float().reshape(-1, N)
        x2 = x.detach().contiguous().reshape(-1, K)
        M = x2.shape[0]
        gx = gW = gb = gA = gB = None

        def ste(grad, q):
            return torch.clamp(grad, -10, 10) if q.quantizer_type == 'log' else grad

        with torch.no_grad():
            lib = _lib.load()
            stream = _lib.stream_ptr(x2.device)
            s = float(lora.scaling) if ctx.use_lora else 0.0
            log_x = qx.quantizer_type == 'log' and qx.num_bits < 32 and not qx.collecting_stats
            limbs = need_x and module.backward_limbs and _LimbGemm.supported(K, N)
            gt = None
            if ctx.use_lora:
                aq = lora.quantize_A(lora.lora_A.detach())                       # [K, r]
                bq = lora.quantize_B(lora.lora_B.detach())                       # [r, N]
                r = aq.shape[1]
                if not (limbs and r <= 128):                                     # else: out of the limb GEMM's activation pass
                    gt = _gemm_nt(g2, bq.contiguous())                           # g . FQ(B)^T  -> [M, r]
            if limbs:
                # g . W_eff on the f16 MFMA kernel: minmax STE is the identity, so both terms share the left operand and
                # W_eff^T = FQ(W)^T + s * FQ(A) . FQ(B)  [K, N] is formed once (0.3 GFLOP); log STE clamps the base term only
                wq_t = module._fq_weight_t(qw)                                   # FQ(W)^T [K, N], kept while W is frozen
                if ctx.use_lora and not log_x:
                    # the rank-r update on this library's fp32-MFMA kernel (no vendor BLAS anywhere on the path)
                    w_t = torch.add(wq_t, _gemm_nt(aq.contiguous(), bq.t().contiguous()), alpha=s)
                else:
                    w_t = wq_t
                if module._bwd_gemm is None:
                    module._bwd_gemm = _LimbGemm()
                if ctx.use_lora and gt is None:
                    gx, gt = module._bwd_gemm(g2, w_t, down=bq, down_scale=_limb_scale(lora.quantize_B)
                                              if lora.quantize_B.num_bits < 32 else None)   # g . FQ(B)^T rides the activation pass
                else:
                    gx = module._bwd_gemm(g2, w_t)
                if log_x:
                    gx = torch.clamp(gx, -10, 10)
                    if ctx.use_lora:
                        gx = gx + s * _gemm_nt(gt, aq.contiguous())
                gx = gx.view(x.shape)
            elif need_x:
                wq_t = qw(module.linear.weight.detach()).t().contiguous()        # FQ(W)^T [K, N], N contiguous
                gx = torch.empty(M, K, dtype=torch.float32, device=x2.device)
                with torch.cuda.device(x2.device):
                    if ctx.use_lora and not log_x:                               # both terms in one launch
                        rc = lib.spq_gemm_f32_nt(g2.data_ptr(), N, wq_t.data_ptr(), N, N, gt.data_ptr(), r,
                                                 aq.contiguous().data_ptr(), r, r, s, None, gx.data_ptr(), K, M, K, stream)
                    else:
                        rc = lib.spq_gemm_f32_nt(g2.data_ptr(), N, wq_t.data_ptr(), N, N, None, 0, None, 0, 0, 1.0, None,
                                                 gx.data_ptr(), K, M, K, stream)
                _lib.check(rc, "spq_gemm_f32_nt(backward)")
                if log_x:                                                        # the clamp applies to the quantizer path only
                    gx = torch.clamp(gx, -10, 10)
                    if ctx.use_lora:
                        gx = gx + s * _gemm_nt(gt, aq.contiguous())
                gx = gx.view(x.shape)
            if ctx.use_lora and need_A:
                gA = ste(_gemm_tn(x2, gt, s), lora.quantize_A)
            if ctx.use_lora and need_B:
                t = t_saved if t_saved is not None else _gemm_nt(x2, aq.t().contiguous())   # x . FQ(A)  -> [M, r]
                gB = ste(_gemm_tn(t, g2, s), lora.quantize_B)
            if need_W:
                xq = qx(x2) if (qx.num_bits < 32 and qx.calibrated and not qx.collecting_stats) else x2
                gW = ste(_gemm_tn(g2, xq.reshape(-1, K).contiguous()), qw)       # g^T . FQ(x) on spq_gemm_f32_tn
            if need_b:
                gb = g2.sum(dim=0)
        return gx, gW, gb, gA, gB, None, None


def _fq_transposed(q: LearnableFakeQuantize, t: torch.Tensor, pad_rows_to: int = 1) -> torch.Tensor:
    """FQ(t)^T for a 2-D LoRA factor whose scale is per column ([1, cols]) or per tensor.  The result may carry
    zero rows up to a multiple of ``pad_rows_to`` (the activation pass copies FQ(A)^T in 64-row pieces)."""
    _lib.require_gpu(t, "LoRA factor")
    rows, cols = t.shape
    out_rows = (cols + pad_rows_to - 1) // pad_rows_to * pad_rows_to
    if q.num_bits >= 32:
        out = torch.zeros(out_rows, rows, dtype=torch.float32, device=t.device)
        out[:cols] = t.t()
        return out
    scale, zero_point = q.qparams_for(cols)
    per_channel = 1 if scale.numel() > 1 else 0
    out = torch.zeros(out_rows, rows, dtype=torch.float32, device=t.device)
    tc = t.contiguous()
    with torch.cuda.device(t.device):
        rc = _lib.load().spq_fakequant_transposed(
            tc.data_ptr(), rows, cols, scale.data_ptr(), zero_point.data_ptr(), per_channel, int(q.num_bits),
            _lib.QTYPE_CODE[q.quantizer_type], 1 if q.symmetric else 0, 1.0, out.data_ptr(),
            _lib.stream_ptr(t.device))
    _lib.check(rc, "spq_fakequant_transposed")
    return out
