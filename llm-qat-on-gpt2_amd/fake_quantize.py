"""``LearnableFakeQuantize`` and the two straight-through quantization functions, on HIP kernels.

Host-side mirror of the reference's ``part1_switchable_precision/quantization.py:15-239`` and
``quantization_methods.py:5-97``: same constructor, attributes, buffers, calibration protocol and error
behaviour; the arithmetic runs in ``libspq.so`` (``spq_minmax_stats``, ``spq_finish_scale``, ``spq_fakequant``).
The module holds only state; it never computes on the CPU.
"""
import math

import torch
import torch.nn as nn

from . import _lib

_BUFFERS = ("scale", "zero_point", "running_min", "running_max")


def _chan_view(shape, chan_axis):
    """shape -> (outer, chan, inner) around ``chan_axis`` (None: everything is 'inner' of one channel)."""
    if chan_axis is None:
        return 1, 1, max(1, math.prod(shape))
    return (max(1, math.prod(shape[:chan_axis])), shape[chan_axis], max(1, math.prod(shape[chan_axis + 1:])))


def _param_axis(x_shape, p_shape):
    """Where a keep-dim scale tensor varies when broadcast against x (None = per-tensor)."""
    axes = [i for i, n in enumerate(p_shape) if n != 1]
    if not axes:
        return None
    if len(axes) > 1:
        raise ValueError(f"scale of shape {tuple(p_shape)} varies along more than one axis")
    ax = axes[0] - len(p_shape) + len(x_shape)           # right-aligned broadcasting
    if ax < 0 or x_shape[ax] != p_shape[axes[0]]:
        raise RuntimeError(f"scale of shape {tuple(p_shape)} does not broadcast against input {tuple(x_shape)}")
    return ax


_eps_cache = {}


def _eps_constants(eps):
    """(fp32(eps), fp32 log2(fp32(eps))) as Python floats -- the values torch.tensor(eps) / torch.log2 give the reference."""
    v = _eps_cache.get(eps)
    if v is None:
        e32 = torch.tensor(eps, dtype=torch.float32)
        v = _eps_cache[eps] = (float(e32), float(torch.log2(e32)))
    return v


def fake_quantize(x, scale, zero_point, num_bits, qtype, symmetric, want_levels=False):
    """One launch of ``spq_fakequant``: returns the dequantised tensor (and int32 levels on request).

    Output shape is ``broadcast(x.shape, scale.shape)``, as the reference's ``x / scale`` gives
    (a 3-D keep-dim scale lifts a 2-D input to 3-D; SURVEY.md §7 hard part 5).
    """
    _lib.require_gpu(x, "fake-quant input")
    _lib.check_device(x.device)
    x = x.contiguous()
    if scale.device != x.device or scale.dtype != torch.float32 or not scale.is_contiguous():
        scale = scale.to(device=x.device, dtype=torch.float32).contiguous()
    if zero_point.device != x.device or zero_point.dtype != torch.float32 or not zero_point.is_contiguous():
        zero_point = zero_point.to(device=x.device, dtype=torch.float32).contiguous()
    out_shape = torch.broadcast_shapes(x.shape, scale.shape)
    tail = tuple(out_shape[len(out_shape) - x.dim():])
    if tail != tuple(x.shape):                      # the scale expands a size-1 axis of x
        x = x.expand(tail).contiguous()
    ax = _param_axis(x.shape, scale.shape)
    outer, chan, inner = _chan_view(x.shape, ax)
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    levels = torch.empty(x.shape, dtype=torch.int32, device=x.device) if want_levels else None
    if x.numel():
        with torch.cuda.device(x.device):
            rc = _lib.load().spq_fakequant(x.data_ptr(), outer, chan, inner, scale.data_ptr(), zero_point.data_ptr(),
                                           0 if ax is None else 1, int(num_bits), _lib.QTYPE_ANY[qtype],
                                           1 if symmetric else 0, out.data_ptr(), _lib.ptr(levels), 4,
                                           _lib.stream_ptr(x.device))
        _lib.check(rc, "spq_fakequant")
    out = out.view(out_shape)
    if want_levels:
        return out, levels.view(out_shape)
    return out


class MinMaxQuantizationFunction(torch.autograd.Function):
    """Forward: quantization_methods.py:8-22 in one kernel.  Backward: un-masked straight-through (:25-28)."""

    @staticmethod
    def forward(ctx, input, scale, zero_point, num_bits, symmetric):
        return fake_quantize(input, scale, zero_point, num_bits, "minmax", symmetric)

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output.clone(), None, None, None, None


class LogQuantizationFunction(torch.autograd.Function):
    """Forward: quantization_methods.py:33-79.  Backward: straight-through clamped to [-10, 10] (:82-90)."""

    @staticmethod
    def forward(ctx, input, log_min, log_range, num_bits, symmetric):
        return fake_quantize(input, log_range, log_min, num_bits, "log", symmetric)

    @staticmethod
    def backward(ctx, grad_output):
        return torch.clamp(grad_output, -10, 10), None, None, None, None


def apply_minmax_quantization(x, scale, zero_point, num_bits, symmetric=True):
    return MinMaxQuantizationFunction.apply(x, scale, zero_point, num_bits, symmetric)


def apply_log_quantization(x, log_min, log_range, num_bits, symmetric=True):
    return LogQuantizationFunction.apply(x, log_min, log_range, num_bits, symmetric)


class LearnableFakeQuantize(nn.Module):
    """Drop-in for the reference class of the same name (quantization.py:15)."""

    def __init__(self, num_bits, channel_dim=0, quantizer_type='minmax', eps=1e-5, symmetric=True,
                 per_channel=True, is_input=False):
        super().__init__()
        self.num_bits = max(1, min(num_bits, 32))
        self.symmetric = symmetric
        self.per_channel = per_channel
        self.channel_dim = channel_dim if per_channel else None
        self.quantizer_type = quantizer_type
        self.eps = eps
        self.is_input = is_input
        self._update_quant_range()

        self.register_buffer('scale', torch.ones(1))
        self.register_buffer('zero_point', torch.zeros(1))
        self.register_buffer('running_min', torch.zeros(1))
        self.register_buffer('running_max', torch.zeros(1))

        self.calibrated = False
        self.collecting_stats = False
        self.num_batches_collected = 0
        self.temp_min = None
        self.temp_max = None
        # bumped whenever scale/zero_point may have changed; prepared GEMM operands key on it
        self._epoch = 0

    # ---- bookkeeping (quantization.py:77-102) -------------------------------------------------------
    def _update_quant_range(self):
        if self.symmetric:
            self.quant_min, self.quant_max = -(2 ** (self.num_bits - 1)), 2 ** (self.num_bits - 1) - 1
        else:
            self.quant_min, self.quant_max = 0, 2 ** self.num_bits - 1

    def set_num_bits(self, value):
        previous = self.num_bits
        self.num_bits = max(1, min(value, 32))
        self._update_quant_range()
        if previous != self.num_bits:
            print(f"    Reset calibration for {self.quantizer_type} quantizer: {previous} -> {self.num_bits} bits")
            self.calibrated = False
            self._epoch += 1

    def start_calibration(self):
        self.collecting_stats = True
        self.calibrated = False
        self.num_batches_collected = 0
        self.temp_min = None
        self.temp_max = None

    # ---- statistics (quantization.py:141-209) -> spq_minmax_stats ----------------------------------
    def _stat_axis(self, ndim):
        if not (self.per_channel and self.channel_dim is not None):
            return None
        ax = self.channel_dim if self.channel_dim >= 0 else ndim + self.channel_dim
        return ax if 0 <= ax < ndim else None

    def _collect_statistics_batch(self, x):
        _lib.require_gpu(x, "calibration input")
        _lib.check_device(x.device)
        if self.quantizer_type not in _lib.QTYPE_CODE:
            raise ValueError(f"Unknown quantizer type: {self.quantizer_type}. Supported types: 'minmax', 'log'")
        with torch.no_grad():
            xc = x.detach().contiguous()
            ax = self._stat_axis(xc.dim())
            outer, chan, inner = _chan_view(xc.shape, ax)
            first = self.num_batches_collected == 0 or self.temp_min is None
            if first:
                keep = [1] * xc.dim()
                if ax is not None:
                    keep[ax] = xc.shape[ax]
                self.temp_min = torch.empty(keep, dtype=torch.float32, device=xc.device)
                self.temp_max = torch.empty(keep, dtype=torch.float32, device=xc.device)
            elif self.temp_min.numel() != (chan if ax is not None else 1):
                raise RuntimeError("calibration batches disagree on the channel count")
            is_log = self.quantizer_type == 'log'
            eps32, log2_eps = _eps_constants(self.eps)
            lib = _lib.load()
            need = lib.spq_stats_workspace_bytes(outer, chan, inner, 0 if ax is None else 1)
            ws = _lib.workspace(xc.device, need)
            with torch.cuda.device(xc.device):
                rc = lib.spq_minmax_stats(xc.data_ptr(), outer, chan, inner, 0 if ax is None else 1,
                                          1 if is_log else 0, eps32, log2_eps,
                                          1 if first else 0, self.temp_min.data_ptr(), self.temp_max.data_ptr(),
                                          ws.data_ptr(), ws.numel(), _lib.stream_ptr(xc.device))
            _lib.check(rc, "spq_minmax_stats")
            self.num_batches_collected += 1

    # ---- scale derivation (quantization.py:104-139) -> spq_finish_scale -------------------------------
    def finish_calibration(self, debug=False):
        if self.num_batches_collected > 0 and self.temp_min is not None:
            dev = self.temp_min.device
            with torch.no_grad():
                for name, src in (("running_min", self.temp_min), ("running_max", self.temp_max)):
                    setattr(self, name, src.detach().clone())
                scale = torch.empty_like(self.running_min)
                zp = torch.empty_like(self.running_min)
                _lib.require_gpu(self.running_min, "calibration statistics")
                with torch.cuda.device(dev):
                    rc = _lib.load().spq_finish_scale(
                        self.running_min.data_ptr(), self.running_max.data_ptr(), self.running_min.numel(),
                        int(self.num_bits), _lib.QTYPE_CODE[self.quantizer_type], 1 if self.symmetric else 0,
                        _eps_constants(self.eps)[0], scale.data_ptr(), zp.data_ptr(),
                        _lib.stream_ptr(dev))
                _lib.check(rc, "spq_finish_scale")
                self.scale = scale
                self.zero_point = zp
                if debug:
                    print(f"         Computed scale: mean={self.scale.mean().item():.6f}")
            self.calibrated = True
            self._epoch += 1
            self.collecting_stats = False
            self.temp_min = None
            self.temp_max = None
        else:
            self.collecting_stats = False
            if debug:
                print(f"      ⚠️ No statistics collected for {self.num_bits}-bit {self.quantizer_type} quantizer")

    # ---- checkpoint hook (quantization.py:40-75): buffers are variable-shape ---------------------------
    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        for name in _BUFFERS:
            key = prefix + name
            if key not in state_dict or getattr(self, name, None) is None:
                continue
            incoming = state_dict[key]
            if self.is_input and incoming.dim() == 3 and incoming.shape[1] > 1:
                # legacy per-position statistics [*, T, *]: collapse T
                take_min = self.quantizer_type != 'log' and 'min' in name
                incoming = (incoming.min(dim=1, keepdim=True)[0] if take_min
                            else incoming.max(dim=1, keepdim=True)[0])
                state_dict[key] = incoming
            getattr(self, name).resize_as_(incoming)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)
        if prefix + 'scale' in state_dict and prefix + 'zero_point' in state_dict:
            self.calibrated = True
        self._epoch += 1

    # ---- forward (quantization.py:211-239) ---------------------------------------------------------------
    def forward(self, x):
        if self.num_bits >= 32:
            return x
        if self.collecting_stats:
            self._collect_statistics_batch(x)
            return x
        if not self.calibrated:
            raise RuntimeError(
                f"Quantizer not calibrated. Please run calibration first for {self.quantizer_type} quantizer.")
        if self.quantizer_type == 'minmax':
            return self._quantize_minmax(x)
        elif self.quantizer_type == 'log':
            return self._quantize_log(x)
        raise ValueError(f"Unknown quantizer type: {self.quantizer_type}. Supported types: 'minmax', 'log'")

    def _quantize_minmax(self, x):
        return apply_minmax_quantization(x, self.scale, self.zero_point, self.num_bits, self.symmetric)

    def _quantize_log(self, x):
        # note the argument order: zero_point carries log_min, scale carries log_range (:237-239)
        return apply_log_quantization(x, self.zero_point, self.scale, self.num_bits, self.symmetric)

    def qparams_for(self, n_channels):
        """``(scale, zero_point)`` in the layout the fused kernels take: one value, or one per channel (``n_channels``).

        Any other shape can only be the reference's log default-fill state (quantization.py:164-172,194-197: a tensor with
        no ``|x| > eps`` -- the zero-initialised ``lora_B`` of lora.py:38 -- gets statistics of shape ``x.shape`` with the
        channel axis set to 1, e.g. ``[r,1]``, every entry ``log2(eps)``), which arrives here through a reference checkpoint.
        Its entries are all equal, so it is the per-tensor quantizer of that value; anything else is refused."""
        s, z = self.scale, self.zero_point
        if s.numel() in (1, n_channels) and z.numel() == s.numel():
            return s, z
        key = (self._epoch, s.data_ptr(), s._version, z.data_ptr(), z._version)
        cached = getattr(self, "_uniform_qparams", None)
        if cached is None or cached[0] != key:
            s1, z1 = s.reshape(-1)[:1].contiguous(), z.reshape(-1)[:1].contiguous()
            uniform = bool(((s == s1).all() & (z == z1).all()).item())     # one host sync per loaded checkpoint
            cached = self._uniform_qparams = (key, (s1, z1) if uniform else None)
        if cached[1] is None:
            raise RuntimeError(f"scale of shape {tuple(s.shape)} does not fit {n_channels} channels")
        return cached[1]

    def quantize_levels(self, x):
        """Integer levels before dequantisation (int32), for level-exactness tests and INT8 export."""
        if not self.calibrated:
            raise RuntimeError(
                f"Quantizer not calibrated. Please run calibration first for {self.quantizer_type} quantizer.")
        return fake_quantize(x, self.scale, self.zero_point, self.num_bits, self.quantizer_type, self.symmetric,
                             want_levels=True)[1]
