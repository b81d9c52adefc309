"""The producers and consumers either side of the quantized linears inside a transformer block:
``SwitchableLayerNorm`` (the tensor c_attn / c_fc quantize) as one HIP pass, and ``SPMLP`` with the GELU between c_fc and
c_proj applied inside c_fc's contraction store.  Drop-ins for the reference's classes of the same names
(part1_switchable_precision/switchable_batchnorm.py:6-109, models_sp.py:76-128); the reference's own ``SPBlock`` /
``SPAttention`` / ``SPModel`` assemble them unchanged (rebind the two names in models_sp).
"""
from typing import List, Union

import torch
import torch.nn as nn

from . import _lib
from .sp_linear import SPLinearWithLoRA


class _ParamView:
    """``ln_layers[key].weight`` / ``.bias`` of the reference (switchable_batchnorm.py:36-89): ``.data`` and ``.requires_grad``
    forwarded to the parameter of that precision."""

    def __init__(self, param):
        self._param = param

    @property
    def data(self):
        return self._param.data

    @data.setter
    def data(self, value):
        self._param.data = value

    @property
    def requires_grad(self):
        return self._param.requires_grad

    @requires_grad.setter
    def requires_grad(self, value):
        self._param.requires_grad = value


class _LayerNormCompat:
    def __init__(self, parent, key):
        self.parent, self.key = parent, key

    @property
    def weight(self):
        return _ParamView(self.parent.weights[self.key])

    @property
    def bias(self):
        return _ParamView(self.parent.biases[self.key])


class SwitchableLayerNorm(nn.Module):
    """One weight/bias pair per precision over a shared normalisation (switchable_batchnorm.py:6-109)."""

    def __init__(self, normalized_shape: Union[int, List[int], torch.Size], precision_levels: List[int] = [6, 8, 16, 32],
                 eps: float = 1e-5):
        super().__init__()
        if isinstance(normalized_shape, int):
            normalized_shape = (normalized_shape,)
        self.normalized_shape = tuple(normalized_shape)
        self.precision_levels = sorted(precision_levels)
        self.eps = eps
        self.weights = nn.ParameterDict()
        self.biases = nn.ParameterDict()
        for precision in self.precision_levels:
            self.weights[str(precision)] = nn.Parameter(torch.ones(normalized_shape))
            self.biases[str(precision)] = nn.Parameter(torch.zeros(normalized_shape))
        self.current_precision = max(self.precision_levels)
        self.ln_layers = {str(p): _LayerNormCompat(self, str(p)) for p in self.precision_levels}

    def set_precision(self, precision: int) -> int:
        if precision not in self.precision_levels:
            raise ValueError(f"Precision {precision} not supported. Available: {self.precision_levels}")
        self.current_precision = precision
        return self.current_precision

    def _composed(self, x, weight, bias):
        """The reference's formula on stock torch ops (autograd-capable; also what a CPU tensor gets)."""
        dims = [-(i + 1) for i in range(len(self.normalized_shape))]
        mean = x.mean(dim=dims, keepdim=True)
        var = x.var(dim=dims, keepdim=True, unbiased=False)
        return weight * ((x - mean) / torch.sqrt(var + self.eps)) + bias

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        weight = self.weights[str(self.current_precision)]
        bias = self.biases[str(self.current_precision)]
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or bias.requires_grad)
        cols = self.normalized_shape[0]
        fused = (x.is_cuda and not needs_grad and len(self.normalized_shape) == 1 and x.dtype == torch.float32
                 and cols % 4 == 0 and cols <= 8192 and x.numel() > 0)
        if not fused:
            return self._composed(x, weight, bias)
        _lib.check_device(x.device)
        xc = x.contiguous()
        out = torch.empty_like(xc)
        with torch.cuda.device(x.device):
            rc = _lib.load().spq_layernorm(xc.data_ptr(), xc.numel() // cols, cols, weight.data_ptr(), bias.data_ptr(),
                                           float(self.eps), out.data_ptr(), _lib.stream_ptr(x.device))
        _lib.check(rc, "spq_layernorm")
        return out


class SPMLP(nn.Module):
    """models_sp.py:76-128: c_fc -> GELU -> c_proj on the drop-in linears, the GELU fused into c_fc's store."""

    def __init__(self, config, bit_widths=None):
        super().__init__()
        if bit_widths is None:
            bit_widths = getattr(config, 'bit_widths', [6, 8, 16, 32])
        self.bit_widths = bit_widths
        try:
            lora_rank_per_bit = config.lora_rank_per_bit
            lora_alpha_per_bit = config.lora_alpha_per_bit
        except AttributeError as e:
            raise AttributeError(f"Config missing required switchable precision attributes: {e}\n"
                                 "Required: lora_rank_per_bit, lora_alpha_per_bit")
        quantizer_per_bit = getattr(config, 'quantizer_per_bit', None)
        per_channel = getattr(config, 'per_channel_quantization', True)
        common = dict(bit_widths=bit_widths, lora_rank_per_bit=lora_rank_per_bit, lora_alpha_per_bit=lora_alpha_per_bit,
                      quantizer_per_bit=quantizer_per_bit, per_channel=per_channel)
        self.c_fc = SPLinearWithLoRA(config.n_embd, 4 * config.n_embd, **common)
        self.c_proj = SPLinearWithLoRA(4 * config.n_embd, config.n_embd, **common)
        self.act = nn.GELU()

    def set_precision(self, bits) -> int:
        if bits not in self.bit_widths:
            raise ValueError(f"Bit width {bits} not in configured widths {self.bit_widths}")
        self.c_fc.set_precision(bits)
        self.c_proj.set_precision(bits)
        return bits

    def forward(self, hidden_states, pre_norm=None):
        """``pre_norm``: the LayerNorm whose output the MLP consumes (ln_2); ``hidden_states`` is then its input and c_fc's
        activation pass applies it (SPLinearWithLoRA.forward)."""
        hidden_states = self.c_fc(hidden_states, activation='gelu', pre_norm=pre_norm)   # LN + c_fc + GELU: two launches
        return self.c_proj(hidden_states)


class SPAttention(nn.Module):
    """models_sp.py:18-74: fused QKV projection, causal softmax attention (stock torch-ROCm ops, not on the quantized
    path), output projection -- on the drop-in linears."""

    def __init__(self, config, bit_widths):
        super().__init__()
        self.n_head = config.n_head
        self.n_embd = config.n_embd
        self.head_dim = self.n_embd // self.n_head
        self.bit_widths = bit_widths
        common = dict(bit_widths=bit_widths, lora_rank_per_bit=config.lora_rank_per_bit,
                      lora_alpha_per_bit=config.lora_alpha_per_bit, quantizer_per_bit=config.quantizer_per_bit,
                      per_channel=getattr(config, 'per_channel_quantization', True))
        self.c_attn = SPLinearWithLoRA(config.n_embd, 3 * config.n_embd, **common)
        self.c_proj = SPLinearWithLoRA(config.n_embd, config.n_embd, **common)
        self.register_buffer("bias", torch.tril(torch.ones(config.n_positions, config.n_positions)))
        self.use_sdpa = True        # False: the reference's explicit q k^T / softmax / v formula (models_sp.py:66-71)

    def set_precision(self, bits) -> int:
        self.current_bit_width = bits
        self.c_attn.set_precision(bits)
        self.c_proj.set_precision(bits)
        return self.current_bit_width

    def forward(self, hidden_states, attention_mask=None, pre_norm=None):
        return self.c_proj(self.core(self.c_attn(hidden_states, pre_norm=pre_norm)))

    def core(self, qkv):
        """models_sp.py:61-75: what sits between the two projections (split heads, causal softmax attention, merge heads)."""
        B, T, _ = qkv.shape
        C = self.n_embd
        q, k, v = qkv.split(self.n_embd, dim=2)
        q = q.view(B, T, self.n_head, self.head_dim).transpose(1, 2)
        k = k.view(B, T, self.n_head, self.head_dim).transpose(1, 2)
        v = v.view(B, T, self.n_head, self.head_dim).transpose(1, 2)
        if self.use_sdpa and q.is_cuda:
            # same causal softmax attention through torch-ROCm's fused kernel (no B x H x T x T matrix in HBM): 0.86 ms vs
            # 3.4 ms at 32 x 12 x 1024 x 64 fp32, |difference| 1.5e-6 against the explicit formula below
            out = torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=True)
        else:
            att = (q @ k.transpose(-2, -1)) / (self.head_dim ** 0.5)
            att = att.masked_fill(self.bias[:T, :T].to(att.device) == 0, float('-inf'))
            out = torch.softmax(att, dim=-1) @ v
        return out.transpose(1, 2).contiguous().view(B, T, C)


class SPBlock(nn.Module):
    """models_sp.py:130-171: pre-LN transformer block on SwitchableLayerNorm, SPAttention and SPMLP."""

    def __init__(self, config, bit_widths):
        super().__init__()
        self.ln_1 = SwitchableLayerNorm(config.n_embd, precision_levels=bit_widths, eps=config.layer_norm_epsilon)
        self.attn = SPAttention(config, bit_widths)
        self.ln_2 = SwitchableLayerNorm(config.n_embd, precision_levels=bit_widths, eps=config.layer_norm_epsilon)
        self.mlp = SPMLP(config, bit_widths)

    def set_precision(self, bits) -> int:
        self.ln_1.set_precision(bits)
        self.attn.set_precision(bits)
        self.ln_2.set_precision(bits)
        self.mlp.set_precision(bits)
        return bits

    def forward(self, hidden_states, attention_mask=None, use_checkpoint=False):
        if use_checkpoint:
            from torch.utils.checkpoint import checkpoint
            return checkpoint(self._forward, hidden_states, attention_mask)
        return self._forward(hidden_states, attention_mask)

    def _forward(self, hidden_states, attention_mask=None):
        # models_sp.py:160-171.  ln_1 / ln_2 are handed to their only consumers (c_attn, c_fc), whose activation pass applies
        # them on the fly where it can (the normalised tensor is then never stored); same values as calling them here.
        hidden_states = hidden_states + self.attn(hidden_states, attention_mask, pre_norm=self.ln_1)
        return hidden_states + self.mlp(hidden_states, pre_norm=self.ln_2)


class SPModel(nn.Module):
    """models_sp.py:173-330: token + position embeddings, ``n_layer`` x SPBlock, final SwitchableLayerNorm -- the caller
    that hosts BASELINE configs 4 / 5.  Same attribute names (``wte, wpe, drop, h, ln_f``) and state-dict keys as the
    reference's class; ``config`` needs ``vocab_size, n_positions, n_embd, n_layer, n_head, layer_norm_epsilon, bit_widths,
    lora_rank_per_bit, lora_alpha_per_bit, quantizer_per_bit`` (``embd_pdrop`` and ``per_channel_quantization`` optional).
    The lm_head of ``SPLMHeadModel`` (models_sp.py:390-458) is a plain tied ``nn.Linear`` outside the quantized path and is
    not rebuilt here."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.bit_widths = config.bit_widths
        self.current_bit_width = max(self.bit_widths)
        self.wte = nn.Embedding(config.vocab_size, config.n_embd)
        self.wpe = nn.Embedding(config.n_positions, config.n_embd)
        self.drop = nn.Dropout(getattr(config, 'embd_pdrop', 0.0))
        self.h = nn.ModuleList([SPBlock(config, bit_widths=self.bit_widths) for _ in range(config.n_layer)])
        self.ln_f = SwitchableLayerNorm(config.n_embd, precision_levels=self.bit_widths, eps=config.layer_norm_epsilon)

    def set_precision(self, bits) -> int:
        if bits not in self.bit_widths:
            raise ValueError(f"Bit width {bits} not in configured widths {self.bit_widths}")
        self.current_bit_width = bits
        for block in self.h:
            block.set_precision(bits)
        self.ln_f.set_precision(bits)
        return self.current_bit_width

    def get_current_precision(self):
        return self.current_bit_width

    def disable_lora_for_calibration(self):
        for module in self.modules():
            if module.__class__.__name__ == 'SPLinearWithLoRA':
                module.calibration_mode = True

    def enable_lora_after_calibration(self):
        for module in self.modules():
            if module.__class__.__name__ == 'SPLinearWithLoRA':
                module.calibration_mode = False

    def forward(self, input_ids=None, inputs_embeds=None, attention_mask=None, use_checkpoint=False,
                output_hidden_states=False):
        if inputs_embeds is not None:
            hidden_states = inputs_embeds
        else:
            if input_ids is None:
                raise ValueError("Either input_ids or inputs_embeds must be provided")
            T = input_ids.shape[1]
            position_ids = torch.arange(0, T, dtype=torch.long, device=input_ids.device).unsqueeze(0)
            hidden_states = self.drop(self.wte(input_ids) + self.wpe(position_ids))
        all_hidden_states = [] if output_hidden_states else None
        for block in self.h:
            if output_hidden_states:
                all_hidden_states.append(hidden_states.clone().detach())
            hidden_states = block(hidden_states, attention_mask, use_checkpoint)
        hidden_states = self.ln_f(hidden_states)
        if output_hidden_states:
            all_hidden_states.append(hidden_states.clone().detach())
            return hidden_states, all_hidden_states
        return hidden_states


class SPLMHeadModel(nn.Module):
    """models_sp.py:390-458: ``transformer`` (SPModel) + the tied, un-quantized ``lm_head`` -- the object the reference's
    checkpoints describe (state-dict keys ``transformer.*`` + ``lm_head.weight``, deploy.py:143, main_sp_eval.py:70)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.transformer = SPModel(config)
        self.lm_head = nn.Linear(config.n_embd, config.vocab_size, bias=False)
        self.lm_head.weight = self.transformer.wte.weight

    def set_precision(self, bits) -> int:
        return self.transformer.set_precision(bits)

    def get_current_precision(self):
        return self.transformer.get_current_precision()

    def disable_lora_for_calibration(self):
        self.transformer.disable_lora_for_calibration()

    def enable_lora_after_calibration(self):
        self.transformer.enable_lora_after_calibration()

    def forward(self, input_ids=None, inputs_embeds=None, labels=None, attention_mask=None, use_checkpoint=False,
                output_hidden_states=False, return_dict=False):
        out = self.transformer(input_ids, inputs_embeds=inputs_embeds, attention_mask=attention_mask,
                               use_checkpoint=use_checkpoint, output_hidden_states=output_hidden_states)
        hidden_states, all_hidden_states = out if output_hidden_states else (out, None)
        logits = self.lm_head(hidden_states)
        loss = None
        if labels is not None:
            shift_logits = logits[..., :-1, :].contiguous()
            shift_labels = labels[..., 1:].contiguous()
            loss = nn.functional.cross_entropy(shift_logits.view(-1, shift_logits.size(-1)), shift_labels.view(-1))
        if return_dict or output_hidden_states:
            return {'loss': loss, 'logits': logits, 'hidden_states': all_hidden_states}
        return {'loss': loss, 'logits': logits} if loss is not None else logits
