"""Checkpoint wire format and INT8 export of the switchable-precision model.

Mirrors ``part1_switchable_precision/deploy.py``: ``save_sp_checkpoints`` (:125-183) writes one ``.pth`` per student
bit-width holding the WHOLE ``state_dict`` plus config dicts; the evaluation loaders (``deploy.py:185-253``,
``part3_eval_sp/main_sp_eval.py:22-78``) rebuild the model with ``per_channel_quantization=False`` and load it with
``strict=True`` -- the variable-shape quantizer buffers come from the file (``LearnableFakeQuantize._load_from_state_dict``).
``convert_to_int8`` (:5-62) is the abs-max INT8 export of every quantized linear; here its levels come straight out of
``spq_fakequant``'s int8 output (one statistics pass + one quantize pass per weight, both HIP).
"""
import collections
import io
import pickle
import time
import types
import zipfile

import torch

from . import _lib
from .blocks import SPLMHeadModel


def _int_keys(d):
    return {int(k) if isinstance(k, str) and k.lstrip('-').isdigit() else k: v for k, v in d.items()} if isinstance(d, dict) else d


def _minmax_per_tensor(w):
    lib = _lib.load()
    mn = torch.empty(1, dtype=torch.float32, device=w.device)
    mx = torch.empty(1, dtype=torch.float32, device=w.device)
    ws = _lib.workspace(w.device, lib.spq_stats_workspace_bytes(1, 1, w.numel(), 0))
    with torch.cuda.device(w.device):
        rc = lib.spq_minmax_stats(w.data_ptr(), 1, 1, w.numel(), 0, 0, 0.0, 0.0, 1, mn.data_ptr(), mx.data_ptr(),
                                  ws.data_ptr(), ws.numel(), _lib.stream_ptr(w.device))
    _lib.check(rc, "spq_minmax_stats")
    return mn, mx


def _levels(w, scale, symmetric, nbytes):
    lib = _lib.load()
    out = torch.empty(w.shape, dtype={1: torch.int8, 2: torch.int16}[nbytes], device=w.device)
    zero = torch.zeros(1, dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        rc = lib.spq_fakequant(w.data_ptr(), 1, 1, w.numel(), scale.data_ptr(), zero.data_ptr(), 0, 8, _lib.QTYPE_ANY["minmax"],
                               1 if symmetric else 0, None, out.data_ptr(), nbytes, _lib.stream_ptr(w.device))
    _lib.check(rc, "spq_fakequant(levels)")
    return out


def convert_to_int8(model):
    """deploy.py:5-62 on the device: for every ``SPLinearWithLoRA`` the frozen weight as INT8 levels of its own abs-max
    scale (symmetric: ``round(w / (max|w| / 127))``; asymmetric: ``round((w - min) / ((max - min) / 255))`` as uint8),
    plus ``scale``, ``zero_point`` and ``bias``; returned on the CPU like the reference's dict."""
    model.eval()
    out = {}
    with torch.no_grad():
        for name, module in model.named_modules():
            if 'SPLinearWithLoRA' not in module.__class__.__name__ or not hasattr(module, 'linear'):
                continue
            if not hasattr(module, 'quantizers_weight'):
                continue
            bits = getattr(module, 'current_bits', 8)
            quantizer = module.quantizers_weight[f'{bits}bit']
            w = module.linear.weight.data.contiguous()
            _lib.require_gpu(w, "weight")
            _lib.check_device(w.device)
            mn, mx = _minmax_per_tensor(w)
            if quantizer.symmetric:
                max_val = torch.maximum(-mn, mx)                                   # == w.abs().max()
                scale = torch.where(max_val > 0, max_val / 127.0, torch.ones_like(max_val))
                zero_point = torch.tensor(0, dtype=torch.int32)
                q = _levels(w, scale, True, 1)                                      # |round(w/scale)| <= 127: the clamp never bites
            else:
                span = mx - mn
                scale = torch.where(span > 0, span / 255.0, torch.ones_like(span))
                zero_point = torch.round(-mn / scale).clamp(0, 255).to(torch.int32).reshape(())
                q = _levels((w - mn).contiguous(), scale, False, 2).to(torch.uint8)
            prefix = f"{name}." if name else ""
            out[f"{prefix}weight_int8"] = q.cpu()
            out[f"{prefix}scale"] = scale.reshape(()).cpu()
            out[f"{prefix}zero_point"] = zero_point.cpu()
            if module.linear.bias is not None:
                out[f"{prefix}bias"] = module.linear.bias.data.cpu()
    return out


def save_sp_checkpoints(model, base_filename, model_config, training_config=None):
    """deploy.py:125-183: one file per student bit-width, each with the whole state_dict (same keys and dict layout; written
    with torch's default pickle protocol so ``torch.load(weights_only=True)`` reads it back)."""
    saved = {}
    timestamp = time.strftime('%Y%m%d_%H%M%S')
    for bits in getattr(model_config, 'bit_widths', [6, 8, 16, 32]):
        if bits == 32:
            continue
        model.set_precision(bits)
        filename = f"{base_filename}_{bits}bit_FP32_{timestamp}.pth"
        torch.save({'model_state_dict': model.state_dict(), 'model_config': dict(model_config.__dict__),
                    'training_config': dict(training_config.__dict__) if training_config else None,
                    'bit_width': bits, 'timestamp': timestamp}, filename)
        saved[bits] = filename
    return saved


# ---- a data-only reader for torch zip checkpoints of any pickle protocol ------------------------------------------------------
# The reference writes its checkpoints with ``pickle_protocol=4`` (deploy.py:152); torch's ``weights_only`` unpickler refuses that
# protocol's opcodes, and ``weights_only=False`` would run whatever the file says.  This reader runs nothing from the file: the
# only globals it resolves are the handful a state dict is made of, and tensor bytes come from the archive's ``data/`` records.
_SAFE_GLOBALS = {
    ("collections", "OrderedDict"): collections.OrderedDict,
    ("torch._utils", "_rebuild_tensor_v2"): torch._utils._rebuild_tensor_v2,
    ("torch._utils", "_rebuild_parameter"): torch._utils._rebuild_parameter,
    ("torch", "Size"): torch.Size,
}
_STORAGE_DTYPES = {
    "FloatStorage": torch.float32, "DoubleStorage": torch.float64, "HalfStorage": torch.float16, "BFloat16Storage": torch.bfloat16,
    "LongStorage": torch.int64, "IntStorage": torch.int32, "ShortStorage": torch.int16, "CharStorage": torch.int8,
    "ByteStorage": torch.uint8, "BoolStorage": torch.bool,
}


class _StorageTag:
    """what ``torch.FloatStorage`` & co. unpickle to: just the dtype of the storage the persistent id refers to"""
    def __init__(self, dtype):
        self.dtype = dtype


class _DataOnlyUnpickler(pickle.Unpickler):
    def __init__(self, file, read_record):
        super().__init__(file)
        self._read_record = read_record
        self._storages = {}

    def find_class(self, module, name):
        if module == "torch" and name in _STORAGE_DTYPES:
            return _StorageTag(_STORAGE_DTYPES[name])
        try:
            return _SAFE_GLOBALS[(module, name)]
        except KeyError:
            raise pickle.UnpicklingError(f"checkpoint refers to {module}.{name}, which a state dict has no business with") from None

    def persistent_load(self, pid):
        # ('storage', <storage type>, key, location, numel): torch.serialization._save
        if not (isinstance(pid, tuple) and len(pid) == 5 and pid[0] == "storage" and isinstance(pid[1], _StorageTag)):
            raise pickle.UnpicklingError(f"unexpected persistent id {pid!r}")
        _, tag, key, _location, numel = pid
        if key not in self._storages:
            raw = self._read_record(str(key))
            nbytes = int(numel) * torch.empty((), dtype=tag.dtype).element_size()
            if len(raw) < nbytes:
                raise pickle.UnpicklingError(f"storage {key}: {len(raw)} bytes in the archive, {nbytes} expected")
            untyped = torch.frombuffer(bytearray(raw[:nbytes]) if nbytes else bytearray(1), dtype=torch.uint8)[:nbytes].untyped_storage()
            self._storages[key] = torch.storage.TypedStorage(wrap_storage=untyped, dtype=tag.dtype, _internal=True)
        return self._storages[key]


def load_checkpoint_data_only(path):
    """Read a ``torch.save`` zip archive (any pickle protocol) without executing anything from it: dicts, lists, numbers,
    strings and CPU tensors only; any other global in the pickle stream raises ``pickle.UnpicklingError``."""
    with zipfile.ZipFile(path) as z:
        names = z.namelist()
        pkl = [n for n in names if n.endswith("/data.pkl") or n == "data.pkl"]
        if len(pkl) != 1:
            raise pickle.UnpicklingError(f"{path}: not a torch zip checkpoint (data.pkl entries: {pkl})")
        prefix = pkl[0][:-len("data.pkl")]
        if prefix + "byteorder" in names and z.read(prefix + "byteorder").strip() not in (b"little", b""):
            raise pickle.UnpicklingError(f"{path}: big-endian checkpoint")
        with z.open(pkl[0]) as f:
            return _DataOnlyUnpickler(io.BytesIO(f.read()), lambda key: z.read(prefix + "data/" + key)).load()


def load_sp_checkpoint(path, device='cuda', target_bits=None, weights_only=True):
    """The evaluation loader (main_sp_eval.py:22-78, deploy.py:185-253): build an ``SPLMHeadModel`` from the checkpoint's
    ``model_config`` with ``per_channel_quantization=False``, ``set_precision(bit_width)``, ``load_state_dict(strict=True)``.

    ``weights_only=True`` executes nothing from the file: torch's safe unpickler first, and -- for the files the reference itself
    writes, which use pickle protocol 4 (deploy.py:152) and which that unpickler refuses -- ``load_checkpoint_data_only``."""
    if weights_only:
        try:
            ck = torch.load(path, map_location='cpu', weights_only=True)
        except pickle.UnpicklingError:
            ck = load_checkpoint_data_only(path)
    else:
        ck = torch.load(path, map_location='cpu', weights_only=False)
    mc = ck['model_config']
    sd = ck.get('model_state_dict', ck)
    n_positions = sd['transformer.wpe.weight'].shape[0] if 'transformer.wpe.weight' in sd else mc.get('n_positions', 1024)
    cfg = types.SimpleNamespace(
        vocab_size=mc['vocab_size'], n_positions=n_positions, n_embd=mc['n_embd'], n_layer=mc['n_layer'], n_head=mc['n_head'],
        layer_norm_epsilon=mc.get('layer_norm_epsilon', 1e-5), embd_pdrop=mc.get('embd_pdrop', 0.0),
        bit_widths=list(mc['bit_widths']), lora_rank_per_bit=_int_keys(mc['lora_rank_per_bit']),
        lora_alpha_per_bit=_int_keys(mc['lora_alpha_per_bit']), quantizer_per_bit=_int_keys(mc['quantizer_per_bit']),
        activation_bits_per_bit=_int_keys(mc.get('activation_bits_per_bit')), per_channel_quantization=False)
    model = SPLMHeadModel(cfg)
    bits = target_bits if target_bits is not None else ck.get('bit_width')
    if bits:
        model.set_precision(bits)
    model.load_state_dict(sd, strict=True)
    return model.to(device).eval(), ck
