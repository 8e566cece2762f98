"""Process-wide execution switches of the MI355X path.

precision:
  'fp32' -- parity mode (default): f32 GEMMs (hipBLASLt), f32 MSDeformAttn values, mask logits as
            3x bf16 MFMA on (hi, lo) split operands (f32-class accuracy), attention on f32 MFMA.
  'bf16' -- throughput mode named by BASELINE.json's north_star: bf16 MFMA contractions
            (mask logits 1x bf16 MFMA, bf16 values for the MSDeformAttn gather, torch GEMMs/convs
            under bf16 autocast); softmax / normalisation / accumulation stay f32.
"""
import contextlib

import torch

_STATE = {'precision': 'fp32'}


def set_precision(p):
    if p not in ('fp32', 'bf16'):
        raise ValueError(f"precision must be 'fp32' or 'bf16', got {p!r}")
    _STATE['precision'] = p


def precision():
    return _STATE['precision']


def is_bf16():
    return _STATE['precision'] == 'bf16'


@contextlib.contextmanager
def precision_scope(p):
    old = _STATE['precision']
    set_precision(p)
    try:
        yield
    finally:
        _STATE['precision'] = old


def autocast():
    """autocast context for the plain library GEMMs / convs (hipBLASLt / MIOpen)."""
    if is_bf16():
        return torch.autocast(device_type='cuda', dtype=torch.bfloat16)
    return contextlib.nullcontext()


_WCACHE = {}


def cast_cached(p, dtype=torch.bfloat16):
    """bf16 copy of a parameter, re-made only when the parameter changes (inference: made once)."""
    key = (id(p), dtype)
    hit = _WCACHE.get(key)
    if hit is not None and hit[0] == p._version and hit[1].device == p.device:
        return hit[1]
    t = p.detach().to(dtype)
    _WCACHE[key] = (p._version, t)
    return t


def linear(x, weight, bias=None):
    """Large-M library GEMM (hipBLASLt): f32 in parity mode, bf16 operands (f32 accumulate) in throughput
    mode with the bf16 weight copy cached. Returns f32."""
    import torch.nn.functional as F
    if is_bf16() and not (torch.is_grad_enabled() and weight.requires_grad):
        y = F.linear(x.to(torch.bfloat16), cast_cached(weight), cast_cached(bias) if bias is not None else None)
        return y.float()
    if is_bf16():
        with torch.autocast(device_type='cuda', dtype=torch.bfloat16):
            return F.linear(x, weight, bias).float()
    return F.linear(x, weight, bias)
