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
