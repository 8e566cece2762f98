"""CPU tests: the C-ABI shared library loads here (no GPU) and exports every symbol that include/cgg_hip.h
declares; the ctypes prototype table covers the header; argument validation returns the documented error
codes without touching a device; the torch wrappers refuse CPU tensors (no silent fallback)."""
import ctypes
import os
import re

import pytest
import torch

import cgg_amd
from cgg_amd import _lib, ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, 'include', 'cgg_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(cgg_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f'{s} declared in include/cgg_hip.h but not exported by libcgg_hip.so'
    assert sorted(_lib.PROTOTYPES) == syms, 'ctypes prototype table and header disagree'
    assert lib.cgg_version() == 100


def test_argument_validation_error_codes_without_a_device():
    lib = _lib.load()
    null = ctypes.c_void_p(None)
    buf = (ctypes.c_char * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.cgg_mask_logits(null, null, null, null, null, 1, 1, 256, 32, null) == -1          # CGG_EINVAL
    assert b'null' in lib.cgg_last_error_string()
    assert lib.cgg_mask_logits(p, p, null, p, null, 1, 100, 128, 32, null) == -2                 # C != 256
    assert lib.cgg_pack_mask_feature(p, p, null, 1, 250, 8, 8, 1, null) == -2                    # C % 16
    assert lib.cgg_pack_mask_feature(p, p, null, 1, 256, 9, 8, 2, null) == -2                    # H % pool
    assert lib.cgg_msda_forward(p, p, p, p, p, p, 1, 10, 8, 30, 3, 5, 4, 0, null) == -2          # D % 4
    assert lib.cgg_msda_forward(p, p, p, p, p, p, 1, 10, 8, 32, 9, 5, 4, 0, null) == -2          # L > 8
    assert lib.cgg_masked_xattn_forward(p, p, null, p, p, 1, 100, 8, 64, 128, 1.0, 0, null) == -2  # D != 32
    assert lib.cgg_masked_xattn_forward(p, p, null, p, p, 1, 200, 8, 32, 128, 1.0, 0, null) == -2  # Q > 128
    assert lib.cgg_linear_rows(p, 30, p, null, null, 0, p, 8, 4, 8, 30, 0, 1, null) == -2        # K % 16
    assert lib.cgg_group_norm(p, p, p, p, p, 1, 30, 4, 4, 32, 1e-5, 0, null) == -1               # C % groups
    assert lib.cgg_masked_xattn_workspace_bytes(2, 100, 8, 32, 16384) > 0


def test_ops_refuse_cpu_tensors():
    x = torch.randn(2, 100, 256)
    feat = torch.randn(2, 256, 8, 8)
    with pytest.raises(_lib.CggError, match='ROCm device'):
        ops.pack_mask_feature(feat)
    with pytest.raises(_lib.CggError):
        ops.linear_rows(x, torch.randn(256, 256))
    with pytest.raises(_lib.CggError):
        ops.masked_xattn(x, torch.randn(2, 64, 512), None, 8)
    with pytest.raises(_lib.CggError):
        ops.msda_forward(torch.randn(1, 80, 8, 32), torch.tensor([[8, 8], [4, 4]]), torch.tensor([0, 64]),
                         torch.rand(1, 5, 8, 2, 4, 2), torch.rand(1, 5, 8, 2, 4))


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libcgg_hip.so')
    with pytest.raises(_lib.CggError, match='no CPU / eager fallback'):
        _lib.load()
