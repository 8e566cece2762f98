"""Checkpoint key-compat loader: the released CGG checkpoints are mmcv-runner files,
`{'meta': ..., 'state_dict': {...}, 'optimizer': ...}`, possibly saved from a DataParallel wrapper (`module.` prefix).
Every module of this package keeps the reference's parameter names ([3P] mmdet / mmcv layer names included), so loading
is name-for-name; this file restates what `mmcv.runner.load_checkpoint` / `save_checkpoint` do around that:
unwrap, strip prefixes, report what did not match, never silently drop a shape mismatch.
"""
import os
import re
import time
from collections import OrderedDict

import torch


def _unwrap(ckpt):
    if not isinstance(ckpt, dict):
        raise RuntimeError(f'No state_dict found in checkpoint of type {type(ckpt)}')
    for key in ('state_dict', 'model'):
        if key in ckpt and isinstance(ckpt[key], dict):
            return ckpt[key]
    return ckpt


def load_state_dict(module, state_dict, strict=False, logger=None):
    """[3P] mmcv.runner.load_state_dict: name-for-name copy; returns (missing, unexpected, mismatched) and warns (or
    raises when `strict`) instead of failing on the first problem."""
    own = module.state_dict()
    missing = [k for k in own if k not in state_dict and not k.endswith('num_batches_tracked')]
    unexpected = [k for k in state_dict if k not in own]
    mismatched = [(k, tuple(state_dict[k].shape), tuple(own[k].shape)) for k in state_dict
                  if k in own and tuple(state_dict[k].shape) != tuple(own[k].shape)]
    good = OrderedDict((k, v) for k, v in state_dict.items() if k in own and tuple(v.shape) == tuple(own[k].shape))
    module.load_state_dict(good, strict=False)
    msgs = []
    if unexpected:
        msgs.append('unexpected key in source state_dict: ' + ', '.join(unexpected))
    if missing:
        msgs.append('missing keys in source state_dict: ' + ', '.join(missing))
    for k, a, b in mismatched:
        msgs.append(f'size mismatch for {k}: checkpoint {a} vs model {b}')
    if msgs:
        text = 'The model and loaded state dict do not match exactly\n' + '\n'.join(msgs)
        if strict:
            raise RuntimeError(text)
        (logger.warning if logger is not None else print)(text)
    return missing, unexpected, mismatched


def load_checkpoint(model, filename, map_location='cpu', strict=False, logger=None,
                    revise_keys=((r'^module\.', ''),)):
    """[3P] mmcv.runner.load_checkpoint (tools/test.py:239, `load_from` / `resume_from` in tools/train.py): `filename` is
    a path or an already loaded dict. Returns the checkpoint dict (with `meta`, e.g. CLASSES)."""
    ckpt = filename if isinstance(filename, dict) else torch.load(filename, map_location=map_location, weights_only=False)
    state = _unwrap(ckpt)
    for pattern, repl in revise_keys:
        state = OrderedDict((re.sub(pattern, repl, k), v) for k, v in state.items())
    load_state_dict(model, state, strict, logger)
    return ckpt


def save_checkpoint(model, filename, optimizer=None, meta=None):
    """[3P] mmcv.runner.save_checkpoint layout: meta + cpu state_dict (+ optimizer)."""
    meta = dict(meta or {})
    meta.setdefault('time', time.asctime())
    if hasattr(model, 'module'):
        model = model.module
    ckpt = {'meta': meta, 'state_dict': OrderedDict((k, v.detach().cpu()) for k, v in model.state_dict().items())}
    if optimizer is not None:
        ckpt['optimizer'] = optimizer.state_dict()
    os.makedirs(os.path.dirname(os.path.abspath(filename)), exist_ok=True)
    torch.save(ckpt, filename)
    return filename
