"""Shared builders for the tests: a product head/detector from the synthetic reference-style config and
an oracle head carrying the SAME weights."""
import copy
import warnings

import torch

import cgg_amd
from cgg_amd import registry, synthetic
from oracle import head as OH


def small_cfg(**kw):
    d = dict(num_things=10, num_stuff=0, num_unknown=3, num_queries=20, depth=18, enc_layers=2, dec_layers=3,
             vocab=500, num_points=256)
    d.update(kw)
    return synthetic.model_config(**d)


def head_cfg(cfg):
    hc = copy.deepcopy(cfg['panoptic_head'])
    hc.update(train_cfg=cfg['train_cfg'], test_cfg=cfg['test_cfg'])
    return hc


def randomize(module, seed=0, scale=None):
    """deterministic non-degenerate weights (the default init zeroes sampling offsets / attention weights)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in sorted(module.named_parameters()):
            if 'word_embeddings' in name:
                continue
            if p.dim() > 1:
                fan = p.shape[1] * (p[0][0].numel() if p.dim() > 2 else 1)
                p.copy_(torch.randn(p.shape, generator=g) / fan**0.5)
            elif 'norm' in name.lower() and name.endswith('weight') or name.endswith('gn.weight'):
                p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
        for name, p in module.named_parameters():
            if name.endswith('sampling_offsets.bias'):
                p.mul_(10.0)  # offsets of a few pixels
    return module


def build_heads(cfg, seed=0):
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        hc = head_cfg(cfg)
        prod = registry.build_head(hc)
        prod.init_weights()
        randomize(prod, seed)
        orc = OH.OracleHead(**hc)
    orc.load_state_dict(prod.state_dict())
    return prod.eval(), orc.eval()
