"""GPU check: in a parity-mode training forward the frozen ResNet stages hand their channel-last maps along and the hand-over is
still valid when the pixel decoder reads it. usage: python scratch/handover_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cgg_amd  # noqa: F401
from cgg_amd import registry, runtime
bb = registry.build_backbone(dict(type='ResNet', depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=3,
                                  norm_cfg=dict(type='BN', requires_grad=False), norm_eval=True, style='pytorch'))
bb.init_weights()
bb = bb.cuda().train()
img = torch.randn(2, 3, 256, 256, device='cuda')
for prec in ('fp32', 'bf16'):
    with runtime.precision_scope(prec):
        feats = bb(img)
    got = [runtime.handed_nhwc(f) is not None for f in feats]
    ok = all(torch.equal(runtime.handed_nhwc(f).permute(0, 3, 1, 2), f) for f, g in zip(feats, got) if g)
    print(prec, 'handed:', got, 'equal:', ok, flush=True)
    assert got[:3] == [True, True, True] and ok
print('handover OK')
