"""x3 vs f32 form of the masked cross-attention backward on one small case: which of grad_q / grad_k / grad_v differ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cgg_amd
from cgg_amd import ops
dev = torch.device('cuda')
for (B, Q, S, H) in ((1, 20, 77, 4), (1, 32, 128, 8), (2, 100, 1050, 8)):
    g = torch.Generator().manual_seed(5)
    E = H * 32
    q = torch.randn(B, Q, E, generator=g).to(dev); kv = torch.randn(B, S, 2 * E, generator=g).to(dev); go = torch.randn(B, Q, E, generator=g).to(dev)
    out, lse = ops.masked_xattn(q, kv, None, H, return_lse=True)
    res = {}
    for form in (True, False):
        ops.XATTN_X3_BWD = form
        res[form] = ops.masked_xattn_backward(q, kv, None, out, lse, go, H)
    (gq3, gkv3), (gq, gkv) = res[True], res[False]
    for name, a, b in (('grad_q', gq3, gq), ('grad_k', gkv3[..., :E], gkv[..., :E]), ('grad_v', gkv3[..., E:], gkv[..., E:])):
        print(B, Q, S, H, name, 'max|x3 - f32| / max|f32| = %.3e' % ((a - b).abs().max().item() / b.abs().max().item()),
              ' ratio of norms %.4f' % (a.norm().item() / b.norm().item()), flush=True)
