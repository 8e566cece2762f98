import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cgg_amd
from cgg_amd import ops
dev = torch.device('cuda')
for (B, Q, H, W) in [(1, 200, 64, 96), (1, 256, 40, 33), (1, 160, 64, 64), (2, 37, 20, 28), (1, 128, 16, 24)]:
    g = torch.Generator().manual_seed(12 + Q)
    embed = torch.randn(B, Q, 256, generator=g).to(dev)
    feat = torch.randn(B, 256, H, W, generator=g).to(dev)
    packed = ops.pack_mask_feature(feat, pool=1, split=False)
    logits, want = ops.mask_logits(embed, packed, want_logits=True, want_bits=True)
    got = ops.mask_logits_bits_astat(embed, packed)
    d = (got != want)
    print((B, Q, H, W), 'mismatching words', int(d.sum()), 'of', d.numel())
    if d.any():
        idx = torch.nonzero(d)
        print(' rows with mismatch:', sorted(set(idx[:, 1].tolist()))[:40], ' tiles:', sorted(set(idx[:, 2].tolist()))[:20])
        i = idx[0]
        print(' first:', i.tolist(), hex(got[tuple(i)].item() & 0xffffffff), hex(want[tuple(i)].item() & 0xffffffff))
