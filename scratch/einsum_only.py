"""the astat / fused einsum rows of bench.py's sweep alone + bit-exactness of the astat kernel against the streamed kernel"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cgg_amd  # noqa: F401
from cgg_amd import ops, runtime
dev = torch.device('cuda')
g = torch.Generator().manual_seed(7)
for Bs, Q, hw in ((2, 100, 256), (2, 200, 256), (4, 200, 256), (16, 100, 256), (2, 200, 100), (1, 37, 50)):
    feat = torch.randn(Bs, 256, hw, hw - (3 if hw < 256 else 0), generator=g).to(dev)
    emb = torch.randn(Bs, Q, 256, generator=g).to(dev)
    with runtime.precision_scope('bf16'):
        packed = ops.pack_mask_feature(feat, 1, False)
        ref = ops.mask_logits(emb, packed, want_logits=False, want_bits=True)[1]
        got = ops.mask_logits_bits_astat(emb, packed)
        torch.cuda.synchronize()
        same = torch.equal(ref, got)
        nd = int((ref != got).sum())
        for name, call in (('fused', lambda: ops.mask_logits(emb, packed, want_logits=False, want_bits=True)),
                           ('astat', lambda: ops.mask_logits_bits_astat(emb, packed))):
            gr = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                call()
            torch.cuda.current_stream().wait_stream(side)
            with torch.cuda.graph(gr):
                for _ in range(20):
                    call()
            gr.replay(); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5):
                gr.replay()
            e.record(); torch.cuda.synchronize()
            ms = s.elapsed_time(e) / 100
            fl = 2.0 * Bs * Q * 256 * feat.shape[2] * feat.shape[3]
            print(f'B={Bs} Q={Q} {feat.shape[2]}x{feat.shape[3]} {name}: {ms*1e3:.1f} us  {fl/ms/1e9:.0f} TF  frac {fl/ms/1e9/2500:.3f}  bits equal {same} ({nd} words differ)')
