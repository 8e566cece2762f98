"""Runs only the hot HIP kernels at BASELINE configs[1] shapes (for rocprofv3 --pmc passes)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import cgg_amd
from cgg_amd import ops
dev = 'cuda'
B, Q, C, H, W = 2, 100, 256, 256, 256
g = torch.Generator().manual_seed(0)
E = torch.randn(B, Q, C, generator=g).to(dev)
F_ = torch.randn(B, C, H, W, generator=g).to(dev)
split = len(sys.argv) > 1 and sys.argv[1] == 'split'
packed = ops.pack_mask_feature(F_, 1, split=split)
for _ in range(20):
    out, _ = ops.mask_logits(E, packed, want_logits=True)
torch.cuda.synchronize()
# MSDeformAttn (fused prologue) at 1024x1024: levels 32^2, 64^2, 128^2
shapes = [(32, 32), (64, 64), (128, 128)]; starts = [0, 1024, 5120]; N = 21504
raw = torch.randn(B, N, 288, generator=g); raw[..., :192] *= 2.0
ref = []
for h, w in shapes:
    ys, xs = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing='ij')
    ref.append(torch.stack([(xs.flatten() + .5) / w, (ys.flatten() + .5) / h], -1))
ref = torch.cat(ref).to(dev); raw = raw.to(dev)
v = torch.randn(B, N, 8, 32, generator=g).to(dev).to(torch.bfloat16)
for _ in range(20):
    ops.msda_forward_fused(v, shapes, starts, raw, ref, 4)
torch.cuda.synchronize()
raw16 = raw.to(torch.bfloat16)
for _ in range(20):      # the encoder-stream variant the bench runs: bf16 value / offsets / output
    ops.msda_forward_fused_bf16(v, shapes, starts, raw16, ref, 4)
torch.cuda.synchronize()
# encoder layer tail (output_proj + LN + FFN + LN) and input projections at the same shapes
M = B * N
mk = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g) * sc).to(dev)
a16, x16, xp16 = mk(M, 256).bfloat16(), mk(M, 256).bfloat16(), mk(M, 256).bfloat16()
wo, w1, w2 = ops.pack_linear_weight(mk(256, 256, sc=0.05)), ops.pack_linear_weight(mk(1024, 256, sc=0.05)), \
    ops.pack_linear_weight(mk(256, 1024, sc=0.03))
bo, b1, b2, g0, g1, be = mk(256, sc=0.1), mk(1024, sc=0.1), mk(256, sc=0.1), mk(256, sc=0.1) + 1, mk(256, sc=0.1) + 1, mk(256, sc=0.1)
pos = mk(N, 256)
for _ in range(20):
    ops.encoder_layer_tail(a16, x16, wo, bo, (g0, be, 1e-5), w1, b1, w2, b2, (g1, be, 1e-5))     # as in the step: y rows only
torch.cuda.synchronize()
wv, wc = ops.pack_encoder_proj_weight(mk(256, 256, sc=0.05)), ops.pack_encoder_proj_weight(mk(288, 256, sc=0.05))
bc288, pos16 = mk(288), pos.bfloat16()
for _ in range(20):
    ops.encoder_proj(x16, None, wv, bo, wc, bc288, pos16=pos16)                                  # x + pos formed in the kernel
torch.cuda.synchronize()
print('done')
