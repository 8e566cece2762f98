"""Kernel-level roofline numbers at BASELINE configs[1] shapes: >= 20 warm-up + 100 timed launches,
HIP events on the launch stream, cache-warm (back-to-back launches) and L2/MALL-flushed (a 1-GiB memset between
launches, timed separately and subtracted) variants. Prints one JSON document."""
import json, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import cgg_amd
from cgg_amd import ops
from cgg_amd.query_decoder import pack_bool_mask
dev = torch.device('cuda')
flush_buf = torch.empty(1 << 30, dtype=torch.uint8, device=dev)


def timed(fn, flush=False, warm=20, n=100):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    if not flush:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    tot = 0.0
    for _ in range(n):
        flush_buf.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n * 1e3


res = {}
B, Q, C, H, W = 2, 100, 256, 256, 256
g = torch.Generator().manual_seed(0)
E = torch.randn(B, Q, C, generator=g).to(dev)
F_ = torch.randn(B, C, H, W, generator=g).to(dev)
for name, split, inb in (('mask_logits_bf16', False, 2), ('mask_logits_split_f32class', True, 4)):
    packed = ops.pack_mask_feature(F_, 1, split=split)
    byts = B * (C * H * W * inb + Q * C * 4 + Q * H * W * 4)
    fl = 2.0 * B * Q * C * H * W
    for mode in ('warm', 'flushed'):
        us = timed(lambda: ops.mask_logits(E, packed, want_logits=True), flush=(mode == 'flushed'))
        res[f'{name}/{mode}'] = dict(us=us, algorithmic_bytes=byts, GBs=byts / us / 1e3, frac_hbm_8TBs=byts / us / 1e3 / 8000,
                                     tflops=fl / us / 1e6, frac_mfma_bf16_2500=fl / us / 1e6 / 2500)
shapes = [(32, 32), (64, 64), (128, 128)]; starts = [0, 1024, 5120]; N = 21504
raw = torch.randn(B, N, 288, generator=g); raw[..., :192] *= 2.0
ref = []
for h, w in shapes:
    ys, xs = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing='ij')
    ref.append(torch.stack([(xs.flatten() + .5) / w, (ys.flatten() + .5) / h], -1))
ref = torch.cat(ref).to(dev)
v16 = torch.randn(B, N, 8, 32, generator=g).to(dev).bfloat16()
raw16 = raw.to(dev).bfloat16()
byts = B * N * (256 + 288 + 256) * 2
for mode in ('warm', 'flushed'):
    us = timed(lambda: ops.msda_forward_fused_bf16(v16, shapes, starts, raw16, ref, 4), flush=(mode == 'flushed'))
    res[f'msda_fwd_bf16_stream/{mode}'] = dict(us=us, algorithmic_bytes=byts, GBs=byts / us / 1e3, frac_hbm_8TBs=byts / us / 1e3 / 8000)
v32 = v16.float(); raw32 = raw.to(dev)
byts = B * N * (256 * 4 + 288 * 4 + 256 * 4)
us = timed(lambda: ops.msda_forward_fused(v32, shapes, starts, raw32, ref, 4))
res['msda_fwd_f32_parity/warm'] = dict(us=us, algorithmic_bytes=byts, GBs=byts / us / 1e3)
for S in (1024, 4096, 16384):
    q = torch.randn(B, Q, 256, generator=g).to(dev)
    k = torch.randn(B, S, 256, generator=g).to(dev).bfloat16()
    vt = torch.randn(B, 256, S, generator=g).to(dev).bfloat16()
    bits = pack_bool_mask(torch.rand(B, Q, S, generator=g) < 0.5).to(dev)
    ops.attn_mask_fix_full_rows(bits, S)
    byts = B * (2 * S * 256 * 2 + 2 * Q * 256 * 4 + Q * S // 8)
    fl = 4.0 * B * Q * S * 256
    us = timed(lambda: ops.masked_xattn_bf16(q, k, vt, bits, 8))
    res[f'masked_xattn_bf16_S{S}/warm'] = dict(us=us, algorithmic_bytes=byts, GBs=byts / us / 1e3, tflops=fl / us / 1e6)
    kv = torch.cat([k.float(), vt.transpose(1, 2).float()], -1).contiguous()
    us = timed(lambda: ops.masked_xattn(q, kv, bits, 8))
    res[f'masked_xattn_f32_parity_S{S}/warm'] = dict(us=us, algorithmic_bytes=B * (2 * S * 256 * 4 + 2 * Q * 256 * 4 + Q * S // 8),
                                                     tflops=fl / us / 1e6)
lg = torch.randn(100, 256, 256, generator=g).to(dev) * 3
sel = torch.arange(100, dtype=torch.int32, device=dev)
us = timed(lambda: ops.instance_masks(lg, sel, (1024, 1024), (1024, 1024), (1024, 1024)))
res['instance_masks_100x1024x1024/warm'] = dict(us=us, algorithmic_bytes=100 * 1024 * 1024 + 100 * 256 * 256 * 4,
                                                 GBs=(100 * 1024 * 1024 + 100 * 256 * 256 * 4) / us / 1e3)
print(json.dumps(res, indent=1))
