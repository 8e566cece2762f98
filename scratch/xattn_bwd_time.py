import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cgg_amd
from cgg_amd import ops
dev = torch.device('cuda')
B, Q, H = 16, 100, 8
E = 256
g = torch.Generator().manual_seed(5)
for S in (1024, 4096, 16384):
    q = torch.randn(B, Q, E, generator=g).to(dev); kv = torch.randn(B, S, 2 * E, generator=g).to(dev); go = (torch.randn(B, Q, E, generator=g) * 1e-5).to(dev)
    out, lse = ops.masked_xattn(q, kv, None, H, return_lse=True)
    ops.XATTN_X3_BWD = False
    rq, rkv = ops.masked_xattn_backward(q, kv, None, out, lse, go, H)
    ops.XATTN_X3_BWD = True
    gq, gkv = ops.masked_xattn_backward(q, kv, None, out, lse, go, H)
    err = max((gq - rq).abs().max().item() / rq.abs().max().item(), (gkv - rkv).abs().max().item() / rkv.abs().max().item())
    torch.cuda.synchronize()
    s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s_.record()
    for _ in range(10):
        ops.masked_xattn_backward(q, kv, None, out, lse, go, H)
    e_.record(); torch.cuda.synchronize()
    print(f'NOP={os.environ.get("CGG_XB_NOP", "plain")} S={S}: {s_.elapsed_time(e_) / 10 * 1e3:.1f} us  err vs f32 form {err:.2e}', flush=True)
