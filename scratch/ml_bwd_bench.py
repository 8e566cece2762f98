"""mask-logit forward / backward at the training shapes of configs[2] (positives of all layers: ~150 rows per image)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgg_amd
from cgg_amd import ops
dev = 'cuda'
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
B, C, h, w = 16, 256, 256, 256
F_ = torch.randn(B, C, h, w, device=dev)
for Q in (100, 150, 200):
    E = torch.randn(B, Q, C, device=dev); go = torch.randn(B, Q, h, w, device=dev)
    for split in (False, True):
        packed = ops.pack_mask_feature(F_, 1, split)
        tf = timeit(lambda: ops.mask_logits(E, packed))
        te = timeit(lambda: ops.mask_logits_backward(E, F_, go, split, True, False))
        tg = timeit(lambda: ops.mask_logits_backward(E, F_, go, split, False, True))
        print(f'Q={Q} split={split}: fwd {tf:.2f} ms, grad_embed {te:.2f} ms, grad_feat {tg:.2f} ms')
    t1 = timeit(lambda: torch.bmm(E, F_.flatten(2)))
    t2 = timeit(lambda: torch.einsum('bqhw,bchw->bqc', go, F_))
    t3 = timeit(lambda: torch.einsum('bqc,bqhw->bchw', E, go))
    print(f'Q={Q} torch f32: fwd {t1:.2f} ms, grad_embed {t2:.2f} ms, grad_feat {t3:.2f} ms')
    E16, F16, g16 = E.bfloat16(), F_.bfloat16(), go.bfloat16()
    t1 = timeit(lambda: torch.bmm(E16, F16.flatten(2)))
    t2 = timeit(lambda: torch.einsum('bqhw,bchw->bqc', g16, F16))
    t3 = timeit(lambda: torch.einsum('bqc,bqhw->bchw', E16, g16))
    print(f'Q={Q} torch bf16: fwd {t1:.2f} ms, grad_embed {t2:.2f} ms, grad_feat {t3:.2f} ms')
