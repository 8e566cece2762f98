"""torch (heuristic top-1 through ATen) vs the tuned hipBLASLt call for the stream path's GEMM shapes."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from cgg_amd import ops
import torch.nn.functional as F
dev = torch.device('cuda')
def timed(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
shapes = [('enc value', 43008, 256, 256, False), ('enc offs', 43008, 384, 256, False), ('enc ffn1', 43008, 1024, 256, True),
          ('enc ffn2', 43008, 256, 1024, False), ('l1 conv1', 131072, 64, 256, True), ('l1 conv3', 131072, 256, 64, True),
          ('l2 conv1', 32768, 128, 512, True), ('l2 conv3', 32768, 512, 128, True), ('l3 conv1', 8192, 256, 1024, True),
          ('l3 conv3', 8192, 1024, 256, True), ('l4 conv1', 2048, 512, 2048, True), ('l4 conv3', 2048, 2048, 512, True),
          ('mask_feature', 131072, 256, 256, False), ('k proj 16k', 32768, 256, 256, False), ('k proj 4k', 8192, 256, 256, False)]
tot_t = tot_l = 0
for name, M, N, K, relu in shapes:
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / 16).bfloat16(); b = torch.randn(N, device=dev).bfloat16()
    ft = (lambda: torch._addmm_activation(b, x, w.t())) if relu else (lambda: F.linear(x, w, b))
    fl = lambda: ops.gemm_bias_res_act_bf16(x, w, b, None, relu)
    fl(); top1, chosen = ops.blaslt_last_tuning()
    tt, tl = timed(ft), timed(fl)
    tot_t += tt; tot_l += tl
    print('%-14s M=%6d N=%4d K=%4d  torch %.1f us  lt-tuned %.1f us  (tuning: top1 %.1f chosen %.1f)' % (name, M, N, K, tt, tl, top1, chosen))
print('sum torch %.1f  lt %.1f' % (tot_t, tot_l))
