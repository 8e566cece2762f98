import sys, os, torch
os.environ['CGG_GEMM_TUNE'] = '32'
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from cgg_amd import ops
dev = torch.device('cuda')
for name, M, N, K in [('l3 3x3 as GEMM', 8192, 256, 2304), ('l4 3x3 as GEMM', 2048, 512, 4608), ('l2 3x3 as GEMM', 32768, 128, 1152),
                      ('l3 s2 first', 8192, 256, 1152), ('l4 s2 first', 2048, 512, 2304)]:
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / 16).bfloat16(); b = torch.randn(N, device=dev).bfloat16()
    ops.gemm_bias_res_act_bf16(x, w, b, None, True)
    print(name, M, N, K, 'top1 %.1f us chosen %.1f us' % ops.blaslt_last_tuning())
