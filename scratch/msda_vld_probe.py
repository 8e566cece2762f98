"""cgg_msda_forward_fused_vld: time of the f32 encoder sampling kernel as a function of the value row stride (configs[1] / configs[2] shapes)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cgg_amd  # noqa: F401
from cgg_amd import ops
dev = torch.device('cuda')
hw = [(32, 32), (64, 64), (128, 128)]
N = sum(h * w for h, w in hw)
starts = [0, 1024, 1024 + 4096]
H, D, L, P = 8, 32, 3, 4
g = torch.Generator().manual_seed(1)
ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing='ij'), -1).flip(-1).reshape(-1, 2)
                 for h, w in hw], 0).to(dev)
for B in (2, 16):
    offs = torch.randn(B, N, 288, generator=g).to(dev)
    offs[..., :192] *= 2.0
    for vld in (256, 288, 320, 384, 512, 544, 576, 640):
        buf = torch.randn(B, N, vld, generator=g).to(dev)
        out = torch.empty(B, N, 256, device=dev)
        hwa = ops._int_array([v for p in hw for v in p]); st = ops._int_array(starts)
        def call():
            rc = ops._lib_().cgg_msda_forward_fused_vld(ctypes.c_void_p(buf.data_ptr()), vld, hwa, st, ctypes.c_void_p(offs.data_ptr()), 288,
                                                        ctypes.c_void_p(ref.data_ptr()), ctypes.c_void_p(out.data_ptr()), B, N, H, D, L, N, P,
                                                        ops.stream_ptr(dev))
            assert rc == 0, ops._lib_().cgg_last_error_string()
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            call()
        e.record(); torch.cuda.synchronize()
        print(f'B={B} value row stride {vld:4d} floats: {s.elapsed_time(e) / 20 * 1e3:7.1f} us')
