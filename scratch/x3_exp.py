"""Elimination runs of cgg_gemm_x3_kernel: the same launches against library builds with one ingredient removed
(XG_EXP 1 = no global loads after the prologue, 2 = no f32 -> f16 split (raw copy), 3 = no LDS stores). Results are garbage
by construction; only the time matters."""
import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch
    import cgg_amd
    from cgg_amd import _lib
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), sys.argv[1])
    from cgg_amd import ops
    dev = torch.device('cuda')
    def timeit(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / n * 1e3
    out = []
    x = torch.randn(2, 256, 256, 256, device=dev); w = torch.randn(256, 256, 3, 3, device=dev) / 48; b = torch.randn(256, device=dev)
    pk = ops.pack_conv_weight_x3(w)
    out.append('conv256 %.1f' % timeit(lambda: ops.conv_x3_nhwc(x, pk, 256, 3, 1, 1, b)))
    for M, N, K in [(43008, 1024, 256), (43008, 256, 1024), (8192, 256, 1024)]:
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / 16; b = torch.randn(N, device=dev)
        pk = ops.pack_linear_weight_x3(w)
        out.append('gemm%dx%dx%d %.1f' % (M, N, K, timeit(lambda: ops.gemm_x3(x, pk, N, b))))
    print(sys.argv[1], ' '.join(out))
else:
    for lib in ['libcgg_hip.so', 'libcgg_exp1.so', 'libcgg_exp2.so', 'libcgg_exp3.so']:
        subprocess.run([sys.executable, os.path.abspath(__file__), lib])
