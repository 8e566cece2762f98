"""Times the parity-mode TRAINING contractions (f16 x 3, f32 A split on the fly) at the configs[2] encoder shapes and prints their
share of the dense f16 MFMA peak (3 MFMA products per f32 product): forward / grad-input GEMM (`ops.gemm_x3`), the masked
grad-input (`ops.gemm_x3_bwd`) and the weight gradient (`ops.wgrad_x3`). argv: [rows] [iters]. GPU box only."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
importlib.import_module('betrayed-by-captions_amd')
ops = importlib.import_module('betrayed-by-captions_amd.ops')
M = int(sys.argv[1]) if len(sys.argv) > 1 else 344064
IT = int(sys.argv[2]) if len(sys.argv) > 2 else 10
PEAK = 2.5e15


def timeit(fn, n=IT):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for K, N in [(256, 256), (256, 1024), (1024, 256), (256, 288)]:
    a = torch.randn(M, K, device='cuda')
    w = torch.randn(N, K, device='cuda') * 0.05
    pk = ops.pack_linear_weight_x3(w)
    out = torch.empty(M, N, device='cuda')
    amax = ops.absmax(a)
    t = timeit(lambda: ops.gemm_x3(a, pk, N, out=out))
    fl = 3 * 2.0 * M * N * K
    print(f'gemm_x3        M={M} K={K:4d} N={N:4d}: {t * 1e6:8.1f} us  {fl / t / 1e12:7.1f} TF/s f16  = {fl / t / PEAK:.3f} of peak', flush=True)
    t = timeit(lambda: ops.gemm_x3(a, pk, N, out=out, amax=amax))
    print(f'gemm_x3 scaled M={M} K={K:4d} N={N:4d}: {t * 1e6:8.1f} us  {fl / t / 1e12:7.1f} TF/s f16  = {fl / t / PEAK:.3f} of peak', flush=True)
    ae = ops.x3a_encode(a)
    t = timeit(lambda: ops.x3a_encode(a))
    print(f'x3a_encode     M={M} K={K:4d}        : {t * 1e6:8.1f} us  {2 * M * K * 4 / t / 1e12:6.2f} TB/s', flush=True)
    for cfg in ([-1] if len(sys.argv) <= 3 else [int(c) for c in sys.argv[3].split(',')]):
        t = timeit(lambda: ops.gemm_x3s(ae, pk, N, out=out, cfg=cfg))
        print(f'gemm_x3s cfg{cfg:2d} M={M} K={K:4d} N={N:4d}: {t * 1e6:8.1f} us  {fl / t / 1e12:7.1f} TF/s f16  = {fl / t / PEAK:.3f} of peak', flush=True)
    dy = torch.randn(M, N, device='cuda') * 1e-3
    am = ops.absmax(dy)
    t = timeit(lambda: ops.wgrad_x3(dy, a, want_bias=True, amax=am))
    print(f'wgrad_x3       M={M} K={K:4d} N={N:4d}: {t * 1e6:8.1f} us  {fl / t / 1e12:7.1f} TF/s f16  = {fl / t / PEAK:.3f} of peak', flush=True)
    t = timeit(lambda: ops.absmax(dy))
    print(f'absmax         M={M} N={N:4d}: {t * 1e6:8.1f} us  {M * N * 4 / t / 1e12:6.2f} TB/s', flush=True)
    del a, w, pk, out, dy
