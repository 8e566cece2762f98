"""Times forward + backward of the parity-mode training linears on the shapes the configs[2] step runs on the f32 LIBRARY GEMM (caption
transformer, vocabulary projection, decoder), against the x3 autograd node (`runtime._X3LinearFn`). GPU box only."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('betrayed-by-captions_amd')
runtime = importlib.import_module('betrayed-by-captions_amd.runtime')
import torch.nn.functional as F

SHAPES = [(16000, 768, 768), (5440, 768, 768), (5440, 768, 2304), (5440, 768, 3072), (5440, 3072, 768), (2048, 768, 30528),
          (1344, 768, 30528), (1600, 256, 256), (1600, 256, 2048), (1600, 2048, 256), (16384 * 16, 512, 256)]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M, K, N in SHAPES:
    x = torch.randn(M, K, device='cuda', requires_grad=True)
    w = (torch.randn(N, K, device='cuda') * 0.02).requires_grad_()
    b = torch.zeros(N, device='cuda', requires_grad=True)
    g = torch.randn(M, N, device='cuda') * 1e-3

    def lib():
        y = F.linear(x, w, b)
        y.backward(g)
        x.grad = w.grad = b.grad = None

    def x3():
        y = runtime._X3LinearFn.apply(x, w, b)
        y.backward(g)
        x.grad = w.grad = b.grad = None

    tl, tx = timeit(lib), timeit(x3)
    # accuracy of both against float64
    xd, wd, gd = x.detach().double(), w.detach().double(), g.double()
    ref_dx, ref_dw = gd @ wd, gd.t() @ xd
    y = runtime._X3LinearFn.apply(x, w, b); y.backward(g)
    ex = ((x.grad.double() - ref_dx).abs().max() / ref_dx.abs().max()).item(), ((w.grad.double() - ref_dw).abs().max() / ref_dw.abs().max()).item()
    x.grad = w.grad = b.grad = None
    y = F.linear(x, w, b); y.backward(g)
    el = ((x.grad.double() - ref_dx).abs().max() / ref_dx.abs().max()).item(), ((w.grad.double() - ref_dw).abs().max() / ref_dw.abs().max()).item()
    x.grad = w.grad = b.grad = None
    print(f'M={M:7d} K={K:5d} N={N:6d}  library {tl:8.1f} us   x3 {tx:8.1f} us   ratio {tl / tx:5.2f}   err dx/dw  x3 {ex[0]:.1e}/{ex[1]:.1e}  lib {el[0]:.1e}/{el[1]:.1e}', flush=True)
