"""Round structure of the encoder layer-tail kernel: time vs row count (512 workgroup slots of 64 rows = 32 768 rows per round)."""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
importlib.import_module('betrayed-by-captions_amd')
ops = importlib.import_module('betrayed-by-captions_amd.ops')
dev = torch.device('cuda:0')
torch.manual_seed(0)
C, FF = 256, 1024
mk = lambda *sh, sc=1.0: torch.randn(*sh, device=dev) * sc
wo, w1, w2 = ops.pack_linear_weight(mk(C, C, sc=0.05)), ops.pack_linear_weight(mk(FF, C, sc=0.05)), ops.pack_linear_weight(mk(C, FF, sc=0.03))
bo, b1, b2, g, be = mk(C, sc=0.1), mk(FF, sc=0.1), mk(C, sc=0.1), mk(C, sc=0.1) + 1, mk(C, sc=0.1)
for M in (10240, 16384, 20480, 32768, 43008, 49152, 65536):
    a16, x16 = mk(M, C).bfloat16(), mk(M, C).bfloat16()
    fn = lambda: ops.encoder_layer_tail(a16, x16, wo, bo, (g, be, 1e-5), w1, b1, w2, b2, (g, be, 1e-5))
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(50): fn()
    torch.cuda.synchronize(); print(M, 'rows  %4d workgroups  %.1f us' % (M // 64, (time.perf_counter() - t) / 50 * 1e6))
