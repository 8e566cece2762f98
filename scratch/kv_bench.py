"""Decoder K / V projection kernel (one launch per memory level) vs the library GEMM pair it replaces."""
import importlib, os, sys, time, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
importlib.import_module('betrayed-by-captions_amd')
ops = importlib.import_module('betrayed-by-captions_amd.ops')
dev = torch.device('cuda:0')
torch.manual_seed(0)
B, C, NK = 2, 256, 768
wk = torch.randn(NK, C, device=dev) * 0.05; wv = torch.randn(NK, C, device=dev) * 0.05; bk = torch.randn(NK, device=dev)
wkp, wvp = ops.pack_decoder_k_weight(wk), ops.pack_linear_weight(wv)
wkb, wvb, bkb = wk.bfloat16(), wv.bfloat16(), bk.bfloat16()
tot = {'lib': 0.0, 'fused': 0.0}
for hw in (1024, 4096, 16384):
    m16 = torch.randn(B, hw, C, device=dev).bfloat16(); mp16 = torch.randn(B, hw, C, device=dev).bfloat16()
    lib = lambda: (F.linear(mp16, wkb, bkb), torch.matmul(wvb, m16.transpose(1, 2)))
    fused = lambda: ops.decoder_kv_proj(m16, mp16, wkp, bk, wvp)
    (k0, v0), (k1, v1) = lib(), fused()
    print(hw, 'max |k - lib|', (k0.float() - k1.float()).abs().max().item(), 'max |vt - lib|', (v0.float() - v1.float()).abs().max().item())
    for name, fn in (('lib', lib), ('fused', fused)):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(50): fn()
        torch.cuda.synchronize(); us = (time.perf_counter() - t) / 50 * 1e6
        tot[name] += us
        print('   ', name, '%.1f us' % us)
print('three levels: library %.1f us, fused %.1f us' % (tot['lib'], tot['fused']))
