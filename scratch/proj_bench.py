"""Fused value_proj + offsets/weights projection kernel vs the two library GEMMs."""
import importlib, os, sys, time, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
importlib.import_module('betrayed-by-captions_amd')
ops = importlib.import_module('betrayed-by-captions_amd.ops')
dev = torch.device('cuda:0')
torch.manual_seed(0)
for M in (43008, 1000):
    C = 256
    x16 = torch.randn(M, C, device=dev).bfloat16(); xp16 = torch.randn(M, C, device=dev).bfloat16()
    wv = torch.randn(256, C, device=dev) * 0.06; bv = torch.randn(256, device=dev) * 0.1
    wc = torch.randn(288, C, device=dev) * 0.06; bc = torch.randn(288, device=dev)
    wvp, wcp = ops.pack_encoder_proj_weight(wv), ops.pack_encoder_proj_weight(wc)
    wvb, wcb, bvb, bcb = wv.bfloat16(), wc.bfloat16(), bv.bfloat16(), bc.bfloat16()
    lib = lambda: (F.linear(x16, wvb, bvb), F.linear(xp16, wcb, bcb))
    fused = lambda: ops.encoder_proj(x16, xp16, wvp, bv, wcp, bc)
    rv = x16.double() @ wvb.double().t() + bv.double(); ro = xp16.double() @ wcb.double().t() + bc.double()
    v, o = fused(); lv, lo = lib(); torch.cuda.synchronize()
    print(M, 'fused err', (v.double() - rv).abs().max().item(), (o.double() - ro).abs().max().item(),
          'lib err', (lv.double() - rv).abs().max().item(), (lo.double() - ro).abs().max().item(),
          'ulp-rel', ((v.double() - rv).abs() / rv.abs().clamp_min(1)).max().item())
    for name, fn in (('lib', lib), ('fused', fused)):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(50): fn()
        torch.cuda.synchronize(); print('  ', name, (time.perf_counter() - t) / 50 * 1e6, 'us')
