"""Fused encoder FFN kernel vs the library path on the real layer shapes / parameter scales, run twice (determinism)."""
import importlib, os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
importlib.import_module('betrayed-by-captions_amd')
ops = importlib.import_module('betrayed-by-captions_amd.ops')
dev = torch.device('cuda:0')
torch.manual_seed(0)
for M, N in ((43008, 21504), (43008 - 40, 21504), (2 * 5376, 5376), (100, 50)):
    C, FF = 256, 1024
    x16 = torch.randn(M, C, device=dev).bfloat16()
    w1 = torch.randn(FF, C, device=dev) * 0.05; b1 = torch.zeros(FF, device=dev)
    w2 = torch.randn(C, FF, device=dev) * 0.03; b2 = torch.zeros(C, device=dev)
    g = torch.ones(C, device=dev); be = torch.zeros(C, device=dev)
    pos = torch.randn(N, C, device=dev)
    w1p, w2p = ops.pack_linear_weight(w1), ops.pack_linear_weight(w2)
    outs = []
    for rep in range(3):
        _, y, yp = ops.encoder_ffn_ln(x16, w1p, b1, w2p, b2, g, be, 1e-5, pos=pos, want_bf16=True, want_pos=True)
        torch.cuda.synchronize()
        outs.append((y.clone(), yp.clone()))
    xd = x16.double(); h = torch.relu(xd @ w1.bfloat16().double().t() + b1.double())
    ref = F.layer_norm(xd + h.bfloat16().double() @ w2.bfloat16().double().t() + b2.double(), (C,), g.double(), be.double(), 1e-5)
    refp = ref + pos.double().repeat(M // N + 1, 1)[:M]
    print(M, N, 'err', (outs[0][0].double() - ref).abs().max().item(), (outs[0][1].double() - refp).abs().max().item(),
          'deterministic', all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:]))
