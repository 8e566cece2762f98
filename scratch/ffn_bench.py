"""Fused encoder FFN + LayerNorm kernel vs the library path (two hipBLASLt GEMMs + residual-LayerNorm pass)."""
import importlib, sys, time, torch, torch.nn.functional as F
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('betrayed-by-captions_amd')
ops = importlib.import_module('betrayed-by-captions_amd.ops')
dev = torch.device('cuda:0')
torch.manual_seed(0)
M, C, FF = 43008, 256, 1024
x = torch.randn(M, C, device=dev)
x16 = x.to(torch.bfloat16)
w1 = torch.randn(FF, C, device=dev) * 0.06; b1 = torch.randn(FF, device=dev) * 0.1
w2 = torch.randn(C, FF, device=dev) * 0.03; b2 = torch.randn(C, device=dev) * 0.1
g = torch.rand(C, device=dev) + 0.5; be = torch.randn(C, device=dev) * 0.1
pos = torch.randn(M // 2, C, device=dev)
w1p, w2p = ops.pack_linear_weight(w1), ops.pack_linear_weight(w2)
w1b, w2b, b1b, b2b = w1.bfloat16(), w2.bfloat16(), b1.bfloat16(), b2.bfloat16()

def lib():
    h = torch._addmm_activation(b1b, x16, w1b.t())
    f = F.linear(h, w2b, b2b)
    return ops.add_layernorm_stream(x16, f, g, be, 1e-5, pos=pos, want_f32=False, want_bf16=True, want_pos=True)

def fused():
    return ops.encoder_ffn_ln(x16, w1p, b1, w2p, b2, g, be, 1e-5, pos=pos, want_bf16=True, want_pos=True)

# f64 reference from the bf16-rounded operands
xd = x16.double(); h = torch.relu(xd @ w1b.double().t() + b1.double())
ref = F.layer_norm(xd + h.to(torch.bfloat16).double() @ w2b.double().t() + b2.double(), (C,), g.double(), be.double(), 1e-5)
_, y_l, yp_l = lib(); _, y_f, yp_f = fused()
torch.cuda.synchronize()
print('lib   err', (y_l.double() - ref).abs().max().item(), 'fused err', (y_f.double() - ref).abs().max().item(),
      'pos err', (yp_f.double() - (ref + pos.double().repeat(2, 1))).abs().max().item())
for name, fn in (('lib', lib), ('fused', fused)):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(50): fn()
    torch.cuda.synchronize(); print(name, (time.perf_counter() - t) / 50 * 1e6, 'us')

# ---- the whole layer tail: output_proj + LayerNorm + FFN + LayerNorm ----
a16 = torch.randn(M, C, device=dev).bfloat16()
wo = torch.randn(C, C, device=dev) * 0.06; bo = torch.randn(C, device=dev) * 0.1
wop, wob, bob = ops.pack_linear_weight(wo), wo.bfloat16(), bo.bfloat16()

def lib_tail():
    o = F.linear(a16, wob, bob)
    _, x1, _ = ops.add_layernorm_stream(x16, o, g, be, 1e-5, want_f32=False)
    h = torch._addmm_activation(b1b, x1, w1b.t())
    f = F.linear(h, w2b, b2b)
    return ops.add_layernorm_stream(x1, f, g, be, 1e-5, pos=pos, want_f32=False, want_bf16=True, want_pos=True)

def fused_tail():
    return ops.encoder_layer_tail(a16, x16, wop, bo, (g, be, 1e-5), w1p, b1, w2p, b2, (g, be, 1e-5), pos=pos, want_pos=True)

_, yl, _ = lib_tail(); _, yf, _ = fused_tail(); torch.cuda.synchronize()
print('layer tail: |fused - library path| max', (yl.float() - yf.float()).abs().max().item())
for name, fn in (('lib_tail (3 GEMMs + 2 LayerNorm passes)', lib_tail), ('fused_tail (one launch)', fused_tail)):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(50): fn()
    torch.cuda.synchronize(); print(name, (time.perf_counter() - t) / 50 * 1e6, 'us')
