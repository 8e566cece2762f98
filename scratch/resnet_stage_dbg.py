"""Where does the x3 ResNet stage's backward leave f32-class accuracy? Gradients at every node boundary against float64."""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cgg_amd
from cgg_amd import ops, runtime
from cgg_amd.backbones import Bottleneck
dev = torch.device('cuda')
_absmax = ops.absmax
def absmax_dbg(t):
    r = _absmax(t)
    nz = t[t != 0].abs()
    print(f'   absmax {tuple(t.shape)}: kernel {r.item():.3e} torch {t.abs().max().item():.3e} median|nz| {nz.median().item():.3e} mean {nz.mean().item():.3e} nz frac {nz.numel() / t.numel():.2f}')
    return r
ops.absmax = absmax_dbg
torch.manual_seed(11)
cin, planes, fs = 128, 32, 1
stage = torch.nn.Sequential(Bottleneck(cin, planes), Bottleneck(planes * 4, planes))
g = torch.Generator().manual_seed(12)
for m in stage.modules():
    if isinstance(m, torch.nn.BatchNorm2d):
        m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.2); m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
        m.weight.data.copy_(torch.rand(m.num_features, generator=g) + 0.5); m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.1)
        m.weight.requires_grad = m.bias.requires_grad = False
stage = stage.to(dev).eval()
B, H, W = 8, 64, 64
x = torch.randn(B, H, W, cin, generator=g).to(dev).requires_grad_()
gm = (torch.randn(B, H, W, planes * 4, generator=g) * 1e-5).to(dev)

def run(mode):
    dt = torch.float64 if mode == 'f64' else torch.float32
    st = copy.deepcopy(stage).to(dt)
    xd = x.detach().to(dt).requires_grad_()
    taps = {}
    def keep(name, t):
        t.retain_grad(); taps[name] = t; return t
    cur = xd
    with runtime.precision_scope('fp32'):
        for bi, blk in enumerate(st):
            if mode == 'x3':
                cb = lambda t, conv, bn, relu, res=None: runtime._X3ConvBnFn.apply(t, conv.weight, *runtime._bn_affine(bn), res, 1, relu)
                o1 = keep(f'{bi}.o1', cb(cur, blk.conv1, blk.bn1, True))
                o2 = keep(f'{bi}.o2', cb(o1, blk.conv2, blk.bn2, True))
                cur = keep(f'{bi}.out', cb(o2, blk.conv3, blk.bn3, True, cur))
            else:
                n = cur.permute(0, 3, 1, 2)
                o1 = keep(f'{bi}.o1', torch.relu(blk.bn1(blk.conv1(n))).permute(0, 2, 3, 1))
                o2 = keep(f'{bi}.o2', torch.relu(blk.bn2(blk.conv2(o1.permute(0, 3, 1, 2)))).permute(0, 2, 3, 1))
                cur = keep(f'{bi}.out', torch.relu(blk.bn3(blk.conv3(o2.permute(0, 3, 1, 2))).permute(0, 2, 3, 1) + cur))
    (cur * gm.to(dt)).sum().backward()
    out = {k: (v.detach().double(), v.grad.double()) for k, v in taps.items()}
    out['x'] = (xd.detach().double(), xd.grad.double())
    for n, p in st.named_parameters():
        if p.grad is not None:
            out['w.' + n] = (p.detach().double(), p.grad.double())
    return out
ref, got, lib = run('f64'), run('x3'), run('f32')
for k in ref:
    sv, sg = ref[k][0].abs().max().item(), ref[k][1].abs().max().item()
    print(f'{k:22s} value err {(got[k][0]-ref[k][0]).abs().max().item()/sv:.1e} (lib {(lib[k][0]-ref[k][0]).abs().max().item()/sv:.1e})   '
          f'grad err {(got[k][1]-ref[k][1]).abs().max().item()/sg:.1e} (lib {(lib[k][1]-ref[k][1]).abs().max().item()/sg:.1e})'
          f'   mask flips {( (got[k][0]>0) != (ref[k][0]>0) ).sum().item() if not k.startswith("w.") else "-"}', flush=True)
