"""Which Swin-B linears take the x3 node in a parity-mode training forward (configs[3] shapes)? GPU box only."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
import cgg_amd  # noqa
from cgg_amd import registry, runtime
from test_fullsize_gpu import swin_b_config
cfg = swin_b_config(200)
bb = registry.build_backbone(cfg['backbone']).cuda().train()
seen = collections.Counter()
real = runtime._X3LinearFn.apply
orig_forward = runtime.ParityLinear.forward


def fwd(self, x):
    before = seen['x3']
    y = orig_forward(self, x)
    return y


calls = []
runtime._X3LinearFn.apply = staticmethod(lambda *a: (calls.append(tuple(a[0].shape)), real(*a))[1])
import torch.nn.functional as F
real_lin = F.linear
lib = []
F.linear = lambda x, w, b=None: (lib.append((tuple(x.shape), tuple(w.shape), x.dtype, x.requires_grad, torch.is_grad_enabled())), real_lin(x, w, b))[1]
x = torch.randn(4, 3, 1024, 1024, device='cuda')
with runtime.precision_scope('fp32'):
    outs = bb(x)
print('x3 node calls', len(calls), collections.Counter(calls).most_common(8))
print('F.linear calls', len(lib), collections.Counter(lib).most_common(8))
