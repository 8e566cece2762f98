"""Fused ResNet layer1 identity Bottleneck vs the three-call path of backbones._conv_nhwc at configs[1] (2 x 256 x 256 x 256)."""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
importlib.import_module('betrayed-by-captions_amd')
ops = importlib.import_module('betrayed-by-captions_amd.ops')
bb = importlib.import_module('betrayed-by-captions_amd.backbones')
rt = importlib.import_module('betrayed-by-captions_amd.runtime')
dev = torch.device('cuda:0')
torch.manual_seed(0)
blk = bb.Bottleneck(256, 64).to(dev).eval()
for m in blk.modules():
    if isinstance(m, torch.nn.BatchNorm2d):
        m.running_var.uniform_(0.5, 1.5); m.running_mean.normal_(0, 0.1); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.1)
x = (torch.randn(2, 256, 256, 256, device=dev) * 0.7).bfloat16()
def fold(conv, bn):
    s = bn.weight.detach().float() / torch.sqrt(bn.running_var.float() + bn.eps)
    w = (conv.weight.detach().float() * s.view(-1, 1, 1, 1)).to(torch.bfloat16)
    b = (bn.bias.detach().float() - bn.running_mean.float() * s).to(torch.bfloat16).contiguous()
    if conv.kernel_size == (1, 1):
        return None, b, w.flatten(1).contiguous()
    return w.contiguous(memory_format=torch.channels_last), b, None
f1, f2, f3 = fold(blk.conv1, blk.bn1), fold(blk.conv2, blk.bn2), fold(blk.conv3, blk.bn3)
packed = ops.pack_bottleneck64(f1[2], f1[1], f2[0], f2[1], f3[2], f3[1])
with torch.no_grad(), rt.precision_scope('bf16'):
    def lib():
        y = bb.ResNet._conv_nhwc(x, blk.conv1, f1, True)
        y = bb.ResNet._conv_nhwc(y, blk.conv2, f2, True)
        return bb.ResNet._conv_nhwc(y, blk.conv3, f3, True, x)
    fused = lambda: ops.bottleneck64(x, packed)
    a, b = lib(), fused()
    torch.cuda.synchronize()
    d = (a.float() - b.float()).abs()
    print('max |fused - three-call| %.4f  mean %.5f  (|y| max %.2f)' % (d.max().item(), d.mean().item(), a.float().abs().max().item()))
    for name, fn in (('three calls', lib), ('fused', fused)):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(50): fn()
        torch.cuda.synchronize(); print(name, '%.1f us' % ((time.perf_counter() - t) / 50 * 1e6))
