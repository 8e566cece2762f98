import sys, warnings, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import cgg_amd
from cgg_amd import registry, runtime, synthetic
from util import head_cfg, randomize, small_cfg
dev = torch.device('cuda')
cfg = small_cfg(num_queries=20, depth=50)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    head = registry.build_head(head_cfg(cfg))
randomize(head, seed=5)
head = head.to(dev).eval()
B, H, W = 2, 128, 160
feats = synthetic.backbone_feats(B, H, W, channels=(256, 512, 1024, 2048), seed=3)
feats16 = [f.to(dev).bfloat16().contiguous(memory_format=torch.channels_last) for f in feats]
feats32 = [f.float() for f in feats16]
metas = synthetic.img_metas(B, H, W)
def d(a, b): return (a - b).abs().max().item()
with torch.no_grad(), runtime.precision_scope('bf16'):
    outs = []
    for name, f in [('mod', feats32), ('mod', feats32), ('str', feats16), ('str', feats16), ('mod', feats32)]:
        c, e, m = head._forward(f, metas, all_masks=False)
        outs.append((name, m[-1].clone(), e[-1].clone()))
    for i in range(1, len(outs)):
        print(outs[0][0], '->', outs[i][0], 'mask diff', d(outs[0][1], outs[i][1]), 'emb diff', d(outs[0][2], outs[i][2]))
    print('qf checksum', head.query_feat.weight.double().sum().item())
print('---- pixel decoder only')
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    head = registry.build_head(head_cfg(cfg))
randomize(head, seed=5)
head = head.to(dev).eval()
pd = head.pixel_decoder
with torch.no_grad(), runtime.precision_scope('bf16'):
    mf_a, mem_a = pd(feats32)
    mf_s, mem_s, _ = pd.forward_stream(feats16)
    mf_b, mem_b = pd(feats32)
    print('mod->mod after stream: mf', d(mf_a, mf_b), 'mems', [d(x, y) for x, y in zip(mem_a, mem_b)])
    print('mod->stream: mf', d(mf_a, mf_s.float().permute(0, 3, 1, 2)), 'mems', [d(x.flatten(2).transpose(1, 2), y) for x, y in zip(mem_a, mem_s)])
    c1 = head._forward(feats32, metas, all_masks=False)
    c2 = head._forward(feats32, metas, all_masks=False)
    print('head mod->mod', d(c1[2][-1], c2[2][-1]))
    c3 = head._forward(feats16, metas, all_masks=False)
    c4 = head._forward(feats32, metas, all_masks=False)
    print('head mod->str', d(c1[2][-1], c3[2][-1]), 'mod->mod after', d(c1[2][-1], c4[2][-1]))
    for i in range(len(c1[2])):
        if c1[2][i] is not None and c3[2][i] is not None:
            print('layer', i, d(c1[2][i], c3[2][i]))
    for i in range(len(c1[1])):
        print('emb layer', i, d(c1[1][i], c3[1][i]), d(c1[1][i], c4[1][i]))
