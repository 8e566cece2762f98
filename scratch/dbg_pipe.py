import sys, warnings, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import cgg_amd
from cgg_amd import registry, runtime, synthetic
from cgg_amd.pipeline import StagePipeline
from util import randomize
dev = torch.device('cuda')
cfg = synthetic.model_config(num_things=10, num_stuff=0, num_unknown=3, num_queries=20, depth=50, enc_layers=2, dec_layers=3, vocab=500, num_points=256)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    model = registry.build_detector(cfg)
randomize(model, seed=9)
for m in model.modules():
    if isinstance(m, torch.nn.BatchNorm2d):
        m.running_var.fill_(1.0); m.running_mean.zero_()
model = model.to(dev).eval()
B, H, W = 2, 128, 192
g = torch.Generator().manual_seed(11)
imgs = [torch.randn(B, 3, H, W, generator=g).to(dev) for _ in range(5)]
head = model.panoptic_head
def d(a, b): return (a.float() - b.float()).abs().max().item()
with torch.no_grad(), runtime.precision_scope('bf16'):
    seq = []
    for im in imgs:
        feats = model.extract_feat(im)
        enc = head._encode(feats)
        seq.append(([f.clone() for f in feats], [k.clone() for kv in enc['kvs'] for k in kv], enc['packed_full'].hi.clone()))
    torch.cuda.synchronize()
    for nstage in (2, 3):
        if nstage == 2:
            fns = [lambda x: head._encode(model.extract_feat(x)), lambda e: (e,)]
        else:
            fns = [model.extract_feat, head._encode, lambda e: (e,)]
        pipe = StagePipeline(fns, imgs[0])
        for k, im in enumerate(imgs):
            slot = pipe.submit(im)
            res = pipe.wait(slot)[0]
            kv = [t.clone() for kv_ in res['kvs'] for t in kv_]
            hi = res['packed_full'].hi.clone()
            torch.cuda.synchronize()
            print(nstage, 'batch', k, 'kv diff', max(d(a, b) for a, b in zip(kv, seq[k][1])), 'packed diff', d(hi, seq[k][2]))
        pipe.flush(); torch.cuda.synchronize()
print('---- full detector pipelines')
from cgg_amd.pipeline import detector_pipeline
metas = synthetic.img_metas(B, H, W)
with torch.no_grad(), runtime.precision_scope('bf16'):
    want = []
    for im in imgs:
        res = model.simple_test(im, metas, rescale=True, device_results=True)
        want.append([{k: tuple(t.clone() for t in v) for k, v in r.items()} for r in res])
    torch.cuda.synchronize()
    for nstage in (2, 3):
        pipe = detector_pipeline(model, imgs[0], metas, stages=nstage, rescale=True, device_results=True)
        for k, im in enumerate(imgs):
            slot = pipe.submit(im)
            res = pipe.wait(slot)
            got = [{kk: tuple(t.clone() for t in v) for kk, v in r.items()} for r in res]
            torch.cuda.synchronize()
            for bi in range(B):
                for key in want[k][bi]:
                    wl, wb, wm = want[k][bi][key]; gl, gb, gm = got[bi][key]
                    print(nstage, 'batch', k, 'img', bi, key, 'labels eq', sorted(wl.tolist()) == sorted(gl.tolist()),
                          'score sum %.4f vs %.4f' % (wb[:, 4].sum().item(), gb[:, 4].sum().item()), 'mask px', int(wm.sum()), int(gm.sum()))
        pipe.flush(); torch.cuda.synchronize()
