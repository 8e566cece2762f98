import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from cgg_amd import ops
dev = torch.device('cuda')
g = torch.Generator().manual_seed(0)
qidx = torch.cat([torch.randint(0, 100, (100,), generator=g) for _ in range(3)]).to(dev)
sc = torch.rand(300, generator=g).to(dev)
dense = (torch.randn(100, 256, 256, generator=g) * 3).to(dev)
sparse = torch.full((100, 256, 256), -6.0)
for q in range(100):                       # one blob per query, ~3 % of the image
    y, x = torch.randint(20, 200, (2,), generator=g).tolist()
    sparse[q, y:y + 45, x:x + 45] = 4.0
sparse = sparse.to(dev)
for name, logits in (('dense random', dense), ('sparse blobs', sparse)):
    f = lambda: ops.instance_masks_picks(logits, qidx, sc, (1024, 1024), (1024, 1024), (1024, 1024))
    for _ in range(3): m, bb = f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print('%s: %.1f us, on-pixels %d, score sum %.6f' % (name, e0.elapsed_time(e1) / 20 * 1e3, int(m.sum()), float(bb[:, 4].sum())))
