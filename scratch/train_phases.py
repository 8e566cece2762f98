import sys, time, warnings, torch
sys.path.insert(0, '/root/repo')
import bench
class A: pass
args = A(); args.queries = 100
dev = torch.device('cuda')
import cgg_amd
from cgg_amd import runtime, synthetic
from cgg_amd.train import GradReducer, build_optimizer, clip_grad_norm_
runtime.set_precision('bf16')
cfg, model = bench.build_model(args, dev)
model.train()
B, H, W = 16, 1024, 1024
img = torch.randn(B, 3, H, W, device=dev)
metas = synthetic.img_metas(B, H, W)
nc = cfg['panoptic_head']['num_things_classes']
batch = synthetic.train_batch(B, H, W, num_classes=nc, seed=77, device=dev)
opt = build_optimizer(model, dict(type='AdamW', lr=1e-4, weight_decay=0.05))
red = GradReducer(model)
head = model.panoptic_head
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for it in range(4):
    red.zero_grad()
    t0 = sync()
    feats = model.extract_feat(img)
    t1 = sync()
    outs = head(feats, metas)
    t2 = sync()
    gl, gm = head.preprocess_gt(batch['gt_labels'], batch['gt_masks'], None, metas)
    ce, cm = head.extract_word_embeddings(batch['gt_caption_ids'], batch['gt_caption_mask'], head.caption_gen_emb_type)
    ne, nm = head.extract_word_embeddings(batch['gt_caption_nouns_ids'], batch['gt_caption_nouns_mask'], head.caption_emb_type)
    losses = head.loss(*outs, gl, gm, batch['gt_caption_ids'], ce, cm, batch['gt_caption_nouns_ids'], ne, nm, metas)
    t3 = sync()
    total = sum(losses.values())
    total.backward()
    t4 = sync()
    red.finish(); clip_grad_norm_(red.flats(), 0.01); opt.step()
    t5 = sync()
    if it >= 2:
        print('backbone fwd %.1f | head fwd %.1f | loss %.1f | backward %.1f | clip+optim %.1f | total %.1f ms' % tuple(1e3 * x for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t5 - t0)))
