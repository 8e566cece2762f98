"""which ingredient of the configs[3] slice (200 queries / batch 4 / Swin channel counts) makes the product's gradients noisier than the
float32 oracle's?  usage: python scratch/slice_dbg.py Q B channels_tag"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
import test_fullsize_gpu as T
from cgg_amd import synthetic
Q, B, tag = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
if len(sys.argv) > 4 and sys.argv[4] == 'einsum_bwd':
    from cgg_amd import ops
    ops.mask_logits_backward_ok = lambda *a: False      # torch.einsum f32 instead of cgg_mask_logits_backward (3 x bf16 split)
ch = (128, 256, 512, 1024) if tag == 'swin' else (256, 512, 1024, 2048)
cfg = T.swin_b_config(Q) if tag == 'swin' else synthetic.model_config(num_things=65, num_stuff=0, num_unknown=17, num_queries=Q, depth=50)
try:
    T._forward_train_slice(torch.device('cuda'), cfg, B, ch, 79, f'dbg Q={Q} B={B} {tag}')
except AssertionError as e:
    print('ASSERT', str(e)[:600])
