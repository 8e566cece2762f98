"""Diagnostic for tests/test_fullsize_gpu.py: where does the fp32-mode error at configs[1] come from, and what is the
oracle's OWN f32 rounding uncertainty (f32 oracle vs f64 oracle) at the same weights?  python scratch/fullsize_diag.py"""
import copy, json, os, sys, time, warnings
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import cgg_amd
from cgg_amd import registry, runtime, synthetic, ops
from oracle import head as OH
from util import MaskTeacher, head_cfg, randomize
import test_fullsize_gpu as T

dev = torch.device('cuda:0')
size = int(os.environ.get('SIZE', 1024))
out = {}
for sharp in [float(v) for v in os.environ.get('SHARP', '1,2,4').split(',')]:
    cfg = synthetic.model_config(num_things=65, num_stuff=0, num_unknown=17, num_queries=100, depth=50)
    B, H, W = 2, size, size
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    img = synthetic.structured_images(B, H, W, seed=1234)
    T.QK_SHARPEN = sharp
    model, orc, backbone = T.build_detector_pair(cfg, 31, img, dev)
    metas = synthetic.img_metas(B, H, W)
    head = model.panoptic_head
    rec = {}
    with torch.no_grad():
        feats = list(backbone(img))
        f64 = [f.double() for f in backbone.double()(img.double())]
        rec['backbone_cpu_f32_vs_f64_rel'] = max(((a.double() - b).abs().max() / b.abs().max()).item() for a, b in zip(feats, f64))
        with runtime.precision_scope('fp32'):
            gf = model.extract_feat(img.to(dev))
        rec['backbone_gpu_vs_f64_rel'] = max(((a.double().cpu() - b).abs().max() / b.abs().max()).item() for a, b in zip(gf, f64))
        # head only, from the SAME f32 features
        t32 = MaskTeacher(orc)
        oc, oe, om = t32.run_oracle(lambda: orc.forward(feats, metas))
        orc64 = copy.deepcopy(orc).double()
        t64 = MaskTeacher(orc64)
        # f64 oracle with the f32 oracle's masks injected is not possible without a hook; compare only up to first flip:
        oc64, oe64, om64 = t64.run_oracle(lambda: orc64.forward([f.double() for f in feats], metas))
        flips = [float(((a < 0) != (b < 0)).float().mean()) for a, b in zip(t32.logits, t64.logits)]
        rec['oracle_f32_vs_f64_attn_bit_flip_frac'] = flips
        rec['oracle_f32_vs_f64_mask_err_per_layer'] = [float((a.double() - b).abs().max()) for a, b in zip(om, om64)]
        rec['logit_scale'] = float(om[-1].abs().max())
        cls = OH.cls_emb_scores(oe[-1], model.panoptic_fusion_head.all_class_embs.cpu()).argmax(-1)
        rec['distinct_classes'] = len(set(cls.flatten().tolist()))
        on = (om[-1] > 0).float().flatten(2).mean(2)
        rec['on_frac_minmax'] = [float(on.min()), float(on.max())]
        # product head (fp32 mode) from the same f32 features, oracle-f32 masks injected
        with runtime.precision_scope('fp32'):
            head.attn_mask_hook = t32.hook
            pc, pe, pm = head.forward([f.to(dev) for f in feats], metas)
            head.attn_mask_hook = None
            pmf, pmem = head.pixel_decoder([f.to(dev) for f in feats])
        omf, omem = orc.pixel_decoder(feats)
        rec['pixel_decoder_mask_feature_err'] = float((pmf.cpu() - omf).abs().max()); rec['mask_feature_scale'] = float(omf.abs().max())
        rec['pixel_decoder_memory_err'] = [float((a.cpu() - b).abs().max()) for a, b in zip(pmem, omem)]
        rec['product_vs_oracle32_mask_err_per_layer'] = [float((a.cpu() - b).abs().max()) for a, b in zip(pm, om)]
        rec['product_vs_oracle32_emb_err_per_layer'] = [float((a.cpu() - b).abs().max()) for a, b in zip(pe, oe)]
        rec['worst_flipped_logit'] = t32.worst
    out[str(sharp)] = rec
    print(sharp, json.dumps(rec), flush=True)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'fullsize_diag.json'), 'w'), indent=1)
