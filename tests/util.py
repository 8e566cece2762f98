"""Shared builders for the tests: a product head/detector from the synthetic reference-style config and
an oracle head carrying the SAME weights."""
import copy
import warnings

import torch

import cgg_amd
from cgg_amd import registry, synthetic
from oracle import head as OH


def small_cfg(**kw):
    d = dict(num_things=10, num_stuff=0, num_unknown=3, num_queries=20, depth=18, enc_layers=2, dec_layers=3,
             vocab=500, num_points=256)
    d.update(kw)
    return synthetic.model_config(**d)


def ag_cfg(**kw):
    """The no-class-embedding / class-agnostic family the reference ships (configs/instance/coco_ag_pretrain_3x.py:97-133, 144-161):
    `use_class_emb=False, pred_emb_norm=True, class_agnostic=True`, no caption heads, `loss_cls` weight 2.0 (class_weight on), the
    assigner's `cls_cost` 2.0 / `cls_emb_cost` 0.0, closed-set fusion head (`ins_results`)."""
    cfg = small_cfg(use_caption=False, use_caption_generation=False, num_unknown=0, **kw)   # :5-9: one known class list, nothing held out
    h = cfg['panoptic_head']
    h.update(use_class_emb=False, class_agnostic=True, pred_emb_norm=True, text_emb_norm=False)
    h['loss_cls']['loss_weight'] = 2.0
    h['loss_cls_emb']['loss_weight'] = 0.0
    cfg['train_cfg']['assigner']['cls_cost']['weight'] = 2.0
    cfg['train_cfg']['assigner']['cls_emb_cost']['weight'] = 0.0
    cfg['panoptic_fusion_head']['use_class_emb'] = False
    cfg['test_cfg'].update(eval_types=['ins_results'], use_class_emb=False)
    return cfg


def g10_inputs():
    """classification logits / blob-shaped mask logits of the G10 closed-set post-processing fixtures."""
    g = torch.Generator().manual_seed(110)
    Q, hh, ww, K = 10, 32, 48, 12
    cls = torch.randn(Q, K + 1, generator=g) * 3
    ys = torch.arange(hh).view(hh, 1).float()
    xs = torch.arange(ww).view(1, ww).float()
    mp = torch.empty(Q, hh, ww)
    for q in range(Q):
        cy, cx = float(torch.rand(1, generator=g)) * hh, float(torch.rand(1, generator=g)) * ww
        r = 4 + float(torch.rand(1, generator=g)) * 10
        mp[q] = (r * r - ((ys - cy)**2 + (xs - cx)**2)) / 8 + 0.3 * torch.randn(hh, ww, generator=g)
    return cls, mp


def head_cfg(cfg):
    hc = copy.deepcopy(cfg['panoptic_head'])
    hc.update(train_cfg=cfg['train_cfg'], test_cfg=cfg['test_cfg'])
    return hc


def randomize(module, seed=0, scale=None):
    """deterministic non-degenerate weights (the default init zeroes sampling offsets / attention weights)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in sorted(module.named_parameters()):
            if 'word_embeddings' in name:       # BERT table: N(0, 0.04^2), padding row 0 zero
                p.copy_(torch.randn(p.shape, generator=g) * 0.04)
                p[0].zero_()
                continue
            if p.dim() > 1:
                fan = p.shape[1] * (p[0][0].numel() if p.dim() > 2 else 1)
                p.copy_(torch.randn(p.shape, generator=g) / fan**0.5)
            elif 'norm' in name.lower() and name.endswith('weight') or name.endswith('gn.weight'):
                p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
        for name, p in module.named_parameters():
            if name.endswith('sampling_offsets.bias'):
                p.mul_(10.0)  # offsets of a few pixels
    return module


def build_heads(cfg, seed=0):
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        hc = head_cfg(cfg)
        prod = registry.build_head(hc)
        prod.init_weights()
        randomize(prod, seed)
        orc = OH.OracleHead(**hc)
    orc.load_state_dict(prod.state_dict())
    return prod.eval(), orc.eval()


class MaskTeacher:
    """Tie-aware parity harness for the decoder loop.

    The attention mask is `resized logit < 0`: a logit within rounding distance of 0 may legitimately fall
    on either side on different hardware, and one flipped key changes everything downstream. So parity is
    split in two checks: (1) the product's own mask bits equal the oracle's wherever the oracle's logit is
    farther than `margin` from 0; (2) with the ORACLE's masks injected into the product's decoder loop, every
    output matches within the stated tolerance."""

    def __init__(self, oracle_head, margin=1e-3):
        self.orc = oracle_head
        self.margin = margin
        self.seen = []
        self.worst = []

    def run_oracle(self, fn):
        self.orc.trace = dict(attn_logits=[])
        try:
            out = fn()
        finally:
            self.logits = self.orc.trace['attn_logits']
            self.orc.trace = None
        return out

    def hook(self, layer_idx, bits):
        from cgg_amd.query_decoder import pack_bool_mask
        from cgg_amd import ops
        lg = self.logits[layer_idx]                       # (B, Q, S) oracle resized logits
        S = lg.shape[-1]
        mine = ops.unpack_bits(bits, S).cpu()
        want = lg < 0
        clear = lg.abs() > self.margin
        wrong = mine != want
        self.worst.append(float(lg.abs()[wrong].max()) if bool(wrong.any()) else 0.0)   # largest |logit| with a flipped bit
        self.seen.append((bool(torch.equal(mine[clear], want[clear])), float(clear.float().mean())))
        return pack_bool_mask(want).to(bits.device).contiguous()

    def check(self):
        assert self.seen, 'hook never called'
        assert all(ok for ok, _ in self.seen), (self.seen, self.worst)
        assert min(frac for _, frac in self.seen) > 0.98, self.seen


class AssignTeacher:
    """Tie-aware parity harness for the Hungarian matching (mask_hungarian_assigner.py:126-131), the second DISCRETE decision of a
    training step besides the attention-mask threshold.

    Two assignments whose total costs differ by less than the float32 noise of the cost matrix are equally "optimal"; which one a
    solver returns depends on the last bits of the costs, and the two choices give different loss graphs for that (layer, image) --
    i.e. gradients that differ by O(1) downstream (round 6, scratch/grad_taps.py: ONE such near-tie at configs[3], total costs
    205.352778 vs 205.352773, was the whole of query_embed's "15 % gradient error"). So, as for the masks: (1) the product's cost
    matrix must equal the float32 oracle's entry-wise within `cost_tol`, and its assignment must be optimal under the ORACLE's matrix
    within `tie_tol` (relative to the optimum); (2) the oracle's assignment is then injected into the product (and replayed in the
    float64 oracle run), so that all three runs differentiate the same graph."""

    def __init__(self, nonempty_images, cost_tol=1e-3, tie_tol=2e-5):
        self.images = list(nonempty_images)                    # image indices with >= 1 ground-truth instance, ascending
        self.cost_tol, self.tie_tol = cost_tol, tie_tol
        self.recorded, self.flips, self.calls, self.worst_cost, self.worst_tie = [], [], 0, 0.0, 0.0

    def _patch(self, fn):
        import contextlib
        from oracle import head as OH

        @contextlib.contextmanager
        def cm():
            old = OH.linear_sum_assignment
            OH.linear_sum_assignment = fn
            try:
                yield self
            finally:
                OH.linear_sum_assignment = old
        return cm()

    def record(self):
        """context: the float32 oracle run -- its solutions, in call order (layer-major over the non-empty images)"""
        from scipy.optimize import linear_sum_assignment as lsa

        def fn(cost):
            r, c = lsa(cost)
            self.recorded.append((r.copy(), c.copy(), cost.detach().clone().float()))
            return r, c
        return self._patch(fn)

    def replay(self):
        """context: another oracle run (float64) that must take the SAME assignments"""
        it = iter(self.recorded)

        def fn(cost):
            r, c, _ = next(it)
            return r, c
        return self._patch(fn)

    def hook(self, layer_idx, image_idx, rows, cols, cost):
        import numpy as np
        r, c, oc = self.recorded[layer_idx * len(self.images) + self.images.index(image_idx)]
        oc = oc.double()
        if cost is not None:
            self.worst_cost = max(self.worst_cost, float(((cost.double().cpu() - oc).abs() / (1 + oc.abs())).max()))
        opt, mine = float(oc[r, c].sum()), float(oc[rows, cols].sum())
        self.worst_tie = max(self.worst_tie, abs(mine - opt) / (1 + abs(opt)))
        o1, o2 = np.argsort(c), np.argsort(cols)
        if not (len(c) == len(cols) and np.array_equal(r[o1], rows[o2])):
            self.flips.append((layer_idx, image_idx, mine - opt))
        self.calls += 1
        return r, c

    def check(self, n_layers):
        assert self.calls == n_layers * len(self.images) == len(self.recorded), (self.calls, len(self.recorded))
        assert self.worst_cost <= self.cost_tol, f'cost matrix differs from the oracle by {self.worst_cost:.2e} (relative)'
        assert self.worst_tie <= self.tie_tol, (f'an assignment of the product is not optimal under the oracle\'s costs '
                                                f'(relative excess {self.worst_tie:.2e}; flips {self.flips})')


# ---- deterministic inputs shared by tests/golden/make_golden.py and the golden tests -----------------
def g4_inputs():
    """(cfg, B, H, W, feats, metas, fh_query, fh_feat) of the G3/G4 head fixtures."""
    cfg = small_cfg(num_queries=8, vocab=120)
    B, H, W = 2, 64, 96
    g = torch.Generator().manual_seed(102)
    feats = [torch.randn(B, c, H // s, W // s, generator=g) for c, s in zip((64, 128, 256, 512), (4, 8, 16, 32))]
    metas = [dict(img_shape=(H, W, 3), ori_shape=(H, W, 3), pad_shape=(H, W, 3), batch_input_shape=(H, W))
             for _ in range(B)]
    qf = torch.randn(8, B, 256, generator=g)
    mf = torch.randn(B, 256, 16, 24, generator=g)
    return cfg, B, H, W, feats, metas, qf, mf


def g6_inputs(H=64, W=96):
    """ground truth of the G5/G6 target / loss fixtures."""
    g = torch.Generator().manual_seed(103)
    gt_labels = [torch.tensor([1, 4, 4]), torch.tensor([0, 6])]
    gt_masks = []
    for n in (3, 2):
        m = torch.zeros(n, H, W, dtype=torch.long)
        for i in range(n):
            y0 = int(torch.randint(0, H - 20, (1,), generator=g))
            x0 = int(torch.randint(0, W - 30, (1,), generator=g))
            m[i, y0:y0 + 12 + 4 * i, x0:x0 + 20 + 3 * i] = 1
        gt_masks.append(m)
    cap_ids = [torch.tensor([101, 7, 9, 11, 102] + [0] * 30), torch.tensor([101, 5, 6, 102] + [0] * 31)]
    cap_mask = [(c != 0).long() for c in cap_ids]
    noun_ids = [torch.tensor([9, 11] + [0] * 33), torch.tensor([0] * 35)]       # second caption: no nouns
    noun_mask = [(c != 0).long() for c in noun_ids]
    return gt_labels, gt_masks, cap_ids, cap_mask, noun_ids, noun_mask


def g7_inputs():
    """query embeddings / blob-shaped mask logits of the G7 post-processing fixtures."""
    g = torch.Generator().manual_seed(104)
    Q, hh, ww = 8, 32, 48
    emb = torch.randn(Q, 768, generator=g) * 0.05
    ys = torch.arange(hh).view(1, hh, 1).float()
    xs = torch.arange(ww).view(1, 1, ww).float()
    cy, cx = torch.rand(Q, 1, 1, generator=g) * hh, torch.rand(Q, 1, 1, generator=g) * ww
    r = 4 + torch.rand(Q, 1, 1, generator=g) * 6
    mp = 6 - ((ys - cy)**2 + (xs - cx)**2) / r**2 * 6 + torch.randn(Q, hh, ww, generator=g) * 0.3
    cls_embs = torch.randn(13, 64, generator=g)
    cls_embs[-1] = 0
    pemb = torch.randn(Q, 64, generator=g)
    return emb, mp, cls_embs, pemb


def plain(t):
    """values of a tensor the product may hand over as x3a rows (`ops.X3ATensor`: round 4's parity-mode ResNet outputs and encoder
    memories) -- a plain float32 tensor either way."""
    from cgg_amd import ops
    return ops.x3a_to_f32(t) if ops.is_x3a(t) else t


class Bank:
    """deterministic random-point source shared by product (device) and oracle (cpu): per-kind call counters."""

    def __init__(self, seed):
        self.seed, self.count = seed, {}

    def __call__(self, kind, shape, device):
        import torch
        i = self.count.get(kind, 0)
        self.count[kind] = i + 1
        g = torch.Generator().manual_seed(self.seed + 1000 * i + {'target': 1, 'oversample': 2, 'random': 3}[kind])
        return torch.rand(*shape, generator=g).to(device)
