"""Losses selected by the reference configs (configs/instance/coco_b48n17.py:110-141).

`GroundingLoss` (open_set/models/losses/grounding_loss.py:9-125) and `CrossEntropyLossOpen`
(losses/cross_entropy_loss.py:251-356) are the reference's own registrations; `CrossEntropyLoss` and
`DiceLoss` are the [3P] mmdet losses the configs name (semantics documented by the reference's fork,
cross_entropy_loss.py:63-196, and SURVEY.md A9).
"""
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import runtime
from .registry import LOSSES

_EPS32 = torch.finfo(torch.float32).eps


def weight_reduce_loss(loss, weight=None, reduction='mean', avg_factor=None):
    """[3P] mmdet: elementwise weight, then mean/sum, or sum/(avg_factor+eps) when avg_factor given."""
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        if reduction == 'mean':
            return loss.mean()
        if reduction == 'sum':
            return loss.sum()
        return loss
    if reduction == 'mean':
        return loss.sum() / (avg_factor + _EPS32)
    if reduction != 'none':
        raise ValueError('avg_factor can not be used with reduction="sum"')
    return loss


def _softmax_ce(pred, label, weight, reduction, avg_factor, class_weight, ignore_index, avg_non_ignore,
                log_input=False):
    ignore_index = -100 if ignore_index is None else ignore_index
    fn = F.nll_loss if log_input else F.cross_entropy
    loss = fn(pred, label, weight=class_weight, reduction='none', ignore_index=ignore_index)
    if avg_factor is None and avg_non_ignore and reduction == 'mean':
        avg_factor = label.numel() - (label == ignore_index).sum().item()
    return weight_reduce_loss(loss, None if weight is None else weight.float(), reduction, avg_factor)


def _sigmoid_ce(pred, label, weight, reduction, avg_factor, class_weight, ignore_index, avg_non_ignore):
    ignore_index = -100 if ignore_index is None else ignore_index
    if pred.dim() != label.dim():
        C = pred.size(-1)
        valid = (label >= 0) & (label != ignore_index)
        onehot = label.new_zeros((label.size(0), C))
        idx = torch.nonzero(valid & (label < C), as_tuple=False).squeeze(1)
        if idx.numel() > 0:
            onehot[idx, label[idx]] = 1
        valid_mask = valid.view(-1, 1).expand(label.size(0), C).float()
        weight = valid_mask if weight is None else weight.view(-1, 1).repeat(1, C) * valid_mask
        label = onehot
    else:
        valid_mask = ((label >= 0) & (label != ignore_index)).float()
        weight = valid_mask if weight is None else weight * valid_mask
    if avg_factor is None and avg_non_ignore and reduction == 'mean':
        avg_factor = valid_mask.sum().item()
    loss = F.binary_cross_entropy_with_logits(pred, label.float(), pos_weight=class_weight, reduction='none')
    return weight_reduce_loss(loss, weight.float(), reduction, avg_factor)


def _mask_ce(pred, target, label, reduction='mean', avg_factor=None, class_weight=None, ignore_index=None,
             **kwargs):
    assert ignore_index is None, 'BCE loss does not support ignore_index'
    assert reduction == 'mean' and avg_factor is None
    inds = torch.arange(pred.size(0), dtype=torch.long, device=pred.device)
    return F.binary_cross_entropy_with_logits(pred[inds, label].squeeze(1), target, weight=class_weight,
                                              reduction='mean')[None]


class _CEBase(nn.Module):

    def __init__(self, use_sigmoid=False, use_mask=False, use_logsoftmax=False, reduction='mean',
                 class_weight=None, ignore_index=None, loss_weight=1.0, avg_non_ignore=False):
        super().__init__()
        assert (use_sigmoid is False) or (use_mask is False)
        self.use_sigmoid, self.use_mask, self.use_logsoftmax = use_sigmoid, use_mask, use_logsoftmax
        self.reduction, self.loss_weight, self.class_weight = reduction, loss_weight, class_weight
        self.ignore_index, self.avg_non_ignore = ignore_index, avg_non_ignore
        if ignore_index is not None and not avg_non_ignore and reduction == 'mean':
            warnings.warn('Default ``avg_non_ignore`` is False, if you would like to ignore the certain '
                          'label and average loss over non-ignore labels, which is the same with PyTorch '
                          'official cross_entropy, set ``avg_non_ignore=True``.')

    def extra_repr(self):
        return f'avg_non_ignore={self.avg_non_ignore}'

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None,
                ignore_index=None, **kwargs):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        if ignore_index is None:
            ignore_index = self.ignore_index
        cw = runtime.const_tensor(self.class_weight, cls_score) if self.class_weight is not None else None
        if self.use_sigmoid:
            loss = _sigmoid_ce(cls_score, label, weight, reduction, avg_factor, cw, ignore_index,
                               self.avg_non_ignore)
        elif self.use_mask:
            loss = _mask_ce(cls_score, label, weight, reduction=reduction, avg_factor=avg_factor,
                            class_weight=cw, ignore_index=ignore_index, **kwargs)
        else:
            loss = _softmax_ce(cls_score, label, weight, reduction, avg_factor, cw, ignore_index,
                               self.avg_non_ignore, log_input=self.use_logsoftmax)
        return self.loss_weight * loss


    def rows_ok(self):
        """True when the loss is the plain softmax cross-entropy that `forward_rows` can finish from per-row losses."""
        return not (self.use_sigmoid or self.use_mask or self.use_logsoftmax) and self.class_weight is None

    def forward_rows(self, loss_rows, label, weight=None, avg_factor=None, reduction_override=None, ignore_index=None):
        """`forward` from ALREADY-COMPUTED per-row losses (`F.cross_entropy(..., reduction='none', ignore_index)`): the
        weighting / averaging rules of the reference's CrossEntropyLoss (cross_entropy_loss.py:63-112) applied to them.
        Used with `CaptionTransformer.generator_ce_rows`, which never materialises the logits."""
        assert self.rows_ok()
        reduction = reduction_override if reduction_override else self.reduction
        if ignore_index is None:
            ignore_index = self.ignore_index
        ii = -100 if ignore_index is None else ignore_index
        if avg_factor is None and self.avg_non_ignore and reduction == 'mean':
            avg_factor = label.numel() - (label == ii).sum().item()
        return self.loss_weight * weight_reduce_loss(loss_rows, None if weight is None else weight.float(), reduction,
                                                     avg_factor)


@LOSSES.register_module()
class CrossEntropyLoss(_CEBase):
    """[3P] mmdet CrossEntropyLoss (`type='CrossEntropyLoss'` in every shipped config)."""

    def __init__(self, use_sigmoid=False, use_mask=False, reduction='mean', class_weight=None,
                 ignore_index=None, loss_weight=1.0, avg_non_ignore=False):
        super().__init__(use_sigmoid, use_mask, False, reduction, class_weight, ignore_index, loss_weight,
                         avg_non_ignore)


@LOSSES.register_module()
class CrossEntropyLossOpen(_CEBase):
    """open_set/models/losses/cross_entropy_loss.py:251 -- adds `use_logsoftmax` (NLL on log-probs)."""


@LOSSES.register_module()
class DiceLoss(nn.Module):
    """[3P] mmdet DiceLoss (naive_dice / eps as in coco_b48n17.py:135-141)."""

    def __init__(self, use_sigmoid=True, activate=True, reduction='mean', naive_dice=False,
                 loss_weight=1.0, eps=1e-3):
        super().__init__()
        self.use_sigmoid, self.activate, self.reduction = use_sigmoid, activate, reduction
        self.naive_dice, self.loss_weight, self.eps = naive_dice, loss_weight, eps

    def forward(self, pred, target, weight=None, reduction_override=None, avg_factor=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        if self.activate:
            if not self.use_sigmoid:
                raise NotImplementedError
            pred = pred.sigmoid()
        x = pred.flatten(1)
        t = target.flatten(1).float()
        inter = (x * t).sum(1)
        if self.naive_dice:
            d = (2 * inter + self.eps) / (x.sum(1) + t.sum(1) + self.eps)
        else:
            d = 2 * inter / ((x * x).sum(1) + self.eps + (t * t).sum(1) + self.eps)
        loss = 1 - d
        if weight is not None:
            assert weight.ndim == loss.ndim and len(weight) == len(pred)
        return self.loss_weight * weight_reduce_loss(loss, weight, reduction, avg_factor)


class _GroundingPairCostsFn(torch.autograd.Function):
    """(caption i, image j) pair costs of grounding_loss.py:32-58 on the HIP kernel (scores, both softmaxes and the
    reductions in one launch; the (B*B, T, Q) tensors never exist). Backward: d cost / d score recomputed by the kernel,
    then ONE batched GEMM against the (constant) caption embeddings."""

    @staticmethod
    def forward(ctx, pred, cap, cap_mask, inv_temperature):
        from . import ops
        pred_c, cap_c = pred.contiguous(), cap.contiguous()
        mask_i = cap_mask.to(torch.int32).contiguous()
        ctx.save_for_backward(pred_c, cap_c, mask_i)
        ctx.inv_t = float(inv_temperature)
        return ops.grounding_pair_costs(pred_c, cap_c, mask_i, ctx.inv_t)

    @staticmethod
    def backward(ctx, grad_cost):
        from . import ops
        pred, cap, mask_i = ctx.saved_tensors
        dsim = ops.grounding_pair_costs_backward(pred, cap, mask_i, grad_cost, ctx.inv_t)     # (Bp, Bc*T, Q)
        grad_pred = torch.matmul(dsim.transpose(1, 2), cap.reshape(-1, cap.shape[-1]))       # (Bp, Q, d)
        grad_cap = None
        if ctx.needs_input_grad[1]:      # not on the CGG path (frozen text encoder); kept for completeness
            Bc, T, d = cap.shape
            grad_cap = torch.einsum('jkq,jqd->kd', dsim, pred).view(Bc, T, d)
        return grad_pred, grad_cap, None, None


def _grounding_tail(d_l2v, d_v2l, num_tokens):
    """grounding_loss.py:60-77: captions without nouns are pushed away (+100, detached), then the four
    log-softmax diagonals over the (caption, image) cost matrices."""
    B = d_l2v.shape[0]
    valid = (num_tokens > 0)[:, None].expand(B, d_l2v.shape[1])
    d_l2v = torch.where(valid, d_l2v, d_l2v.max().detach() + 100.0)
    d_v2l = torch.where(valid, d_v2l, d_v2l.max().detach() + 100.0)
    total = 0.
    for cost in (d_l2v, d_v2l):
        total = total + torch.diag(-torch.log_softmax(-cost, dim=0)).mean() \
            + torch.diag(-torch.log_softmax(-cost, dim=1)).mean()
    return total / 4


def grounding_loss(cls_emb_pred, gt_caption_embs, gt_caption_mask, temperature):
    """Caption grounding loss of open_set/models/losses/grounding_loss.py:9-77, evaluated for all
    (caption i, image j) pairs from ONE (B*T, d) x (d, B*Q) contraction -- the reference's three
    B^2-fold `repeat`s (:23-30) are index arithmetic here, nothing is replicated.

    cls_emb_pred (B,Q,d), gt_caption_embs (B,T,d), gt_caption_mask (B,T) 0/1.
    On a ROCm device (f32, Q <= 256, T <= 64) the pair costs come from `cgg_grounding_pair_costs`; other shapes
    and CPU tensors (host-side unit tests) take the torch formulation below."""
    B, Q, d = cls_emb_pred.shape
    T = gt_caption_mask.shape[1]
    num_tokens = gt_caption_mask.sum(dim=1)                                  # (B,)
    from . import ops
    if ops.grounding_supported(cls_emb_pred, gt_caption_embs) and cls_emb_pred.shape[0] == gt_caption_embs.shape[0]:
        cost = _GroundingPairCostsFn.apply(cls_emb_pred, gt_caption_embs, gt_caption_mask, 1.0 / float(temperature))
        return _grounding_tail(cost[0], cost[1], num_tokens)
    sim = torch.matmul(gt_caption_embs.reshape(B * T, d), cls_emb_pred.reshape(B * Q, d).t())
    sim = sim.view(B, T, B, Q).permute(0, 2, 1, 3)                           # [i, j, t, q]
    sim_t = sim / temperature
    dist_t = (-sim) / temperature
    att_l2v = F.softmax(sim_t, dim=3) * gt_caption_mask[:, None, :, None]
    d_l2v = (att_l2v * dist_t).sum(3).sum(2) / torch.max(num_tokens, torch.ones_like(num_tokens))[:, None]
    att_v2l = F.softmax(sim_t, dim=2)
    d_v2l = (att_v2l * dist_t).sum(3).sum(2) / Q
    return _grounding_tail(d_l2v, d_v2l, num_tokens)


@LOSSES.register_module()
class GroundingLoss(nn.Module):
    """open_set/models/losses/grounding_loss.py:79-125."""

    def __init__(self, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.reduction = reduction
        self.loss_weight = loss_weight
        self.grounding_loss = grounding_loss

    def forward(self, cls_emb_pred, gt_caption_embs, gt_caption_mask, temperature, **kwargs):
        return self.loss_weight * self.grounding_loss(cls_emb_pred, gt_caption_embs, gt_caption_mask,
                                                      temperature, **kwargs)
