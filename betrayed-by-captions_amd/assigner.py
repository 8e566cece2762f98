"""Hungarian mask assignment of the reference (open_set/assigners/mask_hungarian_assigner.py:15-144)
with the [3P] mmdet match costs / sampler / point-sampling helpers its config names
(configs/instance/coco_b48n17.py:165-177; semantics in SURVEY.md A6-A8).

The cost matrix stays f32 (assignment indices must be bit-exact with the reference CPU path); the
solve runs on the host like the reference's scipy `linear_sum_assignment` (:126-131), in the C++ solver of the
extension (`cgg_linear_sum_assignment_f32`: same algorithm, scan order and tie rule -> same indices).
`assign_batch` batches the device->host copies of a whole (layers x images) step into one transfer.
"""
import torch
import torch.nn.functional as F

from .registry import BBOX_ASSIGNERS, BBOX_SAMPLERS, MATCH_COST, build_match_cost

from . import ops


def linear_sum_assignment(cost):
    """scipy.optimize.linear_sum_assignment's contract (and indices) from the C++ solver in libcgg_hip.so."""
    rows, cols = ops.linear_sum_assignment_batch([torch.as_tensor(cost)])[0]
    return rows.numpy(), cols.numpy()


# ---- [3P] mmcv.ops.point_sample / mmdet point utilities -------------------------------------------
def point_sample(input, points, align_corners=False, **kwargs):
    """input (N,C,H,W), points (N,P,2) or (N,Hg,Wg,2) in [0,1] (x,y) -> (N,C,P) / (N,C,Hg,Wg)."""
    add_dim = False
    if points.dim() == 3:
        add_dim = True
        points = points.unsqueeze(2)
    if (input.is_cuda and input.shape[1] == 1 and add_dim and not align_corners and not kwargs and input.dtype == torch.float32
            and points.dtype == torch.float32 and not (torch.is_grad_enabled() and points.requires_grad)):
        # single-channel maps sampled at per-row point lists (uncertainty sampling, mask losses): the HIP gather (+ scatter backward)
        from . import ops
        return ops.point_sample_rows(input[:, 0], points[:, :, 0]).unsqueeze(1)
    out = F.grid_sample(input, points * 2.0 - 1.0, align_corners=align_corners, **kwargs)
    if add_dim:
        out = out.squeeze(3)
    return out


def get_uncertainty(mask_pred, labels):
    if mask_pred.shape[1] == 1:
        gt_class_logits = mask_pred.clone()
    else:
        inds = torch.arange(mask_pred.shape[0], device=mask_pred.device)
        gt_class_logits = mask_pred[inds, labels].unsqueeze(1)
    return -torch.abs(gt_class_logits)


def get_uncertain_point_coords_with_randomness(mask_pred, labels, num_points, oversample_ratio,
                                               importance_sample_ratio, rand_fn=None):
    """[3P] mmdet. `rand_fn(kind, shape, device)` lets tests pin the random draws
    (device RNG streams differ between CPU and GPU, so parity tests feed both sides the same coords)."""
    if rand_fn is None:
        rand_fn = lambda kind, shape, device: torch.rand(*shape, device=device)  # noqa: E731
    assert oversample_ratio >= 1
    assert 0 <= importance_sample_ratio <= 1
    n = mask_pred.shape[0]
    num_sampled = int(num_points * oversample_ratio)
    point_coords = rand_fn('oversample', (n, num_sampled, 2), mask_pred.device)
    point_logits = point_sample(mask_pred, point_coords)
    unc = get_uncertainty(point_logits, labels)
    num_uncertain = int(importance_sample_ratio * num_points)
    num_random = num_points - num_uncertain
    u2 = unc[:, 0, :]
    if u2.is_cuda and u2.dtype == torch.float32 and u2.stride(1) == 1 and 0 < num_uncertain <= u2.shape[1]:
        from . import ops
        idx = ops.topk_select(u2, num_uncertain)          # the SET of the most uncertain points (the loss does not read their order)
    else:
        idx = torch.topk(u2, k=num_uncertain, dim=1)[1]
    shift = num_sampled * torch.arange(n, dtype=torch.long, device=mask_pred.device)
    idx = idx + shift[:, None]
    point_coords = point_coords.view(-1, 2)[idx.view(-1), :].view(n, num_uncertain, 2)
    if num_random > 0:
        rand = rand_fn('random', (n, num_random, 2), mask_pred.device)
        point_coords = torch.cat((point_coords, rand), dim=1)
    return point_coords


# ---- [3P] match costs ---------------------------------------------------------------------------
@MATCH_COST.register_module()
class ClassificationCost:

    def __init__(self, weight=1.):
        self.weight = weight

    def __call__(self, cls_pred, gt_labels):
        return -cls_pred.softmax(-1)[:, gt_labels] * self.weight


@MATCH_COST.register_module()
class CrossEntropyLossCost:

    def __init__(self, weight=1., use_sigmoid=True):
        assert use_sigmoid, 'use_sigmoid = False is not supported yet.'
        self.weight = weight
        self.use_sigmoid = use_sigmoid

    def __call__(self, cls_pred, gt_labels):
        x = cls_pred.flatten(1).float()
        t = gt_labels.flatten(1).float()
        n = x.shape[1]
        pos = F.binary_cross_entropy_with_logits(x, torch.ones_like(x), reduction='none')
        neg = F.binary_cross_entropy_with_logits(x, torch.zeros_like(x), reduction='none')
        cost = torch.einsum('nc,mc->nm', pos, t) + torch.einsum('nc,mc->nm', neg, 1 - t)
        return cost / n * self.weight


@MATCH_COST.register_module()
class DiceCost:

    def __init__(self, weight=1., pred_act=False, eps=1e-3, naive_dice=True):
        self.weight, self.pred_act, self.eps, self.naive_dice = weight, pred_act, eps, naive_dice

    def __call__(self, mask_preds, gt_masks):
        if self.pred_act:
            mask_preds = mask_preds.sigmoid()
        x = mask_preds.flatten(1)
        t = gt_masks.flatten(1).float()
        num = 2 * torch.einsum('nc,mc->nm', x, t)
        if self.naive_dice:
            den = x.sum(-1)[:, None] + t.sum(-1)[None, :]
        else:
            den = x.pow(2).sum(1)[:, None] + t.pow(2).sum(1)[None, :]
        return (1 - (num + self.eps) / (den + self.eps)) * self.weight


# ---- [3P] AssignResult / MaskPseudoSampler ---------------------------------------------------------
class AssignResult:

    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels

    @property
    def num_preds(self):
        return len(self.gt_inds)


class MaskSamplingResult:

    def __init__(self, pos_inds, neg_inds, masks, gt_masks, assign_result, gt_flags):
        self.pos_inds, self.neg_inds = pos_inds, neg_inds
        self.pos_masks, self.neg_masks = masks[pos_inds], masks[neg_inds]
        self.pos_is_gt = gt_flags[pos_inds]
        self.num_gts = gt_masks.shape[0]
        self.pos_assigned_gt_inds = assign_result.gt_inds[pos_inds] - 1
        if gt_masks.numel() == 0:
            assert self.pos_assigned_gt_inds.numel() == 0
            self.pos_gt_masks = torch.empty_like(gt_masks)
        else:
            self.pos_gt_masks = gt_masks[self.pos_assigned_gt_inds, :]
        self.pos_gt_labels = assign_result.labels[pos_inds] if assign_result.labels is not None else None


@BBOX_SAMPLERS.register_module()
class MaskPseudoSampler:

    def __init__(self, **kwargs):
        pass

    def sample(self, assign_result, masks, gt_masks, **kwargs):
        pos_inds = torch.nonzero(assign_result.gt_inds > 0, as_tuple=False).squeeze(-1).unique()
        neg_inds = torch.nonzero(assign_result.gt_inds == 0, as_tuple=False).squeeze(-1).unique()
        gt_flags = masks.new_zeros(masks.shape[0], dtype=torch.uint8)
        return MaskSamplingResult(pos_inds, neg_inds, masks, gt_masks, assign_result, gt_flags)


# ---- the reference's assigner ----------------------------------------------------------------------
@BBOX_ASSIGNERS.register_module()
class MaskHungarianAssignerOpen:
    """cost = cls + cls_emb + mask + dice -> Hungarian on the host -> AssignResult
    (gt_inds: 0 = background, k>0 = matched to gt k-1)."""

    def __init__(self, cls_cost=dict(type='ClassificationCost', weight=1.0),
                 mask_cost=dict(type='FocalLossCost', weight=1.0, binary_input=True),
                 dice_cost=dict(type='DiceCost', weight=1.0),
                 cls_emb_cost=dict(type='ClassficationCost', weight=1.0)):
        self.cls_cost = build_match_cost(cls_cost)
        self.mask_cost = build_match_cost(mask_cost)
        self.dice_cost = build_match_cost(dice_cost)
        self.cls_emb_cost = build_match_cost(cls_emb_cost)

    def cost_matrix(self, cls_pred, cls_emb_pred, mask_pred, gt_labels, gt_mask):
        cost = 0
        if self.cls_cost.weight != 0 and cls_pred is not None:
            cost = cost + self.cls_cost(cls_pred, gt_labels)
        if self.cls_emb_cost.weight != 0 and cls_emb_pred is not None:
            cost = cost + self.cls_emb_cost(cls_emb_pred, gt_labels)
        if self.mask_cost.weight != 0:
            cost = cost + self.mask_cost(mask_pred, gt_mask)
        if self.dice_cost.weight != 0:
            cost = cost + self.dice_cost(mask_pred, gt_mask)
        return cost

    @staticmethod
    def _result_from_match(num_gt, num_query, rows, cols, gt_labels, like):
        gt_inds = like.new_full((num_query, ), 0, dtype=torch.long)
        labels = like.new_full((num_query, ), -1, dtype=torch.long)
        rows = torch.as_tensor(rows, dtype=torch.long, device=like.device)
        cols = torch.as_tensor(cols, dtype=torch.long, device=like.device)
        gt_inds[rows] = cols + 1
        labels[rows] = gt_labels[cols]
        return AssignResult(num_gt, gt_inds, None, labels=labels)

    def assign(self, cls_pred, cls_emb_pred, mask_pred, gt_labels, gt_mask, img_meta,
               gt_bboxes_ignore=None, eps=1e-7):
        assert gt_bboxes_ignore is None, 'Only case when gt_bboxes_ignore is None is supported.'
        num_gt, num_query = gt_labels.shape[0], mask_pred.shape[0]
        if num_gt == 0 or num_query == 0:
            gt_inds = mask_pred.new_full((num_query, ), -1, dtype=torch.long)
            labels = mask_pred.new_full((num_query, ), -1, dtype=torch.long)
            if num_gt == 0:
                gt_inds[:] = 0
            return AssignResult(num_gt, gt_inds, None, labels=labels)
        cost = self.cost_matrix(cls_pred, cls_emb_pred, mask_pred, gt_labels, gt_mask)
        rows, cols = linear_sum_assignment(cost.detach().cpu())
        return self._result_from_match(num_gt, num_query, rows, cols, gt_labels, mask_pred)

    def assign_batch(self, items):
        """items: list of (cls_pred, cls_emb_pred, mask_pred, gt_labels, gt_mask). Builds every cost
        matrix on the device, moves them to the host in ONE transfer (the reference syncs once per
        image and decoder layer, :126), solves, returns a list of AssignResult."""
        costs, todo = [], []
        results = [None] * len(items)
        for i, (cls_pred, cls_emb_pred, mask_pred, gt_labels, gt_mask) in enumerate(items):
            num_gt, num_query = gt_labels.shape[0], mask_pred.shape[0]
            if num_gt == 0 or num_query == 0:
                results[i] = self.assign(cls_pred, cls_emb_pred, mask_pred, gt_labels, gt_mask, None)
                continue
            costs.append(self.cost_matrix(cls_pred, cls_emb_pred, mask_pred, gt_labels, gt_mask)
                         .detach().float().reshape(-1))
            todo.append((i, num_query, num_gt))
        if todo:
            flat = torch.cat(costs).cpu()
            mats, off = [], 0
            for (i, nq, ng) in todo:
                mats.append(flat[off:off + nq * ng].view(nq, ng))
                off += nq * ng
            for (i, nq, ng), (rows, cols) in zip(todo, ops.linear_sum_assignment_batch(mats)):
                results[i] = self._result_from_match(ng, nq, rows.numpy(), cols.numpy(), items[i][3], items[i][2])
        return results
