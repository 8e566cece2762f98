"""Oracle leaf ops (CPU). Each function cites what it restates. TEST INFRASTRUCTURE ONLY."""
import math

import numpy as np
import torch
import torch.nn.functional as F


def msda_core(value, spatial_shapes, sampling_locations, attention_weights):
    """[3P] mmcv `multi_scale_deformable_attn_pytorch` (the pure-PyTorch definition the reference's
    `type='MultiScaleDeformableAttention'` op, configs/instance/coco_b48n17.py:49-58, falls back to):
    per level, F.grid_sample(value_l, 2*loc-1, bilinear, zeros, align_corners=False), weighted sum.

    value (B,Nv,H,D); spatial_shapes (L,2) (H_l,W_l); sampling_locations (B,Nq,H,L,P,2) (x,y);
    attention_weights (B,Nq,H,L,P) -> (B,Nq,H*D)."""
    B, _, H, D = value.shape
    _, Nq, _, L, P, _ = sampling_locations.shape
    shapes = [(int(h), int(w)) for h, w in spatial_shapes.tolist()]
    value_list = value.split([h * w for h, w in shapes], dim=1)
    grids = 2 * sampling_locations - 1
    sampled = []
    for lvl, (h, w) in enumerate(shapes):
        v = value_list[lvl].flatten(2).transpose(1, 2).reshape(B * H, D, h, w)
        g = grids[:, :, :, lvl].transpose(1, 2).flatten(0, 1)  # (B*H, Nq, P, 2)
        sampled.append(F.grid_sample(v, g, mode='bilinear', padding_mode='zeros', align_corners=False))
    aw = attention_weights.transpose(1, 2).reshape(B * H, 1, Nq, L * P)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * aw).sum(-1).view(B, H * D, Nq)
    return out.transpose(1, 2).contiguous()


def msda_core_loops(value, spatial_shapes, sampling_locations, attention_weights):
    """Independent scalar-loop statement of the same op: sample skipped unless
    -1 < y*H-0.5 < H and -1 < x*W-0.5 < W; the 4 integer neighbours, those outside the map give 0.
    numpy float64; small shapes only."""
    v = value.double().numpy()
    loc = sampling_locations.double().numpy()
    aw = attention_weights.double().numpy()
    B, Nv, H, D = v.shape
    _, Nq, _, L, P, _ = loc.shape
    shapes = [(int(h), int(w)) for h, w in spatial_shapes.tolist()]
    starts = np.cumsum([0] + [h * w for h, w in shapes])[:-1]
    out = np.zeros((B, Nq, H, D))
    for b in range(B):
        for q in range(Nq):
            for h in range(H):
                for l, (Hl, Wl) in enumerate(shapes):
                    for p in range(P):
                        x, y = loc[b, q, h, l, p]
                        him, wim = y * Hl - 0.5, x * Wl - 0.5
                        if not (him > -1 and wim > -1 and him < Hl and wim < Wl):
                            continue
                        h0, w0 = math.floor(him), math.floor(wim)
                        lh, lw = him - h0, wim - w0
                        acc = np.zeros(D)
                        for (yy, xx, ww) in ((h0, w0, (1 - lh) * (1 - lw)), (h0, w0 + 1, (1 - lh) * lw),
                                             (h0 + 1, w0, lh * (1 - lw)), (h0 + 1, w0 + 1, lh * lw)):
                            if 0 <= yy < Hl and 0 <= xx < Wl:
                                acc += ww * v[b, starts[l] + yy * Wl + xx, h]
                        out[b, q, h] += aw[b, q, h, l, p] * acc
    return torch.from_numpy(out.reshape(B, Nq, H * D))


def mask_logits(mask_embed, mask_feature):
    """open_set/models/mask2former_head.py:748."""
    return torch.einsum('bqc,bchw->bqhw', mask_embed, mask_feature)


def attn_mask_from_logits(mask_pred, target_size, num_heads=None):
    """open_set/models/mask2former_head.py:749-759: bilinear resize, (optional x num_heads repeat),
    sigmoid < 0.5. Returns bool (B,Q,h*w) when num_heads is None, else (B*num_heads,Q,h*w)."""
    am = F.interpolate(mask_pred, target_size, mode='bilinear', align_corners=False)
    if num_heads is None:
        am = am.flatten(2)
    else:
        am = am.flatten(2).unsqueeze(1).repeat((1, num_heads, 1, 1)).flatten(0, 1)
    return (am.sigmoid() < 0.5).detach()


def attn_mask_logits(mask_pred, target_size):
    """the resized logits whose sign decides the mask (used to exclude numerical ties in tests)."""
    return F.interpolate(mask_pred, target_size, mode='bilinear', align_corners=False).flatten(2)


def fix_full_rows(attn_mask):
    """open_set/models/mask2former_head.py:825-826 (in place)."""
    attn_mask[torch.where(attn_mask.sum(-1) == attn_mask.shape[-1])] = False
    return attn_mask


def masked_attention_core(q, k, v, mask, num_heads, scale=None):
    """softmax(q k^T * scale + (-inf where mask)) v per head -- the core of nn.MultiheadAttention
    ([3P] mmcv MultiheadAttention wrapper; boolean attn_mask True = blocked).
    q (B,Q,E), k,v (B,S,E), mask bool (B,Q,S) or None -> (B,Q,E)."""
    B, Q, E = q.shape
    S = k.shape[1]
    D = E // num_heads
    scale = 1.0 / math.sqrt(D) if scale is None else scale
    qh = (q * scale).view(B, Q, num_heads, D).transpose(1, 2)
    kh = k.view(B, S, num_heads, D).transpose(1, 2)
    vh = v.view(B, S, num_heads, D).transpose(1, 2)
    att = qh @ kh.transpose(-1, -2)
    if mask is not None:
        att = att.masked_fill(mask[:, None], float('-inf'))
    att = att.softmax(-1)
    return (att @ vh).transpose(1, 2).reshape(B, Q, E)
