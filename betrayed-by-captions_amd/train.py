"""Image-parallel training step: one process per GPU, gradients averaged over RCCL.

What the reference gets from `MMDistributedDataParallel` + mmcv's `OptimizerHook` (apis/train.py:150-170,
configs/instance/coco_b48n17.py:270-286) is restated here for an xGMI node:

  * `GradReducer` -- gradients live as VIEWS into a few large flat buckets (default 64 MiB: a ring
    all-reduce over point-to-point xGMI links is per-link bound, so few large messages beat many small ones);
    a post-accumulate hook per parameter marks the bucket ready when its last gradient lands and launches the
    asynchronous all-reduces STRICTLY IN BUCKET ORDER (a ready bucket k waits until buckets 0..k-1 have been
    launched; `finish()` flushes the rest in the same order), so every rank enqueues the same sequence of
    collectives even when the autograd graphs differ between ranks (an image without GT, a parameter unused on
    one rank only) -- RCCL matches collectives by issue order. The exchange overlaps the rest of backward on
    RCCL's own stream. No gradient copy in, no copy out, no per-parameter collectives.
    `broadcast_buffers=False` as in the reference (frozen BN); `broadcast_parameters` = DDP's construction-time
    rank-0 broadcast.
  * `build_optimizer` -- AdamW with the `paramwise_cfg` semantics of [3P] mmcv DefaultOptimizerConstructor
    for the keys the shipped configs use (`custom_keys` lr_mult / decay_mult, `norm_decay_mult`).
  * `clip_grad_norm_` over the flat buckets (`grad_clip=dict(max_norm=0.01, norm_type=2)`).
  * `train_step` -- forward_train -> _parse_losses (ONE vector all-reduce for the log scalars) -> backward
    (overlapped bucket all-reduces) -> clip -> optimizer step.
"""
import torch
import torch.distributed as dist
import torch.nn as nn


def _dist_on():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


class GradReducer:
    """Bucketed, backward-overlapped gradient averaging for a replicated module."""

    def __init__(self, module, bucket_bytes=64 << 20, process_group=None):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        params = [p for p in module.parameters() if p.requires_grad]
        # gradients become ready roughly in reverse registration order -> fill buckets in that order
        params = list(reversed(params))
        self.buckets = []           # dict(flat=..., params=[(p, off, n)], pending=int, handle=None)
        self._owner = {}
        cur, cur_bytes, key = [], 0, None
        for p in params:
            k = (p.dtype, p.device)
            nbytes = p.numel() * p.element_size()
            if cur and (k != key or cur_bytes + nbytes > bucket_bytes):
                self._seal(cur)
                cur, cur_bytes = [], 0
            key = k
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self._seal(cur)
        self._hooks = []
        for bi, b in enumerate(self.buckets):
            for p, _, _ in b['params']:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))
        self._armed = False
        self._next = 0
        self.launch_log = []        # bucket indices in launch order of the last step (tests)

    def _seal(self, plist):
        # 128-byte aligned slots so every view is vector-load friendly
        offs, total = [], 0
        for p in plist:
            offs.append(total)
            total += (p.numel() + 31) // 32 * 32
        flat = torch.zeros(total, dtype=plist[0].dtype, device=plist[0].device)
        entries = []
        for p, off in zip(plist, offs):
            p.grad = flat[off:off + p.numel()].view_as(p)
            entries.append((p, off, p.numel()))
            self._owner[p] = len(self.buckets)
        self.buckets.append(dict(flat=flat, params=entries, pending=len(entries), handle=None, launched=False,
                                 ready=False, index=len(self.buckets)))

    def _make_hook(self, bi):
        def hook(param):
            if not self._armed:
                return
            b = self.buckets[bi]
            # autograd may have replaced the view (first accumulation into a None grad): restore it
            b['pending'] -= 1
            if b['pending'] == 0:
                b['ready'] = True
                self._launch_ready()
        return hook

    def _launch_ready(self):
        """Launch the longest prefix of ready buckets: collective k is only ever issued after 0..k-1."""
        while self._next < len(self.buckets) and self.buckets[self._next]['ready']:
            self._launch(self.buckets[self._next])
            self._next += 1

    def _check_views(self, b):
        flat = b['flat']
        for p, off, n in b['params']:
            g = p.grad
            view = flat[off:off + n].view_as(p)
            if g is None:
                p.grad = view
            elif g.data_ptr() != view.data_ptr():
                view.copy_(g)
                p.grad = view

    def _launch(self, b):
        self._check_views(b)
        b['launched'] = True
        self.launch_log.append(b['index'])
        if self.world > 1:
            b['flat'].div_(self.world)
            b['handle'] = dist.all_reduce(b['flat'], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def zero_grad(self):
        """Zero the buckets in place (the views stay attached) and arm the hooks for the next backward."""
        for b in self.buckets:
            b['flat'].zero_()
            b['pending'] = len(b['params'])
            b['handle'] = None
            b['launched'] = False
            b['ready'] = False
            for p, off, n in b['params']:
                if p.grad is None or p.grad.data_ptr() != b['flat'][off:off + n].data_ptr():
                    p.grad = b['flat'][off:off + n].view_as(p)
        self._next = 0
        self.launch_log = []
        self._armed = True

    def finish(self):
        """After backward: reduce buckets whose parameters did not all receive a gradient this step (their
        missing gradients are zeros, identically on every rank) and wait for every exchange."""
        self._armed = False
        while self._next < len(self.buckets):          # same order as the hooks use: 0, 1, 2, ...
            self._launch(self.buckets[self._next])
            self._next += 1
        for b in self.buckets:
            if b['handle'] is not None:
                b['handle'].wait()
                b['handle'] = None

    def broadcast_parameters(self, module, src=0):
        """Rank `src`'s parameters AND buffers to every rank, once, before the first step ([3P] DDP does this at
        construction; `broadcast_buffers=False` only switches off the per-step re-broadcast)."""
        if self.world <= 1:
            return
        with torch.no_grad():
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, src=src, group=self.group)

    def flats(self):
        return [b['flat'] for b in self.buckets]

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def clip_grad_norm_(flats, max_norm, norm_type=2):
    """[3P] torch.nn.utils.clip_grad_norm_ over the flat gradient buckets (padding is zero). Returns the norm
    as a device tensor (no host sync)."""
    if norm_type != 2:
        raise NotImplementedError('only the L2 norm of the shipped configs (norm_type=2) is implemented')
    sq = torch.stack([torch.linalg.vector_norm(f.float(), 2) for f in flats])
    total = torch.linalg.vector_norm(sq, 2)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for f in flats:
        f.mul_(coef.to(f.dtype))
    return total


_NORM_TYPES = (nn.modules.batchnorm._BatchNorm, nn.GroupNorm, nn.LayerNorm)


def build_optimizer(model, cfg):
    """`optimizer = dict(type='AdamW', lr, weight_decay, eps, betas, paramwise_cfg=dict(custom_keys, norm_decay_mult))`
    (configs/instance/coco_b48n17.py:270-285). custom_keys are matched as substrings of the parameter name,
    longest key first ([3P] DefaultOptimizerConstructor); norm_decay_mult applies to norm-layer parameters that
    no custom key matched."""
    cfg = dict(cfg)
    typ = cfg.pop('type')
    pw = dict(cfg.pop('paramwise_cfg', None) or {})
    base_lr = cfg['lr']
    base_wd = cfg.get('weight_decay', 0.0)
    custom = pw.get('custom_keys', {})
    keys = sorted(custom.keys(), key=lambda k: (-len(k), k))
    norm_mult = pw.get('norm_decay_mult', None)
    norm_param_ids = set()
    for m in model.modules():
        if isinstance(m, _NORM_TYPES):
            norm_param_ids.update(id(p) for p in m.parameters(recurse=False))
    groups = {}
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        lr, wd = base_lr, base_wd
        for k in keys:
            if k in name:
                lr = base_lr * custom[k].get('lr_mult', 1.0)
                wd = base_wd * custom[k].get('decay_mult', 1.0)
                break
        else:
            if norm_mult is not None and id(p) in norm_param_ids:
                wd = base_wd * norm_mult
        groups.setdefault((lr, wd), []).append(p)
    param_groups = [dict(params=ps, lr=lr, weight_decay=wd) for (lr, wd), ps in groups.items()]
    if typ != 'AdamW':
        raise NotImplementedError(f'optimizer type {typ!r}: the shipped configs use AdamW')
    return torch.optim.AdamW(param_groups, foreach=True, **cfg)


class LrSchedule:
    """`lr_config` of the shipped configs (configs/instance/coco_b48n17.py:289-297), [3P] mmcv `StepLrUpdaterHook`
    semantics with `by_epoch=False`: before iteration `it` every param group gets
    `initial_lr * gamma ** #{s in step : it >= s}` (not below `min_lr`), and while `it < warmup_iters` that value is
    scaled by the warmup rule ('linear': `1 - (1 - it / warmup_iters) * (1 - warmup_ratio)`, 'constant':
    `warmup_ratio`, 'exp': `warmup_ratio ** (1 - it / warmup_iters)`). Stateless in `it`, so resuming from
    `meta.iter` restores the schedule."""

    def __init__(self, optimizer, cfg=None):
        cfg = dict(cfg or {})
        self.policy = cfg.get('policy', 'fixed')
        if self.policy not in ('step', 'fixed'):
            raise NotImplementedError(f'lr policy {self.policy!r}: the shipped configs use "step"')
        if cfg.get('by_epoch', False):
            raise NotImplementedError('by_epoch=True needs the dataset length (datasets are out of scope); '
                                      'the shipped configs set by_epoch=False')
        step = cfg.get('step', [])
        self.steps = [step] if isinstance(step, int) else list(step)
        self.gamma = cfg.get('gamma', 0.1)
        self.min_lr = cfg.get('min_lr')
        self.warmup = cfg.get('warmup')
        if self.warmup not in (None, 'linear', 'constant', 'exp'):
            raise ValueError(f'warmup {self.warmup!r}')
        self.warmup_iters = cfg.get('warmup_iters', 0)
        self.warmup_ratio = cfg.get('warmup_ratio', 0.1)
        self.optimizer = optimizer
        # base rates are taken from the optimizer AS BUILT from the config (construct the schedule before loading a
        # resumed optimizer state, whose groups carry the already-decayed rates)
        self.initial = [g.get('initial_lr', g['lr']) for g in optimizer.param_groups]

    def lr_at(self, base, it):
        lr = base
        if self.policy == 'step':
            lr = base * self.gamma ** sum(1 for s in self.steps if it >= s)
            if self.min_lr is not None:
                lr = max(lr, self.min_lr)
        if self.warmup is not None and it < self.warmup_iters:
            if self.warmup == 'constant':
                lr = lr * self.warmup_ratio
            elif self.warmup == 'linear':
                lr = lr * (1 - (1 - it / self.warmup_iters) * (1 - self.warmup_ratio))
            else:
                lr = lr * self.warmup_ratio ** (1 - it / self.warmup_iters)
        return lr

    def apply(self, it):
        for g, base in zip(self.optimizer.param_groups, self.initial):
            g['initial_lr'] = base
            g['lr'] = self.lr_at(base, it)


def train_step(model, optimizer, reducer, data, grad_clip=None):
    """One optimisation step of `MaskFormerOpen`; returns the log_vars of `_parse_losses`."""
    reducer.zero_grad()
    out = model.train_step(data)
    out['loss'].backward()
    reducer.finish()
    if grad_clip is not None:
        clip_grad_norm_(reducer.flats(), grad_clip['max_norm'], grad_clip.get('norm_type', 2))
    optimizer.step()
    return out['log_vars']
