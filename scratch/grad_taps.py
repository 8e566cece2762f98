"""Where along the backward chain of the configs[3] training slice does the product's gradient leave the float64 oracle's by more
than the float32 oracle's does?  Taps: the gradient w.r.t. query_pos at every use (cross / self attention of each decoder layer) and
w.r.t. each decoder layer's output, in the float32 oracle (CPU), the float64 oracle (CPU) and the product (GPU).
usage: python scratch/grad_taps.py [Q B tag]      (tag swin | r50)"""
import os, sys, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
import test_fullsize_gpu as T
from cgg_amd import synthetic, query_decoder as QD
from oracle import modules as OM

TAPS = {}


def who(g):
    return 'prod' if g.is_cuda else ('o64' if g.dtype == torch.float64 else 'o32')


def tap(x, name, batch_first):
    if x is None or not torch.is_grad_enabled():
        return x
    y = x.clone() if not x.requires_grad else x + 0

    def hook(g, name=name, bf=batch_first):
        gg = g.detach()
        if not bf:
            gg = gg.transpose(0, 1)
        TAPS.setdefault(who(g), {})[name] = gg.double().cpu()
    if y.requires_grad:
        y.register_hook(hook)
    return y


# ---- product ----
_attend, _self_attend, _ff = QD.MultiheadAttention.attend, QD.MultiheadAttention.self_attend, QD.DetrTransformerDecoderLayer.forward_fast


def attend(self, query, query_pos, kv, bits):
    return _attend(self, query, tap(query_pos, self._tap + '.pos', True), kv, bits)


def self_attend(self, query, query_pos):
    return _self_attend(self, query, tap(query_pos, self._tap + '.pos', True))


def forward_fast(self, query, query_pos, kv, bits):
    return tap(_ff(self, tap(query, self._tap + '.in', True), query_pos, kv, bits), self._tap + '.out', True)


QD.MultiheadAttention.attend, QD.MultiheadAttention.self_attend = attend, self_attend
QD.DetrTransformerDecoderLayer.forward_fast = forward_fast

# ---- oracle ----
_omha, _olayer = OM.MultiheadAttention.forward, OM.DetrTransformerDecoderLayer.forward


def omha(self, query, key=None, value=None, identity=None, query_pos=None, key_pos=None, **kw):
    if key_pos is None and query_pos is not None and key is not None and query_pos.shape == key.shape:
        key_pos = query_pos                     # (self attention: both uses of the table)
    share = key_pos is query_pos
    qp = tap(query_pos, self._tap + '.pos', self.batch_first)
    return _omha(self, query, key, value, identity, query_pos=qp, key_pos=qp if share else key_pos, **kw)


def olayer(self, query, *a, **kw):
    return tap(_olayer(self, tap(query, self._tap + '.in', False), *a, **kw), self._tap + '.out', False)


OM.MultiheadAttention.forward = omha
OM.DetrTransformerDecoderLayer.forward = olayer

_bh = T.build_heads if hasattr(T, 'build_heads') else None
import util
_build = util.build_heads


def build_heads(cfg, seed=0):
    prod, orc = _build(cfg, seed=seed)
    for m in (prod, orc):
        for i, layer in enumerate(m.transformer_decoder.layers):
            layer._tap = f'L{i}'
            layer.attentions[0]._tap = f'L{i}.cross'
            layer.attentions[1]._tap = f'L{i}.self'
    return prod, orc


util.build_heads = build_heads

# ---- assignment log: (who, layer, image) -> (rows, cols) ----
import numpy as np
from oracle import head as OHm
from cgg_amd import ops as OPS
ASSIGN = {'o32': [], 'o64': [], 'prod': []}
_lsa = OHm.linear_sum_assignment


def lsa(cost):
    r, c = _lsa(cost)
    ASSIGN['o64' if cost.dtype == torch.float64 else 'o32'].append((np.asarray(r), np.asarray(c), cost.clone()))
    return r, c


OHm.linear_sum_assignment = lsa
_lsab = OPS.linear_sum_assignment_batch


def lsab(mats):
    out = _lsab(mats)
    for m, (r, c) in zip(mats, out):
        ASSIGN['prod'].append((r.numpy(), c.numpy(), m.clone()))
    return out


OPS.linear_sum_assignment_batch = lsab

Q, B, tag = (int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]) if len(sys.argv) > 3 else (200, 4, 'swin')
ch = (128, 256, 512, 1024) if tag == 'swin' else (256, 512, 1024, 2048)
cfg = T.swin_b_config(Q) if tag == 'swin' else synthetic.model_config(num_things=65, num_stuff=0, num_unknown=17, num_queries=Q, depth=50)
try:
    T._forward_train_slice(torch.device('cuda'), cfg, B, ch, 79, f'taps Q={Q} B={B} {tag}')
except AssertionError as e:
    print('ASSERT', str(e)[:400])

names = sorted(TAPS.get('o64', {}), key=lambda n: (int(n.split('.')[0][1:]), n))
print('%-16s %10s %10s %10s' % ('tap', 'scale', 'prod_err', 'o32_err'))
rows = {}
for n in names:
    t = TAPS['o64'][n]
    sc = t.abs().max().item() or 1e-30
    pe = (TAPS['prod'][n] - t).abs().max().item() / sc if n in TAPS.get('prod', {}) else float('nan')
    oe = (TAPS['o32'][n] - t).abs().max().item() / sc if n in TAPS.get('o32', {}) else float('nan')
    rows[n] = (sc, pe, oe)
    print('%-16s %10.3g %10.3g %10.3g' % (n, sc, pe, oe))
# the sum over the batch of each pos tap (one term of query_embed's gradient), and the running total
tot = {k: 0 for k in ('o64', 'o32', 'prod')}
print('--- batch-summed pos terms: |term| scale, prod err, o32 err (relative to the FINAL query_embed gradient scale)')
final = sum(TAPS['o64'][n].sum(0) for n in names if n.endswith('.pos'))
fs = final.abs().max().item()
for n in names:
    if not n.endswith('.pos'):
        continue
    t = TAPS['o64'][n].sum(0)
    pe = (TAPS['prod'][n].sum(0) - t).abs().max().item() / fs
    oe = (TAPS['o32'][n].sum(0) - t).abs().max().item() / fs
    print('%-16s %10.3g %10.3g %10.3g' % (n, t.abs().max().item() / fs, pe, oe))
pf = sum(TAPS['prod'][n].sum(0) for n in names if n.endswith('.pos'))
of = sum(TAPS['o32'][n].sum(0) for n in names if n.endswith('.pos'))
print('query_embed total: scale %.3g  prod err %.3g  o32 err %.3g' % (fs, (pf - final).abs().max().item() / fs, (of - final).abs().max().item() / fs))
json.dump({k: list(v) for k, v in rows.items()}, open(os.path.join(R, 'gpurun_out', 'grad_taps.json'), 'w'), indent=1)

# ---- assignments: oracle order = layer-major (layer, image); product = image-major (image, layer) ----
nl = 10
Bn = len(ASSIGN['o32']) // nl
print('assignment problems: o32 %d o64 %d prod %d (B=%d)' % (len(ASSIGN['o32']), len(ASSIGN['o64']), len(ASSIGN['prod']), Bn))
for li in range(nl):
    for b in range(Bn):
        r32, c32, m32 = ASSIGN['o32'][li * Bn + b]
        r64, c64, m64 = ASSIGN['o64'][li * Bn + b]
        rp, cp, mp = ASSIGN['prod'][b * nl + li]
        def key(r, c):
            o = np.argsort(c)
            return tuple(r[o].tolist())
        k32, k64, kp = key(r32, c32), key(r64, c64), key(rp, cp)
        if not (k32 == k64 == kp):
            c_opt = float(m64[r64, c64].sum())
            print('layer %d image %d: o32==o64 %s  prod==o32 %s  prod==o64 %s | f64 cost of: o64 %.9g  o32 %.9g  prod %.9g | max |prod cost - o32 cost| %.3g'
                  % (li, b, k32 == k64, kp == k32, kp == k64, c_opt, float(m64[r32, c32].sum()), float(m64[rp, cp].sum()),
                     float((mp.double() - m32.double()).abs().max())))
